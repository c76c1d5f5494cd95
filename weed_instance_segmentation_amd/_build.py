"""In-tree build of libwm2f.so (hand-written HIP kernels, gfx950 only) with hipcc.

`python -m weed_instance_segmentation_amd._build` or `__graft_entry__.build()`.
hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to
the GPU box with the tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libwm2f.so")
# The profiling build (same sources, -DWM2F_PROFILING): K1 timing ablations and stamped kernels, the stamp buffer,
# environment knobs of K2 / K3 (include/wm2f_prof.h).  tools/ load it; the product never does.
OBJ_PROF = os.path.join(CSRC, "build_prof")
LIB_PROF = os.path.join(HERE, "libwm2f_prof.so")
SOURCES = ["api.hip", "msdeform.hip", "msdeform_tiled.hip", "msdeform_quad.hip", "msdeform_tiled_bwd.hip", "mask_einsum.hip", "mask_einsum_bf16.hip", "token_gemm.hip", "token_wgrad.hip", "layernorm_train.hip", "attn_mask.hip", "masked_xattn.hip", "matcher.hip", "lsa.hip", "fused_elementwise.hip", "postprocess.hip", "mask_loss.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "msdeform_tiled.h"), os.path.join(os.path.dirname(HERE), "include", "wm2f.h"),
           os.path.join(os.path.dirname(HERE), "include", "wm2f_prof.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]
# per-file flags.  masked_xattn: keep MFMA results in VGPRs -- the softmax between the two products reads S with VALU
# instructions, and the default AGPR form cost 40 v_accvgpr moves per key tile and a register tier (120 -> 108).
EXTRA_FLAGS = {"masked_xattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, prof: bool = False) -> str:
    """Compile every kernel source for gfx950 and link libwm2f.so (prof=True: libwm2f_prof.so, the profiling build).
    Returns the library path."""
    hipcc = _hipcc()
    obj_dir, lib, defs = (OBJ_PROF, LIB_PROF, ["-DWM2F_PROFILING"]) if prof else (OBJ, LIB, [])
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s, os.path.abspath(__file__)] + HEADERS):
            jobs.append([hipcc, *FLAGS, *defs, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and (r.stdout or r.stderr):
            print(r.stdout + r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--prof" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, prof=True))
