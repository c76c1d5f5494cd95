"""Random-shape fuzz of the kernels against the oracle / fp32 torch on the CPU (odd sizes, ragged edges, several launches
back to back so that an out-of-bounds write of one launch can show up in the next).  Prints every mismatch.
usage: python tools/probes/fuzz_kernels.py [seed] [cases]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import m2f_oracle as O
from weed_instance_segmentation_amd import ops

bad = 0


def check(name, out, ref, tol, desc):
    global bad
    err = (out - ref).abs().max().item() if out.numel() else 0.0
    if not err <= tol:
        bad += 1
        print(f"MISMATCH {name} {desc}: max err {err:.3g} (tol {tol})")


def run(seed: int, cases: int) -> int:
    global bad
    bad = 0
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    for i in range(cases):
        _round(rnd, g)
    print(f"fuzz seed {seed}: {cases} rounds, {bad} mismatches")
    return bad


def _round(rnd, g):
    global bad
    if True:
        # ---- K3 (fp32) and the attention-mask build
        B, Q, C = rnd.randint(1, 3), rnd.randint(1, 230), rnd.choice([16, 32, 48, 64, 128, 256])
        H, W = rnd.randint(1, 40), 4 * rnd.randint(1, 12)
        emb, pix = torch.randn(B, Q, C, generator=g), torch.randn(B, C, H, W, generator=g)
        guard = torch.full((4096,), 7.0, device="cuda")  # an allocation right behind: an OOB write lands here or in `after`
        out = ops.mask_einsum(emb.cuda(), pix.cuda())
        after = torch.full((4096,), 7.0, device="cuda")
        check("K3", out.cpu(), torch.einsum("bqc,bchw->bqhw", emb, pix), 2e-3, (B, Q, C, H, W))
        if not (bool((guard == 7).all()) and bool((after == 7).all())):
            bad += 1
            print("GUARD OVERWRITTEN after K3", (B, Q, C, H, W))
        size = (rnd.randint(1, 20), rnd.randint(1, 20))
        m, ro = ops.attn_mask_build(out, size)
        exp = O.attention_mask_from_logits(out.cpu(), size)
        mism = (m.cpu().bool() != exp).float().mean().item()
        if mism > 1e-3:
            bad += 1
            print("MISMATCH attn_mask", (B, Q, H, W, size), mism)
        # ---- K3 bf16
        if C % 32 == 0:
            eb, pb = emb.to(torch.bfloat16), pix.to(torch.bfloat16)
            ob = ops.mask_einsum_bf16(eb.cuda(), pb.cuda(), ops.nchw_to_pixel_major_bf16(pb.cuda()))
            check("K3bf16", ob.cpu(), torch.einsum("bqc,bchw->bqhw", eb.float(), pb.float()), 2e-3, (B, Q, C, H, W))
        # ---- K2
        Hh, D = rnd.choice([1, 2, 4, 8]), rnd.choice([16, 32, 64])
        Q2, N = rnd.randint(1, 230), rnd.randint(1, 600)
        q = torch.randn(B, Q2, Hh * D, generator=g) * 0.4
        k, v = torch.randn(B, N, Hh * D, generator=g), torch.randn(B, N, Hh * D, generator=g)
        mask = torch.rand(B, Q2, N, generator=g) < 0.6
        if rnd.random() < 0.5:
            mask[0, rnd.randrange(Q2)] = True
        sh = lambda t, n: t.view(B, n, Hh, D).permute(0, 2, 1, 3)
        ref = O.masked_attention_core(sh(q, Q2), sh(k, N), sh(v, N), mask).permute(0, 2, 1, 3).reshape(B, Q2, Hh * D)
        o2 = ops.masked_xattn(q.cuda(), k.cuda(), v.cuda(), mask.to(torch.uint8).cuda(), (~mask.all(-1)).to(torch.int32).cuda(), Hh)
        check("K2", o2.cpu(), ref, 2e-4, (B, Hh, D, Q2, N))
        # ---- K1: exact 1:2:4 pyramids (streaming kernel) and arbitrary ones (LDS windows / direct gather)
        if rnd.random() < 0.5:
            h0, w0 = rnd.randint(1, 10), rnd.randint(1, 10)
            shapes = [(h0, w0), (2 * h0, 2 * w0), (4 * h0, 4 * w0)]
        else:
            shapes = [(rnd.randint(1, 12), rnd.randint(1, 12)) for _ in range(rnd.randint(1, 4))]
        S = sum(a * b for a, b in shapes)
        L = len(shapes)
        value = torch.randn(B, S, 8, 32, generator=g)
        loc = torch.rand(B, S, 8, L, 4, 2, generator=g) * 1.3 - 0.15
        aw = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4)
        o1 = ops.ms_deform_attn(value.cuda(), shapes, loc.cuda(), aw.cuda())
        check("K1", o1.cpu(), O.msdeform_attn_core(value, shapes, loc, aw), 5e-5, (B, shapes))


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40) else 0)
