"""Which part of K3's VALU remainder path corrupts main-tile rows at C = 64?  (WM2F_K3_DBG: 4 forces the path, +1 skips
the remainder FMAs, +2 skips the remainder epilogue.)  Only main-tile rows (q % 100 < 96) are checked."""
import os, subprocess, sys
if len(sys.argv) == 1:
    for dbg in (4,):
        r = subprocess.run([sys.executable, __file__, str(dbg)], capture_output=True, text=True, env=dict(os.environ, WM2F_K3_DBG=str(dbg)))
        print("dbg", dbg, r.stdout.strip().replace("\n", " | ")[-300:], r.stderr.strip()[-200:] if r.returncode else "")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops
g = torch.Generator().manual_seed(0)
bad = 0
for trial in range(6):
    for B, Q, (H, W) in ((1, 100, (32, 48)), (2, 100, (32, 48)), (1, 200, (24, 32)), (2, 84, (24, 32))):
        emb = torch.randn(B, Q, 64, generator=g); pix = torch.randn(B, 64, H, W, generator=g)
        out = ops.mask_einsum(emb.cuda(), pix.cuda()).cpu()
        ref = torch.einsum("bqc,bchw->bqhw", emb, pix)
        rows = torch.arange(Q)
        per = Q // ((Q + 111) // 112)
        main = (rows % per) < (per // 16) * 16
        err = (out - ref).abs()[:, main]
        bad += int(err.max() > 1e-3)
print("bad cases (main rows):", bad, "of 24")
