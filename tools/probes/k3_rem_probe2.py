"""Follow-up to k3_rem_probe.py: what do the corrupted main-tile rows contain?  (WM2F_K3_DBG=4 forces the VALU remainder
path at C = 64.)"""
import os, sys, torch
os.environ["WM2F_K3_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops
g = torch.Generator().manual_seed(0)
shown = 0
for trial in range(12):
    B, Q, H, W = 1, 100, 32, 48
    emb = torch.randn(B, Q, 64, generator=g); pix = torch.randn(B, 64, H, W, generator=g)
    out = ops.mask_einsum(emb.cuda(), pix.cuda()).cpu().view(Q, H * W)
    ref = torch.einsum("bqc,bchw->bqhw", emb, pix).view(Q, H * W)
    err = (out - ref).abs()
    rows = (err.amax(1) > 1e-3).nonzero().flatten().tolist()
    if not rows:
        continue
    shown += 1
    for r in rows[:4]:
        cols = (err[r] > 1e-3).nonzero().flatten()
        c0, c1 = int(cols.min()), int(cols.max())
        seg = out[r, cols[:4]]
        # does the wrong data equal some other row of the reference at the same pixels?
        match = [(q2, float((ref[q2, cols] - out[r, cols]).abs().max())) for q2 in range(Q)]
        best = min(match, key=lambda t: t[1])
        print(f"trial {trial} row {r}: {len(cols)} wrong pixels in [{c0}, {c1}], values {seg.tolist()}, closest reference row {best[0]} (max diff {best[1]:.3g}); zeros: {bool((out[r, cols] == 0).all())}")
    if shown >= 3:
        break
print("cases shown", shown)
