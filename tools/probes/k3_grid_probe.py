"""K3 fp32 against torch.einsum over a grid of (B, Q, C, H, W): isolates shape-dependent errors."""
import itertools, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops
g = torch.Generator().manual_seed(0)
for B, Q, C, (H, W) in itertools.product((1, 2), (100, 200, 84, 112), (64, 128, 256), ((24, 32), (32, 48), (3, 12))):
    emb = torch.randn(B, Q, C, generator=g)
    pix = torch.randn(B, C, H, W, generator=g)
    out = ops.mask_einsum(emb.cuda(), pix.cuda()).cpu()
    ref = torch.einsum("bqc,bchw->bqhw", emb, pix)
    err = (out - ref).abs()
    if err.max() > 1e-3:
        bad_q = (err.amax(dim=(0, 2, 3)) > 1e-3).nonzero().flatten().tolist()
        bad_b = (err.amax(dim=(1, 2, 3)) > 1e-3).nonzero().flatten().tolist()
        print(f"BAD B={B} Q={Q} C={C} HW={H}x{W}: max err {err.max():.3g}; bad queries {bad_q[:6]}..{bad_q[-3:]} ({len(bad_q)}), bad images {bad_b}")
print("done")
