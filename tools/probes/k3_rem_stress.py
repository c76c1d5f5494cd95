"""Stress of K3's VALU remainder path at the channel counts it is enabled for (C >= 128): many random inputs at the
small shapes where C = 64 showed dropped main-tile stores."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops
g = torch.Generator().manual_seed(1)
bad = n = 0
for trial in range(40):
    for B, Q, C, (H, W) in ((1, 100, 256, (32, 48)), (2, 100, 128, (32, 48)), (1, 200, 256, (24, 32)), (2, 84, 128, (24, 32)), (1, 100, 256, (64, 64))):
        emb = torch.randn(B, Q, C, generator=g); pix = torch.randn(B, C, H, W, generator=g)
        out = ops.mask_einsum(emb.cuda(), pix.cuda()).cpu()
        ref = torch.einsum("bqc,bchw->bqhw", emb, pix)
        n += 1
        bad += int((out - ref).abs().max() > 2e-3)
print(f"bad {bad} of {n}")
