"""Is the loss of gradient DIRECTION inside the ResNet-50 backbone under bf16 autocast (tests/test_configs_gpu.py, config 2:
energy-weighted cosine 0.11 between the bf16 and the fp32 step's backbone gradients, 0.9999 behind the backbone) this package's doing?
The same measurement on the dependency's own eager model (transformers on the GPU box -- a tool, never the product or the tests):
same random-init weights, same batch, the global RNG re-seeded before each step so that both draw the same sampling points.
Usage: python tools/probes/hf_bf16_backbone_grad.py [B]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformers import Mask2FormerConfig, Mask2FormerForUniversalSegmentation, ResNetConfig
from bench import synthetic_labels

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = Mask2FormerConfig(backbone_config=ResNetConfig(out_features=["stage1", "stage2", "stage3", "stage4"]), num_labels=3, num_queries=100)
model = Mask2FormerForUniversalSegmentation(cfg).to(dev).train()
x = torch.randn(B, 3, 1024, 1024, device=dev)
ml, cl = synthetic_labels(B, 1024, 1024, seed=0, device=dev)
grads = {}
for amp in (False, True):
    model.zero_grad(set_to_none=True)
    torch.manual_seed(123)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
    out.loss.backward()
    grads[amp] = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    print(json.dumps({"amp": amp, "loss": float(out.loss)}), flush=True)
for label, sel in (("backbone", lambda n: n.startswith("model.pixel_level_module.encoder.")), ("everything behind it", lambda n: not n.startswith("model.pixel_level_module.encoder."))):
    cs, ws, rs = [], [], []
    for n in grads[False]:
        if not sel(n):
            continue
        a, b = grads[False][n].flatten().double(), grads[True][n].flatten().double()
        na, nb = float(a.norm()), float(b.norm())
        if na == 0:
            continue
        cs.append(float(a @ b) / max(na * nb, 1e-300)); ws.append(na); rs.append(nb / na)
    cs, ws, rs = np.array(cs), np.array(ws), np.array(rs)
    e = ws ** 2 / (ws ** 2).sum()
    print(json.dumps({"model": "transformers eager", "B": B, "group": label, "tensors": len(cs), "cosine_median": round(float(np.median(cs)), 4),
                      "cosine_energy_weighted": round(float((e * cs).sum()), 4), "norm_ratio_energy_weighted": round(float((e * rs).sum()), 4)}), flush=True)
