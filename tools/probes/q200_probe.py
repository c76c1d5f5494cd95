import json, sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_golden
from oracle import m2f_oracle as O
from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation, ops
g = load_golden("full_tiny.npz")
cd = json.loads(str(g["config_json"]))
for Q in (10, 100, 112, 113, 200):
    cd["num_queries"] = Q
    cfg = Mask2FormerConfig.from_dict(cd)
    torch.manual_seed(11)
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(12))
    res = O.forward(sd, cfg.to_dict(), x)
    model = model.cuda()
    with torch.no_grad():
        out = model(pixel_values=x.cuda(), output_auxiliary_logits=True)
    errs = []
    for i, a in enumerate(out.auxiliary_logits):
        r = res["aux_masks"][i]
        errs.append(float((a["masks_queries_logits"].cpu() - r).abs().max() / r.abs().max()))
    ref = res["masks_queries_logits"]
    e = (out.masks_queries_logits.cpu() - ref).abs()
    perq = e.amax(dim=(0, 2, 3))
    print("Q", Q, "aux rel errs", ["%.1e" % v for v in errs], "final %.2e" % float(e.max() / ref.abs().max()), "queries over 1e-4:", int((perq / ref.abs().max() > 1e-4).sum()))
# kernel-level K2/K3 at Q=200
gen = torch.Generator().manual_seed(1)
B, H, D, N = 2, 8, 32, 192
for Q in (112, 200):
    q = torch.randn(B, Q, H*D, generator=gen) * 0.3; k = torch.randn(B, N, H*D, generator=gen); v = torch.randn(B, N, H*D, generator=gen)
    mask = torch.rand(B, Q, N, generator=gen) < 0.6
    ro = (~mask.all(-1)).to(torch.int32)
    sh = lambda t, n: t.view(B, n, H, D).permute(0, 2, 1, 3)
    ref = O.masked_attention_core(sh(q, Q), sh(k, N), sh(v, N), mask).permute(0, 2, 1, 3).reshape(B, Q, H*D)
    out = ops.masked_xattn(q.cuda(), k.cuda(), v.cuda(), mask.to(torch.uint8).cuda(), ro.cuda(), H)
    print("K2 Q", Q, float((out.cpu() - ref).abs().max()))
    emb = torch.randn(B, Q, 64, generator=gen); pix = torch.randn(B, 64, 24, 32, generator=gen)
    print("K3 Q", Q, float((ops.mask_einsum(emb.cuda(), pix.cuda()).cpu() - torch.einsum("bqc,bchw->bqhw", emb, pix)).abs().max()))
