"""Where does a run-to-run difference behind the backbone come from?  The configs[1] forward is run N times on IDENTICAL backbone
features (cached from one call); forward hooks keep every sub-module's output of every run, and the first module (in execution
order) whose output differs from run 0 is reported per run -- together with whether its INPUTS were still identical, which separates
"this op is not reproducible" from "it inherited the difference".  Usage: python tools/probes/determinism_probe.py [runs]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
model = bench.build_model().to(dev).eval()
x = torch.randn(8, 3, 1024, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
plm = model.model.pixel_level_module
with torch.no_grad():
    for _ in range(2):
        model(pixel_values=x)
    feats = [t.clone() for t in plm.encoder(x)]


class Cached(torch.nn.Module):
    channels = getattr(plm.encoder, "channels", None)

    def forward(self, _x):
        return [t.clone() for t in feats]


plm.encoder = Cached()


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for e in o for t in flat(e)]
    return []


records = []
order = []


def hook(name):
    def fn(mod, inp, out):
        cur[name] = ([t.detach().clone() for t in flat(inp)], [t.detach().clone() for t in flat(out)])
        if name not in order:
            order.append(name)
    return fn


leaf_like = (torch.nn.Conv2d, torch.nn.Linear, torch.nn.GroupNorm, torch.nn.LayerNorm)
for name, mod in model.named_modules():
    if name and (isinstance(mod, leaf_like) or type(mod).__name__ in ("MSDeformAttn", "MaskedCrossAttention", "SelfAttention", "PixelDecoderEncoderLayer",
                                                                     "MaskedAttentionDecoderLayer", "Mask2FormerPixelDecoder")):
        mod.register_forward_hook(hook(name))
eq = lambda a, b: len(a) == len(b) and all(p.shape == q.shape and torch.equal(p, q) for p, q in zip(a, b))
ref = None
n_same = 0
for r in range(runs):
    cur = {}
    with torch.no_grad():
        out = model(pixel_values=x)
    cur["__out__"] = ([], [out.masks_queries_logits.clone(), out.class_queries_logits.clone()])
    if ref is None:
        ref = cur
        order.append("__out__")
        continue
    first = None
    for name in order:
        if name in ref and name in cur and not eq(ref[name][1], cur[name][1]):
            first = name
            break
    if first is None:
        n_same += 1
        continue
    ins_same = eq(ref[first][0], cur[first][0])
    o0, o1 = ref[first][1], cur[first][1]
    nd = [int((p != q).sum()) for p, q in zip(o0, o1)]
    md = [float((p.float() - q.float()).abs().max()) for p, q in zip(o0, o1)]
    print(json.dumps({"run": r, "first_differing_module": first, "type": type(dict(model.named_modules()).get(first, model)).__name__,
                      "its_inputs_identical": ins_same, "elements_differing": nd, "max_abs_diff": md}), flush=True)
    del cur
print(json.dumps({"runs": runs, "identical_to_run_0": n_same}), flush=True)
