"""Where a fp32 train step (config 2 shapes, B images) spends its time: forward without loss, matcher + loss forward,
backward, optimizer -- wall clock with a device sync after each part."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = bench.build_model().to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-5)
    x = torch.randn(a.B, 3, 1024, 1024, device=dev)
    ml, cl = bench.synthetic_labels(a.B, 1024, 1024, device=dev)
    acc = {}

    def part(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r

    for it in range(a.iters + 1):
        opt.zero_grad(set_to_none=True)
        out = part("forward (no labels)", lambda: model(pixel_values=x, output_auxiliary_logits=True))
        del out
        out = part("forward + matcher + loss", lambda: model(pixel_values=x, mask_labels=ml, class_labels=cl))
        part("backward", lambda: out.loss.backward())
        part("optimizer", lambda: opt.step())
    res = {k: sum(v[1:]) / len(v[1:]) for k, v in acc.items()}
    res["loss part (difference)"] = res["forward + matcher + loss"] - res["forward (no labels)"]
    print(json.dumps({"B": a.B, "ms": res}))


if __name__ == "__main__":
    main()
