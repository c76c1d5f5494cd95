"""Random-shape fuzz of the device-side instance post-processing against the oracle (odd logit and target sizes, few /
many kept instances, binary maps).  usage: python tools/probes/fuzz_postprocess.py [seed] [cases]"""
import os, random, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import m2f_oracle as O
from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor


def run(seed, cases):
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    proc = Mask2FormerInstancePostProcessor()
    bad = 0
    for _ in range(cases):
        B, Q, C = rnd.randint(1, 3), rnd.randint(1, 40), rnd.randint(1, 5)
        h, w = rnd.randint(3, 70), rnd.randint(3, 70)
        low = torch.randn(B, Q, 3, 4, generator=g) * 4 - rnd.random() * 3
        masks = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)
        cls = torch.randn(B, Q, C + 1, generator=g) * rnd.choice([1.0, 4.0])
        ts = None if rnd.random() < 0.2 else [(rnd.randint(1, 500), rnd.randint(1, 500)) for _ in range(B)]
        maps = rnd.random() < 0.3
        ref = O.post_process_instance_segmentation(cls, masks, 0.5, ts, return_binary_maps=maps)
        out = SimpleNamespace(class_queries_logits=cls.cuda(), masks_queries_logits=masks.cuda())
        res = proc.post_process_instance_segmentation(out, threshold=0.5, target_sizes=ts, return_binary_maps=maps)
        for i, (r, q) in enumerate(zip(res, ref)):
            a, b = r["segments_info"], q["segments_info"]
            ok = len(a) == len(b) and all(x["label_id"] == y["label_id"] and abs(x["score"] - y["score"]) < 3e-6 for x, y in zip(a, b))
            sa, sb = r["segmentation"].cpu().float(), q["segmentation"].float()
            ok = ok and sa.shape == sb.shape and (sa != sb).float().mean().item() <= 1e-3
            if not ok:
                bad += 1
                print("MISMATCH", dict(B=B, Q=Q, C=C, h=h, w=w, ts=ts, maps=maps, image=i, kept=(len(a), len(b)), shapes=(tuple(sa.shape), tuple(sb.shape))))
    print(f"fuzz-postprocess seed {seed}: {cases} cases, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 25) else 0)
