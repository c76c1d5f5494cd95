"""Random-shape fuzz of the backward kernels (K1, K2, point samplers, mask-loss rows) and of K4 against the oracle's
autograd on the CPU.  usage: python tools/probes/fuzz_backward.py [seed] [cases]"""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import m2f_oracle as O
from weed_instance_segmentation_amd import ops

bad = 0


def check(name, out, ref, tol, desc):
    global bad
    scale = max(ref.abs().max().item(), 1e-6)
    err = (out - ref).abs().max().item() / scale
    if not err <= tol:
        bad += 1
        print(f"MISMATCH {name} {desc}: rel err {err:.3g} (tol {tol})")


def run(seed, cases):
    global bad
    bad = 0
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    for _ in range(cases):
        B = rnd.randint(1, 2)
        # ---- K1 forward + backward (1:2:4 -> LDS-window backward; arbitrary -> global-atomic backward)
        if rnd.random() < 0.6:
            h0, w0 = rnd.randint(1, 7), rnd.randint(1, 7)
            shapes = [(h0, w0), (2 * h0, 2 * w0), (4 * h0, 4 * w0)]
        else:
            shapes = [(rnd.randint(1, 9), rnd.randint(1, 9)) for _ in range(rnd.randint(1, 3))]
        S, L = sum(a * b for a, b in shapes), len(shapes)
        value = torch.randn(B, S, 8, 32, generator=g)
        loc = torch.rand(B, S, 8, L, 4, 2, generator=g) * 1.2 - 0.1
        aw = torch.softmax(torch.randn(B, S, 8, L * 4, generator=g), -1).view(B, S, 8, L, 4)
        go = torch.randn(B, S, 256, generator=g)
        vr, lr, wr = (t.clone().requires_grad_(True) for t in (value, loc, aw))
        O.msdeform_attn_core_explicit(vr, shapes, lr, wr).backward(go)
        vg, lg, wg = (t.cuda().requires_grad_(True) for t in (value, loc, aw))
        ops.ms_deform_attn(vg, shapes, lg, wg).backward(go.cuda())
        check("K1 dvalue", vg.grad.cpu(), vr.grad, 2e-4, (B, shapes))
        check("K1 dloc", lg.grad.cpu(), lr.grad, 2e-3, (B, shapes))
        check("K1 dw", wg.grad.cpu(), wr.grad, 2e-4, (B, shapes))
        # ---- K2 backward
        Hh, D, Q, N = rnd.choice([1, 2, 8]), rnd.choice([16, 32, 64]), rnd.randint(1, 150), rnd.randint(1, 300)
        q = torch.randn(B, Q, Hh * D, generator=g) * 0.4
        k, v = torch.randn(B, N, Hh * D, generator=g), torch.randn(B, N, Hh * D, generator=g)
        mask = torch.rand(B, Q, N, generator=g) < 0.5
        mask[0, rnd.randrange(Q)] = True
        go2 = torch.randn(B, Q, Hh * D, generator=g)
        sh = lambda t, n: t.view(B, n, Hh, D).permute(0, 2, 1, 3)
        qr, kr, vr2 = (t.clone().requires_grad_(True) for t in (q, k, v))
        O.masked_attention_core(sh(qr, Q), sh(kr, N), sh(vr2, N), mask).permute(0, 2, 1, 3).reshape(B, Q, Hh * D).backward(go2)
        qg, kg, vg2 = (t.cuda().requires_grad_(True) for t in (q, k, v))
        ops.masked_xattn(qg, kg, vg2, mask.to(torch.uint8).cuda(), (~mask.all(-1)).to(torch.int32).cuda(), Hh).backward(go2.cuda())
        for nme, a, b2 in (("dq", qg, qr), ("dk", kg, kr), ("dv", vg2, vr2)):
            check("K2 " + nme, a.grad.cpu(), b2.grad, 5e-4, (B, Hh, D, Q, N))
        # ---- point samplers over levels + mask-loss rows
        NL, M, P, Nm, H, W = rnd.randint(1, 4), rnd.randint(1, 9), rnd.randint(1, 300), rnd.randint(1, 12), rnd.randint(1, 20), rnd.randint(1, 20)
        maps = [torch.randn(Nm, H, W, generator=g) for _ in range(NL)]
        pts = torch.rand(NL, M, P, 2, generator=g) * 1.2 - 0.1
        idx = torch.randint(0, Nm, (NL, M), generator=g).to(torch.int32)
        labels = (torch.rand(NL * M, P, generator=g) < 0.4).float()
        mr = [m.clone().requires_grad_(True) for m in maps]
        ref_rows = torch.stack([O.sample_point(mr[l][idx[l].long()][:, None], pts[l])[:, 0] for l in range(NL)])
        x = ref_rows.reshape(NL * M, P)
        ref_bce = F.binary_cross_entropy_with_logits(x, labels, reduction="none").mean(1)
        pr = x.sigmoid()
        ref_dice = 1 - (2 * (pr * labels).sum(1) + 1) / (pr.sum(1) + labels.sum(1) + 1)
        (ref_bce.sum() * 0.7 + ref_dice.sum() * 1.3).backward()
        mg = [m.cuda().requires_grad_(True) for m in maps]
        rows = ops.point_sample_levels(mg, pts.cuda(), idx.cuda())
        check("sample_levels", rows.detach().cpu(), ref_rows.detach(), 2e-5, (NL, M, P, Nm, H, W))
        bce, dice = ops.mask_loss_rows(rows.view(NL * M, P), labels.cuda())
        check("bce rows", bce.detach().cpu(), ref_bce.detach(), 2e-5, (NL, M, P))
        check("dice rows", dice.detach().cpu(), ref_dice.detach(), 2e-5, (NL, M, P))
        (bce.sum() * 0.7 + dice.sum() * 1.3).backward()
        for l in range(NL):
            check("dmaps", mg[l].grad.cpu(), mr[l].grad, 2e-4, (NL, M, P, Nm, H, W))
        # ---- K4
        Qm, T_, Pn, hh, ww = rnd.randint(1, 40), rnd.randint(1, 9), rnd.randint(1, 200), rnd.randint(1, 16), rnd.randint(1, 16)
        ml = torch.randn(1, 1, Qm, hh, ww, generator=g) * 2
        cl = torch.randn(1, 1, Qm, 4, generator=g)
        tg = (torch.rand(T_, 2 * hh, 3 * ww, generator=g) < 0.4).float()
        tc = torch.randint(0, 3, (T_,), generator=g)
        pp = torch.rand(1, 1, Pn, 2, generator=g)
        cost = ops.matcher_cost(ml.cuda(), cl.cuda(), tg.cuda(), [T_], tc.cuda(), pp.cuda(), 2.0, 5.0, 5.0).cpu()
        check("K4", cost[0, 0, :, :T_], O.matcher_cost(ml[0, 0], cl[0, 0], tg, tc, pp[0], 2.0, 5.0, 5.0), 1e-4, (Qm, T_, Pn, hh, ww))
    print(f"fuzz-backward seed {seed}: {cases} rounds, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20) else 0)
