"""Random-shape fuzz of K5 (soft-argmin + confidence) and K6 (depth hypotheses) against the CPU oracle.
Usage: fuzz_regress.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import ops
from oracle import cds_oracle as O
dev = torch.device("cuda:0")


def case_k5(rng):
    D = rng.choice([1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 48, 96, 192, 200]); h = rng.randint(1, 40); w = rng.randint(1, 90)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    sharp = rng.choice([0.5, 3.0, 20.0])
    pre = torch.randn(1, D, h, w, generator=g) * sharp
    if rng.random() < 0.3:   # peaks at the ends of the depth range (window clipped by the zero padding)
        pre[:, rng.choice([0, D - 1])] += 30.0
    per_pixel = rng.random() < 0.7
    hyp = (torch.linspace(425, 900, D).view(1, D, 1, 1) + 3 * torch.rand(1, D, h, w, generator=g)) if per_pixel \
        else torch.linspace(425, 900, D).view(1, D)
    prob, depth, conf = O.softargmin(pre, hyp)
    d, c, p = ops.softargmin_conf(pre[0].to(dev), (hyp[0] if per_pixel else hyp[0]).to(dev).contiguous(), want_prob=True)
    # the confidence gathers at trunc(sum p*idx): a pixel whose index lands within rounding of an integer may pick the
    # neighbouring window -> compare where the oracle's index is not within 1e-4 of an integer
    idxf = (prob * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).sum(1)[0]
    safe = ((idxf - idxf.round()).abs() > 1e-4) | (idxf.round() > idxf)
    e_d = float((d.cpu() - depth[0]).abs().max()); e_p = float((p.cpu() - prob[0]).abs().max())
    e_c = float(((c.cpu() - conf[0]).abs() * safe).max())
    return f"K5 D={D:3d} h={h:2d} w={w:2d} sharp={sharp:4.1f} pp={int(per_pixel)}", max(e_d / 900.0 * 1e1, e_p, e_c), (e_d, e_p, e_c)


def case_k6(rng):
    scale = rng.choice([1, 2, 4]); D = rng.choice([8, 32, 48, 5]); H = 4 * rng.randint(1, 24); W = 4 * rng.randint(1, 30)
    prev_scale = rng.choice([1, 2, 4])
    hp, wp = H // prev_scale, W // prev_scale
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    prev = 425 + 475 * torch.rand(1, hp, wp, generator=g)
    interval = rng.choice([0.75, 1.5, 4.0]) * 2.5
    dmin, dmax = 425.0, 902.5
    want = O.stage_hypotheses(prev, D, torch.tensor(interval).view(1, 1, 1), torch.tensor(dmin).view(1, 1, 1),
                              torch.tensor(dmax).view(1, 1, 1), H, W, scale)[0]
    got = ops.depth_hypotheses(prev[0].to(dev).contiguous(), D, H, W, scale, interval, dmin, dmax).cpu()
    e = float((got - want).abs().max())
    return f"K6 D={D:2d} H={H:3d} W={W:3d} scale={scale} prev=1/{prev_scale} interval={interval}", e / 900.0 * 1e1, (e,)


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    worst = {}
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
        name, score, errs = (case_k5 if i % 2 == 0 else case_k6)(rng)
        k = name[:2]
        worst[k] = tuple(max(a, b) for a, b in zip(worst.get(k, errs), errs))
        flag = "" if score < 2e-5 else "  <-- CHECK"
        print(f"{name}: " + " ".join(f"{e:.2e}" for e in errs) + flag)
    print("worst (K5: depth, prob, conf | K6: hypotheses):", worst)
