"""Race hunt: repeat the 2D training ops on fixed inputs and compare every repetition with the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cds_mvsnet_amd import train2d_ops as t2
dev = torch.device("cuda:0")
cases = [(3, 2, 8, 16, 24, 40), (3, 2, 16, 32, 22, 36), (3, 1, 8, 16, 21, 40), (7, 1, 8, 11, 23, 52), (11, 1, 3, 11, 26, 72), (1, 1, 24, 8, 20, 36), (5, 1, 8, 11, 19, 36), (3, 1, 16, 16, 18, 44)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for (k, s, cin, cout, h, w) in cases:
    torch.manual_seed(k * 100 + cin)
    x = torch.randn(3, cin, h, w, dtype=torch.float64); wt = torch.randn(cout, cin, k, k, dtype=torch.float64) * 0.2
    pad = (k - 1) // 2
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, pad); g = torch.randn_like(yr); yr.backward(g)
    worst = [0.0, 0.0, 0.0]; bad = 0
    xd, wd, gd = x.float().to(dev), wt.float().to(dev), g.float().to(dev)
    for r in range(reps):
        xg, wg = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        y = t2.Conv2d.apply(xg, wg, None, s, pad); y.backward(gd)
        errs = [(y.detach().double().cpu() - yr.detach()).abs().max().item() / yr.abs().max().item(),
                (xg.grad.double().cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item(),
                (wg.grad.double().cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()]
        if max(errs) > 5e-4:
            bad += 1
            if bad <= 3: print("   BAD rep", r, errs)
        worst = [max(a, b) for a, b in zip(worst, errs)]
    print(f"k={k} s={s} {cin}->{cout} {h}x{w}: worst rel err y/dx/dw = {worst[0]:.2e} {worst[1]:.2e} {worst[2]:.2e}, bad {bad}/{reps}")
