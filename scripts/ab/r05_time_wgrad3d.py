"""conv3d weight gradient (cds_conv3d_wgrad_f32) at the CostRegNet shapes of the config-5 training step."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import train_ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = []
for (C, D, h, w) in ((32, 48, 72, 96), (16, 32, 144, 192), (8, 8, 288, 384)):
    shapes += [("conv0", 8, C, D, h, w, 1), ("conv1", 16, 8, D, h, w, 2), ("conv2", 16, 16, D // 2, h // 2, w // 2, 1),
               ("conv3", 32, 16, D // 2, h // 2, w // 2, 2), ("conv4", 32, 32, D // 4, h // 4, w // 4, 1)]
tot = 0.0
for name, Ca, Cb, D, h, w, S in shapes:
    x = torch.randn(1, Cb, D, h, w, device=dev)
    g = torch.randn(1, Ca, D // S, h // S, w // S, device=dev)
    ts = []
    for i in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); dw = train_ops.conv3d_wgrad(g, x, S); b.record(); b.synchronize()
        if i >= 2: ts.append(a.elapsed_time(b) * 1e3)
    t = statistics.median(ts); tot += t
    print(f"{name} {Cb:2d}->{Ca:2d} s{S} {D}x{h}x{w}: {t:7.1f} us  ({2 * 27 * Ca * Cb * g[0, 0].numel() / t / 1e6:6.1f} TFLOP/s)  sum {float(dw.double().sum()):.5e}")
print(f"total {tot:.0f} us")
