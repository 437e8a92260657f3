"""conv2d weight gradient (cds_conv2d_wgrad_f32) at FeatureNet shapes of the config-5 training step (8 stacked images of 384x288)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import train2d_ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
tot = 0.0
for name, N, Co, Cin, H, W, k in (("conv00 k11", 8, 11, 3, 288, 384, 11), ("conv00 k7", 8, 11, 3, 288, 384, 7), ("conv01 k7", 8, 11, 8, 288, 384, 7),
                                  ("conv01 k5", 8, 11, 8, 288, 384, 5), ("conv01 k3", 8, 11, 8, 288, 384, 3), ("conv10 k5", 8, 19, 16, 144, 192, 5),
                                  ("conv10 k3", 8, 19, 16, 144, 192, 3), ("conv20 k3", 8, 35, 32, 72, 96, 3), ("out3 k1", 8, 11, 8, 288, 384, 1)):
    x = torch.randn(N, Cin, H, W, device=dev)
    g = torch.randn(N, Co, H, W, device=dev)
    ts = []
    for i in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); dw = train2d_ops.conv2d_wgrad(g, x, k, 1, (k - 1) // 2); b.record(); b.synchronize()
        if i >= 2: ts.append(a.elapsed_time(b) * 1e3)
    t = statistics.median(ts); tot += t
    print(f"{name:10s} {Cin:2d}->{Co:2d} {H}x{W} x{N}: {t:7.1f} us  ({2 * k * k * Co * Cin * N * H * W / t / 1e6:6.1f} TFLOP/s)  sum {float(dw.double().sum()):.5e}")
print(f"total {tot:.0f} us")
