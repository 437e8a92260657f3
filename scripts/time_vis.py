"""Time the matrix-core 16 -> 16 3x3 layer (visibility CNN) at a given size.  Usage: time_vis.py N H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops
N, H, W = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(N, 16, H, W, device=dev); wcl = torch.randn(9, 16, 16, device=dev) * 0.1; b = torch.randn(16, device=dev)
hw_, hb_ = torch.randn(16, device=dev), torch.randn(1, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
t0 = timeit(lambda: ops.conv2d_k3_c16(x, wcl, b, 1))
t1 = timeit(lambda: ops.conv2d_k3_c16(x, wcl, b, 1, head_w=hw_, head_b=hb_))
fl = 2.0 * N * H * W * 16 * 16 * 9
print(f"{os.environ.get('TAG','')} vis layer N={N} {W}x{H}: plain {t0*1e3:.0f} us ({fl/t0/1e9:.0f} TF, {8.0*x.numel()/t0/1e9:.2f} TB/s)  with head {t1*1e3:.0f} us")
