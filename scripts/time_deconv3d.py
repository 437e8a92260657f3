"""Time one cds_deconv3d_k3s2 launch shape with HIP events.  Usage: time_deconv3d.py Cin Cout D H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops
Cin, Cout, D, H, W = (int(a) for a in sys.argv[1:6])
dev = torch.device("cuda:0")
x = torch.randn(Cin, D, H, W, device=dev)
w = torch.randn(Cin, 27, Cout, device=dev) * 0.1
b = torch.randn(Cout, device=dev)
skip = torch.randn(Cout, 2 * D, 2 * H, 2 * W, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
t = timeit(lambda: ops.deconv3d_k3s2(x, w, b, relu=True, skip=skip))
fl = 2.0 * x[0].numel() * Cin * 27 * Cout
by = 4.0 * (x.numel() + 2 * skip.numel())
print(f"{os.environ.get('TAG','')} deconv3d {Cin}->{Cout} {W}x{H}x{D}: {t*1e3:.0f} us  {fl/t/1e9:.1f} TF  {by/t/1e9:.2f} TB/s (compulsory bytes)")
