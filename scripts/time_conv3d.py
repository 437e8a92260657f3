"""Time one cds_conv3d_k3 launch shape with HIP events.  Usage: time_conv3d.py Cin Cout D H W stride"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops
Cin, Cout, D, H, W, stride = (int(a) for a in sys.argv[1:7])
dev = torch.device("cuda:0")
x = torch.randn(Cin, D, H, W, device=dev)
w = torch.randn(Cin, 27, Cout, device=dev) * 0.1
b = torch.randn(Cout, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
out = ops.conv3d_k3(x, w, b, stride=stride, relu=True)
t = timeit(lambda: ops.conv3d_k3(x, w, b, stride=stride, relu=True))
fl = 2.0 * out[0].numel() * Cin * 27 * Cout
by = 4.0 * (x.numel() + out.numel())
print(f"{os.environ.get('TAG','')} conv3d {Cin}->{Cout} s={stride} {W}x{H}x{D}: {t*1e3:.0f} us  {fl/t/1e9:.1f} TF  {by/t/1e9:.2f} TB/s (compulsory bytes)")
