"""Time one cds_conv2d launch shape with HIP events.  Usage: time_conv2d.py N Cin Cout k H W [stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops
N, Cin, Cout, k, H, W = (int(a) for a in sys.argv[1:7])
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
dev = torch.device("cuda:0")
x = torch.randn(N, Cin, H, W, device=dev)
coutp = (Cout + 7) // 8 * 8
w = torch.zeros(Cin, k * k, coutp, device=dev); w[:, :, :Cout] = torch.randn(Cin, k * k, Cout, device=dev) * 0.1
aff = torch.rand(N, Cin, 3, device=dev)
pad = (k - 1) // 2
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
out = torch.empty(N, Cout, Ho, Wo, device=dev)
for name, a in (("plain", None), ("affine", aff)):
    t = timeit(lambda: ops.conv2d(x, w, None, Cout, k, stride, pad, 0, out=out, in_affine=a))
    fl = 2.0 * N * Ho * Wo * Cin * k * k * Cout
    by = 4.0 * (x.numel() + out.numel())
    print(f"{os.environ.get('TAG','')} conv2d N={N} {Cin}->{Cout} k={k} s={stride} {W}x{H} {name}: {t*1e3:.0f} us  {fl/t/1e9:.1f} TF  {by/t/1e9:.2f} TB/s")
