"""Time the prob layer: the planar VALU kernel (conv3d_k3 on conv11's planar output) vs the matrix-core z-march on channels-last input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import ops
D, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 512, 640)
dev = torch.device("cuda:0")
torch.manual_seed(0)
x_cl = torch.randn(D, H, W, 8, device=dev)
x_pl = x_cl.permute(3, 0, 1, 2).contiguous()
w = torch.randn(1, 8, 3, 3, 3, device=dev) * 0.3
wpk = w.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous()
ws = ops.split_pack_prob_toeplitz(w)
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
a = ops.conv3d_k3(x_pl, wpk, None, relu=False)[0]
b = ops.conv3d_prob_sbf(x_cl, ws)
print("max diff", (a - b).abs().max().item(), "scale", a.abs().max().item())
print(f"{D}x{H}x{W}: planar VALU {timeit(lambda: ops.conv3d_k3(x_pl, wpk, None, relu=False)):.0f} us, matrix-core z-march {timeit(lambda: ops.conv3d_prob_sbf(x_cl, ws)):.0f} us "
      f"(TY={os.environ.get('CDS_PROB_TY', 'auto')}, zchunk={os.environ.get('CDS_PROB_ZCHUNK', 'auto')})")
