"""Time the fused conv11 + prob kernel at the headline shape (input cells 96 x 256 x 320 x 16).  CDS_MVSNET_LIB selects a probe build."""
import os, sys
import torch
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops
torch.manual_seed(0)
D, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (96, 256, 320)))
dev = "cuda"
x = torch.randn(D, H, W, 16, device=dev)
skip = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev)
ws = ops.split_pack_deconv_prob(torch.randn(16, 8, 3, 3, 3, device=dev) * 0.1)
b = torch.randn(8, device=dev) * 0.1
tab = ops.pack_prob_table(torch.randn(1, 8, 3, 3, 3, device=dev) * 0.1)
fn = lambda: ops.deconv_prob_zm(x, ws, b, skip, tab)
for _ in range(3):
    fn()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
print(f"{os.environ.get('CDS_MVSNET_LIB', 'product').split('.')[-2] if os.environ.get('CDS_MVSNET_LIB') else 'product':>12s}  nseg={os.environ.get('CDS_DPZ_NSEG', 'auto'):>4s}  {best:8.1f} us", flush=True)
