"""conv00 (5 staged slots -> 8 output images, 1600x1184) in split-f16: time per launch (A/B of probe builds through CDS_MVSNET_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
H, W, V = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1184, 1600, 4)
imgs = torch.rand(1 + V, 3, H, W, generator=g).to(dev)
ks = (3, 7, 11)
ws = [torch.cat((torch.randn(8, 3, k, k, generator=g) / (3 * k * k) ** 0.5, torch.randn(3, 3, k, k, generator=g) * 0.1)).to(dev) for k in ks]
w1, b1, w2 = torch.randn(4, 3, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(3, 4, generator=g).to(dev)
epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(2 * V)], dtype=torch.float32).to(dev)
wh, winv = ops.split_pack_conv00(ws, f16=True)
bound = imgs.abs().amax().reshape(1)
fn = lambda: ops.conv00_cl(imgs, wh, None, w1, b1, w2, epi, 0.01, V, 0.1, in_bound=bound, w_inv_scale=winv)
for _ in range(3): fn()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(10): fn()
b.record(); torch.cuda.synchronize()
print(f"{os.environ.get('TAG', 'base')}: conv00 f16 {W}x{H} {1 + V} slots -> {2 * V} images: {a.elapsed_time(b) / 10 * 1e3:.1f} us (incl. the records reduction)")
