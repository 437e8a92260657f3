"""Fused conv11 + prob vs the two separate kernels at the cascade's stage shapes (input cells D, H, W)."""
import sys
import torch
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (D, H, W) in {"M1": (96, 256, 320), "M3 stage1": (24, 148, 200), "M3 stage2": (16, 296, 400), "M3 stage3": (4, 592, 800),
                        "M2 stage1": (24, 64, 80), "M2 stage2": (16, 128, 160), "M2 stage3": (4, 256, 320), "M4 stage3": (4, 528, 960)}.items():
    x = torch.randn(D, H, W, 16, device=dev); skip = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev)
    w11 = torch.randn(16, 8, 3, 3, 3, device=dev) * 0.1; b = torch.randn(8, device=dev) * 0.1; wp = torch.randn(1, 8, 3, 3, 3, device=dev) * 0.1
    wo, wn, tab = ops.split_pack_deconv3d(w11), ops.split_pack_deconv_prob(w11), ops.pack_prob_table(wp)
    wpk = wp.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous()
    sep = t(lambda: ops.conv3d_k3(ops.deconv3d_sbf(x, wo, b, 8, skip=skip, out_planar=True), wpk, None, relu=False))
    fus = t(lambda: ops.deconv_prob_zm(x, wn, b, skip, tab))
    print(f"{name:10s} cells {D}x{H}x{W}: separate {sep:8.1f} us   fused {fus:8.1f} us", flush=True)
