"""Fused conv11 + residual + prob kernel (csrc/deconv_prob_zm.hip) against the two separate kernels and a float64 torch
restatement; then timings at the headline shape.  GPU only."""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops

torch.manual_seed(0)
dev = "cuda"


def run(D, H, W, check64=True, timing=False):
    x = torch.randn(D, H, W, 16, device=dev)
    skip = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev)
    w11 = torch.randn(16, 8, 3, 3, 3, device=dev) * 0.1
    b11 = torch.randn(8, device=dev) * 0.1
    wp = torch.randn(1, 8, 3, 3, 3, device=dev) * 0.1
    ws_old = ops.split_pack_deconv3d(w11)
    ws_new = ops.split_pack_deconv_prob(w11)
    tab = ops.pack_prob_table(wp)
    wpk = wp.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous()
    y = ops.deconv3d_sbf(x, ws_old, b11, 8, skip=skip, out_planar=True)
    ref = ops.conv3d_k3(y, wpk, None, relu=False)[0]
    got = ops.deconv_prob_zm(x, ws_new, b11, skip, tab)
    torch.cuda.synchronize()
    d = (got - ref).abs().max().item()
    msg = f"D{D} H{H} W{W}: max |fused - separate| {d:.3e} (max |ref| {ref.abs().max().item():.2f})"
    if check64:
        x64 = x.permute(3, 0, 1, 2)[None].double()
        y64 = F.relu(F.conv_transpose3d(x64, w11.double(), b11.double(), stride=2, padding=1, output_padding=1)) \
            + skip.permute(3, 0, 1, 2)[None].double()
        r64 = F.conv3d(y64, wp.double(), padding=1)[0, 0]
        e_f = (got.double() - r64).abs().max().item()
        e_s = (ref.double() - r64).abs().max().item()
        msg += f"   vs float64: fused {e_f:.3e}, separate {e_s:.3e}"
    print(msg, flush=True)
    if timing:
        for name, fn in (("separate", lambda: ops.conv3d_k3(ops.deconv3d_sbf(x, ws_old, b11, 8, skip=skip, out_planar=True), wpk, None, relu=False)),
                         ("fused", lambda: ops.deconv_prob_zm(x, ws_new, b11, skip, tab))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"   {name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
    return d


for shp in ((1, 1, 1), (2, 3, 5), (3, 7, 31), (5, 13, 61), (4, 6, 30), (9, 12, 64), (7, 20, 33)):
    run(*shp)
run(12, 64, 80, check64=True, timing=True)
run(96, 256, 320, check64=False, timing=True)
