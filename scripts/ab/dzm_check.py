"""Class-per-wave z-marching conv9 kernel (csrc/deconv3d_zm.hip) against the tiled kernel and float64; timings at the M1 / cascade shapes."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops
dev = "cuda"
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(D, H, W, check64=True, timing=False):
    x = torch.randn(D, H, W, 32, device=dev)
    skip = torch.randn(2 * D, 2 * H, 2 * W, 16, device=dev)
    w = torch.randn(32, 16, 3, 3, 3, device=dev) * 0.08
    b = torch.randn(16, device=dev) * 0.1
    wo, wn = ops.split_pack_deconv3d(w), ops.split_pack_deconv_cls(w)
    ref = ops.deconv3d_sbf(x, wo, b, 16, skip=skip)
    got = ops.deconv3d_zm(x, wn, b, skip=skip)
    torch.cuda.synchronize()
    msg = f"D{D} H{H} W{W}: max |zm - tiled| {(got - ref).abs().max().item():.3e}"
    if check64:
        r64 = F.relu(F.conv_transpose3d(x.permute(3, 0, 1, 2)[None].double(), w.double(), b.double(), stride=2, padding=1, output_padding=1))[0] \
            + skip.permute(3, 0, 1, 2).double()
        msg += f"  vs float64: zm {(got.permute(3, 0, 1, 2).double() - r64).abs().max().item():.3e}, tiled {(ref.permute(3, 0, 1, 2).double() - r64).abs().max().item():.3e}"
    if timing:
        msg += f"   tiled {t(lambda: ops.deconv3d_sbf(x, wo, b, 16, skip=skip)):.1f} us, zm {t(lambda: ops.deconv3d_zm(x, wn, b, skip=skip)):.1f} us"
    print(msg, flush=True)
for shp in ((1, 1, 1), (2, 3, 5), (3, 7, 17), (5, 9, 33), (4, 8, 16), (6, 20, 50)):
    run(*shp)
run(48, 128, 160, check64=False, timing=True)      # M1 conv9
run(12, 74, 100, check64=False, timing=True)       # 1600x1184 stage 1
run(8, 148, 200, check64=False, timing=True)       # stage 2
run(2, 296, 400, check64=False, timing=True)       # stage 3
