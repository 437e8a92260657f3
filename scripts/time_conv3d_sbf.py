"""Per-layer timing of the split-bf16 3x3x3 convolutions (channels-last) against the fp32 kernels (M1 CostRegNet shapes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
sel = sys.argv[1:]
for name, cin, cout, stride, D, H, W in (("conv0", 8, 8, 1, 192, 512, 640), ("conv1", 8, 16, 2, 192, 512, 640), ("conv2", 16, 16, 1, 96, 256, 320),
                                         ("conv3", 16, 32, 2, 96, 256, 320), ("conv4", 32, 32, 1, 48, 128, 160), ("conv5", 32, 64, 2, 48, 128, 160),
                                         ("conv6", 64, 64, 1, 24, 64, 80),
                                         ("s2conv0", 16, 8, 1, 32, 592, 800), ("s1conv0", 32, 8, 1, 48, 296, 400), ("s3conv0", 8, 8, 1, 8, 1184, 1600)):
    if sel and name not in sel: continue
    x = torch.randn(cin, D, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
    b = torch.randn(cout, device=dev)
    wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous()
    pair = cout == 8 and stride == 1
    ws = ops.split_pack_conv3d_pair(w) if pair else ops.split_pack_conv3d(w)
    code = ops.SBF_PAIR if pair else stride
    x_cl = x.permute(1, 2, 3, 0).contiguous()
    t32 = t(lambda: ops.conv3d_k3(x, wpk, b, stride=stride))
    tsb = t(lambda: ops.conv3d_sbf(x_cl, ws, b, cout, stride=code))
    o32 = ops.conv3d_k3(x, wpk, b, stride=stride)
    d = (o32 - ops.conv3d_sbf(x_cl, ws, b, cout, stride=code).permute(3, 0, 1, 2)).abs().max().item()
    fl = 2.0 * 27 * cin * cout * o32[0].numel()
    by = 4.0 * (x.numel() + o32.numel())
    print(f"{name}: fp32 kernel {t32:8.1f} us ({fl / t32 / 1e6:6.1f} TF)   split-bf16 {tsb:8.1f} us ({fl / tsb / 1e6:6.1f} TF-equivalent, {by / tsb / 1e6:5.2f} TB/s compulsory)   max |diff| {d:.2e}")
for name, cin, cout, D, H, W in (("conv7", 64, 32, 24, 64, 80), ("conv9", 32, 16, 48, 128, 160), ("conv11", 16, 8, 96, 256, 320)):
    if sel and name not in sel: continue
    x = torch.randn(cin, D, H, W, device=dev)
    w = torch.randn(cin, cout, 3, 3, 3, device=dev) / (27 * cin / 8) ** 0.5
    b = torch.randn(cout, device=dev)
    skip = torch.randn(cout, 2 * D, 2 * H, 2 * W, device=dev)
    wpk = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous()
    ws = ops.split_pack_deconv3d(w)
    x_cl = x.permute(1, 2, 3, 0).contiguous()
    skip_cl = skip.permute(1, 2, 3, 0).contiguous()
    t32 = t(lambda: ops.deconv3d_k3s2(x, wpk, b, skip=skip))
    tsb = t(lambda: ops.deconv3d_sbf(x_cl, ws, b, cout, skip=skip_cl, out_planar=(cout == 8)))
    got = ops.deconv3d_sbf(x_cl, ws, b, cout, skip=skip_cl, out_planar=(cout == 8))
    d = (ops.deconv3d_k3s2(x, wpk, b, skip=skip) - (got if cout == 8 else got.permute(3, 0, 1, 2))).abs().max().item()
    fl = 2.0 * 27 * cin * cout * D * H * W
    by = 4.0 * (x.numel() + 2 * skip.numel())
    tail = ""
    if (cin, cout) == (32, 16):     # the z-marching class-per-wave kernel CostRegNet runs conv9 on (csrc/deconv3d_zm.hip)
        wc = ops.split_pack_deconv_cls(w)
        tzm = t(lambda: ops.deconv3d_zm(x_cl, wc, b, skip=skip_cl))
        dz = (ops.deconv3d_zm(x_cl, wc, b, skip=skip_cl) - got).abs().max().item()
        tail = f"   z-march {tzm:8.1f} us ({by / tzm / 1e6:5.2f} TB/s compulsory, max |diff to tiled| {dz:.2e})"
    print(f"{name}: fp32 kernel {t32:8.1f} us ({fl / t32 / 1e6:6.1f} TF)   split-bf16 {tsb:8.1f} us ({fl / tsb / 1e6:6.1f} TF-equivalent, {by / tsb / 1e6:5.2f} TB/s compulsory)   max |diff| {d:.2e}{tail}")
if not sel or "tail" in sel:
    # conv11 + conv0 residual + prob in one launch (csrc/deconv_prob_zm.hip) against the two kernels timed above / below
    D, H, W = 96, 256, 320
    x_cl = torch.randn(D, H, W, 16, device=dev)
    skip_cl = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev)
    w11 = torch.randn(16, 8, 3, 3, 3, device=dev) / 54 ** 0.5
    b11 = torch.randn(8, device=dev)
    wp = torch.randn(1, 8, 3, 3, 3, device=dev) / 216 ** 0.5
    wz, tab = ops.split_pack_deconv_prob(w11), ops.pack_prob_table(wp)
    tf = t(lambda: ops.deconv_prob_zm(x_cl, wz, b11, skip_cl, tab))
    by = 4.0 * (x_cl.numel() + skip_cl.numel() + 8 * D * H * W)
    print(f"conv11 + residual + prob fused: {tf:8.1f} us ({by / tf / 1e6:5.2f} TB/s compulsory)")
if not sel or "prob" in sel:
    x = torch.randn(8, 192, 512, 640, device=dev)
    w = torch.randn(1, 8, 3, 3, 3, device=dev) / 216 ** 0.5
    wpk = w.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous()
    t32 = t(lambda: ops.conv3d_k3(x, wpk, None, relu=False))
    print(f"prob: planar fp32 kernel {t32:8.1f} us ({4.0 * 9 * x[0].numel() / t32 / 1e6:5.2f} TB/s compulsory)")
