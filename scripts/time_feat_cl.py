"""FeatureNet layers at the 1600x1184 cascade shapes (8 images): channels-last kernels (feat_cl.hip) vs the planar ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cds_mvsnet_amd import ops
from cds_mvsnet_amd.model import _pack2d
dev = torch.device("cuda")
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
N = 8
only = sys.argv[1:]
g = torch.Generator().manual_seed(0)
for name, c, ks, H, W in (("conv01", 8, (3, 5, 7), 1184, 1600), ("conv10", 16, (3, 5), 592, 800), ("conv20", 32, (1, 3), 296, 400),
                          ("out2", 16, (1, 3), 592, 800), ("out3", 8, (1, 3), 1184, 1600)):
    if only and name not in only:
        continue
    K, co3 = len(ks), c + 3
    x = torch.randn(N, c, H, W, generator=g).to(dev)
    xcl = x.permute(0, 2, 3, 1).contiguous()
    aff = torch.stack((0.5 + torch.rand(N, c, generator=g), 0.3 * torch.randn(N, c, generator=g), torch.full((N, c), 0.1)), -1).to(dev).contiguous()
    wsp = ops.split_pack_dynconv([(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks])
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    tp = t(lambda: ops.dynconv_fused_sbf(x, wsp, None, c, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff))
    tc = t(lambda: ops.dynconv_cl(xcl, wsp, None, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff))
    o2 = ops.dynconv_fused_sbf(x, wsp, None, c, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff)[0]
    o1 = ops.dynconv_cl(xcl, wsp, None, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff)[0].permute(0, 3, 1, 2)
    nks = sum((k * k + 3) // 4 for k in ks) * (c // 8)
    mfma = nks * ((co3 + 15) // 16) * 6 * (N * H * W / 16)          # v_mfma_f32_16x16x32_bf16 count
    floor = mfma * 16 / (1024 * 2.1e9) * 1e6                        # us at 16 cycles per MFMA and SIMD, 2.1 GHz
    print(f"{name}: planar fused {tp:8.1f} us   channels-last {tc:8.1f} us   (matrix-pipe floor {floor:6.1f} us = {floor / tc * 100:4.1f} %)   "
          f"max |diff| {(o1 - o2).abs().max().item():.2e} equal {torch.equal(o1, o2)}")
if not only or "conv00" in only:
    H, W, V = 1184, 1600, 4
    imgs = torch.rand(1 + V, 3, H, W, generator=g).to(dev)
    ks = (3, 7, 11)
    ws = [torch.cat((torch.randn(8, 3, k, k, generator=g) / (3 * k * k) ** 0.5, torch.randn(3, 3, k, k, generator=g) * 0.1)).to(dev) for k in ks]
    w1, b1, w2 = torch.randn(4, 3, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(3, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(2 * V)], dtype=torch.float32)
    wpk = [_pack2d(w) for w in ws]
    wsp = ops.split_pack_conv00(ws)
    def valu():
        br = torch.empty((3, 1 + V, 11, H, W), device=dev)
        for i, k in enumerate(ks):
            ops.conv2d(imgs, wpk[i], None, 11, k, 1, (k - 1) // 2, ops.ACT_NONE, out=br[i])
        return ops.dynconv_blend_cl(br, w1, b1, w2, epi, 0.01, V, 0.1)
    mfma = 26 * 6 * ((1 + V) * H * W / 16)
    floor = mfma * 16 / (1024 * 2.1e9) * 1e6
    tm = t(lambda: ops.conv00_cl(imgs, wsp, None, w1, b1, w2, epi, 0.01, V, 0.1))
    print(f"conv00 (5 slots -> 8 images): VALU branches + blend {t(valu):8.1f} us   matrix cores {tm:8.1f} us   (matrix-pipe floor {floor:6.1f} us = {floor / tm * 100:4.1f} %)")
if not only or "small" in only:
    # the small layers: downsample1 / 2, inner1 / 2, the tanh outputs
    for name, cin, cout, H, W in (("downsample1", 8, 16, 1184, 1600), ("downsample2", 16, 32, 592, 800)):
        x = torch.randn(N, cin, H, W, generator=g).to(dev); xcl = x.permute(0, 2, 3, 1).contiguous()
        aff = torch.stack((torch.ones(N, cin), torch.zeros(N, cin), torch.full((N, cin), 0.1)), -1).to(dev).contiguous()
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        wpk, w9 = _pack2d(w), w.permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()
        def planar():
            y = ops.conv2d(x, wpk, None, cout, 3, 2, 1, in_affine=aff)
            return ops.instnorm_affine(y, 0.1)
        def cl():
            y = ops.conv2d_k3s2_cl(xcl, w9, cout, aff)
            return ops.instnorm_stats_cl(y, 0.1)
        print(f"{name}: planar conv + stats {t(planar):8.1f} us   channels-last {t(cl):8.1f} us")
    for name, ca, cb, cout, H, W in (("inner1", 32, 16, 16, 592, 800), ("inner2", 16, 8, 8, 1184, 1600)):
        xa, xb = torch.randn(N, ca, H // 2, W // 2, generator=g).to(dev), torch.randn(N, cb, H, W, generator=g).to(dev)
        w = (torch.randn(cout, ca + cb, 1, 1, generator=g) / (ca + cb) ** 0.5).to(dev)
        aa = torch.stack((torch.ones(N, ca), torch.zeros(N, ca), torch.full((N, ca), 0.1)), -1).to(dev).contiguous()
        ab = torch.stack((torch.ones(N, cb), torch.zeros(N, cb), torch.full((N, cb), 0.1)), -1).to(dev).contiguous()
        wpk, wt = _pack2d(w), w.reshape(cout, ca + cb).t().contiguous()
        xacl, xbcl = xa.permute(0, 2, 3, 1).contiguous(), xb.permute(0, 2, 3, 1).contiguous()
        print(f"{name}: planar {t(lambda: ops.conv2d_fpn(xa, xb, wpk, cout, aa, ab, stats_slope=0.1)):8.1f} us   "
              f"channels-last {t(lambda: ops.conv2d_fpn_cl(xacl, xbcl, wt, cout, aa, ab, 0.1)):8.1f} us")
    for name, C, H, W in (("out1 tanh", 32, 296, 400), ("out2 tanh", 16, 592, 800), ("out3 tanh", 8, 1184, 1600)):
        x = torch.randn(N, C, H, W, generator=g).to(dev); xcl = x.permute(0, 2, 3, 1).contiguous()
        st = ops.instnorm_stats_cl(xcl, 0.1)[0]
        def planar():
            a = ops.instnorm_apply(x[:4], st[:4], ops.ACT_TANH)
            return a, ops.instnorm_apply(x[4:], st[4:], ops.ACT_TANH, out_hwc=True)
        print(f"{name}: planar (chw + hwc) {t(planar):8.1f} us   channels-last {t(lambda: ops.instnorm_apply_cl(xcl, st, ops.ACT_TANH, 4, cl_from=4)):8.1f} us")
