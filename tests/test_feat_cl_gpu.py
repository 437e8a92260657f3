"""GPU tests of the channels-last FeatureNet kernels (csrc/feat_cl.hip) against the planar kernels they replace (which the goldens
G5 / G9 and float64 tests of test_hip_parity.py pin) on the same seeded inputs.  Same arithmetic in another data layout: the
matrix-core DynamicConv must agree to the last bits, the VALU layers to fp32 summation-order round-off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from cds_mvsnet_amd import ops as o
    return o


def _cl(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _affine(N, C, g):
    return torch.stack((0.5 + torch.rand(N, C, generator=g), 0.3 * torch.randn(N, C, generator=g), torch.full((N, C), 0.1)), dim=-1).contiguous()


@pytest.mark.parametrize("c,ks,N,H,W,bias", [(8, (3, 5, 7), 2, 21, 44, True), (16, (3, 5), 3, 16, 36, True), (32, (1, 3), 2, 9, 20, False),
                                             (8, (1, 3), 1, 8, 64, True), (16, (3, 5), 1, 40, 100, False), (16, (1, 3), 2, 17, 33, True),
                                             (32, (1, 3), 1, 24, 70, True), (8, (3, 5, 7), 1, 5, 7, False)])
def test_dynconv_cl_equals_planar_fused(c, ks, N, H, W, bias, dev, ops):
    """cds_dynconv_cl_f32 against cds_dynconv_fused_sbf_f32 (planar, W % 4 == 0) or branches + blend (any W): the same K-loop and
    epilogue arithmetic (feat_common.hpp) on another activation layout: blended output and norm-curvature bit-identical, the
    InstanceNorm statistics to fp64 regrouping.
    Partial tiles, widths that are not multiples of 4, tiles smaller than the halo, several images, both temperatures."""
    g = torch.Generator().manual_seed(c * 3 + len(ks) + H)
    K = len(ks)
    x = torch.randn(N, c, H, W, generator=g)
    aff = _affine(N, c, g)
    co3 = c + 3
    wlist = [(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks]
    ws = ops.split_pack_dynconv(wlist)
    bs = torch.randn(K, co3, generator=g)
    bs[:, c:] = 0.0                                  # att_convs have no bias (dynamic_conv.py:86)
    bs = bs.to(dev) if bias else None
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    xd, ad = x.to(dev), aff.to(dev)
    for T in (1.0, 0.01):
        if W % 4 == 0:
            o2, n2, s2, a2 = ops.dynconv_fused_sbf(xd, ws, bs, c, ks, w1, b1, w2, epi, T, 0.1, in_affine=ad)
        else:                                        # the planar matrix-core kernels need W % 4 == 0: VALU branches + blend
            br = torch.empty((K, N, co3, H, W), device=dev)
            from cds_mvsnet_amd.model import _pack2d
            for i, k in enumerate(ks):
                ops.conv2d(xd, _pack2d(wlist[i]), bs[i].contiguous() if bs is not None else None, co3, k, 1, (k - 1) // 2, ops.ACT_NONE,
                           out=br[i], in_affine=ad)
            o2, n2, s2, a2 = ops.dynconv_blend(br, w1, b1, w2, epi, T, 1, stats_slope=0.1)
        o1, n1, s1, a1 = ops.dynconv_cl(_cl(xd), ws, bs, ks, w1, b1, w2, epi, T, 0.1, in_affine=ad)
        o1 = o1.permute(0, 3, 1, 2)
        tol = 1e-6 if W % 4 == 0 else 2e-4 / min(T, 1.0) * 1e-1    # VALU fp32 chains vs split-bf16: fp32-class, amplified by 1 / T
        scale = max(1.0, o2.abs().max().item())
        assert (o1 - o2).abs().max().item() <= tol * scale, (T, (o1 - o2).abs().max().item())
        assert (n1 - n2).abs().max().item() <= tol * max(1.0, n2.abs().max().item())
        if W % 4 == 0:
            assert torch.allclose(s1, s2, rtol=1e-9, atol=1e-6)
            assert torch.allclose(a1, a2, rtol=1e-5, atol=1e-6)
            print(f"dynconv_cl C={c} k={ks} T={T}: bit-identical to the planar fused kernel: {torch.equal(o1, o2)} / {torch.equal(n1, n2)}")


@pytest.mark.parametrize("N,n_shared,H,W", [(4, 2, 24, 40), (1, 1, 9, 13), (5, 3, 16, 100)])
def test_blend_cl_equals_planar_blend(N, n_shared, H, W, dev, ops):
    """cds_dynconv_blend_cl_f32 (conv00's epilogue, channels-last result) against cds_dynconv_blend_stats_f32: bit-identical output
    and norm-curvature (same per-pixel arithmetic), statistics to fp64 regrouping; shared reference slots."""
    g = torch.Generator().manual_seed(N * 7 + W)
    br = torch.randn(3, N - n_shared + 1, 11, H, W, generator=g).to(dev)
    w1, b1, w2 = torch.randn(4, 3, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(3, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.4 + 3.0 * n, H * 2.1 + n] for n in range(N)], dtype=torch.float32)
    for T in (1.0, 0.01):
        o2, n2, s2, a2 = ops.dynconv_blend(br, w1, b1, w2, epi, T, n_shared, stats_slope=0.1)
        o1, n1, s1, a1 = ops.dynconv_blend_cl(br, w1, b1, w2, epi, T, n_shared, 0.1)
        assert torch.equal(o1.permute(0, 3, 1, 2), o2) and torch.equal(n1, n2)
        assert torch.allclose(s1, s2, rtol=1e-10, atol=1e-8) and torch.allclose(a1, a2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("cin,cout,N,H,W", [(8, 16, 2, 20, 36), (16, 32, 3, 14, 22), (8, 16, 1, 7, 9), (16, 32, 1, 32, 64), (8, 16, 2, 70, 133),
                                            (16, 32, 2, 41, 67)])
def test_downsample_cl_vs_planar(cin, cout, N, H, W, dev, ops):
    """cds_conv2d_k3s2_cl_f32 against cds_conv2d_affine_f32 (k 3, stride 2, pad 1) and float64; statistics pass against
    cds_instnorm_affine_f32."""
    import torch.nn.functional as F
    from cds_mvsnet_amd.model import _pack2d
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    aff = _affine(N, cin, g)
    xn = x * aff[:, :, 0, None, None] + aff[:, :, 1, None, None]
    xn = torch.where(xn > 0, xn, xn * 0.1)
    want = F.conv2d(xn.double(), w.double(), stride=2, padding=1)
    old = ops.conv2d(x.to(dev), _pack2d(w.to(dev)), None, cout, 3, 2, 1, in_affine=aff.to(dev))
    w9 = w.permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous().to(dev)
    new = ops.conv2d_k3s2_cl(_cl(x.to(dev)), w9, cout, aff.to(dev))
    assert tuple(new.shape) == (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, cout)
    e_new = (new.permute(0, 3, 1, 2).cpu().double() - want).abs().max().item()
    e_old = (old.cpu().double() - want).abs().max().item()
    assert e_new <= 1.5 * e_old + want.abs().max().item() * 2.0 ** -22, (e_new, e_old)
    # round 6: the same layer on the matrix cores in split-f16 arithmetic (what FeatureNet runs): fp32-class against float64, with the
    # tight bound and with the loose one the model uses (sqrt(H W) of an InstanceNorm-ed map; here the input is not normalised: 8 x max)
    wh, winv = ops.split_pack_dynconv([w.to(dev)], f16=True)
    for bound in (float(xn.abs().max()), 8.0 * float(xn.abs().max())):
        mf = ops.conv2d_k3s2_cl(_cl(x.to(dev)), None, cout, aff.to(dev), wsplit=wh, w_inv_scale=winv, x_bound=bound)
        assert tuple(mf.shape) == tuple(new.shape)
        e_mf = (mf.permute(0, 3, 1, 2).cpu().double() - want).abs().max().item()
        assert e_mf <= 1.5 * e_old + want.abs().max().item() * 2.0 ** -22, (bound, e_mf, e_old)
    st, a_new = ops.instnorm_stats_cl(new, 0.1)
    a_old = ops.instnorm_affine(new.permute(0, 3, 1, 2).contiguous(), 0.1)
    assert torch.allclose(a_new, a_old, rtol=1e-5, atol=1e-6)
    y = new.permute(0, 3, 1, 2).cpu().double()
    assert torch.allclose(st[..., 0].cpu(), y.sum((2, 3)), rtol=1e-10, atol=1e-7)
    assert torch.allclose(st[..., 1].cpu(), (y * y).sum((2, 3)), rtol=1e-10, atol=1e-7)


@pytest.mark.parametrize("ca,cb,cout,N,H,W,coarse_aff", [(32, 16, 16, 2, 12, 20, True), (16, 8, 8, 3, 18, 26, False), (16, 8, 8, 1, 64, 34, True)])
def test_fpn_lateral_cl_vs_planar(ca, cb, cout, N, H, W, coarse_aff, dev, ops):
    """cds_conv2d_fpn_cl_f32 against cds_conv2d_fpn_f32: same products, fmaf chain over the same channel order: bit-identical output
    and statistics to fp64 regrouping."""
    from cds_mvsnet_amd.model import _pack2d
    g = torch.Generator().manual_seed(ca + H)
    xa, xb = torch.randn(N, ca, H // 2, W // 2, generator=g), torch.randn(N, cb, H, W, generator=g)
    w = torch.randn(cout, ca + cb, 1, 1, generator=g) / (ca + cb) ** 0.5
    aa = _affine(N, ca, g).to(dev) if coarse_aff else None
    ab = _affine(N, cb, g).to(dev)
    o2, a2 = ops.conv2d_fpn(xa.to(dev), xb.to(dev), _pack2d(w.to(dev)), cout, aa, ab, stats_slope=0.1)
    wt = w.reshape(cout, ca + cb).t().contiguous().to(dev)
    o1, a1 = ops.conv2d_fpn_cl(_cl(xa.to(dev)), _cl(xb.to(dev)), wt, cout, aa, ab, 0.1)
    assert torch.equal(o1.permute(0, 3, 1, 2), o2)
    assert torch.allclose(a1, a2, rtol=1e-5, atol=1e-6)
    o3 = ops.conv2d_fpn_cl(_cl(xa.to(dev)), _cl(xb.to(dev)), wt, cout, aa, ab, None)
    assert torch.equal(o3, o1)


@pytest.mark.parametrize("C,N,H,W,n_chw", [(8, 4, 20, 36, 2), (16, 3, 9, 13, 0), (32, 2, 16, 24, 2), (8, 1, 7, 300, 1)])
def test_instnorm_apply_cl(C, N, H, W, n_chw, dev, ops):
    """cds_instnorm_apply_cl_f32 against cds_instnorm_apply_f32 on the same statistics: bit-identical values in both output
    layouts; the channels-last part covers the images n >= cl_from only."""
    g = torch.Generator().manual_seed(C + W)
    x = (torch.randn(N, C, H, W, generator=g) * 2.0 + 0.5).to(dev)
    st, _ = ops.instnorm_stats_cl(_cl(x), 0.1)
    want = ops.instnorm_apply(x, st, ops.ACT_TANH)
    for cl_from in (0, n_chw):
        if cl_from >= N:
            continue
        cl, chw = ops.instnorm_apply_cl(_cl(x), st, ops.ACT_TANH, n_chw, cl_from=cl_from)
        assert torch.equal(cl.permute(0, 3, 1, 2), want[cl_from:])
        if n_chw:
            assert torch.equal(chw, want[:n_chw])
        else:
            assert chw is None
    if n_chw:
        cl, chw = ops.instnorm_apply_cl(_cl(x), st, ops.ACT_TANH, n_chw, cl_from=None)
        assert cl is None and torch.equal(chw, want[:n_chw])


def test_feature_runner_layouts_agree(dev, ops, monkeypatch):
    """The whole FeatureNet on channels-last activations against the planar runner of rounds 1-4 on the same weights and images
    (three pairs: shared reference copies, CHW + HWC outputs, partial tiles at every level): mean difference <= 1e-5 at T = 0.01 (the
    two paths differ by conv00's arithmetic - split-bf16 matrix cores vs fp32 fmaf chains - and by fp32 summation order in the VALU
    layers, amplified by softmax(./T)), curvature maps alike."""
    import cds_mvsnet_amd.model as cm
    from cds_mvsnet_amd import FeatureNet, seeded_init_
    net = seeded_init_(FeatureNet(8), 7).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    H, W, V = 96, 160, 3
    low = torch.rand(V + 1, 3, H // 8, W // 8, generator=g)
    imgs = torch.nn.functional.interpolate(low, (H, W), mode="bicubic", align_corners=False).clamp(0, 1)
    batch = torch.stack([imgs[0]] * V + [imgs[v + 1] for v in range(V)]).to(dev)
    epi = torch.tensor([[300.0 + 40 * i, -120.0 + 35 * i] for i in range(2 * V)], dtype=torch.float32)
    outs = {}
    for layout in (True, False):
        monkeypatch.setattr(cm, "USE_FEAT_CL", layout)
        outs[layout] = cm._FeatureRunner(net)(batch, epi, 0.01, n_chw=V, n_shared=V)
    for s in ("stage1", "stage2", "stage3"):
        a, b = outs[True][s], outs[False][s]
        for j in range(4):
            assert a[j].shape == b[j].shape, (s, j, a[j].shape, b[j].shape)
            err = (a[j] - b[j]).abs()
            assert err.mean().item() < 1e-5 and err.max().item() < 2e-3, (s, j, err.mean().item(), err.max().item())


@pytest.mark.parametrize("c,ks,N,H,W", [(8, (1, 3), 8, 592, 800), (8, (3, 5, 7), 8, 296, 400), (16, (3, 5), 8, 296, 400), (16, (1, 3), 8, 296, 400),
                                        (32, (1, 3), 8, 296, 400)])
def test_dynconv_cl_is_bit_stable_at_scale(c, ks, N, H, W, dev, ops):
    """Thousands of workgroups (several per CU, in different phases: MFMA K-loops beside epilogues), four repetitions, every output
    bit compared with the unfused planar path (branches kernel + blend kernel).  Round 5: the first form of this kernel (transposed
    GEMM, weights from global loads as the MFMA's A operand) produced wrong values in lanes 48-63 for about one 16-pixel tile in 10^4
    - in its own waves and in OTHER kernels sharing the CU - only at this scale, differently in every run; small-shape tests never
    saw it (profiles/r05_experiments.md)."""
    g = torch.Generator().manual_seed(c + len(ks))
    K, co3 = len(ks), c + 3
    x = torch.randn(N, c, H, W, generator=g).to(dev)
    xcl = _cl(x)
    aff = _affine(N, c, g).to(dev)
    wsp = ops.split_pack_dynconv([(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks])
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    for T in (1.0, 0.01):
        br = ops.dynconv_branches_sbf(x, wsp, None, co3, ks, in_affine=aff)
        o2, n2, _, _ = ops.dynconv_blend(br, w1, b1, w2, epi, T, 1, stats_slope=0.1)
        del br
        for rep in range(4):
            o1, n1, _, _ = ops.dynconv_cl(xcl, wsp, None, ks, w1, b1, w2, epi, T, 0.1, in_affine=aff)
            bad = int((o1.permute(0, 3, 1, 2) != o2).sum()), int((n1 != n2).sum())
            assert bad == (0, 0), (T, rep, bad)
        o3, n3, _, _ = ops.dynconv_fused_sbf(x, wsp, None, c, ks, w1, b1, w2, epi, T, 0.1, in_affine=aff)
        assert torch.equal(o3, o2) and torch.equal(n3, n2)          # the planar fused kernel as well


@pytest.mark.parametrize("V,H,W", [(4, 20, 36), (1, 8, 33), (3, 13, 100), (6, 37, 50)])
def test_visibility_layers_channels_last(V, H, W, dev, ops):
    """cds_vis_layer1_cl_f32 and cds_conv2d_k3_relu_cl_f32 (the visibility CNN on channels-last activations) against float64 and the
    planar kernels they replace: layer 1 (exact fp32 fmaf chains) to round-off, the matrix-core layers fp32-class, with and without the
    fused head; widths that are not multiples of 4, partial tiles."""
    import torch.nn.functional as F
    from cds_mvsnet_amd.model import _pack2d
    g = torch.Generator().manual_seed(V * 100 + W)
    ent, nc = torch.rand(V, H, W, generator=g) * 3.0, torch.rand(V, H, W, generator=g)
    w0, b0 = torch.randn(16, 2, 3, 3, generator=g) / 4.0, torch.randn(16, generator=g) * 0.2
    y0 = F.conv2d(torch.stack((ent, nc), 1).double(), w0.double(), b0.double(), padding=1).clamp_min(0)
    x1 = ops.vis_layer1_cl(ent.to(dev), nc.to(dev), _pack2d(w0.to(dev)), b0.to(dev))
    assert (x1.permute(0, 3, 1, 2).cpu().double() - y0).abs().max().item() <= 1e-5 * max(1.0, y0.abs().max().item())
    w, b = torch.randn(16, 16, 3, 3, generator=g) / 12.0, torch.randn(16, generator=g) * 0.2
    hw, hb = torch.randn(16, generator=g) * 0.3, torch.randn(1, generator=g)
    xin = x1.permute(0, 3, 1, 2).cpu()
    y64 = F.conv2d(xin.double(), w.double(), b.double(), padding=1).clamp_min(0)
    ws = ops.split_pack_dynconv([w.to(dev)])
    got = ops.conv2d_k3_relu_cl(x1, ws, b.to(dev))
    ref32 = F.conv2d(xin, w, b, padding=1).clamp_min(0)
    e_new, e_ref = (got.permute(0, 3, 1, 2).cpu().double() - y64).abs().max().item(), (ref32.double() - y64).abs().max().item()
    assert e_new <= 1.5 * e_ref + y64.abs().max().item() * 2.0 ** -23, (e_new, e_ref)
    h64 = torch.sigmoid((y64 * hw.double().view(1, 16, 1, 1)).sum(1) + hb.double())
    goth = ops.conv2d_k3_relu_cl(x1, ws, b.to(dev), head_w=hw.to(dev), head_b=hb.to(dev))
    assert tuple(goth.shape) == (V, H, W)
    assert (goth.cpu().double() - h64).abs().max().item() <= 2e-6
    if W % 4 == 0:      # the planar matrix-core kernel: same K-loop -> bit-identical
        old = ops.conv2d_k3_relu_sbf(xin.to(dev).contiguous(), ws, b.to(dev))
        assert torch.equal(got.permute(0, 3, 1, 2), old)
        oldh = ops.conv2d_k3_relu_sbf(xin.to(dev).contiguous(), ws, b.to(dev), head_w=hw.to(dev), head_b=hb.to(dev))
        assert torch.equal(goth, oldh)


@pytest.mark.parametrize("N,n_shared,H,W", [(4, 2, 24, 40), (1, 1, 9, 13), (5, 3, 16, 100), (3, 1, 40, 70), (8, 4, 64, 96)])
def test_conv00_on_matrix_cores(N, n_shared, H, W, dev, ops):
    """cds_conv00_cl_f32 (conv00: 3 -> 8 + 3, kernel sizes 3 / 7 / 11, tap-pair K-steps on the matrix cores, blend fused, shared
    reference slots) against the exact-fp32 VALU branch kernels + blend kernel it replaces and against float64: fp32-class
    (error vs float64 <= 1.5x the VALU kernels'), blended output / curvature within the softmax(./T) amplification of that round-off;
    partial tiles, widths that are not multiples of 4, several slots."""
    import torch.nn.functional as F
    from cds_mvsnet_amd.model import _pack2d
    g = torch.Generator().manual_seed(N * 10 + W)
    S = N - n_shared + 1
    imgs = torch.rand(S, 3, H, W, generator=g)
    ks = (3, 7, 11)
    ws = [torch.cat((torch.randn(8, 3, k, k, generator=g) / (3 * k * k) ** 0.5, torch.randn(3, 3, k, k, generator=g) * 0.1)) for k in ks]
    w1, b1, w2 = torch.randn(4, 3, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(3, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.4 + 3.0 * n, H * 2.1 + n] for n in range(N)], dtype=torch.float32)
    br = torch.empty((3, S, 11, H, W), device=dev)
    for i, k in enumerate(ks):
        ops.conv2d(imgs.to(dev), _pack2d(ws[i].to(dev)), None, 11, k, 1, (k - 1) // 2, ops.ACT_NONE, out=br[i])
    wsp = ops.split_pack_conv00([w.to(dev) for w in ws])
    # branch responses themselves: T -> infinity makes the blend the plain mean of the three branches
    for T in (1e9, 1.0, 0.01):
        o2, n2, s2, a2 = ops.dynconv_blend_cl(br, w1, b1, w2, epi, T, n_shared, 0.1)
        o1, n1, s1, a1 = ops.conv00_cl(imgs.to(dev), wsp, None, w1, b1, w2, epi, T, n_shared, 0.1)
        assert o1.shape == o2.shape == (N, H, W, 8)
        tol = 3e-6 if T >= 1.0 else 3e-4
        assert (o1 - o2).abs().max().item() <= tol * max(1.0, o2.abs().max().item()), (T, (o1 - o2).abs().max().item())
        assert (n1 - n2).abs().max().item() <= tol * max(1.0, n2.abs().max().item()), (T, (n1 - n2).abs().max().item())
        if T >= 1.0:
            assert torch.allclose(a1, a2, rtol=2e-4, atol=2e-5)
    # float64: the mean-of-branches output (T = 1e9) against F.conv2d
    want = sum(F.conv2d(imgs.double(), w[:8].double(), padding=(k - 1) // 2) for w, k in zip(ws, ks)) / 3.0
    o1 = ops.conv00_cl(imgs.to(dev), wsp, None, w1, b1, w2, epi, 1e9, n_shared, 0.1)[0].permute(0, 3, 1, 2).cpu().double()
    o2 = ops.dynconv_blend_cl(br, w1, b1, w2, epi, 1e9, n_shared, 0.1)[0].permute(0, 3, 1, 2).cpu().double()
    slot = [0] * n_shared + list(range(1, S))
    e1 = max((o1[n] - want[slot[n]]).abs().max().item() for n in range(N))
    e2 = max((o2[n] - want[slot[n]]).abs().max().item() for n in range(N))
    assert e1 <= 1.5 * e2 + want.abs().max().item() * 2.0 ** -22, (e1, e2)
