"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle on seeded inputs.  Run with ``pytest -m gpu`` on an MI355X.

Tolerances (fp32): warped / aggregated volume <= 1e-5 abs (observed ~1e-7),
entropy / visibility <= 2e-5, CostRegNet output <= 1e-4, single DynamicConv <= 5e-5, FeatureNet features
mean <= 2e-5 (max <= 2e-3 at T=0.01, round-off amplified by softmax(./T)), stage depth mean-L1 <= 1e-3
(the reference is not bit-stable against itself below ~2e-4).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from cds_mvsnet_amd import ops as o
    assert o.version() >= 100
    return o


def _stage_inputs(g, dev, ops):
    ref = g["ref_fea"].to(dev).contiguous()
    src_hwc = torch.stack([ops.chw_to_hwc(s.contiguous()) for s in g["src_fea"].to(dev)])
    return ref, src_hwc, g["mats"].contiguous(), g["hyp"][0].to(dev).contiguous()


def test_chw_to_hwc(dev, ops):
    x = torch.randn(16, 24, 40, device=dev)
    assert torch.equal(ops.chw_to_hwc(x), x.permute(1, 2, 0).contiguous())


def test_homo_warp_matches_reference_grid_sample(dev, ops):
    g = load_golden("g1_warp_aggregate_a")
    _, src_hwc, mats, hyp = _stage_inputs(g, dev, ops)
    out = ops.homo_warp(src_hwc[0], mats[0], hyp).cpu()
    diff = (out - g["warped0"]).abs()
    assert diff.max() < 1e-6, diff.max()
    assert (out != g["warped0"]).float().mean() < 1e-3  # op order is pinned: almost every voxel is bit-identical


def test_homo_warp_plane_hypotheses(dev, ops):
    """[D] (per-plane) hypotheses == the same planes broadcast per pixel (warping.py accepts both)."""
    g = load_golden("g1_warp_aggregate_a")
    _, src_hwc, mats, _ = _stage_inputs(g, dev, ops)
    planes = torch.linspace(425, 900, 12, device=dev)
    h, w, _ = src_hwc[0].shape
    a = ops.homo_warp(src_hwc[0], mats[0], planes)
    b = ops.homo_warp(src_hwc[0], mats[0], planes.view(-1, 1, 1).expand(-1, h, w).contiguous())
    assert torch.equal(a, b)


# Both sample-position modes of the LDS-staged K1 / K3 kernels are held to the reference goldens at the SAME tolerances:
# exact=True is the product default (the reference's fp32 operation order, ops.WARP_EXACT) and must in addition stay at round-off
# level (bit-identical positions); exact=False is the opt-in fast form (CDS_WARP_FAST=1).
POSITION_MODES = [pytest.param(False, id="fast"), pytest.param(True, id="exact")]


@pytest.mark.parametrize("exact", POSITION_MODES)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_warp_entropy(tag, exact, dev, ops):
    g = load_golden(f"g1_warp_aggregate_{tag}")
    ref, src_hwc, mats, hyp = _stage_inputs(g, dev, ops)
    ent = ops.warp_entropy(ref, src_hwc, mats, hyp, exact=exact).cpu()
    assert (ent - g["entropy"]).abs().max() < (5e-6 if exact else 2e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_visibility_cnn(tag, dev, seeded_state):
    g = load_golden(f"g1_warp_aggregate_{tag}")
    model = seeded_state(False).to(dev)
    vis = model.stage_net.visibility(g["entropy"].to(dev), g["ref_nc"][:, 0].to(dev).contiguous(), int(g["stage"])).cpu()
    assert (vis - g["vis_w"]).abs().max() < 1e-5


@pytest.mark.parametrize("exact", POSITION_MODES)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_warp_aggregate(tag, exact, dev, ops):
    g = load_golden(f"g1_warp_aggregate_{tag}")
    ref, src_hwc, mats, hyp = _stage_inputs(g, dev, ops)
    vol, vis_sum = ops.warp_aggregate(ref, src_hwc, g["vis_w"].to(dev).contiguous(), mats, hyp, exact=exact)
    diff = (vol.cpu() - g["volume_mean"]).abs()
    assert diff.max() < (2e-6 if exact else 1e-5), diff.max()
    assert (vis_sum.cpu() - g["vis_w"].sum(0)).abs().max() < 1e-6


@pytest.mark.parametrize("tag", ["a", "c"])
def test_view_shards_sum_to_unsharded(tag, dev, ops):
    """partial volume sums over disjoint view subsets add up (fp32 re-association only)."""
    g = load_golden(f"g1_warp_aggregate_{tag}")
    ref, src_hwc, mats, hyp = _stage_inputs(g, dev, ops)
    vis = g["vis_w"].to(dev).contiguous()
    full, _ = ops.warp_aggregate(ref, src_hwc, vis, mats, hyp)
    V = ref.shape[0]
    parts = [ops.warp_aggregate(ref[v:v + 1], src_hwc[v:v + 1], vis[v:v + 1], mats[v:v + 1].contiguous(), hyp,
                                normalize=False) for v in range(V)]
    vol = sum(p[0] for p in parts)
    vs = sum(p[1] for p in parts)
    ops.volume_normalize_(vol, vs)
    assert (vol - full).abs().max() < 1e-6
    # accumulate flag: second call adds onto the first
    v0, s0 = ops.warp_aggregate(ref[:1], src_hwc[:1], vis[:1], mats[:1].contiguous(), hyp, normalize=False)
    v1, s1 = ops.warp_aggregate(ref[1:], src_hwc[1:], vis[1:], mats[1:].contiguous(), hyp, normalize=True, volume=v0,
                                vis_sum=s0, accumulate=True)
    assert (v1 - full).abs().max() < 1e-6


@pytest.mark.parametrize("tag,C", [("c8", 8), ("c16", 16), ("c32", 32), ("c8_wide", 8)])
def test_costreg(tag, C, dev):
    from cds_mvsnet_amd import CostRegNet, seeded_init_
    g = load_golden(f"g2_costreg_{tag}")
    net = CostRegNet(C, 8)
    seeded_init_(net, int(g["seed"]))
    net = net.to(dev).eval()
    out = net(g["volume"].to(dev).contiguous()).cpu()               # planar volume: exact-fp32 kernels (fmaf chains)
    err = (out - g["cost_reg"]).abs().max()
    assert err < 1e-4, err
    # channels-last volume: the split-bf16 matrix-core kernels (what the model runs); same golden, same tolerance, and
    # the two arithmetic paths agree with each other far inside it
    assert net.split_bf16_supported()
    out_cl = net(g["volume"].permute(1, 2, 3, 0).contiguous().to(dev), channels_last=True).cpu()
    err_cl = (out_cl - g["cost_reg"]).abs().max()
    assert err_cl < 1e-4, err_cl
    assert (out_cl - out).abs().max() < 2e-5
    # ... and with conv0 - conv3 in split-f16 (what the model runs since round 6: a bound of the volume's magnitudes is handed in)
    vol_cl = g["volume"].permute(1, 2, 3, 0).contiguous().to(dev)
    out_h = net(vol_cl, channels_last=True, bound=vol_cl.abs().amax().reshape(1)).cpu()
    err_h = (out_h - g["cost_reg"]).abs().max()
    assert err_h < 1e-4, err_h
    from cds_mvsnet_amd import ops as _ops
    assert (out_h - out).abs().max() < 2e-5
    assert not torch.equal(out_h, out_cl) or not _ops.USE_SPLIT_F16         # (not equal: the split-f16 layers really ran)
    print(f"CostRegNet {tag}: max |HIP - reference| exact-fp32 {err:.2e}, split-bf16 {err_cl:.2e}, split-f16 conv0-3 {err_h:.2e}")


def test_conv3d_layers_vs_torch(dev, ops):
    """Every conv flavour of K4 against the plain fp32 torch op on the CPU (odd sizes exercise the tile edges)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    for (cin, cout, D, H, W, stride) in [(8, 8, 5, 9, 70, 1), (8, 16, 6, 10, 22, 2), (16, 16, 3, 7, 13, 1),
                                          (32, 64, 4, 6, 10, 2), (8, 1, 4, 9, 50, 1), (64, 64, 1, 2, 3, 1),
                                          # W % 4 == 0 and wide: the register-prefetch "pipe" kernels (partial tiles in x/y/z)
                                          (8, 8, 5, 9, 64, 1), (6, 16, 3, 5, 100, 1), (8, 16, 6, 10, 72, 2),
                                          (3, 8, 9, 7, 128, 2), (8, 1, 4, 9, 68, 1), (16, 32, 2, 3, 36, 1),
                                          # Cout % 16 == 0, Cin % 4 == 0: the fp32-MFMA implicit-GEMM kernels
                                          (8, 16, 5, 9, 64, 1), (64, 64, 3, 5, 40, 1), (32, 64, 9, 7, 128, 2),
                                          (4, 48, 2, 6, 76, 2), (12, 16, 7, 3, 132, 1), (32, 64, 6, 16, 40, 2), (64, 64, 3, 8, 20, 1),
                                          (16, 16, 2, 4, 12, 1),
                                          # same kernels with unaligned rows (W % 4 != 0): the 50-wide deepest level of a
                                          # 1600-wide scene and friends
                                          (64, 64, 6, 37, 50, 1), (32, 64, 5, 9, 50, 2), (16, 16, 3, 5, 18, 1), (8, 16, 2, 3, 33, 1)]:
        x = torch.randn(cin, D, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        ref = F.relu(F.conv3d(x[None], w, b, stride=stride, padding=1))[0]
        skip = torch.randn(ref.shape, generator=g)
        wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().to(dev)
        out = ops.conv3d_k3(x.to(dev), wpk, b.to(dev), stride=stride, relu=True, skip=skip.to(dev)).cpu()
        assert (out - (ref + skip)).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (cin, cout, stride)
    for (cin, cout, D, H, W) in [(16, 8, 3, 5, 50), (64, 32, 1, 2, 3), (32, 16, 2, 6, 9), (16, 8, 3, 5, 64),
                                 (10, 16, 2, 3, 36), (32, 16, 3, 3, 100), (64, 32, 3, 5, 40), (8, 48, 1, 6, 68), (64, 32, 3, 8, 20), (32, 16, 2, 4, 8),
                                 (64, 32, 6, 37, 50), (32, 16, 3, 5, 18), (16, 16, 2, 4, 10),   # W even, not a multiple of 4
                                 (32, 32, 8, 40, 132), (16, 32, 10, 44, 66)]:   # >= 256 workgroups: the four-classes-per-workgroup kernel
        x = torch.randn(cin, D, H, W, generator=g)
        w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        ref = F.relu(F.conv_transpose3d(x[None], w, b, stride=2, padding=1, output_padding=1))[0]
        skip = torch.randn(ref.shape, generator=g)
        wpk = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous().to(dev)
        out = ops.deconv3d_k3s2(x.to(dev), wpk, b.to(dev), relu=True, skip=skip.to(dev)).cpu()
        assert (out - (ref + skip)).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (cin, cout)


def test_conv2d_vs_torch(dev, ops):
    import torch.nn.functional as F
    from cds_mvsnet_amd.model import _pack2d
    g = torch.Generator().manual_seed(1)
    for (n, cin, cout, H, W, k, s) in [(1, 3, 11, 37, 70, 11, 1), (2, 8, 11, 20, 33, 7, 1), (1, 16, 19, 18, 30, 5, 1),
                                       (3, 2, 16, 16, 24, 3, 1), (1, 8, 16, 21, 35, 3, 2), (1, 48, 16, 9, 14, 1, 1),
                                       (2, 16, 1, 9, 14, 1, 1),
                                       # W % 4 == 0: the aligned, register-prefetched "pipe" kernels (partial chunks, tile
                                       # edges, exact widths 11 / 19 / 35, stride 2, a 4-pixel-wide image)
                                       (2, 8, 11, 37, 72, 3, 1), (1, 3, 11, 20, 68, 7, 1), (1, 16, 19, 18, 36, 5, 1),
                                       (1, 32, 35, 19, 44, 3, 1), (1, 8, 16, 21, 36, 3, 2), (2, 16, 32, 17, 132, 3, 2),
                                       (1, 8, 11, 9, 4, 3, 1), (1, 6, 11, 33, 200, 5, 1), (1, 16, 24, 16, 64, 3, 1)]:
        x = torch.randn(n, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        pad = (k - 1) // 2
        ref = F.conv2d(x, w, b, stride=s, padding=pad)
        out = ops.conv2d(x.to(dev), _pack2d(w).to(dev), b.to(dev), cout, k, s, pad).cpu()
        assert out.shape == ref.shape
        assert (out - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (cin, cout, k, s)


@pytest.mark.parametrize("tag", ["d48", "d8", "d192"])
def test_softargmin_conf(tag, dev, ops):
    g = load_golden(f"g3_regress_{tag}")
    depth, conf, prob = ops.softargmin_conf(g["prob_pre"].to(dev), g["hyp"].to(dev), want_prob=True)
    assert (prob.cpu() - g["prob"]).abs().max() < 1e-6
    assert (depth.cpu() - g["depth"]).abs().max() < 5e-4
    # the gathered window index is floor(sum p*idx): a 1-ulp difference can flip it for isolated pixels
    bad = ((conf.cpu() - g["conf"]).abs() > 1e-5).float().mean()
    assert bad <= 0.01, bad
    # edge rows: peaks at d=0 / d=D-1 (window clipped by the zero padding) must match exactly enough
    assert (conf.cpu()[:5, :8] - g["conf"][:5, :8]).abs().max() < 1e-5


def test_depth_hypotheses(dev, ops):
    g = load_golden("g4_hypotheses")
    dv = g["depth_values"]
    H, W = int(g["H"]), int(g["W"])
    dmin, dmax, dint = float(dv[0, 0]), float(dv[0, -1]), dv[0, 1] - dv[0, 0]
    s1 = ops.depth_planes(48, H // 4, W // 4, dmin, dmax, dev).cpu()
    assert torch.equal(s1, g["s1"][0])
    for tag in ("s2", "s3", "s2x4"):
        D, ratio, scale = g[tag + "_meta"]
        interval = float(float(ratio) * dint)
        out = ops.depth_hypotheses(g[tag + "_prev"][0].to(dev).contiguous(), int(D), H, W, int(scale), interval, dmin,
                                   dmax).cpu()
        diff = (out - g[tag][0]).abs()
        assert diff.max() < 1e-4, (tag, diff.max())
        assert (out != g[tag][0]).float().mean() < 0.01, tag  # op order pinned: (nearly) bit-identical


def test_dynconv_and_epipoles(dev, ops, seeded_state):
    from cds_mvsnet_amd import geometry, seeded_init_
    from cds_mvsnet_amd.model import DynamicConv, FeatureNet, _FeatureRunner
    g = load_golden("g5_dynconv")
    e_ref, e_src = geometry.pair_epipoles(g["cams"][0, 0], g["cams"][0, 1])
    assert abs(e_ref[0] - float(g["epipole_ref"][0, 0])) <= 1e-3 * abs(e_ref[0]) + 1e-3
    assert abs(e_src[1] - float(g["epipole_src"][0, 1])) <= 1e-3 * abs(e_src[1]) + 1e-3
    # a single K=3 DynamicConv with bias through conv2d + blend
    dc = DynamicConv(3, 8, (3, 7, 11))
    seeded_init_(dc, 7)
    dc = dc.to(dev).eval()

    class _Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dc = dc
    runner = _FeatureRunner.__new__(_FeatureRunner)
    packed = {}
    for i, k in enumerate(dc.size_kernels):
        from cds_mvsnet_amd.model import _pack2d, _bn_fold
        packed[f"dc.w{i}"] = _pack2d(torch.cat((dc.convs[i].weight.detach(), dc.att_convs[i].weight.detach()), 0))
        packed[f"dc.b{i}"] = torch.cat((dc.convs[i].bias.detach(), torch.zeros(3, device=dev))).contiguous()
    scale, shift = _bn_fold(dc.att_weights[1])
    packed["dc.m1"] = (dc.att_weights[0].weight.detach().reshape(4, 3) * scale.view(4, 1)).contiguous()
    packed["dc.mb"] = shift.contiguous()
    packed["dc.m2"] = dc.att_weights[3].weight.detach().reshape(3, 4).contiguous()
    # batch of two images with different epipoles: row 0 must match the golden, row 1 must differ from it
    epi = torch.cat((g["epipole_ref"], g["epipole_src"])).contiguous()
    x2 = torch.stack((g["img"], g["img"])).to(dev).contiguous()
    for T in (1.0, 0.1, 0.01):
        y, nc = runner._dynamic(packed, "dc", dc, x2, epi, T, stats_slope=None)
        assert (y[0].cpu() - g[f"y_T{T}"]).abs().max() < 5e-5, T
        # the statistics-carrying variant (4 pixels per thread) gives the same tensors, exact fp64 sums of its output and
        # the normalisation table / normalised tensor of the two-pass path
        y2, nc2, st, aff = runner._dynamic(packed, "dc", dc, x2, epi, T, stats_slope=0.1)
        assert torch.equal(y2, y) and torch.equal(nc2, nc)
        yd = y.double()
        want = torch.stack((yd.sum((2, 3)), (yd * yd).sum((2, 3))), dim=-1)
        assert ((st - want).abs() / want.abs().clamp_min(1.0)).max() < 1e-12
        assert (aff - ops.instnorm_affine(y, 0.1)).abs().max() < 1e-6
        for hwc in (False, True):
            assert (ops.instnorm_apply(y, st, 4, out_hwc=hwc) - ops.instnorm_act(y, 4, out_hwc=hwc)).abs().max() < 1e-6
        again = runner._dynamic(packed, "dc", dc, x2, epi, T, stats_slope=0.1)[2]
        assert torch.equal(again, st)                                       # fixed summation order: bit-reproducible
        assert (nc[0].cpu() - g[f"nc_T{T}"]).abs().max() < 5e-5, T
        assert (nc[1] - nc[0]).abs().max() > 1e-4


@pytest.mark.parametrize("layout", ["channels_last", "planar"])
def test_featurenet(layout, dev, seeded_state, monkeypatch):
    """FeatureNet against the reference golden G5 and the float64 envelope G9, on the channels-last kernels (feat_cl.hip, the
    default since round 5) and on the planar kernels of rounds 1-4 (CDS_FEAT_CL=0)."""
    from cds_mvsnet_amd import FeatureNet, seeded_init_
    import cds_mvsnet_amd.model as cm
    from cds_mvsnet_amd.model import _FeatureRunner
    monkeypatch.setattr(cm, "USE_FEAT_CL", layout == "channels_last")
    g = load_golden("g5_featurenet")
    net = FeatureNet(8)
    seeded_init_(net, 7)
    net = net.to(dev).eval()
    run = _FeatureRunner(net)
    epi = g["epipole"].contiguous()
    img = g["img"].to(dev).unsqueeze(0).contiguous()
    gn = load_golden("g9_featurenet_noise")
    worst = {}
    for T in (1.0, 0.01):
        o = run(img, epi, T)                                   # one image, CHW
        o2 = run(torch.cat((img, img)), torch.cat((epi, epi)), T, n_chw=1)   # same image twice: CHW + HWC
        out = {s: (o[s][0][0], o[s][2][0], o[s][3][0]) for s in o}
        out_hwc = {s: (o2[s][1][0],) for s in o2}
        # Tolerances are DERIVED from the reference's own fp32 round-off on this input (g9_featurenet_noise, captured by
        # make_golden.py): the module in float64 is the exact answer, and the reference's two fp32 evaluations of it
        # (oneDNN / native convolutions) miss it by up to 1.5e-4 at T = 0.01, where the softmax(./T) blend amplifies
        # convolution round-off by up to 0.25 / T per layer (2.5e-5 at T = 1).  The HIP path must sit inside 3x that
        # envelope around the float64 answer; the mean error is bounded at the fp32 round-off level as before.
        for s in ("stage1", "stage2", "stage3"):
            for j, key in enumerate(("fea", "ncsum", "nc")):
                k = f"{s}_{key}_T{T}"
                got = out[s][j].cpu()
                err32 = (got - g[k]).abs()
                err64 = (got - gn[k + "_f64"]).abs()
                envelope = max(float(gn[k + "_ref32_vs_f64_max"]), float(gn[k + "_native32_vs_f64_max"]))
                scale = max(1.0, g[k].abs().max().item())
                assert err32.mean() < 2e-5 * scale, (s, key, T, err32.mean())
                assert err64.max() <= 3.0 * envelope, (s, key, T, float(err64.max()), envelope)
                worst[k] = (float(err64.max()), envelope)
            assert torch.equal(out_hwc[s][0].permute(2, 0, 1), out[s][0])
    print("FeatureNet max |HIP - float64 reference| vs the reference's own fp32 envelope:",
          {k: f"{a:.1e} / {b:.1e}" for k, (a, b) in worst.items()})


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_stage_net_reference_signature(tag, dev, seeded_state):
    """StageNet.forward with the reference's own call signature (model.py:16) vs the reference output."""
    g = load_golden(f"g1_warp_aggregate_{tag}")
    model = seeded_state(False).to(dev)
    stage = int(g["stage"])
    V = g["ref_fea"].shape[0]
    feats = [{"ref": (g["ref_fea"][v:v + 1].to(dev), g["ref_nc_sum"][v:v + 1].to(dev), g["ref_nc"][v:v + 1].to(dev)),
              "src": (g["src_fea"][v:v + 1].to(dev), g["src_nc_sum"][v:v + 1].to(dev), None)} for v in range(V)]
    hyp = g["hyp"].to(dev)
    out = model.stage_net(feats, g["cams"].to(dev), depth_values=hyp, num_depth=hyp.shape[1],
                          cost_regularization=model.cost_regularization[stage], stage_idx=stage)
    assert (out["depth"].cpu() - g["depth"]).abs().mean() < 1e-3
    assert (out["photometric_confidence"].cpu() - g["conf"]).abs().mean() < 1e-3
    assert (out["norm_curv"].cpu() - g["norm_curv"]).abs().max() < 1e-6


@pytest.mark.parametrize("tag,refine", [("norefine", False), ("refine", True)])
def test_full_forward(tag, refine, dev, seeded_state):
    g = load_golden(f"g6_forward_{tag}")
    model = seeded_state(refine).to(dev)
    cams = {k[4:]: v.to(dev) for k, v in g.items() if k.startswith("cam_")}
    with torch.no_grad():
        out = model(g["imgs"].to(dev), cams, g["depth_values"].to(dev), temperature=0.01)
    assert set(out.keys()) >= {"stage1", "stage2", "stage3", "depth", "photometric_confidence", "norm_curv",
                               "refined_depth"}
    for s in (1, 2, 3):
        st = out[f"stage{s}"]
        l1 = (st["depth"].cpu() - g[f"stage{s}_depth"]).abs().mean().item()
        assert l1 < 1e-3, (s, l1)
        assert (st["photometric_confidence"].cpu() - g[f"stage{s}_conf"]).abs().mean() < 1e-3, s
        assert (st["norm_curv"].cpu() - g[f"stage{s}_norm_curv"]).abs().max() < 1e-4, s
    assert (out["refined_depth"].cpu() - g["refined_depth"]).abs().mean() < 1e-3
    assert out["depth"].shape == g["stage3_depth"].shape


def test_oracle_parity_midsize(dev, ops, seeded_state):
    """Seeded 64x80, D=96, C=8, N=5 stage against the CPU oracle (sizes the goldens do not cover)."""
    from cds_mvsnet_amd import synth
    from oracle import cds_oracle as O
    h, w, D, C, N = 64, 80, 96, 8, 5
    model = seeded_state(False)
    sd = model.state_dict()
    feats = synth.make_pair_features(N - 1, C, h, w, seed=21, sharp=True)
    cams = synth.stage_cameras(N, h, w, seed=22)
    hyp = synth.make_hypotheses(D, h, w, seed=23)
    ref_out = O.stage_forward(feats, cams, hyp, sd, 2, exact=False)
    model = model.to(dev)
    dfe = [{k: tuple(t.to(dev) if t is not None else None for t in f[k]) for k in ("ref", "src")} for f in feats]
    out = model.stage_net(dfe, cams.to(dev), depth_values=hyp.to(dev), num_depth=D,
                          cost_regularization=model.cost_regularization[2], stage_idx=2)
    assert (out["depth"].cpu() - ref_out["depth"]).abs().mean() < 1e-3
    assert (out["photometric_confidence"].cpu() - ref_out["photometric_confidence"]).abs().mean() < 1e-3


@pytest.mark.parametrize("name,h,w,D,C,N", [("M1b", 128, 160, 192, 32, 5), ("M1 crop", 256, 320, 192, 8, 5),
                                            ("M1 full size (the headline workload: 640x512, D=192, C=8, N=5)", 512, 640, 192, 8, 5)])
def test_oracle_parity_large_depth_range(name, h, w, D, C, N, dev, ops, seeded_state):
    """the D = 192 single-stage workloads against the CPU oracle under an assertion (until round 5 only bench.py's
    cpu_baseline compared them): M1b (BASELINE config 2 read as the 160x128, C = 32 grid) at full size, and a 320x256 window of M1
    (640x512, C = 8) - the same D = 192 chunking of K1 / K3 (48-plane LDS chunks, CDS_K3_NSEG depth segments) and the same
    CostRegNet kernels as the full size at a quarter of the oracle's run time - and, since round 6, M1 itself at full
    size: the workload bench.py's headline number is quoted on is under an assertion, not only under bench.py's `abs_depth_l1_vs_gpu`
    (~20 s and ~20 GB of the oracle on the box's host cores).  Tolerances are SURVEY 8(c)'s: aggregated volume <= 1e-5 abs, depth
    mean-L1 <= 1e-3, confidence mean <= 1e-3."""
    from cds_mvsnet_amd import synth
    from oracle import cds_oracle as O
    model = seeded_state(False)
    sd = model.state_dict()
    stage = {8: 2, 16: 1, 32: 0}[C]
    feats = synth.make_pair_features(N - 1, C, h, w, seed=31)
    cams = synth.stage_cameras(N, h, w, seed=32)
    hyp = synth.make_hypotheses(D, h, w, seed=33)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        want = O.stage_forward(feats, cams, hyp, sd, stage, exact=False)
    model = model.to(dev)
    dfe = [{k: tuple(t.to(dev) if t is not None else None for t in f[k]) for k in ("ref", "src")} for f in feats]
    with torch.no_grad():
        out = model.stage_net(dfe, cams, depth_values=hyp.to(dev), num_depth=D, cost_regularization=model.cost_regularization[stage],
                              stage_idx=stage)
        # the aggregated volume itself (K1 -> visibility CNN -> K3), planar
        from cds_mvsnet_amd import geometry
        ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
        src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
        ref_nc = torch.stack([f["ref"][2][0, 0] for f in feats]).to(dev).contiguous()
        vol, _, _, _ = model.stage_net.aggregate(ref, src, ref_nc, geometry.warp_matrices(cams[0]), hyp[0].to(dev).contiguous(), stage)
    assert (vol.cpu() - want["_volume_mean"][0]).abs().max().item() <= 1e-5, name
    assert (out["depth"].cpu() - want["depth"]).abs().mean().item() <= 1e-3, name
    assert (out["photometric_confidence"].cpu() - want["photometric_confidence"]).abs().mean().item() <= 1e-3, name
    assert (out["norm_curv"].cpu() - want["norm_curv"]).abs().max().item() <= 1e-6, name


def test_cpu_tensors_fail_loudly(ops):
    with pytest.raises(RuntimeError):
        ops.chw_to_hwc(torch.zeros(8, 4, 4))


# ------------------------------------------------------------------------------------------------
# edge cases and size-independent properties (shapes the goldens do not cover, up to BASELINE's full M1 size)
# ------------------------------------------------------------------------------------------------
def _random_stage(ops, dev, V, C, D, h, w, seed, sharp=True):
    from cds_mvsnet_amd import geometry, synth
    feats = synth.make_pair_features(V, C, h, w, seed=seed, sharp=sharp)
    cams = synth.stage_cameras(V + 1, h, w, seed=seed + 1)
    hyp = synth.make_hypotheses(D, h, w, seed=seed + 2)
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
    return feats, cams, hyp, ref, src, geometry.warp_matrices(cams[0]), hyp[0].to(dev).contiguous()


@pytest.mark.parametrize("V,C,D,h,w", [(1, 8, 7, 9, 70), (4, 8, 33, 12, 130), (6, 8, 5, 16, 24), (3, 16, 9, 10, 50),
                                       (2, 32, 4, 8, 66), (4, 8, 2, 5, 3),
                                       # C = 8 with 2 / 3 views: the LDS kernels are specialised per view count; D = 70
                                       # spans three chunks (32 + 32 + 6) and an odd plane pair at the end
                                       (2, 8, 70, 11, 67), (3, 8, 35, 6, 129),
                                       # 5 - 7 views (BASELINE config 4: N = 7): two launches, the second accumulating onto the first
                                       (5, 8, 50, 20, 90), (6, 32, 10, 16, 40), (6, 16, 49, 24, 72), (7, 8, 12, 12, 70)])
@pytest.mark.parametrize("exact", POSITION_MODES)
def test_warp_kernels_vs_oracle_odd_shapes(V, C, D, h, w, exact, dev, ops):
    """Odd D / widths that are not tile multiples / V = 1, 5, 6, 7 (more than 4: two launches) / C = 16, 32 / tiny
    images: K1 and K3 against the CPU oracle (explicit fp32 gather), both position modes."""
    from oracle import cds_oracle as O
    feats, cams, hyp, ref, src, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=40 + V + C)
    vis = torch.rand(V, h, w, generator=torch.Generator().manual_seed(1)) * 0.8 + 0.1
    ent = ops.warp_entropy(ref, src, mats, hyp_d, exact=exact).cpu()
    vol, vis_sum = ops.warp_aggregate(ref, src, vis.to(dev).contiguous(), mats, hyp_d, exact=exact)
    P_ref = O.compose_projection(cams[:, 0])
    want_vol, want_ent = 0.0, []
    for v in range(V):
        warped = O.warp_volume(feats[v]["src"][0], O.compose_projection(cams[:, v + 1]), P_ref, hyp)
        in_prod, e = O.correlation_entropy(feats[v]["ref"][0], warped)
        want_vol = want_vol + in_prod * vis[v].view(1, 1, 1, h, w)
        want_ent.append(e[0, 0])
    want_vol = want_vol[0] / (vis.sum(0).view(1, 1, h, w) + 1e-6)
    # fast positions differ from the reference's by ~1e-7 w px: on these white-noise-sharp maps that is up to ~3e-5 in the volume
    assert (vol.cpu() - want_vol).abs().max() < (1e-5 if exact else 5e-5)
    assert (ent - torch.stack(want_ent)).abs().max() < (5e-5 if exact else 2e-4)
    assert (vis_sum.cpu() - vis.sum(0)).abs().max() < 1e-6


@pytest.mark.parametrize("V,C,D,h,w", [(6, 16, 49, 24, 72), (7, 8, 13, 12, 70), (5, 32, 7, 16, 40)])
def test_warp_aggregate_two_launches_channels_last_equals_planar(V, C, D, h, w, dev, ops):
    """More than four source views = two launches, the second one reading the first one's partial sums one plane pair ahead (round 6).
    The channels-last volume (what CostRegNet's matrix-core kernels read; the layout of the cascade forward at N = 7) must be the planar
    one bit for bit, also for an odd number of planes (a last pair with one plane) and chunks shorter than a pair's look-ahead."""
    feats, cams, hyp, ref, src, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=70 + V)
    vis = (torch.rand(V, h, w, generator=torch.Generator().manual_seed(3)) * 0.8 + 0.1).to(dev).contiguous()
    planar, vs0 = ops.warp_aggregate(ref, src, vis, mats, hyp_d, exact=True)
    cl, vs1 = ops.warp_aggregate(ref, src, vis, mats, hyp_d, exact=True, channels_last=True)
    assert torch.equal(cl.permute(3, 0, 1, 2), planar) and torch.equal(vs0, vs1)


def test_fast_positions_leave_the_parity_tolerance_at_full_width(dev, ops):
    """Why the reference-order positions are the default: at w = 640 the fast form p.xy * rcp(z) and the
    reference's normalise / de-normalise round trip place the samples up to ~1e-4 px apart; on sharp feature maps that moves
    the volume past the 1e-5 parity tolerance, while the exact mode stays at round-off level against the CPU oracle."""
    from oracle import cds_oracle as O
    V, C, D, h, w = 2, 8, 6, 16, 640
    feats, cams, hyp, ref, src, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=11)
    vis = (torch.rand(V, h, w, generator=torch.Generator().manual_seed(1)) * 0.8 + 0.1)
    P_ref = O.compose_projection(cams[:, 0])
    want = 0.0
    for v in range(V):
        warped = O.warp_volume(feats[v]["src"][0], O.compose_projection(cams[:, v + 1]), P_ref, hyp)
        want = want + O.correlation_entropy(feats[v]["ref"][0], warped)[0] * vis[v].view(1, 1, 1, h, w)
    want = want[0] / (vis.sum(0).view(1, 1, h, w) + 1e-6)
    err = {}
    for exact in (True, False):
        vol, _ = ops.warp_aggregate(ref, src, vis.to(dev).contiguous(), mats, hyp_d, exact=exact)
        err[exact] = (vol.cpu() - want).abs().max().item()
    print("volume max-abs vs oracle at w=640: exact", err[True], "fast", err[False])
    assert err[True] < 2e-6
    assert err[False] < 1e-3          # bounded (a position error, not a bug) ...
    assert err[False] > err[True]     # ... but not at round-off level


def test_warp_paths_agree_lds_vs_direct(dev, ops):
    """The LDS-staged kernels and the direct (L1 gather) kernels are two implementations of the same arithmetic:
    identical positions / interpolation, accumulation differs by one rounding per view."""
    import os, subprocess, sys
    code = ("import torch,sys; sys.path.insert(0,'.'); from cds_mvsnet_amd import ops, synth, geometry;"
            "dev=torch.device('cuda:0'); V,C,D,h,w=4,8,40,64,200;"
            "f=synth.make_pair_features(V,C,h,w,seed=9,sharp=True); cams=synth.stage_cameras(V+1,h,w,seed=8);"
            "hyp=synth.make_hypotheses(D,h,w,seed=7)[0].to(dev);"
            "ref=torch.stack([x['ref'][0][0] for x in f]).to(dev).contiguous();"
            "src=torch.stack([ops.chw_to_hwc(x['src'][0][0].to(dev).contiguous()) for x in f]);"
            "vis=torch.rand(V,h,w,generator=torch.Generator().manual_seed(3)).to(dev);"
            "m=geometry.warp_matrices(cams[0]); e=ops.warp_entropy(ref,src,m,hyp); v,_=ops.warp_aggregate(ref,src,vis,m,hyp);"
            "torch.save((e.cpu(),v.cpu()), sys.argv[1])")
    outs = []
    for flag in ("0", "1"):
        path = f"/tmp/cds_paths_{flag}.pt"
        env = dict(os.environ, CDS_WARP_DIRECT=flag, CDS_WARP_EXACT="1")   # the direct kernels always use the reference order
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
        outs.append(torch.load(path))
    assert (outs[0][0] - outs[1][0]).abs().max() < 2e-5
    assert (outs[0][1] - outs[1][1]).abs().max() < 2e-6


def test_per_plane_hypotheses_equal_broadcast(dev, ops):
    feats, cams, hyp, ref, src, mats, _ = _random_stage(ops, dev, 2, 8, 6, 16, 40, seed=77)
    planes = torch.linspace(430, 890, 6, device=dev)
    full = planes.view(-1, 1, 1).expand(-1, 16, 40).contiguous()
    vis = torch.rand(2, 16, 40, device=dev)
    # [D] hypotheses run on the direct kernels (always the reference-order positions): compare like with like
    assert (ops.warp_entropy(ref, src, mats, planes) - ops.warp_entropy(ref, src, mats, full, exact=True)).abs().max() < 2e-5
    assert (ops.warp_aggregate(ref, src, vis, mats, planes)[0] - ops.warp_aggregate(ref, src, vis, mats, full, exact=True)[0]).abs().max() < 2e-6
    assert (ops.warp_aggregate(ref, src, vis, mats, planes)[0] - ops.warp_aggregate(ref, src, vis, mats, full, exact=False)[0]).abs().max() < 1e-5


def test_full_size_M1_properties(dev, ops):
    """BASELINE's full single-stage size (640x512, D=192, C=8, N=5; 2 GB volume) through size-independent properties:
    (1) the visibility-weighted mean is invariant to a common scale of the weights; (2) the partial sums of two view
    shards add up to the unsharded volume; (3) |volume| <= max|ref|*max|src| (convex combination of products);
    (4) entropy in [0, log D]; (5) depth inside the hypothesis range and confidence in [0, 1] after CostRegNet."""
    import math
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_
    V, C, D, h, w = 4, 8, 192, 512, 640
    feats, cams, hyp, ref, src, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=5, sharp=False)
    vis = (torch.rand(V, h, w, generator=torch.Generator().manual_seed(2)) * 0.9 + 0.05).to(dev)
    ent = ops.warp_entropy(ref, src, mats, hyp_d)
    assert ent.min() >= -1e-5 and ent.max() <= math.log(D) + 1e-4
    vol, _ = ops.warp_aggregate(ref, src, vis, mats, hyp_d)
    bound = ref.abs().max() * src.abs().max()
    assert vol.abs().max() <= bound * (1 + 1e-5)
    vol2, _ = ops.warp_aggregate(ref, src, (vis * 0.5).contiguous(), mats, hyp_d)      # (1) exact: scaling by 0.5
    assert (vol - vol2).abs().max() < 2e-6
    del vol2
    pa, sa = ops.warp_aggregate(ref[:2], src[:2], vis[:2].contiguous(), mats[:2].contiguous(), hyp_d, normalize=False)
    pb, sb = ops.warp_aggregate(ref[2:], src[2:], vis[2:].contiguous(), mats[2:].contiguous(), hyp_d, normalize=False)
    pa += pb
    del pb
    ops.volume_normalize_(pa, sa + sb)                                               # (2)
    assert (pa - vol).abs().max() < 1e-6
    del pa
    model = seeded_init_(CDSMVSNet(), 0).eval().to(dev)
    prob_pre = model.cost_regularization[2](vol)
    depth, conf = ops.softargmin_conf(prob_pre, hyp_d)
    assert torch.isfinite(depth).all() and depth.min() >= hyp_d.min() - 1e-3 and depth.max() <= hyp_d.max() + 1e-3
    assert conf.min() >= 0 and conf.max() <= 1 + 1e-5
    # CostRegNet is translation covariant along x away from the borders: shift the volume by 8 voxels
    shifted = model.cost_regularization[2](torch.roll(vol, 8, dims=3))
    assert (shifted[:, :, 40:-40] - torch.roll(prob_pre, 8, dims=2)[:, :, 40:-40]).abs().max() < 1e-4


def test_model_input_validation(dev):
    from cds_mvsnet_amd import CDSMVSNet, synth
    m = CDSMVSNet().eval().to(dev)
    with pytest.raises(ValueError):
        m(synth.make_images(3, 72, 96).to(dev), {k: v.to(dev) for k, v in synth.make_cameras(3, 72, 96).items()},
          synth.make_depth_values().to(dev))   # 72 is not a multiple of 32


def test_batch_of_two_equals_two_singles(dev, seeded_state):
    """The reference forward is batched (B > 1 in training / DataParallel); each batch item must equal its own run."""
    from cds_mvsnet_amd import synth
    model = seeded_state(False).to(dev)
    a = (synth.make_images(3, 64, 96, seed=1), synth.make_cameras(3, 64, 96, seed=1))
    b = (synth.make_images(3, 64, 96, seed=2), synth.make_cameras(3, 64, 96, seed=2))
    dv = synth.make_depth_values()
    imgs = torch.cat((a[0], b[0])).to(dev)
    cams = {k: torch.cat((a[1][k], b[1][k])).to(dev) for k in a[1]}
    dv2 = torch.cat((dv, dv + 10.0)).to(dev)
    with torch.no_grad():
        both = model(imgs, cams, dv2, temperature=0.01)
        one = model(b[0].to(dev), {k: v.to(dev) for k, v in b[1].items()}, (dv + 10.0).to(dev), temperature=0.01)
    assert both["depth"].shape == (2, 64, 96) and both["stage1"]["norm_curv"].shape == (2, 1, 16, 24)
    assert torch.equal(both["depth"][1], one["depth"][0])
    assert torch.equal(both["photometric_confidence"][1], one["photometric_confidence"][0])


@pytest.mark.parametrize("V,C,D,h,w", [(2, 8, 6, 12, 40), (3, 16, 4, 9, 70), (1, 32, 3, 8, 20), (2, 8, 40, 24, 130), (4, 16, 20, 33, 65)])
def test_warp_aggregate_backward_vs_autograd(V, C, D, h, w, dev, ops):
    """Training step: gradients of the un-normalised warp-aggregate w.r.t. ref / src / vis against
    torch autograd through the CPU oracle's warp (F.grid_sample)."""
    from oracle import cds_oracle as O
    feats, cams, hyp, _, _, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=60 + V)
    g = torch.Generator().manual_seed(5)
    ref = torch.stack([f["ref"][0][0] for f in feats]).requires_grad_(True)
    src = torch.stack([f["src"][0][0] for f in feats]).requires_grad_(True)
    vis = (torch.rand(V, h, w, generator=g) * 0.8 + 0.1).requires_grad_(True)
    G = torch.randn(C, D, h, w, generator=g)
    P_ref = O.compose_projection(cams[:, 0])
    vol = 0.0
    for v in range(V):
        warped = O.warp_volume(src[v:v + 1], O.compose_projection(cams[:, v + 1]), P_ref, hyp, exact=False)[0]
        vol = vol + ref[v].unsqueeze(1) * warped * vis[v].view(1, 1, h, w)
    (vol * G).sum().backward()
    ref_d = ref.detach().to(dev).requires_grad_(True)
    src_d = src.detach().to(dev).requires_grad_(True)
    vis_d = vis.detach().to(dev).requires_grad_(True)
    vol_d = ops.WarpAggregate.apply(ref_d, src_d.permute(0, 2, 3, 1).contiguous(), vis_d, mats, hyp_d)
    assert (vol_d.detach().cpu() - vol.detach()).abs().max() < 1e-5 * max(1.0, vol.detach().abs().max().item())   # un-normalised sum over V views
    (vol_d * G.to(dev)).sum().backward()
    for name, a, b in (("ref", ref_d.grad, ref.grad), ("src", src_d.grad, src.grad), ("vis", vis_d.grad, vis.grad)):
        err = (a.cpu() - b).abs().max().item()
        assert err < 2e-4 * max(1.0, b.abs().max().item()), (name, err)


def test_training_step_matches_reference(dev):
    """model.train() forward + final_loss + backward against the reference's CPU training step
    (tests/golden/g7_training_step.npz): loss, depth, and every parameter's gradient norm."""
    from cds_mvsnet_amd import CDSMVSNet, final_loss, seeded_init_
    g = load_golden("g7_training_step")
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).train()
    cams = {k[4:]: v.to(dev) for k, v in g.items() if k.startswith("cam_")}
    gt = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("gt_")}
    mask = {k[5:]: v.to(dev) for k, v in g.items() if k.startswith("mask_")}
    dv = g["depth_values"].to(dev)
    out = model(g["imgs"].to(dev), cams, dv, gt_depths=gt, temperature=0.1)
    assert set(out["stage1"].keys()) == {"depth", "photometric_confidence", "feat_distance", "norm_curv", "feat_target"}
    assert out["stage2"]["feat_distance"].shape == (2, 33, 32, 48) and out["stage2"]["feat_target"].shape == (2, 33, 32, 48)
    loss, depth_loss = final_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0], depth_interval=dv[:, 1] - dv[:, 0])
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))
    assert abs(depth_loss.item() - float(g["depth_loss"])) < 2e-3 * abs(float(g["depth_loss"]))
    assert (out["stage3"]["depth"].detach().cpu() - g["stage3_depth"]).abs().mean() < 5e-3
    names = [str(n) for n in g["param_names"]]
    want = dict(zip(names, g["grad_norms"]))
    got = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    assert set(got) == set(want)
    want = {n: float(v) for n, v in want.items()}
    rel = {n: abs(got[n] - want[n]) / max(want[n], 1e-6) for n in want}
    # observed on MI355X: loss equal to 1e-6, median gradient-norm error 2.3e-4, worst 2.9 % on a parameter whose
    # gradient norm is 0.005 (visibility-CNN bias): bound relatively, with an absolute floor for such tiny gradients
    for n in want:
        assert rel[n] < 0.08 or abs(got[n] - want[n]) < 5e-4, (n, got[n], want[n])
    assert sorted(rel.values())[len(rel) // 2] < 2e-3          # median parameter: 0.2 %
    assert (model.cost_regularization[2].prob.weight.grad.cpu() - g["grad_prob3"]).abs().max() < 2e-3 * g["grad_prob3"].abs().max()
    # full gradient TENSORS of one parameter per kind of layer (captured from the reference's step): direction, not only norm.
    # fp32 against fp32 through a cascade whose hypothesis ranges switch discretely: the cosine is the robust measure; the
    # element-wise bound is relative to the tensor's largest entry
    params = dict(model.named_parameters())
    full = [k for k in g if k.startswith("fullgrad:")]
    assert len(full) >= 8
    for key in full:
        want_g = g[key].double()
        got_g = params[key[len("fullgrad:"):]].grad.detach().cpu().double()
        assert got_g.shape == want_g.shape, key
        cos = float((got_g * want_g).sum() / (got_g.norm() * want_g.norm() + 1e-30))
        err = float((got_g - want_g).abs().max() / (want_g.abs().max() + 1e-30))
        assert cos > 0.999 and err < 0.05, (key, cos, err)
    # BatchNorm running statistics were updated by the step (training-mode BN, momentum 0.1)
    assert model.cost_regularization[0].conv0.bn.num_batches_tracked.item() == 101


@pytest.mark.parametrize("h,w", [(32, 48), (19, 27)])
def test_refinement_hip_vs_oracle(h, w, seeded_state):
    """a15 on the HIP kernels (conv2d + deconv2d + finish) against the oracle restatement, incl. odd sizes."""
    from oracle import cds_oracle as O
    model = seeded_state(True)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    net = model.refine_network.eval().cuda()
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 2 * h, 2 * w, generator=g)
    dmin, dmax = torch.tensor([170.0, 160.0]), torch.tensor([361.0, 350.0])
    depth0 = dmin.view(2, 1, 1, 1) + (dmax - dmin).view(2, 1, 1, 1) * torch.rand(2, 1, h, w, generator=g)
    exp = O.refinement(img, depth0, dmin, dmax, sd)
    with torch.no_grad():
        got = net(img.cuda(), depth0.cuda(), dmin.cuda(), dmax.cuda()).cpu()
    assert got.shape == exp.shape
    assert (got - exp).abs().max() < 2e-3 and (got - exp).abs().mean() < 1e-4      # depths ~160-360


def test_conv3d_channels_last_mfma_vs_torch(dev, ops):
    """cds_conv3d_k3_cl_f32 (channels-last LDS tile, one b128 LDS read per four MFMAs) vs torch fp32: one and several
    16-channel chunks, several cout blocks, partial tiles in x / y / z, with and without bias / skip / ReLU."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    for (cin, cout, D, H, W, relu, use_skip, use_bias) in [(16, 16, 2, 4, 32, True, True, True), (16, 16, 3, 7, 36, True, False, True),
                                                           (32, 32, 5, 6, 40, True, True, True), (64, 64, 3, 5, 20, False, False, False),
                                                           (16, 48, 1, 1, 16, True, True, False), (32, 16, 7, 9, 100, True, True, True)]:
        assert ops.conv3d_cl_supported(cin, cout, W, 1)
        x = torch.randn(cin, D, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g) if use_bias else None
        ref = F.conv3d(x[None], w, b, padding=1)[0]
        if relu:
            ref = F.relu(ref)
        skip = torch.randn(ref.shape, generator=g) if use_skip else None
        wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().to(dev)
        wcl = w.permute(2, 3, 4, 0, 1).reshape(27, cout, cin).contiguous().to(dev)
        os.environ["CDS_CONV_CL"] = "2"          # force the channels-last kernel for every covered shape
        try:
            out = ops.conv3d_k3(x.to(dev), wpk, b.to(dev) if use_bias else None, relu=relu,
                                skip=skip.to(dev) if use_skip else None, wcl=wcl).cpu()
        finally:
            os.environ.pop("CDS_CONV_CL")
        want = ref + skip if use_skip else ref
        assert (out - want).abs().max() < 2e-5 * max(1.0, ref.abs().max().item()), (cin, cout, D, H, W)


@pytest.mark.parametrize("H,W,N,refine", [(1184, 1600, 5, False), (1056, 1920, 7, False), (1152, 1536, 5, True),
                                          (512, 640, 3, False), (512, 640, 5, False)])
def test_full_size_cascade_configs(H, W, N, refine):
    """BASELINE configs 3 / 4 / 3alt, config 1's shape (640x512, N=3) and M2 (640x512, N=5) at full size through size-independent
    properties (the CPU oracle needs minutes there): finite outputs, depths inside the hypothesis range, confidences in [0,1],
    output shapes, run-to-run agreement (two forwards with stage 1 on its side stream next to FeatureNet: this is the test that caught
    a kernel corrupting its stream neighbours in round 5) and view-order covariance of the stage-1 depth."""
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
    dev = torch.device("cuda")
    model = seeded_init_(CDSMVSNet(refine=refine, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
    imgs = synth.make_images(N, H, W, seed=4).to(dev)
    cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=refine, seed=4).items()}
    dv = synth.make_depth_values().to(dev)
    with torch.no_grad():
        a = model(imgs, cams, dv, temperature=0.01)
        b = model(imgs, cams, dv, temperature=0.01)
        perm = [0] + list(range(N - 1, 0, -1))          # same views, reversed source order
        c = model(imgs[:, perm], {k: v[:, perm] for k, v in cams.items()}, dv, temperature=0.01)
    Hs, Ws = (H // 2, W // 2) if refine else (H, W)
    assert a["depth"].shape == (1, Hs, Ws) and a["refined_depth"].shape == (1, H, W)
    assert a["stage1"]["depth"].shape == (1, Hs // 4, Ws // 4) and a["stage2"]["depth"].shape == (1, Hs // 2, Ws // 2)
    lo, hi = float(dv[0, 0]), float(dv[0, -1])
    for k in ("stage1", "stage2", "stage3"):
        d, cf = a[k]["depth"], a[k]["photometric_confidence"]
        assert torch.isfinite(d).all() and torch.isfinite(cf).all() and torch.isfinite(a[k]["norm_curv"]).all()
        assert d.min() >= lo - 1e-3 and d.max() <= hi + 1e-3
        assert cf.min() >= 0 and cf.max() <= 1 + 1e-5
    assert torch.isfinite(a["refined_depth"]).all()
    assert (a["depth"] - b["depth"]).abs().mean() < 1e-3
    # the aggregation is a sum over source views: their order only changes fp32 summation order
    assert (a["stage1"]["depth"] - c["stage1"]["depth"]).abs().mean() < 5e-3


@pytest.mark.parametrize("exact", POSITION_MODES)
def test_warp_lds_fallbacks_wild_geometry(exact, dev, ops):
    """Geometry that defeats the LDS fast path: a wide baseline and near depths give tens of pixels of parallax per plane
    (boxes over the LDS budget even after the chunk is halved to 8 planes -> global-memory path), randomly permuted
    (non-monotone) hypotheses, and a source camera whose principal plane cuts the depth range (projective pole: samples
    jump across the image).  K1 / K3 must still agree with the CPU oracle."""
    from oracle import cds_oracle as O
    from cds_mvsnet_amd import synth, geometry
    V, C, D, h, w = 3, 8, 40, 24, 136
    feats = synth.make_pair_features(V, C, h, w, seed=91, sharp=True)
    cams = synth.make_cameras(V + 1, h, w, refine=False, seed=91, baseline=(900.0, 300.0, 40.0))["stage3"]
    cams[0, 3, 0, :3, 3] += torch.tensor([0.0, 0.0, -260.0])            # view 3: its z = 0 plane lies inside the sweep
    g = torch.Generator().manual_seed(91)
    hyp = 120.0 + 600.0 * torch.rand(1, D, h, w, generator=g)             # 120..720, unordered per pixel
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
    mats = geometry.warp_matrices(cams[0])
    hyp_d = hyp[0].to(dev).contiguous()
    P_ref = O.compose_projection(cams[:, 0])
    ent = ops.warp_entropy(ref, src, mats, hyp_d, exact=exact).cpu()
    vis = torch.rand(V, h, w, generator=g)
    vol, _ = ops.warp_aggregate(ref, src, vis.to(dev), mats, hyp_d, normalize=False, exact=exact)
    want = torch.zeros(C, D, h, w)
    for v in range(V):
        warped = O.warp_volume(feats[v]["src"][0], O.compose_projection(cams[:, v + 1]), P_ref, hyp)
        in_prod, e = O.correlation_entropy(feats[v]["ref"][0], warped)
        ok = torch.isfinite(e[0, 0])
        assert ((ent[v] - e[0, 0]).abs()[ok]).max() < 5e-5
        want += (in_prod * vis[v].view(1, 1, 1, h, w))[0]
    fin = torch.isfinite(want)
    assert fin.float().mean() > 0.99
    assert ((vol.cpu() - want).abs()[fin]).max() < 2e-5


@pytest.mark.parametrize("k,stride,cin,cout,H,W", [(3, 1, 8, 11, 37, 70), (1, 1, 24, 8, 20, 33), (5, 1, 16, 19, 18, 40), (3, 2, 16, 32, 21, 50),
                                                     (3, 1, 8, 11, 37, 72), (7, 1, 8, 11, 20, 36), (3, 2, 16, 32, 21, 52)])
def test_conv2d_normalise_on_load_equals_materialised(k, stride, cin, cout, H, W, dev, ops):
    """cds_instnorm_affine_f32 + cds_conv2d_affine_f32 (InstanceNorm + LeakyReLU applied while the input tile is loaded)
    must give what the materialised path gives (cds_instnorm_act_f32 then cds_conv2d_f32), including the zero
    padding, mixed with an untouched (1, 0, 1) channel group, at odd sizes."""
    from cds_mvsnet_amd.ops import ACT_LEAKY01, ACT_NONE
    g = torch.Generator().manual_seed(k * 10 + stride)
    N = 3
    x = (torch.randn(N, cin, H, W, generator=g) * 2 + 0.5).to(dev)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    coutp = (cout + 7) // 8 * 8
    wpk = torch.nn.functional.pad(w.permute(1, 2, 3, 0).reshape(cin, k * k, cout), (0, coutp - cout)).contiguous().to(dev)
    pad = (k - 1) // 2
    want = ops.conv2d(ops.instnorm_act(x, ACT_LEAKY01), wpk, None, cout, k, stride, pad, ACT_NONE)
    aff = ops.instnorm_affine(x, 0.1)
    got = ops.conv2d(x, wpk, None, cout, k, stride, pad, ACT_NONE, in_affine=aff)
    # same arithmetic; the two statistics passes add their block partials with atomics, so allow the last bit of 1/std
    assert (got - want).abs().max() < 1e-5
    # half of the channels already materialised: identity rows leave them as they are
    xm = x.clone()
    xm[:, : cin // 2] = ops.instnorm_act(x[:, : cin // 2].contiguous(), ACT_LEAKY01)
    aff2 = aff.clone()
    aff2[:, : cin // 2] = torch.tensor([1.0, 0.0, 1.0], device=dev)
    assert (ops.conv2d(xm, wpk, None, cout, k, stride, pad, ACT_NONE, in_affine=aff2) - want).abs().max() < 1e-5


@pytest.mark.parametrize("seed", [3, 17, 91])
def test_warp_kernels_random_shapes_against_oracle(dev, seed):
    """Seeded fuzz of K1 / K3 (C = 8 LDS kernels and their direct fallbacks): random view counts 1..4,
    plane counts around the 32-plane chunk boundary, ragged widths / heights, short and long baselines and
    per-pixel hypothesis jitter, each compared with the oracle's explicit gather (tests/tools/fuzz_warp.py)."""
    import importlib.util
    import random
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz_warp.py")
    spec = importlib.util.spec_from_file_location("fuzz_warp", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = random.Random(seed)
    for _ in range(6):
        vol_err, ent_err = fuzz.one_case(rng, dev, verbose=False)
        assert vol_err < 1e-5, vol_err
        assert ent_err < 2e-5, ent_err


@pytest.mark.parametrize("ca,cb,cout,h,w", [(32, 16, 16, 12, 38), (16, 8, 8, 34, 70), (16, 8, 12, 6, 2)])
def test_fpn_lateral_equals_conv_of_concatenation(dev, ops, ca, cb, cout, h, w):
    """cds_conv2d_fpn_f32 == the 1x1 convolution of cat(nearest2x(coarse), skip) (module.py:253-254,260-261), with and
    without pending normalisation tables, bit for bit (same channel and fma order)."""
    g = torch.Generator().manual_seed(ca * 100 + w)
    N = 3
    coarse = torch.randn(N, ca, h, w, generator=g).to(dev)
    skip = torch.randn(N, cb, 2 * h, 2 * w, generator=g).to(dev)
    weight = torch.randn(cout, ca + cb, 1, 1, generator=g) * 0.2
    coutp = (cout + 7) // 8 * 8
    wpk = torch.zeros(ca + cb, 1, coutp)
    wpk[:, 0, :cout] = weight[:, :, 0, 0].t()
    wpk = wpk.to(dev)
    up = torch.nn.functional.interpolate(coarse, scale_factor=2, mode="nearest")
    cat = torch.cat((up, skip), dim=1).contiguous()
    a_c, a_s = ops.instnorm_affine(coarse, 0.1), ops.instnorm_affine(skip, 0.1)
    ident = torch.tensor([1.0, 0.0, 1.0], device=dev).expand(N, ca, 3)
    for ac, asx, acat in ((None, None, None), (a_c, a_s, torch.cat((a_c, a_s), 1).contiguous()),
                          (None, a_s, torch.cat((ident, a_s), 1).contiguous())):
        want = ops.conv2d(cat, wpk, None, cout, 1, 1, 0, in_affine=acat)
        got = ops.conv2d_fpn(coarse, skip, wpk, cout, ac, asx)
        assert torch.equal(got, want), (got - want).abs().max()
    # statistics taken inside the kernel: same output, the normalisation table of the two-pass path
    got2, aff2 = ops.conv2d_fpn(coarse, skip, wpk, cout, a_c, a_s, stats_slope=0.1)
    want2 = ops.conv2d_fpn(coarse, skip, wpk, cout, a_c, a_s)
    assert torch.equal(got2, want2)
    assert (aff2 - ops.instnorm_affine(want2, 0.1)).abs().max() < 1e-6
    # and against plain PyTorch
    ref = torch.nn.functional.conv2d(cat.cpu(), weight)
    assert (ops.conv2d_fpn(coarse, skip, wpk, cout).cpu() - ref).abs().max() < 1e-4
    with pytest.raises(ValueError):
        ops.conv2d_fpn(coarse, skip[:, :, :-1], wpk, cout)


@pytest.mark.parametrize("N,H,W", [(1, 8, 64), (3, 37, 100), (2, 5, 4), (4, 128, 160)])
def test_conv2d_c16_mfma_against_torch(dev, ops, N, H, W):
    """Matrix-core 16 -> 16 3x3 convolution (+ fused 1x1 head) against PyTorch fp32 and the VALU kernel; tolerance 2e-5
    abs on O(1) outputs (fp32 MFMA accumulation, different summation order)."""
    g = torch.Generator().manual_seed(N * 1000 + W)
    x = torch.randn(N, 16, H, W, generator=g)
    weight = torch.randn(16, 16, 3, 3, generator=g) * 0.1
    bias = torch.randn(16, generator=g) * 0.1
    hw_, hb_ = torch.randn(16, generator=g) * 0.3, torch.randn(1, generator=g)
    wcl = weight.permute(2, 3, 0, 1).reshape(9, 16, 16).contiguous().to(dev)
    want = torch.relu(torch.nn.functional.conv2d(x, weight, bias, padding=1))
    got = ops.conv2d_k3_c16(x.to(dev), wcl, bias.to(dev), 1)
    assert (got.cpu() - want).abs().max() < 2e-5
    wpk = weight.permute(1, 2, 3, 0).reshape(16, 9, 16).contiguous().to(dev)
    valu = ops.conv2d(x.to(dev), wpk, bias.to(dev), 16, 3, 1, 1, 1)
    assert (got - valu).abs().max() < 2e-5
    head = torch.sigmoid(torch.nn.functional.conv2d(want, hw_.view(1, 16, 1, 1), hb_))[:, 0]
    got_h = ops.conv2d_k3_c16(x.to(dev), wcl, bias.to(dev), 1, head_w=hw_.to(dev), head_b=hb_.to(dev))
    assert got_h.shape == (N, H, W)
    assert (got_h.cpu() - head).abs().max() < 1e-5
    with pytest.raises(ValueError):
        ops.conv2d_k3_c16(x[:, :, :, :3].contiguous().to(dev), wcl, None)


@pytest.mark.parametrize("N", [2, 10])
def test_full_forward_view_counts(dev, seeded_state, N):
    """Whole model against the CPU oracle with a single source view (V = 1 kernels) and with 9 source views: more than
    one batched FeatureNet pass (CDS_MAX_IMAGES / 2 pairs each) and more than CDS_MAX_VIEWS views in K1 / K3 (chunked
    launches, two-pass aggregation)."""
    from cds_mvsnet_amd import synth
    from oracle import cds_oracle as O
    H, W = 64, 96
    model = seeded_state(False)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    imgs = synth.make_images(N, H, W, seed=13)
    cams = synth.make_cameras(N, H, W, refine=False, seed=13)
    dv = synth.make_depth_values()
    want = O.forward(imgs, cams, dv, sd, refine=False, temperature=0.01, exact=False)
    model = model.to(dev)
    with torch.no_grad():
        got = model(imgs.to(dev), cams, dv, temperature=0.01)
    for s in (1, 2, 3):
        a, b = got[f"stage{s}"], want[f"stage{s}"]
        assert (a["depth"].cpu() - b["depth"]).abs().mean() < 1e-3, s
        assert (a["photometric_confidence"].cpu() - b["photometric_confidence"]).abs().mean() < 1e-3, s
        assert (a["norm_curv"].cpu() - b["norm_curv"]).abs().max() < 1e-4, s


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,stride,D,H,W", [(8, 8, 1, 9, 13, 40), (8, 8, 1, 16, 32, 64), (16, 16, 1, 6, 10, 36),
                                                  (32, 32, 1, 5, 9, 20), (64, 64, 1, 4, 6, 16), (8, 4, 1, 7, 8, 44),
                                                  (8, 16, 2, 9, 17, 35), (16, 32, 2, 8, 16, 32), (32, 64, 2, 6, 9, 21),
                                                  (8, 16, 2, 16, 32, 64), (8, 8, 101, 9, 13, 40), (8, 8, 101, 50, 9, 70), (16, 8, 101, 5, 8, 33),
                                                  (32, 8, 101, 8, 16, 64),
                                                  # ragged volumes over several columns / z segments of the z-marching kernels (conv3d_zmg.hip)
                                                  (32, 8, 101, 20, 23, 100), (16, 8, 101, 17, 29, 70), (16, 16, 1, 21, 27, 70),
                                                  (8, 16, 2, 21, 37, 131), (16, 32, 2, 19, 31, 67), (8, 8, 101, 23, 19, 67)])
def test_conv3d_split_bf16_is_fp32_class(cin, cout, stride, D, H, W, dev, ops):
    """csrc/conv3d_sbf.hip: 3 x 3 x 3 convolution with every fp32 operand split exactly into three bf16 terms and six
    error-compensated partial products on the bf16 matrix cores (fp32 accumulate).  The claim is fp32-CLASS accuracy,
    measured against a float64 convolution: its error must not exceed 1.5x the error of a plain fp32 evaluation of the same
    convolution (+ one fp32 ulp of the result scale) — the larger of PyTorch's fp32 CPU convolution (blocked summation)
    and this library's exact-fp32 kernel (cds_conv3d_k3_f32: one sequential fmaf chain over the 27 Cin products per output) —
    with bias, ReLU and residual fused like the fp32 kernels."""
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(cin, D, H, W, generator=g) * torch.exp(torch.randn(cin, 1, 1, 1, generator=g))     # uneven channel scales
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    pair = stride == ops.SBF_PAIR        # stride 1, Cout = 8, voxel-pair columns
    code, stride = stride, (1 if pair else stride)
    want64 = F.conv3d(x.double().unsqueeze(0), w.double(), b.double(), padding=1, stride=stride)[0]
    want32 = F.conv3d(x.unsqueeze(0), w, b, padding=1, stride=stride)[0]
    skip = torch.randn(want32.shape, generator=g)
    ws = ops.split_pack_conv3d_pair(w.to(dev)) if pair else ops.split_pack_conv3d(w.to(dev))
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    got = ops.conv3d_sbf(x_cl, ws, b.to(dev), cout, stride=code, relu=False).cpu().permute(3, 0, 1, 2)
    wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().to(dev)
    try:
        chain32 = ops.conv3d_k3(x.to(dev), wpk, b.to(dev), stride=stride, relu=False).cpu()
    except ValueError:          # Cout = 4 is outside the fp32 kernels' coverage
        chain32 = want32
    err_sbf = (got.double() - want64).abs().max().item()
    err_f32 = max((want32.double() - want64).abs().max().item(), (chain32.double() - want64).abs().max().item())
    ulp = want64.abs().max().item() * 2.0 ** -23
    print(f"conv3d {cin}->{cout} s{stride}: max err vs float64: split-bf16 {err_sbf:.2e}, torch fp32 "
          f"{(want32.double() - want64).abs().max().item():.2e}, fp32 fmaf-chain kernel {(chain32.double() - want64).abs().max().item():.2e}")
    assert err_sbf <= 1.5 * err_f32 + ulp, (err_sbf, err_f32)
    got2 = ops.conv3d_sbf(x_cl, ws, b.to(dev), cout, stride=code, relu=True,
                          skip=skip.permute(1, 2, 3, 0).contiguous().to(dev)).cpu().permute(3, 0, 1, 2)
    want2 = skip.double() + want64.clamp_min(0)
    assert (got2.double() - want2).abs().max().item() <= 1.5 * err_f32 + 2 * ulp


@pytest.mark.gpu
@pytest.mark.parametrize("loose", [1.0, 6.5])
@pytest.mark.parametrize("cin,cout,stride,D,H,W", [(8, 8, 101, 9, 13, 40), (8, 8, 101, 50, 9, 70), (16, 8, 101, 5, 8, 33), (32, 8, 101, 8, 16, 64),
                                                  (32, 8, 101, 20, 23, 100), (16, 8, 101, 17, 29, 70), (16, 16, 1, 6, 10, 36),
                                                  (16, 16, 1, 21, 27, 70), (8, 16, 2, 9, 17, 35), (8, 16, 2, 21, 37, 131),
                                                  (16, 32, 2, 8, 16, 32), (16, 32, 2, 19, 31, 67),
                                                  # the tiled kernels (deep layers): conv4 / conv5 / conv6
                                                  (32, 32, 1, 5, 9, 20), (32, 32, 1, 9, 13, 70), (64, 64, 1, 4, 6, 16), (64, 64, 1, 6, 11, 37),
                                                  (32, 64, 2, 6, 9, 21), (32, 64, 2, 11, 18, 45)])
def test_conv3d_split_f16_is_fp32_class(cin, cout, stride, D, H, W, loose, dev, ops):
    """Round 6: the z-marching kernels in SPLIT-F16 arithmetic (csrc/sbf_common.hpp: two fp16 terms of operand x power-of-two tensor
    scale, three products per K-step instead of split-bf16's six, exact rescaling) - the same claim and the same bar as
    test_conv3d_split_bf16_is_fp32_class: against a float64 convolution no worse than 1.5x a plain fp32 evaluation (+ one ulp of
    the result scale), on the same inputs (uneven channel scales: magnitudes spread over ~3 decades inside one tensor scale).
    `loose`: the input bound handed to the kernel is that many times the true maximum (any upper bound is valid: it only moves the
    tensor scale - here across a power of two).  The kernel's out_bound must be the exact maximum magnitude of what it stored."""
    if not ops.USE_SPLIT_F16:
        pytest.skip("CDS_SPLIT_F16=0: the split-f16 convolution entry is switched off")
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(cin, D, H, W, generator=g) * torch.exp(torch.randn(cin, 1, 1, 1, generator=g))     # uneven channel scales
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    pair = stride == ops.SBF_PAIR
    code, stride = stride, (1 if pair else stride)
    assert ops.conv3d_sf16_supported(cin, cout, code)
    want64 = F.conv3d(x.double().unsqueeze(0), w.double(), b.double(), padding=1, stride=stride)[0]
    want32 = F.conv3d(x.unsqueeze(0), w, b, padding=1, stride=stride)[0]
    wh, w_inv = (ops.split_pack_conv3d_pair if pair else ops.split_pack_conv3d)(w.to(dev), f16=True)
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    in_bound = (x_cl.abs().amax() * loose).reshape(1)
    for relu in (False, True):
        out_bound = torch.zeros(1, device=dev)
        got = ops.conv3d_sbf(x_cl, wh, b.to(dev), cout, stride=code, relu=relu, in_bound=in_bound, w_inv_scale=w_inv, out_bound=out_bound)
        assert float(out_bound) == float(got.abs().max()), (relu, float(out_bound), float(got.abs().max()))
        got = got.cpu().permute(3, 0, 1, 2)
        ref64 = want64.clamp_min(0) if relu else want64
        wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().to(dev)
        chain32 = ops.conv3d_k3(x.to(dev), wpk, b.to(dev), stride=stride, relu=False).cpu()
        err = (got.double() - ref64).abs().max().item()
        err_f32 = max((want32.double() - want64).abs().max().item(), (chain32.double() - want64).abs().max().item())
        ulp = want64.abs().max().item() * 2.0 ** -23
        if not relu:
            ws = ops.split_pack_conv3d_pair(w.to(dev)) if pair else ops.split_pack_conv3d(w.to(dev))
            sbf = ops.conv3d_sbf(x_cl, ws, b.to(dev), cout, stride=code, relu=False).cpu().permute(3, 0, 1, 2)
            print(f"conv3d {cin}->{cout} s{stride} bound x{loose}: max err vs float64: split-f16 {err:.2e}, split-bf16 "
                  f"{(sbf.double() - want64).abs().max().item():.2e}, fp32 {err_f32:.2e}")
        assert err <= 1.5 * err_f32 + ulp, (relu, err, err_f32)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,D,H,W", [(64, 32, 2, 5, 18), (64, 32, 5, 9, 37), (32, 16, 3, 6, 20), (32, 16, 7, 13, 37)])
def test_deconv3d_split_f16_is_fp32_class(cin, cout, D, H, W, dev, ops):
    """conv7 (64 -> 32, tiled transposed kernel) and conv9 (32 -> 16, z-marching class-per-wave kernel) in split-f16 arithmetic against
    float64: the bar of test_deconv3d_split_bf16_is_fp32_class (no worse than 1.5x a plain fp32 evaluation), with ReLU + residual, and the
    kernel's out_bound = the largest magnitude it stored."""
    g = torch.Generator().manual_seed(cin * 10 + cout)
    x = torch.randn(cin, D, H, W, generator=g) * torch.exp(torch.randn(cin, 1, 1, 1, generator=g))
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    b = torch.randn(cout, generator=g)
    want64 = F.conv_transpose3d(x.double().unsqueeze(0), w.double(), b.double(), stride=2, padding=1, output_padding=1)[0]
    want32 = F.conv_transpose3d(x.unsqueeze(0), w, b, stride=2, padding=1, output_padding=1)[0]
    skip = torch.randn(want32.shape, generator=g)
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    skip_cl = skip.permute(1, 2, 3, 0).contiguous().to(dev)
    in_bound = x_cl.abs().amax().reshape(1)
    wpk = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous().to(dev)
    chain32 = ops.deconv3d_k3s2(x.to(dev), wpk, b.to(dev), relu=False).cpu()
    err_f32 = max((want32.double() - want64).abs().max().item(), (chain32.double() - want64).abs().max().item())
    ulp = want64.abs().max().item() * 2.0 ** -23
    if cout == 32:
        wh, winv = ops.split_pack_deconv3d(w.to(dev), f16=True)
        run = lambda **kw: ops.deconv3d_sbf(x_cl, wh, b.to(dev), cout, in_bound=in_bound, w_inv_scale=winv, **kw)
    else:
        wh, winv = ops.split_pack_deconv_cls(w.to(dev), f16=True)
        run = lambda **kw: ops.deconv3d_zm(x_cl, wh, b.to(dev), in_bound=in_bound, w_inv_scale=winv, **kw)
    ob = torch.zeros(1, device=dev)
    got = run(relu=False, out_bound=ob)
    assert float(ob) == float(got.abs().max())
    err = (got.cpu().permute(3, 0, 1, 2).double() - want64).abs().max().item()
    print(f"deconv3d {cin}->{cout}: max err vs float64: split-f16 {err:.2e}, fp32 {err_f32:.2e}")
    assert err <= 1.5 * err_f32 + ulp, (err, err_f32)
    ob.zero_()
    got2 = run(relu=True, skip=skip_cl, out_bound=ob)
    assert float(ob) == float(got2.abs().max())
    want2 = skip.double() + want64.clamp_min(0)
    assert (got2.cpu().permute(3, 0, 1, 2).double() - want2).abs().max().item() <= 1.5 * err_f32 + 2 * ulp


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,D,H,W", [(16, 8, 5, 7, 37), (16, 8, 4, 8, 32), (32, 16, 3, 6, 20), (64, 32, 2, 5, 18),
                                            (8, 8, 3, 4, 16)])
def test_deconv3d_split_bf16_is_fp32_class(cin, cout, D, H, W, dev, ops):
    """ConvTranspose3d k3 s2 p1 op1 on the split-bf16 kernel (all 8 output parity classes per workgroup; Cout = 8: the x
    parities share an MFMA) against float64, same acceptance as the forward convolution."""
    g = torch.Generator().manual_seed(cin * 10 + cout)
    x = torch.randn(cin, D, H, W, generator=g) * torch.exp(torch.randn(cin, 1, 1, 1, generator=g))
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    b = torch.randn(cout, generator=g)
    want64 = F.conv_transpose3d(x.double().unsqueeze(0), w.double(), b.double(), stride=2, padding=1, output_padding=1)[0]
    want32 = F.conv_transpose3d(x.unsqueeze(0), w, b, stride=2, padding=1, output_padding=1)[0]
    skip = torch.randn(want32.shape, generator=g)
    ws = ops.split_pack_deconv3d(w.to(dev))
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    got = ops.deconv3d_sbf(x_cl, ws, b.to(dev), cout, relu=False).cpu().permute(3, 0, 1, 2)
    wpk = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous().to(dev)
    chain32 = ops.deconv3d_k3s2(x.to(dev), wpk, b.to(dev), relu=False).cpu()
    err_sbf = (got.double() - want64).abs().max().item()
    err_f32 = max((want32.double() - want64).abs().max().item(), (chain32.double() - want64).abs().max().item())
    ulp = want64.abs().max().item() * 2.0 ** -23
    print(f"deconv3d {cin}->{cout}: max err vs float64: split-bf16 {err_sbf:.2e}, torch fp32 "
          f"{(want32.double() - want64).abs().max().item():.2e}, fp32 fmaf-chain kernel {(chain32.double() - want64).abs().max().item():.2e}")
    assert err_sbf <= 1.5 * err_f32 + ulp, (err_sbf, err_f32)
    got2 = ops.deconv3d_sbf(x_cl, ws, b.to(dev), cout, relu=True,
                            skip=skip.permute(1, 2, 3, 0).contiguous().to(dev)).cpu().permute(3, 0, 1, 2)
    want2 = skip.double() + want64.clamp_min(0)
    assert (got2.double() - want2).abs().max().item() <= 1.5 * err_f32 + 2 * ulp
    got3 = ops.deconv3d_sbf(x_cl, ws, b.to(dev), cout, relu=True, skip=skip.permute(1, 2, 3, 0).contiguous().to(dev),
                            out_planar=True).cpu()                      # same values, planar output (conv11 -> prob)
    assert torch.equal(got3, got2.contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,W,nseg", [(1, 1, 1, 0), (2, 3, 5, 0), (3, 8, 16, 1), (4, 9, 17, 2), (5, 13, 33, 0), (7, 24, 47, 3), (6, 5, 70, 1)])
def test_conv9_class_per_wave_z_march(D, H, W, nseg, dev, ops, monkeypatch):
    """csrc/deconv3d_zm.hip: ConvTranspose3d 32 -> 16 (k3 s2 p1 op1) + shift + ReLU + residual as a z-marching kernel with one wave
    per output parity class (models/module.py:125-160) against float64: same acceptance as the tiled split-bf16 kernel (no worse
    than 1.5x PyTorch's fp32 layer), on volumes that exercise partial 16 x 8-cell columns and the z segmentation; and without the
    residual / activation."""
    if nseg:
        monkeypatch.setenv("CDS_DZM_NSEG", str(nseg))
    else:
        monkeypatch.delenv("CDS_DZM_NSEG", raising=False)
    g = torch.Generator().manual_seed(D * 100 + W)
    x = torch.randn(32, D, H, W, generator=g) * torch.exp(0.5 * torch.randn(32, 1, 1, 1, generator=g))
    w = torch.randn(32, 16, 3, 3, 3, generator=g) / (27 * 4) ** 0.5
    b = torch.randn(16, generator=g)
    skip = torch.randn(16, 2 * D, 2 * H, 2 * W, generator=g)
    want64 = F.conv_transpose3d(x.double()[None], w.double(), b.double(), stride=2, padding=1, output_padding=1)[0]
    want32 = F.conv_transpose3d(x[None], w, b, stride=2, padding=1, output_padding=1)[0]
    wc = ops.split_pack_deconv_cls(w.to(dev))
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    got = ops.deconv3d_zm(x_cl, wc, b.to(dev), relu=False).cpu().permute(3, 0, 1, 2)
    err = (got.double() - want64).abs().max().item()
    wpk = w.permute(0, 2, 3, 4, 1).reshape(32, 27, 16).contiguous().to(dev)
    chain32 = ops.deconv3d_k3s2(x.to(dev), wpk, b.to(dev), relu=False).cpu()          # the exact-fp32 fmaf-chain kernel
    err32 = max((want32.double() - want64).abs().max().item(), (chain32.double() - want64).abs().max().item())
    ulp = want64.abs().max().item() * 2.0 ** -23
    print(f"conv9 z-march D{D} H{H} W{W}: max err vs float64 {err:.2e} (fp32 implementations {err32:.2e})")
    assert err <= 1.5 * err32 + 2 * ulp, (err, err32)       # 2 ulp: a handful of voxels make the fp32 reference's max a lucky draw
    got2 = ops.deconv3d_zm(x_cl, wc, b.to(dev), relu=True, skip=skip.permute(1, 2, 3, 0).contiguous().to(dev)).cpu().permute(3, 0, 1, 2)
    assert (got2.double() - (skip.double() + want64.clamp_min(0))).abs().max().item() <= 1.5 * err32 + 3 * ulp
    tiled = ops.deconv3d_sbf(x_cl, ops.split_pack_deconv3d(w.to(dev)), b.to(dev), 16, relu=True,
                             skip=skip.permute(1, 2, 3, 0).contiguous().to(dev)).cpu().permute(3, 0, 1, 2)
    assert (got2 - tiled).abs().max().item() <= 8 * ulp + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,W,nseg", [(1, 1, 1, 0), (2, 3, 5, 0), (3, 7, 31, 1), (4, 6, 30, 0), (5, 13, 61, 2), (9, 12, 64, 4),
                                        (7, 20, 33, 1), (6, 40, 95, 0)])
def test_conv11_residual_prob_fused(D, H, W, nseg, dev, ops, monkeypatch):
    """csrc/deconv_prob_zm.hip: conv11 (ConvTranspose3d 16 -> 8 + folded BN + ReLU), the conv0 residual and prob (Conv3d
    8 -> 1) in one z-marching launch (models/module.py:125-160, :495-499) against a float64 restatement: fp32-class (no worse
    than 1.5x the two separate fp32 layers of PyTorch), on ragged volumes that exercise the 30 x 6-cell column tiling (partial
    tiles, one-cell volumes, exact multiples) and the z segmentation (segment seams; nseg = 0: the launcher's own choice,
    which cuts small volumes into one-plane segments)."""
    if nseg:
        monkeypatch.setenv("CDS_DPZ_NSEG", str(nseg))
    else:
        monkeypatch.delenv("CDS_DPZ_NSEG", raising=False)
    g = torch.Generator().manual_seed(D * 1000 + H * 10 + W)
    x = torch.randn(16, D, H, W, generator=g) * torch.exp(0.5 * torch.randn(16, 1, 1, 1, generator=g))
    skip = torch.randn(8, 2 * D, 2 * H, 2 * W, generator=g)
    w11 = torch.randn(16, 8, 3, 3, 3, generator=g) / (27 * 2) ** 0.5
    b11 = torch.randn(8, generator=g)
    wp = torch.randn(1, 8, 3, 3, 3, generator=g) / 27 ** 0.5
    y64 = F.conv_transpose3d(x.double()[None], w11.double(), b11.double(), stride=2, padding=1, output_padding=1).clamp_min(0) \
        + skip.double()[None]
    want64 = F.conv3d(y64, wp.double(), padding=1)[0, 0]
    y32 = F.conv_transpose3d(x[None], w11, b11, stride=2, padding=1, output_padding=1).clamp_min(0) + skip[None]
    want32 = F.conv3d(y32, wp, padding=1)[0, 0]
    got = ops.deconv_prob_zm(x.permute(1, 2, 3, 0).contiguous().to(dev), ops.split_pack_deconv_prob(w11.to(dev)), b11.to(dev),
                             skip.permute(1, 2, 3, 0).contiguous().to(dev), ops.pack_prob_table(wp.to(dev))).cpu()
    assert got.shape == want32.shape
    err = (got.double() - want64).abs().max().item()
    err32 = (want32.double() - want64).abs().max().item()
    ulp = want64.abs().max().item() * 2.0 ** -23
    print(f"conv11+prob fused D{D} H{H} W{W}: max err vs float64 {err:.2e} (torch fp32: {err32:.2e})")
    assert err <= 1.5 * err32 + 2 * ulp, (err, err32)
    # the same launch with the transposed convolution in split-f16 arithmetic (round 6): the same bar
    x_cl = x.permute(1, 2, 3, 0).contiguous().to(dev)
    wh, winv = ops.split_pack_deconv_prob(w11.to(dev), f16=True)
    got_h = ops.deconv_prob_zm(x_cl, wh, b11.to(dev), skip.permute(1, 2, 3, 0).contiguous().to(dev), ops.pack_prob_table(wp.to(dev)),
                               in_bound=x_cl.abs().amax().reshape(1), w_inv_scale=winv).cpu()
    err_h = (got_h.double() - want64).abs().max().item()
    assert err_h <= 1.5 * err32 + 2 * ulp, (err_h, err32)
    # and equal to the two separate product kernels to fp32 rounding (different summation order in prob)
    ysep = ops.deconv3d_sbf(x.permute(1, 2, 3, 0).contiguous().to(dev), ops.split_pack_deconv3d(w11.to(dev)), b11.to(dev), 8,
                            skip=skip.permute(1, 2, 3, 0).contiguous().to(dev), out_planar=True)
    sep = ops.conv3d_k3(ysep, wp.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous().to(dev), None, relu=False)[0].cpu()
    assert (got - sep).abs().max().item() <= 16 * ulp + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,ks,N,H,W,bias", [(8, 8, (3, 5, 7), 2, 21, 44, False), (16, 16, (3, 5), 1, 16, 36, False),
                                                   (32, 32, (1, 3), 2, 9, 20, True), (16, 16, (1, 3), 1, 12, 32, True),
                                                   (8, 8, (1, 3), 3, 8, 64, True)])
def test_dynconv_branches_on_matrix_cores(cin, cout, ks, N, H, W, bias, dev, ops):
    """csrc/conv2d_sbf.hip: all branch convolutions of a DynamicConv (convs[k] + att_convs[k] for every kernel size,
    dynamic_conv.py:112,116) in one launch on the bf16 matrix cores in split-bf16 arithmetic, with the producer's
    InstanceNorm + LeakyReLU applied on load.  fp32-class: against float64 no worse than 1.5x PyTorch's / the VALU kernel's
    fp32 error, and within 2e-5 of the exact-fp32 VALU kernel (cds_conv2d_affine_f32) it replaces."""
    g = torch.Generator().manual_seed(cin + 7 * len(ks))
    x = torch.randn(N, cin, H, W, generator=g)
    aff = torch.stack((0.5 + torch.rand(N, cin, generator=g), 0.3 * torch.randn(N, cin, generator=g),
                       torch.full((N, cin), 0.1)), dim=-1).contiguous()
    t = x * aff[:, :, 0, None, None] + aff[:, :, 1, None, None]
    xin = torch.where(t > 0, t, t * 0.1)
    co3 = cout + 3
    ws = [torch.randn(co3, cin, k, k, generator=g) / (cin * k * k) ** 0.5 for k in ks]
    bs = torch.randn(len(ks), co3, generator=g) if bias else None
    got = ops.dynconv_branches_sbf(x.to(dev), ops.split_pack_dynconv([w.to(dev) for w in ws]), bs.to(dev) if bias else None,
                                   co3, ks, in_affine=aff.to(dev)).cpu()
    for i, k in enumerate(ks):
        want64 = F.conv2d(xin.double(), ws[i].double(), bs[i].double() if bias else None, padding=(k - 1) // 2)
        want32 = F.conv2d(xin, ws[i], bs[i] if bias else None, padding=(k - 1) // 2)
        pad = (-co3) % 8
        wpk = F.pad(ws[i].permute(1, 2, 3, 0).reshape(cin, k * k, co3), (0, pad)).contiguous().to(dev)
        valu = ops.conv2d(x.to(dev), wpk, bs[i].to(dev) if bias else None, co3, k, 1, (k - 1) // 2, in_affine=aff.to(dev)).cpu()
        err = (got[i].double() - want64).abs().max().item()
        ref = max((want32.double() - want64).abs().max().item(), (valu.double() - want64).abs().max().item())
        ulp = want64.abs().max().item() * 2.0 ** -23
        assert err <= 1.5 * ref + ulp, (k, err, ref)
        assert (got[i] - valu).abs().max().item() < 2e-5 * max(1.0, want64.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,ks,N,H,W,bias", [(8, 8, (3, 5, 7), 2, 21, 44, True), (16, 16, (3, 5), 3, 16, 36, True),
                                                   (32, 32, (1, 3), 2, 9, 20, False), (8, 8, (1, 3), 1, 8, 64, True),
                                                   (16, 16, (3, 5), 1, 40, 100, False)])
def test_dynconv_fused_equals_branches_then_blend(cin, cout, ks, N, H, W, bias, dev, ops):
    """cds_dynconv_fused_sbf_f32 (branch convolutions + the blend epilogue on their accumulators, one kernel) against the two
    kernels it replaces (cds_dynconv_branches_sbf_f32 -> cds_dynconv_blend_stats_f32): same arithmetic in the same order, so the
    blended output and the norm-curvature map must be bit-identical; the InstanceNorm statistics are summed in another grouping
    (fp64) and agree to rounding.  Sizes with partial tiles, several images, T that saturates and T that does not."""
    g = torch.Generator().manual_seed(cin * 3 + len(ks) + H)
    K = len(ks)
    x = torch.randn(N, cin, H, W, generator=g)
    aff = torch.stack((0.5 + torch.rand(N, cin, generator=g), 0.3 * torch.randn(N, cin, generator=g),
                       torch.full((N, cin), 0.1)), dim=-1).contiguous()
    co3 = cout + 3
    ws = ops.split_pack_dynconv([(torch.randn(co3, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev) for k in ks])
    bs = torch.randn(K, co3, generator=g).to(dev) if bias else None
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    for T in (1.0, 0.01):
        br = ops.dynconv_branches_sbf(x.to(dev), ws, bs, co3, ks, in_affine=aff.to(dev))
        o2, n2, s2, a2 = ops.dynconv_blend(br, w1, b1, w2, epi, T, 1, stats_slope=0.1)
        o1, n1, s1, a1 = ops.dynconv_fused_sbf(x.to(dev), ws, bs, cout, ks, w1, b1, w2, epi, T, 0.1, in_affine=aff.to(dev))
        assert torch.equal(o1, o2) and torch.equal(n1, n2)
        assert torch.allclose(s1, s2, rtol=1e-12, atol=1e-9)
        assert torch.allclose(a1, a2, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(4, 20, 36), (1, 8, 32), (3, 13, 100)])
def test_visibility_layers_split_bf16(N, H, W, dev, ops):
    """cds_conv2d_k3_relu_sbf_f32 (visibility CNN layers 2 / 3 + head, model.py:14, split-bf16 on the bf16 matrix cores) against
    float64 and the fp32-MFMA kernel it replaces (cds_conv2d_k3_c16_f32): fp32-class, with and without the fused 1x1 head."""
    g = torch.Generator().manual_seed(N * 100 + W)
    x = torch.randn(N, 16, H, W, generator=g).clamp_min(0)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    b = torch.randn(16, generator=g) * 0.2
    hw, hb = torch.randn(16, generator=g) * 0.3, torch.randn(1, generator=g)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1).clamp_min(0)
    ws = ops.split_pack_dynconv([w.to(dev)])
    wcl = w.permute(2, 3, 0, 1).reshape(9, 16, 16).contiguous().to(dev)
    got = ops.conv2d_k3_relu_sbf(x.to(dev), ws, b.to(dev)).cpu()
    old = ops.conv2d_k3_c16(x.to(dev), wcl, b.to(dev)).cpu()
    e_new, e_old = (got.double() - y64).abs().max().item(), (old.double() - y64).abs().max().item()
    ulp = y64.abs().max().item() * 2.0 ** -23
    assert e_new <= 1.5 * e_old + ulp, (e_new, e_old)
    h64 = torch.sigmoid((y64 * hw.double().view(1, 16, 1, 1)).sum(1) + hb.double())
    goth = ops.conv2d_k3_relu_sbf(x.to(dev), ws, b.to(dev), head_w=hw.to(dev), head_b=hb.to(dev)).cpu()
    oldh = ops.conv2d_k3_c16(x.to(dev), wcl, b.to(dev), head_w=hw.to(dev), head_b=hb.to(dev)).cpu()
    assert (goth.double() - h64).abs().max().item() <= 1.5 * (oldh.double() - h64).abs().max().item() + 2.0 ** -23


@pytest.mark.gpu
@pytest.mark.parametrize("V,C,D,h,w,y0,y1", [(4, 8, 40, 64, 136, 16, 40), (2, 16, 9, 40, 72, 0, 8), (3, 32, 12, 32, 48, 24, 32),
                                            (6, 16, 20, 48, 64, 8, 48), (1, 8, 70, 24, 200, 8, 24)])
def test_warp_row_windows_equal_full_grid_rows(V, C, D, h, w, y0, y1, dev, ops):
    """Row-window forms of K1 / K3 (pixel-slab sharding, exchange='slab'): reference-side tensors restricted to rows [y0, y1),
    source maps whole.  The window's entropy and volume must equal the same rows of the full-grid call BIT FOR BIT (positions come
    from the global pixel row), planar and channels-last, both position modes."""
    feats, cams, hyp, ref, src, mats, hyp_d = _random_stage(ops, dev, V, C, D, h, w, seed=60 + V)
    vis = (torch.rand(V, h, w, generator=torch.Generator().manual_seed(4)) * 0.8 + 0.1).to(dev)
    for exact in (True, False):
        ent_full = ops.warp_entropy(ref, src, mats, hyp_d, exact=exact)
        ref_w, hyp_w, vis_w = ref[:, :, y0:y1].contiguous(), hyp_d[:, y0:y1].contiguous(), vis[:, y0:y1].contiguous()
        ent_w = ops.warp_entropy(ref_w, src, mats, hyp_w, exact=exact, window=(h, y0))
        assert torch.equal(ent_w, ent_full[:, y0:y1])
        for cl in (False, True):
            vol_full, vs_full = ops.warp_aggregate(ref, src, vis, mats, hyp_d, channels_last=cl, exact=exact)
            vol_w, vs_w = ops.warp_aggregate(ref_w, src, vis_w, mats, hyp_w, channels_last=cl, exact=exact, window=(h, y0))
            assert torch.equal(vs_w, vs_full[y0:y1])
            assert torch.equal(vol_w, vol_full[:, y0:y1] if cl else vol_full[:, :, y0:y1])
