"""GPU tests of cds_stage_inputs_f32 (csrc/warp.hip): the reference-signature boundary of one stage (models/model.py:16-40: a list
over the source views of {'ref': (fea, nc_sum, nc), 'src': (fea, nc_sum, _)}) gathered for K1 / K3 in one launch.  It only moves
data and takes maxima, so every output is compared BIT for bit with the separate launches it replaces (torch.stack,
cds_chw_to_hwc_f32 per view, abs / amax / product, cds_pair_mean_f32 + cds_view_mean_f32), and StageNet.forward through it is
compared bit for bit with StageNet.forward through the old harness."""
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


def _maps(V, C, h, w, seed, dev):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: [(torch.randn(*s, generator=g) * (0.2 + i)).to(dev) for i in range(V)]
    return mk(C, h, w), mk(C, h, w), mk(h, w), mk(h, w), mk(h, w)


@pytest.mark.parametrize("V,C,h,w", [(1, 8, 5, 7), (4, 8, 64, 80), (2, 16, 37, 53), (3, 32, 16, 24), (8, 8, 33, 129), (5, 16, 1, 300)])
def test_stage_inputs_equal_the_separate_launches(V, C, h, w, dev, ops):
    rf, sf, nc, ncs_r, ncs_s = _maps(V, C, h, w, 17 * V + C + h, dev)
    ref, src, ref_nc, nc_mean, bound = ops.stage_inputs(rf, sf, nc, ncs_r, ncs_s)
    assert torch.equal(ref, torch.stack(rf))
    assert torch.equal(src, torch.stack([ops.chw_to_hwc(s) for s in sf]))
    assert torch.equal(ref_nc, torch.stack(nc))
    assert torch.equal(nc_mean, ops.view_mean(ops.pair_mean(torch.stack(ncs_r + ncs_s), V)))
    want = (torch.stack(rf).abs().amax() * torch.stack(sf).abs().amax()).reshape(1)
    assert bound.shape == (1,) and torch.equal(bound, want)
    # the optional parts
    ref2, src2, none_nc, none_mean, none_bound = ops.stage_inputs(rf, sf, want_bound=False)
    assert torch.equal(ref2, ref) and torch.equal(src2, src) and none_nc is None and none_mean is None and none_bound is None


def test_stage_inputs_bound_edge_values(dev, ops):
    """The maximum is taken on |x| as an ordered integer: negative extremes count, an all-zero map gives 0, a NaN gives NaN (amax's
    behaviour), an infinity gives inf; repeated calls do not leak state."""
    rf, sf, *_ = _maps(3, 8, 9, 11, 5, dev)
    rf[1][3, 4, 5] = -77.5
    sf[2][0, 0, 0] = -3.0
    for _ in range(3):
        b = ops.stage_inputs(rf, sf)[4]
        assert torch.equal(b, (torch.tensor(77.5, device=dev) * torch.stack(sf).abs().amax().clamp_min(3.0)).reshape(1))
    z = [torch.zeros_like(t) for t in rf]
    assert float(ops.stage_inputs(z, sf)[4]) == 0.0
    sf[0][7, 8, 10] = float("inf")
    assert float(ops.stage_inputs(rf, sf)[4]) == float("inf")
    rf[0][0, 0, 0] = float("nan")
    assert torch.isnan(ops.stage_inputs(rf, sf)[4]).item()


def test_stage_inputs_rejects_what_it_does_not_cover(dev, ops):
    rf, sf, nc, a, b = _maps(2, 8, 6, 10, 3, dev)
    assert ops.stage_inputs_supported(rf, sf, (nc, a, b))
    assert not ops.stage_inputs_supported(rf, sf[:1])
    assert not ops.stage_inputs_supported([t.double() for t in rf], sf)
    assert not ops.stage_inputs_supported([t.transpose(1, 2) for t in rf], [t.transpose(1, 2) for t in sf])
    assert not ops.stage_inputs_supported([t[:4] for t in rf], [t[:4] for t in sf])          # C = 4
    assert not ops.stage_inputs_supported(rf * 5, sf * 5)                                     # 10 views > MAX_VIEWS
    with pytest.raises(ValueError):
        ops.stage_inputs(rf, sf[:1])
    with pytest.raises(ValueError):
        ops.stage_inputs(rf, sf, nc, a, None)


@pytest.mark.parametrize("B", [1, 2])
def test_stage_net_forward_is_unchanged_by_the_fused_boundary(B, dev, ops, seeded_state, monkeypatch):
    """StageNet.forward (reference signature) through cds_stage_inputs_f32 against the same call with the fused boundary disabled
    (the stack / transpose / amax harness): depth, confidence and norm_curv bit-identical."""
    from cds_mvsnet_amd import synth
    model = seeded_state(False).to(dev)
    V, C, h, w, D = 3, 8, 32, 40, 16
    g = torch.Generator().manual_seed(11)
    feats = []
    for _ in range(V):
        mk = lambda c: torch.tanh(torch.randn(B, c, h, w, generator=g)).to(dev)
        feats.append({"ref": (mk(C), mk(1), mk(1)), "src": (mk(C), mk(1), mk(1))})
    pm = torch.stack([synth.make_cameras(V + 1, h, w, refine=False, seed=s)["stage3"][0] for s in range(B)])
    dv = torch.linspace(425.0, 905.0, D).view(1, D, 1, 1).expand(B, D, h, w).contiguous().to(dev)
    run = lambda: model.stage_net(feats, pm, depth_values=dv, num_depth=D, cost_regularization=model.cost_regularization[2], stage_idx=2)
    with torch.no_grad():
        fused = run()
        monkeypatch.setattr(ops, "stage_inputs_supported", lambda *a, **k: False)
        plain = run()
    for k in ("depth", "photometric_confidence", "norm_curv"):
        assert fused[k].shape == plain[k].shape
        assert torch.equal(fused[k], plain[k]), k
