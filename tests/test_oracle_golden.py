"""The CPU oracle (oracle/cds_oracle.py) against the golden vectors captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle; it runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cds_oracle as O


def _pairs_from(g):
    V = g["ref_fea"].shape[0]
    return [{"ref": (g["ref_fea"][v:v + 1], g["ref_nc_sum"][v:v + 1], g["ref_nc"][v:v + 1]),
             "src": (g["src_fea"][v:v + 1], g["src_nc_sum"][v:v + 1], None)} for v in range(V)]


def test_warp_exact_mode_is_bit_identical_to_reference():
    g = load_golden("g1_warp_aggregate_a")
    cams = g["cams"]
    P_ref, P_src = O.compose_projection(cams[:, 0]), O.compose_projection(cams[:, 1])
    M = O.relative_projection(P_src, P_ref)[0]
    assert torch.equal(torch.cat((M[:3, :3].reshape(9), M[:3, 3])), g["mats"][0])
    w_exact = O.warp_volume(g["src_fea"][0:1], P_src, P_ref, g["hyp"], exact=True)[0]
    w_fast = O.warp_volume(g["src_fea"][0:1], P_src, P_ref, g["hyp"], exact=False)[0]
    assert torch.equal(w_fast, g["warped0"])
    mism = (w_exact != g["warped0"]).float().mean().item()
    assert mism < 1e-4, f"exact-mode warp differs from F.grid_sample on {mism:.2e} of the voxels"
    assert (w_exact - g["warped0"]).abs().max().item() < 1e-6


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("exact", [True, False])
def test_aggregate_and_stage(tag, exact, seeded_state):
    g = load_golden(f"g1_warp_aggregate_{tag}")
    sd = seeded_state(False).state_dict()
    stage = int(g["stage"])
    agg = O.aggregate_views(_pairs_from(g), g["cams"], g["hyp"], sd, stage, exact=exact)
    assert (torch.cat(agg["entropy"])[:, 0] - g["entropy"]).abs().max() < 5e-6
    assert (torch.cat(agg["vis_w"])[:, 0] - g["vis_w"]).abs().max() < 5e-6
    assert (agg["volume_mean"][0] - g["volume_mean"]).abs().max() < 2e-6
    out = O.stage_forward(_pairs_from(g), g["cams"], g["hyp"], sd, stage, exact=exact)
    assert (out["depth"] - g["depth"]).abs().mean() < 1e-3
    assert (out["photometric_confidence"] - g["conf"]).abs().mean() < 1e-4
    assert (out["norm_curv"] - g["norm_curv"]).abs().max() < 1e-6


@pytest.mark.parametrize("tag,C", [("c8", 8), ("c16", 16), ("c32", 32), ("c8_wide", 8)])
def test_costreg(tag, C):
    from cds_mvsnet_amd import CostRegNet, seeded_init_
    g = load_golden(f"g2_costreg_{tag}")
    net = CostRegNet(C, 8)
    seeded_init_(net, int(g["seed"]))
    sd = {"cr." + k: v for k, v in net.state_dict().items()}
    out = O.cost_regularization(g["volume"].unsqueeze(0), sd, "cr")[0, 0]
    assert (out - g["cost_reg"]).abs().max() < 1e-5


@pytest.mark.parametrize("tag", ["d48", "d8", "d192"])
def test_regression(tag):
    g = load_golden(f"g3_regress_{tag}")
    prob, depth, conf = O.softargmin(g["prob_pre"].unsqueeze(0), g["hyp"].unsqueeze(0))
    assert (prob[0] - g["prob"]).abs().max() < 1e-6
    assert (depth[0] - g["depth"]).abs().max() < 2e-4
    assert (conf[0] - g["conf"]).abs().max() < 1e-6
    assert torch.allclose(O.confidence_window_reference(prob)[0].gather(
        0, (prob[0] * torch.arange(prob.shape[1]).view(-1, 1, 1)).sum(0).long().clamp(0, prob.shape[1] - 1).unsqueeze(0))[0],
        conf[0], atol=1e-6)


def test_hypotheses():
    g = load_golden("g4_hypotheses")
    dv = g["depth_values"]
    H, W = int(g["H"]), int(g["W"])
    dmin, dmax = dv[:, 0].view(1, 1, 1), dv[:, -1].view(1, 1, 1)
    dint = (dv[:, 1] - dv[:, 0]).view(1, 1, 1)
    s1 = O.stage_hypotheses(dv, 48, 4.0 * dint, dmin, dmax, H, W, 4)
    assert torch.equal(s1, g["s1"])
    for tag in ("s2", "s3", "s2x4"):
        D, ratio, scale = g[tag + "_meta"]
        out = O.stage_hypotheses(g[tag + "_prev"], int(D), float(ratio) * dint, dmin, dmax, H, W, int(scale))
        assert torch.equal(out, g[tag]), tag
        assert out.min() >= 425.0 and out.max() <= 902.5


def test_epipoles_and_dynconv():
    from cds_mvsnet_amd.model import DynamicConv
    from cds_mvsnet_amd import seeded_init_
    g = load_golden("g5_dynconv")
    Fm = O.fundamental_matrix(g["cams"][:, 0], g["cams"][:, 1])
    assert torch.equal(Fm, g["fmatrix"])
    assert torch.equal(O.epipole_from_F(Fm), g["epipole_ref"])
    assert torch.equal(O.epipole_from_F(Fm.transpose(1, 2)), g["epipole_src"])
    dc = DynamicConv(3, 8, (3, 7, 11))
    seeded_init_(dc, 7)
    sd = {"dc." + k: v for k, v in dc.state_dict().items()}
    for T in (1.0, 0.1, 0.01):
        y, nc = O.dynamic_conv(g["img"].unsqueeze(0), g["epipole_ref"], T, sd, "dc", (3, 7, 11))
        assert (y[0] - g[f"y_T{T}"]).abs().max() < 1e-5, T
        assert (nc[0, 0] - g[f"nc_T{T}"]).abs().max() < 1e-5, T


def test_featurenet():
    from cds_mvsnet_amd import FeatureNet, seeded_init_
    g = load_golden("g5_featurenet")
    net = FeatureNet(8)
    seeded_init_(net, 7)
    sd = {"feature." + k: v for k, v in net.state_dict().items()}
    gn = load_golden("g9_featurenet_noise")
    for T in (1.0, 0.01):
        out = O.feature_net(g["img"].unsqueeze(0), g["epipole"], T, sd)
        for s in ("stage1", "stage2", "stage3"):
            assert (out[s][0][0] - g[f"{s}_fea_T{T}"]).abs().max() < 2e-5, (s, T)
            assert (out[s][1][0, 0] - g[f"{s}_ncsum_T{T}"]).abs().max() < 2e-5, (s, T)
            assert (out[s][2][0, 0] - g[f"{s}_nc_T{T}"]).abs().max() < 2e-5, (s, T)
            # and against the float64 evaluation of the reference module: inside the reference's own fp32 envelope
            for j, key in enumerate(("fea", "ncsum", "nc")):
                k = f"{s}_{key}_T{T}"
                got = out[s][j][0] if j == 0 else out[s][j][0, 0]
                env = max(float(gn[k + "_ref32_vs_f64_max"]), float(gn[k + "_native32_vs_f64_max"]))
                assert (got - gn[k + "_f64"]).abs().max() <= 1.5 * env, (k, env)


@pytest.mark.parametrize("tag,refine", [("norefine", False), ("refine", True)])
def test_full_forward(tag, refine, seeded_state):
    g = load_golden(f"g6_forward_{tag}")
    sd = seeded_state(refine).state_dict()
    cams = {k[4:]: v for k, v in g.items() if k.startswith("cam_")}
    out = O.forward(g["imgs"], cams, g["depth_values"], sd, refine=refine, temperature=0.01, exact=False)
    for s in (1, 2, 3):
        st = out[f"stage{s}"]
        assert torch.equal(st["_hyp"][0, :, ::4, ::4], g[f"stage{s}_hyp"]) or \
            (st["_hyp"][0, :, ::4, ::4] - g[f"stage{s}_hyp"]).abs().mean() < 1e-3
        assert (st["depth"] - g[f"stage{s}_depth"]).abs().mean() < 1e-3, s
        assert (st["photometric_confidence"] - g[f"stage{s}_conf"]).abs().mean() < 1e-3, s
        assert (st["norm_curv"] - g[f"stage{s}_norm_curv"]).abs().max() < 1e-4, s
    assert (out["refined_depth"] - g["refined_depth"]).abs().mean() < 1e-3


def test_final_loss_equals_reference_g11(golden):
    """losses.final_loss (masked sums instead of the reference's boolean gathers) against the reference's loss, depth loss and
    input gradients captured in G11 (models/losses.py:6-48)."""
    from cds_mvsnet_amd import final_loss
    g = golden("g11_loss")
    stages = ("stage1", "stage2", "stage3")
    for tag, kw in (("w", dict(dlossw=[0.5, 1.0, 2.0])), ("nw", dict())):
        inp = {k: {"depth": g[f"{k}.depth"].clone().requires_grad_(True), "norm_curv": g[f"{k}.norm_curv"].clone().requires_grad_(True),
                   "feat_distance": g[f"{k}.feat_distance"].clone().requires_grad_(True), "feat_target": g[f"{k}.feat_target"]} for k in stages}
        inp["refined_depth"] = g["refined_depth"].clone().requires_grad_(True)
        gt = {k: g[f"{k}.gt"] for k in stages + ("stage4",)}
        mask = {k: g[f"{k}.mask"] for k in stages + ("stage4",)}
        loss, dl = final_loss(inp, gt, mask, depth_interval=g["interval"], **kw)
        loss.backward()
        assert abs(loss.item() - g[f"{tag}.loss"].item()) <= 1e-5 * abs(g[f"{tag}.loss"].item())
        assert abs(dl.item() - g[f"{tag}.depth_loss"].item()) <= 1e-5 * abs(g[f"{tag}.depth_loss"].item())
        for k in stages:
            for n in ("depth", "norm_curv", "feat_distance"):
                ref = g[f"{tag}.grad.{k}.{n}"]
                assert (inp[k][n].grad - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), (tag, k, n)
        ref = g[f"{tag}.grad.refined_depth"]
        assert (inp["refined_depth"].grad - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
