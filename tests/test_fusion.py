"""Depth-map filtering / fusion: oracle vs the reference golden on CPU, HIP kernel vs both on the GPU,
and the file-level harness (depth_est / confidence / cams / images -> PLY)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from cds_mvsnet_amd import fusion, mvs_io, synth


def _golden():
    g = load_golden("g8_fusion")                 # float arrays come back as tensors, scalars as numpy
    return g, g


def test_oracle_matches_reference_fusion():
    from oracle import cds_oracle as O
    g, T = _golden()
    out = O.fuse_view(T["depths"][0], T["confs"][0], T["cams"][0], T["depths"][1:], T["confs"][1:], T["cams"][1:],
                      conf=g["conf"].tolist(), thres_disp=float(g["thres_disp"]), thres_view=int(g["thres_view"]))
    assert torch.equal(out["view_masks"], T["view_masks"])          # per-view geometric masks: identical
    assert torch.equal(out["mask"], T["mask"])
    assert (out["depth"] - T["fused"]).abs().max() < 5e-3            # depths ~650: a few fp32 ulps
    assert (out["points"] - T["points"]).abs().max() < 5e-3
    assert 0.2 < float(T["mask"].mean()) < 0.6                       # the fixture exercises both outcomes


def test_camera_chains_layout():
    sc = synth.make_fusion_scene(3, 16, 24, seed=1)
    ch = fusion.camera_chains(sc["cams"][0], sc["cams"][1:])
    assert ch.shape == (2, 100)
    k_ref, e_ref = sc["cams"][0, 1, :3, :3], sc["cams"][0, 0]
    assert torch.allclose(ch[0, :9].view(3, 3) @ k_ref, torch.eye(3), atol=1e-5)
    assert torch.allclose(ch[1, 9:25].view(4, 4) @ e_ref, torch.eye(4), atol=1e-4)
    assert torch.equal(ch[1, 25:41].view(4, 4), sc["cams"][2, 0]) and torch.equal(ch[1, 41:50].view(3, 3), sc["cams"][2, 1, :3, :3])
    assert torch.allclose(ch[1, 59:75].view(4, 4) @ sc["cams"][2, 0], torch.eye(4), atol=1e-4)
    assert torch.equal(ch[0, 75:91].view(4, 4), e_ref) and torch.equal(ch[0, 91:].view(3, 3), k_ref)


def test_ply_roundtrip_and_header(tmp_path):
    rs = np.random.RandomState(0)
    pts, col = rs.randn(11, 3).astype(np.float32), rs.randint(0, 256, (11, 3)).astype(np.uint8)
    p = str(tmp_path / "a.ply")
    fusion.write_ply(p, pts, col)
    raw = open(p, "rb").read()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty float x\n")
    assert len(raw) == raw.index(b"end_header\n") + len(b"end_header\n") + 11 * 15
    p2, c2 = fusion.read_ply(p)
    assert np.array_equal(p2, pts) and np.array_equal(c2, col)


def test_fusion_needs_gpu():
    sc = synth.make_fusion_scene(3, 16, 24, seed=1)
    with pytest.raises(RuntimeError):
        fusion.fuse_view(sc["depths"][0], sc["confs"][0], sc["cams"][0], sc["depths"][1:], sc["confs"][1:], sc["cams"][1:])


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_fusion_vs_reference_golden():
    g, T = _golden()
    dev = "cuda"
    out = fusion.fuse_view(T["depths"][0].to(dev), T["confs"][0].to(dev), T["cams"][0], T["depths"][1:].to(dev),
                           T["confs"][1:].to(dev), T["cams"][1:], conf=g["conf"].tolist(),
                           thres_disp=float(g["thres_disp"]), thres_view=int(g["thres_view"]), want_view_masks=True)
    vm, mask = out["view_masks"].cpu(), out["mask"].cpu()
    # thresholded quantities: a different fp32 summation order may flip pixels that sit exactly on a threshold
    assert (vm != T["view_masks"]).float().mean() < 2e-3
    assert (mask != T["mask"]).float().mean() < 2e-3
    same = (vm == T["view_masks"]).all(0)
    assert ((out["depth"].cpu() - T["fused"]).abs()[same]).max() < 2e-2        # depths ~650 (3e-5 relative)
    assert ((out["points"].cpu() - T["points"]).abs()[:, same]).max() < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,V,conf,tv", [(37, 53, 1, (0.0, 0.0, 0.0), 1), (64, 80, 7, (0.3, 0.2, 0.1), 4)])
def test_hip_fusion_vs_oracle_shapes(h, w, V, conf, tv):
    from oracle import cds_oracle as O
    sc = synth.make_fusion_scene(V + 1, h, w, seed=11 + V)
    exp = O.fuse_view(sc["depths"][0], sc["confs"][0], sc["cams"][0], sc["depths"][1:], sc["confs"][1:], sc["cams"][1:],
                      conf=conf, thres_disp=1.0, thres_view=tv)
    out = fusion.fuse_view(sc["depths"][0].cuda(), sc["confs"][0].cuda(), sc["cams"][0], sc["depths"][1:].cuda(),
                           sc["confs"][1:].cuda(), sc["cams"][1:], conf=conf, thres_disp=1.0, thres_view=tv,
                           want_view_masks=True)
    assert (out["view_masks"].cpu() != exp["view_masks"]).float().mean() < 3e-3
    assert (out["mask"].cpu() != exp["mask"]).float().mean() < 3e-3
    same = (out["view_masks"].cpu() == exp["view_masks"]).all(0)
    assert (out["depth"].cpu() - exp["depth"]).abs()[same].max() < 2e-2


@pytest.mark.gpu
def test_filter_depth_harness(tmp_path):
    """Files in the layout infer.py writes -> fused PLY; point count and coordinates against the oracle."""
    from oracle import cds_oracle as O
    from PIL import Image
    n, h, w = 4, 48, 64
    sc = synth.make_fusion_scene(n, h, w, seed=5, outlier_frac=0.05)
    scan = tmp_path / "out" / "scan1"
    for sub in ("depth_est", "confidence", "cams", "images"):
        os.makedirs(scan / sub)
    for i in range(n):
        mvs_io.write_pfm(str(scan / "depth_est" / f"{i:08d}.pfm"), sc["depths"][i].numpy())
        mvs_io.write_pfm(str(scan / "confidence" / f"{i:08d}.pfm"), np.ascontiguousarray(sc["confs"][i].permute(1, 2, 0).numpy()))
        mvs_io.write_cam_file(str(scan / "cams" / f"{i:08d}_cam.txt"), sc["cams"][i].numpy())
        Image.fromarray((sc["imgs"][i].numpy() * 255).astype(np.uint8)).save(str(scan / "images" / f"{i:08d}.jpg"))
    pairs = tmp_path / "in" / "scan1"
    os.makedirs(pairs)
    with open(pairs / "pair.txt", "w") as f:
        f.write(f"{n}\n")
        for i in range(n):
            others = [j for j in range(n) if j != i]
            f.write(f"{i}\n{len(others)} " + " ".join(f"{j} 1.0" for j in others) + "\n")
    ply = str(tmp_path / "scan1.ply")
    info = fusion.filter_depth(str(pairs), str(scan), ply, conf=(0.1, 0.1, 0.1), thres_disp=1.0, thres_view=2)
    pts, col = fusion.read_ply(ply)
    assert pts.shape[0] == info["points"] and col.shape == pts.shape and np.isfinite(pts).all()
    n_exp, exp_pts = 0, []
    for i in range(n):
        others = [j for j in range(n) if j != i]
        cams = torch.stack([torch.from_numpy(fusion.read_fusion_cam(str(scan / "cams" / f"{j:08d}_cam.txt"))) for j in [i] + others])
        e = O.fuse_view(sc["depths"][i], sc["confs"][i], cams[0], sc["depths"][others], sc["confs"][others], cams[1:],
                        conf=(0.1, 0.1, 0.1), thres_disp=1.0, thres_view=2)
        n_exp += int(e["mask"].sum())
        exp_pts.append(e["points"][:, e["mask"] > 0.5].t().numpy())
    assert abs(pts.shape[0] - n_exp) <= max(3, 0.003 * n_exp)
    assert n_exp > 0.2 * n * h * w
    # the surface is Z = 650 + 40 sin(X/60) cos(Y/50) in world coordinates
    z = 650.0 + 40.0 * np.sin(pts[:, 0] / 60.0) * np.cos(pts[:, 1] / 50.0)
    assert np.median(np.abs(pts[:, 2] - z)) < 2.0       # one pixel is ~11 world units here: bilinear depth error only
    if pts.shape[0] == n_exp:                           # no pixel sat exactly on a threshold: same points, same order
        assert np.abs(pts - np.concatenate(exp_pts)).max() < 5e-2
