"""File formats either side of the path: PFM, camera files, pair lists, evaluation samples."""
import os
import struct

import numpy as np
import pytest
import torch

from cds_mvsnet_amd import mvs_io, synth


def test_pfm_layout_and_roundtrip(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4) + 0.5
    p = str(tmp_path / "a.pfm")
    mvs_io.write_pfm(p, a)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n4 3\n-1.000000\n")                    # grey, "w h", negative scale = little endian
    payload = np.frombuffer(raw[len(b"Pf\n4 3\n-1.000000\n"):], dtype="<f4").reshape(3, 4)
    assert np.array_equal(payload[0], a[2])                           # rows are stored bottom-up
    b, scale = mvs_io.read_pfm(p)
    assert scale == 1.0 and np.array_equal(a, b)
    c = np.random.RandomState(0).rand(5, 7, 3).astype(np.float32)
    mvs_io.write_pfm(str(tmp_path / "c.pfm"), c)
    assert open(tmp_path / "c.pfm", "rb").read(2) == b"PF"
    assert np.array_equal(mvs_io.read_pfm(str(tmp_path / "c.pfm"))[0], c)
    # a big-endian file written by another tool
    with open(tmp_path / "be.pfm", "wb") as f:
        f.write(b"Pf\n2 1\n1.0\n" + struct.pack(">2f", 3.0, 4.0))
    assert np.array_equal(mvs_io.read_pfm(str(tmp_path / "be.pfm"))[0], np.array([[3.0, 4.0]], np.float32))
    with pytest.raises(TypeError):
        mvs_io.write_pfm(p, a.astype(np.float64))
    with open(tmp_path / "bad.pfm", "wb") as f:
        f.write(b"P6\n1 1\n1\n")
    with pytest.raises(ValueError):
        mvs_io.read_pfm(str(tmp_path / "bad.pfm"))


def _write_scene(root, scan, n_views, H, W, seed=0):
    from PIL import Image
    cams = synth.make_cameras(n_views, H, W, refine=False, seed=seed)["stage3"][0].numpy()   # full-res intrinsics
    imgs = synth.make_images(n_views, H, W, seed=seed)[0].numpy()
    os.makedirs(os.path.join(root, scan, "images"))
    os.makedirs(os.path.join(root, scan, "cams"))
    for v in range(n_views):
        Image.fromarray((imgs[v].transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(root, scan, "images", f"{v:08d}.jpg"), quality=98)
        with open(os.path.join(root, scan, "cams", f"{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n" + "\n".join(" ".join(f"{x:.8f}" for x in r) for r in cams[v, 0]) + "\n\nintrinsic\n")
            f.write("\n".join(" ".join(f"{x:.8f}" for x in r[:3]) for r in cams[v, 1, :3]) + "\n\n425.0 2.5\n")
    with open(os.path.join(root, scan, "pair.txt"), "w") as f:
        f.write(f"{n_views}\n")
        for v in range(n_views):
            others = [u for u in range(n_views) if u != v]
            f.write(f"{v}\n{len(others)} " + " ".join(f"{u} {100.0 - u:.1f}" for u in others) + "\n")
    return cams


def test_cam_pair_and_sample(tmp_path):
    root = str(tmp_path)
    cams = _write_scene(root, "scan1", 4, 64, 96)
    intr, extr, dmin, dint = mvs_io.read_cam_file(os.path.join(root, "scan1", "cams", "00000002_cam.txt"), interval_scale=1.06)
    assert np.allclose(extr, cams[2, 0], atol=1e-6) and np.allclose(intr, cams[2, 1, :3, :3], atol=1e-6)
    assert dmin == 425.0 and abs(dint - 2.5 * 1.06) < 1e-6
    pairs = mvs_io.read_pair_file(os.path.join(root, "scan1", "pair.txt"))
    assert pairs[1] == (1, [0, 2, 3])
    ds = mvs_io.EvalScenes(root, ["scan1"], nviews=3, ndepths=192, interval_scale=1.0, max_h=64, max_w=96)
    assert len(ds) == 4
    s = ds[0]
    assert s["imgs"].shape == (3, 3, 64, 96) and 0 <= s["imgs"].min() and s["imgs"].max() <= 1
    assert s["depth_values"].shape == (192,) and s["depth_values"][0] == 425.0 and abs(s["depth_values"][1] - 427.5) < 1e-4
    # multi-scale intrinsics: stage1 = K/4, stage2 = K/2, stage3 = K (refine=False)
    K = cams[0, 1, :2, :3]
    assert np.allclose(s["proj_matrices"]["stage3"][0, 1, :2, :3], K, atol=1e-5)
    assert np.allclose(s["proj_matrices"]["stage1"][0, 1, :2, :3], K / 4, atol=1e-5)
    assert s["filename"].format("depth_est", ".pfm") == "scan1/depth_est/00000000.pfm"
    # refine=True shifts the pyramid by one level and adds stage4
    sr = mvs_io.EvalScenes(root, ["scan1"], nviews=3, max_h=64, max_w=96, refine=True)[0]
    assert set(sr["proj_matrices"]) == {"stage1", "stage2", "stage3", "stage4"}
    assert np.allclose(sr["proj_matrices"]["stage4"][0, 1, :2, :3], K, atol=1e-5)
    # write_cam_file round trip
    mvs_io.write_cam_file(os.path.join(root, "c.txt"), s["proj_matrices"]["stage3"][1])
    lines = open(os.path.join(root, "c.txt")).read().split("\n")
    assert lines[0] == "extrinsic" and lines[6] == "intrinsic"
    assert np.allclose(np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4), s["proj_matrices"]["stage3"][1, 0])


@pytest.mark.gpu
def test_inference_harness_end_to_end(tmp_path):
    """The reference's save_depth loop on a synthetic scene: PFM / cam / jpg files land in the reference layout and the
    depth PFM is exactly what the model returned."""
    from cds_mvsnet_amd import CDSMVSNet, infer, seeded_init_
    root = str(tmp_path / "scenes")
    os.makedirs(root)
    _write_scene(root, "scanA", 4, 128, 160, seed=3)
    with open(tmp_path / "list.txt", "w") as f:
        f.write("scanA\n")
    out = str(tmp_path / "out")
    infer.main(["--testpath", root, "--testlist", str(tmp_path / "list.txt"), "--outdir", out, "--num_view", "3",
                "--max_h", "128", "--max_w", "160", "--interval_scale", "1.0", "--fuse", "--thres_view", "1",
                "--conf", "0.0,0.0,0.0"])
    # step 2 of the harness (test.py:386-396): the filtered / fused point cloud of the scan
    from cds_mvsnet_amd import fusion
    pts, col = fusion.read_ply(os.path.join(out, "scanA.ply"))
    assert pts.shape == col.shape and pts.shape[1] == 3 and np.isfinite(pts).all()
    for sub, ext in (("depth_est", ".pfm"), ("confidence", ".pfm"), ("cams", "_cam.txt"), ("images", ".jpg")):
        assert os.path.exists(os.path.join(out, "scanA", sub, f"00000002{ext}"))
    depth, _ = mvs_io.read_pfm(os.path.join(out, "scanA", "depth_est", "00000000.pfm"))
    conf, _ = mvs_io.read_pfm(os.path.join(out, "scanA", "confidence", "00000000.pfm"))
    assert depth.shape == (128, 160) and conf.shape == (128, 160, 3)
    ds = mvs_io.EvalScenes(root, ["scanA"], nviews=3, max_h=128, max_w=160, interval_scale=1.0)
    s = ds[0]
    dev = torch.device("cuda:0")
    model = seeded_init_(CDSMVSNet(depth_interals_ratio=(4.0, 1.5, 0.75)), 0).to(dev).eval()
    with torch.no_grad():
        o = model(torch.from_numpy(s["imgs"])[None].to(dev), {k: torch.from_numpy(v)[None].to(dev) for k, v in s["proj_matrices"].items()},
                  torch.from_numpy(s["depth_values"])[None].to(dev), temperature=0.01)
    assert np.array_equal(depth, o["refined_depth"][0].cpu().numpy())
    assert np.array_equal(conf[..., 2], o["photometric_confidence"][0].cpu().numpy())
    assert 425 <= depth.min() and depth.max() <= 425 + 2.5 * 192


# ------------------------------------------------------------------------------------------------
# G10: bytes written / arrays parsed by the REFERENCE (tests/golden/make_golden.py::g10_formats)
# ------------------------------------------------------------------------------------------------
def test_pfm_and_cam_files_match_reference_bytes(tmp_path, golden):
    """write_pfm / write_cam_file are byte-identical to the reference's save_pfm (datasets/data_io.py:42-71) and write_cam
    (test.py:132-146); read_pfm / read_cam_file return what the reference's readers return for the same files."""
    g = golden("g10_formats")
    for tag in ("grey", "color", "hw1"):
        arr = g[f"pfm_{tag}_array"].numpy()
        p = str(tmp_path / f"{tag}.pfm")
        mvs_io.write_pfm(p, arr)
        assert open(p, "rb").read() == g[f"pfm_{tag}_bytes"].tobytes(), tag
    for tag in ("grey", "color", "be"):
        p = str(tmp_path / f"r_{tag}.pfm")
        with open(p, "wb") as f:
            f.write(g[f"pfm_{tag}_bytes"].tobytes())
        got, scale = mvs_io.read_pfm(p)
        assert np.array_equal(got, g[f"pfm_{tag}_read"].numpy()) and scale == float(g[f"pfm_{tag}_scale"]), tag
    p = str(tmp_path / "00000000_cam.txt")
    mvs_io.write_cam_file(p, g["cam_array"].numpy())
    assert open(p, "rb").read() == g["cam_bytes"].tobytes()
    # the reference's fusion step reads that file back with read_camera_parameters (test.py:82-93): same numbers here
    with open(p) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    assert np.array_equal(np.array(" ".join(lines[1:5]).split(), np.float32).reshape(4, 4), g["cam_read_extrinsic"].numpy())
    assert np.array_equal(np.array(" ".join(lines[7:10]).split(), np.float32).reshape(3, 3), g["cam_read_intrinsic"].numpy())


@pytest.mark.parametrize("dataset", ["dtu", "tt"])
@pytest.mark.parametrize("refine", [False, True])
def test_eval_scene_samples_match_reference_dataset(tmp_path, golden, dataset, refine):
    """EvalScenes on the fixture scene == the sample dict of the reference's MVSDataset (datasets/general_eval.py:118-215):
    images, per-stage projection matrices (incl. the Tanks&Temples +4 px principal point and edge padding, and the
    4-field depth line), depth values and the output file-name pattern, exactly."""
    g = golden("g10_formats")
    for i, name in enumerate(g["scene_file_names"]):
        p = tmp_path / str(name)
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(g[f"scene_file_{i}"].tobytes())
    scan = "scan_" + dataset
    ds = mvs_io.EvalScenes(str(tmp_path / dataset), [scan], nviews=4, ndepths=192, interval_scale=1.06, max_h=64, max_w=80,
                           refine=refine, dataset=dataset)
    assert len(ds) == 4
    for idx in (0, 2):
        key = f"scene_{dataset}_{'refine' if refine else 'norefine'}_{idx}"
        smp = ds[idx]
        assert smp["filename"] == str(g[key + "_filename"])
        assert np.array_equal(smp["imgs"], g[key + "_imgs"].numpy())
        assert np.array_equal(smp["depth_values"], g[key + "_depth_values"].numpy())
        stages = [k[len(key) + 6:] for k in g if k.startswith(key + "_proj_")]
        assert sorted(smp["proj_matrices"]) == sorted(stages)
        for st in stages:
            assert np.array_equal(smp["proj_matrices"][st], g[key + "_proj_" + st].numpy()), st
