"""Depth-map filtering and point-cloud fusion on the GPU.

Mirrors step 2 of the reference's ``test.py`` (``filter_depth``, test.py:324-383, built on fusion.py): for every
reference view read back ``depth_est/ confidence/ cams/ images/`` written by :mod:`cds_mvsnet_amd.infer`, keep the
pixels whose three stage confidences exceed ``conf`` and that re-project consistently (pixel distance < ``thres_disp``,
relative depth difference < 1 %) into at least ``thres_view`` source views, average the consistent depths and emit the
world-space points with their colours as one binary PLY.

All per-pixel work is one launch of ``cds_depth_fusion_f32`` per reference view; the host side only reads files, inverts
the 3x3 / 4x4 camera matrices (fp32, CPU) and compacts the masked points.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .mvs_io import read_pair_file, read_pfm


def read_fusion_cam(path: str) -> np.ndarray:
    """``cams/%08d_cam.txt`` as written by save_outputs -> [2,4,4] (extrinsic; intrinsic in [:3,:3], [3,3]=1)
    (test.py:85-95, 285-288)."""
    with open(path) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    cam[0] = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    cam[1, :3, :3] = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    cam[1, 3, 3] = 1.0
    return cam


def camera_chains(ref_cam: torch.Tensor, src_cams: torch.Tensor) -> torch.Tensor:
    """ref_cam [2,4,4], src_cams [V,2,4,4] (CPU fp32) -> [V,100] blocks for cds_depth_fusion_f32:
    Kinv_ref Einv_ref E_src K_src | Kinv_src Einv_src E_ref K_ref (fusion.py:24-46 applies them in this order)."""
    ref_cam = ref_cam.detach().to("cpu", torch.float32)
    src_cams = src_cams.detach().to("cpu", torch.float32)
    k_ref, e_ref = ref_cam[1, :3, :3], ref_cam[0]
    kinv_ref, einv_ref = torch.inverse(k_ref), torch.inverse(e_ref)
    rows = []
    for cam in src_cams:
        k_src, e_src = cam[1, :3, :3], cam[0]
        rows.append(torch.cat([kinv_ref.reshape(9), einv_ref.reshape(16), e_src.reshape(16), k_src.reshape(9),
                               torch.inverse(k_src).reshape(9), torch.inverse(e_src).reshape(16), e_ref.reshape(16),
                               k_ref.reshape(9)]))
    return torch.stack(rows).contiguous()


def fuse_view(ref_depth: torch.Tensor, ref_conf: torch.Tensor, ref_cam: torch.Tensor, src_depths: torch.Tensor,
              src_confs: torch.Tensor, src_cams: torch.Tensor, conf: Sequence[float] = (0.0, 0.0, 0.0),
              thres_disp: float = 1.0, thres_view: int = 3, want_view_masks: bool = False) -> Dict[str, torch.Tensor]:
    """One reference view: device tensors ref_depth [h,w], ref_conf [3,h,w], src_depths [V,h,w], src_confs [V,3,h,w];
    cameras ref_cam [2,4,4], src_cams [V,2,4,4] (any device; they are inverted on the host)."""
    dev = ref_depth.device
    cams = camera_chains(ref_cam, src_cams).to(dev)
    fused, mask, points, vm = ops.depth_fusion(ref_depth.contiguous(), ref_conf.contiguous(), src_depths.contiguous(),
                                               src_confs.contiguous(), cams, conf, thres_disp, 0.01, thres_view,
                                               want_view_masks)
    return {"depth": fused, "mask": mask, "points": points, "view_masks": vm}


def write_ply(path: str, points: np.ndarray, colors: np.ndarray) -> None:
    """Binary little-endian PLY with x,y,z float32 + red,green,blue uint8 vertices (what plyfile writes in
    test.py:370-382)."""
    n = int(points.shape[0])
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    rec["red"], rec["green"], rec["blue"] = colors[:, 0], colors[:, 1], colors[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\n" f"element vertex {n}\n"
              "property float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        rec.tofile(f)


def read_ply(path: str) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        rec = np.frombuffer(f.read(), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"),
                                             ("blue", "u1")], count=n)
    return np.stack([rec["x"], rec["y"], rec["z"]], -1), np.stack([rec["red"], rec["green"], rec["blue"]], -1)


def _load_view(scan_folder: str, vid: int):
    from PIL import Image
    depth = read_pfm(os.path.join(scan_folder, "depth_est", f"{vid:08d}.pfm"))[0]
    conf = read_pfm(os.path.join(scan_folder, "confidence", f"{vid:08d}.pfm"))[0]
    cam = read_fusion_cam(os.path.join(scan_folder, "cams", f"{vid:08d}_cam.txt"))
    return np.ascontiguousarray(depth, dtype=np.float32), np.ascontiguousarray(conf.transpose(2, 0, 1)), cam, \
        lambda: np.asarray(Image.open(os.path.join(scan_folder, "images", f"{vid:08d}.jpg")), dtype=np.float32) / 255.0


def filter_depth(pair_folder: str, scan_folder: str, plyfilename: str, conf: Sequence[float] = (0.0, 0.0, 0.0),
                 thres_disp: float = 1.0, thres_view: int = 3, n_src_views: int = 10, device: str = "cuda",
                 verbose: bool = False) -> Dict[str, float]:
    """The reference's ``filter_depth`` for one scan: -> PLY at ``plyfilename`` and mean photo/geo/final mask rates."""
    pairs = read_pair_file(os.path.join(pair_folder, "pair.txt"))
    cache: Dict[int, tuple] = {}

    def view(vid):
        if vid not in cache:
            d, c, cam, img = _load_view(scan_folder, vid)
            cache[vid] = (torch.from_numpy(d).to(device), torch.from_numpy(c).to(device), torch.from_numpy(cam), img)
        return cache[vid]

    pts_all: List[np.ndarray] = []
    col_all: List[np.ndarray] = []
    rates = []
    for ref, srcs in pairs:
        srcs = srcs[:n_src_views]
        if not srcs:
            continue
        rd, rc, rcam, rimg = view(ref)
        sv = [view(s) for s in srcs]
        out = fuse_view(rd, rc, rcam, torch.stack([s[0] for s in sv]), torch.stack([s[1] for s in sv]),
                        torch.stack([s[2] for s in sv]), conf, thres_disp, thres_view)
        keep = out["mask"] > 0.5
        pts = out["points"][:, keep].t().contiguous().cpu().numpy()
        img = torch.from_numpy(np.ascontiguousarray(rimg())).to(device)            # [h,w,3]
        col = (img[keep] * 255).to(torch.uint8).cpu().numpy()
        pts_all.append(pts)
        col_all.append(col)
        rates.append(float(keep.float().mean()))
        if verbose:
            print(f"processing {scan_folder}, ref-view{ref:02d}, final-mask:{rates[-1]:.4f}")
    p_all = np.concatenate(pts_all, 0) if pts_all else np.zeros((0, 3), np.float32)
    c_all = np.concatenate(col_all, 0) if col_all else np.zeros((0, 3), np.uint8)
    write_ply(plyfilename, p_all, c_all)
    return {"points": int(p_all.shape[0]), "mean_final_mask": float(np.mean(rates)) if rates else 0.0}
