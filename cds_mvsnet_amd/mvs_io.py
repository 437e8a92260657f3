"""On-disk formats either side of the plane-sweep path: MVSNet-style scenes in, depth maps out.

* PFM images (reference: datasets/data_io.py:6-71): header ``Pf`` (1 channel) / ``PF`` (3 channels), ``w h``, a
  scale whose SIGN gives the byte order (negative = little endian), then rows stored bottom-up.
* camera files ``cams/%08d_cam.txt`` (datasets/general_eval.py:64-86): ``extrinsic`` + 4x4, ``intrinsic`` + 3x3,
  then ``depth_min depth_interval [num_depth [depth_max]]``.
* ``pair.txt`` (general_eval.py:40-55): number of viewpoints, then per viewpoint the reference id and
  ``N id0 score0 id1 score1 ...``.
* :class:`EvalScenes` builds the sample dict the model consumes (general_eval.py:118-215): ``imgs [N,3,H,W]`` in
  [0,1], ``proj_matrices`` {stageK: [N,2,4,4]} with per-stage intrinsics, ``depth_values [D]``, ``filename``.

Image decoding / resizing uses PIL (OpenCV is not available in this image): bilinear resizes are not bit-identical to
``cv2.resize``; everything else (camera scaling, crops, depth ranges, file layout) follows the reference.
"""
from __future__ import annotations

import os
import re
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------
# PFM
# ------------------------------------------------------------------------------------------------
def read_pfm(path: str) -> Tuple[np.ndarray, float]:
    with open(path, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError(f"{path}: not a PFM file")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError(f"{path}: malformed PFM header")
        w, h = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (h, w, 3) if header == "PF" else (h, w)
    if data.size != int(np.prod(shape)):
        raise ValueError(f"{path}: payload size does not match the header")
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def write_pfm(path: str, image: np.ndarray, scale: float = 1.0) -> None:
    image = np.asarray(image)
    if image.dtype != np.float32:
        raise TypeError("PFM payload must be float32")
    if image.ndim == 3 and image.shape[2] == 3:
        header = "PF"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        header = "Pf"
    else:
        raise ValueError("PFM image must be HxW, HxWx1 or HxWx3")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder in "=|" and sys.byteorder == "little")
    with open(path, "wb") as f:
        f.write(f"{header}\n{image.shape[1]} {image.shape[0]}\n{(-scale if little else scale):f}\n".encode("utf-8"))
        np.ascontiguousarray(np.flipud(image)).tofile(f)


# ------------------------------------------------------------------------------------------------
# cameras and pairs
# ------------------------------------------------------------------------------------------------
def read_cam_file(path: str, ndepths: int = 192, interval_scale: float = 1.0) -> Tuple[np.ndarray, np.ndarray, float, float]:
    """-> (intrinsic 3x3 at FULL image resolution, extrinsic 4x4, depth_min, depth_interval)."""
    with open(path) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extr = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intr = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    fields = lines[11].split()
    dmin, dint = float(fields[0]), float(fields[1])
    if len(fields) >= 3:  # "depth_min interval num_depth [depth_max]": re-spread the range over ndepths planes
        dmax = dmin + int(float(fields[2])) * dint
        dint = (dmax - dmin) / ndepths
    return intr, extr, dmin, dint * interval_scale


def write_cam_file(path: str, cam: np.ndarray) -> None:
    """cam [2,4,4] (extrinsic, intrinsic in [:3,:3]) in the layout the reference's fusion step reads back; byte-identical
    to the reference's writer (test.py:132-149: every number as ``str(np.float32)`` followed by one space, then row 3
    of the intrinsic slot on the depth-range line)."""
    cam = np.asarray(cam)
    with open(path, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
        # the reference closes the file with row 3 of the intrinsic slot (the dataset leaves it zero): "0.0 0.0 0.0 0.0"
        f.write("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")


def read_pair_file(path: str) -> List[Tuple[int, List[int]]]:
    with open(path) as f:
        n = int(f.readline())
        out = []
        for _ in range(n):
            ref = int(f.readline().rstrip())
            src = [int(x) for x in f.readline().rstrip().split()[1::2]]
            out.append((ref, src))
    return out


# ------------------------------------------------------------------------------------------------
# evaluation scenes -> model samples
# ------------------------------------------------------------------------------------------------
class EvalScenes:
    """MVSNet-format test scenes (``<root>/<scan>/{images|images_post}/%08d.jpg, cams/%08d_cam.txt, pair.txt``)."""

    def __init__(self, root: str, scans: Sequence[str], nviews: int = 5, ndepths: int = 192, interval_scale: float = 1.06,
                 max_h: int = 512, max_w: int = 640, refine: bool = False, dataset: str = "dtu"):
        self.root, self.nviews, self.ndepths = root, nviews, ndepths
        self.interval_scale, self.max_h, self.max_w = interval_scale, max_h, max_w
        self.refine, self.dataset = refine, dataset
        self.metas: List[Tuple[str, int, List[int]]] = []
        for scan in scans:
            for ref, src in read_pair_file(os.path.join(root, scan, "pair.txt")):
                if not src:
                    continue
                if len(src) < nviews - 1:  # pad with the best view like the reference does
                    src = src + [src[0]] * (nviews - 1 - len(src))
                self.metas.append((scan, ref, src[:nviews - 1]))

    def __len__(self) -> int:
        return len(self.metas)

    def _image(self, scan: str, vid: int) -> np.ndarray:
        from PIL import Image
        for sub in ("images_post", "images"):
            p = os.path.join(self.root, scan, sub, f"{vid:08d}.jpg")
            if os.path.exists(p):
                break
        img = np.asarray(Image.open(p).convert("RGB"), dtype=np.float32) / 255.0
        if self.dataset == "tt":  # Tanks & Temples: 1080 -> 1088 rows (general_eval.py:92-93)
            img = np.pad(img, ((4, 4), (0, 0), (0, 0)), "edge")
        return img

    def __getitem__(self, idx: int) -> Dict[str, object]:
        from PIL import Image
        scan, ref, srcs = self.metas[idx]
        imgs, mats, depth_values = [], [], None
        for i, vid in enumerate([ref] + srcs):
            img = self._image(scan, vid)
            intr, extr, dmin, dint = read_cam_file(os.path.join(self.root, scan, "cams", f"{vid:08d}_cam.txt"),
                                                   self.ndepths, self.interval_scale)
            if self.dataset == "tt":
                intr[1, 2] += 4
            intr[:2, :] /= 4.0  # cam files hold full-resolution intrinsics; stage1 works at 1/4 (general_eval.py:75)
            h, w = img.shape[:2]
            if (h, w) != (self.max_h, self.max_w):
                intr[0, :] *= self.max_w / w
                intr[1, :] *= self.max_h / h
                pil = Image.fromarray(np.clip(img * 255.0, 0, 255).astype(np.uint8)).resize((self.max_w, self.max_h), Image.BILINEAR)
                img = np.asarray(pil, dtype=np.float32) / 255.0
            imgs.append(img)
            m = np.zeros((2, 4, 4), dtype=np.float32)
            m[0] = extr
            m[1, :3, :3] = intr
            mats.append(m)
            if i == 0:
                depth_values = np.arange(dmin, dint * (self.ndepths - 0.5) + dmin, dint, dtype=np.float32)
        imgs = np.stack(imgs).transpose(0, 3, 1, 2)
        base = np.stack(mats)

        def scaled(f):
            m = base.copy()
            m[:, 1, :2, :] = base[:, 1, :2, :] * f
            return m
        if self.refine:
            proj = {"stage1": scaled(0.5), "stage2": base, "stage3": scaled(2), "stage4": scaled(4)}
        else:
            proj = {"stage1": base, "stage2": scaled(2), "stage3": scaled(4)}
        return {"imgs": imgs, "proj_matrices": proj, "depth_values": depth_values,
                "filename": scan + "/{}/" + f"{ref:08d}" + "{}"}


def nearest_resize(a: np.ndarray, h: int, w: int) -> np.ndarray:
    """cv2.INTER_NEAREST-style resize of [H,W(,C)] used for the confidence / image side outputs (test.py:232-243)."""
    H, W = a.shape[:2]
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return a[ys][:, xs]


def save_outputs(outdir: str, filename: str, depth: np.ndarray, confs: Sequence[np.ndarray], cam: np.ndarray,
                 img_chw: np.ndarray) -> None:
    """Write depth_est / confidence PFMs, the camera and the image where the reference's fusion step expects them
    (test.py:216-248).  ``confs``: per-stage confidence maps, resized to the depth resolution and stacked as 3 channels."""
    from PIL import Image
    h, w = depth.shape
    paths = {k: os.path.join(outdir, filename.format(k, ext)) for k, ext in
             (("depth_est", ".pfm"), ("confidence", ".pfm"), ("cams", "_cam.txt"), ("images", ".jpg"))}
    for p in paths.values():
        os.makedirs(os.path.dirname(p), exist_ok=True)
    write_pfm(paths["depth_est"], depth.astype(np.float32))
    conf3 = np.stack([nearest_resize(c.astype(np.float32), h, w) for c in confs], axis=-1)
    write_pfm(paths["confidence"], np.ascontiguousarray(conf3))
    write_cam_file(paths["cams"], cam)
    img = nearest_resize(np.transpose(img_chw, (1, 2, 0)), h, w)
    Image.fromarray(np.clip(img * 255, 0, 255).astype(np.uint8)).save(paths["images"], quality=95)
