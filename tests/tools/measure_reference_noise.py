"""Build-container-only diagnostic (needs /root/reference): the reference FeatureNet's OWN fp32 round-off — oneDNN vs
native convolutions, and fp32 vs a float64 evaluation of the same module — with this repo's seeded weights and with the
three trained checkpoints, at T = 1 and the evaluation temperature T = 0.01.  Appends the table to
profiles/r02_trained_blend_regime.md (run measure_blend_regime.py first).

    PYTHONDONTWRITEBYTECODE=1 python tests/tools/measure_reference_noise.py
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
torch.set_num_threads(8)

from models.module import FeatureNet as RefFeat  # noqa: E402  (reference, imported read-only)

from cds_mvsnet_amd import seeded_init_, synth  # noqa: E402
from cds_mvsnet_amd.infer import _placeholder_pickle  # noqa: E402

CKPTS = {"dtu_only": "dtu_only/checkpoint-epoch24.pth", "both_dtu_blended": "both_dtu_blended/cds_mvsnet.ckpt",
         "fine_tuning_on_blended": "fine_tuning_on_blended/cds_mvsnet.ckpt"}


def nets():
    n = RefFeat(base_channels=8)
    seeded_init_(n, 7)
    out = {"seeded (seed 7)": n.eval()}
    for name, pth in CKPTS.items():
        ck = torch.load("/root/reference/pretrained/" + pth, map_location="cpu", weights_only=False, pickle_module=_placeholder_pickle)
        sd = {k.replace("module.", "")[len("feature."):]: v for k, v in ck["state_dict"].items()
              if k.replace("module.", "").startswith("feature.")}
        n = RefFeat(base_channels=8)
        n.load_state_dict(sd, strict=True)
        out[name] = n.eval()
    return out


def worst(a, b):
    m = 0.0
    for s in a:
        for j in range(3):
            sc = max(1.0, float(b[s][j].abs().max()))
            m = max(m, float((a[s][j].double() - b[s][j].double()).abs().max()) / sc)
    return m


@torch.no_grad()
def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g5_featurenet.npz"))
    small = torch.from_numpy(g["img"]).unsqueeze(0)
    epi = torch.from_numpy(g["epipole"])
    big = synth.make_images(1, 256, 320, seed=4)[0, :1]
    rows = []
    for name, net in nets().items():
        n64 = copy.deepcopy(net).double()
        for T in (1.0, 0.01):
            for tag, im in (("64x96", small), ("256x320", big)):
                a = net(im, epi, T)
                with torch.backends.mkldnn.flags(enabled=False):
                    b = net(im, epi, T)
                c = n64(im.double(), epi.double(), T)
                rows.append(f"| {name} | {T} | {tag} | {worst(a, b):.1e} | {worst(a, c):.1e} | {worst(b, c):.1e} |")
    lines = ["", "## The reference FeatureNet's own fp32 round-off (max abs over all nine outputs, relative to max(1, |x|))", "",
             "| weights | T | image | fp32 oneDNN vs fp32 native conv | fp32 oneDNN vs float64 | fp32 native vs float64 |", "|---|---|---|---|---|---|"] + rows
    lines += ["", "Reading: with the seeded weights the reference's fp32 forward is itself only accurate to 1.5e-4 (64x96) / 3.8e-4",
              "(256x320) at T = 0.01 — the softmax(./T) blend amplifies convolution round-off — and the trained checkpoints are",
              "*friendlier* (1e-5 .. 7e-5), not harsher, than the seeded initialiser.  tests/golden/g9_featurenet_noise.npz stores the",
              "per-output envelope for the G5 input; the FeatureNet parity tests bound |HIP - float64 reference| by 3x that envelope."]
    path = os.path.join(ROOT, "profiles", "r02_trained_blend_regime.md")
    open(path, "a").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
