"""Build-container-only parity of the CPU oracle against the REFERENCE on its own TRAINED checkpoints (SURVEY §8(c):
"trained-weight parity is checked in this container only").  Skipped wherever /root/reference is absent (the GPU box):
the checkpoints are not vendored, nothing here travels.

For each of the three shipped checkpoints (`pretrained/*/`, loaded by test.py:180-187's rules: 'module.' prefix stripped)
the reference's `CDSMVSNet.forward` and `oracle.cds_oracle.forward` run on the same synthetic 3-view scene at the
evaluation temperature T = 0.01, refine=True (the checkpoints' own arch args), and must agree to the tolerances of
SURVEY §8(c): stage depth mean-L1 <= 1e-3, confidence <= 1e-3, curvature maps <= 2e-5."""
import os
import sys
import warnings

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pretrained")),
                                reason="needs the reference tree and its checkpoints (build container only)")

CKPTS = ["dtu_only/checkpoint-epoch24.pth", "both_dtu_blended/cds_mvsnet.ckpt", "fine_tuning_on_blended/cds_mvsnet.ckpt"]


def _reference_model(path):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True                       # the reference tree is read-only
    warnings.filterwarnings("ignore")
    from models.model import CDSMVSNet as RefNet  # reference, imported read-only
    from cds_mvsnet_amd.infer import _placeholder_pickle
    ck = torch.load(os.path.join(REF, "pretrained", path), map_location="cpu", weights_only=False,
                    pickle_module=_placeholder_pickle)
    sd = {k.replace("module.", ""): v for k, v in ck["state_dict"].items()}
    m = RefNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75))
    m.load_state_dict(sd, strict=True)                   # 387 entries, 0 missing / 0 unexpected
    return m.eval(), sd


@pytest.mark.parametrize("path", CKPTS)
def test_oracle_equals_reference_on_trained_checkpoint(path):
    from cds_mvsnet_amd import CDSMVSNet, synth
    from cds_mvsnet_amd.infer import load_checkpoint
    from oracle import cds_oracle as O
    torch.set_num_threads(8)
    ref, sd = _reference_model(path)
    # the product's loader accepts the same file (placeholder unpickler) and fills every one of the 387 entries
    mine = CDSMVSNet(refine=True, depth_interals_ratio=(4.0, 1.5, 0.75))
    load_checkpoint(mine, os.path.join(REF, "pretrained", path), trust_pickle=True)
    assert all(torch.equal(v, sd[k]) for k, v in mine.state_dict().items())

    N, H, W = 3, 256, 320
    imgs = synth.make_images(N, H, W, seed=4)
    cams = synth.make_cameras(N, H, W, refine=True, seed=4)
    dv = synth.make_depth_values()
    with torch.no_grad():
        want = ref(imgs, cams, dv, temperature=0.01)
        got = O.forward(imgs, cams, dv, sd, refine=True, temperature=0.01, exact=True)
    for k in ("stage1", "stage2", "stage3"):
        assert (got[k]["depth"] - want[k]["depth"]).abs().mean() < 1e-3, (path, k)
        assert (got[k]["photometric_confidence"] - want[k]["photometric_confidence"]).abs().mean() < 1e-3, (path, k)
        assert (got[k]["norm_curv"] - want[k]["norm_curv"]).abs().max() < 2e-5, (path, k)
    assert (got["refined_depth"] - want["refined_depth"]).abs().mean() < 1e-3, path
