"""Losses of 6 training steps with / without the side-stream weight gradients, twice each (run-to-run spread = chaos baseline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
from test_train_harness import _train_sample
dev = torch.device("cuda:0")
sample = _train_sample(dev)
for side in (False, False, True, True):
    T.SIDE_STREAM_WGRAD = side
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
    opt = T.make_optimizer(model, lr=1e-3)
    red = T.GradAllReducer(model.parameters())
    ls = [T.train_step(model, opt, sample, temperature=0.1, reducer=red)[0] for _ in range(6)]
    print("side" if side else "main", " ".join(f"{l:.5f}" for l in ls))
