"""Deviation of one training step (loss, per-parameter gradient norms) from the reference's CPU step (golden G7), for the two
FeatureNet call groupings of training.forward_train (CDS_TRAIN_BATCH_FEATURES=1 | 0).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cds_mvsnet_amd import CDSMVSNet, final_loss, seeded_init_, training

g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "g7_training_step.npz")).items() if v.dtype.kind != "U"}
dev = torch.device("cuda")
for mode in (True, False):
    training.BATCH_FEATURES = mode
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).train()
    cams = {k[4:]: v.to(dev) for k, v in g.items() if k.startswith("cam_")}
    gt = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("gt_")}
    mask = {k[5:]: v.to(dev) for k, v in g.items() if k.startswith("mask_")}
    dv = g["depth_values"].to(dev)
    out = model(g["imgs"].to(dev), cams, dv, gt_depths=gt, temperature=0.1)
    loss, _ = final_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0], depth_interval=dv[:, 1] - dv[:, 0])
    loss.backward()
    names = [str(n) for n in np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "g7_training_step.npz"))["param_names"]]
    want = dict(zip(names, g["grad_norms"].tolist()))
    devs = [abs(float(p.grad.norm()) - want[n]) / max(want[n], 1e-6) for n, p in model.named_parameters() if n in want]
    print(f"batched={mode}: loss {loss.item():.5f} (golden {float(g['loss']):.5f}, rel {abs(loss.item() - float(g['loss'])) / float(g['loss']):.2e}); "
          f"grad-norm rel deviation: median {np.median(devs) if devs else float('nan'):.2e}, max {max(devs) if devs else float('nan'):.2e} over {len(devs)} parameters; "
          f"depth mean abs dev {(out['stage3']['depth'].detach().cpu() - g['stage3_depth']).abs().mean().item():.3e}")

