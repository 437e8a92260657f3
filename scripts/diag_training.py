import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cds_mvsnet_amd import CDSMVSNet, final_loss, seeded_init_
z = np.load("tests/golden/g7_training_step.npz")
g = {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" and z[k].ndim > 0 else z[k]) for k in z.files}
dev = torch.device("cuda:0")
for it in range(4):
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).train()
    cams = {k[4:]: v.to(dev) for k, v in g.items() if k.startswith("cam_")}
    gt = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("gt_")}
    mask = {k[5:]: v.to(dev) for k, v in g.items() if k.startswith("mask_")}
    dv = g["depth_values"].to(dev)
    out = model(g["imgs"].to(dev), cams, dv, gt_depths=gt, temperature=0.1)
    loss, dl = final_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0], depth_interval=dv[:, 1] - dv[:, 0])
    loss.backward()
    want = dict(zip([str(n) for n in g["param_names"]], [float(x) for x in g["grad_norms"]]))
    got = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    rel = {n: abs(got[n] - want[n]) / max(want[n], 1e-6) for n in want}
    worst = sorted(rel, key=rel.get)[-3:]
    print(it, "loss", loss.item(), float(g["loss"]), "dl", dl.item(), float(g["depth_loss"]), "median", sorted(rel.values())[len(rel)//2],
          "worst", [(w, round(rel[w], 4), want[w]) for w in worst])
