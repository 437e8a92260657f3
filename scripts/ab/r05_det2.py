import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cds_mvsnet_amd.model as cm
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, geometry
dev = torch.device("cuda")
H, W, N = 1184, 1600, 5
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=4).to(dev)
cams = synth.make_cameras(N, H, W, refine=False, seed=4)
dv = synth.make_depth_values()
with torch.no_grad():
    for layout in (True, False):
        cm.USE_FEAT_CL = layout
        cm.OVERLAP_STAGE1 = False
        ref = model(imgs, cams, dv, temperature=0.01)
        cm.OVERLAP_STAGE1 = True
        for rep in range(3):
            a = model(imgs, cams, dv, temperature=0.01)
            msg = []
            for k in ("stage1", "stage2", "stage3"):
                d = (a[k]["depth"] - ref[k]["depth"]).abs()
                nc = (a[k]["norm_curv"] - ref[k]["norm_curv"]).abs()
                msg.append(f"{k}: depth differing {int((d > 0).sum())} mean {d.mean().item():.2e} max {d.max().item():.2e} | norm_curv differing {int((nc > 0).sum())} max {nc.max().item():.2e}")
            print("channels-last" if layout else "planar", "overlap run", rep, "vs single-stream:", " ; ".join(msg))
