"""Four single forwards of the 1600x1184 cascade, each after a device synchronisation (target of a kernel trace: where does the GPU idle?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = 1184, 1600, 5
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=0).to(dev)
cams = synth.make_cameras(N, H, W, refine=False, seed=0)
dv = synth.make_depth_values()
with torch.no_grad():
    for _ in range(3): model(imgs, cams, dv, temperature=0.01)
    for i in range(4):
        torch.cuda.synchronize(); time.sleep(0.02); t0 = time.perf_counter()
        model(imgs, cams, dv, temperature=0.01)
        torch.cuda.synchronize(); print(f"single forward {1e3 * (time.perf_counter() - t0):.2f} ms")
