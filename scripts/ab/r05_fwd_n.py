"""Does the per-forward time of the 1600x1184 cascade depend on how many forwards are timed back to back (bench.py times 3)?"""
import os, sys, time, gc
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
    for n in (1, 2, 3, 5, 10, 3, 1):
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); gc.collect(); gc.disable(); t0 = time.perf_counter()
            for _ in range(n): out = model(imgs, cams, dv, temperature=0.01)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3); gc.enable()
        print(f"n={n:2d}: min {min(ts):.2f} ms  all {[round(t, 2) for t in ts]}")
    # host time to enqueue one forward
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = model(imgs, cams, dv, temperature=0.01); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.2f} ms, then {1e3 * (t2 - t1):.2f} ms until the GPU is done")
