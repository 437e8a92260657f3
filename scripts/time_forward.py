"""Time the full CDSMVSNet forward (M2: 640x512 images, N=5, cascade 48/32/8) on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 640, 5)))
refine = len(sys.argv) > 4 and sys.argv[4] == "refine"
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=refine, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=0).to(dev)
cams = synth.make_cameras(N, H, W, refine=refine, seed=0)     # host tensors: no readback in the forward
dv = synth.make_depth_values()
with torch.no_grad():
    for _ in range(2): out = model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): out = model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"full forward {W}x{H} N={N} refine={refine}: {dt*1e3:.2f} ms/depth-map = {1/dt:.2f} depth-maps/s; depth mean {out['depth'].mean().item():.3f}")
