"""cProfile of the host side of the cascade forward (the 640x512 forward is launch-bound: ~7 ms of kernels, ~10 ms wall)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = 512, 640, 5
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=0).to(dev)
cams = synth.make_cameras(N, H, W, refine=False, seed=0)     # host tensors, as infer.py / bench.py pass them
dv = synth.make_depth_values()
with torch.no_grad():
    for _ in range(3): model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5): model(imgs, cams, dv, temperature=0.01)
    t_host = (time.perf_counter() - t0) / 5          # host time to ENQUEUE a forward (no sync inside)
    torch.cuda.synchronize()
    print(f"host enqueue time per forward: {t_host * 1e3:.2f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
