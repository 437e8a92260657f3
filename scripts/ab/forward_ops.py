"""ATen ops issued by one cascade forward (torch profiler): which of them launch kernels, how often."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1184, 1600, 5)
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=0).to(dev); pm = synth.make_cameras(N, H, W, refine=False, seed=0); dv = synth.make_depth_values()
with torch.no_grad():
    for _ in range(2): model(imgs, pm, dv, temperature=0.01)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model(imgs, pm, dv, temperature=0.01); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and (e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key)]
for k, c, t in sorted(rows, key=lambda r: -r[1])[:30]:
    print(f"{c:5d} x {k:45s} {t:9.1f} us")
