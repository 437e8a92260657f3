"""ATen ops (and the launches they cause) of ONE cascade forward: the glue around the HIP kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
from torch.profiler import profile, ProfilerActivity
H, W, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 640, 5)
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).eval()
imgs = synth.make_images(N, H, W, seed=3).to(dev); pm = synth.make_cameras(N, H, W, refine=False, seed=3); dv = synth.make_depth_values()
with torch.no_grad():
    for _ in range(3):
        model(imgs, pm, dv, temperature=0.01)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model(imgs, pm, dv, temperature=0.01)
        torch.cuda.synchronize()
ka = prof.key_averages()
print("device-time ops (name, calls, device us):")
for e in sorted(ka, key=lambda e: -e.count):
    if e.self_device_time_total > 0 and (e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key):
        print(f"  {e.key[:60]:60s} {e.count:5d} {e.self_device_time_total:9.0f}")
