"""Eager vs hipGraph-replayed inference forward: steady-state ms per forward (back to back) and one forward after a synchronisation."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
from cds_mvsnet_amd.graphed import CapturedForward
dev = torch.device("cuda")
shapes = [(512, 640, 5), (512, 640, 3), (1184, 1600, 5), (1056, 1920, 7)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
for H, W, N in shapes:
    imgs = synth.make_images(N, H, W, seed=0).to(dev)
    pm, dv = synth.make_cameras(N, H, W, refine=False, seed=0), synth.make_depth_values()
    res = {}
    for name, fn in (("eager", lambda: model(imgs, pm, dv, temperature=0.01)), ("graph", None), ("graph_nocheck", None)):
        if name == "graph":
            r = CapturedForward(model); fn = lambda r=r: r(imgs, pm, dv, temperature=0.01)
        if name == "graph_nocheck":
            r = CapturedForward(model, check_weights=False); fn = lambda r=r: r(imgs, pm, dv, temperature=0.01)
        with torch.no_grad():
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize(); steady = (time.perf_counter() - t0) / 20 * 1e3
            single = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); single.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            for _ in range(20): fn()
            host = (time.perf_counter() - t0) / 20 * 1e3
            torch.cuda.synchronize()
        res[name] = (steady, sorted(single)[2], host)
    print(f"{W}x{H} N={N}: " + "; ".join(f"{k}: steady {v[0]:.3f} ms, single {v[1]:.3f} ms, host enqueue {v[2]:.3f} ms" for k, v in res.items()), flush=True)
