"""Does a hipGraph replay of CostRegNet + soft-argmin beat the eager launch sequence?  (M1 and a cascade-stage-1 shape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, ops
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False), 0).eval().to(dev)
for (C, D, h, w, s) in [(8, 192, 512, 640, 2), (32, 48, 128, 160, 0), (16, 32, 256, 320, 1)]:
    cr = model.cost_regularization[s]
    vol = torch.randn(C, D, h, w, device=dev)
    hyp = torch.rand(D, h, w, device=dev) * 400 + 400
    def run():
        p = cr(vol)
        return ops.softargmin_conf(p, hyp)
    with torch.no_grad():
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            with torch.cuda.graph(g, stream=st):
                out = run()
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    print(f"C={C} D={D} {w}x{h}: eager {te*1e3:.3f} ms, graph replay {tg*1e3:.3f} ms")
