"""Does running two independent depth-map pipelines on two HIP streams raise throughput at M1?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
_, cams, hyp, dfe = bench.make_workload("M1", 0, dev)
cams_d, hyp_d = cams.to(dev), hyp.to(dev)
D = bench.WORKLOADS["M1"][2]
def step():
    return model.stage_net(dfe, cams_d, depth_values=hyp_d, num_depth=D, cost_regularization=model.cost_regularization[2], stage_idx=2)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / n
    print(f"one stream: {t1*1e3:.2f} ms per depth map")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rounds in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n // 2):
            with torch.cuda.stream(s1): step()
            with torch.cuda.stream(s2): step()
        torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / n
    print(f"two streams: {t2*1e3:.2f} ms per depth map ({t1/t2:.3f}x)")
