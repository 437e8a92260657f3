"""Which ATen ops / copies does one M1 step issue? (torch profiler, one step after warm-up)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 0).eval().to(dev)
h, w, D, C, n = bench.WORKLOADS["M1"]
_, cams, hyp, dfe = bench.make_workload("M1", 0, dev)
hyp_d = hyp.to(dev)
step = lambda: model.stage_net(dfe, cams, depth_values=hyp_d, num_depth=D, cost_regularization=model.cost_regularization[2], stage_idx=2)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
