"""torch.profiler view of one T5 training step: which ATen ops (and which of our call sites) issue the small launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
H, W, n_views, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 7).to(dev)
model.train()
sample = bench.train_sample(H, W, n_views, refine, dev, seed=21)
opt = T.make_optimizer(model)
reducer = T.GradAllReducer(model.parameters(), module=model)
for _ in range(3):
    T.train_step(model, opt, sample, temperature=0.1, reducer=reducer, bf16=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(model, opt, sample, temperature=0.1, reducer=reducer, bf16=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=50, max_src_column_width=110))
