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
    T.train_step(model, opt, sample, temperature=0.1, reducer=reducer)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(model, opt, sample, temperature=0.1, reducer=reducer)
    torch.cuda.synchronize()
import collections
ks = prof.key_averages(group_by_stack_n=12)
agg = collections.Counter(); dev_t = collections.Counter()
for e in ks:
    if e.key not in ("aten::copy_", "aten::clone", "aten::div", "aten::fill_", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::zeros", "aten::flip", "aten::item", "aten::neg", "aten::where", "aten::sub", "aten::stack", "aten::index", "aten::mean"):
        continue
    site = next((fr for fr in e.stack if "/cds_mvsnet_amd/" in fr or "bench.py" in fr), (e.stack[0] if e.stack else "?"))
    site = site.replace("/root/repo/", "").split("/scratch")[-1][-90:]
    agg[(e.key, site)] += e.count; dev_t[(e.key, site)] += e.self_device_time_total
print("small ATen ops by call site (op, site, calls, device us):")
for (k, site), c in agg.most_common(70):
    print(f"  {k:14s} {c:4d} {dev_t[(k, site)]:8.0f}  {site}")
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("ops by call count (name, calls, self cpu us, self device us):")
for e in rows[:45]:
    print(f"  {e.key[:70]:70s} {e.count:5d} {e.self_cpu_time_total:9.0f} {e.self_device_time_total:9.0f}")
print(ka.table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
