"""cProfile of the CPU side of the config-5 training step (the step is CPU-bound: 24 of 28 ms are enqueue time)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
dev = torch.device("cuda:0")
H, W, n_views, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 7).to(dev)
sample = bench.train_sample(H, W, n_views, refine, dev, seed=21)
opt = T.make_optimizer(model)
red = T.GradAllReducer(model.parameters(), module=model)
for _ in range(3):
    T.train_step(model, opt, sample, temperature=0.1, reducer=red)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    T.train_step(model, opt, sample, temperature=0.1, reducer=red)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
