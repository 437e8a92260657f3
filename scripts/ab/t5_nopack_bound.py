"""Upper bound of what batching the per-layer weight packing could buy the training step: the pack helpers return cached results
(stale weights: timing only)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T, train_ops, train2d_ops
dev = torch.device("cuda:0")
H, W, n_views, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 0).to(dev)
sample = bench.train_sample(H, W, n_views, refine, dev, seed=21)
opt = T.make_optimizer(model)
def timeit(n=8):
    for _ in range(4): T.train_step(model, opt, sample, temperature=0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): T.train_step(model, opt, sample, temperature=0.1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
base = timeit()
cache = {}
o3, o2 = train_ops.pack_conv3d, train2d_ops.pack_conv2d
def p3(w, mode, dgrad):
    k = ("3", w.data_ptr(), mode, dgrad)
    if k not in cache: cache[k] = o3(w, mode, dgrad)
    return cache[k]
def p2(wa, wb, fwd=True, dgrad=False):
    k = ("2", wa.data_ptr(), wb.data_ptr() if wb is not None else 0, fwd, dgrad)
    if k not in cache: cache[k] = o2(wa, wb, fwd, dgrad)
    return cache[k]
train_ops.pack_conv3d, train2d_ops.pack_conv2d = p3, p2
cached = timeit()
print(f"T5: {base:.2f} ms with per-layer packing, {cached:.2f} ms with cached packs ({len(cache)} packs per step)")
