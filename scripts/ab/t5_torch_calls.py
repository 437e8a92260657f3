"""Which lines of the training step call torch functions (forward + loss; the backward runs on the autograd thread and is listed by
op name from the profiler)?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.overrides import TorchFunctionMode
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
dev = torch.device("cuda:0")
H, W, n_views, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 7).to(dev)
sample = bench.train_sample(H, W, n_views, refine, dev, seed=21)
opt = T.make_optimizer(model)
for _ in range(3): T.train_step(model, opt, sample, temperature=0.1)
cnt = collections.Counter()
SKIP = {"__get__", "size", "dim", "is_contiguous", "data_ptr", "view", "reshape", "__getitem__", "shape", "device", "dtype", "stride", "numel",
        "is_cuda", "unsqueeze", "squeeze", "permute", "transpose", "expand", "narrow", "select", "detach", "is_floating_point", "element_size",
        "__float__", "requires_grad", "grad", "__set__", "view_as", "item"}
class Log(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            fr = [f for f in traceback.extract_stack() if "cds_mvsnet_amd/" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    T.train_step(model, opt, sample, temperature=0.1)
gpu = [(k, n) for k, n in cnt.most_common() if "geometry.py" not in k[1]]
for (k, s), n in gpu[:90]: print(f"{n:4d} {k:26s} {s}")
print("total logged torch calls (excl. geometry)", sum(n for _, n in gpu))
