"""ATen launches (not our HIP kernels) inside one config-5 training step: op, output shape, source line (forward) or autograd node."""
import os, sys, collections, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import torch.nn.functional as F
from torch.utils._python_dispatch import TorchDispatchMode
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, train as T
H, W, N, B = 576, 768, 5, 1
dev = torch.device("cuda:0")
imgs = synth.make_images(N, H, W, seed=1).to(dev)
cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=True, seed=1).items()}
dv = synth.make_depth_values().to(dev)
g = torch.Generator().manual_seed(9)
base = 600.0 + 120.0 * F.interpolate(torch.rand(B, 1, 4, 6, generator=g), (H, W), mode="bicubic", align_corners=False)[:, 0]
gt, mask = {}, {}
for s, sc in (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)):
    gt[s] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous().to(dev)
    mask[s] = torch.ones(B, H // sc, W // sc, device=dev)
sample = {"imgs": imgs, "proj_matrices": cams, "depth_values": dv, "depth": gt, "mask": mask}
model = seeded_init_(CDSMVSNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
opt = T.make_optimizer(model)
red = T.GradAllReducer(model.parameters())
for _ in range(2): T.train_step(model, opt, sample, 0.1, reducer=red)
torch.cuda.synchronize()
agg = collections.Counter()
SKIP = ("view", "empty", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "select", "slice", "expand", "permute", "transpose", "reshape",
        "unbind", "t.default", "is_pinned", "_local_scalar", "record_stream", "narrow", "split", "is_same_size", "sym_", "stride", "size")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        t = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, (tuple, list)) and out and isinstance(out[0], torch.Tensor) else None)
        on_gpu = (t is not None and t.is_cuda) or any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
        name = str(func)
        if on_gpu and not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if "cds_mvsnet_amd" in f.filename][-2:]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)) or "(autograd engine)"
            shp = tuple(t.shape) if t is not None else ()
            agg[(name, where, shp if len(where) > 20 or True else ())] += 1
        return out
with Log():
    T.train_step(model, opt, sample, 0.1, reducer=red)
torch.cuda.synchronize()
print("aten ops touching cuda tensors in one training step:", sum(agg.values()))
byline = collections.Counter()
for (name, where, shp), n in agg.items(): byline[(name, where)] += n
for (name, where), n in sorted(byline.items(), key=lambda kv: -kv[1]):
    shapes = sorted({str(s) for (nm, wh, s), _ in agg.items() if nm == name and wh == where})
    print(f"n={n:3d} {name:34s} {where:60s} {' '.join(shapes[:4])}{' ...' if len(shapes) > 4 else ''}")
