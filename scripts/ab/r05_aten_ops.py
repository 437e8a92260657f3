"""Which ATen launches (not our HIP kernels) are inside one cascade forward, and where in model.py do they come from."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1184, 1600, 5)))
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=0).to(dev)
cams = synth.make_cameras(N, H, W, refine=False, seed=0)
dv = synth.make_depth_values()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        t = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, (tuple, list)) and out and isinstance(out[0], torch.Tensor) else None)
        on_gpu = (t is not None and t.is_cuda) or any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if on_gpu and not any(k in name for k in ("view", "empty", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "select", "slice", "expand", "permute", "transpose", "reshape", "_unsafe_view", "unbind", "t.default", "is_pinned", "_local_scalar", "record_stream", "narrow", "split", "contiguous")):
            fr = [f for f in traceback.extract_stack() if "cds_mvsnet_amd" in f.filename][-2:]
            agg[(name, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return out
with torch.no_grad():
    for _ in range(2): model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize()
    with Log():
        model(imgs, cams, dv, temperature=0.01)
print("aten ops touching cuda tensors in one forward:", sum(agg.values()))
for (name, fr), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"n={n:3d} {name:28s} {fr}")
