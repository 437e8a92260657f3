"""Which lines of the product forward call torch ops (the glue around the HIP kernels)?  A TorchFunctionMode logs every torch
function called during ONE cascade forward together with the innermost cds_mvsnet_amd frame that called it."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.overrides import TorchFunctionMode
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
H, W, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 640, 5)
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).eval()
imgs = synth.make_images(N, H, W, seed=3).to(dev); pm = synth.make_cameras(N, H, W, refine=False, seed=3); dv = synth.make_depth_values()
cnt = collections.Counter()
SKIP = {"__get__", "size", "dim", "is_contiguous", "data_ptr", "view", "reshape", "__getitem__", "shape", "device", "dtype", "stride", "numel",
        "is_cuda", "unsqueeze", "squeeze", "permute", "transpose", "expand", "narrow", "select", "detach", "is_floating_point", "element_size"}
class Log(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            fr = [f for f in traceback.extract_stack() if "cds_mvsnet_amd/" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with torch.no_grad():
    for _ in range(2): model(imgs, pm, dv, temperature=0.01)
    with Log():
        model(imgs, pm, dv, temperature=0.01)
for (k, s), n in cnt.most_common(400): print(f"{n:4d} {k:24s} {s}")
print("total logged torch calls", sum(cnt.values()))
