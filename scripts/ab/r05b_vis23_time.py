"""Visibility CNN layers 2 + 3 + head: one fused launch (cds_vis23_cl_f32) against the two launches it replaces, at the stage shapes of
the 1600x1184 / 1920x1056 cascades and at M1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator().manual_seed(0)
w2, w3 = (ops.split_pack_dynconv([(torch.randn(16, 16, 3, 3, generator=g) / 12).to(dev)]) for _ in range(2))
b2, b3, hw, hb = (torch.randn(n, generator=g).to(dev) for n in (16, 16, 16, 1))
for name, (V, H, W) in {"M1": (4, 512, 640), "M3 stage1": (4, 296, 400), "M3 stage2": (4, 592, 800), "M3 stage3": (4, 1184, 1600),
                        "M4 stage3": (6, 1056, 1920)}.items():
    x = torch.rand(V, H, W, 16, generator=g).to(dev)
    two = lambda: ops.conv2d_k3_relu_cl(ops.conv2d_k3_relu_cl(x, w2, b2), w3, b3, head_w=hw, head_b=hb)
    one = lambda: ops.vis23_cl(x, w2, b2, w3, b3, hw, hb)
    print(f"{name:10s} V={V} {W}x{H}: two launches {t(two):7.1f} us   fused {t(one):7.1f} us   equal {torch.equal(one(), two())}", flush=True)
