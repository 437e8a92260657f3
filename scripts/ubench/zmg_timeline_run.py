"""Run one layer through the timeline probe library and print per-stage phase durations (cycles) of workgroup 300.
usage: CDS_MVSNET_LIB=.../libcdsmvs_hip.probe_timeline.so python scripts/ubench/zmg_timeline_run.py conv0"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops, _lib
import numpy as np
name = sys.argv[1]
shapes = {"conv0": (8, 8, 1, 192, 512, 640), "conv1": (8, 16, 2, 192, 512, 640), "conv2": (16, 16, 1, 96, 256, 320), "s2conv0": (16, 8, 1, 32, 592, 800)}
cin, cout, stride, D, H, W = shapes[name]
dev = torch.device("cuda")
x = torch.randn(D, H, W, cin, device=dev)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
b = torch.randn(cout, device=dev)
pair = cout == 8 and stride == 1
ws = ops.split_pack_conv3d_pair(w) if pair else ops.split_pack_conv3d(w)
code = ops.SBF_PAIR if pair else stride
for _ in range(3): ops.conv3d_sbf(x, ws, b, cout, stride=code)
torch.cuda.synchronize()
lib = _lib.load()
n = 3 * 8 * 256
buf = (ctypes.c_longlong * n)()
lib.cds_zmg_probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cds_zmg_probe_dump(buf, n) == 0
a = np.array(buf[:], dtype=np.int64).reshape(3, 8, 256)
ns = int((a[0, 0] > 0).sum())
print(f"{name}: {ns} stages stamped")
c = a[:, :, :ns]
t0 = c[0, 0, 0]
print("st | early consumer: top->kstart kloop epi | late consumer: top->kstart kloop epi | producer: deposit issue wait | stage period")
for st in range(1, min(ns - 1, 40)):
    e, l, p = c[0], c[1], c[2]
    per = e[0, st + 1] - e[0, st]
    print(f"{st:3d} | {e[1,st]-e[0,st]:6d} {e[2,st]-e[1,st]:6d} {e[3,st]-e[2,st]:6d} (barrier wait {e[0,st+1]-e[3,st]:6d}) | "
          f"{l[1,st]-l[0,st]:6d} {l[2,st]-l[1,st]:6d} {l[3,st]-l[2,st]:6d} (barrier wait {l[0,st+1]-l[3,st]:6d}) | "
          f"{p[1,st]-p[0,st]:6d} {p[2,st]-p[1,st]:6d} (barrier wait {p[0,st+1]-p[2,st]:6d}) | {per:6d}")
