"""Run the fused conv11 + prob kernel through the timeline probe library and print per-half-step phase durations (cycles) of
workgroup 100.  usage: CDS_MVSNET_LIB=.../libcdsmvs_hip.dpz_timeline.so python scripts/ubench/dpz_timeline_run.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops, _lib
import numpy as np
D, H, W = 96, 256, 320
dev = "cuda"
x = torch.randn(D, H, W, 16, device=dev)
skip = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev)
ws = ops.split_pack_deconv_prob(torch.randn(16, 8, 3, 3, 3, device=dev) * 0.1)
b = torch.randn(8, device=dev) * 0.1
tab = ops.pack_prob_table(torch.randn(1, 8, 3, 3, 3, device=dev) * 0.1)
for _ in range(3):
    ops.deconv_prob_zm(x, ws, b, skip, tab)
torch.cuda.synchronize()
lib = _lib.load()
n = 3 * 8 * 256
buf = (ctypes.c_longlong * n)()
lib.cds_dpz_probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cds_dpz_probe_dump(buf, n) == 0
a = np.array(buf[:], dtype=np.int64).reshape(3, 8, 256)
ns = int((a[2, 0] > 0).sum())
print(f"{ns} half-steps stamped (s_memtime ticks: 100 MHz constant clock -> x ~19 for shader cycles)" )
print(" t | consumer 0: operands mfma epilogue (barrier wait) | consumer 7: operands mfma epilogue (barrier wait) | prob: process deposit store (barrier wait) | period")
for t in range(2, min(ns - 1, 44)):
    e, l, p = a[0], a[1], a[2]
    per = p[0, t + 1] - p[0, t]
    # even half-steps: the consumers' third phase = epilogue + input staging (stamp 4 separates them)
    stg = f" [stg {e[3,t]-e[4,t]:5d} {l[3,t]-l[4,t]:5d}]" if (e[4, t] > e[2, t] and e[4, t] <= e[3, t]) else ""
    print(f"{t:3d} | {e[1,t]-e[0,t]:6d} {e[2,t]-e[1,t]:6d} {e[3,t]-e[2,t]:6d} ({e[0,t+1]-e[3,t]:6d}) | "
          f"{l[1,t]-l[0,t]:6d} {l[2,t]-l[1,t]:6d} {l[3,t]-l[2,t]:6d} ({l[0,t+1]-l[3,t]:6d}) | "
          f"{p[1,t]-p[0,t]:6d} {p[2,t]-p[1,t]:6d} {p[3,t]-p[2,t]:6d} ({p[0,t+1]-p[3,t]:6d}) | {per:6d}")
