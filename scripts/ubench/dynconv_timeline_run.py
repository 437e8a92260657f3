"""Phase durations (cycles) of the DynamicConv kernel's workgroups, fused mode, one layer at the 1600x1184 cascade shapes.
usage: CDS_MVSNET_LIB=.../libcdsmvs_hip.probe_dyn.so python scripts/ubench/dynconv_timeline_run.py out3"""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops, _lib
name = sys.argv[1]
shapes = {"conv01": (8, 8, (3, 5, 7), 1184, 1600), "conv10": (16, 16, (3, 5), 592, 800), "conv20": (32, 32, (1, 3), 296, 400),
          "out2": (16, 16, (1, 3), 592, 800), "out3": (8, 8, (1, 3), 1184, 1600)}
cin, cout, ks, H, W = shapes[name]
N, dev = 8, torch.device("cuda")
co3 = cout + 3
x = torch.randn(N, cin, H, W, device=dev)
aff = torch.stack((torch.ones(N, cin), torch.zeros(N, cin), torch.full((N, cin), 0.1)), -1).to(dev).contiguous()
ws = [torch.randn(co3, cin, k, k, device=dev) / (cin * k * k) ** 0.5 for k in ks]
wsp = ops.split_pack_dynconv(ws)
out = torch.empty(len(ks), N, co3, H, W, device=dev)
for _ in range(3): ops.dynconv_branches_sbf(x, wsp, None, co3, ks, out=out, in_affine=aff)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); ops.dynconv_branches_sbf(x, wsp, None, co3, ks, out=out, in_affine=aff); b.record(); torch.cuda.synchronize()
lib = _lib.load()
n = 8 * 4096
buf = (ctypes.c_longlong * n)()
lib.cds_dyn_probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cds_dyn_probe_dump(buf, n) == 0
t = np.array(buf[:], dtype=np.int64).reshape(8, 4096)
sel = slice(512, 4096)     # skip the first wave of workgroups
d = lambda i, j: np.median(t[j, sel] - t[i, sel])
print(f"{name}: kernel {a.elapsed_time(b)*1e3:.0f} us; median cycles per workgroup (last round's stamps): start->first loads landed {d(0,1):.0f}, "
      f"split+LDS stores {d(1,2):.0f}, barrier {d(2,3):.0f}, K-loop {d(3,4):.0f}, whole {np.median(t[4, sel]-t[0, sel]):.0f} (+ epilogue)")
