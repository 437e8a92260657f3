"""Does a layer run faster when its input was just written (Infinity-Cache / MALL hot) than when it comes from HBM?
conv1 (8 -> 16, stride 2) and conv0 (8 -> 8) on z-chunks of the M1 volume; 'hot' = producer-like write immediately before,
'cold' = a 1.5 GB copy in between."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
H, W = 512, 640
trash_a = torch.empty(384 * 1024 * 1024, device=dev); trash_b = torch.empty_like(trash_a)
def ev(): return torch.cuda.Event(enable_timing=True)
for D in (8, 16, 24, 48):
    x = torch.randn(D, H, W, 8, device=dev); src = torch.randn(D, H, W, 8, device=dev)
    w0 = torch.randn(8, 8, 3, 3, 3, device=dev) / 216 ** 0.5; ws0 = ops.split_pack_conv3d_pair(w0)
    w1 = torch.randn(16, 8, 3, 3, 3, device=dev) / 216 ** 0.5; ws1 = ops.split_pack_conv3d(w1)
    b0 = torch.zeros(8, device=dev); b1 = torch.zeros(16, device=dev)
    res = {}
    for name, fn in (("conv0", lambda: ops.conv3d_sbf(x, ws0, b0, 8, stride=ops.SBF_PAIR)), ("conv1", lambda: ops.conv3d_sbf(x, ws1, b1, 16, stride=2))):
        for mode in ("hot", "cold"):
            ts = []
            for it in range(6):
                if mode == "hot":
                    trash_b.copy_(trash_a); x.copy_(src)        # x written last: hot
                else:
                    x.copy_(src); trash_b.copy_(trash_a)        # 3 GB of other traffic after x was written
                a, b = ev(), ev(); a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[(name, mode)] = sorted(ts)[len(ts) // 2]
    mb = x.numel() * 4 / 1e6
    print(f"D={D} ({mb:.0f} MB input): conv0 hot {res[('conv0','hot')]:.0f} us cold {res[('conv0','cold')]:.0f} us | conv1 hot {res[('conv1','hot')]:.0f} us cold {res[('conv1','cold')]:.0f} us")
