"""Time K1 / K3 alone with HIP events (A/B of kernel variants through env knobs).
Usage: time_warp.py [h w D C [lo hi]]   (default: the M1 shape; lo/hi = hypothesis range, narrow for cascade stages 2/3)
Env: NVIEWS (default 5), EXACT=0|1 (sample-position mode), CL=1 (channels-last volume)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops, synth, geometry
a = sys.argv[1:]
h, w, D, C = (int(v) for v in a[:4]) if len(a) >= 4 else (512, 640, 192, 8)
N = int(os.environ.get('NVIEWS', '5'))
EXACT = os.environ.get('EXACT', '0') == '1'
CL = os.environ.get('CL', '0') == '1'
rng = dict(lo=float(a[4]), hi=float(a[5])) if len(a) >= 6 else {}
dev = torch.device("cuda:0")
feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
cams = synth.stage_cameras(N, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1, **rng)[0].to(dev)
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
vis = torch.rand(N - 1, h, w, device=dev)
mats = ops.geo(geometry.warp_matrices(cams[0]), "cuda", "mats")   # device data since round 6 (geometry block)
vol = torch.empty((D, h, w, C) if CL else (C, D, h, w), device=dev); vs = torch.empty(h, w, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t1 = timeit(lambda: ops.warp_entropy(ref, src, mats, hyp, exact=EXACT))
t3 = timeit(lambda: ops.warp_aggregate(ref, src, vis, mats, hyp, volume=vol, vis_sum=vs, channels_last=CL, exact=EXACT))
b_alg = 4.0 * h * w * (C * D + D + 2 * (N - 1) * C + (N - 1))   # volume + hypotheses + features + weights
print(f"{os.environ.get('TAG','')} N={N} exact={int(EXACT)} cl={int(CL)} {w}x{h} D={D} C={C}: K1 {t1:.3f} ms  K3 {t3:.3f} ms  (K3 {b_alg/1e6:.0f} MB, roofline frac {b_alg/(t3*1e-3)/8e12:.3f})")
