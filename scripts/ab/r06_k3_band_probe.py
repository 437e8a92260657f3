"""Probe (round 6): does the 256 MB memory-side cache pay for K3's second pass at N = 7?  Six source views run as two launches of three;
over the whole image the second launch re-reads the first one's partial sums (0.8-1.0 GB per stage) from HBM.  Through the row-window
entry point the same two launches can run band by band (a band's volume ~100 MB), so the partial sums of a band are read back right
after they were written.  Timing only: the band volumes are separate tensors here.
Usage: r06_k3_band_probe.py h w D C band_rows [lo hi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import ops, synth, geometry
a = sys.argv[1:]
h, w, D, C, band = (int(v) for v in a[:5])
rng = dict(lo=float(a[5]), hi=float(a[6])) if len(a) >= 7 else {}
N = 7
dev = torch.device("cuda:0")
feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
cams = synth.stage_cameras(N, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1, **rng)[0].to(dev)
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
vis = torch.rand(N - 1, h, w, device=dev)
mats = ops.geo(geometry.warp_matrices(cams[0]), "cuda", "mats")
vol = torch.empty((D, h, w, C), device=dev); vs = torch.empty(h, w, device=dev)
bands = []
for y0 in range(0, h, band):
    y1 = min(h, y0 + band)
    bands.append((y0, ref[:, :, y0:y1].contiguous(), hyp[:, y0:y1].contiguous(), vis[:, y0:y1].contiguous(),
                  torch.empty((D, y1 - y0, w, C), device=dev), torch.empty(y1 - y0, w, device=dev)))
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def full():
    ops.warp_aggregate(ref, src, vis, mats, hyp, volume=vol, vis_sum=vs, channels_last=True, exact=True)
def banded():
    for y0, r, hy, vi, bv, bs in bands:
        ops.warp_aggregate(r, src, vi, mats, hy, volume=bv, vis_sum=bs, channels_last=True, exact=True, window=(h, y0))
tf, tb = timeit(full), timeit(banded)
mb = D * band * w * C * 4 / 1e6
print(f"N=7 {w}x{h} D={D} C={C}: whole image {tf:.3f} ms; {len(bands)} bands of {band} rows ({mb:.0f} MB each) {tb:.3f} ms")
