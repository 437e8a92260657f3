"""Time K1 / K3 alone at the M1 shape with HIP events (A/B of kernel variants through env knobs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops, synth, geometry
h, w, D, C, N = 512, 640, 192, 8, 5
dev = torch.device("cuda:0")
feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
cams = synth.stage_cameras(N, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev)
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
vis = torch.rand(N - 1, h, w, device=dev)
mats = geometry.warp_matrices(cams[0])
vol = torch.empty(C, D, h, w, device=dev); vs = torch.empty(h, w, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t1 = timeit(lambda: ops.warp_entropy(ref, src, mats, hyp))
t3 = timeit(lambda: ops.warp_aggregate(ref, src, vis, mats, hyp, volume=vol, vis_sum=vs))
print(f"{os.environ.get('TAG','')} K1 {t1:.3f} ms  K3 {t3:.3f} ms  (K3 roofline frac {2354053120/(t3*1e-3)/8e12:.3f})")
