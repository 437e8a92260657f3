"""K3 once + calibration kernels with known byte counts (target of the FETCH_SIZE / WRITE_SIZE PMC passes)."""
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
mats = ops.geo(geometry.warp_matrices(cams[0]), "cuda", "mats")   # device data since round 6 (geometry block)
vol = torch.empty(D, h, w, C, device=dev); vs = torch.empty(h, w, device=dev)   # channels-last: what the model runs
for _ in range(3):
    ops.warp_aggregate(ref, src, vis, mats, hyp, volume=vol, vis_sum=vs, channels_last=True)      # B_alg = 2 354 053 120 B
    ops.warp_entropy(ref, src, mats, hyp)
    ops.volume_normalize_(vol.view(C, D, h, w), vs + 1.0)   # calibration (planar view of the same bytes): reads 2 013 265 920 B (+1.3 MB), writes 2 013 265 920 B
torch.cuda.synchronize()
print("ok")
