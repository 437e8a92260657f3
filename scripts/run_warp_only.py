"""Run only K1 / K3 at the M1 shape a few times (target of rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cds_mvsnet_amd import ops, synth, geometry
h, w, D, C, N = 512, 640, 192, 8, 5
if len(sys.argv) > 1 and sys.argv[1] == "small":
    h, w, D = 256, 320, 96
dev = torch.device("cuda:0")
feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
cams = synth.stage_cameras(N, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev)
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
vis = torch.rand(N - 1, h, w, device=dev)
mats = ops.geo(geometry.warp_matrices(cams[0]), "cuda", "mats")   # device data since round 6 (geometry block)
for i in range(3):
    ent = ops.warp_entropy(ref, src, mats, hyp)
    vol, vs = ops.warp_aggregate(ref, src, vis, mats, hyp)
torch.cuda.synchronize()
print("done", float(ent.mean()), float(vol.abs().mean()))
