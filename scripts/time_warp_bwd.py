"""K3 backward at the three cascade-stage shapes of the config-5 training step (768x576, refine -> 384x288 base).
CDS_K3BWD_DIRECT=1 selects the direct-scatter kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cds_mvsnet_amd import ops, synth, geometry
dev = torch.device("cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, h, w, D, C in (("stage1", 72, 96, 48, 32), ("stage2", 144, 192, 32, 16), ("stage3", 288, 384, 8, 8), ("gt3", 288, 384, 1, 8)):
    V = 4
    feats = synth.make_pair_features(V, C, h, w, seed=3)
    cams = synth.stage_cameras(V + 1, h, w, seed=1)
    hyp = synth.make_hypotheses(D, h, w, seed=2)[0].to(dev).contiguous()
    mats = geometry.warp_matrices(cams[0])
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev)
    src = torch.stack([f["src"][0][0] for f in feats]).permute(0, 2, 3, 1).contiguous().to(dev)
    vis = torch.rand(V, h, w, device=dev)
    g = torch.randn(C, D, h, w, device=dev)
    us = t(lambda: ops.warp_aggregate_bwd(ref, src, vis, mats, hyp, g))
    print(f"{name}: {h}x{w} D={D} C={C} V={V}: {us:8.1f} us")
