"""K3 backward (warp_aggregate_bwd) at the three stage shapes of the config-5 training step (768x576 / 2, N=5) + the gt-depth (D = 1) calls."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import ops, synth, geometry
dev = torch.device("cuda:0")
H, W, N = 288, 384, 5
g = torch.Generator().manual_seed(5)
for s, (sc, D, C, ratio) in enumerate(zip((4, 2, 1), (48, 32, 8), (32, 16, 8), (4.0, 2.0, 1.0))):
    h, w = H // sc, W // sc
    feats = synth.make_pair_features(N - 1, C, h, w, seed=11 + s)
    cams = synth.stage_cameras(N, h, w, seed=s)
    base = 600.0 + 120.0 * torch.nn.functional.interpolate(torch.rand(1, 1, 6, 8, generator=g), (h, w), mode="bicubic", align_corners=False)[0, 0]
    if s == 0:
        hyp = torch.linspace(425.0, 902.5, D).view(D, 1, 1).expand(D, h, w).contiguous()
    else:
        hyp = (base.unsqueeze(0) + (torch.arange(D, dtype=torch.float32).view(D, 1, 1) - (D - 1) // 2) * (ratio * 2.5)).contiguous()
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
    vis = (torch.rand(N - 1, h, w, generator=g) * 0.9 + 0.05).to(dev)
    mats = geometry.warp_matrices(cams[0])
    for name, hy in (("main", hyp.to(dev)), ("gt  ", base.unsqueeze(0).contiguous().to(dev))):
        Dh = hy.shape[0]
        gv = torch.randn(C, Dh, h, w, device=dev)
        ts = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(); out = ops.warp_aggregate_bwd(ref, src, vis, mats, hy, gv); b.record(); b.synchronize()
            if i >= 2: ts.append(a.elapsed_time(b) * 1e3)
        print(f"stage {s + 1} {name} {w}x{h} D={Dh} C={C}: {statistics.median(ts):8.1f} us (incl. three zero fills)  checksum {sum(float(o.double().sum()) for o in out):.6e}")
