"""A/B of the exact vs relaxed sample-position arithmetic of the warp kernels: run once per library build,

    python tests/tools/relaxed_ab.py exact   > gpurun_out/ab_exact.json
    CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.relaxed.so python tests/tools/relaxed_ab.py relaxed > gpurun_out/ab_relaxed.json

For every warp golden (G1 a/b/c, captured from the reference) it reports the volume max-abs error, the entropy / visibility
errors and the stage depth mean-L1; for the full-forward goldens (G6) the per-stage depth mean-L1; plus K1 / K3 times at M1.
Uses the goldens only (the reference's own outputs), so it lives with the tests."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from cds_mvsnet_amd import CDSMVSNet, geometry, ops, seeded_init_, synth  # noqa: E402

dev = torch.device("cuda")
res = {"build": sys.argv[1] if len(sys.argv) > 1 else "?"}
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 7).eval().to(dev)
with torch.no_grad():
    for tag in ("a", "b", "c"):
        g = load_golden(f"g1_warp_aggregate_{tag}")
        stage = int(g["stage"])
        ref = g["ref_fea"].to(dev).contiguous()
        src = torch.stack([ops.chw_to_hwc(g["src_fea"][v].to(dev).contiguous()) for v in range(ref.shape[0])])
        ref_nc = g["ref_nc"][:, 0].to(dev).contiguous()
        mats = geometry.warp_matrices(g["cams"][0])
        hyp = g["hyp"][0].to(dev).contiguous()
        vol, _, ent, vis = model.stage_net.aggregate(ref, src, ref_nc, mats, hyp, stage)
        V = ref.shape[0]
        feats = [{"ref": (g["ref_fea"][v:v + 1].to(dev), g["ref_nc_sum"][v:v + 1].to(dev), g["ref_nc"][v:v + 1].to(dev)),
                  "src": (g["src_fea"][v:v + 1].to(dev), g["src_nc_sum"][v:v + 1].to(dev), None)} for v in range(V)]
        out = model.stage_net(feats, g["cams"].to(dev), depth_values=g["hyp"].to(dev), num_depth=hyp.shape[0],
                              cost_regularization=model.cost_regularization[stage], stage_idx=stage)
        res[f"g1_{tag}"] = {"volume_max_abs": float((vol.cpu() - g["volume_mean"]).abs().max()),
                            "entropy_max_abs": float((ent.cpu() - g["entropy"]).abs().max()),
                            "vis_max_abs": float((vis.cpu() - g["vis_w"]).abs().max()),
                            "depth_mean_l1": float((out["depth"].cpu() - g["depth"]).abs().mean())}
    for tag, refine in (("norefine", False), ("refine", True)):
        g = load_golden(f"g6_forward_{tag}")
        m = seeded_init_(CDSMVSNet(refine=refine, depth_interals_ratio=(4.0, 1.5, 0.75)), 7).eval().to(dev)
        cams = {k[4:]: v for k, v in g.items() if k.startswith("cam_")}
        out = m(g["imgs"].to(dev), cams, g["depth_values"], temperature=0.01)
        res[f"g6_{tag}"] = {f"stage{s}_depth_mean_l1": float((out[f"stage{s}"]["depth"].cpu() - g[f"stage{s}_depth"]).abs().mean())
                            for s in (1, 2, 3)}
    # timing at M1
    h, w, D, C, N = 512, 640, 192, 8, 5
    feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
    cams = synth.stage_cameras(N, h, w, seed=0)
    hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
    vis = torch.rand(N - 1, h, w, device=dev)
    mats = geometry.warp_matrices(cams[0])

    def t(fn, n=10):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    res["M1_ms"] = {"K1_warp_entropy": t(lambda: ops.warp_entropy(ref, src, mats, hyp)),
                    "K3_warp_aggregate": t(lambda: ops.warp_aggregate(ref, src, vis, mats, hyp)),
                    "K3_warp_aggregate_channels_last": t(lambda: ops.warp_aggregate(ref, src, vis, mats, hyp, channels_last=True))}
print(json.dumps(res))
