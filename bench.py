#!/usr/bin/env python
"""bench.py — depth-maps/sec of the CDS-MVSNet plane-sweep hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY §8(d) "M1"): one single-stage plane sweep per step — per-pair
feature maps [C=8, 512, 640] for N=5 views, D=192 per-pixel hypotheses -> K1 warp-correlate-entropy, visibility
CNN, K3 warp-aggregate (2.0 GB volume), CostRegNet (3D U-Net, 625 GFLOP fp32), soft-argmin depth + confidence.
Inputs are synthetic (seeded), resident in HBM before the timed region; weights are seeded random (no network).

Multi-GPU: default ``--parallelism replicas`` — every rank computes the depth map of its own reference view
(the reference's only multi-GPU mode, nn.DataParallel batch scatter, as one process per GPU; no data-path
collective; weak scaling) — that is `value`.  Every line with N > 1 ALSO carries a `viewshard` object: the
north_star exchange (source views of ONE depth map sharded over the ranks, one RCCL all-reduce of
volume_sum ++ vis_sum ++ nc_sum per stage) measured after the timed region on the M1 stage and on the BASELINE
config-4 cascade (Tanks&Temples 1920x1056, N=7, through `shard_views(model)`): ms per depth map, all-reduce bytes
and ms, ranks in the RCCL communicator, NCCL_ALGO.  ``--parallelism viewshard`` makes that mode the timed one
(strong scaling).  ``--workload M2|M3|M4`` times the full three-stage cascade (640x512 N=5 / DTU 1600x1184 N=5 /
Tanks&Temples 1920x1056 N=7) instead of the single-stage M1; ``--workload T5`` times BASELINE config 5, one
BlendedMVS-shaped training step (768x576, N=5, refine=True, data-parallel gradient all-reduce).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (fused warp-aggregate kernel,
HBM bound) and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample) objects.
"""
import argparse
import gc
import json
import os
import sys
import time
import statistics

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md); 6.29e12 measured copy ceiling
FP32_PEAK = 157.3e12       # FLOP/s vector / f32-MFMA

WORKLOADS = {
    # name: (h, w, D, C, n_views)
    "M1": (512, 640, 192, 8, 5),
    "M1b": (128, 160, 192, 32, 5),
    "tiny": (64, 80, 48, 8, 3),
}
CASCADES = {
    # name: (H, W, n_views): full three-stage forward (FeatureNet + 3 x StageNet), refine=False
    "M2": (512, 640, 5),
    "M3": (1184, 1600, 5),
    "M4": (1056, 1920, 7),
    "tinycascade": (128, 160, 3),
}
TRAIN = {
    # name: (H, W, n_views, refine)   BASELINE config 5 (configs/config_blended.json)
    "T5": (576, 768, 5, True),
    "tinytrain": (64, 96, 3, False),
}
NDEPTHS, RATIOS, STAGE_C = (48, 32, 8), (4.0, 1.5, 0.75), (32, 16, 8)


def cascade_algorithmic_bytes(H, W, n_views):
    """Sum of the three stages' warp-aggregate algorithmic bytes (stage s: H/scale x W/scale, D_s planes, C_s channels)."""
    return sum(algorithmic_bytes(H // sc, W // sc, D, C, n_views) for sc, D, C in zip((4, 2, 1), NDEPTHS, STAGE_C))


def algorithmic_bytes(h, w, D, C, n_views):
    """volume write + hypotheses read + ref/src feature maps + per-view visibility maps, fp32."""
    hw, V = h * w, n_views - 1
    return 4 * (C * D * hw + D * hw + 2 * V * C * hw + V * hw)


def costreg_flops(h, w, D, C):
    """Multiply-adds x2 of the 11 3x3x3 (de)convolutions of CostRegNet (module.py:270-315), base 8."""
    v = D * h * w
    b = 8
    f = 27 * C * b * v                                   # conv0
    f += 27 * b * 2 * b * v / 8 + 27 * 2 * b * 2 * b * v / 8       # conv1, conv2
    f += 27 * 2 * b * 4 * b * v / 64 + 27 * 4 * b * 4 * b * v / 64  # conv3, conv4
    f += 27 * 4 * b * 8 * b * v / 512 + 27 * 8 * b * 8 * b * v / 512  # conv5, conv6
    f += 27 * 8 * b * 4 * b * v / 512 + 27 * 4 * b * 2 * b * v / 64 + 27 * 2 * b * b * v / 8  # deconvs (per input voxel)
    f += 27 * b * 1 * v                                  # prob
    return 2.0 * f


def costreg_min_bytes(h, w, D, C):
    """Compulsory fp32 activation traffic of CostRegNet: every layer reads its input (+ skip) and writes its output once."""
    v = D * h * w
    b = 8
    t = (C + b) * v                                   # conv0
    t += b * v + 2 * b * v / 8 + 2 * (2 * b * v / 8)  # conv1, conv2
    t += 2 * b * v / 8 + 4 * b * v / 64 + 2 * (4 * b * v / 64)   # conv3, conv4
    t += 4 * b * v / 64 + 8 * b * v / 512 + 2 * (8 * b * v / 512)  # conv5, conv6
    t += 8 * b * v / 512 + 2 * (4 * b * v / 64)       # conv7: in, skip + out
    t += 4 * b * v / 64 + 2 * (2 * b * v / 8)         # conv9
    t += 2 * b * v / 8 + 2 * (b * v)                  # conv11
    t += b * v + v                                    # prob
    return 4.0 * t


def make_workload(name, seed, device):
    from cds_mvsnet_amd import synth
    h, w, D, C, n_views = WORKLOADS[name]
    feats = synth.make_pair_features(n_views - 1, C, h, w, seed=seed + 1)
    cams = synth.stage_cameras(n_views, h, w, seed=seed)
    hyp = synth.make_hypotheses(D, h, w, seed=seed + 1)
    dfe = [{k: tuple(t.to(device) for t in f[k]) for k in ("ref", "src")} for f in feats]
    return feats, cams, hyp, dfe


def other_workloads(model, dev):
    """BASELINE's config 2 can be read as the 640x512 volume grid (M1, the headline above), as its 160x128 /
    C=32 cousin (M1b) or as the full three-stage cascade on 640x512 images (M2).  The other two are reported here,
    untimed by the driver, measured after the timed region (5 iterations each after 2 warm-ups), followed by the
    cascade at BASELINE's config 3 / 4 image sizes (M3: DTU 1600x1184, N=5; M4: Tanks&Temples 1920x1056, N=7)."""
    from cds_mvsnet_amd import synth
    out = {}

    def timeit(fn, n=5, warm=2):
        gc.collect()
        gc.disable()          # a generation-2 collection inside a 20-100 ms timed batch showed up as 3x outliers
        # at least one untimed call AFTER the collection, then synchronise and start the clock at once: the collection pause (tens of ms
        # of an idle GPU) used to sit between the warm-up and the timed batch, and the first timed forward then ran at ramping clocks
        # (scripts/ab/r05_fwd_n.py: 18.3 ms for one forward after an idle gap against 15.2 ms per forward in a batch of ten)
        for _ in range(max(warm, 1)):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        return dt / n * 1e3

    with torch.no_grad():
        # the headline workload once more with CostRegNet on the exact-fp32 kernels (sequential fmaf chains on the fp32
        # matrix / vector pipes, planar volumes) instead of the split-bf16 matrix-core kernels
        import cds_mvsnet_amd.model as cm
        if cm.USE_SPLIT_BF16:
            h, w, D, C, n_views = WORKLOADS["M1"]
            _, cams, hyp, dfe = make_workload("M1", 0, dev)
            hyp_d = hyp.to(dev)
            one = lambda: model.stage_net(dfe, cams, depth_values=hyp_d, num_depth=D,  # noqa: E731
                                          cost_regularization=model.cost_regularization[2], stage_idx=2)
            # two independent depth maps in flight on two HIP streams (latency-bound kernels of one fill the other's gaps)
            lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]

            def two():
                for st in lanes:
                    st.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(st):
                        one()
                for st in lanes:
                    torch.cuda.current_stream(dev).wait_stream(st)
            out["M1_two_streams_depth_maps_per_s"] = 2e3 / timeit(two)
            cm.USE_SPLIT_BF16 = False
            model.repack()
            try:
                out["M1_exact_fp32_costreg_ms"] = timeit(one)
            finally:
                cm.USE_SPLIT_BF16 = True
                model.repack()
            del dfe, hyp_d
        h, w, D, C, n_views = WORKLOADS["M1b"]
        _, cams, hyp, dfe = make_workload("M1b", 0, dev)
        hyp_d = hyp.to(dev)
        out["M1b_single_stage_160x128_D192_C32_ms"] = timeit(lambda: model.stage_net(
            dfe, cams, depth_values=hyp_d, num_depth=D, cost_regularization=model.cost_regularization[0], stage_idx=0))
        imgs = synth.make_images(5, 512, 640, seed=0).to(dev)
        pm = synth.make_cameras(5, 512, 640, refine=False, seed=0)
        dv = synth.make_depth_values()
        from cds_mvsnet_amd.graphed import CapturedForward
        runner = CapturedForward(model)            # hipGraph replay of the same forward (graphed.py; torch.equal to eager under -m gpu)

        def single(fn, n=5):
            """one forward issued to an idle GPU (host camera algebra + enqueue + execution), median of n"""
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return statistics.median(ts)
        out["M2_cascade_forward_640x512_N5_ms"] = timeit(lambda: model(imgs, pm, dv, temperature=0.01))
        out["M2_cascade_forward_640x512_N5_graph_ms"] = timeit(lambda: runner(imgs, pm, dv, temperature=0.01))
        # two depth maps in flight: two replayed graphs (own static buffers) on two streams - the 640x512 kernels do not fill 256 CUs, and
        # with ~0.7 ms of host work per replay (eager: ~2.3 ms of enqueue per forward) the host keeps both streams fed
        runner2 = CapturedForward(model)
        lanes2 = [torch.cuda.Stream(device=dev) for _ in range(2)]

        def two_maps():
            for st_, r_ in zip(lanes2, (runner, runner2)):
                st_.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(st_):
                    r_(imgs, pm, dv, temperature=0.01)
            for st_ in lanes2:
                torch.cuda.current_stream(dev).wait_stream(st_)
        out["M2_two_graph_streams_ms_per_depth_map"] = timeit(two_maps) / 2
        del runner2
        out["M2_single_forward_after_sync_eager_ms"] = single(lambda: model(imgs, pm, dv, temperature=0.01))
        out["M2_single_forward_after_sync_graph_ms"] = single(lambda: runner(imgs, pm, dv, temperature=0.01))
        # BASELINE configs[2] / [3] on one GPU: the DTU and Tanks&Temples image sizes through the full cascade
        for key, (hh, ww, nn) in {"M3_cascade_forward_1600x1184_N5_ms": (1184, 1600, 5),
                                  "M4_cascade_forward_1920x1056_N7_ms": (1056, 1920, 7)}.items():
            imgs = synth.make_images(nn, hh, ww, seed=0).to(dev)
            pm = synth.make_cameras(nn, hh, ww, refine=False, seed=0)
            # best of three 3-iteration batches: the first passes at a new size occasionally pay for allocator growth
            out[key] = min(timeit(lambda: model(imgs, pm, dv, temperature=0.01), n=3, warm=2 if r == 0 else 0) for r in range(3))
            out[key.replace("_ms", "_graph_ms")] = min(timeit(lambda: runner(imgs, pm, dv, temperature=0.01), n=3, warm=2 if r == 0 else 0)
                                                       for r in range(3))
            if key.startswith("M3"):
                out["M3_single_forward_after_sync_eager_ms"] = single(lambda: model(imgs, pm, dv, temperature=0.01))
                out["M3_single_forward_after_sync_graph_ms"] = single(lambda: runner(imgs, pm, dv, temperature=0.01))
            del imgs
        del runner
        k3 = k3_stage_rooflines(model, dev)
    # BASELINE configs[4] on one GPU: the BlendedMVS training step (768x576, N=5, refine, fp32: forward + final_loss + backward +
    # SGD; the weight-gradient side stream is audited on the first step) -- the driver-timed figure of SURVEY 8(f)-2
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
    H5, W5, n5, refine5 = TRAIN["T5"]
    tmodel = seeded_init_(CDSMVSNet(refine=refine5, ndepths=NDEPTHS, depth_interals_ratio=RATIOS), 0).to(dev)
    sample = train_sample(H5, W5, n5, refine5, dev)
    opt = T.make_optimizer(tmodel)
    out["T5_train_step_768x576_N5_fp32_ms"] = min(timeit(lambda: T.train_step(tmodel, opt, sample, temperature=0.1), n=5, warm=3 if r == 0 else 0)
                                                 for r in range(2))
    # the same step under the bf16-storage / f32-accumulate policy of the FeatureNet activations (config 5's "bf16" label; train.py)
    out["T5_train_step_768x576_N5_bf16storage_ms"] = min(
        timeit(lambda: T.train_step(tmodel, opt, sample, temperature=0.1, activation_storage="bf16"), n=5, warm=3 if r == 0 else 0)
        for r in range(2))
    # the same fp32 step as ONE hipGraph (train.CapturedTrainStep: forward + loss + backward + SGD recorded once, replayed with the
    # sample's geometry block); the loss is read every step, like train_step does
    tmodel = seeded_init_(CDSMVSNet(refine=refine5, ndepths=NDEPTHS, depth_interals_ratio=RATIOS), 0).to(dev)
    opt = T.make_optimizer(tmodel)
    cstep = T.CapturedTrainStep(tmodel, opt)
    out["T5_train_step_768x576_N5_fp32_graph_ms"] = min(timeit(lambda: float(cstep(sample, 0.1)[0]), n=5, warm=4 if r == 0 else 0) for r in range(2))
    # ... and with the loss read once per five steps and the cameras on the host (as a data loader delivers them): the host prepares the
    # next step's geometry while the GPU trains
    hsample = dict(sample, proj_matrices={k: v.cpu() for k, v in sample["proj_matrices"].items()}, depth_values=sample["depth_values"].cpu())
    out["T5_train_step_768x576_N5_fp32_graph_deferred_loss_ms"] = min(timeit(lambda: cstep(hsample, 0.1), n=5, warm=1) for r in range(2))
    # the bf16-storage policy under the same capture (GPU-bound there: fewer stored bytes can show)
    tmodel2 = seeded_init_(CDSMVSNet(refine=refine5, ndepths=NDEPTHS, depth_interals_ratio=RATIOS), 0).to(dev)
    opt2 = T.make_optimizer(tmodel2)
    cstep2 = T.CapturedTrainStep(tmodel2, opt2, activation_storage="bf16")
    out["T5_train_step_768x576_N5_bf16storage_graph_ms"] = min(timeit(lambda: float(cstep2(sample, 0.1)[0]), n=5, warm=4 if r == 0 else 0)
                                                              for r in range(2))
    del tmodel, opt, sample, cstep, hsample, tmodel2, opt2, cstep2
    res = {k: round(v, 3) for k, v in out.items()}
    res["M3_K3_roofline_by_stage"] = k3
    return res


def k3_stage_rooflines(model, dev, H=1184, W=1600, n_views=5):
    """`roofline` objects of the fused warp + aggregation kernel (K3) at the three stage shapes of the BASELINE config-3
    cascade (C = 32 / 16 / 8: 1.94 ms of an 18 ms forward), with the cascade's own hypothesis ranges (stage 1: the 48 planes of the
    whole depth range; stages 2 / 3: 32 / 8 planes at 1.5 / 0.75 intervals around a smooth depth map).  HIP events on the launch stream,
    median of 7 launches after 2 warm-ups; algorithmic bytes per SURVEY 8(d)."""
    import statistics
    from cds_mvsnet_amd import geometry, ops, synth
    out = {}
    V = n_views - 1
    g = torch.Generator().manual_seed(5)
    for s, (sc, D, C, ratio) in enumerate(zip((4, 2, 1), NDEPTHS, STAGE_C, RATIOS)):
        h, w = H // sc, W // sc
        feats = synth.make_pair_features(V, C, h, w, seed=11 + s)
        cams = synth.stage_cameras(n_views, h, w, seed=s)
        if s == 0:
            hyp = torch.linspace(425.0, 902.5, D).view(D, 1, 1).expand(D, h, w).contiguous()
        else:
            base = 600.0 + 120.0 * torch.nn.functional.interpolate(torch.rand(1, 1, 6, 8, generator=g), (h, w), mode="bicubic",
                                                                   align_corners=False)[0, 0]
            hyp = (base.unsqueeze(0) + (torch.arange(D, dtype=torch.float32).view(D, 1, 1) - (D - 1) // 2) * (ratio * 2.5)).contiguous()
        ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
        src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
        vis = (torch.rand(V, h, w, generator=g) * 0.9 + 0.05).to(dev)
        mats, hyp_d = ops.geo(geometry.warp_matrices(cams[0]), dev, "mats"), hyp.to(dev)   # homographies are device data (geometry block)
        ts = []
        for i in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            vol, _ = ops.warp_aggregate(ref, src, vis, mats, hyp_d, channels_last=True)
            b.record()
            b.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b))
            del vol
        ms = statistics.median(ts)
        b_alg = algorithmic_bytes(h, w, D, C, n_views)
        out[f"stage{s + 1}"] = {"shape": f"{w}x{h}, D={D}, C={C}, N={n_views}", "kernel_ms": round(ms, 4), "algorithmic_bytes": b_alg,
                               "bound": "hbm", "achieved": round(b_alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": round(b_alg / (ms * 1e-3) / HBM_PEAK, 4)}
        del ref, src, vis, hyp_d
    return out


def cpu_baseline(model_cpu, name, budget_frac, seed=0, gpu_depth=None):
    """The CPU oracle (proved equal to the reference, tests/test_oracle_golden.py) on the same workload — by default
    the FULL M1 size, one run (~20 s and ~20 GB on 64 cores); `budget_frac` < 1 (or a host with less than
    48 GB of free memory) takes the top-left (h*f) x (w*f) window instead: all D planes, all views, cost linear in the
    pixel count (profiles/r02_cpu_baseline_linearity.md).  The inputs are the ones of the timed GPU workload (same seeds), so at full
    size the oracle's depth map is also the parity reference of the run: `abs_depth_l1_vs_gpu` = mean |depth_cpu - depth_gpu| over
    the depth map the timed region produced (BASELINE.json's metric: "...; abs-depth L1 vs ref").  Returns the JSON object."""
    from cds_mvsnet_amd import synth
    from oracle import cds_oracle as O
    h, w, D, C, n_views = WORKLOADS[name]
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    need = 10.0 * C * D * h * w * 4 * budget_frac ** 2          # ~5 full-volume temporaries per view, with slack
    if avail and need > 0.8 * avail:
        budget_frac = min(budget_frac, 0.5)
    hs, ws = max(8, int(h * budget_frac) // 8 * 8), max(8, int(w * budget_frac) // 8 * 8)
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    feats = synth.make_pair_features(n_views - 1, C, hs, ws, seed=seed + 1)
    cams = synth.stage_cameras(n_views, hs, ws, seed=seed)
    hyp = synth.make_hypotheses(D, hs, ws, seed=seed + 1)
    sd = model_cpu.state_dict()
    stage = {8: 2, 16: 1, 32: 0}[C]
    with torch.no_grad():
        O.stage_forward(feats, cams, hyp[:, :8], sd, stage, exact=False)  # warm-up (thread pool, oneDNN primitives)
        t0 = time.time()                                # ONE timed run after the warm-up (~18 s on 64 cores): a stated baseline, not a statistic
        ref = O.stage_forward(feats, cams, hyp, sd, stage, exact=False)
        dt = time.time() - t0
        times = [dt]
    frac = (hs * ws) / float(h * w)
    what = "full size" if frac == 1.0 else f"window {ws}x{hs} of {w}x{h} ({frac:.4f} of the pixels, linear extrapolation)"
    out = {"value": frac / dt, "unit": "depth-maps/s", "cores": cores, "kind": "port",
           "sample": f"{name} {what}, D={D}, C={C}, N={n_views}, one run ({dt:.2f} s) after a small warm-up of the torch-CPU "
                     f"oracle (F.grid_sample path); runs: 1 (a stated baseline, not a statistic)", "runs": len(times),
           "run_seconds": [round(t, 2) for t in times]}
    if gpu_depth is not None and frac == 1.0:
        dcpu = ref["depth"].reshape(h, w).float()
        dgpu = gpu_depth.reshape(h, w).float().cpu()
        err = (dcpu - dgpu).abs()
        out["abs_depth_l1_vs_gpu"] = float(err.mean())
        out["abs_depth_max_vs_gpu"] = float(err.max())
        out["depth_range"] = [float(hyp.min()), float(hyp.max())]
    return out


def build_id():
    """sha256 (16 hex) of the loaded libcdsmvs_hip.so and of the kernel sources it was built from."""
    import hashlib
    from cds_mvsnet_amd import _lib
    out = {}
    try:
        out["lib_sha16"] = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        out["lib_sha16"] = None
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "cds_mvsnet_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".hpp", ".h")) or fn == "Makefile":
            hsh.update(open(os.path.join(csrc, fn), "rb").read())
    out["csrc_sha16"] = hsh.hexdigest()[:16]
    out["abi_version"] = int(_lib.load().cds_version())
    return out


def train_sample(H, W, n_views, refine, dev, seed=21):
    """Synthetic BlendedMVS-shaped training sample (images, multi-scale cameras, depth range, GT depth + masks per stage)."""
    import torch.nn.functional as F
    from cds_mvsnet_amd import synth
    imgs = synth.make_images(n_views, H, W, seed=seed).to(dev)
    cams = {k: v.to(dev) for k, v in synth.make_cameras(n_views, H, W, refine=refine, seed=seed).items()}
    dv = synth.make_depth_values().to(dev)
    g = torch.Generator().manual_seed(9)
    base = 600.0 + 120.0 * F.interpolate(torch.rand(1, 1, 6, 8, generator=g), (H, W), mode="bicubic", align_corners=False)[:, 0]
    gt, mask = {}, {}
    scales = (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)) if refine else (("stage1", 4), ("stage2", 2), ("stage3", 1))
    for sname, sc in scales:
        gt[sname] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous().to(dev)
        mask[sname] = (torch.rand(1, H // sc, W // sc, generator=g) > 0.15).float().to(dev)
    if not refine:
        gt["stage4"], mask["stage4"] = gt["stage3"], mask["stage3"]
    return {"imgs": imgs, "proj_matrices": cams, "depth_values": dv, "depth": gt, "mask": mask}


def measure_k3_traffic(timeout_s: float = 300.0):
    """HBM bytes of ONE K3 launch at M1, measured in THIS run: two counter-only rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc
    WRITE_SIZE`; no trace domains) over scripts/run_k3_traffic.py in a child process, collected and corrected as
    MI355X_MICROARCH.md prescribes (KiB units; on gfx950 FETCH_SIZE tallies a wide streaming read at half its bytes -> doubled),
    with the correction re-checked on the spot against `volume_normalize` (reads and writes 2 013 265 920 B exactly).
    Returns (dict | None, note)."""
    import collections, glob, shutil, sqlite3, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="cds_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    raw = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            subprocess.run([exe, "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "scripts", "run_k3_traffic.py")],
                           cwd="/tmp", env=env, timeout=timeout_s, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None, f"rocprofv3 wrote no database for {ctr}"
            agg = collections.defaultdict(list)
            for k, c, v in sqlite3.connect(dbs[0]).cursor().execute("select kernel_name, counter_name, value from counters_collection"):
                if c == ctr:
                    agg[k].append(v)
            raw[ctr] = {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}
    except Exception as e:   # a failed profiler run must not take the benchmark down
        return None, f"{type(e).__name__}: {str(e)[:160]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def pick(d, sub):
        return next((v for k, v in d.items() if sub in k), None)
    f3, w3 = pick(raw["FETCH_SIZE"], "warp_aggregate"), pick(raw["WRITE_SIZE"], "warp_aggregate")
    fc, wc = pick(raw["FETCH_SIZE"], "volume_normalize"), pick(raw["WRITE_SIZE"], "volume_normalize")
    if None in (f3, w3, fc, wc):
        return None, "kernels not found in the counter databases"
    known = 4.0 * 8 * 192 * 512 * 640
    return {"bytes": w3 + 2.0 * f3, "write_bytes": w3, "fetch_bytes_raw": f3, "fetch_bytes_corrected": 2.0 * f3,
            "calibration_volume_normalize": {"known_read_bytes": known, "fetch_raw_over_known": fc / known,
                                             "write_raw_over_known": wc / known}}, "measured"


def measure_viewshard(model, dev, dist, rank, world, exchange, stage_name="M1", cascade_name="M4"):
    """The north_star exchange on N > 1 ranks, measured outside the headline's timed region: (i) the single-stage
    workload through `ViewShardedStage`, (ii) the BASELINE config-4 cascade through `shard_views(model)`.  Same
    bracket as the headline: barrier + synchronize on both sides, max over ranks."""
    from cds_mvsnet_amd import distributed as cdist, ops, synth
    import datetime
    # the side measurement runs in its OWN communicator with a short collective timeout: a rank that fails alone inside one of the modes
    # (never run on RCCL hardware so far) leaves the others stuck for two minutes, not for the ten of the default group
    grp = dist.new_group(ranks=list(range(dist.get_world_size())), timeout=datetime.timedelta(minutes=2))

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        ops.PROFILE.clear()
        ops.PROFILE_HOST_MS.clear()
        ops.PROFILE_ON = True
        dist.barrier(group=grp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dist.barrier(group=grp); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ops.PROFILE_ON = False
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        kern = {k: sum(a.elapsed_time(b) for a, b in v) / n for k, v in ops.PROFILE.items() if v}
        for k, v in ops.PROFILE_HOST_MS.items():          # host-timed sections (gloo dry run of the halo exchange)
            kern.setdefault(k, sum(v) / n)
        kern["_halo_exchanges_timed"] = (len(ops.PROFILE.get("halo_exchange", ())) or len(ops.PROFILE_HOST_MS.get("halo_exchange", ()))) / n
        return float(t.item()) / n * 1e3, kern

    out = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "exchange": exchange,
           "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"),
           "collective": "one fp32 SUM all-reduce of volume_sum ++ vis_sum ++ nc_sum per stage (models/model.py:57-60,74)"}
    # self-check of the first RCCL run: one rank per GPU in ONE communicator, and that communicator is RCCL
    assert dist.get_world_size() == world, f"communicator of {dist.get_world_size()} ranks on a --gpus {world} run"
    out["self_check"] = {"ranks_equal_n_gpus": dist.get_world_size() == world, "backend_is_nccl": dist.get_backend() == "nccl"}
    if dev.type == "cuda" and torch.cuda.device_count() >= world and not all(out["self_check"].values()):
        raise RuntimeError(f"viewshard self-check failed: {out['self_check']} (ranks {dist.get_world_size()}, N {world}, "
                           f"backend {dist.get_backend()})")
    try:
        out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        out["rccl_version"] = None

    def one_mode(exch):
        res = {}
        key = {"reduce_scatter": "reduce_scatter", "slab": "gather_rows"}.get(exch, "allreduce")
        with torch.no_grad():
            h, w, D, C, n_views = WORKLOADS[stage_name]
            stage = {8: 2, 16: 1, 32: 0}[C]
            _, cams, hyp, dfe = make_workload(stage_name, 0, dev)
            hyp_d = hyp.to(dev)
            runner = cdist.ViewShardedStage(model, grp, exchange=exch)
            ms, kern = timed(lambda: runner(dfe, cams, hyp_d, D, stage), 5, 2)
            nbytes = 4 * cdist.ViewShard.flat_size(C, D, h, w)
            res["stage"] = {"workload": f"{stage_name}: {w}x{h}, D={D}, C={C}, N={n_views}", "ms_per_depth_map": ms,
                            "depth_maps_per_s": 1e3 / ms, "exchange_bytes_per_rank": nbytes if key == "allreduce" else
                            runner.shard.exchanged_bytes // max(1, runner.shard.exchanges),
                            "exchange_ms": kern.get(key), "local_source_views": len(runner.shard.local_views(n_views - 1)),
                            "halo_exchanges_per_depth_map": runner.shard.halo_exchanges // max(1, runner.shard.exchanges),
                            "halo_bytes_sent_per_rank": runner.shard.halo_bytes // max(1, runner.shard.exchanges),
                            "halo_exchange_us_measured": (1e3 * kern["halo_exchange"] / kern["_halo_exchanges_timed"]
                                                          if kern.get("halo_exchange") and kern.get("_halo_exchanges_timed") else None),
                            "kernel_ms": {k: round(v, 4) for k, v in sorted(kern.items()) if not k.startswith("_")}}
            if key == "allreduce" and kern.get("allreduce"):
                res["stage"]["allreduce_busbw_GBps"] = 2.0 * (world - 1) / world * nbytes / (kern["allreduce"] * 1e-3) / 1e9
            del dfe, hyp_d, runner
            H, W, nv = CASCADES[cascade_name]
            imgs = synth.make_images(nv, H, W, seed=0).to(dev)
            pm = synth.make_cameras(nv, H, W, refine=False, seed=0)
            dv = synth.make_depth_values()
            sh = cdist.shard_views(model, grp, exchange=exch)
            try:
                ms, kern = timed(lambda: model(imgs, pm, dv, temperature=0.01), 3, 2)
            finally:
                model._view_shard = None
            bytes_per_map = sum(4 * cdist.ViewShard.flat_size(Cs, Ds, H // sc, W // sc)
                                for sc, Ds, Cs in zip((4, 2, 1), NDEPTHS, STAGE_C))
            res["cascade"] = {"workload": f"{cascade_name}: cascade {W}x{H}, N={nv}, D={NDEPTHS}", "ms_per_depth_map": ms,
                              "depth_maps_per_s": 1e3 / ms, "volume_bytes": bytes_per_map, "exchanges_per_depth_map": 3,
                              "exchange_ms": kern.get(key), "local_source_views": len(sh.local_views(nv - 1)),
                              "halo_exchange_us_measured": (1e3 * kern["halo_exchange"] / kern["_halo_exchanges_timed"]
                                                            if kern.get("halo_exchange") and kern.get("_halo_exchanges_timed") else None),
                              "kernel_ms": {k: round(v, 4) for k, v in sorted(kern.items()) if not k.startswith("_")}}
            if key == "allreduce" and kern.get("allreduce"):
                res["cascade"]["allreduce_busbw_GBps"] = 2.0 * (world - 1) / world * bytes_per_map / (kern["allreduce"] * 1e-3) / 1e9
        return res

    out.update(one_mode(exchange if exchange != "reduce_scatter" else "allreduce"))     # north-star form: stage / cascade keys
    # the form that scales (round 3): rows of the sum + slab-parallel CostRegNet with per-layer halo exchange + row gather
    def guarded(exch, what):
        # a (rank-symmetric) failure of a mode that has never run on RCCL must not take the headline line down with it
        try:
            res = one_mode(exch)
        except Exception as e:       # noqa: BLE001
            model._view_shard = None
            res = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        res["collective"] = what
        return res

    out["reduce_scatter"] = guarded("reduce_scatter", "point-to-point reduce-scatter of the partial sums by rows (+ 8 halo rows per side), 11 "
                                    "one-row halo exchanges inside the slab-parallel CostRegNet, gather of 3 h w floats")
    # pixel slabs: no volume on any link (view-sharded FeatureNet + all-gather of the feature maps, then all views for own rows)
    out["slab"] = guarded("slab", "all-gather of the per-view feature maps (cascade only), 11 one-row halo exchanges per stage, gather of "
                          "3 h w floats; no cost volume crosses a link")
    # the RCCL algorithm / protocol actually in force: what the environment pins, else RCCL's own choice (NCCL_DEBUG=INFO prints it per
    # communicator; the bench does not parse logs)
    out["rccl"] = {"NCCL_ALGO": os.environ.get("NCCL_ALGO") or "unset (RCCL chooses per message size)",
                   "NCCL_PROTO": os.environ.get("NCCL_PROTO") or "unset", "NCCL_DEBUG": os.environ.get("NCCL_DEBUG") or "unset",
                   "version": out.get("rccl_version")}
    return out


def strong_scaling_summary(vs):
    """Top-level view of the north-star measurement: ms per depth map of ONE depth map sharded over the ranks, per
    exchange mode, for the single-stage M1 workload and the config-4 cascade - next to the `replicas` headline, so that the first
    N-GPU run explains itself.  Keys exist on every N > 1 line (None where a mode failed; its error is in `viewshard`)."""
    def pick(obj, part):
        try:
            return round(float(obj[part]["ms_per_depth_map"]), 4)
        except Exception:      # noqa: BLE001
            return None
    def halo(obj, part):
        try:
            v = obj[part]["halo_exchange_us_measured"]
            return None if v is None else round(float(v), 2)
        except Exception:      # noqa: BLE001
            return None
    s = {"ranks": vs.get("ranks"), "backend": vs.get("backend"), "rccl": vs.get("rccl"),
         "stage_M1_ms_per_depth_map": {"allreduce": pick(vs, "stage"), "reduce_scatter": pick(vs.get("reduce_scatter", {}), "stage"),
                                       "slab": pick(vs.get("slab", {}), "stage")},
         "cascade_M4_ms_per_depth_map": {"allreduce": pick(vs, "cascade"), "reduce_scatter": pick(vs.get("reduce_scatter", {}), "cascade"),
                                         "slab": pick(vs.get("slab", {}), "cascade")},
         "halo_exchange_us_measured": {"reduce_scatter": halo(vs.get("reduce_scatter", {}), "stage"), "slab": halo(vs.get("slab", {}), "stage")}}
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="M1", choices=sorted(WORKLOADS) + sorted(CASCADES) + sorted(TRAIN))
    ap.add_argument("--parallelism", default="replicas", choices=["replicas", "viewshard"])
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "p2p", "reduce_scatter", "slab"],
                    help="viewshard exchange: one RCCL all-reduce; reduce-scatter + all-gather as direct P2P sends; or "
                         "reduce_scatter = rows of the sum per rank + slab-parallel CostRegNet (the form that scales)")
    ap.add_argument("--train-act-storage", default="f32", choices=["f32", "bf16"],
                    help="--workload T5: storage of the FeatureNet activations between the passes (bf16 = bf16 storage / f32 accumulate)")
    ap.add_argument("--no-extras", action="store_true", help="skip the M1b / M2 / M3 / M4 side measurements")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 in this run")
    ap.add_argument("--no-viewshard", action="store_true", help="N > 1: skip the north-star view-shard measurement")
    ap.add_argument("--streams", type=int, default=1,
                    help="depth maps in flight per GPU on separate HIP streams (a step = that many depth maps; "
                         "per-kernel event timing and the roofline object need 1)")
    ap.add_argument("--cpu-sample", type=float, default=1.0,
                    help="linear window fraction for the CPU baseline (1 = the full workload once; 0 = skip)")
    ap.add_argument("--viewshard-timeout", type=float, default=420.0,
                    help="N > 1: seconds the view-shard side measurement may take before a watchdog prints the headline line without it "
                         "and ends the process (below the process group's 10-minute timeout, which aborts without output)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) in production; gloo only for single-GPU dry runs")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the view-shard side measurements also with ONE rank (the RCCL dry run a "
                         "one-GPU box allows: communicator, device collectives and the nccl branches of every exchange)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # CDS_BENCH_DEVICE pins every rank to one device (dry run of the N>1 control flow on a 1-GPU box, with gloo)
    dev_index = int(os.environ.get("CDS_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a bounded collective timeout: a rank that dies inside an exchange must not leave the others hanging until the driver's limit
        import datetime
        tmo = datetime.timedelta(minutes=10)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.dist_backend, timeout=tmo)

    from cds_mvsnet_amd import CDSMVSNet, ops, seeded_init_, synth
    kind = "stage" if args.workload in WORKLOADS else ("cascade" if args.workload in CASCADES else "train")
    refine = kind == "train" and TRAIN[args.workload][3]
    model_cpu = seeded_init_(CDSMVSNet(refine=refine, ndepths=NDEPTHS, depth_interals_ratio=RATIOS), 0).eval()
    import copy
    model = copy.deepcopy(model_cpu).to(dev)
    # replicas: every rank owns a different reference view (different seed); viewshard: same depth map everywhere
    seed = rank if args.parallelism == "replicas" else 0
    viewshard_timed = world > 1 and args.parallelism == "viewshard"
    metric = "depth-maps/sec per ref view at 640x512 N=5 D=192 (single-stage plane sweep incl. CostRegNet + regression)"
    unit = "depth-maps/s"
    if kind == "stage":
        h, w, D, C, n_views = WORKLOADS[args.workload]
        stage = {8: 2, 16: 1, 32: 0}[C]
        _, cams, hyp, dfe = make_workload(args.workload, seed, dev)
        # cameras stay on the host (the module turns them into 12 kernel-argument floats per view there); a device copy
        # is accepted as well but costs a readback per call
        cams_d, hyp_d = cams, hyp.to(dev)
        workload_desc = f"{args.workload}: single-stage StageNet volume {w}x{h}, D={D}, C={C}, N={n_views} views"
        b_alg = algorithmic_bytes(h, w, D, C, n_views)
        if viewshard_timed:
            from cds_mvsnet_amd import distributed as cdist
            runner = cdist.ViewShardedStage(model, dist.group.WORLD, exchange=args.exchange)
            def step():
                return runner(dfe, cams_d, hyp_d, D, stage)
        else:
            def step():
                return model.stage_net(dfe, cams_d, depth_values=hyp_d, num_depth=D,
                                       cost_regularization=model.cost_regularization[stage], stage_idx=stage)
    elif kind == "cascade":
        H, W, n_views = CASCADES[args.workload]
        imgs = synth.make_images(n_views, H, W, seed=seed).to(dev)
        pm = synth.make_cameras(n_views, H, W, refine=False, seed=seed)
        dv = synth.make_depth_values()
        workload_desc = f"{args.workload}: three-stage cascade forward {W}x{H}, N={n_views} views, D={NDEPTHS}, ratios {RATIOS}"
        metric = f"depth-maps/sec per ref view, full cascade at {W}x{H} N={n_views}"
        b_alg = cascade_algorithmic_bytes(H, W, n_views)
        if viewshard_timed:
            from cds_mvsnet_amd import distributed as cdist
            cdist.shard_views(model, dist.group.WORLD, exchange=args.exchange)
        def step():
            return model(imgs, pm, dv, temperature=0.01)
    else:
        from cds_mvsnet_amd import train as T
        H, W, n_views, _ = TRAIN[args.workload]
        sample = train_sample(H, W, n_views, refine, dev, seed=21 + seed)
        opt = T.make_optimizer(model)
        reducer = T.GradAllReducer(model.parameters(), module=model)       # broadcasts rank 0's weights when world > 1
        workload_desc = (f"{args.workload}: BlendedMVS-shaped training step {W}x{H}, N={n_views}, refine={refine}, "
                         + ("fp32 (the reference's precision; see cds_mvsnet_amd/train.py), " if args.train_act_storage == "f32" else
                            "bf16 storage of the FeatureNet activations / f32 accumulate (cds_mvsnet_amd/train.py), ")
                         + "SGD, flat-bucket gradient all-reduce")
        metric, unit = f"training samples/sec ({W}x{H} N={n_views} step: forward + loss + backward + all-reduce + SGD)", "samples/s"
        b_alg = None
        def step():
            l, _ = T.train_step(model, opt, sample, temperature=0.1, reducer=reducer, activation_storage=args.train_act_storage)
            return {"depth": torch.tensor([l])}

    if args.streams > 1:   # independent pipelines on separate streams: memory-bound and issue-bound kernels overlap
        one_map = step
        lanes = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]

        def step():
            out = None
            for st in lanes:
                with torch.cuda.stream(st):
                    out = one_map()
            return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    grad_ctx = torch.enable_grad() if kind == "train" else torch.no_grad()
    with grad_ctx:
        gc.collect()
        gc.disable()          # no cyclic-GC pause inside the timed region (the steps allocate no reference cycles); collected BEFORE the
        for _ in range(args.warmup):      # warm-up so that no idle gap sits between the warm-up steps and the timed ones
            out = step()
        ops.PROFILE.clear()
        ops.PROFILE_ON = args.streams == 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        ops.PROFILE_ON = False
    if dist is not None:
        t = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    maps_per_step = (1 if viewshard_timed else world) * args.streams
    value = maps_per_step * args.steps / dt
    depth_mean = float(out["depth"].float().mean().item())

    # per-kernel durations from the HIP events recorded inside the timed region (same stream as the launches); per STEP
    # (a cascade step launches every kernel family three times)
    kern = {k: sum(a.elapsed_time(b) for a, b in v) / args.steps for k, v in ops.PROFILE.items() if v}
    roof = None
    if "warp_aggregate" in kern and b_alg is not None:
        t_k3 = kern["warp_aggregate"] * 1e-3
        achieved = b_alg / t_k3
        traffic, traffic_detail, traffic_note = None, None, "static"
        if world == 1 and args.workload == "M1" and not args.no_extras and not args.no_pmc:
            torch.cuda.synchronize()
            traffic_detail, traffic_note = measure_k3_traffic()
            if traffic_detail is not None:
                traffic = traffic_detail["bytes"]
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("warp_aggregate_hbm_bytes")
            except Exception:
                traffic = None
        roof = {"kernel": "warp_aggregate_lds_kernel (K3, fused homography warp + visibility-weighted aggregation)",
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "frac_of_measured_copy_ceiling": achieved / 6.29e12,
                "algorithmic_bytes": b_alg, "kernel_ms": kern["warp_aggregate"], "traffic": traffic,
                "traffic_source": ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate counter-only "
                                   "passes over scripts/run_k3_traffic.py in a child process), WRITE + 2 x FETCH per the gfx950 "
                                   "correction, re-checked on volume_normalize's known byte count (traffic_detail)"
                                   if traffic_detail is not None else
                                   f"profiles/pmc_traffic.json: static, from the committed rocprofv3 --pmc passes of this kernel "
                                   f"(FETCH_SIZE x2 + WRITE_SIZE); in-run measurement: {traffic_note}"),
                "traffic_detail": traffic_detail}
        span = sum(kern.get(k, 0.0) for k in ("warp_entropy", "visibility_cnn", "warp_aggregate"))
        roof["k1_vis_k3_span_ms"] = span
        roof["k1_vis_k3_span_frac"] = b_alg / (span * 1e-3) / HBM_PEAK if span > 0 else None
    extra = {}
    if "costreg" in kern and kind == "stage":
        fl = costreg_flops(h, w, D, C)
        import cds_mvsnet_amd.model as cm
        tf = fl / (kern["costreg"] * 1e-3) / 1e12
        act = costreg_min_bytes(h, w, D, C)
        # split-f16 (round 6): 3 fp16 MFMAs per fp32-equivalent product set -> 2.5 PFLOP/s / 3; split-bf16: 6 -> / 6; exact path: the fp32 pipes
        nprod = 3.0 if ops.USE_SPLIT_F16 else 6.0
        extra["roofline_costreg"] = {
            "bound": (f"mfma {'f16 / 3 (split-f16)' if ops.USE_SPLIT_F16 else 'bf16 / 6 (split-bf16)'} and hbm") if cm.USE_SPLIT_BF16 else "fp32",
            "achieved": tf, "unit": "TFLOP/s (fp32-equivalent algorithmic flops)",
            "peak": (2500.0 / nprod) if cm.USE_SPLIT_BF16 else FP32_PEAK / 1e12,
            "frac": tf / ((2500.0 / nprod) if cm.USE_SPLIT_BF16 else FP32_PEAK / 1e12),
            "frac_of_fp32_peak": tf / (FP32_PEAK / 1e12), "kernel_ms": kern["costreg"], "flops": fl,
            "min_activation_bytes": act, "activation_gbs": act / (kern["costreg"] * 1e-3) / 1e9,
            "frac_of_hbm_peak": act / (kern["costreg"] * 1e-3) / HBM_PEAK}
    extra["kernel_ms"] = {k: round(v, 4) for k, v in sorted(kern.items())}

    others = None
    if world == 1 and args.streams == 1 and args.workload == "M1" and not args.no_extras:
        others = other_workloads(model, dev)
    def make_line(vs, cpu):
        line = {
            "metric": metric,
            "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if viewshard_timed else "weak",
            "vs_baseline": None,
            "dtype": "bf16-storage/f32-accumulate" if (args.workload in TRAIN and args.train_act_storage == "bf16") else "f32",
            "data": "synthetic (seeded features/cameras/hypotheses, seeded random weights)",
            "config": {"workload": workload_desc,
                       "parallelism": (args.parallelism if kind != "train" else "data-parallel") if world > 1 else "single",
                       "depth_maps_per_step_per_gpu": args.streams, "depth_mean": depth_mean,
                       "costreg_arithmetic": (("split-f16: every fp32 operand as 2 fp16 terms of (operand x power-of-two tensor scale; the "
                                               "scale from the producing kernel's measured max), 3 partial products on "
                                               "v_mfma_f32_16x16x32_f16, fp32 accumulate, exact rescaling (22 of 24 significand bits per operand: "
                                               "error vs float64 0.45-0.82x an fp32 convolution's own on every layer shape tested, "
                                               "tests/test_hip_parity.py; CDS_SPLIT_F16=0: split-bf16, CDS_CONV_EXACT=1: exact fp32)"
                                               if ops.USE_SPLIT_F16 else
                                               "split-bf16: every fp32 operand split exactly into 3 bf16 terms, 6 error-compensated "
                                               "partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate (error vs float64 <= the "
                                               "exact-fp32 fmaf-chain kernels'; CDS_CONV_EXACT=1 selects those)")
                                              if __import__("cds_mvsnet_amd.model", fromlist=["x"]).USE_SPLIT_BF16
                                              else "exact fp32 (fmaf chains on the fp32 matrix / vector pipes)")},
            "roofline": roof, "cpu_baseline": cpu, "build": build_id(),
        }
        line.update(extra)
        if others is not None:
            line["other_workloads"] = others
        if vs is not None:
            line["viewshard"] = vs
            line["strong_scaling"] = strong_scaling_summary(vs)
        return line

    vs = None
    if (world > 1 or args.force_dist) and kind != "train" and not args.no_viewshard:
        # The view-shard side measurement is the first code of a node run that exchanges data over RCCL between kernels.  It must not
        # take the timed headline down with it: an exception is recorded in the line; a HANG (ranks disagreeing about a collective) is
        # cut by a watchdog on every rank that prints the headline line with the failure noted and ends the process before the
        # process group's own timeout aborts it without output.
        import threading
        finished, emit_lock = threading.Event(), threading.Lock()

        def watchdog():
            if finished.wait(args.viewshard_timeout):
                return
            with emit_lock:
                if finished.is_set():
                    return
                if rank == 0:
                    print(json.dumps(make_line({"error": f"the view-shard side measurement did not finish within {args.viewshard_timeout:.0f} s "
                                                         "and was abandoned (watchdog); the timed steps above are complete",
                                                "ranks": world, "backend": args.dist_backend}, None)), flush=True)
                sys.stdout.flush()
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        model._view_shard = None
        try:
            if os.environ.get("CDS_BENCH_TEST_HANG") == "1":       # tests/test_sharded_gpu.py: the watchdog path itself
                time.sleep(3600)
            vs = measure_viewshard(model, dev, dist, rank, world, args.exchange)
        except Exception as e:       # noqa: BLE001  (the side measurement must not take the timed headline line down with it)
            model._view_shard = None
            vs = {"error": f"{type(e).__name__}: {str(e)[:400]}", "ranks": world, "backend": dist.get_backend()}
        with emit_lock:
            finished.set()
    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_sample > 0 and kind == "stage":
            cpu = cpu_baseline(model_cpu, args.workload, args.cpu_sample, seed=seed, gpu_depth=out["depth"][0] if args.streams == 1 else None)
        print(json.dumps(make_line(vs, cpu)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
