"""The view-sharded PRODUCT path on the HIP kernels (reference sums: models/model.py:57-60,74):
``shard_views(model)`` / ``ViewShard.run_stage`` = zero-initialised flat buffer, K1, visibility CNN, un-normalised K3
written into a view of it, ONE all-reduce, ``volume_normalize_``, CostRegNet, regression.

* world size 1 (a single-process gloo group): the sharded cascade must equal the unsharded one;
* world size 2 on ONE device (two processes pinned to cuda:0, gloo — RCCL refuses two ranks on one GPU): every rank's
  sharded forward against the unsharded forward, depth mean <= 1e-3, stage volume <= 1e-6 (fp32 re-association only);
* ``nn.DataParallel`` (the reference's own multi-GPU mode, base/base_trainer.py:17-18, test.py:185-186) with two
  replicas on cuda:0, and the packed-weight cache invalidation rules."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _inputs(N, H, W, seed, dev):
    from cds_mvsnet_amd import synth
    imgs = synth.make_images(N, H, W, seed=seed).to(dev)
    cams = synth.make_cameras(N, H, W, refine=False, seed=seed)     # host-side cameras, as the harness passes them
    return imgs, cams, synth.make_depth_values()


def _model(dev, refine=False, seed=7):
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_
    return seeded_init_(CDSMVSNet(refine=refine, depth_interals_ratio=(4.0, 1.5, 0.75)), seed).eval().to(dev)


def _stage_inputs(V, C, D, h, w, dev):
    from cds_mvsnet_amd import synth
    feats = synth.make_pair_features(V, C, h, w, seed=31, sharp=True)
    cams = synth.stage_cameras(V + 1, h, w, seed=32)
    hyp = synth.make_hypotheses(D, h, w, seed=33)
    dfe = [{k: tuple(t.to(dev) for t in f[k]) for k in ("ref", "src")} for f in feats]
    return dfe, cams, hyp.to(dev)


def _unsharded_volume(model, dfe, cams, hyp, stage):
    from cds_mvsnet_amd import geometry, ops
    ref = torch.stack([f["ref"][0][0] for f in dfe]).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].contiguous()) for f in dfe])
    ref_nc = torch.stack([f["ref"][2][0, 0] for f in dfe]).contiguous()
    vol, _, _, _ = model.stage_net.aggregate(ref, src, ref_nc, geometry.warp_matrices(cams[0]), hyp[0].contiguous(), stage)
    return vol


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_shard_views_world1_equals_unsharded(backend):
    """backend "nccl" = RCCL with a one-rank communicator: the only RCCL execution a one-GPU box allows (two ranks on one device are
    refused as duplicate GPUs).  It runs the library, the device all-reduce and the nccl branches of the exchanges (no staging through
    the host, the communication stream of HaloComm) at least once per round; the N > 1 traffic itself needs a node."""
    from cds_mvsnet_amd import distributed as cdist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = _model(dev)
        imgs, cams, dv = _inputs(4, 128, 160, 3, dev)
        with torch.no_grad():
            want = model(imgs, cams, dv, temperature=0.01)
            for exchange in ("allreduce", "reduce_scatter", "slab"):   # world 1: one slab = the whole grid, no neighbours
                sh = cdist.shard_views(model, exchange=exchange)
                sh.keep_volume = True
                got = model(imgs, cams, dv, temperature=0.01)
                assert sh.exchanges == 3 and sh.local_views(3) == [0, 1, 2]
                for k in ("stage1", "stage2", "stage3"):
                    assert (got[k]["depth"] - want[k]["depth"]).abs().mean() < 1e-3, k
                    assert (got[k]["photometric_confidence"] - want[k]["photometric_confidence"]).abs().mean() < 1e-3, k
                    assert (got[k]["norm_curv"] - want[k]["norm_curv"]).abs().max() < 1e-5, k
            # single stage, volume level: normalising after the (identity) exchange == the fused normalisation of K3
            model._view_shard = None
            dfe, scams, hyp = _stage_inputs(3, 8, 16, 40, 64, dev)
            vol = _unsharded_volume(model, dfe, scams, hyp, 2)
            runner = cdist.ViewShardedStage(model)
            runner.shard.keep_volume = True
            out = runner(dfe, scams, hyp, 16, 2)
            full = model.stage_net(dfe, scams, depth_values=hyp, num_depth=16,
                                   cost_regularization=model.cost_regularization[2], stage_idx=2)
            assert (runner.shard.last_volume - vol).abs().max() < 1e-6
            assert (out["depth"] - full["depth"]).abs().mean() < 1e-3
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cds_mvsnet_amd import distributed as cdist
        dev = torch.device("cuda:0")                                # both ranks on the one GPU of the box
        torch.cuda.set_device(dev)
        model = _model(dev)
        # world 3: N = 4 (one source view per rank); world 2: N = 7, BASELINE config 4's view count at a reduced image size - six
        # source views, three per rank ({0, 2, 4} | {1, 3, 5}); the slab exchanges then sweep all six on every rank: K3 in two passes
        # (more than four views), the feature all-gather over two FeatureNet groups
        N = 7 if world == 2 else 4
        imgs, cams, dv = _inputs(N, 128, 160, 5, dev)
        res = {}
        with torch.no_grad():
            want = model(imgs, cams, dv, temperature=0.01)          # unsharded, on this rank
            for exchange in ("allreduce", "p2p", "reduce_scatter", "slab"):
                sh = cdist.shard_views(model, exchange=exchange)
                got = model(imgs, cams, dv, temperature=0.01)
                res[exchange] = [float((got[k]["depth"] - want[k]["depth"]).abs().mean()) for k in ("stage1", "stage2", "stage3")]
                res[exchange + "_conf"] = max(float((got[k]["photometric_confidence"] - want[k]["photometric_confidence"]).abs().mean())
                                              for k in ("stage1", "stage2", "stage3"))
                res[exchange + "_nc"] = max(float((got[k]["norm_curv"] - want[k]["norm_curv"]).abs().max()) for k in ("stage1", "stage2", "stage3"))
                res[exchange + "_views"] = sh.local_views(N - 1)
                res[exchange + "_n"] = sh.exchanges
                res[exchange + "_halo"] = sh.halo_exchanges
            model._view_shard = None
            # stage level: the all-reduced, normalised volume against the unsharded K3 output
            dfe, scams, hyp = _stage_inputs(3, 8, 16, 40, 64, dev)
            vol = _unsharded_volume(model, dfe, scams, hyp, 2)
            runner = cdist.ViewShardedStage(model)
            runner.shard.keep_volume = True
            runner(dfe, scams, hyp, 16, 2)
            res["volume"] = float((runner.shard.last_volume - vol).abs().max())
        torch.cuda.synchronize()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_views_two_ranks_on_one_device(world):
    """The full CASCADE under every exchange (all-reduce, p2p, reduce_scatter, slab) against the unsharded forward of the same rank: depth
    mean-L1 <= 1e-3 per stage.  world 2: N = 7 (config 4's view count), views {0, 2, 4} | {1, 3, 5}; world 3: one view per rank and UNEVEN row slabs (stage 1 has 32 rows = 4 groups of 8 over 3 ranks:
    16 + 8 + 8; a rank's coarsest CostRegNet level then holds a single row)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if world == 2:
        assert res[0]["allreduce_views"] == [0, 2, 4] and res[1]["allreduce_views"] == [1, 3, 5]
    else:
        assert [res[r]["allreduce_views"] for r in range(3)] == [[0], [1], [2]]
    for r in range(world):
        for exchange in ("allreduce", "p2p", "reduce_scatter", "slab"):
            assert res[r][exchange + "_n"] == 3                     # ONE volume exchange per cascade stage
            assert max(res[r][exchange]) < 1e-3, (r, exchange, res[r][exchange])
            assert res[r][exchange + "_conf"] < 1e-3 and res[r][exchange + "_nc"] < 1e-5, (r, exchange)
        # reduce_scatter: rows of the sum + slab-parallel CostRegNet on the HIP kernels: 11 one-row halo exchanges per stage
        assert res[r]["reduce_scatter_halo"] == 33 and res[r]["slab_halo"] == 33 and res[r]["allreduce_halo"] == 0
        assert res[r]["volume"] < 1e-6, res[r]["volume"]


def test_data_parallel_two_replicas_one_device():
    """The reference wraps the model in nn.DataParallel when n_gpu > 1: replicas are shallow copies that receive freshly
    broadcast weights each forward — they must pack from THEIR tensors (not the original module's cached, device-0
    pointers) and launch on their own device."""
    dev = torch.device("cuda:0")
    model = _model(dev)
    from cds_mvsnet_amd import synth
    B, N, H, W = 2, 3, 64, 96
    imgs = torch.cat([synth.make_images(N, H, W, seed=40 + b) for b in range(B)]).to(dev)
    cams_l = [synth.make_cameras(N, H, W, seed=40 + b) for b in range(B)]
    cams = {k: torch.cat([c[k] for c in cams_l]).to(dev) for k in cams_l[0]}   # device tensors: DataParallel scatters them
    dv = synth.make_depth_values().repeat(B, 1).to(dev)
    with torch.no_grad():
        want = model(imgs, cams, dv, temperature=0.01)
        dp = torch.nn.DataParallel(model, device_ids=[0, 0])
        got = dp(imgs, cams, dv, temperature=0.01)
    assert got["depth"].shape == want["depth"].shape
    assert (got["depth"] - want["depth"]).abs().mean() < 1e-3
    assert (got["stage1"]["photometric_confidence"] - want["stage1"]["photometric_confidence"]).abs().mean() < 1e-3


def test_packed_weights_follow_parameter_updates():
    """load_state_dict / .to() / optimizer-style in-place updates are tracked; `.data` edits need repack()."""
    dev = torch.device("cuda:0")
    model = _model(dev)
    imgs, cams, dv = _inputs(3, 64, 96, 9, dev)
    with torch.no_grad():
        a = model(imgs, cams, dv, temperature=0.01)["depth"].clone()
        other = _model(dev, seed=8)
        model.load_state_dict(other.state_dict())                   # same storage, new values: the post-hook invalidates
        b = model(imgs, cams, dv, temperature=0.01)["depth"].clone()
        want_b = other(imgs, cams, dv, temperature=0.01)["depth"]
        assert (b - want_b).abs().mean() < 1e-3 and (a - b).abs().mean() > 1e-2
        for p in model.cost_regularization.parameters():
            p.data.mul_(0.5)                                        # bypasses the version counter...
        model.repack()                                              # ...so the documented contract is an explicit repack
        c = model(imgs, cams, dv, temperature=0.01)["depth"]
        assert (c - b).abs().mean() > 1e-3
        for p in model.cost_regularization.parameters():
            p.mul_(2.0)                                             # in-place through autograd's counter: tracked
        d = model(imgs, cams, dv, temperature=0.01)["depth"]
        assert (d - b).abs().mean() < 1e-3


def test_config5_training_step_768x576():
    """BASELINE config 5: BlendedMVS training shape 768x576, N=5, refine=True (configs/config_blended.json), one
    optimisation step (fp32 kernels; two fresh models): finite losses that agree, every layer updated."""
    import numpy as np
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, train as T
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    B, N, H, W = 1, 5, 576, 768
    imgs = synth.make_images(N, H, W, seed=21).to(dev)
    cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=True, seed=21).items()}
    dv = synth.make_depth_values().to(dev)
    g = torch.Generator().manual_seed(9)
    base = 600.0 + 120.0 * F.interpolate(torch.rand(B, 1, 6, 8, generator=g), (H, W), mode="bicubic", align_corners=False)[:, 0]
    gt, mask = {}, {}
    for s, sc in (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)):     # refine=True: stages at 1/8, 1/4, 1/2, 1
        gt[s] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous().to(dev)
        mask[s] = (torch.rand(B, H // sc, W // sc, generator=g) > 0.15).float().to(dev)
    sample = {"imgs": imgs, "proj_matrices": cams, "depth_values": dv, "depth": gt, "mask": mask}
    losses = {}
    for rep in (0, 1):
        model = seeded_init_(CDSMVSNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75)), 7).to(dev)
        opt = T.make_optimizer(model, lr=1e-3)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        l0, d0 = T.train_step(model, opt, sample, temperature=0.1, reducer=T.GradAllReducer(model.parameters(), module=model))
        assert np.isfinite(l0) and np.isfinite(d0) and d0 > 0
        moved = sum(1 for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n]))
        assert moved > 0.9 * len(before)
        losses[rep] = l0
        del model, opt
        torch.cuda.empty_cache()
    assert abs(losses[1] - losses[0]) <= 1e-4 * abs(losses[0]), losses


def test_bench_two_ranks_dry_run_on_one_device(tmp_path):
    """the WHOLE N > 1 bench path - torch.distributed.run with two ranks, the timed replicas, and the `viewshard`
    measurements of all three exchanges (all-reduce, reduce_scatter, slab: slab-parallel CostRegNet with its 11 halo exchanges per
    stage) incl. the self-check - executed every round as a dry run: both ranks on cuda:0, gloo instead of RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    env = dict(os.environ, CDS_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dist-backend", "gloo", "--no-extras", "--no-pmc", "--cpu-sample", "0.1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    vs = line["viewshard"]
    assert vs["ranks"] == 2 and vs["backend"] == "gloo"
    for mode in ("reduce_scatter", "slab"):
        assert "error" not in vs[mode], vs[mode]
        assert vs[mode]["stage"]["halo_exchanges_per_depth_map"] == 11
        assert vs[mode]["stage"]["ms_per_depth_map"] > 0 and vs[mode]["cascade"]["ms_per_depth_map"] > 0
    assert vs["stage"]["ms_per_depth_map"] > 0 and vs["cascade"]["exchanges_per_depth_map"] == 3
    # the first node run explains itself - the north-star strong-scaling figures at top level, per exchange mode, with the
    # measured time of a halo exchange and the RCCL settings in force
    ss = line["strong_scaling"]
    assert ss["ranks"] == 2 and set(ss["rccl"]) >= {"NCCL_ALGO", "NCCL_PROTO", "version"}
    for key in ("stage_M1_ms_per_depth_map", "cascade_M4_ms_per_depth_map"):
        assert set(ss[key]) == {"allreduce", "reduce_scatter", "slab"} and all(v is not None and v > 0 for v in ss[key].values()), ss
    assert all(v is not None and v > 0 for v in ss["halo_exchange_us_measured"].values()), ss


def test_bench_one_rank_over_rccl(tmp_path):
    """`bench.py --force-dist`: the view-shard side measurements (all three exchanges, stage + cascade) with a ONE-rank RCCL communicator -
    every nccl branch of the bench and of the exchanges runs on the device at least once per round (the self-check must pass:
    ranks == N, backend nccl)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist", "--no-extras",
           "--no-pmc", "--cpu-sample", "0"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    vs = line["viewshard"]
    assert "error" not in vs, vs
    assert vs["backend"] == "nccl" and vs["ranks"] == 1 and all(vs["self_check"].values())
    for mode in ("reduce_scatter", "slab"):
        assert "error" not in vs[mode], vs[mode]
    assert vs["stage"]["ms_per_depth_map"] > 0


def test_bench_watchdog_cuts_a_hung_side_measurement(tmp_path):
    """A view-shard side measurement that hangs (ranks disagreeing about a collective on the first node run) must not cost the timed
    headline: the watchdog of `bench.py` prints the line with the failure noted and ends the process with exit code 0, well before the
    process group's own timeout aborts it without output.  `CDS_BENCH_TEST_HANG=1` puts a sleep where the measurement would run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", CDS_BENCH_TEST_HANG="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-dist", "--no-extras",
           "--no-pmc", "--cpu-sample", "0", "--viewshard-timeout", "3"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)     # (the sleep is an hour)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["roofline"] is not None
    assert "watchdog" in line["viewshard"]["error"] and line["strong_scaling"]["stage_M1_ms_per_depth_map"]["allreduce"] is None
