"""World-size-2 gloo test of the source-view shard exchange on CPU.

The HIP kernels cannot run here, so each rank produces its shard's partial sums with the CPU oracle, the
product's ``ViewShard`` does the bookkeeping (view -> rank map, flat buffer layout) and the single all-reduce,
and the result must equal the unsharded aggregate up to fp32 re-association."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
        from cds_mvsnet_amd.distributed import ViewShard
        from oracle import cds_oracle as O
        torch.set_num_threads(2)
        h, w, D, C, N, stage = 16, 24, 8, 8, 4, 2
        sd = seeded_init_(CDSMVSNet(), 0).state_dict()
        feats = synth.make_pair_features(N - 1, C, h, w, seed=5, sharp=True)
        cams = synth.stage_cameras(N, h, w, seed=6)
        hyp = synth.make_hypotheses(D, h, w, seed=7)
        sh = ViewShard()
        mine = sh.local_views(N - 1)
        assert mine == [v for v in range(N - 1) if v % world == rank]
        flat = torch.zeros(sh.flat_size(C, D, h, w))
        vol, vis_sum, nc_sum = sh.split_flat(flat, C, D, h, w)
        P_ref = O.compose_projection(cams[:, 0])
        for v in mine:
            warped = O.warp_volume(feats[v]["src"][0], O.compose_projection(cams[:, v + 1]), P_ref, hyp)
            in_prod, ent = O.correlation_entropy(feats[v]["ref"][0], warped)
            vis = O.vis_cnn(torch.cat((ent, feats[v]["ref"][2]), 1), sd, f"stage_net.vis.{stage}")
            vol += (in_prod * vis.unsqueeze(1))[0]
            vis_sum += vis[0, 0]
            nc_sum += ((feats[v]["ref"][1] + feats[v]["src"][1]) / 2)[0, 0]
        sh.all_reduce_partials(flat)
        mean = vol / (vis_sum.unsqueeze(0).unsqueeze(0) + 1e-6)
        want = O.aggregate_views(feats, cams, hyp, sd, stage)
        err = (mean - want["volume_mean"][0]).abs().max().item()
        err_nc = (nc_sum / (N - 1) - want["nc_mean"][0, 0]).abs().max().item()
        q.put((rank, err, err_nc, len(mine)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_view_shard_allreduce(world):
    """world 2: 3 source views over 2 ranks; world 8 (a full node): ranks 3..7 own no view and contribute zeros."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[3] for r in res) == ([1, 2] if world == 2 else [0, 0, 0, 0, 0, 1, 1, 1])
    for _, err, err_nc, _ in res:
        assert err < 1e-6 and err_nc < 1e-6


def test_more_ranks_than_views_bookkeeping():
    from cds_mvsnet_amd.distributed import ViewShard

    class _Fake(ViewShard):
        def __init__(self, rank, world):
            self.rank, self.world, self.group = rank, world, None
    owners = [_Fake(r, 8).local_views(6) for r in range(8)]
    assert sorted(v for o in owners for v in o) == list(range(6))
    assert owners[6] == [] and owners[7] == []
    C, D, h, w = 8, 16, 8, 8
    flat = torch.arange(ViewShard.flat_size(C, D, h, w), dtype=torch.float32)
    vol, vs, nc = ViewShard.split_flat(flat, C, D, h, w)
    assert vol.shape == (C, D, h, w) and vs.shape == (h, w) and nc.shape == (h, w)
    assert vol.data_ptr() == flat.data_ptr() and nc[-1, -1] == flat[-1]


def _p2p_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from cds_mvsnet_amd.distributed import ViewShard
        res = []
        for n in (1, 5, 1000, 4099):                      # fewer elements than ranks, uneven slices
            g = torch.Generator().manual_seed(100 + rank)
            a = torch.randn(n, generator=g)
            ref = a.clone()
            dist.all_reduce(ref)
            got = ViewShard(exchange="p2p").all_reduce_partials(a.clone())
            res.append(float((got - ref).abs().max()))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_p2p_reduce_scatter_all_gather_equals_allreduce(world):
    """The point-to-point exchange (reduce-scatter + all-gather as direct sends) against the library
    all-reduce on gloo, world sizes 2, 3 and 8 (a full node)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, errs in res:
        assert max(errs) < 1e-6 * max(1.0, world / 2)      # fp32 re-association of `world` addends of magnitude ~1


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from cds_mvsnet_amd import CDSMVSNet, seeded_init_
        from cds_mvsnet_amd.train import GradAllReducer
        model = seeded_init_(CDSMVSNet(refine=True), 10 + rank)        # every rank starts from DIFFERENT weights
        before = torch.cat([t.detach().reshape(-1).double() for t in model.state_dict().values()])
        red = GradAllReducer(model.parameters(), module=model)          # broadcasts rank 0's parameters + buffers
        after = torch.cat([t.detach().reshape(-1).double() for t in model.state_dict().values()])
        gathered = [torch.empty_like(after) for _ in range(world)]
        dist.all_gather(gathered, after)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        changed = not torch.equal(before, after)
        # gradient averaging on the now identical replicas: rank r contributes (r + 1), the mean is (world + 1) / 2
        for p in model.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        n_coll = red.reduce()
        gerr = max(float((p.grad - (world + 1) / 2).abs().max()) for p in model.parameters())
        q.put((rank, same, changed, n_coll, gerr))
    finally:
        dist.destroy_process_group()


def test_broadcast_initial_weights_then_average_gradients_world2():
    """replicas built from different seeds must start identical (rank 0's parameters AND BatchNorm buffers),
    then the flat-bucket gradient all-reduce averages (one collective for the 3.9 MB of gradients)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "replicas differ after the broadcast"
    assert not res[0][2] and res[1][2], "rank 0 must keep its weights, rank 1 must take them"
    assert all(r[3] == 1 and r[4] < 1e-6 for r in res)
