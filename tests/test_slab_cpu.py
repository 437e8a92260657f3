"""Slab-parallel CostRegNet and the reduce-scatter exchange on CPU over gloo.

The HIP kernels cannot run here: the layer ARITHMETIC is plugged in as torch reference ops (float64, BN-folded weights of the
same CostRegNet holder), so what these tests pin is the product's slab bookkeeping -- row partition, which halo row every layer
needs, the even-row alignment of the stride-2 layers, the coarse bottom row of the transposed layers, the skip rows, the
point-to-point exchanges themselves and the final gather -- against the UNSHARDED network, at world sizes 1, 2, 3 and 8."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TorchLayers:
    """models/module.py:80-160 with torch ops on channels-last tensors, BatchNorm (eval) folded like CostRegNet._pack."""
    conv11_planar = False

    def __init__(self, cr):
        self.p = {k: v.double() for k, v in cr._pack().items() if k.endswith(".w") or k.endswith(".b")}

    def _w(self, name, transposed):
        w = self.p[name + ".w"]                       # [Cin, 27, Cout]
        w = w.reshape(w.shape[0], 3, 3, 3, w.shape[2])
        return w.permute(0, 4, 1, 2, 3).contiguous() if transposed else w.permute(4, 0, 1, 2, 3).contiguous()

    def conv(self, name, x, stride):
        y = F.relu(F.conv3d(x.permute(3, 0, 1, 2)[None], self._w(name, False), self.p[name + ".b"], stride=stride, padding=1))
        return y[0].permute(1, 2, 3, 0).contiguous()

    def deconv(self, name, x, skip, planar):
        y = F.relu(F.conv_transpose3d(x.permute(3, 0, 1, 2)[None], self._w(name, True), self.p[name + ".b"], stride=2, padding=1,
                                      output_padding=1))
        return skip + y[0].permute(1, 2, 3, 0)

    def prob(self, x):
        return F.conv3d(x.permute(3, 0, 1, 2)[None], self._w("prob", False), None, padding=1)[0, 0]


def _full_costreg(layers, vol):
    c0 = layers.conv("conv0", vol, 1)
    c2 = layers.conv("conv2", layers.conv("conv1", c0, 2), 1)
    c4 = layers.conv("conv4", layers.conv("conv3", c2, 2), 1)
    x = layers.conv("conv6", layers.conv("conv5", c4, 2), 1)
    x = layers.deconv("conv7", x, c4, False)
    x = layers.deconv("conv9", x, c2, False)
    x = layers.deconv("conv11", x, c0, False)
    return layers.prob(x)


def _softargmin(prob_pre, hyp):
    p = torch.softmax(prob_pre, dim=0)
    depth = (p * hyp).sum(0)
    return depth, p.max(dim=0).values          # any per-pixel statistic serves as the second gathered map here


def _worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from cds_mvsnet_amd import CostRegNet, seeded_init_
        from cds_mvsnet_amd.distributed import ViewShard
        from cds_mvsnet_amd.slab import HaloComm, slab_cost_regularization, slab_rows, slab_window
        torch.set_num_threads(1)
        C, D, h, w, V = case
        cr = seeded_init_(CostRegNet(C, 8), 3).eval()
        layers = TorchLayers(cr)
        g = torch.Generator().manual_seed(11)
        partial = torch.randn(V, D, h, w, C, generator=g, dtype=torch.float64)      # per-view partial volume sums
        vis = torch.rand(V, h, w, generator=g, dtype=torch.float64) + 0.1
        nc = torch.rand(V, h, w, generator=g, dtype=torch.float64)
        hyp = 400.0 + 500.0 * torch.rand(D, h, w, generator=g, dtype=torch.float64)
        # ---- unsharded reference ----
        vol_full = partial.sum(0) / (vis.sum(0)[None, :, :, None] + 1e-6)
        want_pre = _full_costreg(layers, vol_full)
        want_depth, want_conf = _softargmin(want_pre, hyp)
        # ---- (1) slab CostRegNet alone: every rank is handed its rows of the normalised volume ----
        rows = slab_rows(h, world)
        a, b = rows[rank]
        comm = HaloComm(None, rows)
        lo, hi = slab_window(a, b, h)
        vol_win = vol_full[:, lo:hi].clone()
        if b > a:                 # only the rows a - 1 .. b of the window have to be valid: poison the rest of the halo
            vol_win[:, :max(0, a - 1 - lo)] = float("nan")
            vol_win[:, b + 1 - lo:] = float("nan")
        got = slab_cost_regularization(layers if b > a else None, comm, vol_win, a, b, h)
        err_pre = (got - want_pre[:, a:b]).abs().max().item() if b > a else 0.0
        assert comm.exchanges == 11
        # ---- (2) the product's reduce_scatter stage: view shard -> rows -> slab CostRegNet -> regression -> gather ----
        sh = ViewShard(exchange="reduce_scatter")
        sh.layers_factory = lambda _cr: layers
        sh._normalize_rows = staticmethod(lambda v, s: v.div_(s[None, :, :, None] + 1e-6))
        sh._regress_rows = staticmethod(_softargmin)
        mine = sh.local_views(V)
        vol_p = partial[mine].sum(0) if mine else torch.zeros(D, h, w, C, dtype=torch.float64)
        vis_p = vis[mine].sum(0) if mine else torch.zeros(h, w, dtype=torch.float64)
        nc_p = nc[mine].sum(0) if mine else torch.zeros(h, w, dtype=torch.float64)

        class _M:
            cost_regularization = [cr]
        depth, conf, ncm = sh._run_stage_slabs(_M, vol_p, vis_p, nc_p, hyp, 0, V)
        err_depth = (depth.double() - want_depth).abs().max().item()
        err_conf = (conf.double() - want_conf).abs().max().item()
        err_nc = (ncm.double() - nc.sum(0) / V).abs().max().item()
        expect_sent = sum((slab_window(ra, rb, h)[1] - slab_window(ra, rb, h)[0]) for r, (ra, rb) in enumerate(rows) if r != rank and rb > ra)
        q.put((rank, err_pre, err_depth, err_conf, err_nc, sh.exchanges, sh.halo_exchanges, sh.exchanged_bytes, b - a,
               expect_sent * (D * w * C + 2 * w) * 8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(1, (8, 8, 16, 8, 2)), (2, (8, 8, 32, 16, 3)), (3, (16, 8, 40, 8, 4)), (8, (8, 16, 64, 8, 5)),
                                        (4, (32, 8, 16, 16, 2))])
def test_slab_costreg_and_reduce_scatter_equal_unsharded(world, case):
    """Slabs of 8 / 16 / 24 rows incl. uneven partitions (40 rows over 3 ranks = 16 + 16 + 8), a full node (8 ranks x 8 rows: every
    coarse level has ONE row per rank), and more ranks than row groups (16 rows over 4 ranks: two ranks own nothing)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C, D, h, w, V = case
    assert sum(r[8] for r in res) == h
    for rank, err_pre, err_depth, err_conf, err_nc, nx, nhalo, sent, n, expect_sent in res:
        assert err_pre < 1e-10, (rank, err_pre)            # float64: the slab network IS the unsharded network
        assert err_depth < 1e-8 and err_conf < 1e-10 and err_nc < 1e-12, (rank, err_depth, err_conf, err_nc)
        assert nx == 1 and nhalo == 11                     # one volume exchange + 11 one-row halo exchanges per stage
        # what leaves a rank, once: every other rank's rows plus the 8 halo rows per side its slab network is laid out on
        assert sent == expect_sent


def test_slab_rows_partition():
    from cds_mvsnet_amd.slab import slab_rows
    assert slab_rows(64, 8) == [(8 * r, 8 * r + 8) for r in range(8)]
    assert slab_rows(40, 3) == [(0, 16), (16, 32), (32, 40)]
    assert slab_rows(16, 4) == [(0, 8), (8, 16), (16, 16), (16, 16)]
    with pytest.raises(ValueError):
        slab_rows(20, 2)


# ------------------------------------------------------------------------------------------------
# pixel-slab mode (exchange="slab"): feature all-gather + row windows + slab CostRegNet + gather
# ------------------------------------------------------------------------------------------------
def _pixel_slab_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
        from cds_mvsnet_amd.distributed import ViewShard
        from oracle import cds_oracle as O
        torch.set_num_threads(1)
        h, w, D, C, V, stage = 40, 24, 8, 8, 3, 2
        model = seeded_init_(CDSMVSNet(), 0).eval()
        sd = model.state_dict()
        feats = synth.make_pair_features(V, C, h, w, seed=5, sharp=True)
        cams = synth.stage_cameras(V + 1, h, w, seed=6)
        hyp = synth.make_hypotheses(D, h, w, seed=7)
        want = O.stage_forward(feats, cams, hyp, sd, stage)
        # ---- (1) gather_features: view-sharded per-view maps -> all views, in view order, on every rank ----
        sh = ViewShard(exchange="slab")
        mine = sh.local_views(V)
        full = {"stageX": (torch.stack([f["ref"][0][0] for f in feats]), torch.stack([f["src"][0][0].permute(1, 2, 0) for f in feats]),
                           torch.stack([f["ref"][1][0, 0] for f in feats] + [f["src"][1][0, 0] for f in feats]),
                           torch.stack([f["ref"][2][0, 0] for f in feats] + [f["src"][2][0, 0] if f["src"][2] is not None else f["src"][1][0, 0] for f in feats]))}
        local = None
        if mine:
            idx = torch.tensor(mine)
            local = {"stageX": (full["stageX"][0][idx], full["stageX"][1][idx], torch.cat((full["stageX"][2][idx], full["stageX"][2][idx + V])),
                                torch.cat((full["stageX"][3][idx], full["stageX"][3][idx + V])))}
        sh.set_feature_shapes({"stageX": ((C, h, w), (h, w, C), (h, w), (h, w))}, torch.device("cpu"))
        got = sh.gather_features(local, V)
        gather_ok = all(torch.equal(a, b) for a, b in zip(got["stageX"], full["stageX"]))
        # ---- (2) the pixel-slab stage with the device ops replaced by the oracle on row windows ----
        P_ref = O.compose_projection(cams[:, 0])
        ent_full, prod_full = [], []
        for v in range(V):
            warped = O.warp_volume(feats[v]["src"][0], O.compose_projection(cams[:, v + 1]), P_ref, hyp)
            in_prod, ent = O.correlation_entropy(feats[v]["ref"][0], warped)
            ent_full.append(ent[0, 0])
            prod_full.append(in_prod[0])                                   # [C,D,h,w]
        calls = {}

        def warp_entropy_rows(ref_w, src, mats, hyp_w, window):
            hs, y0 = window
            calls["k1"] = (y0, y0 + ref_w.shape[2])
            assert hs == h and src.shape[1] == h and torch.equal(hyp_w, hyp[0][:, y0:y0 + ref_w.shape[2]])
            return torch.stack([e[y0:y0 + ref_w.shape[2]] for e in ent_full])

        def visibility_rows(_model, ent, ref_nc_w, s):
            return torch.stack([O.vis_cnn(torch.stack((ent[v], ref_nc_w[v]))[None], sd, f"stage_net.vis.{s}")[0, 0] for v in range(V)])

        def warp_aggregate_rows(ref_w, src, vis_w, mats, hyp_w, window):
            hs, y0 = window
            n = ref_w.shape[2]
            calls["k3"] = (y0, y0 + n)
            num = sum(prod_full[v][:, :, y0:y0 + n] * vis_w[v][None, None] for v in range(V))
            return (num / (vis_w.sum(0)[None, None] + 1e-6)).permute(1, 2, 3, 0).contiguous()

        layers = TorchLayers(model.cost_regularization[stage])

        class _L:       # float32 volumes in, float64 layer arithmetic
            conv11_planar = False
            conv = staticmethod(lambda name, x, s: layers.conv(name, x.double(), s))
            deconv = staticmethod(lambda name, x, skip, planar: layers.deconv(name, x.double(), skip.double(), planar))
            prob = staticmethod(lambda x: layers.prob(x.double()))
        sh.layers_factory = lambda _cr: _L
        sh._warp_entropy_rows = staticmethod(warp_entropy_rows)
        sh._visibility_rows = staticmethod(visibility_rows)
        sh._warp_aggregate_rows = staticmethod(warp_aggregate_rows)
        sh._regress_rows = staticmethod(lambda p, hy: tuple(t[0] for t in O.softargmin(p.float()[None], hy[None])[1:]))
        ref = full["stageX"][0]
        src = full["stageX"][1].contiguous()
        ref_nc = full["stageX"][3][:V]
        nc_sums = (full["stageX"][2][:V] + full["stageX"][2][V:]) / 2
        depth, conf, ncm = sh.run_stage(model_stub(model), ref, src, ref_nc, nc_sums, torch.zeros(V, 12), hyp[0], stage, V)
        from cds_mvsnet_amd.slab import slab_rows
        a, b = slab_rows(h, world)[rank]
        q.put((rank, gather_ok, float((depth - want["depth"][0]).abs().mean()), float((conf - want["photometric_confidence"][0]).abs().mean()),
               float((ncm - want["norm_curv"][0, 0]).abs().max()), calls, (a, b), sh.halo_exchanges, sh.feature_gather_bytes))
    finally:
        dist.destroy_process_group()


def model_stub(model):
    class _CR:
        def __init__(self, cr):
            self.cr = cr

        def split_bf16_supported(self):
            return True                     # the CPU stand-in layers are channels-last like the split-bf16 kernels

    class _M:
        cost_regularization = [_CR(c) for c in model.cost_regularization]
    # layers_factory receives the wrapped holder; the test's factory ignores it
    return _M


@pytest.mark.parametrize("world", [2, 3])
def test_pixel_slab_stage_equals_oracle(world):
    """exchange="slab": (1) the feature all-gather returns every view's maps in view order on every rank (uneven shards: 3 views
    over 2 / 3 ranks); (2) a rank asks K1 for its rows plus one 8-row tile of margin (the visibility CNN looks 3 px sideways; its
    zero padding at the WINDOW edge must not reach the rank's own rows), K3 for exactly its rows, and the gathered depth /
    confidence / curvature equal the CPU oracle's unsharded stage."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pixel_slab_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gather_ok, e_depth, e_conf, e_nc, calls, (a, b), nhalo, fbytes in res:
        assert gather_ok
        assert e_depth < 1e-3 and e_conf < 1e-3 and e_nc < 1e-6, (rank, e_depth, e_conf, e_nc)
        assert calls["k3"] == (max(0, a - 8), min(40, b + 8)) and calls["k1"] == (max(0, a - 8), min(40, b + 8))
        assert nhalo == 11 and fbytes > 0
