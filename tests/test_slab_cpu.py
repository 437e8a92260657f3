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
        from cds_mvsnet_amd.slab import HaloComm, slab_cost_regularization, slab_rows
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
        got = slab_cost_regularization(layers if b > a else None, comm, vol_full[:, a:b].contiguous())
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
        q.put((rank, err_pre, err_depth, err_conf, err_nc, sh.exchanges, sh.halo_exchanges, sh.exchanged_bytes, b - a))
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
    for rank, err_pre, err_depth, err_conf, err_nc, nx, nhalo, sent, n in res:
        assert err_pre < 1e-10, (rank, err_pre)            # float64: the slab network IS the unsharded network
        assert err_depth < 1e-8 and err_conf < 1e-10 and err_nc < 1e-12, (rank, err_depth, err_conf, err_nc)
        assert nx == 1 and nhalo == 11                     # one volume exchange + 11 one-row halo exchanges per stage
        # exactly the rows a rank does not own leave it, once ((world-1)/world of the partial sums for an even partition)
        assert sent == (D * w * C + 2 * w) * (h - n) * 8 * (1 if world > 1 else 0)


def test_slab_rows_partition():
    from cds_mvsnet_amd.slab import slab_rows
    assert slab_rows(64, 8) == [(8 * r, 8 * r + 8) for r in range(8)]
    assert slab_rows(40, 3) == [(0, 16), (16, 32), (32, 40)]
    assert slab_rows(16, 4) == [(0, 8), (8, 16), (16, 16), (16, 16)]
    with pytest.raises(ValueError):
        slab_rows(20, 2)
