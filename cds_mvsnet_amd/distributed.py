"""Source-view sharding across the GPUs of one node (north_star).

``volume_sum = sum_v vis_v * (ref_v (x) warp_v)``, ``vis_sum = sum_v vis_v`` and ``nc_sum`` are plain sums
over source views (models/model.py:57-60) and ``vis_v`` only depends on view v's own data
(model.py:44-51).  One process per GPU; rank r owns the source views ``{v : v % world == r}``, runs
FeatureNet for its pairs, K1, the visibility CNN and a partial K3 WITHOUT the final division, then ONE fp32
SUM all-reduce (RCCL over xGMI on the compute stream; gloo in the CPU tests) of a single flat buffer
``[C*D*h*w | h*w | h*w]`` = volume_sum ++ vis_sum ++ nc_sum, after which every rank normalises, runs
CostRegNet + regression and holds the depth map that seeds the next stage.  The result differs from the
unsharded sum only by fp32 re-association.

xGMI is point-to-point (7 links per GPU): a ring all-reduce of the 0.5-2 GB volume is single-link bound, which
at the single-stage M1 size costs more than the warp work it parallelises — this mode
exists for the many-view / large-image configs; bench.py defaults to independent replicas.

``exchange="reduce_scatter"`` (round 3) is the form that can scale: the partial sums are reduce-scattered BY ROWS (rank r
receives rows [a_r, b_r) of the sum: (world-1)/world of the volume leaves each rank once, nothing comes back), every rank
normalises and regularises only its own rows (``slab.slab_cost_regularization``: CostRegNet with one one-row halo exchange
per layer between neighbouring ranks), runs the soft-argmin on them, and the depth / confidence / curvature rows are
gathered at the end (3 h w floats).  ``"allreduce"`` / ``"p2p"`` keep the round-1/2 behaviour (everybody regularises all).

``exchange="slab"`` shards PIXELS instead of partial sums (the plane sweep is independent per reference pixel: K1's entropy,
the visibility weight, K3's sum over views and the soft-argmin all live on one pixel; only the visibility CNN (3 px) and
CostRegNet (one row per layer) look sideways).  FeatureNet stays sharded by view, the per-view feature maps are all-gathered
once per depth map (1/C/D of a volume each), and then every rank runs K1 -> visibility CNN -> K3 over ALL views for its own
rows only (row-window kernels, ``cds_warp_*_window_f32``), the slab-parallel CostRegNet and the soft-argmin.  No cost volume
ever crosses a link.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import geometry, ops
from .slab import HaloComm, HipCostRegLayers, slab_cost_regularization, slab_rows, slab_window

Tensor = torch.Tensor


class ViewShard:
    """`exchange`: "allreduce" (one collective, the library picks ring / tree) or "p2p" (reduce-scatter + all-gather
    written as direct point-to-point transfers: every rank sends slice j of its buffer to rank j, sums the world-1
    slices it receives, then sends its reduced slice to everybody).  On xGMI every pair of GPUs has its own link, so the
    p2p form moves 2 (world-1)/world of the buffer per rank with all links busy, where a ring is bound by one link
    (SURVEY §8(e)).  Both give the same sums up to fp32 re-association."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, exchange: str = "allreduce"):
        if exchange not in ("allreduce", "p2p", "reduce_scatter", "slab"):
            raise ValueError("exchange must be 'allreduce', 'p2p', 'reduce_scatter' or 'slab'")
        self.group = group
        self.exchange = exchange
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # diagnostics: the last exchange (bytes, events) and, on request, the normalised volume of the last stage
        self.keep_volume = False
        self.last_volume: Optional[Tensor] = None
        self.exchanged_bytes = 0
        self.exchanges = 0
        self.halo_exchanges = 0          # reduce_scatter mode: one-row exchanges of the slab-parallel CostRegNet
        self.halo_bytes = 0
        self.gather_bytes = 0
        self.feature_gather_bytes = 0    # slab mode: all-gather of the per-view feature maps
        self.layers_factory = HipCostRegLayers     # tests plug torch reference layers in here

    # ---- bookkeeping (device independent) ----------------------------------------------------
    def local_views(self, n_src: int) -> List[int]:
        return [v for v in range(n_src) if v % self.world == self.rank]

    @staticmethod
    def flat_size(C: int, D: int, h: int, w: int) -> int:
        return C * D * h * w + 2 * h * w

    @staticmethod
    def split_flat(flat: Tensor, C: int, D: int, h: int, w: int) -> Tuple[Tensor, Tensor, Tensor]:
        n = C * D * h * w
        return flat[:n].view(C, D, h, w), flat[n:n + h * w].view(h, w), flat[n + h * w:].view(h, w)

    def all_reduce_partials(self, flat: Tensor) -> Tensor:
        """The single exchange step: SUM over ranks of volume_sum ++ vis_sum ++ nc_sum (in place)."""
        self.exchanges += 1
        self.exchanged_bytes += flat.numel() * flat.element_size()
        if self.exchange == "allreduce" or self.world == 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return flat
        return self._p2p_all_reduce(flat)

    def _slices(self, n: int) -> List[Tuple[int, int]]:
        step = (n + self.world - 1) // self.world
        return [(min(r * step, n), min((r + 1) * step, n)) for r in range(self.world)]

    def _p2p_all_reduce(self, flat: Tensor) -> Tensor:
        if flat.is_cuda and dist.get_backend(self.group) != "nccl":
            # gloo has no device-tensor send / recv (dry runs of the N > 1 control flow on one GPU): stage through the
            # host.  RCCL ("nccl") sends straight from HBM over xGMI.
            host = flat.cpu()
            self._p2p_all_reduce(host)
            flat.copy_(host)
            return flat
        W, me = self.world, self.rank
        sl = self._slices(flat.numel())
        a, b = sl[me]
        peers = [r for r in range(W) if r != me]
        # reduce-scatter: my slice of every peer's buffer comes to me; theirs go out
        recv = [torch.empty(b - a, dtype=flat.dtype, device=flat.device) for _ in peers]
        ops_ = []
        for i, r in enumerate(peers):
            ra, rb = sl[r]
            if rb > ra:
                ops_.append(dist.P2POp(dist.isend, flat[ra:rb], self._global(r), self.group))
            if b > a:
                ops_.append(dist.P2POp(dist.irecv, recv[i], self._global(r), self.group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        mine = flat[a:b]
        for i, r in enumerate(peers):          # rank order: the same association on every rank
            if b > a:
                mine.add_(recv[i])
        # all-gather of the reduced slices
        ops_ = []
        for r in peers:
            ra, rb = sl[r]
            if b > a:
                ops_.append(dist.P2POp(dist.isend, mine, self._global(r), self.group))
            if rb > ra:
                ops_.append(dist.P2POp(dist.irecv, flat[ra:rb], self._global(r), self.group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        return flat

    def _global(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def reduce_scatter_rows(self, vol_cl: Tensor, vis_sum: Tensor, nc_sum: Tensor, rows):
        """SUM over ranks, scattered by rows: returns this rank's WINDOW rows (own rows + 8 halo rows each side, `slab.slab_window`) of
        (volume [D][m][w][C], vis_sum [m][w], nc_sum [m][w]).
        Point-to-point: rank r sends rows [a_j, b_j) of its partial sums to every rank j (packed into one message per peer)
        and adds the world-1 messages it receives in rank order.  (world-1)/world of the buffer leaves each rank, once."""
        self.exchanges += 1
        W, me = self.world, self.rank
        D, h, w, C = vol_cl.shape
        rows = [slab_window(ra, rb, h) if rb > ra else (ra, rb) for ra, rb in rows]     # own rows + the 8-row halo the slab network reads
        a, b = rows[me]

        def pack(ra, rb):
            return torch.cat((vol_cl[:, ra:rb].reshape(-1), vis_sum[ra:rb].reshape(-1), nc_sum[ra:rb].reshape(-1)))

        mine = pack(a, b)
        if W == 1:
            parts = [mine]
        else:
            staged = vol_cl.is_cuda and dist.get_backend(self.group) != "nccl"     # gloo dry runs: through the host
            peers = [r for r in range(W) if r != me]
            send = {r: pack(*rows[r]) for r in peers if rows[r][1] > rows[r][0]}
            if staged:
                send = {r: t.cpu() for r, t in send.items()}
            recv = {r: torch.empty(mine.numel(), dtype=mine.dtype, device="cpu" if staged else mine.device)
                    for r in peers if b > a}
            ops_ = [dist.P2POp(dist.isend, t, self._global(r), self.group) for r, t in send.items()]
            ops_ += [dist.P2POp(dist.irecv, t, self._global(r), self.group) for r, t in recv.items()]
            self.exchanged_bytes += sum(t.numel() * t.element_size() for t in send.values())
            if ops_:
                for req in dist.batch_isend_irecv(ops_):
                    req.wait()
            parts = [mine if r == me else recv[r].to(mine.device) for r in range(W)] if b > a else [mine]
        tot = parts[0].clone()
        for t in parts[1:]:                     # rank order
            tot.add_(t)
        n = b - a
        nv = D * n * w * C
        return tot[:nv].view(D, n, w, C), tot[nv:nv + n * w].view(n, w), tot[nv + n * w:].view(n, w)

    def _run_stage_slabs(self, model, vol_cl: Tensor, vis_sum: Tensor, nc_sum: Tensor, hyp: Tensor, stage_idx: int,
                         n_src_total: int):
        """reduce_scatter mode after the local partial sums: rows -> normalise -> slab CostRegNet -> soft-argmin -> gather."""
        D, h, w, C = vol_cl.shape
        rows = slab_rows(h, self.world)
        a, b = rows[self.rank]
        with ops.prof("reduce_scatter"):
            vol_r, vis_r, nc_r = self.reduce_scatter_rows(vol_cl, vis_sum, nc_sum, rows)
        comm = HaloComm(self.group, rows)
        out = torch.zeros((3, h, w), dtype=hyp.dtype, device=hyp.device)
        if b > a:
            lo, _ = slab_window(a, b, h)
            vol_r = vol_r.contiguous()
            self._normalize_rows(vol_r, vis_r.contiguous())
            if self.keep_volume:
                self.last_volume = vol_r[:, a - lo:b - lo].permute(3, 0, 1, 2)
            prob_pre = slab_cost_regularization(self.layers_factory(model.cost_regularization[stage_idx]), comm, vol_r, a, b, h)
            depth, conf = self._regress_rows(prob_pre.contiguous(), hyp[:, a:b].contiguous())
            out[0, a:b], out[1, a:b], out[2, a:b] = depth, conf, nc_r[a - lo:b - lo] / n_src_total
        else:
            slab_cost_regularization(None, comm, vol_r, a, b, h)
        self.halo_exchanges += comm.exchanges
        self.halo_bytes += comm.bytes_sent
        with ops.prof("gather_rows"):
            if self.world > 1:
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)     # disjoint rows: a gather written as a sum
                self.gather_bytes += out.numel() * out.element_size()
        return out[0], out[1], out[2]

    # ---- pixel-slab mode ("slab") --------------------------------------------------------------
    def gather_features(self, feats, n_src: int):
        """All-gather of the view-sharded FeatureNet outputs: ``feats[stage] = (ref_chw [Vl,C,h,w], src_hwc [Vl,h,w,C],
        nc_sum [2 Vl,h,w], nc_abs [2 Vl,h,w])`` for this rank's views (None if it has none; per-image maps are ordered
        [reference copies ..., sources ...]) -> the same structure for ALL n_src views in view order.  ONE collective:
        every view travels as one flat record of fixed size, every rank contributes ceil(n_src / world) records."""
        W = self.world
        if W == 1:
            return feats
        mine = self.local_views(n_src)
        per_rank = (n_src + W - 1) // W
        names = sorted(self._feat_shapes) if feats is None else sorted(feats)
        shapes = self._feat_shapes if feats is None else {k: tuple(tuple(t.shape[1:]) for t in feats[k]) for k in names}
        rec = sum(int(torch.Size(sh[0]).numel()) + int(torch.Size(sh[1]).numel()) + 2 * int(torch.Size(sh[2]).numel()) +
                  2 * int(torch.Size(sh[3]).numel()) for sh in shapes.values())
        dev = self._feat_device if feats is None else feats[names[0]][0].device
        buf = torch.zeros((per_rank, rec), dtype=torch.float32, device=dev)
        Vl = len(mine)
        for i in range(Vl):
            parts = []
            for k in names:
                ref, src, ncs, nca = feats[k]
                parts += [ref[i].reshape(-1), src[i].reshape(-1), ncs[i].reshape(-1), ncs[Vl + i].reshape(-1),
                          nca[i].reshape(-1), nca[Vl + i].reshape(-1)]
            buf[i] = torch.cat(parts)
        got = [torch.empty_like(buf) for _ in range(W)]
        dist.all_gather(got, buf, group=self.group)
        self.feature_gather_bytes += buf.numel() * 4 * (W - 1)
        out = {}
        recs = [got[v % W][v // W] for v in range(n_src)]
        off = 0
        for k in names:
            s_ref, s_src, s_nc, s_na = shapes[k]
            n_ref, n_src_el, n_nc, n_na = (int(torch.Size(x).numel()) for x in (s_ref, s_src, s_nc, s_na))
            ref = torch.stack([r[off:off + n_ref].view(s_ref) for r in recs])
            o = off + n_ref
            src = torch.stack([r[o:o + n_src_el].view(s_src) for r in recs])
            o += n_src_el
            ncs = torch.stack([r[o:o + n_nc].view(s_nc) for r in recs] + [r[o + n_nc:o + 2 * n_nc].view(s_nc) for r in recs])
            o += 2 * n_nc
            nca = torch.stack([r[o:o + n_na].view(s_na) for r in recs] + [r[o + n_na:o + 2 * n_na].view(s_na) for r in recs])
            off = o + 2 * n_na
            out[k] = (ref, src, ncs, nca)
        return out

    def set_feature_shapes(self, shapes, device) -> None:
        """Ranks without a view of their own still take part in ``gather_features``: they need the record layout
        ({stage: (ref, src, nc_sum, nc_abs) per-view shapes})."""
        self._feat_shapes, self._feat_device = shapes, device

    def _run_stage_pixel_slab(self, model, ref: Tensor, src: Tensor, ref_nc: Tensor, nc_sums: Tensor, mats: Tensor, hyp: Tensor,
                              stage_idx: int, n_src_total: int):
        """slab mode: ALL views (gathered features) for this rank's rows.  ref [V,C,h,w], src [V,h,w,C], ref_nc / nc_sums
        [V,h,w], hyp [D,h,w]."""
        D, h, w = hyp.shape
        rows = slab_rows(h, self.world)
        a, b = rows[self.rank]
        comm = HaloComm(self.group, rows)
        out = torch.zeros((3, h, w), dtype=hyp.dtype, device=hyp.device)
        self.exchanges += 1
        if b > a:
            # K1 + visibility CNN on the rows plus a margin of one 8-row tile (the three 3x3 layers look 3 px sideways)
            a1, b1 = max(0, a - 8), min(h, b + 8)
            ent = self._warp_entropy_rows(ref[:, :, a1:b1].contiguous(), src, mats, hyp[:, a1:b1].contiguous(), (h, a1))
            # the visibility weight is exact on the rows [a1 + 3, b1 - 3) (three 3x3 layers): the own rows need a margin of >= 3, the
            # slab window's halo rows [lo, hi) are only read by conv0 of the halo rows themselves ... which are discarded: the
            # weights there may be the window-edge approximation
            vis_w = self._visibility_rows(model, ent, ref_nc[:, a1:b1].contiguous(), stage_idx)
            lo, hi = slab_window(a, b, h)                   # = (a1, b1): K3 sweeps the slab network's halo rows as well
            vol = self._warp_aggregate_rows(ref[:, :, lo:hi].contiguous(), src, vis_w[:, lo - a1:hi - a1].contiguous(), mats,
                                            hyp[:, lo:hi].contiguous(), (h, lo))
            if self.keep_volume:
                self.last_volume = vol[:, a - lo:b - lo].permute(3, 0, 1, 2)
            prob_pre = slab_cost_regularization(self.layers_factory(model.cost_regularization[stage_idx]), comm, vol, a, b, h)
            depth, conf = self._regress_rows(prob_pre.contiguous(), hyp[:, a:b].contiguous())
            out[0, a:b], out[1, a:b], out[2, a:b] = depth, conf, nc_sums[:, a:b].sum(dim=0) / n_src_total
        else:
            slab_cost_regularization(None, comm, hyp.new_zeros((D, 0, w, 1)), a, b, h)
        self.halo_exchanges += comm.exchanges
        self.halo_bytes += comm.bytes_sent
        with ops.prof("gather_rows"):
            if self.world > 1:
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
                self.gather_bytes += out.numel() * out.element_size()
        return out[0], out[1], out[2]

    # device ops of the pixel-slab path on row windows, overridable by the CPU tests (window = (grid rows, first row))
    @staticmethod
    def _warp_entropy_rows(ref_w, src, mats, hyp_w, window):
        return ops.warp_entropy(ref_w, src, mats, hyp_w, window=window)

    @staticmethod
    def _visibility_rows(model, ent, ref_nc_w, stage_idx):
        return model.stage_net.visibility(ent, ref_nc_w, stage_idx)

    @staticmethod
    def _warp_aggregate_rows(ref_w, src, vis_w, mats, hyp_w, window):
        return ops.warp_aggregate(ref_w, src, vis_w, mats, hyp_w, normalize=True, channels_last=True, window=window)[0]

    # the two per-pixel device ops of the slab path, overridable by the CPU tests
    @staticmethod
    def _normalize_rows(vol_rows: Tensor, vis_rows: Tensor) -> None:
        ops.volume_normalize_(vol_rows, vis_rows, channels_last=True)

    @staticmethod
    def _regress_rows(prob_pre: Tensor, hyp_rows: Tensor):
        return ops.softargmin_conf(prob_pre, hyp_rows)

    # ---- one stage on this rank's views (GPU) --------------------------------------------------
    def run_stage(self, model, ref: Optional[Tensor], src: Optional[Tensor], ref_nc: Optional[Tensor],
                  nc_sums: Optional[Tensor], mats: Optional[Tensor], hyp: Tensor, stage_idx: int, n_src_total: int,
                  C: Optional[int] = None, vol_bound: Optional[Tensor] = None):
        """vol_bound: a device scalar >= max |normalised volume| over ALL views of all ranks (the model passes 1: tanh features) - the
        all-reduce / p2p exchanges then run CostRegNet in split-f16 arithmetic like the unsharded forward (same bound on every rank: same
        result on every rank); None, and always for the row-slab forms (a rank's halo rows come from a neighbour whose maximum it does
        not know): split-bf16."""
        D, h, w = hyp.shape
        if ref is not None:
            C = ref.shape[1]
        cr = model.cost_regularization[stage_idx]
        cl = cr.split_bf16_supported()          # channels-last volume for the split-bf16 CostRegNet kernels
        if self.exchange in ("reduce_scatter", "slab") and not cl:
            raise RuntimeError(f"exchange='{self.exchange}' needs the channels-last split-bf16 CostRegNet path")
        if self.exchange == "slab":             # ref .. mats hold ALL views here (gather_features ran in the model's forward)
            return self._run_stage_pixel_slab(model, ref, src, ref_nc, nc_sums, mats, hyp, stage_idx, n_src_total)
        flat = torch.zeros(self.flat_size(C, D, h, w), dtype=torch.float32, device=hyp.device)
        vol, vis_sum, nc_sum = self.split_flat(flat, C, D, h, w)
        if cl:
            vol = vol.view(D, h, w, C)          # same bytes of the flat buffer, [D][h][w][C]
        if ref is not None and ref.shape[0] > 0:
            ent = ops.warp_entropy(ref, src, mats, hyp)
            vis = model.stage_net.visibility(ent, ref_nc, stage_idx).contiguous()
            ops.warp_aggregate(ref, src, vis, mats, hyp, normalize=False, volume=vol, vis_sum=vis_sum, channels_last=cl)
            nc_sum.copy_(nc_sums.sum(dim=0))
        if self.exchange == "reduce_scatter":
            return self._run_stage_slabs(model, vol, vis_sum, nc_sum, hyp, stage_idx, n_src_total)
        with ops.prof("allreduce"):
            self.all_reduce_partials(flat)
        ops.volume_normalize_(vol, vis_sum, channels_last=cl)
        if self.keep_volume:
            self.last_volume = vol.permute(3, 0, 1, 2) if cl else vol
        if cl:
            depth, conf = cr.regress(vol, hyp, bound=vol_bound)
        else:
            depth, conf = ops.softargmin_conf(cr(vol), hyp)
        return depth, conf, nc_sum / n_src_total


def shard_views(model, group: Optional["dist.ProcessGroup"] = None, exchange: str = "allreduce") -> ViewShard:
    """Make ``model.forward`` shard the source views of every depth map over the ranks of ``group``."""
    sh = ViewShard(group, exchange)
    model._view_shard = sh
    return sh


class ViewShardedStage:
    """Single-stage driver with the reference's StageNet argument layout (used by bench.py --parallelism
    viewshard): every rank is handed all views and picks its own."""

    def __init__(self, model, group=None, exchange: str = "allreduce"):
        self.model = model
        self.shard = ViewShard(group, exchange)

    def __call__(self, features, proj_matrices: Tensor, depth_values: Tensor, num_depth: int, stage_idx: int):
        sh = self.shard
        V = len(features)
        mine = list(range(V)) if sh.exchange == "slab" else sh.local_views(V)   # pixel slabs: every rank sweeps all views for its rows
        cams = proj_matrices.detach().float().cpu()
        hyp = depth_values[0].contiguous()
        C = features[0]["ref"][0].shape[1]
        if mine:
            ref = torch.stack([features[v]["ref"][0][0] for v in mine]).contiguous()
            src = torch.stack([ops.chw_to_hwc(features[v]["src"][0][0].contiguous()) for v in mine])
            ref_nc = torch.stack([features[v]["ref"][2][0, 0] for v in mine]).contiguous()
            nc_sums = torch.stack([(features[v]["ref"][1][0, 0] + features[v]["src"][1][0, 0]) / 2 for v in mine])
            mats = ops.geo(geometry.warp_matrices(cams[0])[mine].contiguous(), ref.device, "mats")   # one upload for K1 and K3
        else:
            ref = src = ref_nc = nc_sums = mats = None
        depth, conf, nc = sh.run_stage(self.model, ref, src, ref_nc, nc_sums, mats, hyp, stage_idx, V, C=C)
        return {"depth": depth.unsqueeze(0), "photometric_confidence": conf.unsqueeze(0),
                "norm_curv": nc.view(1, 1, *nc.shape)}
