"""h-slab-parallel cost-volume regularisation: CostRegNet (models/module.py:270-315) on a horizontal slab of the stage grid
per rank, with ONE one-row halo exchange per layer between neighbouring ranks (11 point-to-point exchanges per stage).

Why: with the all-reduce of SURVEY §8(e) every rank regularises the WHOLE volume (models/model.py:83-92 after the sum), so
CostRegNet + regression -- 6.2 of 8.2 ms at the 640x512x192 single stage -- is not parallelised at all and the exchange moves the
full 2 GB volume.  Here the volume is reduce-scattered by rows (rank r receives the rows [a_r, b_r) of the SUM), every rank
regularises only its own rows, and the per-pixel soft-argmin runs on them too; what is gathered at the end is the depth /
confidence rows (a few MB).

Geometry.  The stage grid has h rows, h % 8 == 0 (CostRegNet has three stride-2 levels).  Slab boundaries are multiples of 8,
so a slab owns whole rows at every level L = 0..3: rows [a / 2^L, b / 2^L).  What a layer needs beyond its own rows:

  3x3x3 convolution, stride 1 (conv0/2/4/6, prob):  one row above, one below (true zero padding at the grid border);
  3x3x3 convolution, stride 2 (conv1/3/5):           output row o reads input rows 2o-1 .. 2o+1: ONE row above, none below.  The
        kernels pair rows (2o'-1, 2o', 2o'+1) from the top of the tensor they are given, so the local tensor starts at an EVEN
        global row: two rows are prepended (the outer one only feeds a discarded output row and is zero);
  transposed convolution k3 s2 p1 op1 (conv7/9/11): fine row r reads coarse rows (r-1)/2 .. (r+1)/2: ONE coarse row below, none
        above; the U-Net skip rows are the rank's own.

Buffers (round 3, second form): the first version rebuilt every layer's input with ``torch.cat`` (own rows + halo rows): at two ranks the
copies doubled the slab CostRegNet (6.8 vs 3.5 ms at 256 rows of the M1 volume).  Now every level-L activation is ONE dense buffer with
8 >> L halo rows on each side (`slab_cost_regularization`), layers run on whole buffers and an exchanged row is written in place.

The layer arithmetic itself is not in this file: `layers` is any object with conv / deconv / prob methods on dense channels-last
tensors (product: `HipCostRegLayers`, the split-bf16 matrix-core kernels; the CPU tests plug in torch reference ops to check the
slab bookkeeping and the exchanges against the unsharded network over gloo).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def slab_rows(h: int, world: int) -> List[Tuple[int, int]]:
    """Rows [a, b) of every rank: whole groups of 8 rows, as even as possible, contiguous in rank order.  Ranks beyond the
    number of groups get an empty slab (a == b)."""
    if h % 8:
        raise ValueError(f"slab_rows: h must be a multiple of 8, got {h}")
    groups = h // 8
    base, rem = divmod(groups, world)
    out, a = [], 0
    for r in range(world):
        n = (base + (1 if r < rem else 0)) * 8
        out.append((a, a + n))
        a += n
    return out


class HaloComm:
    """One-row halo exchange between vertically neighbouring slabs (point-to-point; RCCL send / recv on xGMI, gloo in the CPU
    tests).  Counts what it moves.

    The send / receive rows are PREALLOCATED per (row shape, role) and reused by every exchange of that shape (a stage has 11
    exchanges over 7 distinct row shapes, a forward three stages): an exchange is one strided copy into the send row, the
    point-to-point batch, and the caller's in-place write of the received row -- no allocation, no `.contiguous()` temporary.  On a
    GPU with RCCL the batch is issued on a dedicated communication stream; the compute stream waits for its completion EVENT (the
    host never blocks), so the only serialisation is the data dependency of the next layer on its halo row."""

    def __init__(self, group: Optional["dist.ProcessGroup"], rows: List[Tuple[int, int]]):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.rows = rows
        active = [r for r, (a, b) in enumerate(rows) if b > a]
        me = active.index(self.rank) if self.rank in active else -1
        self.upper = active[me - 1] if me > 0 else None              # the slab above (smaller y)
        self.lower = active[me + 1] if 0 <= me < len(active) - 1 else None
        self.active = me >= 0
        self.exchanges = 0
        self.bytes_sent = 0
        self._rows: dict = {}                                         # (shape, dtype, device, role) -> preallocated row
        self._stream = None

    def _global(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _row(self, own: Tensor, row_dim: int, role: str, staged: bool) -> Tensor:
        shape = list(own.shape)
        shape[row_dim] = 1
        dev = torch.device("cpu") if staged else own.device
        key = (tuple(shape), own.dtype, dev, role)
        t = self._rows.get(key)
        if t is None:
            t = torch.empty(shape, dtype=own.dtype, device=dev, pin_memory=staged)
            self._rows[key] = t
        return t

    def exchange(self, own: Tensor, row_dim: int, top: bool, bottom: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """own: this rank's rows of an activation (any layout, rows along `row_dim`).  Returns (row above my first row,
        row below my last row): the upper neighbour's LAST row if `top`, the lower neighbour's FIRST row if `bottom`; None where
        there is no neighbour (grid border) or the row was not asked for.  Every active rank calls this with the same flags.
        The returned rows are the communicator's own buffers: consume (copy) them before the next exchange of the same shape."""
        self.exchanges += 1
        if not self.active:
            return None, None
        n = own.shape[row_dim]
        nccl = own.is_cuda and dist.get_backend(self.group) == "nccl"
        staged = own.is_cuda and not nccl                            # gloo dry runs of the GPU path go through (pinned) host rows
        send_last = top and self.lower is not None                   # my last row is the lower neighbour's top halo
        send_first = bottom and self.upper is not None               # my first row is the upper neighbour's bottom halo
        want_top = top and self.upper is not None
        want_bot = bottom and self.lower is not None
        if not (send_last or send_first or want_top or want_bot):
            return None, None
        main = side = None
        if nccl:
            main = torch.cuda.current_stream(own.device)
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=own.device)
            side = self._stream
            side.wait_stream(main)                                   # the producing layer
        ctx = torch.cuda.stream(side) if nccl else contextlib.nullcontext()
        recv_top = recv_bot = None
        from . import ops as _ops
        t_host = ev0 = None
        if _ops.PROFILE_ON:                                          # bench.py: measured us per exchange (events on the communication stream)
            if nccl:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(side)
            else:
                import time as _time
                t_host = _time.perf_counter()
        with ctx:
            ops_ = []

            def send(i: int, role: str, peer: int) -> None:
                row = own.select(row_dim, i).unsqueeze(row_dim)
                if staged:
                    dv = self._row(own, row_dim, role + "_dev", False)
                    dv.copy_(row)
                    t = self._row(own, row_dim, role, True)
                    t.copy_(dv)                                      # device -> pinned host, synchronous for the host
                else:
                    t = self._row(own, row_dim, role, False)
                    t.copy_(row)
                ops_.append(dist.P2POp(dist.isend, t, self._global(peer), self.group))
                self.bytes_sent += t.numel() * t.element_size()

            if send_last:
                send(n - 1, "send_last", self.lower)
            if send_first:
                send(0, "send_first", self.upper)
            if want_top:
                recv_top = self._row(own, row_dim, "recv_top", staged)
                ops_.append(dist.P2POp(dist.irecv, recv_top, self._global(self.upper), self.group))
            if want_bot:
                recv_bot = self._row(own, row_dim, "recv_bot", staged)
                ops_.append(dist.P2POp(dist.irecv, recv_bot, self._global(self.lower), self.group))
            for req in dist.batch_isend_irecv(ops_):
                req.wait()                                           # RCCL: a stream-side wait on the communication stream; gloo: host
            if staged:
                if recv_top is not None:
                    recv_top = self._row(own, row_dim, "recv_top_dev", False).copy_(recv_top)
                if recv_bot is not None:
                    recv_bot = self._row(own, row_dim, "recv_bot_dev", False).copy_(recv_bot)
        if t_host is not None:
            import time as _time
            _ops.PROFILE_HOST_MS.setdefault("halo_exchange", []).append((_time.perf_counter() - t_host) * 1e3)
        if nccl:
            ev = torch.cuda.Event(enable_timing=ev0 is not None)
            ev.record(side)
            if ev0 is not None:
                _ops.PROFILE.setdefault("halo_exchange", []).append((ev0, ev))
            main.wait_event(ev)                                      # the consuming layer waits for the rows, the host does not
            for t in (recv_top, recv_bot):
                if t is not None:
                    t.record_stream(main)
        return recv_top, recv_bot


HALO0 = 8     # rows of halo of a level-0 slab buffer; level L carries HALO0 >> L (8, 4, 2, 1)


def slab_window(a: int, b: int, h: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the level-0 buffer of a slab [a, b): HALO0 rows beyond each side, clipped at the grid border."""
    return max(0, a - HALO0), min(h, b + HALO0)


def slab_cost_regularization(layers, comm: HaloComm, vol: Tensor, a: int, b: int, h: int) -> Tensor:
    """CostRegNet on this rank's rows [a, b) of an h-row grid, WITHOUT copying activations.

    vol: [D][hi - lo][w][C] channels-last, the rows ``slab_window(a, b, h)`` of the normalised volume -- every row of it VALID (the
    plane sweep is per pixel: the rank computes / receives its 8 halo rows itself).  Every activation of level L lives in one dense
    buffer that covers the rows [a_L - H_L, b_L + H_L) with H_L = 8 >> L (clipped at the grid border), which makes every layer a plain
    dense call on whole buffers: a stride-1 layer keeps the rows, a stride-2 layer maps a level-L buffer onto exactly the level-(L+1)
    buffer (it starts at an even global row), a transposed layer maps it back and its skip tensor IS the level-L buffer.  Only the own
    rows (and, for the input volume, the halo) of a buffer are valid; what a layer needs beyond them -- ONE row -- is fetched from the
    neighbour and written over the garbage row in place (`HaloComm.exchange`, 11 per stage); garbage further out only ever feeds rows
    that are overwritten or discarded.  Redundant arithmetic: 16 rows per slab (1.25x at 64 rows), no `torch.cat`.
    Returns prob_pre for the rows [a, b): [D][b - a][w] (a view).  `layers`: conv / deconv / prob on dense channels-last tensors
    (product: `HipCostRegLayers`; the CPU tests plug in torch reference ops)."""
    n = b - a
    if not comm.active or n == 0:
        for _ in range(11):
            comm.exchanges += 1
        return vol.new_zeros((vol.shape[0], 0, vol.shape[2]))
    top = [HALO0 >> L if a > 0 else 0 for L in range(4)]            # halo rows above / below the own rows, per level
    bot = [HALO0 >> L if b < h else 0 for L in range(4)]
    own = [n >> L for L in range(4)]
    if vol.shape[1] != top[0] + n + bot[0]:
        raise ValueError(f"slab volume has {vol.shape[1]} rows, expected the window {slab_window(a, b, h)}")

    def refresh(buf: Tensor, L: int, row_dim: int, need_top: bool, need_bot: bool) -> None:
        """Overwrite the row just above / below the own rows of a level-L buffer with the neighbour's edge row."""
        t, m = top[L], own[L]
        view = buf.narrow(row_dim, t, m)
        r_top, r_bot = comm.exchange(view, row_dim, need_top, need_bot)
        if r_top is not None:
            buf.narrow(row_dim, t - 1, 1).copy_(r_top)
        if r_bot is not None:
            buf.narrow(row_dim, t + m, 1).copy_(r_bot)

    comm.exchanges += 1                                   # conv0's halo rows are valid already: no message (counted as a no-op)
    c0 = layers.conv("conv0", vol, 1)
    refresh(c0, 0, 1, True, False)                        # conv1 (stride 2) reads one row above
    c1 = layers.conv("conv1", c0, 2)
    refresh(c1, 1, 1, True, True)
    c2 = layers.conv("conv2", c1, 1)
    del c1
    refresh(c2, 1, 1, True, False)
    c3 = layers.conv("conv3", c2, 2)
    refresh(c3, 2, 1, True, True)
    c4 = layers.conv("conv4", c3, 1)
    del c3
    refresh(c4, 2, 1, True, False)
    c5 = layers.conv("conv5", c4, 2)
    refresh(c5, 3, 1, True, True)
    x = layers.conv("conv6", c5, 1)
    del c5
    refresh(x, 3, 1, False, True)                         # a transposed layer reads one coarse row below
    x = layers.deconv("conv7", x, c4, False)
    del c4
    refresh(x, 2, 1, False, True)
    x = layers.deconv("conv9", x, c2, False)
    del c2
    refresh(x, 1, 1, False, True)
    planar = bool(getattr(layers, "conv11_planar", False))
    x = layers.deconv("conv11", x, c0, planar)
    del c0
    refresh(x, 0, 2 if planar else 1, True, True)
    y = layers.prob(x)
    return y[:, top[0]:top[0] + n]


class HipCostRegLayers:
    """The product layer ops of `slab_cost_regularization`: the split-bf16 matrix-core kernels (csrc/conv3d_sbf.hip) exactly as
    CostRegNet._run_cl calls them, conv11 writing planar for the prob kernel."""
    conv11_planar = True

    def __init__(self, cost_reg):
        from . import ops
        self.ops = ops
        if not cost_reg.split_bf16_supported():
            raise RuntimeError("slab-parallel CostRegNet needs the split-bf16 kernels (base channels 8, CDS_CONV_EXACT unset)")
        cost_reg.want_slab_operands()
        self.p = cost_reg._packed.get(cost_reg, cost_reg._pack)
        self.cout = {name: getattr(cost_reg, name).conv.out_channels for name in
                     ("conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11")}

    def conv(self, name: str, x: Tensor, stride: int) -> Tensor:
        code = self.ops.SBF_PAIR if name == "conv0" else stride
        return self.ops.conv3d_sbf(x, self.p[name + ".ws"], self.p[name + ".b"], self.cout[name], stride=code)

    def deconv(self, name: str, x: Tensor, skip: Tensor, planar: bool) -> Tensor:
        return self.ops.deconv3d_sbf(x, self.p[name + ".ws"], self.p[name + ".b"], self.cout[name], skip=skip, out_planar=planar)

    def prob(self, x: Tensor) -> Tensor:
        return self.ops.conv3d_k3(x, self.p["prob.w"], None, relu=False)[0]
