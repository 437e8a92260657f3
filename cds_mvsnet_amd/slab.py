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

The layer arithmetic itself is not in this file: `layers` is any object with conv / deconv / prob methods on dense channels-last
tensors (product: `HipCostRegLayers`, the split-bf16 matrix-core kernels; the CPU tests plug in torch reference ops to check the
slab bookkeeping and the exchanges against the unsharded network over gloo).
"""
from __future__ import annotations

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
    tests).  Counts what it moves."""

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

    def _global(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def exchange(self, own: Tensor, row_dim: int, top: bool, bottom: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """own: this rank's rows of an activation (any layout, rows along `row_dim`).  Returns (row above my first row,
        row below my last row): the upper neighbour's LAST row if `top`, the lower neighbour's FIRST row if `bottom`; None where
        there is no neighbour (grid border) or the row was not asked for.  Every active rank calls this with the same flags."""
        self.exchanges += 1
        if not self.active:
            return None, None
        n = own.shape[row_dim]
        ops_, recv_top, recv_bot = [], None, None
        staged = own.is_cuda and dist.get_backend(self.group) != "nccl"   # gloo dry runs of the GPU path go through the host
        keep = []

        def buf_like():
            shape = list(own.shape)
            shape[row_dim] = 1
            return torch.empty(shape, dtype=own.dtype, device="cpu" if staged else own.device)

        def out_row(i):
            t = own.select(row_dim, i).unsqueeze(row_dim).contiguous()
            return t.cpu() if staged else t

        if top and self.lower is not None:            # my last row is the lower neighbour's top halo
            t = out_row(n - 1)
            keep.append(t)
            ops_.append(dist.P2POp(dist.isend, t, self._global(self.lower), self.group))
            self.bytes_sent += t.numel() * t.element_size()
        if bottom and self.upper is not None:         # my first row is the upper neighbour's bottom halo
            t = out_row(0)
            keep.append(t)
            ops_.append(dist.P2POp(dist.isend, t, self._global(self.upper), self.group))
            self.bytes_sent += t.numel() * t.element_size()
        if top and self.upper is not None:
            recv_top = buf_like()
            ops_.append(dist.P2POp(dist.irecv, recv_top, self._global(self.upper), self.group))
        if bottom and self.lower is not None:
            recv_bot = buf_like()
            ops_.append(dist.P2POp(dist.irecv, recv_bot, self._global(self.lower), self.group))
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        if staged:
            recv_top = recv_top.to(own.device) if recv_top is not None else None
            recv_bot = recv_bot.to(own.device) if recv_bot is not None else None
        return recv_top, recv_bot


def _with_halo(own: Tensor, row_dim: int, above: List[Optional[Tensor]], below: List[Optional[Tensor]]) -> Tensor:
    """Dense tensor [above rows ..., own rows, below rows ...]; None entries are dropped."""
    parts = [t for t in above if t is not None] + [own] + [t for t in below if t is not None]
    return parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, dim=row_dim)


def slab_cost_regularization(layers, comm: HaloComm, vol: Tensor) -> Tensor:
    """CostRegNet on this rank's rows.  vol: [D][n][w][C] channels-last, the rank's own rows of the normalised volume.
    Returns prob_pre for the same rows, [D][n][w].  `layers`:
        conv(name, x [D][R][W][Cin], stride) -> [Do][Ro][Wo][Cout]            (BN folded, ReLU; zero padding 1)
        deconv(name, x [D][R][W][Cin], skip [2D][2R][2W][Cout]) -> same shape as skip, plus `planar` = True for a
                                                                  [Cout][2D][2R][2W] result (conv11 feeds the planar prob kernel)
        prob(x) -> [D][R][W]                                                   (x in the layout deconv('conv11') returned)
        prob_row_dim: the row dimension of that layout (1 channels-last, 2 planar)."""
    n = vol.shape[1]

    def zeros_row(t: Tensor, row_dim: int) -> Tensor:
        shape = list(t.shape)
        shape[row_dim] = 1
        return torch.zeros(shape, dtype=t.dtype, device=t.device)

    def conv_s1(name: str, own: Tensor) -> Tensor:
        top, bot = comm.exchange(own, 1, True, True)
        x = _with_halo(own, 1, [top], [bot])
        y = layers.conv(name, x, 1)
        t = 0 if top is None else 1
        return y[:, t:t + own.shape[1]]

    def conv_s2(name: str, own: Tensor) -> Tensor:
        top, _ = comm.exchange(own, 1, True, False)
        # the local tensor must start at an even global row: [zero row, row above, own rows]
        x = _with_halo(own, 1, [zeros_row(own, 1), top] if top is not None else [], [])
        y = layers.conv(name, x, 2)
        t = 0 if top is None else 1
        return y[:, t:t + own.shape[1] // 2]

    def deconv(name: str, own: Tensor, skip_own: Tensor, planar: bool = False) -> Tensor:
        _, bot = comm.exchange(own, 1, False, True)
        x = _with_halo(own, 1, [], [bot])
        skip = skip_own.contiguous()
        if bot is not None:     # two more (discarded) output rows: pad the skip tensor to the kernel's output shape
            pad = torch.zeros((skip.shape[0], 2, skip.shape[2], skip.shape[3]), dtype=skip.dtype, device=skip.device)
            skip = torch.cat((skip, pad), dim=1)
        y = layers.deconv(name, x, skip, planar)
        rows = 2 * own.shape[1]
        return y[:, :, :rows] if planar else y[:, :rows]

    if not comm.active or n == 0:
        # a rank without rows still takes part in nothing: the exchanges are between active neighbours only
        for _ in range(11):
            comm.exchanges += 1
        return vol.new_zeros((vol.shape[0], 0, vol.shape[2]))
    c0 = conv_s1("conv0", vol)
    c1 = conv_s2("conv1", c0)
    c2 = conv_s1("conv2", c1)
    del c1
    c3 = conv_s2("conv3", c2)
    c4 = conv_s1("conv4", c3)
    del c3
    c5 = conv_s2("conv5", c4)
    x = conv_s1("conv6", c5)
    del c5
    x = deconv("conv7", x, c4)
    del c4
    x = deconv("conv9", x, c2)
    del c2
    planar = bool(getattr(layers, "conv11_planar", False))
    x = deconv("conv11", x, c0, planar)
    del c0
    rd = 2 if planar else 1
    top, bot = comm.exchange(x, rd, True, True)
    y = layers.prob(_with_halo(x, rd, [top], [bot]))
    t = 0 if top is None else 1
    return y[:, t:t + n].contiguous()


class HipCostRegLayers:
    """The product layer ops of `slab_cost_regularization`: the split-bf16 matrix-core kernels (csrc/conv3d_sbf.hip) exactly as
    CostRegNet._run_cl calls them, conv11 writing planar for the prob kernel."""
    conv11_planar = True

    def __init__(self, cost_reg):
        from . import ops
        self.ops = ops
        if not cost_reg.split_bf16_supported():
            raise RuntimeError("slab-parallel CostRegNet needs the split-bf16 kernels (base channels 8, CDS_CONV_EXACT unset)")
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
