"""Per-rank compute of the slab modes, measured on ONE GPU: K1 + visibility CNN + K3 on a row window and CostRegNet + soft-argmin on a
slab of h / N rows (+ the halo rows the layers see), N = 1, 2, 4, 8, at the M1 shape.  No communication: this is the compute term of the
scaling prediction in DESIGN.md section 6."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, ops, geometry
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 0).eval().to(dev)
h, w, D, C, n = bench.WORKLOADS["M1"]
_, cams, hyp, dfe = bench.make_workload("M1", 0, dev)
hyp_d = hyp[0].to(dev).contiguous()
V = n - 1
ref = torch.stack([f["ref"][0][0] for f in dfe]).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].contiguous()) for f in dfe])
ref_nc = torch.stack([f["ref"][2][0, 0] for f in dfe]).contiguous()
mats = geometry.warp_matrices(cams[0])
cr = model.cost_regularization[2]
def t(fn, k=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
with torch.no_grad():
    for N in (1, 2, 4, 8):
        rows = h // N
        a, b = (h - rows) // 2 // 8 * 8, (h - rows) // 2 // 8 * 8 + rows          # an interior slab
        a1, b1 = max(0, a - 8), min(h, b + 8)
        def warp():
            ent = ops.warp_entropy(ref[:, :, a1:b1].contiguous(), src, mats, hyp_d[:, a1:b1].contiguous(), window=(h, a1))
            vis = model.stage_net.visibility(ent, ref_nc[:, a1:b1].contiguous(), 2)[:, a - a1:b - a1].contiguous()
            return ops.warp_aggregate(ref[:, :, a:b].contiguous(), src, vis, mats, hyp_d[:, a:b].contiguous(), channels_last=True, window=(h, a))[0]
        vol = warp()
        t_w = t(warp)
        halo = 0 if N == 1 else 2
        volh = torch.randn(D, rows + (8 if halo else 0), w, C, device=dev)             # slab + a row group standing in for the halo rows
        t_c = t(lambda: cr(volh, channels_last=True))
        pre = cr(volh, channels_last=True)
        t_s = t(lambda: ops.softargmin_conf(pre[:, :rows].contiguous(), hyp_d[:, a:b].contiguous()))
        print(f"N={N}: rows {rows}: K1+vis+K3 on the window {t_w:.3f} ms | CostRegNet on {volh.shape[1]} rows {t_c:.3f} ms | soft-argmin {t_s:.3f} ms | sum {t_w + t_c + t_s:.3f} ms")

# the slab CostRegNet as the product runs it (level buffers with 8 >> L halo rows, exchanged rows written in place), with a stand-in
# communicator that hands back zero rows: the per-rank compute INCLUDING the halo bookkeeping
from cds_mvsnet_amd.slab import HipCostRegLayers, slab_cost_regularization, slab_window
class FakeComm:
    active = True
    exchanges = 0
    bytes_sent = 0
    def __init__(self, top, bottom): self.t, self.b = top, bottom
    def exchange(self, own, row_dim, top, bottom):
        self.exchanges += 1
        shape = list(own.shape); shape[row_dim] = 1
        z = torch.zeros(shape, dtype=own.dtype, device=own.device)
        return (z if (top and self.t) else None), (z if (bottom and self.b) else None)
with torch.no_grad():
    layers = HipCostRegLayers(cr)
    for N in (1, 2, 4, 8):
        rows = h // N
        a = (h - rows) // 2 // 8 * 8; b = a + rows
        lo, hi = slab_window(a, b, h)
        vol = torch.randn(D, hi - lo, w, C, device=dev)
        comm = FakeComm(a > 0, b < h)
        t_c = t(lambda: slab_cost_regularization(layers, comm, vol, a, b, h))
        print(f"N={N}: slab_cost_regularization on rows [{a},{b}) (buffer {hi - lo} rows, halo rows written in place): {t_c:.3f} ms")
