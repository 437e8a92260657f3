"""Which kernel, running on another stream, perturbs K1 (warp_entropy)?  Victim on stream A, aggressor on stream B, the victim's
output compared bit for bit with its output in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops, synth, geometry
from cds_mvsnet_amd.model import _pack2d
dev = torch.device("cuda")
if os.environ.get("AGGR_FILL_TAPS"):
    # experiment: the padded taps (beyond k^2) of the weight operand are 1.0 in every row instead of 0 (the kernel variant zeroes the data)
    def _pack_taps(ws):
        Co3, Cin = ws[0].shape[:2]
        rounds, nblk = Cin // 8, (Co3 + 15) // 16
        parts = []
        for w in ws:
            k = w.shape[-1]
            nks = (k * k + 3) // 4
            taps = torch.ones((nblk * 16, rounds, nks * 4, 8), dtype=torch.float32, device=w.device)
            taps[:, :, :k * k] = 0.0
            taps[:Co3, :, :k * k] = w.detach().float().reshape(Co3, rounds, 8, k * k).permute(0, 1, 3, 2)
            if os.environ["AGGR_FILL_TAPS"] == "rows":
                taps[Co3:, :, :k * k] = taps[:1, :, :k * k]
            parts.append(taps.reshape(nblk, 16, rounds, nks, 4, 8).permute(2, 3, 0, 4, 1, 5).reshape(rounds, nks, nblk, 64, 8))
        return ops._split3(torch.cat(parts, dim=1))
    ops.split_pack_dynconv = _pack_taps
if os.environ.get("AGGR_FILL_PAD"):
    # experiment: no all-zero rows in the weight (A) operand: the padded output channels repeat channel 0
    _orig_pack = ops.split_pack_dynconv
    def _pack_filled(ws):
        Co3 = ws[0].shape[0]
        pad = (-Co3) % 16
        mode = os.environ["AGGR_FILL_PAD"]
        if mode == "rows":
            ws2 = [torch.cat([w] + [w[:1]] * pad, 0) for w in ws]
        else:   # "quad": only make the LAST row of the block nonzero (does one nonzero row per 4-row group suffice?)
            ws2 = [torch.cat([w, torch.zeros_like(w[:1]).repeat(pad - 1, 1, 1, 1), w[:1]], 0) if pad > 1 else torch.cat([w, w[:1]], 0) for w in ws]
        out = _orig_pack(ws2)
        return out
    ops.split_pack_dynconv = _pack_filled
g = torch.Generator().manual_seed(0)
# victim: stage-1 K1 of the 1600x1184 cascade
V, C, D, h, w = 4, 32, 48, 296, 400
feats = synth.make_pair_features(V, C, h, w, seed=1)
cams = synth.stage_cameras(V + 1, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
mats = geometry.warp_matrices(cams[0])
victim = lambda: ops.warp_entropy(ref, src, mats, hyp)
want = victim().clone()
torch.cuda.synchronize()

def dyn_setup(c, ks, N, H, W, cl):
    K, co3 = len(ks), c + 3
    x = torch.randn(N, c, H, W, generator=g).to(dev)
    xcl = x.permute(0, 2, 3, 1).contiguous()
    aff = torch.stack((0.5 + torch.rand(N, c, generator=g), 0.3 * torch.randn(N, c, generator=g), torch.full((N, c), 0.1)), -1).to(dev).contiguous()
    wsp = ops.split_pack_dynconv([(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks])
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    if cl:
        return lambda: ops.dynconv_cl(xcl, wsp, None, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff)
    return lambda: ops.dynconv_fused_sbf(x, wsp, None, c, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff)

N = 8
aggr = {}
for name, c, ks, H, W in (("conv01", 8, (3, 5, 7), 592, 800), ("conv10", 16, (3, 5), 592, 800), ("conv20", 32, (1, 3), 296, 400), ("out2", 16, (1, 3), 592, 800),
                          ("out3", 8, (1, 3), 592, 800)):
    aggr["cl " + name] = dyn_setup(c, ks, N, H, W, True)
    aggr["planar " + name] = dyn_setup(c, ks, N, H, W, False)
# round 6: the split-f16 kernels (v_mfma_f32_16x16x32_f16, same operand roles as their split-bf16 forms) as aggressors
def dyn_f16_setup(c, ks, N, H, W):
    K, co3 = len(ks), c + 3
    xcl = torch.randn(N, H, W, c, generator=g).to(dev)
    aff = torch.stack((0.5 + torch.rand(N, c, generator=g), 0.3 * torch.randn(N, c, generator=g), torch.full((N, c), 0.1)), -1).to(dev).contiguous()
    wh, winv = ops.split_pack_dynconv([(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks], f16=True)
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    return lambda: ops.dynconv_cl(xcl, wh, None, ks, w1, b1, w2, epi, 0.01, 0.1, in_affine=aff, x_bound=(H * W) ** 0.5, w_inv_scale=winv)
for name, c, ks, H, W in (("conv01", 8, (3, 5, 7), 592, 800), ("conv10", 16, (3, 5), 592, 800), ("conv20", 32, (1, 3), 296, 400),
                          ("out2", 16, (1, 3), 592, 800), ("out3", 8, (1, 3), 592, 800)):
    aggr["f16 cl " + name] = dyn_f16_setup(c, ks, N, H, W)
def conv3d_f16_setup(cin, cout, code, D, H, W):
    x = torch.randn(D, H, W, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).to(dev)
    wh, winv = (ops.split_pack_conv3d_pair if code == ops.SBF_PAIR else ops.split_pack_conv3d)(w, f16=True)
    b = torch.randn(cout, generator=g).to(dev)
    bound = x.abs().amax().reshape(1)
    return lambda: ops.conv3d_sbf(x, wh, b, cout, stride=code, in_bound=bound, w_inv_scale=winv)
def down_f16_setup(cin, cout, N, H, W):
    xcl = torch.randn(N, H, W, cin, generator=g).to(dev)
    aff = torch.stack((0.5 + torch.rand(N, cin, generator=g), 0.3 * torch.randn(N, cin, generator=g), torch.full((N, cin), 0.1)), -1).to(dev).contiguous()
    wh, winv = ops.split_pack_dynconv([(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)], f16=True)
    return lambda: ops.conv2d_k3s2_cl(xcl, None, cout, aff, wsplit=wh, w_inv_scale=winv, x_bound=(H * W) ** 0.5)
aggr["f16 cl downsample1"] = down_f16_setup(8, 16, N, 592, 800)
aggr["f16 cl downsample2"] = down_f16_setup(16, 32, N, 296, 400)
aggr["f16 conv3d 8->8 pair"] = conv3d_f16_setup(8, 8, ops.SBF_PAIR, 48, 296, 400)
aggr["f16 conv3d 16->16"] = conv3d_f16_setup(16, 16, 1, 24, 148, 200)
aggr["f16 conv3d 32->32"] = conv3d_f16_setup(32, 32, 1, 24, 148, 200)
# round 6: the weight-gradient kernels (fp32 matrix pipe, v_mfma_f32_16x16x4_f32): they run on the side stream of the training step
from cds_mvsnet_amd import train_ops, train2d_ops
def wgrad2d_setup(Co, Cin, H, W, k):
    x, gg = torch.randn(8, Cin, H, W, generator=g).to(dev), torch.randn(8, Co, H, W, generator=g).to(dev)
    return lambda: train2d_ops.conv2d_wgrad(gg, x, k, 1, (k - 1) // 2)
def wgrad3d_setup(Ca, Cb, D, h, w, S):
    x, gg = torch.randn(1, Cb, D, h, w, generator=g).to(dev), torch.randn(1, Ca, D // S, h // S, w // S, generator=g).to(dev)
    return lambda: train_ops.conv3d_wgrad(gg, x, S)
aggr["wgrad2d k3 16->19"] = wgrad2d_setup(19, 16, 144, 192, 3)
aggr["wgrad2d k5 8->11"] = wgrad2d_setup(11, 8, 288, 384, 5)
aggr["wgrad2d k7 8->11"] = wgrad2d_setup(11, 8, 288, 384, 7)
aggr["wgrad3d s1 8->8"] = wgrad3d_setup(8, 8, 8, 288, 384, 1)
aggr["wgrad3d s2 8->16"] = wgrad3d_setup(16, 8, 32, 144, 192, 2)
xa, xb = torch.randn(N, 296, 400, 32, generator=g).to(dev), torch.randn(N, 592, 800, 16, generator=g).to(dev)
wt = torch.randn(48, 16, generator=g).to(dev)
aggr["cl fpn inner1"] = lambda: ops.conv2d_fpn_cl(xa, xb, wt, 16, None, None, 0.1)
w9 = torch.randn(9, 16, 32, generator=g).to(dev)
aggr["cl downsample2"] = lambda: ops.conv2d_k3s2_cl(xb, w9, 32, None)
aggr["cl instnorm stats"] = lambda: ops.instnorm_stats_cl(xb, 0.1)
st = ops.instnorm_stats_cl(xb, 0.1)[0]
aggr["cl instnorm apply"] = lambda: ops.instnorm_apply_cl(xb, st, ops.ACT_TANH, 4, cl_from=4)
aggr["victim itself"] = victim
big = torch.randn(64 * 1024 * 1024, device=dev)
aggr["torch copy 256 MB"] = lambda: big.clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
only = os.environ.get("AGGR_ONLY")
for name, fn in aggr.items():
    if only and not name.startswith(only):
        continue
    fn(); torch.cuda.synchronize()
    nag = 6
    if os.environ.get("AGGR_COVER"):      # as many aggressor launches as cover the 12 victim launches in time
        def _ms(f, n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        nag = max(6, min(2000, int(12 * _ms(victim, 6) / max(_ms(fn, 6), 1e-3)) + 1))
    bad = tot = 0
    for rep in range(int(os.environ.get("AGGR_REPS", "3"))):
        with torch.cuda.stream(sb):
            for _ in range(nag):
                fn()
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(12):
                outs.append(victim())
        torch.cuda.synchronize()
        for o in outs:
            tot += 1
            bad += int(not torch.equal(o, want))
    print(f"aggressor {name:22s}: victim outputs differing from isolation: {bad} / {tot}")
