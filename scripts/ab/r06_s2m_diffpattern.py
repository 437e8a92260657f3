"""What do the victim's (K1) differing values look like while the stride-2 matrix-core kernel runs beside it?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops, synth, geometry
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
V, C, D, h, w = 4, 32, 48, 296, 400
feats = synth.make_pair_features(V, C, h, w, seed=1)
cams = synth.stage_cameras(V + 1, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
mats = geometry.warp_matrices(cams[0]).to(dev)
victim = lambda: ops.warp_entropy(ref, src, mats, hyp)
want = victim().clone()
torch.cuda.synchronize()
cin, cout, N, H, W = 16, 32, 8, 296, 400
xcl = torch.randn(N, H, W, cin, generator=g).to(dev)
wh, winv = ops.split_pack_dynconv([(torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)], f16=True)
outbuf = None
def aggr():
    return ops.conv2d_k3s2_cl(xcl, None, cout, None, wsplit=wh, w_inv_scale=winv, x_bound=(H * W) ** 0.5)
a_want = aggr().clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(3):
    with torch.cuda.stream(sb):
        aouts = [aggr() for _ in range(300)]
    with torch.cuda.stream(sa):
        outs = [victim() for _ in range(6)]
    torch.cuda.synchronize()
    abad = sum(int(not torch.equal(o, a_want)) for o in aouts)
    print(f"rep {rep}: aggressor outputs differing from isolation: {abad} / {len(aouts)}")
    for i, o in enumerate(outs):
        d = (o != want)
        n = int(d.sum())
        if n == 0:
            print(f"  victim {i}: identical"); continue
        idx = d.nonzero()
        xs = idx[:, 2]
        err = (o - want)[d].abs()
        print(f"  victim {i}: {n} of {o.numel()} differ; views {sorted(set(idx[:, 0].tolist()))}; x mod 64 histogram (16-bins) "
              f"{[int(((xs % 64) // 16 == k).sum()) for k in range(4)]}; max |err| {float(err.max()):.3e} median {float(err.median()):.3e}; "
              f"nan {int(torch.isnan(o).sum())}; first {idx[:4].tolist()}")
