"""Overlap on vs off: how do the outputs of stage 1's K1 (warp_entropy) differ?"""
import os, sys, torch, functools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cds_mvsnet_amd.model as cm
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, ops
dev = torch.device("cuda")
H, W, N = 1184, 1600, 5
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=4).to(dev)
cams = synth.make_cameras(N, H, W, refine=False, seed=4)
dv = synth.make_depth_values()
keep = collections.defaultdict(list)
for n in ("warp_entropy", "warp_aggregate", "depth_planes", "pair_mean"):
    f = getattr(ops, n)
    def g(*a, _f=f, _n=n, **k):
        r = _f(*a, **k)
        keep[_n].append((a, r))
        return r
    setattr(ops, n, g)
def run(ov):
    keep.clear()
    cm.OVERLAP_STAGE1 = ov
    with torch.no_grad():
        model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize()
    return {k: list(v) for k, v in keep.items()}
run(False)
base = run(False)
for rep in range(2):
    t = run(True)
    (a0, e0), (a1, e1) = base["warp_entropy"][0], t["warp_entropy"][0]
    print("rep", rep, "K1 inputs equal:", [torch.equal(x, y) for x, y in zip(a0[:4], a1[:4]) if isinstance(x, torch.Tensor)])
    d = (e0 - e1).abs()
    nz = d > 0
    print("   entropy [V,h,w]", tuple(e0.shape), "differing", int(nz.sum()), "max", float(d.max()), "mean over differing", float(d[nz].mean()) if nz.any() else 0)
    idx = nz.nonzero()[:16].tolist()
    print("   first differing (v, y, x):", idx, " (y%8, x%32):", [(i[1] % 8, i[2] % 32) for i in idx])
    import collections as C
    print("   by view:", C.Counter(int(i[0]) for i in nz.nonzero()[:100000]))
    hyp0, hyp1 = base["depth_planes"][0][1], t["depth_planes"][0][1]
    print("   hypotheses equal:", torch.equal(hyp0, hyp1))
