"""Stress: every dynconv_cl configuration at sizes with thousands of workgroups, several repetitions, bit-compared with the
unfused planar path (branches kernel + blend kernel)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tot = 0
for (c, ks, N, H, W) in ((8, (1, 3), 8, 1184, 1600), (8, (3, 5, 7), 8, 592, 800), (32, (1, 3), 8, 296, 400), (16, (3, 5), 8, 592, 800), (16, (1, 3), 8, 592, 800),
                         (32, (1, 3), 8, 288, 384), (8, (3, 5, 7), 2, 256, 512)):
    K, co3 = len(ks), c + 3
    x = torch.randn(N, c, H, W, generator=g).to(dev)
    xcl = x.permute(0, 2, 3, 1).contiguous()
    aff = torch.stack((0.5 + torch.rand(N, c, generator=g), 0.3 * torch.randn(N, c, generator=g), torch.full((N, c), 0.1)), -1).to(dev).contiguous()
    wsp = ops.split_pack_dynconv([(torch.randn(co3, c, k, k, generator=g) / (c * k * k) ** 0.5).to(dev) for k in ks])
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[W * 0.3 + 5.0 * n, -H * 1.7 - n] for n in range(N)], dtype=torch.float32)
    for T in (1.0, 0.01):
        br = ops.dynconv_branches_sbf(x, wsp, None, co3, ks, in_affine=aff)
        ref = ops.dynconv_blend(br, w1, b1, w2, epi, T, 1, stats_slope=0.1)[:2]
        del br
        res = []
        for rep in range(reps):
            cl = ops.dynconv_cl(xcl, wsp, None, ks, w1, b1, w2, epi, T, 0.1, in_affine=aff)[:2]
            bad = cl[0].permute(0, 3, 1, 2) != ref[0]
            res.append((int(bad.sum()), int((cl[1] != ref[1]).sum())))
            if res[-1][0]:
                idx = bad.nonzero()
                print("     bad channels:", sorted(collections.Counter(int(i[1]) for i in idx).items()))
        tot += sum(a + b for a, b in res)
        print(c, ks, N, H, W, "T", T, "bad (out elements, norm_curv pixels) per rep:", res)
print("TOTAL BAD", tot)
