"""Random-shape fuzz of the fused conv11 + residual + prob kernel against the two separate product kernels (GPU only).
usage: python scripts/fuzz_dpz.py [cases] [seed]"""
import os, sys, random
import torch
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
worst = 0.0
for case in range(n):
    D = rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 12, 17])
    H = rng.randint(1, 45)
    W = rng.randint(1, 130)
    nseg = rng.choice([0, 0, 1, 2, 3, 5])
    if nseg:
        os.environ["CDS_DPZ_NSEG"] = str(nseg)
    else:
        os.environ.pop("CDS_DPZ_NSEG", None)
    g = torch.Generator(device=dev).manual_seed(case)
    x = torch.randn(D, H, W, 16, device=dev, generator=g) * (1.0 + 3.0 * rng.random())
    skip = torch.randn(2 * D, 2 * H, 2 * W, 8, device=dev, generator=g)
    w11 = torch.randn(16, 8, 3, 3, 3, device=dev, generator=g) * 0.15
    b = torch.randn(8, device=dev, generator=g)
    wp = torch.randn(1, 8, 3, 3, 3, device=dev, generator=g) * 0.2
    y = ops.deconv3d_sbf(x, ops.split_pack_deconv3d(w11), b, 8, skip=skip, out_planar=True)
    ref = ops.conv3d_k3(y, wp.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous(), None, relu=False)[0]
    got = ops.deconv_prob_zm(x, ops.split_pack_deconv_prob(w11), b, skip, ops.pack_prob_table(wp))
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    worst = max(worst, err)
    if err > 3e-6:
        print(f"case {case}: D{D} H{H} W{W} nseg {nseg}: relative max diff {err:.3e}  <-- FAIL", flush=True)
        sys.exit(1)
print(f"{n} cases, worst relative max diff {worst:.3e}")
