"""Random-shape fuzz of the conv9 class-per-wave z-march kernel against the tiled split-bf16 kernel (GPU only).
usage: python scripts/fuzz_dzm.py [cases] [seed]"""
import os, sys, random
import torch
sys.path.insert(0, ".")
from cds_mvsnet_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
worst = 0.0
for case in range(n):
    D = rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 12, 25])
    H = rng.randint(1, 40)
    W = rng.randint(1, 90)
    nseg = rng.choice([0, 0, 1, 2, 3, 5])
    if nseg:
        os.environ["CDS_DZM_NSEG"] = str(nseg)
    else:
        os.environ.pop("CDS_DZM_NSEG", None)
    g = torch.Generator(device=dev).manual_seed(case)
    x = torch.randn(D, H, W, 32, device=dev, generator=g) * (1.0 + 3.0 * rng.random())
    use_skip, relu, use_bias = rng.random() < 0.7, rng.random() < 0.7, rng.random() < 0.8
    skip = torch.randn(2 * D, 2 * H, 2 * W, 16, device=dev, generator=g) if use_skip else None
    w = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) * 0.1
    b = torch.randn(16, device=dev, generator=g) if use_bias else None
    ref = ops.deconv3d_sbf(x, ops.split_pack_deconv3d(w), b, 16, relu=relu, skip=skip)
    got = ops.deconv3d_zm(x, ops.split_pack_deconv_cls(w), b, relu=relu, skip=skip)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    worst = max(worst, err)
    if err > 3e-6:
        print(f"case {case}: D{D} H{H} W{W} nseg {nseg} skip {use_skip} relu {relu} bias {use_bias}: relative max diff {err:.3e}  <-- FAIL", flush=True)
        sys.exit(1)
print(f"{n} cases, worst relative max diff {worst:.3e}")
