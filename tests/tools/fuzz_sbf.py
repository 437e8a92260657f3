"""Random-shape fuzz of the split-bf16 matrix-core kernels (csrc/conv3d_sbf.hip, conv2d_sbf.hip) against float64 PyTorch:
3D convolutions (stride 1 / 2, 1-4 output blocks, the voxel-pair form incl. the z-marching Cin = 8 kernel, with / without
residual), transposed convolutions (Cout 8 warp-specialised / 16 / 32, planar output), the fused DynamicConv against the
branch + blend kernels, the visibility layers.  Reports the worst error relative to max|reference| (fp32-class: ~1e-6).
Usage: fuzz_sbf.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from cds_mvsnet_amd import ops
dev = torch.device("cuda:0")


def rel(a, b):
    return (a.double() - b).abs().max().item() / max(1e-9, b.abs().max().item())


def conv_case(rng, g):
    kind = rng.choice(["s1", "s2", "pair"])
    if kind == "pair":
        cin, cout, stride, code = rng.choice([8, 8, 16, 32]), 8, 1, ops.SBF_PAIR
    else:
        cin, cout = rng.choice([(8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
        stride = 1 if kind == "s1" else 2
        code = stride
    D, H, W = rng.randint(1, 40), rng.randint(1, 30), rng.randint(2, 90)
    x = torch.randn(cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    want = F.conv3d(x.double().unsqueeze(0), w.double(), b.double(), stride=stride, padding=1)[0]
    use_skip = rng.random() < 0.4
    skip = torch.randn(want.shape, generator=g) if use_skip else None
    ws = ops.split_pack_conv3d_pair(w.to(dev)) if kind == "pair" else ops.split_pack_conv3d(w.to(dev))
    got = ops.conv3d_sbf(x.permute(1, 2, 3, 0).contiguous().to(dev), ws, b.to(dev), cout, stride=code, relu=use_skip,
                         skip=skip.permute(1, 2, 3, 0).contiguous().to(dev) if use_skip else None).cpu().permute(3, 0, 1, 2)
    ref = skip.double() + want.clamp_min(0) if use_skip else want
    return f"conv3d {kind} {cin}->{cout} D={D} H={H} W={W} skip={int(use_skip)}", rel(got, ref)


def deconv_case(rng, g):
    cin, cout = rng.choice([(16, 8), (8, 8), (32, 16), (64, 32), (32, 8)])
    D, H, W = rng.randint(1, 12), rng.randint(1, 14), rng.randint(1, 50)
    x = torch.randn(cin, D, H, W, generator=g)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) / (27 * cin / 8) ** 0.5
    b = torch.randn(cout, generator=g)
    want = F.conv_transpose3d(x.double().unsqueeze(0), w.double(), b.double(), stride=2, padding=1, output_padding=1)[0]
    skip = torch.randn(want.shape, generator=g)
    planar = rng.random() < 0.5
    got = ops.deconv3d_sbf(x.permute(1, 2, 3, 0).contiguous().to(dev), ops.split_pack_deconv3d(w.to(dev)), b.to(dev), cout, relu=True,
                           skip=skip.permute(1, 2, 3, 0).contiguous().to(dev), out_planar=planar).cpu()
    got = got if planar else got.permute(3, 0, 1, 2)
    return f"deconv3d {cin}->{cout} D={D} H={H} W={W} planar={int(planar)}", rel(got, skip.double() + want.clamp_min(0))


def dyn_case(rng, g):
    cin, cout, ks = rng.choice([(8, 8, (3, 5, 7)), (16, 16, (3, 5)), (32, 32, (1, 3)), (8, 8, (1, 3)), (16, 16, (1, 3))])
    N, H, W = rng.randint(1, 4), rng.randint(1, 40), 4 * rng.randint(1, 30)
    K, co3 = len(ks), cout + 3
    x = torch.randn(N, cin, H, W, generator=g)
    ws = ops.split_pack_dynconv([(torch.randn(co3, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev) for k in ks])
    bs = torch.randn(K, co3, generator=g).to(dev)
    w1, b1, w2 = torch.randn(4, K, generator=g).to(dev), torch.randn(4, generator=g).to(dev), torch.randn(K, 4, generator=g).to(dev)
    epi = torch.tensor([[rng.uniform(-2 * W, 3 * W), rng.uniform(-2 * H, 3 * H)] for _ in range(N)], dtype=torch.float32)
    T = rng.choice([1.0, 0.1, 0.01])
    br = ops.dynconv_branches_sbf(x.to(dev), ws, bs, co3, ks)
    o2, n2, s2, a2 = ops.dynconv_blend(br, w1, b1, w2, epi, T, 1, stats_slope=0.1)
    o1, n1, s1, a1 = ops.dynconv_fused_sbf(x.to(dev), ws, bs, cout, ks, w1, b1, w2, epi, T, 0.1)
    bad = 0.0 if (torch.equal(o1, o2) and torch.equal(n1, n2) and torch.allclose(a1, a2, rtol=1e-6, atol=1e-7)) else 1.0
    return f"dynconv fused {cin}->{cout} ks={ks} N={N} H={H} W={W} T={T}", bad


def vis_case(rng, g):
    N, H, W = rng.randint(1, 5), rng.randint(1, 40), 4 * rng.randint(1, 40)
    x = torch.randn(N, 16, H, W, generator=g).clamp_min(0)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    b = torch.randn(16, generator=g) * 0.2
    hw, hb = torch.randn(16, generator=g) * 0.3, torch.randn(1, generator=g)
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1).clamp_min(0)
    want = torch.sigmoid((y * hw.double().view(1, 16, 1, 1)).sum(1) + hb.double())
    got = ops.conv2d_k3_relu_sbf(x.to(dev), ops.split_pack_dynconv([w.to(dev)]), b.to(dev), head_w=hw.to(dev), head_b=hb.to(dev)).cpu()
    return f"visibility layer N={N} H={H} W={W}", (got.double() - want).abs().max().item()


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    worst = {}
    for i in range(n):
        g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
        fn = rng.choice([conv_case, conv_case, deconv_case, dyn_case, vis_case])
        name, err = fn(rng, g)
        fam = name.split()[0] + " " + name.split()[1]
        worst[fam] = max(worst.get(fam, 0.0), err)
        flag = "  <-- FAIL" if err > 2e-5 else ""
        print(f"{name}: {err:.2e}{flag}")
    print("worst per family:", {k: f"{v:.2e}" for k, v in worst.items()})
