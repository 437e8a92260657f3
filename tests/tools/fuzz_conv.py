"""Random-shape fuzz of the 3D / 2D convolution entry points against PyTorch fp32 on the CPU (tile edges, unaligned widths,
every kernel family the dispatchers can pick).  Usage: fuzz_conv.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from cds_mvsnet_amd import ops
dev = torch.device("cuda:0")


def case3d(rng):
    cin = rng.choice([3, 4, 8, 12, 16, 32, 64]); cout = rng.choice([1, 8, 16, 32, 48, 64])
    D = rng.randint(1, 9); H = rng.randint(1, 20); W = rng.choice([rng.randint(1, 40), rng.choice([32, 36, 40, 48, 50, 64, 68, 72, 100, 132])])
    kind = rng.choice(["conv1", "conv2", "deconv"])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = torch.randn(cin, D, H, W, generator=g); b = torch.randn(cout, generator=g)
    if kind == "deconv":
        if cout % 8: cout = 8; b = torch.randn(cout, generator=g)
        w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.1
        ref = F.relu(F.conv_transpose3d(x[None], w, b, stride=2, padding=1, output_padding=1))[0]
        skip = torch.randn(ref.shape, generator=g)
        wpk = w.permute(0, 2, 3, 4, 1).reshape(cin, 27, cout).contiguous().to(dev)
        out = ops.deconv3d_k3s2(x.to(dev), wpk, b.to(dev), relu=True, skip=skip.to(dev)).cpu()
    else:
        s = 1 if kind == "conv1" else 2
        if (cout != 1 and cout % 8) or (cout == 1 and s == 2):   # Cout = 1 exists for stride 1 only (the prob layer)
            cout = 8; b = torch.randn(cout, generator=g)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
        ref = F.relu(F.conv3d(x[None], w, b, stride=s, padding=1))[0]
        skip = torch.randn(ref.shape, generator=g)
        wpk = w.permute(1, 2, 3, 4, 0).reshape(cin, 27, cout).contiguous().to(dev)
        out = ops.conv3d_k3(x.to(dev), wpk, b.to(dev), stride=s, relu=True, skip=skip.to(dev)).cpu()
    err = float((out - (ref + skip)).abs().max()) / max(1.0, float(ref.abs().max()))
    return f"{kind:6s} {cin:2d}->{cout:2d} D={D} H={H:2d} W={W:3d}", err


def case2d(rng):
    from cds_mvsnet_amd.model import _pack2d
    cin = rng.choice([2, 3, 8, 16, 24, 32, 48]); cout = rng.choice([1, 8, 11, 16, 19, 32, 35])
    k, s = rng.choice([(1, 1), (3, 1), (3, 2), (5, 1), (7, 1), (11, 1)])
    n = rng.randint(1, 3); H = rng.randint(1, 40); W = rng.choice([rng.randint(1, 70), rng.choice([4, 8, 36, 64, 68, 72, 100, 132, 200])])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = torch.randn(n, cin, H, W, generator=g); w = torch.randn(cout, cin, k, k, generator=g) * 0.1; b = torch.randn(cout, generator=g)
    pad = (k - 1) // 2
    ref = F.conv2d(x, w, b, stride=s, padding=pad)
    out = ops.conv2d(x.to(dev), _pack2d(w).to(dev), b.to(dev), cout, k, s, pad).cpu()
    err = float((out - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    return f"conv2d {cin:2d}->{cout:2d} k={k:2d} s={s} n={n} H={H:2d} W={W:3d}", err


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    worst = 0.0
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
        name, err = (case3d if i % 2 == 0 else case2d)(rng)
        worst = max(worst, err)
        print(f"{name}: {err:.2e}{'' if err < 2e-5 else '  <-- FAIL'}")
    print("worst relative error", worst)
