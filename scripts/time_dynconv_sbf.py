"""DynamicConv branch convolutions at the 1600x1184 cascade shapes (8 images): matrix-core kernel vs the VALU kernels."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
N = 8
for name, cin, cout, ks, H, W in (("conv01", 8, 8, (3, 5, 7), 1184, 1600), ("conv10", 16, 16, (3, 5), 592, 800),
                                  ("conv20", 32, 32, (1, 3), 296, 400), ("out2", 16, 16, (1, 3), 592, 800), ("out3", 8, 8, (1, 3), 1184, 1600)):
    co3 = cout + 3
    x = torch.randn(N, cin, H, W, device=dev)
    aff = torch.stack((torch.ones(N, cin), torch.zeros(N, cin), torch.full((N, cin), 0.1)), -1).to(dev).contiguous()
    ws = [torch.randn(co3, cin, k, k, device=dev) / (cin * k * k) ** 0.5 for k in ks]
    wsp = ops.split_pack_dynconv(ws)
    pad = (-co3) % 8
    wpk = [F.pad(w.permute(1, 2, 3, 0).reshape(cin, k * k, co3), (0, pad)).contiguous() for w, k in zip(ws, ks)]
    out = torch.empty(len(ks), N, co3, H, W, device=dev)
    def valu():
        for i, k in enumerate(ks):
            ops.conv2d(x, wpk[i], None, co3, k, 1, (k - 1) // 2, out=out[i], in_affine=aff)
    tv = t(valu)
    o1 = out.clone()
    ts = t(lambda: ops.dynconv_branches_sbf(x, wsp, None, co3, ks, out=out, in_affine=aff))
    fl = 2.0 * sum(k * k for k in ks) * cin * co3 * N * H * W
    print(f"{name}: VALU {tv:8.1f} us ({fl / tv / 1e6:6.1f} TF)   matrix cores {ts:8.1f} us ({fl / ts / 1e6:6.1f} TF-equivalent)   max |diff| {(out - o1).abs().max().item():.2e}")
