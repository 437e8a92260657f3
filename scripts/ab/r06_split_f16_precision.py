import torch, math
torch.manual_seed(0)
def split_bf16(x):
    a1 = x.to(torch.bfloat16).to(torch.float32); r = x - a1
    a2 = r.to(torch.bfloat16).to(torch.float32); a3 = (r - a2)   # exact residual (fits bf16? assume stored as bf16)
    a3 = a3.to(torch.bfloat16).to(torch.float32)
    return a1, a2, a3
def split_f16(x, scale):
    xs = x * scale
    h = xs.to(torch.float16).to(torch.float32); l = (xs - h).to(torch.float16).to(torch.float32)
    return h / scale, l / scale
def pow2scale(x, target=2.0**14):
    m = x.abs().max().item()
    return 2.0 ** math.floor(math.log2(target / m))
for cin, cout in ((8, 8), (16, 16), (32, 32), (64, 64)):
    x = torch.relu(torch.randn(1, cin, 12, 20, 24)) * 3.0          # post-ReLU activations
    w = torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5
    ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    f32 = torch.nn.functional.conv3d(x, w, padding=1).double()
    e32 = (f32 - ref).abs()
    # representation error only (products summed in float64)
    def conv64(a, b): return torch.nn.functional.conv3d(a.double(), b.double(), padding=1)
    a1, a2, a3 = split_bf16(x); b1, b2, b3 = split_bf16(w)
    sb = conv64(a1, b1) + conv64(a1, b2) + conv64(a2, b1) + conv64(a1, b3) + conv64(a2, b2) + conv64(a3, b1)
    sx, sw = pow2scale(x), pow2scale(w)
    xh, xl = split_f16(x, sx); wh, wl = split_f16(w, sw)
    f3 = conv64(xh, wh) + conv64(xh, wl) + conv64(xl, wh)
    f4 = f3 + conv64(xl, wl)
    scale = ref.abs().max().item()
    print(f"cin {cin:2d} cout {cout:2d}: |ref|max {scale:.2f}  fp32 conv err max {e32.max():.2e} rms {e32.pow(2).mean().sqrt():.2e} | "
          f"split-bf16(6) repr err max {(sb-ref).abs().max():.2e} rms {(sb-ref).pow(2).mean().sqrt():.2e} | "
          f"fp16x2 (3 prod) max {(f3-ref).abs().max():.2e} rms {(f3-ref).pow(2).mean().sqrt():.2e} | (4 prod) max {(f4-ref).abs().max():.2e} rms {(f4-ref).pow(2).mean().sqrt():.2e}")
