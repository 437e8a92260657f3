"""Timing of the fused conv11 + prob kernel against the two shipped kernels at the M1 shapes (+ K5 on both logit forms)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
Da, Ha, Wa = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (96, 256, 320)
x = torch.randn(Da, Ha, Wa, 16, device=dev); skip = torch.randn(2 * Da, 2 * Ha, 2 * Wa, 8, device=dev)
w = torch.randn(16, 8, 3, 3, 3, device=dev) / 54 ** 0.5; b = torch.randn(8, device=dev)
wp = torch.randn(1, 8, 3, 3, 3, device=dev) / 216 ** 0.5
ws, pws = ops.split_pack_deconv3d(w), ops.split_pack_prob_toeplitz(wp)
wpk = wp.permute(1, 2, 3, 4, 0).reshape(8, 27, 1).contiguous()
hyp = 400 + 500 * torch.rand(2 * Da, 2 * Ha, 2 * Wa, device=dev)
t_f = t(lambda: ops.deconv3d_prob_sbf(x, ws, b, skip, pws))
p3 = ops.deconv3d_prob_sbf(x, ws, b, skip, pws)
t_k5f = t(lambda: ops.softargmin_conf_p3(p3, hyp))
t_d = t(lambda: ops.deconv3d_sbf(x, ws, b, 8, skip=skip, out_planar=True))
y = ops.deconv3d_sbf(x, ws, b, 8, skip=skip, out_planar=True)
t_p = t(lambda: ops.conv3d_k3(y, wpk, None, relu=False))
pre = ops.conv3d_k3(y, wpk, None, relu=False)[0]
t_k5 = t(lambda: ops.softargmin_conf(pre, hyp))
got = p3[1].clone(); got[1:] += p3[0][:-1]; got[:-1] += p3[2][1:]
print(f"{os.environ.get('TAG','')} fused conv11+prob {t_f:.0f} us + K5(p3) {t_k5f:.0f} us = {t_f + t_k5f:.0f} | conv11 {t_d:.0f} + prob {t_p:.0f} + K5 {t_k5:.0f} = {t_d + t_p + t_k5:.0f} us | max diff {(got - pre).abs().max().item():.2e}")
