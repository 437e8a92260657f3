import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import ops
dev = torch.device("cuda")
Da, Ha, Wa = 96, 256, 320
x = torch.randn(Da, Ha, Wa, 16, device=dev); skip = torch.randn(2 * Da, 2 * Ha, 2 * Wa, 8, device=dev)
w = torch.randn(16, 8, 3, 3, 3, device=dev) / 54 ** 0.5; b = torch.randn(8, device=dev)
wp = torch.randn(1, 8, 3, 3, 3, device=dev) / 216 ** 0.5
ws, pws = ops.split_pack_deconv3d(w), ops.split_pack_prob_toeplitz(wp)
for _ in range(4):
    p3 = ops.deconv3d_prob_sbf(x, ws, b, skip, pws)
torch.cuda.synchronize()
print("ok")
