"""Time one training step (BASELINE config 5 shape: 768x576, N=5, refine=True) under each activation-storage policy (train.py):
fp32, bf16 storage / f32 accumulate, and its strict form; also prints the peak allocated memory of the steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, train as T
import torch.nn.functional as F
H, W, N, B = 576, 768, 5, 1
dev = torch.device("cuda:0")
imgs = synth.make_images(N, H, W, seed=1).to(dev)
cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=True, seed=1).items()}
dv = synth.make_depth_values().to(dev)
g = torch.Generator().manual_seed(9)
base = 600.0 + 120.0 * F.interpolate(torch.rand(B, 1, 4, 6, generator=g), (H, W), mode="bicubic", align_corners=False)[:, 0]
gt, mask = {}, {}
for s, sc in (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)):
    gt[s] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous().to(dev)
    mask[s] = torch.ones(B, H // sc, W // sc, device=dev)
sample = {"imgs": imgs, "proj_matrices": cams, "depth_values": dv, "depth": gt, "mask": mask}
for kind in (sys.argv[1:] or ["f32", "bf16", "bf16-forward"]):
    model = seeded_init_(CDSMVSNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
    opt = T.make_optimizer(model)
    red = T.GradAllReducer(model.parameters())
    for _ in range(3):
        loss = T.train_step(model, opt, sample, 0.1, reducer=red, activation_storage=kind)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = T.train_step(model, opt, sample, 0.1, reducer=red, activation_storage=kind)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"train step {W}x{H} N={N} B={B} {kind}: {dt*1e3:.1f} ms  loss {loss[0]:.4f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.3f} GiB")
    del model, opt, red
