"""Gradient agreement of the full training step: HIP 2D stacks vs torch 2D stacks, and torch vs torch (run-to-run noise floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, training, losses, synth
dev = torch.device("cuda:0")
refine = len(sys.argv) > 1 and sys.argv[1] == "refine"
B, N = 1, 3
Hm, Wm = (128, 192) if refine else (64, 96)
H, W = (Hm // 2, Wm // 2) if refine else (Hm, Wm)
imgs = synth.make_images(N, Hm, Wm, seed=31).to(dev)
cams = {k: v.to(dev) for k, v in synth.make_cameras(N, Hm, Wm, refine=refine, seed=31).items()}
dv = synth.make_depth_values().to(dev)
g = torch.Generator().manual_seed(4)
base = 600.0 + 120.0 * F.interpolate(torch.rand(B, 1, 4, 6, generator=g), (Hm, Wm), mode="bicubic", align_corners=False)[:, 0]
gt, mask = {}, {}
for s, sc in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
    gt[s] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous().to(dev)
    mask[s] = (torch.rand(B, H // sc, W // sc, generator=g) > 0.15).float().to(dev)
gt["stage4"] = F.interpolate(base.unsqueeze(1), (Hm, Wm) if refine else (H, W), mode="nearest")[:, 0].contiguous().to(dev)
mask["stage4"] = torch.ones_like(gt["stage4"])
def run(hip, eps=0.0):
    global imgs
    training.USE_HIP_TRAIN2D = hip
    imgs_ = imgs * (1.0 + eps)
    model = seeded_init_(CDSMVSNet(refine=refine, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
    model.train()
    out = model(imgs_, cams, dv, gt_depths=gt, temperature=0.1)
    loss, _ = losses.final_loss(out, gt, mask, depth_interval=dv[:, 1] - dv[:, 0], dlossw=[0.5, 1.0, 2.0])
    loss.backward()
    return loss.item(), {k: out[k]["depth"].detach().clone() for k in ("stage1", "stage2", "stage3")}, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    print(f"== {tag}: loss {a[0]:.6f} vs {b[0]:.6f}")
    for k in a[1]:
        print(f"   depth {k}: max abs diff {(a[1][k] - b[1][k]).abs().max().item():.3e}")
    rows = []
    for n in b[2]:
        x, y = a[2][n].double(), b[2][n].double()
        rows.append(((x - y).abs().max().item() / max(y.abs().max().item(), 1e-9), n, y.abs().max().item()))
    rows.sort(reverse=True)
    va = torch.cat([a[2][n].double().flatten() for n in b[2]]); vb = torch.cat([b[2][n].double().flatten() for n in b[2]])
    print(f"   cosine of the whole gradient {torch.dot(va, vb).item() / (va.norm().item() * vb.norm().item()):.8f}")
    rows2 = [r for r in rows if "cost_regularization" not in r[1]]
    for r in rows2[:8]:
        print(f"   [2D] {r[0]:.3e}  {r[1]}  (scale {r[2]:.3e})")
    for r in rows[:4]:
        print(f"   {r[0]:.3e}  {r[1]}  (scale {r[2]:.3e})")
t1 = run(False); t2 = run(False, 1e-6); h1 = run(True); t3 = run(False, 1e-5)
cmp(t2, t1, "torch(images * (1 + 1e-6)) vs torch"); cmp(t3, t1, "torch(images * (1 + 1e-5)) vs torch"); cmp(h1, t1, "hip vs torch")
