"""Config-5 training step: host enqueue time (train_step returns float(loss): it synchronises at the end, so the host side is measured with
the loss read deferred) against the GPU time of the same step, per phase.  Prints ms per step: forward enqueue, loss + backward enqueue,
optimiser, and the synchronised total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T, final_loss

dev = torch.device("cuda:0")
H, W, n, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 0).to(dev).train()
sample = bench.train_sample(H, W, n, refine, dev)
opt = T.make_optimizer(model)
for _ in range(4):
    T.train_step(model, opt, sample, temperature=0.1)
torch.cuda.synchronize()
dv = sample["depth_values"]
interval = dv[:, 1] - dv[:, 0]
acc = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "opt": 0.0, "total": 0.0}
N = 6
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model(sample["imgs"], sample["proj_matrices"], dv, gt_depths=sample["depth"], temperature=0.1)
    t1 = time.perf_counter()
    loss, dl = final_loss(out, sample["depth"], sample["mask"], dlossw=[0.5, 1.0, 2.0], depth_interval=interval)
    t2 = time.perf_counter()
    T._backward(model, loss)
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    for k, v in (("fwd", t1 - t0), ("loss", t2 - t1), ("bwd", t3 - t2), ("opt", t4 - t3), ("total", t5 - t0)):
        acc[k] += v * 1e3 / N
print("host enqueue ms per step:", {k: round(v, 2) for k, v in acc.items()}, "| host sum", round(sum(v for k, v in acc.items() if k != "total"), 2))

if os.environ.get("T5_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(4):
        T.train_step(model, opt, sample, temperature=0.1)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)
