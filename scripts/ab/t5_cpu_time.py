"""Is the config-5 training step CPU-bound?  CPU time to ENQUEUE forward / loss+backward / optimiser vs. the GPU's finish time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T, _scratch
from cds_mvsnet_amd.losses import final_loss
dev = torch.device("cuda:0")
H, W, n_views, refine = bench.TRAIN["T5"]
model = seeded_init_(CDSMVSNet(refine=refine, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 7).to(dev)
model.train()
sample = bench.train_sample(H, W, n_views, refine, dev, seed=21)
opt = T.make_optimizer(model)
dv = sample["depth_values"]; interval = dv[:, 1] - dv[:, 0]
rows = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model(sample["imgs"], sample["proj_matrices"], dv, gt_depths=sample["depth"], temperature=0.1)
    t1 = time.perf_counter()
    loss, _ = final_loss(out, sample["depth"], sample["mask"], dlossw=[0.5, 1.0, 2.0], depth_interval=interval)
    with _scratch.side_stream_weight_gradients(loss.device, T.SIDE_STREAM_WGRAD):
        loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3])
for r in rows[3:]:
    print("enqueue forward %.1f ms, loss+backward %.1f, optimiser %.1f; GPU still busy for %.1f; step %.1f" % tuple(r))
