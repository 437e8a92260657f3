"""Does a hipGraph replay of the M1 step (25 kernel launches) beat eager enqueue?  Captures model.stage_net(...) into a torch.cuda.CUDAGraph."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model = seeded_init_(CDSMVSNet(refine=False, ndepths=bench.NDEPTHS, depth_interals_ratio=bench.RATIOS), 0).eval().to(dev)
h, w, D, C, n_views = bench.WORKLOADS["M1"]
_, cams, hyp, dfe = bench.make_workload("M1", 0, dev)
hyp_d = hyp.to(dev)
def step():
    return model.stage_net(dfe, cams, depth_values=hyp_d, num_depth=D, cost_regularization=model.cost_regularization[2], stage_idx=2)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    eager = timeit(step)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    ref = step()
    g.replay(); torch.cuda.synchronize()
    print("max |graph - eager| depth:", (out["depth"] - ref["depth"]).abs().max().item())
    graph = timeit(g.replay)
    eager2 = timeit(step)
print(f"eager {eager:.3f} ms, graph replay {graph:.3f} ms, eager again {eager2:.3f} ms")
