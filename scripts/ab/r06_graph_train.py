"""Eager vs hipGraph-replayed training step (BASELINE config 5 shape 768x576 N=5 refine=True): ms per step with the loss read every
step (as train_step does) and with the loss read only at the end (host runs ahead)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
sys.argv = sys.argv[:1] + sys.argv[1:]
import bench
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
dev = torch.device("cuda")
H, W, N = 576, 768, 5
kinds = sys.argv[1:] or ["f32"]
for host_cams in (False, True):
    sample = bench.train_sample(H, W, N, True, dev, seed=21)
    if host_cams:
        sample["proj_matrices"] = {k: v.cpu() for k, v in sample["proj_matrices"].items()}
        sample["depth_values"] = sample["depth_values"].cpu()
    for kind in kinds:
        for mode in ("eager", "graph"):
            model = seeded_init_(CDSMVSNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
            opt = T.make_optimizer(model)
            red = T.GradAllReducer(model.parameters())
            if mode == "eager":
                fn = lambda: T._step_tensors(model, opt, sample, 0.1, (0.5, 1.0, 2.0), red, kind)
            else:
                cs = T.CapturedTrainStep(model, opt, reducer=red, activation_storage=kind)
                fn = lambda: cs(sample, 0.1)
            for _ in range(4): l = fn()
            float(l[0]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): l = fn(); float(l[0])
            every = (time.perf_counter() - t0) / 10 * 1e3
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): l = fn()
            host = (time.perf_counter() - t0) / 10 * 1e3
            float(l[0]); torch.cuda.synchronize()
            deferred = (time.perf_counter() - t0) / 10 * 1e3
            print(f"{W}x{H} N={N} {kind} cams on {'host' if host_cams else 'device'} {mode}: loss read every step {every:.2f} ms; "
                  f"loss read at the end {deferred:.2f} ms (host enqueue {host:.2f} ms); loss {float(l[0]):.4f}", flush=True)
            del model, opt, red
