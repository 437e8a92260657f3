"""Random-size fuzz of the FeatureNet runner (DynamicConv branches, blends, in-kernel InstanceNorm statistics, FPN laterals,
cross-pair sharing) against the CPU oracle.  Sizes are multiples of 4, so the coarse levels get widths that are not multiples
of 4 (the aligned 'pipe' kernels step aside there).  Usage: fuzz_feature.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import FeatureNet, seeded_init_
from cds_mvsnet_amd.model import _FeatureRunner
from oracle import cds_oracle as O
dev = torch.device("cuda:0")
net = seeded_init_(FeatureNet(8), 7).eval()
sd = {"feature." + k: v.clone() for k, v in net.state_dict().items()}
run = _FeatureRunner(net.to(dev))

if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    worst_mean = worst_max = 0.0
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
        H, W = 4 * rng.randint(2, 30), 4 * rng.randint(2, 50)
        V = rng.randint(1, 3); T = rng.choice([1.0, 0.1, 0.01]); shared = rng.random() < 0.6
        g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
        ref = torch.rand(3, H, W, generator=g)
        srcs = [torch.rand(3, H, W, generator=g) for _ in range(V)]
        epi = torch.tensor([[rng.uniform(-3 * W, 4 * W), rng.uniform(-3 * H, 4 * H)] for _ in range(2 * V)], dtype=torch.float32)
        batch = torch.stack([ref] * V + srcs).to(dev)
        out = run(batch, epi, T, n_chw=V, n_shared=V if shared else 1)
        e_mean = e_max = 0.0
        for n in range(2 * V):
            want = O.feature_net(batch[n:n + 1].cpu(), epi[n:n + 1], T, sd)
            for s in ("stage1", "stage2", "stage3"):
                fea = out[s][0][n] if n < V else out[s][1][n - V].permute(2, 0, 1)
                for got, ref_t in ((fea, want[s][0][0]), (out[s][2][n], want[s][1][0, 0]), (out[s][3][n], want[s][2][0, 0])):
                    err = (got.cpu() - ref_t).abs()
                    scale = max(1.0, float(ref_t.abs().max()))
                    e_mean = max(e_mean, float(err.mean()) / scale); e_max = max(e_max, float(err.max()) / scale)
        worst_mean, worst_max = max(worst_mean, e_mean), max(worst_max, e_max)
        flag = "" if (e_mean < 2e-5 and e_max < 2e-3) else "  <-- CHECK"
        print(f"H={H:3d} W={W:3d} V={V} T={T:4.2f} shared={int(shared)}: mean {e_mean:.2e} max {e_max:.2e}{flag}")
    print("worst mean / max error (relative to max(1, |ref|max)):", worst_mean, worst_max)
