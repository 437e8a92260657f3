#!/bin/bash
# FeatureNet / cascade forward with split-f16 on and off (same box): eager forward ms at the cascade sizes
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for f in 0 1; do
  echo "CDS_SPLIT_F16=$f"
  CDS_SPLIT_F16=$f python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
dev = torch.device("cuda")
model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
for H, W, N in ((512, 640, 5), (1184, 1600, 5), (1056, 1920, 7)):
    imgs = synth.make_images(N, H, W, seed=0).to(dev)
    pm, dv = synth.make_cameras(N, H, W, refine=False, seed=0), synth.make_depth_values()
    with torch.no_grad():
        for _ in range(3): out = model(imgs, pm, dv, temperature=0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = model(imgs, pm, dv, temperature=0.01)
        torch.cuda.synchronize()
    print(f"  {W}x{H} N={N}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per forward; depth mean {float(out['depth'].mean()):.5f}")
PY
done
