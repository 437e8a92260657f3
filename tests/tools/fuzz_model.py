"""Random-configuration fuzz of the whole CDSMVSNet forward (sizes, view counts, refinement, batch) against the CPU oracle.
Usage: fuzz_model.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
from oracle import cds_oracle as O
dev = torch.device("cuda:0")

if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    models = {}
    worst = 0.0
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
        refine = rng.random() < 0.4
        m = 64 if refine else 32
        H, W = m * rng.randint(1, 4 if not refine else 2), m * rng.randint(1, 5 if not refine else 3)
        N = rng.randint(2, 6); B = rng.choice([1, 1, 2])
        if refine not in models:
            cpu = seeded_init_(CDSMVSNet(refine=refine, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75)), 7).eval()
            models[refine] = ({k: v.clone() for k, v in cpu.state_dict().items()}, cpu.to(dev))
        sd, model = models[refine]
        seed = rng.randint(0, 10_000)
        imgs = torch.cat([synth.make_images(N, H, W, seed=seed + b) for b in range(B)])
        cams = [synth.make_cameras(N, H, W, refine=refine, seed=seed + b) for b in range(B)]
        cams = {k: torch.cat([c[k] for c in cams]) for k in cams[0]}
        dv = synth.make_depth_values().repeat(B, 1)
        want = O.forward(imgs, cams, dv, sd, refine=refine, temperature=0.01, exact=False)
        with torch.no_grad():
            got = model(imgs.to(dev), cams, dv, temperature=0.01)
        errs = [float((got[f"stage{s}"]["depth"].cpu() - want[f"stage{s}"]["depth"]).abs().mean()) for s in (1, 2, 3)]
        errs.append(float((got["refined_depth"].cpu() - want["refined_depth"]).abs().mean()))
        errs.append(float((got["photometric_confidence"].cpu() - want["photometric_confidence"]).abs().mean()))
        worst = max(worst, max(errs))
        flag = "" if max(errs) < 1e-3 else "  <-- FAIL"
        print(f"{W}x{H} N={N} B={B} refine={int(refine)}: depth L1 per stage " + " ".join(f"{e:.1e}" for e in errs[:3]) +
              f"  refined {errs[3]:.1e}  conf {errs[4]:.1e}{flag}")
    print("worst mean-L1", worst)
