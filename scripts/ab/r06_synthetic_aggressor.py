"""K1 beside a kernel that only issues matrix-core instructions (scripts/ubench/mfma_aggressor.hip): is dense MFMA activity itself
enough to make K1's packed-fp32 results wrong, and which activity (type, operand data, burst length, pauses)?"""
import ctypes, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from cds_mvsnet_amd import ops, synth, geometry
so = os.path.join(R, "scripts", "ubench", "mfma_aggressor.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", so[:-3] + ".hip", "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_aggressor.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda")
V, C, D, h, w = 4, 32, 48, 296, 400
feats = synth.make_pair_features(V, C, h, w, seed=1)
cams = synth.stage_cameras(V + 1, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
mats = geometry.warp_matrices(cams[0]).to(dev)
victim = lambda: ops.warp_entropy(ref, src, mats, hyp)
want = victim().clone()
out = torch.zeros(16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
KIND = {0: "f16 16x16x32", 1: "bf16 16x16x32", 2: "fp32 16x16x4", 3: "VALU fma"}


def run(kind, grid, iters, burst8, pause, zero):
    def aggr():
        rc = lib.mfma_aggressor(kind, out.data_ptr(), grid, iters, burst8, pause, zero, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    aggr(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); aggr(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)
    n = max(1, int(9.0 / max(t, 1e-3)))
    bad = tot = nel = 0
    for rep in range(2):
        with torch.cuda.stream(sa):
            outs = [victim() for _ in range(8)]
        with torch.cuda.stream(sb):
            for _ in range(n):
                aggr()
        torch.cuda.synchronize()
        for o in outs:
            d = int((o != want).sum())
            bad += d > 0; nel += d; tot += 1
    rate = 8 * burst8 * iters * grid * 4 / (t * 1e-3) / 1e12      # MFMA instructions per second, in 1e12 wave-instructions
    print(f"{KIND[kind]:14s} grid {grid:5d} burst {8 * burst8:3d} pause {pause} {'zero operands' if zero else 'random operands':15s}: "
          f"{t:6.2f} ms per launch ({rate:5.2f} T wave-MFMA/s): {bad:2d} / {tot} victim launches differ ({nel} values)")


for kind in (0, 1, 2, 3):
    run(kind, 2048, 400, 4, 0, 0)
run(0, 2048, 400, 4, 0, 1)
run(1, 2048, 400, 4, 0, 1)
for pause in (1, 2, 3):
    run(0, 2048, 400, 4, pause, 0)
for burst8 in (1, 2):
    run(0, 2048, 1600 // burst8, burst8, 1, 0)
for grid in (256, 512, 1024):
    run(0, grid, 400 * 2048 // grid, 4, 0, 0)
