"""No kernel of the library may be perturbed by - or perturb - its stream neighbours.

On the MI355X a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) delivers wrong results in some lanes when
another wave of its SIMD has 16x16x32 f16 / bf16 MFMAs in flight (profiles/r06_experiments.md; first seen in round 5 as "one kernel
corrupts its stream neighbours", profiles/r05_experiments.md).  A kernel that is bit-exact alone then computes wrong values beside a
matrix-core kernel on another stream: K1, 0.7 ms of packed position / interpolation arithmetic, showed wrong entropies in ~4 % of its
pixels.  No test of one kernel alone can see that.  The library is therefore built without the instruction class (csrc/Makefile:
NOPK; tests/test_build_flags.py checks the binary), and this file is the behavioural check: the most sensitive victims (K1 and K3 at
the stage-1 shape of the 1600x1184 cascade) run in a loop on one stream while (a) a whole inference forward, (b) a whole training step,
(c) a synthetic kernel that does nothing but MFMAs runs on another, and every victim output must equal, bit for bit, the output the
kernel gives alone.  scripts/ab/r06_replay_aggressors.py is the per-launch form (which launch of a forward is it?).
"""
import ctypes
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def victim():
    from cds_mvsnet_amd import ops, synth, geometry
    dev = torch.device("cuda")
    V, C, D, h, w = 4, 32, 48, 296, 400
    feats = synth.make_pair_features(V, C, h, w, seed=1)
    cams = synth.stage_cameras(V + 1, h, w, seed=0)
    hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
    ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
    src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
    mats = geometry.warp_matrices(cams[0]).to(dev)
    vis = torch.rand(V, h, w, generator=torch.Generator().manual_seed(3)).to(dev)
    k1 = lambda: ops.warp_entropy(ref, src, mats, hyp)
    k3 = lambda: ops.warp_aggregate(ref, src, vis, mats, hyp, channels_last=True)[0]
    want1, want3 = k1().clone(), k3().clone()
    torch.cuda.synchronize()
    assert torch.equal(k1(), want1) and torch.equal(k3(), want3)      # deterministic alone
    return {"K1": (k1, want1), "K3": (k3, want3)}


def _beside(victim_fn, want, aggressor, n_victims, reps=2):
    """Victim loop on stream A (enqueued first: it is asynchronous), aggressor on stream B; returns (differing outputs, outputs)."""
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    aggressor()                                                        # allocator and packed weights warm: the timed runs overlap
    torch.cuda.synchronize()
    bad = tot = 0
    for _ in range(reps):
        with torch.cuda.stream(sa):
            outs = [victim_fn() for _ in range(n_victims)]
        with torch.cuda.stream(sb):
            aggressor()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, want)) for o in outs)
        tot += len(outs)
    return bad, tot


@pytest.mark.parametrize("which", ["K1", "K3"])
@pytest.mark.parametrize("H,W,N", [(1184, 1600, 5), (512, 640, 5)])
def test_inference_forward_does_not_perturb_its_stream_neighbours(which, H, W, N, victim):
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
    dev = torch.device("cuda")
    model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
    imgs = synth.make_images(N, H, W, seed=4).to(dev)
    cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=False, seed=4).items()}
    dv = synth.make_depth_values().to(dev)
    nfw = 2 if H > 1000 else 6

    def forward():
        with torch.no_grad():
            for _ in range(nfw):
                model(imgs, cams, dv, temperature=0.01)

    fn, want = victim[which]
    bad, tot = _beside(fn, want, forward, n_victims=40)
    assert bad == 0, f"{bad} of {tot} {which} launches beside a {W}x{H} forward differ from {which} alone"


@pytest.mark.parametrize("which", ["K1", "K3"])
def test_training_step_does_not_perturb_its_stream_neighbours(which, victim):
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
    from test_train_harness import _train_sample
    dev = torch.device("cuda")
    sample = _train_sample(dev, B=2, N=5, H=288, W=384)
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
    opt = T.make_optimizer(model)
    fn, want = victim[which]
    bad, tot = _beside(fn, want, lambda: T.train_step(model, opt, sample, temperature=0.1), n_victims=40)
    assert bad == 0, f"{bad} of {tot} {which} launches beside a training step differ from {which} alone"


@pytest.fixture(scope="module")
def mfma_aggressor():
    """scripts/ubench/mfma_aggressor.hip: MFMAs on register operands and nothing else (built by __graft_entry__.build())."""
    ub = os.path.join(ROOT, "scripts", "ubench", "mfma_aggressor")
    if not os.path.exists(ub + ".so"):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", ub + ".hip", "-o", ub + ".so"])
    lib = ctypes.CDLL(ub + ".so")
    lib.mfma_aggressor.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("which", ["K1", "K3"])
@pytest.mark.parametrize("kind", [0, 1], ids=["f16", "bf16"])
def test_dense_mfma_kernel_does_not_perturb_k1_k3(kind, which, victim, mfma_aggressor):
    """The worst neighbour there is: 2048 workgroups that issue 16x16x32 MFMAs back to back for ~1.5 ms per launch.  With packed-fp32
    instructions in K1 this gave 6-16 wrong launches of 16 and ~10^5 wrong values (profiles/r06_experiments.md); the library is built
    without them (tests/test_build_flags.py), and K1 / K3 must come out bit-identical."""
    out = torch.zeros(16, device="cuda")

    def aggressor():
        for _ in range(8):
            rc = mfma_aggressor.mfma_aggressor(kind, out.data_ptr(), 2048, 400, 4, 0, 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0

    fn, want = victim[which]
    bad, tot = _beside(fn, want, aggressor, n_victims=12)
    assert bad == 0, f"{bad} of {tot} {which} launches beside a dense {'bf16' if kind else 'f16'} MFMA kernel differ from {which} alone"


@pytest.mark.parametrize("H,W,N", [(512, 640, 5), (1184, 1600, 5)])
def test_forward_beside_dense_mfma_kernel_is_bit_identical(H, W, N, mfma_aggressor):
    """Every kernel of the forward as the VICTIM: the whole cascade beside the synthetic matrix-core kernel must give, bit for bit, the
    depth / confidence maps it gives alone (the forward is deterministic: no floating-point atomics on the inference path)."""
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth
    dev = torch.device("cuda")
    model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
    imgs = synth.make_images(N, H, W, seed=4).to(dev)
    cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=False, seed=4).items()}
    dv = synth.make_depth_values().to(dev)
    keys = [(s_, k) for s_ in ("stage1", "stage2", "stage3") for k in ("depth", "photometric_confidence")]
    with torch.no_grad():
        model(imgs, cams, dv, temperature=0.01)
        alone = model(imgs, cams, dv, temperature=0.01)
        again = model(imgs, cams, dv, temperature=0.01)
    torch.cuda.synchronize()
    for s_, k in keys:
        assert torch.equal(alone[s_][k], again[s_][k]), f"the forward alone is not deterministic in {s_}.{k}"
    out = torch.zeros(16, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    n_aggr = 4 if H < 1000 else 12                                      # ~1.5 ms per launch: covers the forward
    for kind in (0, 1):
        for rep in range(2):
            with torch.cuda.stream(sb):
                for _ in range(n_aggr):
                    assert mfma_aggressor.mfma_aggressor(kind, out.data_ptr(), 2048, 400, 4, 0, 0, sb.cuda_stream) == 0
            with torch.cuda.stream(sa), torch.no_grad():
                got = model(imgs, cams, dv, temperature=0.01)
            torch.cuda.synchronize()
            for s_, k in keys:
                n = int((got[s_][k] != alone[s_][k]).sum())
                assert n == 0, f"{n} values of {s_}.{k} differ beside a dense {'bf16' if kind else 'f16'} MFMA kernel ({W}x{H}, run {rep})"
