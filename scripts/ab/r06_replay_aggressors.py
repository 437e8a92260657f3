"""Which launch of a forward perturbs K1?  Every C-ABI call of ONE inference forward is recorded (entry point + arguments), then each
distinct launch is replayed in a loop on stream B while K1 runs in a loop on stream A; K1's outputs are compared bit for bit with its
output alone (the per-launch form of tests/test_concurrency_gpu.py; scripts/ab/r05_aggressor.py holds hand-built shapes).

    python scripts/ab/r06_replay_aggressors.py [H W N]           # default 512 640 5
    REPLAY_ONLY=cds_conv3d python ...                             # entry points starting with ...

The forward allocates from a private memory pool, so its (stale) buffers stay valid and apart from the victim's while the launches are
replayed; a replayed launch computes on whatever the forward left there - the effect looked for does not depend on the data."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, synth, ops, geometry, _lib
from cds_mvsnet_amd import model as M

H, W, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 640, 5)
dev = torch.device("cuda")
real = _lib.load()
calls = []


class Recorder:
    def __getattr__(self, name):
        fn = getattr(real, name)
        def wrapped(*args):
            calls.append((name, tuple(a.value if isinstance(a, ctypes._SimpleCData) else a for a in args)))
            return fn(*args)
        return wrapped


# victim first: its buffers live outside the forward's pool
V, C, D, h, w = 4, 32, 48, 296, 400
feats = synth.make_pair_features(V, C, h, w, seed=1)
cams_v = synth.stage_cameras(V + 1, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev).contiguous()
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
mats = geometry.warp_matrices(cams_v[0]).to(dev)
victim = lambda: ops.warp_entropy(ref, src, mats, hyp)
want = victim().clone()

model = seeded_init_(CDSMVSNet(refine=False, depth_interals_ratio=(4.0, 1.5, 0.75)), 0).eval().to(dev)
imgs = synth.make_images(N, H, W, seed=4).to(dev)
cams = {k: v.to(dev) for k, v in synth.make_cameras(N, H, W, refine=False, seed=4).items()}
dv = synth.make_depth_values().to(dev)
with torch.no_grad():
    model(imgs, cams, dv, temperature=0.01)                 # packs the weights (outside the pool, they stay)
torch.cuda.synchronize()
pool = torch.cuda.MemPool()
rec_stream = torch.cuda.Stream()
side = M._side_stream(dev)
_lib._lib = Recorder()
with torch.cuda.use_mem_pool(pool), torch.cuda.stream(rec_stream), torch.no_grad():
    out = model(imgs, cams, dv, temperature=0.01)
torch.cuda.synchronize()
_lib._lib = real
handles = {rec_stream.cuda_stream, side.cuda_stream}
print(f"{len(calls)} C-ABI calls in a {W}x{H} N={N} forward")

distinct, seen = [], set()
for name, args in calls:
    sig = (name, tuple(a for a in args if isinstance(a, (int, float)) and not (isinstance(a, int) and a > 1 << 32)))
    if sig in seen:
        continue
    seen.add(sig)
    distinct.append((name, args))
only = os.environ.get("REPLAY_ONLY")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def ms(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_victim = ms(victim, 6)
flagged = []
for idx, (name, args) in enumerate(distinct):
    if only and not name.startswith(only):
        continue
    if not args or not isinstance(args[-1], int) or args[-1] not in handles:
        continue                                              # not a launch on a stream (cds_loss_records and the like)
    fn = getattr(real, name)

    def replay(fn=fn, args=args):
        a = list(args)
        a[-1] = torch.cuda.current_stream().cuda_stream
        rc = fn(*a)
        assert rc == 0, (name, rc)

    nag = max(4, min(3000, int(8 * t_victim / max(ms(replay, 4), 1e-3)) + 1))
    bad = tot = 0
    for rep in range(2):
        with torch.cuda.stream(sa):
            outs = [victim() for _ in range(8)]
        with torch.cuda.stream(sb):
            for _ in range(nag):
                replay()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, want)) for o in outs)
        tot += len(outs)
    ints = [a for a in args[:-1] if isinstance(a, (int, float)) and not (isinstance(a, int) and a > 1 << 32)]
    mark = "AGGRESSOR" if bad else "         "
    print(f"{mark} #{idx:3d} {name:34s} {str(ints):60s} x{nag:4d}: {bad:2d} / {tot}")
    if bad:
        flagged.append((name, ints, bad, tot))
print(f"{len(flagged)} aggressor launches" + "".join(f"\n  {n} {i}: {b} / {t}" for n, i, b, t in flagged))
