"""how much of K3 (VALU / LDS bound) and conv0 (matrix-pipe bound) can share the machine?  The fused
K3-inside-conv0-producer kernel wins only by (a) dropping the volume's write + read and (b) running K3's VALU work
under conv0's MFMA work on the same CUs.  (b) has a cheap upper-bound proxy: launch the two kernels of TWO DIFFERENT depth maps on two
streams and compare the pair's makespan with the two run back to back.  M1 shape (640x512, D=192, C=8, N=5)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, ops, synth, geometry
h, w, D, C, N = 512, 640, 192, 8, 5
dev = torch.device("cuda:0")
model = seeded_init_(CDSMVSNet(refine=False), 0).eval().to(dev)
cr = model.cost_regularization[0] if hasattr(model.cost_regularization, "__getitem__") else model.cost_regularization
cr = [m for m in model.modules() if type(m).__name__ == "CostRegNet" and m.conv0.conv.in_channels == C][0]
p = cr._packed.get(cr, cr._pack)
feats = synth.make_pair_features(N - 1, C, h, w, seed=1)
cams = synth.stage_cameras(N, h, w, seed=0)
hyp = synth.make_hypotheses(D, h, w, seed=1)[0].to(dev)
ref = torch.stack([f["ref"][0][0] for f in feats]).to(dev).contiguous()
src = torch.stack([ops.chw_to_hwc(f["src"][0][0].to(dev).contiguous()) for f in feats])
vis = torch.rand(N - 1, h, w, device=dev)
mats = geometry.warp_matrices(cams[0])
volA = torch.empty(D, h, w, C, device=dev); vsA = torch.empty(h, w, device=dev)
volB = torch.randn(D, h, w, C, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
REP = 6

def k3():
    ops.warp_aggregate(ref, src, vis, mats, hyp, volume=volA, vis_sum=vsA, channels_last=True)

def conv0():
    return ops.conv3d_sbf(volB, p["conv0.ws"], p["conv0.b"], 8, stride=ops.SBF_PAIR)

def timed(fa, fb):
    """REP launches of fa on stream A and of fb on stream B (either may be None); makespan in ms per (fa, fb) pair."""
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    sA.wait_event(t0); sB.wait_event(t0)
    for _ in range(REP):
        if fa is not None:
            with torch.cuda.stream(sA): fa()
        if fb is not None:
            with torch.cuda.stream(sB): fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    t1.record(); t1.synchronize()
    return t0.elapsed_time(t1) / REP

with torch.no_grad():
    for _ in range(2): timed(k3, conv0)
    rows = {"K3 alone": [], "conv0 alone": [], "K3 || conv0 (two streams)": []}
    for _ in range(7):
        rows["K3 alone"].append(timed(k3, None))
        rows["conv0 alone"].append(timed(None, conv0))
        rows["K3 || conv0 (two streams)"].append(timed(k3, conv0))
med = {k: statistics.median(v) for k, v in rows.items()}
for k, v in med.items(): print(f"{k:28s} {v:.3f} ms per launch (median of 7 x {REP})")
seq = med["K3 alone"] + med["conv0 alone"]
par = med["K3 || conv0 (two streams)"]
print(f"back to back {seq:.3f} ms, concurrent {par:.3f} ms: overlap hides {seq - par:.3f} ms = {100 * (seq - par) / seq:.1f} % of the pair")
vol_bytes = 2 * C * D * h * w * 4
print(f"volume write + read the fused kernel would drop: {vol_bytes / 1e9:.2f} GB")
