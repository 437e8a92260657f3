"""CostRegNet layer by layer (HIP events around each launch, 10 runs) at the headline and cascade stage shapes: us, compulsory bytes
(input + output + skip, fp32), TB/s, algorithmic TFLOP/s.  Usage: r06_costreg_layers.py [name ...]   names: M1 M3s1 M3s2 M3s3 M2s1 M2s2 M2s3 M4s1.."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, ops
dev = torch.device("cuda")
SH = {"M1": (8, 192, 512, 640), "M3s1": (32, 48, 296, 400), "M3s2": (16, 32, 592, 800), "M3s3": (8, 8, 1184, 1600),
      "M2s1": (32, 48, 128, 160), "M2s2": (16, 32, 256, 320), "M2s3": (8, 8, 512, 640),
      "M4s1": (32, 48, 264, 480), "M4s2": (16, 32, 528, 960), "M4s3": (8, 8, 1056, 1920)}
model = seeded_init_(CDSMVSNet(refine=False), 0).eval().to(dev)
def t(fn, n=10):
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): out = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, out
for name in (sys.argv[1:] or list(SH)):
    C, D, h, w = SH[name]
    cr = model.cost_regularization[{8: 2, 16: 1, 32: 0}[C]]
    p = cr._packed.get(cr, cr._pack)
    v = torch.randn(D, h, w, C, device=dev)
    rows, tot = [], 0.0
    def run(label, fn, cin, cout, outvox, extra_bytes=0, k=27):
        global tot
        us, out = t(fn)
        by = 4.0 * (fn.in_numel + out.numel()) + extra_bytes
        fl = 2.0 * k * cin * cout * outvox
        rows.append(f"  {label:8s} {us:8.1f} us  {by / us / 1e6:5.2f} TB/s compulsory  {fl / us / 1e6:6.1f} TF")
        tot += us
        return out
    def mk(f, n):
        f.in_numel = n
        return f
    with torch.no_grad():
        if ops.USE_SPLIT_F16 and "conv0.wh" in p:      # conv0 - conv3 in split-f16 (CDS_SPLIT_F16=0: split-bf16)
            bnd = torch.zeros(4, device=dev); vb = v.abs().amax().reshape(1)
            def z(i): bnd[i:i + 1].zero_(); return bnd[i:i + 1]
            c0 = run("conv0 h", mk(lambda: ops.conv3d_sbf(v, p["conv0.wh"], p["conv0.b"], 8, stride=ops.SBF_PAIR, in_bound=vb, w_inv_scale=p["conv0.whs"], out_bound=bnd[0:1]), v.numel()), C, 8, D * h * w)
            c1 = run("conv1 h", mk(lambda: ops.conv3d_sbf(c0, p["conv1.wh"], p["conv1.b"], 16, stride=2, in_bound=bnd[0:1], w_inv_scale=p["conv1.whs"], out_bound=bnd[1:2]), c0.numel()), 8, 16, D * h * w // 8)
            c2 = run("conv2 h", mk(lambda: ops.conv3d_sbf(c1, p["conv2.wh"], p["conv2.b"], 16, in_bound=bnd[1:2], w_inv_scale=p["conv2.whs"], out_bound=bnd[2:3]), c1.numel()), 16, 16, D * h * w // 8)
            c3 = run("conv3 h", mk(lambda: ops.conv3d_sbf(c2, p["conv3.wh"], p["conv3.b"], 32, stride=2, in_bound=bnd[2:3], w_inv_scale=p["conv3.whs"]), c2.numel()), 16, 32, D * h * w // 64)
            c4 = run("conv4 h", mk(lambda: ops.conv3d_sbf(c3, p["conv4.wh"], p["conv4.b"], 32, in_bound=vb * 0 + 1e3, w_inv_scale=p["conv4.whs"], out_bound=bnd[3:4]), c3.numel()), 32, 32, D * h * w // 64)
            c5 = run("conv5 h", mk(lambda: ops.conv3d_sbf(c4, p["conv5.wh"], p["conv5.b"], 64, stride=2, in_bound=vb * 0 + 1e3, w_inv_scale=p["conv5.whs"]), c4.numel()), 32, 64, D * h * w // 512)
            c6 = run("conv6 h", mk(lambda: ops.conv3d_sbf(c5, p["conv6.wh"], p["conv6.b"], 64, in_bound=vb * 0 + 1e3, w_inv_scale=p["conv6.whs"]), c5.numel()), 64, 64, D * h * w // 512)
        else:
            c0 = run("conv0", mk(lambda: ops.conv3d_sbf(v, p["conv0.ws"], p["conv0.b"], 8, stride=ops.SBF_PAIR), v.numel()), C, 8, D * h * w)
            c1 = run("conv1", mk(lambda: ops.conv3d_sbf(c0, p["conv1.ws"], p["conv1.b"], 16, stride=2), c0.numel()), 8, 16, D * h * w // 8)
            c2 = run("conv2", mk(lambda: ops.conv3d_sbf(c1, p["conv2.ws"], p["conv2.b"], 16), c1.numel()), 16, 16, D * h * w // 8)
            c3 = run("conv3", mk(lambda: ops.conv3d_sbf(c2, p["conv3.ws"], p["conv3.b"], 32, stride=2), c2.numel()), 16, 32, D * h * w // 64)
        c4 = c4 if "c4" in dir() and False else run("conv4", mk(lambda: ops.conv3d_sbf(c3, p["conv4.ws"], p["conv4.b"], 32), c3.numel()), 32, 32, D * h * w // 64)
        c5 = run("conv5", mk(lambda: ops.conv3d_sbf(c4, p["conv5.ws"], p["conv5.b"], 64, stride=2), c4.numel()), 32, 64, D * h * w // 512)
        c6 = run("conv6", mk(lambda: ops.conv3d_sbf(c5, p["conv6.ws"], p["conv6.b"], 64), c5.numel()), 64, 64, D * h * w // 512)
        if ops.USE_SPLIT_F16 and "conv7.wh" in p:
            b7 = c6.abs().amax().reshape(1)
            run("conv7 h", mk(lambda: ops.deconv3d_sbf(c6, p["conv7.wh"], p["conv7.b"], 32, skip=c4, in_bound=b7, w_inv_scale=p["conv7.whs"]), c6.numel() + c4.numel()), 64, 32, D * h * w // 512)
        x7 = run("conv7", mk(lambda: ops.deconv3d_sbf(c6, p["conv7.ws"], p["conv7.b"], 32, skip=c4), c6.numel() + c4.numel()), 64, 32, D * h * w // 512)
        x9 = run("conv9", mk(lambda: ops.deconv3d_zm(x7, p["conv9.wc"], p["conv9.b"], skip=c2), x7.numel() + c2.numel()), 32, 16, D * h * w // 64)
        pr = run("tail", mk(lambda: ops.deconv_prob_zm(x9, p["conv11.wz"], p["conv11.b"], c0, p["prob.tab"]), x9.numel() + c0.numel()), 16, 8, D * h * w // 8)
    print(f"{name}: C={C} D={D} {w}x{h}: CostRegNet layer sum {tot:.1f} us")
    print("\n".join(rows), flush=True)
    del v, c0, c1, c2, c3, c4, c5, c6, x7, x9, pr
