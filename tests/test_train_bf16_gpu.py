"""SURVEY §8(f)-2, BASELINE config 5's "bf16": the bf16-STORAGE / fp32-ACCUMULATE policy of the training step
(cds_mvsnet_amd/train2d_ops.py ``activation_storage``, csrc/train2d.hip).  The policy's kernels against torch formulations of the
same arithmetic, the tensors autograd keeps between the passes, and the acceptance against the reference's own training step
(tests/golden/g7_training_step.npz): loss within 1e-2 relative, cosine >= 0.99 on the nine full gradient tensors."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_conversion_kernel_is_round_to_nearest_even(dev):
    from cds_mvsnet_amd import train2d_ops as T2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(100003, generator=g) * torch.logspace(-30, 30, 100003)
    # ties, values just around them, infinities, NaN, denormals, the largest finite value (rounds to inf)
    special = torch.tensor([1.0, 1.00390625, 1.001953125, 1.005859375, -1.001953125, float("inf"), -float("inf"), float("nan"), 1e-40, -1e-45,
                            3.4028234e38, 0.0, -0.0])
    x = torch.cat((x, special)).to(dev)
    got = T2.to_bf16(x)
    want = x.bfloat16()
    assert got.dtype == torch.bfloat16
    a, b = got.view(torch.int16).cpu(), want.view(torch.int16).cpu()
    nan = torch.isnan(want.float().cpu())
    assert torch.equal(a[~nan], b[~nan]) and bool(torch.isnan(got.float().cpu()[nan]).all())


@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("act", ["leaky", "tanh"])
def test_instnorm_b16_forward_backward(dev, act, strict):
    """y -> y16 = bf16(y) kept, z = act(InstanceNorm(y)) with z16 = bf16(z) kept; default: the forward values are the fp32 path's;
    strict ("bf16-forward"): statistics of y16, z32 = widened z16.  Backward from y16 in both."""
    from cds_mvsnet_amd import train2d_ops as T2, ops
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(3, 8, 37, 52, generator=g) * 2 + 0.3).to(dev)
    code = ops.ACT_LEAKY01 if act == "leaky" else ops.ACT_TANH
    yr = y.clone().requires_grad_(True)
    with T2.activation_storage("bf16-forward" if strict else "bf16"):
        z = T2.instnorm_act(yr, code)
    z16 = T2.stored(z)
    z_f32 = T2.InstNormAct.apply(y, code)                         # the fp32 path's kernel
    src = (y.bfloat16() if strict else y).double().requires_grad_(True)     # float64 autograd reference on what the forward consumed
    zn = F.instance_norm(src, eps=1e-5)
    want = F.leaky_relu(zn, 0.1) if act == "leaky" else torch.tanh(zn)
    if not strict:
        assert (z.detach() - z_f32).abs().max() < 1e-6            # the forward pass is the fp32 step's
    if act == "leaky":
        assert z16 is not None and z16.dtype == torch.bfloat16
        if strict:
            assert torch.equal(z16.float(), z.detach())           # the fp32 transient IS the stored value
            assert (z.detach().double() - want.detach()).abs().max() < 2.0 ** -8 * want.detach().abs().max()      # one bf16 rounding
        else:
            assert torch.equal(z16.view(torch.int16), z.detach().bfloat16().view(torch.int16))
    else:
        assert z16 is None                                        # stage outputs stay fp32 for the cost volume
        assert (z.detach().double() - want.detach()).abs().max() < 2e-6
    gz = torch.randn(y.shape, generator=g).to(dev)
    z.backward(gz)
    want.backward(gz.double())
    err = (yr.grad.double() - src.grad).abs().max() / src.grad.abs().max()
    # strict: the kernel differentiates exactly what the forward consumed; default: xhat is rebuilt from bf16(y), one rounding (2^-9) off
    assert err < (2e-5 if strict else 2e-2), err
    cosv = float((yr.grad.double() * src.grad).sum() / (yr.grad.double().norm() * src.grad.norm()))
    assert cosv > 0.99999


def _saved_bytes(fn):
    sizes = {}

    def pack(t):
        sizes[t.dtype] = sizes.get(t.dtype, 0) + t.numel() * t.element_size()
        return t
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        out = fn()
    return out, sizes


def _cosines(ga, gb):
    assert set(ga) == set(gb)
    return {n: float((ga[n] * gb[n]).sum() / (ga[n].norm() * gb[n].norm() + 1e-30)) for n in ga}


def test_dynconv_unit_strict_policy_equals_its_torch_emulation(dev):
    """ONE DynamicConv unit (branches -> epilogue -> InstanceNorm + LeakyReLU) under "bf16-forward" against the same policy written with
    torch ops (tests/torch_training_ref.py: rounding where the kernels round, straight-through gradient).  One level of rounding, so
    the two implementations can only differ by the rare value that fp32 summation order pushes across a bf16 rounding boundary.
    (Across the whole FeatureNet such flips multiply layer by layer - quantised forward passes are chaotic - which is why the network-
    level comparison below is against bounds, not against the emulation.)"""
    import torch_training_ref as TR
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, training, train2d_ops as T2
    model = seeded_init_(CDSMVSNet(refine=False), 3).to(dev).train()
    unit = model.feature.conv01
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(4, 8, 64, 96, generator=g).bfloat16().float().to(dev)           # a stored activation
    epi = (torch.rand(4, 2, generator=g) * 200 - 50).to(dev)
    wt = torch.randn(4, 8, 64, 96, generator=g).to(dev)
    res = {}
    for name in ("hip", "torch"):
        for p in unit.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if name == "hip":
            with T2.activation_storage("bf16-forward"):
                z, nc = training._unit(unit, T2.with_store(x, x0.bfloat16()), epi, 0.1, 2)
        else:
            TR.BF16_STORAGE = True
            try:
                with TR.torch_layers(two_d=True, three_d=False):
                    z, nc = training._unit(unit, x, epi, 0.1, 2)
            finally:
                TR.BF16_STORAGE = False
        ((z * wt).sum() + nc.sum()).backward()
        res[name] = (z.detach(), nc.detach(), x.grad.detach().double().cpu(),
                     {n: p.grad.detach().double().cpu() for n, p in unit.named_parameters() if p.grad is not None})
    zh, nh, gxh, gh = res["hip"]
    zt, nt, gxt, gt = res["torch"]
    flips = ((zh - zt).abs() > 1e-6).float().mean().item()
    assert flips < 2e-3, flips                                    # a few values on the other side of a bf16 boundary
    assert (nh - nt).abs().max() < 1e-3 * nt.abs().max()
    cos = _cosines(gh, gt)
    assert min(cos.values()) > 0.9995, cos
    assert float((gxh * gxt).sum() / (gxh.norm() * gxt.norm())) > 0.9995


def test_featurenet_keeps_bf16_between_the_passes_and_gradients_agree(dev):
    """What autograd holds for FeatureNet's backward under the policies (the activation bytes halve, nothing large stays fp32) and the
    parameter gradients against the fp32 policy: "bf16" (the shipped form) has the fp32 forward pass and per-parameter cosines
    > 0.99 (median > 0.9995; the lowest are the 4 x K tensors of the attention MLPs, whose ReLU gates are rebuilt from rounded curvatures); "bf16-forward" is bounded loosely (see train2d_ops: every activation carries 2^-9 through nine DynamicConvs)."""
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, training, train2d_ops as T2
    model = seeded_init_(CDSMVSNet(refine=False), 3).to(dev).train()
    net = model.feature
    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 96, 128, generator=g).to(dev)
    epi = (torch.rand(4, 2, generator=g) * 200 - 50).to(dev)
    wts = [torch.randn(4, c, 96 // s, 128 // s, generator=g).to(dev) for c, s in ((32, 4), (16, 2), (8, 1))]

    def run(kind):
        for p in net.parameters():
            p.grad = None
        with T2.activation_storage(kind):
            out, sizes = _saved_bytes(lambda: training.feature_net(net, x, epi, 0.1, groups=2))
        loss = sum((out[f"stage{i + 1}"][0] * wts[i]).sum() + out[f"stage{i + 1}"][1].sum() for i in range(3))
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), sizes, {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}

    l32, s32, g32 = run("f32")
    assert s32.get(torch.bfloat16, 0) == 0
    act32 = s32[torch.float32]
    for kind in ("bf16", "bf16-forward"):
        l16, s16, g16 = run(kind)
        # under the policy: what is left in fp32 are the weights, the 3-channel images, statistics rows and the unrounded tanh outputs
        assert s16[torch.bfloat16] > 0.45 * act32 and s16.get(torch.float32, 0) < 0.06 * act32, (kind, s32, s16)
        cos = _cosines(g16, g32)
        vals = sorted(cos.values())
        print(kind, "vs f32: saved bytes", s16, "(f32:", s32, ") loss", l16, l32, "lowest cosines",
              {n: round(c, 5) for n, c in sorted(cos.items(), key=lambda kv: kv[1])[:4]})
        if kind == "bf16":
            assert abs(l16 - l32) <= 1e-6 * abs(l32), (l16, l32)       # the forward pass is the fp32 step's
            assert vals[0] > 0.99 and vals[len(vals) // 2] > 0.9995, vals[:4]
        else:
            assert abs(l16 - l32) < 2e-2 * abs(l32)
            assert vals[0] > 0.9 and vals[len(vals) // 2] > 0.99, vals[:4]


def test_wgrad_reads_the_stored_input(dev):
    """cds_conv2d_wgrad_xb16_f32 == cds_conv2d_wgrad_f32 on the widened input, bit for bit per workgroup partition (same kernel, same
    order; the final atomics make the last bits order-dependent, hence a tolerance of a few ulps of the largest entry)."""
    from cds_mvsnet_amd import train2d_ops as T2
    g = torch.Generator().manual_seed(4)
    for (cin, co, k, s) in ((8, 11, 3, 1), (16, 19, 5, 1), (3, 11, 11, 1), (8, 16, 3, 2), (24, 8, 1, 1)):
        x = torch.randn(2, cin, 40, 56, generator=g).to(dev)
        x16 = T2.to_bf16(x)
        pad = (k - 1) // 2 if s == 1 else 1
        ho, wo = (40 + 2 * pad - k) // s + 1, (56 + 2 * pad - k) // s + 1
        gy = torch.randn(2, co, ho, wo, generator=g).to(dev)
        a = T2.conv2d_wgrad(gy, x16, k, s, pad).clone()
        b = T2.conv2d_wgrad(gy, x16.float(), k, s, pad).clone()
        assert (a - b).abs().max() <= 1e-5 * b.abs().max(), (cin, co, k, s)


@pytest.mark.parametrize("kind", ["bf16", "bf16-forward"])
def test_training_step_bf16_storage_against_reference_step(dev, kind):
    """The acceptance of the policy: the reference's own training step (G7, fp32 on the CPU) against model.train() forward + final_loss +
    backward with bf16-stored FeatureNet activations: loss within 1e-2 relative, cosine >= 0.99 on every full gradient tensor of the
    fixture, median gradient norm within 2 %.  "bf16-forward" is held to the loss bound only (measured cosines 0.76-0.98: see
    train2d_ops)."""
    from cds_mvsnet_amd import CDSMVSNet, final_loss, seeded_init_, train2d_ops as T2
    g = load_golden("g7_training_step")
    model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev).train()
    cams = {k[4:]: v.to(dev) for k, v in g.items() if k.startswith("cam_")}
    gt = {k[3:]: v.to(dev) for k, v in g.items() if k.startswith("gt_")}
    mask = {k[5:]: v.to(dev) for k, v in g.items() if k.startswith("mask_")}
    dv = g["depth_values"].to(dev)
    with T2.activation_storage(kind):
        out = model(g["imgs"].to(dev), cams, dv, gt_depths=gt, temperature=0.1)
    loss, depth_loss = final_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0], depth_interval=dv[:, 1] - dv[:, 0])
    loss.backward()
    rel = abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))
    assert rel < 1e-2, rel
    params = dict(model.named_parameters())
    full = [k for k in g if k.startswith("fullgrad:")]
    assert len(full) >= 9
    cos = {}
    for key in full:
        want_g = g[key].double()
        got_g = params[key[len("fullgrad:"):]].grad.detach().cpu().double()
        cos[key] = float((got_g * want_g).sum() / (got_g.norm() * want_g.norm() + 1e-30))
    print(kind, "step vs G7: loss rel", rel, "cosines", {k[9:]: round(v, 5) for k, v in cos.items()})
    if kind == "bf16-forward":
        assert min(cos.values()) > 0.5
        return
    assert min(cos.values()) >= 0.99, cos
    names = [str(n) for n in g["param_names"]]
    want = {n: float(v) for n, v in zip(names, g["grad_norms"])}
    got = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    relg = sorted(abs(got[n] - want[n]) / max(want[n], 1e-6) for n in want)
    assert relg[len(relg) // 2] < 2e-2                            # median gradient norm within 2 %


def test_train_step_accepts_the_policy(dev):
    """train.train_step(activation_storage="bf16") runs optimisation steps, the loss moves as under fp32 and the process default is
    restored afterwards."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T, train2d_ops as T2
    losses = {}
    for kind in ("f32", "bf16"):
        model = seeded_init_(CDSMVSNet(refine=True), 0).to(dev)
        sample = bench.train_sample(256, 320, 3, True, dev)
        opt = T.make_optimizer(model, lr=1e-3)
        losses[kind] = [T.train_step(model, opt, sample, temperature=0.1, activation_storage=kind)[0] for _ in range(3)]
    assert T2.activation_storage_dtype() == torch.float32
    assert all(np.isfinite(losses["bf16"]))
    assert abs(losses["bf16"][0] - losses["f32"][0]) < 2e-2 * abs(losses["f32"][0])
    assert abs(losses["bf16"][2] - losses["f32"][2]) < 5e-2 * abs(losses["f32"][2])
