"""Capture golden input/output vectors from the REFERENCE implementation (CPU, fp32).

Run in the build container only (``/root/reference`` is not present on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported read-only; weights come from this repo's own seeded initialiser
(``cds_mvsnet_amd.seeded_init_``) and are loaded into the reference through the shared state-dict
keys, so no reference weights or source text are vendored.  Only data (inputs + expected outputs) is
written, as ``tests/golden/*.npz``.  Fixtures: G1 warp/aggregate, G2 CostRegNet,
G3 regression, G4 hypotheses, G5 DynamicConv/FeatureNet/epipoles, G6 full forward.
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")
torch.set_num_threads(8)

from models.model import CDSMVSNet as RefNet  # noqa: E402  (reference)
from models.module import CostRegNet as RefCostReg, FeatureNet as RefFeatureNet  # noqa: E402
from models.module import conf_regression, depth_regression, get_depth_range_samples  # noqa: E402
from models.dynamic_conv import DynamicConv as RefDynConv, compute_Fmatrix, compute_epipole  # noqa: E402
from models.utils.warping import homo_warping_3D  # noqa: E402

from cds_mvsnet_amd import seeded_init_, synth  # noqa: E402

SEED = 7


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB", {k: v.shape for k, v in out.items()})


def ref_model(refine):
    m = RefNet(refine=refine, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75))
    seeded_init_(m, SEED)
    return m.eval()


@torch.no_grad()
def g1_warp_aggregate():
    model = ref_model(False)
    for tag, (h, w, D, C, N, stage) in {"a": (32, 40, 16, 8, 3, 2), "b": (16, 24, 48, 32, 3, 0),
                                        "c": (24, 32, 16, 16, 4, 1)}.items():
        feats = synth.make_pair_features(N - 1, C, h, w, seed=11 + stage, sharp=(tag != "a"))
        cams = synth.stage_cameras(N, h, w, seed=3 + stage)
        hyp = synth.make_hypotheses(D, h, w, seed=5 + stage)
        captured = {}

        def vis_hook(mod, inp, out, store=captured):
            store.setdefault("vis_in", []).append(inp[0].clone())
            store.setdefault("vis_out", []).append(out.clone())

        def cr_hook(mod, inp, store=captured):
            store["volume_mean"] = inp[0].clone()

        h1 = model.stage_net.vis[stage].register_forward_hook(vis_hook)
        h2 = model.cost_regularization[stage].register_forward_pre_hook(cr_hook)
        out = model.stage_net(feats, cams, depth_values=hyp, num_depth=D,
                              cost_regularization=model.cost_regularization[stage], stage_idx=stage)
        h1.remove()
        h2.remove()
        # plain warp of view 0 + the matrices the reference derives (model.py:40-43, warping.py:80)
        projs = []
        for v in range(N):
            P = cams[:, v, 0].clone()
            P[:, :3, :4] = torch.matmul(cams[:, v, 1, :3, :3], cams[:, v, 0, :3, :4])
            projs.append(P)
        mats = []
        for v in range(1, N):
            M = torch.matmul(projs[v], torch.inverse(projs[0]))[0]
            mats.append(torch.cat((M[:3, :3].reshape(9), M[:3, 3].reshape(3))))
        warped0 = homo_warping_3D(feats[0]["src"][0], projs[1], projs[0], hyp)
        save(f"g1_warp_aggregate_{tag}",
             ref_fea=torch.cat([f["ref"][0] for f in feats]), src_fea=torch.cat([f["src"][0] for f in feats]),
             ref_nc_sum=torch.cat([f["ref"][1] for f in feats]), src_nc_sum=torch.cat([f["src"][1] for f in feats]),
             ref_nc=torch.cat([f["ref"][2] for f in feats]), cams=cams, hyp=hyp, mats=torch.stack(mats),
             stage=np.int64(stage), warped0=warped0[0] if tag == "a" else np.zeros(1, np.float32),
             entropy=torch.cat([x[:, 0] for x in captured["vis_in"]]), vis_w=torch.cat([x[:, 0] for x in captured["vis_out"]]),
             volume_mean=captured["volume_mean"][0], depth=out["depth"], conf=out["photometric_confidence"],
             norm_curv=out["norm_curv"])


@torch.no_grad()
def g2_costreg():
    for tag, (C, D, h, w) in {"c8": (8, 8, 16, 24), "c16": (16, 16, 16, 24), "c32": (32, 8, 16, 24),
                              "c8_wide": (8, 8, 8, 64)}.items():
        net = RefCostReg(in_channels=C, base_channels=8)
        seeded_init_(net, SEED + C)
        net.eval()
        g = torch.Generator().manual_seed(C)
        vol = 0.3 * torch.randn(1, C, D, h, w, generator=g)
        save(f"g2_costreg_{tag}", volume=vol[0], cost_reg=net(vol)[0, 0], seed=np.int64(SEED + C))


@torch.no_grad()
def g3_regress():
    g = torch.Generator().manual_seed(3)
    for tag, (D, h, w) in {"d48": (48, 16, 24), "d8": (8, 16, 24), "d192": (192, 8, 16)}.items():
        pre = 2.0 * torch.randn(1, D, h, w, generator=g)
        # edge cases: sharp peaks at the first / last plane and at an interior plane
        pre[0, 0, 0, :8] += 30.0
        pre[0, D - 1, 1, :8] += 30.0
        pre[0, D // 2, 2, :8] += 30.0
        pre[0, 1, 3, :8] += 12.0
        pre[0, D - 2, 4, :8] += 12.0
        hyp = synth.make_hypotheses(D, h, w, seed=D)
        prob = F.softmax(pre, dim=1)
        save(f"g3_regress_{tag}", prob_pre=pre[0], hyp=hyp[0], prob=prob[0], depth=depth_regression(prob, hyp)[0],
             conf=conf_regression(prob)[0])


@torch.no_grad()
def g4_hypotheses():
    g = torch.Generator().manual_seed(4)
    H, W = 64, 96
    dv = synth.make_depth_values()
    dmin, dmax = dv[:, [0]].unsqueeze(-1).unsqueeze(-1), dv[:, [-1]].unsqueeze(-1).unsqueeze(-1)
    dint = (dv[:, 1] - dv[:, 0]).unsqueeze(-1).unsqueeze(-1)
    cases = {}
    # first stage: global planes
    full = get_depth_range_samples(cur_depth=dv, ndepth=48, depth_inteval_pixel=4.0 * dint, dtype=torch.float32,
                                   device=torch.device("cpu"), shape=[1, H, W], max_depth=dmax, min_depth=dmin)
    cases["s1"] = F.interpolate(full.unsqueeze(1), [48, H // 4, W // 4], mode="trilinear", align_corners=False).squeeze(1)
    # later stages, previous depth spans beyond both clamps
    for tag, (hp, wp, D, ratio, scale) in {"s2": (H // 4, W // 4, 32, 1.5, 2), "s3": (H // 2, W // 2, 8, 0.75, 1),
                                           "s2x4": (H // 4, W // 4, 16, 4.0, 4)}.items():
        prev = 415.0 + 500.0 * torch.rand(1, hp, wp, generator=g)
        cur = F.interpolate(prev.unsqueeze(1), [H, W], mode="bilinear", align_corners=False).squeeze(1)
        full = get_depth_range_samples(cur_depth=cur, ndepth=D, depth_inteval_pixel=ratio * dint, dtype=torch.float32,
                                       device=torch.device("cpu"), shape=[1, H, W], max_depth=dmax, min_depth=dmin)
        cases[tag] = F.interpolate(full.unsqueeze(1), [D, H // scale, W // scale], mode="trilinear",
                                   align_corners=False).squeeze(1)
        cases[tag + "_prev"] = prev
        cases[tag + "_meta"] = np.array([D, ratio, scale], dtype=np.float64)
    save("g4_hypotheses", depth_values=dv, H=np.int64(H), W=np.int64(W), **cases)


@torch.no_grad()
def g5_features():
    H, W = 64, 96
    cams = synth.make_cameras(3, H, W, refine=False, seed=5)["stage3"]
    Fm = compute_Fmatrix(cams[:, 0], cams[:, 1])
    e_ref, e_src = compute_epipole(Fm), compute_epipole(torch.transpose(Fm, 1, 2))
    img = synth.make_images(1, H, W, seed=5)[:, 0]
    # single DynamicConv (K = 3) with bias
    dc = RefDynConv(3, 8, size_kernels=(3, 7, 11))
    seeded_init_(dc, SEED)
    dc.eval()
    dyn = {}
    for T in (1.0, 0.1, 0.01):
        y, nc = dc(img, epipole=e_ref, temperature=T)
        dyn[f"y_T{T}"] = y[0]
        dyn[f"nc_T{T}"] = nc[0, 0]
    save("g5_dynconv", img=img[0], cams=cams, fmatrix=Fm, epipole_ref=e_ref, epipole_src=e_src, **dyn)
    net = RefFeatureNet(base_channels=8, arch_mode="fpn")
    seeded_init_(net, SEED)
    net.eval()
    feats = {}
    for T in (1.0, 0.01):
        out = net(img, epipole=e_ref, temperature=T)
        for s in ("stage1", "stage2", "stage3"):
            feats[f"{s}_fea_T{T}"] = out[s][0][0]
            feats[f"{s}_ncsum_T{T}"] = out[s][1][0, 0]
            feats[f"{s}_nc_T{T}"] = out[s][2][0, 0]
    save("g5_featurenet", img=img[0], epipole=e_ref, **feats)


@torch.no_grad()
def g9_feature_noise():
    """The reference's OWN fp32 round-off on the G5 FeatureNet case, so the FeatureNet tolerances are derived instead of
    chosen: the same module evaluated (i) in float64 (stored rounded to fp32: the 'exact' answer), (ii) in fp32 with the
    oneDNN convolutions (= G5) and (iii) in fp32 with ATen's native convolutions (``mkldnn`` off: another valid fp32
    summation order).  At T = 0.01 the softmax(./T) blend amplifies convolution round-off by up to 0.25 / T per layer,
    and the reference's two fp32 evaluations already differ by ~1.6e-4 on this image (3.5e-4 at 256x320)."""
    import copy
    H, W = 64, 96
    cams = synth.make_cameras(3, H, W, refine=False, seed=5)["stage3"]
    e_ref = compute_epipole(compute_Fmatrix(cams[:, 0], cams[:, 1]))
    img = synth.make_images(1, H, W, seed=5)[:, 0]
    net = RefFeatureNet(base_channels=8, arch_mode="fpn")
    seeded_init_(net, SEED)
    net.eval()
    net64 = copy.deepcopy(net).double()
    arrays = {}
    for T in (1.0, 0.01):
        a = net(img, epipole=e_ref, temperature=T)
        with torch.backends.mkldnn.flags(enabled=False):
            b = net(img, epipole=e_ref, temperature=T)
        c = net64(img.double(), epipole=e_ref.double(), temperature=T)
        for s in ("stage1", "stage2", "stage3"):
            for j, key in enumerate(("fea", "ncsum", "nc")):
                x64 = c[s][j][0] if j == 0 else c[s][j][0, 0]
                xa = a[s][j][0] if j == 0 else a[s][j][0, 0]
                xb = b[s][j][0] if j == 0 else b[s][j][0, 0]
                arrays[f"{s}_{key}_T{T}_f64"] = x64.float()
                arrays[f"{s}_{key}_T{T}_ref32_vs_f64_max"] = np.float64((xa.double() - x64).abs().max())
                arrays[f"{s}_{key}_T{T}_native32_vs_f64_max"] = np.float64((xb.double() - x64).abs().max())
                arrays[f"{s}_{key}_T{T}_ref32_vs_native32_max"] = np.float64((xa - xb).abs().max())
    save("g9_featurenet_noise", **arrays)
    for k, v in arrays.items():
        if k.endswith("_max"):
            print(f"  {k}: {float(v):.2e}")


@torch.no_grad()
def g6_forward():
    for tag, (refine, H, W) in {"norefine": (False, 128, 160), "refine": (True, 128, 192)}.items():
        model = ref_model(refine)
        N = 3
        imgs = synth.make_images(N, H, W, seed=6)
        cams = synth.make_cameras(N, H, W, refine=refine, seed=6)
        dv = synth.make_depth_values()
        hyps = []

        def pre_hook(mod, args, kwargs, store=hyps):
            store.append(kwargs["depth_values"].clone())

        hk = model.stage_net.register_forward_pre_hook(pre_hook, with_kwargs=True)
        out = model(imgs, cams, dv, temperature=0.01)
        hk.remove()
        arrays = {"imgs": imgs, "depth_values": dv, "refined_depth": out["refined_depth"]}
        for k, v in cams.items():
            arrays["cam_" + k] = v
        for s in range(3):
            st = out[f"stage{s + 1}"]
            arrays[f"stage{s + 1}_depth"] = st["depth"]
            arrays[f"stage{s + 1}_conf"] = st["photometric_confidence"]
            arrays[f"stage{s + 1}_norm_curv"] = st["norm_curv"]
            arrays[f"stage{s + 1}_hyp"] = hyps[s][0, :, ::4, ::4]  # sub-sampled: pins the hypothesis generator
        save(f"g6_forward_{tag}", **arrays)


G7_FULL_GRADIENTS = ("feature.conv01.conv.att_convs.1.weight", "feature.conv01.conv.convs.2.weight",
                     "feature.conv10.conv.att_weights.0.weight", "feature.conv10.conv.att_weights.3.weight",
                     "feature.conv01.conv.att_weights.1.weight", "feature.out1.convs.1.weight",
                     "cost_regularization.0.conv4.conv.weight", "cost_regularization.0.conv4.bn.weight",
                     "stage_net.vis.1.0.conv.weight")


def g7_training_step():
    """Reference training step on CPU (model.train(), gt depths, final_loss, backward): loss and per-parameter gradient
    norms.  B = 2 so every BatchNorm sees real batch statistics."""
    from models.losses import final_loss as ref_loss
    torch.manual_seed(0)
    B, N, H, W = 2, 3, 64, 96
    model = RefNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0))
    seeded_init_(model, SEED)
    model.train()
    imgs = torch.cat([synth.make_images(N, H, W, seed=20 + b) for b in range(B)])
    cams_l = [synth.make_cameras(N, H, W, refine=False, seed=20 + b) for b in range(B)]
    cams = {k: torch.cat([c[k] for c in cams_l]) for k in cams_l[0]}
    dv = synth.make_depth_values().repeat(B, 1)
    g = torch.Generator().manual_seed(9)
    gt, mask = {}, {}
    base = 600.0 + 120.0 * F.interpolate(torch.rand(B, 1, 4, 6, generator=g), (H, W), mode="bicubic", align_corners=False)[:, 0]
    for s, sc in (("stage1", 4), ("stage2", 2), ("stage3", 1)):
        gt[s] = F.interpolate(base.unsqueeze(1), (H // sc, W // sc), mode="nearest")[:, 0].contiguous()
        mask[s] = (torch.rand(B, H // sc, W // sc, generator=g) > 0.15).float()
    gt["stage4"], mask["stage4"] = gt["stage3"], mask["stage3"]   # refine=False: refined_depth is the stage-3 depth
    out = model(imgs, cams, dv, gt_depths=gt, temperature=0.1)
    loss, depth_loss = ref_loss(out, gt, mask, dlossw=[0.5, 1.0, 2.0], depth_interval=dv[:, 1] - dv[:, 0])
    loss.backward()
    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else float(p.grad.norm()))
    arrays = {"imgs": imgs, "depth_values": dv, "loss": loss.detach(), "depth_loss": depth_loss.detach(),
              "grad_norms": np.array(norms, dtype=np.float64), "param_names": np.array(names),
              "stage3_depth": out["stage3"]["depth"].detach(), "stage1_feat_distance_mean": out["stage1"]["feat_distance"].detach().mean(),
              "grad_prob3": model.cost_regularization[2].prob.weight.grad.clone(),
              "grad_vis0": model.stage_net.vis[0][3].weight.grad.clone()}
    # full gradient tensors of one parameter of every kind of layer the step trains (norms alone do not pin a
    # gradient's direction): a DynamicConv curvature branch, a feature branch, the attention MLP's two 1x1 layers and its BatchNorm2d,
    # a CostRegNet convolution and its BatchNorm3d
    params = dict(model.named_parameters())
    for n in G7_FULL_GRADIENTS:
        arrays["fullgrad:" + n] = params[n].grad.clone()
    for k, v in cams.items():
        arrays["cam_" + k] = v
    for s in gt:
        arrays["gt_" + s] = gt[s]
        arrays["mask_" + s] = mask[s]
    save("g7_training_step", **arrays)
    print("loss", float(loss), "depth_loss", float(depth_loss), "zero-grad params", sum(1 for x in norms if x == 0.0), "of", len(norms))


def g8_fusion():
    """Reference depth filtering / fusion (fusion.py, driven as test.py:334-351 drives it) on a synthetic consistent
    scene.  fusion.py builds its pixel grids with ``.cuda()``; there is no GPU here, so ``Tensor.cuda`` is made an
    identity for the duration of the call (CPU fp32 run of the same functions)."""
    import fusion as ref_fusion
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        sc = synth.make_fusion_scene(5, 96, 128, seed=3)
        conf, thres_disp, thres_view = [0.1, 0.05, 0.02], 1.0, 3
        ref_depth = sc["depths"][0][None, None]                 # n1hw
        ref_conf = sc["confs"][0][None]                          # n3hw
        src_depths = sc["depths"][1:][None, :, None].clone()    # nv1hw
        src_confs = sc["confs"][1:][None]                        # nv3hw
        ref_cam, src_cams = sc["cams"][0][None], sc["cams"][1:][None]
        for ids in range(src_depths.size(1)):
            m = ref_fusion.prob_filter(src_confs[:, ids, ...], conf)
            src_depths[:, ids, ...] *= m.float()
        prob_mask = ref_fusion.prob_filter(ref_conf, conf)
        reproj_xyd, in_range = ref_fusion.get_reproj(ref_depth, src_depths, ref_cam, src_cams)
        vis_masks, vis_mask = ref_fusion.vis_filter(ref_depth, reproj_xyd, in_range, thres_disp, 0.01, thres_view)
        ave = ref_fusion.ave_fusion(ref_depth, reproj_xyd, vis_masks)
        mask = ref_fusion.bin_op_reduce([prob_mask, vis_mask], torch.min)
        idx_img = ref_fusion.get_pixel_grids(*ave.size()[-2:]).unsqueeze(0)
        idx_cam = ref_fusion.idx_img2cam(idx_img, ave, ref_cam)
        points = ref_fusion.idx_cam2world(idx_cam, ref_cam)[..., :3, 0].permute(0, 3, 1, 2)
    finally:
        torch.Tensor.cuda = orig
    save("g8_fusion", depths=sc["depths"], confs=sc["confs"], cams=sc["cams"], conf=np.array(conf, np.float32),
         thres_disp=thres_disp, thres_view=thres_view, fused=ave[0, 0], mask=mask[0, 0].float(), points=points[0],
         view_masks=vis_masks[0, :, 0], reproj_xyd=reproj_xyd[0])
    print("photo/geo/final", float(prob_mask.float().mean()), float(vis_mask.float().mean()), float(mask.float().mean()),
          "per-view", vis_masks[0, :, 0].mean(dim=(1, 2)))


def _reference_function(path, name, extra_globals=None):
    """Compile ONE top-level function of a reference script that cannot be imported as a module (test.py parses
    ``sys.argv`` and imports cv2 / plyfile at import time) straight from the file where it lies.  Nothing is copied:
    the function object is built from the reference's own source at capture time."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    glb = {"np": np, "__builtins__": __builtins__}
    glb.update(extra_globals or {})
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), glb)
    return glb[name]


def g10_formats():
    """On-disk formats written / parsed BY THE REFERENCE: PFM bytes from ``datasets/data_io.save_pfm``,
    arrays from its ``read_pfm``, the camera text from ``test.py:write_cam``, and the sample dict its ``MVSDataset``
    (``datasets/general_eval.py``) builds from a small synthetic MVSNet-format scene (dtu and tt layouts, refine on/off).
    ``cv2`` does not exist in this image: it is stubbed in ``sys.modules`` (as torchvision is for the checkpoints); the only
    cv2 call on this path, ``cv2.resize`` to the size the image already has, is an identity checked by the stub."""
    import io
    import tempfile
    import types
    from PIL import Image

    cv2 = types.ModuleType("cv2")

    def _resize(img, size, interpolation=None):
        assert (img.shape[1], img.shape[0]) == tuple(size), "the fixture never resizes (cv2 is absent)"
        return img
    cv2.resize = _resize
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    sys.modules.setdefault("cv2", cv2)
    from datasets import data_io as ref_io                        # reference
    from datasets.general_eval import MVSDataset as RefDataset    # reference

    out = {}
    rs = np.random.RandomState(10)
    tmp = tempfile.mkdtemp()
    # ---- PFM ----
    grey = (rs.rand(5, 7).astype(np.float32) * 900 + 400)
    col = rs.rand(6, 4, 3).astype(np.float32)
    one = rs.rand(3, 5, 1).astype(np.float32)
    for tag, arr in (("grey", grey), ("color", col), ("hw1", one)):
        pth = os.path.join(tmp, tag + ".pfm")
        ref_io.save_pfm(pth, arr)
        out[f"pfm_{tag}_array"] = arr
        out[f"pfm_{tag}_bytes"] = np.frombuffer(open(pth, "rb").read(), dtype=np.uint8)
        back, scale = ref_io.read_pfm(pth) if tag != "hw1" else (None, None)
        if back is not None:
            out[f"pfm_{tag}_read"] = np.ascontiguousarray(back)
            out[f"pfm_{tag}_scale"] = np.float64(scale)
    be = os.path.join(tmp, "be.pfm")                             # a big-endian file with scale 2.5, as another tool writes it
    with open(be, "wb") as f:
        f.write(b"Pf\n3 2\n2.5\n" + np.arange(6, dtype=">f4").tobytes())
    back, scale = ref_io.read_pfm(be)
    out["pfm_be_bytes"] = np.frombuffer(open(be, "rb").read(), dtype=np.uint8)
    out["pfm_be_read"] = np.ascontiguousarray(back).astype(np.float32)
    out["pfm_be_scale"] = np.float64(scale)
    # ---- camera text written by the reference's inference script ----
    write_cam = _reference_function("/root/reference/test.py", "write_cam")
    read_params = _reference_function("/root/reference/test.py", "read_camera_parameters")
    cam = np.zeros((2, 4, 4), np.float32)
    cam[0] = np.eye(4, dtype=np.float32)
    cam[0, :3, :3] += (rs.rand(3, 3).astype(np.float32) - 0.5) * 0.1
    cam[0, :3, 3] = rs.rand(3).astype(np.float32) * 100 - 50
    cam[1, :3, :3] = np.array([[361.54125, 0, 82.900625], [0, 360.3975, 66.383875], [0, 0, 1]], np.float32)
    pth = os.path.join(tmp, "w_cam.txt")
    write_cam(pth, cam)
    out["cam_array"] = cam
    out["cam_bytes"] = np.frombuffer(open(pth, "rb").read(), dtype=np.uint8)
    intr, extr = read_params(pth)
    out["cam_read_intrinsic"], out["cam_read_extrinsic"] = intr, extr
    # ---- a scene parsed by the reference's evaluation dataset ----
    names, blobs = [], []
    for ds_name, H, W, depth_line in (("dtu", 64, 80, "425.0 2.5"), ("tt", 56, 80, "0.35 0.0125 96 1.55")):
        root = os.path.join(tmp, ds_name)
        scan = "scan_" + ds_name
        os.makedirs(os.path.join(root, scan, "images"))
        os.makedirs(os.path.join(root, scan, "cams"))
        nv = 4
        for v in range(nv):
            img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, scan, "images", f"{v:08d}.jpg"), quality=95)
            R = np.eye(4)
            R[:3, :3] += (rs.rand(3, 3) - 0.5) * 0.05
            R[:3, 3] = rs.rand(3) * 60 - 30
            K = np.array([[2892.33 / 16 + v, 0, W / 2 + 0.37], [0, 2883.18 / 16 - v, H / 2 - 0.21], [0, 0, 1]])
            with open(os.path.join(root, scan, "cams", f"{v:08d}_cam.txt"), "w") as f:
                f.write("extrinsic\n" + "\n".join(" ".join(repr(float(x)) for x in r) for r in R) + "\n\nintrinsic\n")
                f.write("\n".join(" ".join(repr(float(x)) for x in r) for r in K) + "\n\n" + depth_line + "\n")
        with open(os.path.join(root, scan, "pair.txt"), "w") as f:
            f.write(f"{nv}\n")
            for v in range(nv):
                others = [u for u in range(nv) if u != v]
                f.write(f"{v}\n{len(others)} " + " ".join(f"{u} {1000.0 / (1 + abs(u - v)):.3f}" for u in others) + "\n")
        for dirpath, _, files in os.walk(root):
            for fn in sorted(files):
                full = os.path.join(dirpath, fn)
                names.append(os.path.relpath(full, tmp))
                blobs.append(np.frombuffer(open(full, "rb").read(), dtype=np.uint8))
        Hm = 64                                                  # tt: 56 rows + 4 + 4 edge padding
        for refine in (False, True):
            ds = RefDataset(root, [scan], "test", 4, ndepths=192, interval_scale=1.06, max_h=Hm, max_w=W, fix_res=False,
                            dataset=ds_name, refine=refine)
            assert len(ds) == nv
            for idx in (0, 2):
                smp = ds[idx]
                key = f"scene_{ds_name}_{'refine' if refine else 'norefine'}_{idx}"
                out[key + "_imgs"] = smp["imgs"].astype(np.float32)
                for st, m in smp["proj_matrices"].items():
                    out[key + "_proj_" + st] = m
                out[key + "_depth_values"] = smp["depth_values"]
                out[key + "_filename"] = np.array(smp["filename"])
    out["scene_file_names"] = np.array(names)
    for i, b in enumerate(blobs):
        out[f"scene_file_{i}"] = b
    save("g10_formats", **out)


def g11_loss():
    """models/losses.py:6-48 (final_loss) on random stage outputs: the loss, the last depth loss and the gradient of every input
    (two batch items, ragged masks, with and without stage weights / the refined depth)."""
    from models.losses import final_loss as ref_loss
    g = torch.Generator().manual_seed(SEED + 11)
    shapes = {"stage1": (8, 12, 9), "stage2": (16, 24, 7), "stage3": (32, 48, 5)}
    out = {}
    inp, gt, mask = {}, {}, {}
    for k, (h, w, D) in shapes.items():
        inp[k] = {"depth": (500 + 50 * torch.rand(2, h, w, generator=g)).requires_grad_(True),
                  "norm_curv": torch.rand(2, 1, h, w, generator=g).requires_grad_(True),
                  "feat_distance": torch.randn(2, D, h, w, generator=g).requires_grad_(True),
                  "feat_target": (torch.rand(2, D, h, w, generator=g) > 0.8).float()}
        gt[k] = 500 + 50 * torch.rand(2, h, w, generator=g)
        mask[k] = (torch.rand(2, h, w, generator=g) > 0.3).float()
    inp["refined_depth"] = (500 + 50 * torch.rand(2, 32, 48, generator=g)).requires_grad_(True)
    gt["stage4"] = 500 + 50 * torch.rand(2, 32, 48, generator=g)
    mask["stage4"] = (torch.rand(2, 32, 48, generator=g) > 0.3).float()
    interval = torch.tensor([2.5, 2.0])
    for k in shapes:
        for n in ("depth", "norm_curv", "feat_distance", "feat_target"):
            out[f"{k}.{n}"] = inp[k][n].detach().clone()
        out[f"{k}.gt"], out[f"{k}.mask"] = gt[k], mask[k]
    out["refined_depth"], out["stage4.gt"], out["stage4.mask"], out["interval"] = inp["refined_depth"].detach().clone(), gt["stage4"], mask["stage4"], interval
    for tag, kw in (("w", dict(dlossw=[0.5, 1.0, 2.0])), ("nw", dict())):
        for t in [inp[k][n] for k in shapes for n in ("depth", "norm_curv", "feat_distance")] + [inp["refined_depth"]]:
            t.grad = None
        loss, dl = ref_loss(inp, gt, mask, depth_interval=interval, **kw)
        loss.backward()
        out[f"{tag}.loss"], out[f"{tag}.depth_loss"] = loss.detach().reshape(1), dl.detach().reshape(1)
        for k in shapes:
            for n in ("depth", "norm_curv", "feat_distance"):
                out[f"{tag}.grad.{k}.{n}"] = inp[k][n].grad.clone()
        out[f"{tag}.grad.refined_depth"] = inp["refined_depth"].grad.clone()
    save("g11_loss", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_warp_aggregate, g2_costreg, g3_regress, g4_hypotheses, g5_features, g6_forward, g7_training_step,
               g8_fusion, g9_feature_noise, g10_formats, g11_loss):
        if not only or fn.__name__.split("_")[0] in only:
            fn()
