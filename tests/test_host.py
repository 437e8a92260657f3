"""CPU-only checks of the host side: state-dict layout, the C-ABI library (loads, exports every symbol the
header declares — no compute calls without a GPU), camera algebra, loud failure on CPU tensors."""
import ctypes
import json
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden


def test_state_dict_matches_reference_layout():
    from cds_mvsnet_amd import CDSMVSNet
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    m = CDSMVSNet(refine=True, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 1.5, 0.75))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(mine.keys()) == list(ref.keys())
    assert mine == ref
    assert sum(p.numel() for p in m.parameters()) == 981622
    # refine=False drops exactly the refine_network entries
    m2 = CDSMVSNet(refine=False)
    assert set(m2.state_dict().keys()) == {k for k in ref if not k.startswith("refine_network.")}


def test_checkpoint_roundtrip_with_module_prefix(tmp_path):
    """Checkpoints saved under DataParallel carry a 'module.' prefix that the harness strips (test.py:182-184)."""
    from cds_mvsnet_amd import CDSMVSNet, seeded_init_
    a = seeded_init_(CDSMVSNet(refine=True), 3)
    ck = {"state_dict": {"module." + k: v for k, v in a.state_dict().items()}}
    torch.save(ck, tmp_path / "ck.pth")
    sd = {k[len("module."):]: v for k, v in torch.load(tmp_path / "ck.pth")["state_dict"].items()}
    b = CDSMVSNet(refine=True)
    missing, unexpected = b.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


def test_library_exports_every_declared_symbol():
    from cds_mvsnet_amd import _lib
    header = open(os.path.join(ROOT, "include", "cds_mvsnet_hip.h")).read()
    declared = set(re.findall(r"^int\s+(cds_\w+)\s*\(", header, re.M))
    assert declared == set(_lib.SIGNATURES.keys())
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # argument counts in the ctypes table match the header prototypes
    for name, args in re.findall(r"^int\s+(cds_\w+)\s*\(([^;]*?)\);", header, re.M | re.S):
        n = 0 if args.strip() == "void" else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name]), name
    assert _lib.load().cds_version() >= 100


def test_geometry_matches_reference_numbers():
    from cds_mvsnet_amd import geometry
    g = load_golden("g1_warp_aggregate_c")
    assert torch.equal(geometry.warp_matrices(g["cams"][0]), g["mats"])
    g5 = load_golden("g5_dynconv")
    Fm = geometry.fundamental(g5["cams"][0, 0], g5["cams"][0, 1])
    assert torch.equal(Fm, g5["fmatrix"])
    e_ref, e_src = geometry.pair_epipoles(g5["cams"][0, 0], g5["cams"][0, 1])
    assert e_ref == (float(g5["epipole_ref"][0, 0]), float(g5["epipole_ref"][0, 1]))
    assert e_src == (float(g5["epipole_src"][0, 0]), float(g5["epipole_src"][0, 1]))


def test_no_cpu_fallback():
    from cds_mvsnet_amd import CDSMVSNet, ops, synth
    with pytest.raises(RuntimeError):
        ops.chw_to_hwc(torch.zeros(8, 4, 4))
    m = CDSMVSNet().eval()
    imgs = synth.make_images(3, 64, 64)
    with pytest.raises(RuntimeError):
        m(imgs, synth.make_cameras(3, 64, 64), synth.make_depth_values())
    with pytest.raises(RuntimeError):
        m.train()(imgs, synth.make_cameras(3, 64, 64), synth.make_depth_values())


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "cds_mvsnet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("SURVEY", ""), fn
            assert "/root/reference" not in src, fn
    # helper scripts outside tests/ (timing, profiling, variant builds) must not touch the oracle either: the fuzzers
    # that compare against it live under tests/tools/
    scripts = os.path.join(ROOT, "scripts")
    for fn in os.listdir(scripts):
        if fn.endswith((".py", ".sh")):
            assert "oracle" not in open(os.path.join(scripts, fn)).read(), fn


def test_synthetic_inputs_are_deterministic():
    from cds_mvsnet_amd import synth
    a, b = synth.make_cameras(5, 512, 640, seed=0), synth.make_cameras(5, 512, 640, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(a["stage1"][0, :, 1, :2, :3] * 4, a["stage3"][0, :, 1, :2, :3])
    f1 = synth.make_pair_features(2, 8, 16, 24, seed=1)
    f2 = synth.make_pair_features(2, 8, 16, 24, seed=1)
    assert torch.equal(f1[1]["src"][0], f2[1]["src"][0]) and f1[0]["ref"][0].abs().max() < 1


def _unsplit(packed):
    """int16 [..., 3, 64, 8] split-bf16 operand -> float [..., 64, 8] (hi + mid + lo)."""
    return packed.view(torch.bfloat16).float().sum(dim=-3)


def test_split_bf16_packers_reconstruct_the_weights_exactly():
    """Host side of the split-bf16 kernels (ops.split_pack_*): the three bf16 terms sum back to the fp32 weight bit for bit,
    and every (round, K-step, block, lane group, row) slot holds the tap / channel the kernels' address tables expect."""
    from cds_mvsnet_amd import ops
    g = torch.Generator().manual_seed(0)
    # 3D convolution: lane l = 16 g + i of K-step t multiplies cout 16 mb + i by tap 4 t + g, channels 8 rd + j
    w = torch.randn(24, 16, 3, 3, 3, generator=g)
    a = _unsplit(ops.split_pack_conv3d(w))                        # [rd][t][mb][64][8]
    assert a.shape == (2, 7, 2, 64, 8)
    for rd, t, mb, gg, i in ((0, 0, 0, 0, 0), (1, 3, 1, 2, 5), (1, 6, 0, 2, 15), (0, 5, 1, 3, 7)):
        tap, co = 4 * t + gg, 16 * mb + i
        want = w[co, 8 * rd:8 * rd + 8].reshape(8, 27)[:, tap] if (tap < 27 and co < 24) else torch.zeros(8)
        assert torch.equal(a[rd, t, mb, 16 * gg + i], want), (rd, t, mb, gg, i)
    assert a[:, 6, :, 48:].abs().max() == 0                       # tap 27 is the zero pad
    assert a[:, :, 1, :, :].reshape(2, 7, 4, 16, 8)[:, :, :, 8:].abs().max() == 0     # couts 24..31 of block 1
    # voxel-pair packing (Cout = 8): row i = 8 p + co, K-step = (kz, ky), lane group g -> x' = (0, 2, 1, 3)[g]
    w8 = torch.randn(8, 8, 3, 3, 3, generator=g)
    p = _unsplit(ops.split_pack_conv3d_pair(w8))                  # [1][9][1][64][8]
    for kz, ky, gg, par, co in ((0, 0, 0, 0, 3), (1, 2, 1, 1, 7), (2, 1, 3, 0, 0), (2, 2, 2, 1, 4), (0, 1, 3, 1, 2)):
        xp = (0, 2, 1, 3)[gg]
        kx = xp - par
        want = w8[co, :, kz, ky, kx] if 0 <= kx <= 2 else torch.zeros(8)
        assert torch.equal(p[0, kz * 3 + ky, 0, 16 * gg + 8 * par + co], want), (kz, ky, gg, par, co)
    # transposed convolution, Cout = 16: class (pz, py, px) K-steps, parity 1 axes take (k=2, d=0) then (k=0, d=1)
    wt = torch.randn(8, 16, 3, 3, 3, generator=g)
    d = _unsplit(ops.split_pack_deconv3d(wt))                     # [1][9][1][64][8]
    assert d.shape == (1, 9, 1, 64, 8)
    assert torch.equal(d[0, 0, 0, 5], wt[:, 5, 1, 1, 1])          # class (0,0,0): the single tap (1,1,1)
    assert d[0, 0, 0, 16:].abs().max() == 0                       # its three padded slots
    assert torch.equal(d[0, 7, 0, 16 * 3 + 2], wt[:, 2, 2, 0, 0])  # class (1,1,1), slot 3 = (iz, iy, ix) = (0, 1, 1) -> k = (2, 0, 0)
    assert torch.equal(d[0, 8, 0, 16 * 0 + 9], wt[:, 9, 0, 2, 2])  # second K-step of class 7, slot 4 = (1, 0, 0) -> k = (0, 2, 2)
    # DynamicConv branches: K-steps of branch b start at ks0_b, tap = ky * k + kx
    ws = [torch.randn(11, 8, k, k, generator=g) for k in (3, 5, 7)]
    q = _unsplit(ops.split_pack_dynconv(ws))                      # [1][3 + 7 + 13][1][64][8]
    assert q.shape == (1, 23, 1, 64, 8)
    assert torch.equal(q[0, 3 + 2, 0, 16 * 1 + 4], ws[1][4, :, 1, 4])       # 5x5: tap 4*2+1 = 9 -> (ky, kx) = (1, 4)
    assert torch.equal(q[0, 10 + 12, 0, 16 * 0 + 10], ws[2][10, :, 6, 6])   # 7x7: tap 48 -> (6, 6)
    assert q[0, 10 + 12, 0, 16:].abs().max() == 0                           # taps 49..51 padded
    # exactness of the three-term split on awkward values
    x = torch.tensor([1.0, 1 + 2 ** -23, 3.1415927, -1e-30, 6.5e37, 2 ** -120, 0.0]).reshape(1, 1, 7).expand(1, 64, 7)
    x = torch.cat((x, torch.zeros(1, 64, 1)), dim=-1).contiguous()
    assert torch.equal(_unsplit(ops._split3(x)), x)


def test_fused_conv11_prob_operands_compute_the_reference_layers():
    """Host side of csrc/deconv_prob_zm.hip: the five matrix operands per round from ops.split_pack_deconv_prob, multiplied the way
    the kernel's K-steps read the input cells (half-step 1: slots (dy, dx) of cell plane a; half-step 2: slots (dz, dx) at dy = 0 and
    at dy = 1; rows = (x parity, cout)), reproduce ConvTranspose3d(16 -> 8, k3 s2 p1 op1) of the reference (models/module.py:125-160),
    and the prob table of ops.pack_prob_table read as [kx][half][ky][kz][4] reproduces Conv3d(8 -> 1, k3, p1) (module.py:499)."""
    import torch.nn.functional as F
    from cds_mvsnet_amd import ops
    g = torch.Generator().manual_seed(3)
    D, H, W = 2, 3, 4
    x = torch.randn(16, D, H, W, generator=g).double()
    w = torch.randn(16, 8, 3, 3, 3, generator=g)
    a = _unsplit(ops.split_pack_deconv_prob(w)).double()           # [rd][operand][64 = 16 g + row][8 ci]
    assert a.shape == (2, 5, 64, 8)
    xp = F.pad(x, (0, 1, 0, 1, 0, 1))                              # cells beyond the volume are zero
    def cells(dz, dy, dx):                                         # [16][D][H][W]: cell (a + dz, y + dy, x + dx)
        return xp[:, dz:dz + D, dy:dy + H, dx:dx + W]
    out = torch.zeros(8, 2 * D, 2 * H, 2 * W, dtype=torch.float64)
    # (operand, [slot g -> (dz, dy, dx)], z parity, y parity)
    plan = [(0, lambda gg: (0, gg >> 1, gg & 1), 0, 0), (1, lambda gg: (0, gg >> 1, gg & 1), 0, 1),
            (2, lambda gg: (gg >> 1, 0, gg & 1), 1, 0), (3, lambda gg: (gg >> 1, 0, gg & 1), 1, 1),
            (4, lambda gg: (gg >> 1, 1, gg & 1), 1, 1)]
    for k, slot, pz, py in plan:
        for rd in range(2):
            for gg in range(4):
                b = cells(*slot(gg))[8 * rd:8 * rd + 8]                         # [8 ci][D][H][W]
                rows = a[rd, k, 16 * gg:16 * gg + 16]                           # [16 rows = 8 px + co][8 ci]
                r = torch.einsum("ic,cdhw->idhw", rows, b)                      # [16][D][H][W]
                for px in range(2):
                    out[:, pz::2, py::2, px::2] += r[8 * px:8 * px + 8]
    want = F.conv_transpose3d(x[None], w.double(), stride=2, padding=1, output_padding=1)[0]
    assert (out - want).abs().max().item() < 1e-12
    # prob table
    wp = torch.randn(1, 8, 3, 3, 3, generator=g)
    tab = ops.pack_prob_table(wp)
    assert tab.shape == (3, 2, 3, 3, 4)
    for kx, hh, ky, kz, i in ((0, 0, 0, 0, 0), (2, 1, 1, 0, 3), (1, 0, 2, 2, 1), (2, 1, 2, 1, 2)):
        assert tab[kx, hh, ky, kz, i] == wp[0, 4 * hh + i, kz, ky, kx]


def test_restricted_unpickler_does_not_resolve_load_from_bytes():
    """`torch.storage._load_from_bytes` is a full `torch.load(..., weights_only=False)` in disguise; the allowlisted
    unpickler of infer.load_checkpoint must turn it (like every other non-tensor global) into an inert placeholder."""
    import io
    import pickle
    from cds_mvsnet_amd.infer import _Opaque, _placeholder_pickle
    up = _placeholder_pickle.Unpickler(io.BytesIO(b""))
    assert up.find_class("torch.storage", "_load_from_bytes") is _Opaque
    assert up.find_class("os", "system") is _Opaque
    assert up.find_class("torch.hub", "load") is _Opaque
    assert up.find_class("collections", "OrderedDict") is __import__("collections").OrderedDict
    # a pickle stream that tries to CALL it only builds a placeholder
    class Evil:
        def __reduce__(self):
            import torch.storage
            return (torch.storage._load_from_bytes, (b"not a checkpoint",))
    obj = _placeholder_pickle.load(io.BytesIO(pickle.dumps(Evil())))
    assert isinstance(obj, _Opaque)

