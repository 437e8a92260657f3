"""Tensor-level wrappers around the C ABI (include/cds_mvsnet_hip.h).

PyTorch is used here only for device memory (allocation through the caching allocator) and for the
current HIP stream; all arithmetic happens in libcdsmvs_hip.so.  Every function requires fp32,
contiguous, ROCm-resident tensors and raises otherwise — there is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (ACT_LEAKY01, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, AGG_ACCUMULATE, AGG_CHANNELS_LAST, AGG_FAST_POSITIONS,
                   AGG_NORMALIZE,
                   MAX_IMAGES, MAX_VIEWS, check)

Tensor = torch.Tensor

# Optional per-op timing with HIP events on the launch stream (bench.py turns it on inside its timed region;
# events are recorded asynchronously, nothing synchronises here).
PROFILE_ON = False
PROFILE: dict = {}
PROFILE_HOST_MS: dict = {}     # host-timed sections (gloo dry runs of the communication paths): name -> [ms, ...]


class prof:
    """``with prof("name"):`` brackets the enclosed launches with two events when PROFILE_ON."""

    def __init__(self, name: str):
        self.name = name
        self.start = None

    def __enter__(self):
        if PROFILE_ON:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            PROFILE.setdefault(self.name, []).append((self.start, end))
        return False


# the current device index without torch.cuda.current_device()'s lazy-init bookkeeping (~1 000 calls per training step)
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _dev(t: Tensor, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: cds_mvsnet_amd ops need a ROCm (cuda) tensor; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if t.device.index != _current_device():
        raise RuntimeError(f"{name}: tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                           "(the launch goes to the current device; use `with torch.cuda.device(t.device):`)")
    return t.data_ptr()


def _dev64(t: Tensor, name: str) -> int:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float64 device tensor")
    return t.data_ptr()


def _host(t: Tensor, name: str) -> int:
    if t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 CPU tensor")
    return t.data_ptr()


def geo(t, device, name: str = "geometry") -> Tensor:
    """Per-call GEOMETRY operand (homographies [V,12], epipoles [N,2], depth-range scalars) as the float32 DEVICE tensor the kernels
    read.  Since round 6 these are device data, not by-value kernel arguments: the model writes all of a forward's geometry into one
    block with one host -> device copy (geometry.GeoBlock) and hands out slices of it, and a captured hipGraph is replayed for new
    cameras by rewriting that block.  Direct callers (tests, scripts) may still pass a CPU tensor or a list of Python floats: it is
    uploaded here through pinned memory (asynchronous, the host does not wait for the stream) - which a stream capture must not
    contain (the captured copy would replay stale host bytes), hence the error there."""
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return t
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"{name}: host-side geometry inside a stream capture - build a geometry.GeoBlock before capturing and "
                           "pass its device slices")
    if not isinstance(t, torch.Tensor):
        t = torch.tensor([float(v) for v in t], dtype=torch.float32)
    return t.detach().to(dtype=torch.float32).contiguous().pin_memory().to(device=device, non_blocking=True)


# the raw hipStream_t of the current stream without building a torch.cuda.Stream object (5 us -> 0.5 us per launch: the training step is
# CPU-bound at ~1 500 launches)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _stream(t: Tensor) -> int:
    """The current HIP stream of t's device.  The C entry points launch on the calling thread's current device, so that
    device must be t's: the model's forward makes it so (``torch.cuda.device``); direct callers get a clear error
    instead of a launch on the wrong GPU."""
    idx = t.device.index
    cur = torch.cuda.current_device()
    if idx is not None and idx != cur:
        raise RuntimeError(f"cds_mvsnet_amd ops launch on the current device (cuda:{cur}) but the "
                           f"tensors live on {t.device}: wrap the call in `with torch.cuda.device(t.device):`")
    return _raw_stream(cur)


def version() -> int:
    return _lib.load().cds_version()


def chw_to_hwc(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """[C,h,w] -> channels-last [h,w,C] (the layout K1 / K3 gather from); written into `out` if given."""
    C, h, w = x.shape
    if out is None:
        out = torch.empty((h, w, C), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (h, w, C) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device:
        raise ValueError(f"chw_to_hwc: out must be a contiguous float32 [{h},{w},{C}] tensor on {x.device}")
    check(_lib.load().cds_chw_to_hwc_f32(_dev(x, "x"), out.data_ptr(), C, h, w, _stream(x)), "cds_chw_to_hwc_f32")
    return out


USE_STAGE_INPUTS = os.environ.get("CDS_STAGE_INPUTS", "1") != "0"     # A/B knob: 0 = the stack / transpose / amax launches


def stage_inputs_supported(ref_feas, src_feas, maps=()) -> bool:
    """Whether :func:`stage_inputs` covers these per-view tensors (contiguous fp32 maps on one CUDA device, C in 8 / 16 / 32)."""
    V = len(ref_feas)
    if not (USE_STAGE_INPUTS and 1 <= V <= MAX_VIEWS and len(src_feas) == V and all(len(m) == V for m in maps)):
        return False
    r0 = ref_feas[0]
    if r0.dim() != 3 or r0.shape[0] not in (8, 16, 32) or not r0.is_cuda:
        return False
    ok = lambda t, shape: (t.is_cuda and t.device == r0.device and t.dtype == torch.float32 and t.is_contiguous()
                           and tuple(t.shape) == shape)
    return (all(ok(t, tuple(r0.shape)) for t in list(ref_feas) + list(src_feas))
            and all(ok(t, tuple(r0.shape[1:])) for m in maps for t in m))


def stage_inputs(ref_feas, src_feas, ref_nc=None, ref_ncsum=None, src_ncsum=None, want_bound: bool = True):
    """The reference's per-view feature dicts (models/model.py:16-40) gathered for K1 / K3 in ONE launch (cds_stage_inputs_f32):
    lists over the V source views of [C,h,w] reference-copy / source features (+ optionally the [h,w] curvature maps) ->
    (ref_chw [V,C,h,w], src_hwc [V,h,w,C], ref_nc [V,h,w] | None, nc_mean [h,w] | None, bound [1] | None) where
    bound = max |ref| * max |src| (what ``ref.abs().amax() * src.abs().amax()`` gives: the split-f16 CostRegNet's volume bound)."""
    maps = [m for m in (ref_nc, ref_ncsum, src_ncsum) if m is not None]
    if (ref_ncsum is None) != (src_ncsum is None) or not stage_inputs_supported(ref_feas, src_feas, maps):
        raise ValueError("stage_inputs: V <= MAX_VIEWS contiguous float32 CUDA maps of one shape, C in (8, 16, 32)")
    V = len(ref_feas)
    C, h, w = ref_feas[0].shape
    dev = ref_feas[0].device
    arr = lambda ts: (ctypes.c_void_p * V)(*[t.data_ptr() for t in ts]) if ts is not None else None
    ref = torch.empty((V, C, h, w), dtype=torch.float32, device=dev)
    src = torch.empty((V, h, w, C), dtype=torch.float32, device=dev)
    nc = torch.empty((V, h, w), dtype=torch.float32, device=dev) if ref_nc is not None else None
    ncm = torch.empty((h, w), dtype=torch.float32, device=dev) if ref_ncsum is not None else None
    state = torch.empty(_lib.STAGE_STATE_WORDS, dtype=torch.float32, device=dev) if want_bound else None
    check(_lib.load().cds_stage_inputs_f32(arr(ref_feas), arr(src_feas), arr(ref_nc), arr(ref_ncsum), arr(src_ncsum), ref.data_ptr(),
                                           src.data_ptr(), nc.data_ptr() if nc is not None else None,
                                           ncm.data_ptr() if ncm is not None else None,
                                           state.data_ptr() if state is not None else None, V, C, h, w, _stream(ref)),
          "cds_stage_inputs_f32")
    return ref, src, nc, ncm, (state[0:1] if state is not None else None)


def _hyp_args(hyp: Tensor, D_expected: Optional[int], h: int, w: int) -> Tuple[int, int]:
    if hyp.dim() == 1:
        return hyp.shape[0], 0
    if hyp.dim() == 3 and hyp.shape[1] == h and hyp.shape[2] == w:
        return hyp.shape[0], 1
    raise ValueError(f"hypotheses must be [D] or [D,{h},{w}], got {tuple(hyp.shape)}")


def homo_warp(src_hwc: Tensor, mat: Tensor, hyp: Tensor) -> Tensor:
    """models/utils/warping.py:69-104 for one view.  src_hwc [h,w,C], mat [12] (device; a CPU tensor is uploaded) -> [C,D,h,w]."""
    h, w, C = src_hwc.shape
    mat = geo(mat, src_hwc.device, "mat")
    D, pp = _hyp_args(hyp, None, h, w)
    out = torch.empty((C, D, h, w), dtype=torch.float32, device=src_hwc.device)
    check(_lib.load().cds_homo_warp_f32(_dev(src_hwc, "src_hwc"), _dev(mat, "mat"), _dev(hyp, "hyp"), out.data_ptr(),
                                        C, D, h, w, pp, _stream(out)), "cds_homo_warp_f32")
    return out


# Sample-position arithmetic of the LDS-staged K1 / K3 kernels (decision and measurements: DESIGN.md section 4).
# Default: the reference's fp32 operation order (true divisions + ATen's normalise / de-normalise round trip), sample positions
# bit-identical to F.grid_sample's.  CDS_WARP_FAST=1 (or exact=False per call) samples at p.xy * rcp(p.z + 1e-6) directly:
# K1 / K3 7-12 % faster, but the positions move by up to ~1e-4 px at w = 640, which moves the volume of SHARP feature maps past the
# 1e-5 parity tolerance (the small goldens, w = 40, do not show it) -- so it stays opt-in.
WARP_EXACT = os.environ.get("CDS_WARP_FAST", "0") != "1"


def _pos_flag(exact: Optional[bool]) -> int:
    return 0 if (WARP_EXACT if exact is None else exact) else AGG_FAST_POSITIONS


def _window_shape_ok(C: int, D: int, h: int, w: int, hs: int) -> None:
    """The row-window kernels are the LDS-staged ones only (no direct-kernel fallback): name the shape instead of a bare EINVAL."""
    if C not in (8, 16, 32) or w < 2 or hs < 2 or D * h * w * 4 >= 1 << 32:
        raise ValueError(f"row-window K1 / K3 (the pixel-slab shard) cover C in (8, 16, 32), w >= 2, a grid of >= 2 rows and D*h*w < 2^30 "
                         f"voxels per window; got C={C}, D={D}, window {h}x{w} of {hs} rows - use exchange='allreduce' for this stage")


def _window_args(window: Optional[Tuple[int, int]], h: int) -> Tuple[int, int]:
    """window = (hs, y_off): the reference-side tensors cover rows [y_off, y_off + h) of an hs-row image grid."""
    if window is None:
        return h, 0
    hs, y_off = int(window[0]), int(window[1])
    if y_off < 0 or y_off + h > hs:
        raise ValueError(f"window rows [{y_off}, {y_off + h}) outside the {hs}-row grid")
    return hs, y_off


def warp_entropy(ref_chw: Tensor, src_hwc: Tensor, mats: Tensor, hyp: Tensor, exact: Optional[bool] = None,
                 window: Optional[Tuple[int, int]] = None) -> Tensor:
    """K1.  ref_chw [V,C,h,w], src_hwc [V,h,w,C], mats [V,12] (device slice of the geometry block; a CPU tensor is uploaded), hyp [D,h,w]|[D] -> entropy [V,h,w].
    exact: sample-position arithmetic (None = the module default ``WARP_EXACT``).
    window = (hs, y_off): ref_chw / hyp / the result cover rows [y_off, y_off + h) of the grid, src_hwc is [V,hs,w,C]."""
    V, C, h, w = ref_chw.shape
    hs, y_off = _window_args(window, h)
    if tuple(src_hwc.shape) != (V, hs, w, C) or tuple(mats.shape) != (V, 12):
        raise ValueError("warp_entropy: inconsistent shapes")
    mats = geo(mats, ref_chw.device, "mats")
    D, pp = _hyp_args(hyp, None, h, w)
    if window is not None and not pp:
        raise ValueError("warp_entropy: a row window needs per-pixel hypotheses [D,h,w]")
    if window is not None:
        _window_shape_ok(C, D, h, w, hs)
    ent = torch.empty((V, h, w), dtype=torch.float32, device=ref_chw.device)
    lib = _lib.load()
    with prof("warp_entropy"):
        for v0 in range(0, V, MAX_VIEWS):
            v1 = min(V, v0 + MAX_VIEWS)
            if window is not None:
                check(lib.cds_warp_entropy_window_f32(_dev(ref_chw[v0:v1], "ref"), _dev(src_hwc[v0:v1], "src"),
                                                      _dev(mats[v0:v1], "mats"), _dev(hyp, "hyp"), ent[v0:v1].data_ptr(),
                                                      v1 - v0, C, D, h, w, hs, y_off, _pos_flag(exact), _stream(ent)),
                      "cds_warp_entropy_window_f32")
                continue
            check(lib.cds_warp_entropy_flags_f32(_dev(ref_chw[v0:v1], "ref"), _dev(src_hwc[v0:v1], "src"),
                                                 _dev(mats[v0:v1], "mats"), _dev(hyp, "hyp"), ent[v0:v1].data_ptr(),
                                                 v1 - v0, C, D, h, w, pp, _pos_flag(exact), _stream(ent)),
                  "cds_warp_entropy_flags_f32")
    return ent


def warp_aggregate(ref_chw: Tensor, src_hwc: Tensor, vis_w: Tensor, mats: Tensor, hyp: Tensor,
                   normalize: bool = True, volume: Optional[Tensor] = None, vis_sum: Optional[Tensor] = None,
                   accumulate: bool = False, channels_last: bool = False,
                   exact: Optional[bool] = None, window: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, Tensor]:
    """K3.  Returns (volume [C,D,h,w] — or [D,h,w,C] with channels_last, the layout the split-bf16 CostRegNet kernels
    read — and vis_sum [h,w]).  With normalize=False the raw visibility-weighted sums are returned (what a source-view
    shard contributes to the all-reduce)."""
    V, C, h, w = ref_chw.shape
    hs, y_off = _window_args(window, h)          # window: see warp_entropy
    if tuple(src_hwc.shape) != (V, hs, w, C) or tuple(mats.shape) != (V, 12) or tuple(vis_w.shape) != (V, h, w):
        raise ValueError("warp_aggregate: inconsistent shapes")
    mats = geo(mats, ref_chw.device, "mats")
    D, pp = _hyp_args(hyp, None, h, w)
    if window is not None and not pp:
        raise ValueError("warp_aggregate: a row window needs per-pixel hypotheses [D,h,w]")
    if window is not None:
        _window_shape_ok(C, D, h, w, hs)
    dev = ref_chw.device
    vshape = (D, h, w, C) if channels_last else (C, D, h, w)
    if volume is None:
        if accumulate:
            raise ValueError("accumulate=True needs an existing volume")
        volume = torch.empty(vshape, dtype=torch.float32, device=dev)
    if vis_sum is None:
        vis_sum = torch.empty((h, w), dtype=torch.float32, device=dev)
    if tuple(volume.shape) != vshape or tuple(vis_sum.shape) != (h, w):
        raise ValueError("warp_aggregate: bad output buffers")
    lib = _lib.load()
    nchunks = (V + MAX_VIEWS - 1) // MAX_VIEWS
    with prof("warp_aggregate"):
        for i, v0 in enumerate(range(0, V, MAX_VIEWS)):
            v1 = min(V, v0 + MAX_VIEWS)
            flags = (AGG_CHANNELS_LAST if channels_last else 0) | _pos_flag(exact)
            if accumulate or i > 0:
                flags |= AGG_ACCUMULATE
            if normalize and i == nchunks - 1:
                flags |= AGG_NORMALIZE
            if window is not None:
                check(lib.cds_warp_aggregate_window_f32(_dev(ref_chw[v0:v1], "ref"), _dev(src_hwc[v0:v1], "src"),
                                                        _dev(vis_w[v0:v1], "vis"), _dev(mats[v0:v1], "mats"), _dev(hyp, "hyp"),
                                                        _dev(volume, "volume"), _dev(vis_sum, "vis_sum"), v1 - v0, C, D, h, w,
                                                        hs, y_off, flags, _stream(volume)), "cds_warp_aggregate_window_f32")
                continue
            check(lib.cds_warp_aggregate_f32(_dev(ref_chw[v0:v1], "ref"), _dev(src_hwc[v0:v1], "src"),
                                             _dev(vis_w[v0:v1], "vis"), _dev(mats[v0:v1], "mats"), _dev(hyp, "hyp"),
                                             _dev(volume, "volume"), _dev(vis_sum, "vis_sum"), v1 - v0, C, D, h, w, pp,
                                             flags, _stream(volume)), "cds_warp_aggregate_f32")
    return volume, vis_sum


def warp_aggregate_bwd(ref_chw: Tensor, src_hwc: Tensor, vis_w: Tensor, mats: Tensor, hyp: Tensor,
                       grad_volume: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Backward of the un-normalised K3: returns (grad_ref [V,C,h,w], grad_src_hwc [V,h,w,C], grad_vis [V,h,w])."""
    V, C, h, w = ref_chw.shape
    D, pp = _hyp_args(hyp, None, h, w)
    if tuple(grad_volume.shape) != (C, D, h, w):
        raise ValueError(f"warp_aggregate_bwd: grad_volume must be {(C, D, h, w)}, got {tuple(grad_volume.shape)}")
    if tuple(src_hwc.shape) != (V, h, w, C) or tuple(mats.shape) != (V, 12) or tuple(vis_w.shape) != (V, h, w):
        raise ValueError("warp_aggregate_bwd: inconsistent shapes")
    mats = geo(mats, ref_chw.device, "mats")
    # the kernel accumulates partial sums (channel groups x depth segments) with atomics: ONE zero fill for the three gradients
    flat = torch.zeros((ref_chw.numel() + src_hwc.numel() + vis_w.numel(),), dtype=torch.float32, device=ref_chw.device)
    g_ref = flat[:ref_chw.numel()].view_as(ref_chw)
    g_src = flat[ref_chw.numel():ref_chw.numel() + src_hwc.numel()].view_as(src_hwc)
    g_vis = flat[ref_chw.numel() + src_hwc.numel():].view_as(vis_w)
    lib = _lib.load()
    # every gradient is per view: groups of MAX_VIEWS views are independent launches (like the forward's chunks)
    for v0 in range(0, V, MAX_VIEWS):
        v1 = min(V, v0 + MAX_VIEWS)
        check(lib.cds_warp_aggregate_bwd_f32(_dev(ref_chw[v0:v1], "ref"), _dev(src_hwc[v0:v1], "src"),
                                             _dev(vis_w[v0:v1], "vis"), _dev(mats[v0:v1], "mats"), _dev(hyp, "hyp"),
                                             _dev(grad_volume, "grad_volume"), g_ref[v0:v1].data_ptr(),
                                             g_src[v0:v1].data_ptr(), g_vis[v0:v1].data_ptr(), v1 - v0, C, D, h, w, pp,
                                             _stream(g_ref)), "cds_warp_aggregate_bwd_f32")
    return g_ref, g_src, g_vis


class WarpAggregate(torch.autograd.Function):
    """volume_sum = sum_v vis_v * ref_v (x) warp(src_v) with hand-written forward (K3) and backward kernels.
    Gradients: ref, src (channels-last), vis.  mats (CPU) and hyp carry none (warping.py:79)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # fp32 kernels, also under bf16 autocast
    def forward(ctx, ref_chw, src_hwc, vis_w, mats, hyp):
        ref_chw, src_hwc, vis_w, hyp = (t.contiguous() for t in (ref_chw, src_hwc, vis_w, hyp))
        ctx.save_for_backward(ref_chw, src_hwc, vis_w, hyp)
        ctx.mats = mats
        volume, _ = warp_aggregate(ref_chw, src_hwc, vis_w, mats, hyp, normalize=False, exact=True)   # the backward kernel differentiates the reference-order positions
        return volume

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_volume):
        ref_chw, src_hwc, vis_w, hyp = ctx.saved_tensors
        g_ref, g_src, g_vis = warp_aggregate_bwd(ref_chw, src_hwc, vis_w, ctx.mats, hyp, grad_volume.contiguous())
        return g_ref, g_src, g_vis, None, None


class VolumeFinish(torch.autograd.Function):
    """(volume, feat_distance) of the training step from the un-normalised K3 outputs (models/model.py:56-78): volume_sum [C,D,h,w],
    gt_sum [C,1,h,w] or None, vis [V,h,w] -> volume_sum / (sum_v vis + 1e-6) and [D (+1),h,w] = channel sums / the same denominator.
    One launch forward, two backward (8 + ~15 ATen launches otherwise)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, volume_sum, gt_sum, vis):
        volume_sum, vis = volume_sum.contiguous(), vis.contiguous()
        gt_sum = gt_sum.contiguous() if gt_sum is not None else None
        C, D, h, w = volume_sum.shape
        V = vis.shape[0]
        if tuple(vis.shape) != (V, h, w) or (gt_sum is not None and tuple(gt_sum.shape) != (C, 1, h, w)):
            raise ValueError("VolumeFinish: inconsistent shapes")
        vol = torch.empty_like(volume_sum)
        fd = torch.empty((D + (1 if gt_sum is not None else 0), h, w), dtype=torch.float32, device=vol.device)
        check(_lib.load().cds_volume_finish_f32(_dev(volume_sum, "volume_sum"), gt_sum.data_ptr() if gt_sum is not None else None,
                                                _dev(vis, "vis"), V, C, D, h * w, vol.data_ptr(), fd.data_ptr(), _stream(vol)),
              "cds_volume_finish_f32")
        ctx.save_for_backward(volume_sum, gt_sum, vis)
        return vol, fd

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_vol, g_fd):
        volume_sum, gt_sum, vis = ctx.saved_tensors
        C, D, h, w = volume_sum.shape
        V = vis.shape[0]
        g_vol = g_vol.contiguous().float() if g_vol is not None else None
        g_fd = g_fd.contiguous().float() if g_fd is not None else None
        g_vs = torch.empty_like(volume_sum)
        g_gt = torch.empty_like(gt_sum) if gt_sum is not None else None
        g_vis = torch.empty_like(vis)
        scratch = torch.empty((D, h, w), dtype=torch.float32, device=vis.device)
        p = lambda t: t.data_ptr() if t is not None else None                                       # noqa: E731
        check(_lib.load().cds_volume_finish_bwd_f32(p(g_vol), p(g_fd), p(volume_sum), p(gt_sum), p(vis), V, C, D, h * w, p(g_vs), p(g_gt),
                                                    p(g_vis), p(scratch), _stream(vis)), "cds_volume_finish_bwd_f32")
        return g_vs, g_gt, g_vis


def volume_normalize_(volume: Tensor, vis_sum: Tensor, channels_last: bool = False) -> Tensor:
    """volume /= (vis_sum + 1e-6) per pixel, in place (model.py:74); volume [C,D,h,w], or [D,h,w,C] with channels_last."""
    if channels_last:
        D, h, w, C = volume.shape
        check(_lib.load().cds_volume_normalize_cl_f32(_dev(volume, "volume"), _dev(vis_sum, "vis_sum"), C, D, h * w,
                                                      _stream(volume)), "cds_volume_normalize_cl_f32")
        return volume
    C, D, h, w = volume.shape
    check(_lib.load().cds_volume_normalize_f32(_dev(volume, "volume"), _dev(vis_sum, "vis_sum"), C, D, h * w,
                                               _stream(volume)), "cds_volume_normalize_f32")
    return volume


def softargmin_conf(prob_pre: Tensor, hyp: Tensor, want_prob: bool = False):
    """K5.  prob_pre [D,h,w] -> depth [h,w], confidence [h,w] (and prob [D,h,w] if requested)."""
    D, h, w = prob_pre.shape
    D2, pp = _hyp_args(hyp, D, h, w)
    if D2 != D:
        raise ValueError("softargmin_conf: hypotheses / volume depth mismatch")
    depth = torch.empty((h, w), dtype=torch.float32, device=prob_pre.device)
    conf = torch.empty_like(depth)
    prob = torch.empty_like(prob_pre) if want_prob else None
    with prof("softargmin_conf"):
        check(_lib.load().cds_softargmin_conf_f32(_dev(prob_pre, "prob_pre"), _dev(hyp, "hyp"), depth.data_ptr(),
                                                  conf.data_ptr(), prob.data_ptr() if want_prob else None, D, h, w, pp,
                                                  _stream(depth)), "cds_softargmin_conf_f32")
    return (depth, conf, prob) if want_prob else (depth, conf)


def depth_hypotheses(prev_depth: Tensor, D: int, H: int, W: int, scale: int, interval, dmin, dmax=None) -> Tensor:
    """K6.  prev_depth [hp,wp] -> hypotheses [D,H/scale,W/scale].  interval: the stage's hypothesis spacing, dmin / dmax: the clamp
    range - device scalars of the geometry block (interval = a 1-element device tensor, dmin = a 2-element device tensor (dmin, dmax),
    dmax None) or Python floats (uploaded)."""
    hp, wp = prev_depth.shape
    dev = prev_depth.device
    out = torch.empty((D, H // scale, W // scale), dtype=torch.float32, device=dev)
    if dmax is None:
        ival, rng = geo(interval, dev, "interval"), geo(dmin, dev, "depth_range")
    else:
        ival = geo([interval, dmin, dmax], dev, "interval")
        rng = ival[1:]
    if ival.numel() < 1 or rng.numel() < 2:
        raise ValueError("depth_hypotheses: interval needs 1 value, the depth range 2")
    check(_lib.load().cds_depth_hypotheses_f32(_dev(prev_depth, "prev_depth"), out.data_ptr(), D, hp, wp, H, W, scale,
                                               _dev(ival, "interval"), _dev(rng, "depth_range"), _stream(out)),
          "cds_depth_hypotheses_f32")
    return out


def depth_planes(D: int, h: int, w: int, lo, hi, device) -> Tensor:
    """First-stage planes lo + k (hi - lo) / (D - 1).  lo = a 2-element device tensor (lo, hi) of the geometry block with hi None,
    or two Python floats (uploaded)."""
    out = torch.empty((D, h, w), dtype=torch.float32, device=device)
    rng = geo(lo if hi is None else [lo, hi], out.device, "depth_range")
    if rng.numel() < 2:
        raise ValueError("depth_planes: the depth range needs 2 values")
    check(_lib.load().cds_depth_planes_f32(out.data_ptr(), D, h, w, _dev(rng, "depth_range"), _stream(out)),
          "cds_depth_planes_f32")
    return out


USE_CONV3D_CL = os.environ.get("CDS_CONV_CL", "1") != "0"     # A/B knob


def conv3d_cl_supported(Cin: int, Cout: int, W: int, stride: int) -> bool:
    """Shapes covered by cds_conv3d_k3_cl_f32 (channels-last LDS tile on the matrix cores)."""
    return stride == 1 and Cin % 16 == 0 and Cout % 16 == 0 and W % 4 == 0 and W >= 16 and USE_CONV3D_CL


def conv3d_cl_preferred(Cin: int) -> bool:
    """Measured at M1: the channels-last kernel stages a whole 16-channel chunk before its first MFMA, which is only
    amortised from 4 chunks on (conv6, 64 -> 64: 345 vs 462 us; conv4 32 -> 32: 596 vs 580; conv2 16 -> 16: 1386 vs 1127)."""
    return Cin >= 64 or os.environ.get("CDS_CONV_CL", "1") == "2"


def conv3d_k3(x: Tensor, wpk: Tensor, bias: Optional[Tensor], stride: int = 1, relu: bool = True,
              skip: Optional[Tensor] = None, wcl: Optional[Tensor] = None) -> Tensor:
    """K4.  x [Cin,D,H,W], wpk packed [Cin,27,Cout] -> [Cout,Do,Ho,Wo].  wcl: optional ci-fastest copy [27,Cout,Cin]
    of the same weights; when given and the shape is covered the channels-last MFMA kernel is used."""
    Cin, D, H, W = x.shape
    if (wcl is not None and conv3d_cl_supported(Cin, wpk.shape[2], W, stride) and conv3d_cl_preferred(Cin)
            and Cin * D * H * W < 0x7fffffff):
        Cout = wpk.shape[2]
        if tuple(wcl.shape) != (27, Cout, Cin):
            raise ValueError("conv3d_k3: wcl must be [27,Cout,Cin]")
        out = torch.empty((Cout, D, H, W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("conv3d_k3: residual shape mismatch")
        check(_lib.load().cds_conv3d_k3_cl_f32(_dev(x, "x"), _dev(wcl, "weight_cl"), _dev(bias, "bias") if bias is not None else None,
                                               _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, Cout,
                                               D, H, W, ACT_RELU if relu else ACT_NONE, _stream(x)), "cds_conv3d_k3_cl_f32")
        return out
    if wpk.shape[0] != Cin or wpk.shape[1] != 27:
        raise ValueError("conv3d_k3: packed weight must be [Cin,27,Cout]")
    Cout = wpk.shape[2]
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((Cout, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("conv3d_k3: residual shape mismatch")
    check(_lib.load().cds_conv3d_k3_f32(_dev(x, "x"), _dev(wpk, "weight"), _dev(bias, "bias") if bias is not None else None,
                                        _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, Cout, D, H,
                                        W, stride, ACT_RELU if relu else ACT_NONE, _stream(x)), "cds_conv3d_k3_f32")
    return out


def split_pack_conv3d(w: Tensor, f16: bool = False):
    """Pack a (BN-folded) Conv3d weight [Cout,Cin,3,3,3] for cds_conv3d_sbf_f32: every weight is split exactly into
    three bf16 terms (hi = RN(w), mid = RN(w - hi), lo = w - hi - mid) and laid out as the A operands of
    v_mfma_f32_16x16x32_bf16: int16 [Cin/8][7 ksteps][ceil(Cout/16)][3][64 lanes][8].  Lane l = 16 g + i multiplies output
    channel 16 mb + i by tap 4 t + g (tap = (kz*3 + ky)*3 + kx; tap 27 is a zero pad) of input channels 8 rd + 0..7."""
    Cout, Cin = w.shape[:2]
    if Cin % 8:
        raise ValueError("split_pack_conv3d: Cin must be a multiple of 8")
    rounds, mbl = Cin // 8, (Cout + 15) // 16
    taps = torch.zeros((mbl * 16, rounds, 28, 8), dtype=torch.float32, device=w.device)         # [co][rd][tap][j]
    taps[:Cout, :, :27] = w.detach().float().reshape(Cout, rounds, 8, 27).permute(0, 1, 3, 2)
    # -> [rd][t][mb][g][i][j] -> lanes l = 16 g + i
    a = taps.reshape(mbl, 16, rounds, 7, 4, 8).permute(2, 3, 0, 4, 1, 5).reshape(rounds, 7, mbl, 64, 8)
    return _split2_f16(a) if f16 else _split3(a)                                                 # [rd][t][mb][3][64][8] (f16: + 1 / scale)


def _split2_f16(a: Tensor) -> Tuple[Tensor, float]:
    """Split-f16 form of a weight operand [...,64,8] (csrc/sbf_common.hpp): two fp16 terms of a * s for the power of two s that puts
    max |a| into (2^14, 2^15] -> (int16 [...,3,64,8] = (hi, lo, unused), 1 / s).  The third slot keeps the split-bf16 geometry."""
    import math
    m = float(a.abs().max())
    e = math.frexp(m)[1] if (m > 0.0 and math.isfinite(m)) else 15
    e = max(-100, min(100, e))
    s = 2.0 ** (15 - e)
    v = a * s
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return torch.stack((hi, lo, torch.zeros_like(hi)), dim=-3).contiguous().view(torch.int16), 1.0 / s


def _split3(a: Tensor) -> Tensor:
    """Exact three-term bf16 split of a float tensor [...,64,8] -> int16 [...,3,64,8] (hi, mid, lo)."""
    hi = a.to(torch.bfloat16)
    r1 = a - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack((hi, mid, lo), dim=-3).contiguous().view(torch.int16)


def _deconv_tables(merge: bool):
    """Mirror of DTab in csrc/conv3d_sbf.hip: per K-step (class, first tap slot); per class its tap list of per-axis
    (kernel index k, cell offset d) pairs.  Output parity 0 of an axis takes (k=1, d=0); parity 1 takes (2, 0) and (0, 1)."""
    ax = {0: [(1, 0)], 1: [(2, 0), (0, 1)]}
    if merge:   # class = (pz, py); x slots are the cell offsets d = 0, 1 shared by both x parities
        classes = [(pz, py, None) for pz in (0, 1) for py in (0, 1)]
        ksteps = [(0, 0), (1, 0), (2, 0), (3, 0), (3, 4)]
    else:
        classes = [(pz, py, px) for pz in (0, 1) for py in (0, 1) for px in (0, 1)]
        ksteps = [(c, 0) for c in range(8)] + [(7, 4)]
    taps = []
    for pz, py, px in classes:
        tl = []
        for kz, dz in ax[pz]:
            for ky, dy in ax[py]:
                for xs in ([0, 1] if merge else ax[px]):
                    tl.append((kz, ky, xs))        # merge: xs = cell offset dx; else (kx, dx)
        taps.append(tl)
    return classes, ksteps, taps


def split_pack_deconv3d(w: Tensor, f16: bool = False):
    """Pack a (BN-folded) ConvTranspose3d weight [Cin,Cout,3,3,3] for cds_deconv3d_sbf_f32 (Cout == 8: the two x parities
    share an MFMA, rows = (px, cout); Cout in {16, 32}: rows = couts).  int16 [Cin/8][nks][mb][3][64][8]."""
    Cin, Cout = w.shape[:2]
    if Cin % 8 or Cout not in (8, 16, 32):
        raise ValueError("split_pack_deconv3d: Cin % 8 == 0 and Cout in {8, 16, 32}")
    merge = Cout == 8
    mbl = 1 if merge else Cout // 16
    rounds = Cin // 8
    _, ksteps, taps = _deconv_tables(merge)
    wf = w.detach().float().reshape(rounds, 8, Cout, 3, 3, 3)                       # [rd][j][co][kz][ky][kx]
    a = torch.zeros((rounds, len(ksteps), mbl, 4, 16, 8), dtype=torch.float32, device=w.device)   # [rd][ks][mb][g][i][j]
    for ks, (c, s0) in enumerate(ksteps):
        for gg in range(4):
            s = s0 + gg
            if s >= len(taps[c]):
                continue
            kz, ky, xs = taps[c][s]
            if merge:
                dx = xs
                if dx == 0:
                    a[:, ks, 0, gg, 0:8] = wf[:, :, :, kz, ky, 1].permute(0, 2, 1)      # px = 0: tap 1 of cell a
                    a[:, ks, 0, gg, 8:16] = wf[:, :, :, kz, ky, 2].permute(0, 2, 1)     # px = 1: tap 2 of cell a
                else:
                    a[:, ks, 0, gg, 8:16] = wf[:, :, :, kz, ky, 0].permute(0, 2, 1)     # px = 1: tap 0 of cell a + 1
            else:
                kx = xs[0]
                for mb in range(mbl):
                    a[:, ks, mb, gg] = wf[:, :, mb * 16:(mb + 1) * 16, kz, ky, kx].permute(0, 2, 1)
    a = a.reshape(rounds, len(ksteps), mbl, 64, 8)
    return _split2_f16(a) if f16 else _split3(a)      # f16: (tensor, 1 / weight scale) for cds_deconv3d_sf16_f32


def deconv3d_sbf(x_cl: Tensor, wsplit: Tensor, bias: Optional[Tensor], cout: int, relu: bool = True,
                 skip: Optional[Tensor] = None, out_planar: bool = False, in_bound: Optional[Tensor] = None, w_inv_scale: float = 1.0,
                 out_bound: Optional[Tensor] = None) -> Tensor:
    """ConvTranspose3d k3 s2 p1 op1 in split-bf16 arithmetic, channels-last: x_cl [D,H,W,Cin] -> [2D,2H,2W,cout]
    (or planar [cout,2D,2H,2W] with out_planar; the residual `skip` is channels-last either way)."""
    D, H, W, Cin = x_cl.shape
    out = torch.empty((cout, 2 * D, 2 * H, 2 * W) if out_planar else (2 * D, 2 * H, 2 * W, cout), dtype=torch.float32,
                      device=x_cl.device)
    if skip is not None and tuple(skip.shape) != (2 * D, 2 * H, 2 * W, cout):
        raise ValueError("deconv3d_sbf: residual shape mismatch")
    if wsplit.dtype != torch.int16 or not wsplit.is_cuda or not wsplit.is_contiguous():
        raise ValueError("deconv3d_sbf: wsplit must be the contiguous int16 device tensor from split_pack_deconv3d")
    if in_bound is not None:       # split-f16 arithmetic (cout == 32: conv7 of CostRegNet); operands from split_pack_deconv3d(..., f16=True)
        if cout != 32 or out_planar:
            raise ValueError("deconv3d_sbf: the split-f16 form exists for cout == 32, channels-last output")
        check(_lib.load().cds_deconv3d_sf16_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                                _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, cout, D, H, W,
                                                ACT_RELU if relu else ACT_NONE, _dev(in_bound, "in_bound"), float(w_inv_scale),
                                                _dev(out_bound, "out_bound") if out_bound is not None else None, _stream(x_cl)),
              "cds_deconv3d_sf16_f32")
        return out
    check(_lib.load().cds_deconv3d_sbf_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                           _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, cout,
                                           D, H, W, ACT_RELU if relu else ACT_NONE, 1 if out_planar else 0, _stream(x_cl)),
          "cds_deconv3d_sbf_f32")
    return out


def split_pack_deconv_cls(w: Tensor, f16: bool = False):
    """Pack a (BN-folded) ConvTranspose3d weight [32,16,3,3,3] for cds_deconv3d_zm_f32: per output parity class c = 4 pz + 2 py + px
    and 8-channel round two matrix operands (cell offset dz = 0 | 1), rows = couts, K slot g = (dy, dx) = (g >> 1, g & 1); a slot the
    class does not reach (d > parity on an axis) is zero.  Per axis: parity 0 takes kernel tap 1 of cell a; parity 1 takes tap 2 of
    cell a and tap 0 of cell a + 1.  int16 [8][4][2][3][64][8]."""
    if tuple(w.shape) != (32, 16, 3, 3, 3):
        raise ValueError("split_pack_deconv_cls: ConvTranspose3d weight [32,16,3,3,3]")
    wf = w.detach().float().reshape(4, 8, 16, 3, 3, 3)                              # [rd][j][co][kz][ky][kx]
    a = torch.zeros((8, 4, 2, 4, 16, 8), dtype=torch.float32, device=w.device)      # [class][rd][dz][g][co][j]
    tap = lambda par, d: (1 if d == 0 else None) if par == 0 else (2 if d == 0 else 0)
    for c in range(8):
        pz, py, px = c >> 2, (c >> 1) & 1, c & 1
        for dz in range(2):
            for g in range(4):
                kz, ky, kx = tap(pz, dz), tap(py, g >> 1), tap(px, g & 1)
                if kz is None or ky is None or kx is None:
                    continue
                a[c, :, dz, g] = wf[:, :, :, kz, ky, kx].permute(0, 2, 1)
    a = a.reshape(8, 4, 2, 64, 8)
    return _split2_f16(a) if f16 else _split3(a)      # f16: (tensor, 1 / weight scale) for cds_deconv3d_zm_sf16_f32


def deconv3d_zm(x_cl: Tensor, wcls: Tensor, bias: Optional[Tensor], relu: bool = True, skip: Optional[Tensor] = None,
                in_bound: Optional[Tensor] = None, w_inv_scale: float = 1.0, out_bound: Optional[Tensor] = None) -> Tensor:
    """ConvTranspose3d 32 -> 16 (k3 s2 p1 op1) in split-bf16 arithmetic on the z-marching class-per-wave kernel, channels-last:
    x_cl [D,H,W,32] -> [2D,2H,2W,16].  in_bound given: split-f16 arithmetic (wcls / w_inv_scale from split_pack_deconv_cls(..., f16=True);
    bounds as in conv3d_sbf)."""
    D, H, W, Cin = x_cl.shape
    out = torch.empty((2 * D, 2 * H, 2 * W, 16), dtype=torch.float32, device=x_cl.device)
    if skip is not None and tuple(skip.shape) != tuple(out.shape):
        raise ValueError("deconv3d_zm: residual shape mismatch")
    if wcls.dtype != torch.int16 or not wcls.is_cuda or not wcls.is_contiguous() or wcls.numel() != 8 * 4 * 2 * 3 * 64 * 8:
        raise ValueError("deconv3d_zm: wcls must be the contiguous int16 device tensor from split_pack_deconv_cls")
    if in_bound is not None:
        check(_lib.load().cds_deconv3d_zm_sf16_f32(_dev(x_cl, "x"), wcls.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                                   _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, 16, D, H, W,
                                                   ACT_RELU if relu else ACT_NONE, _dev(in_bound, "in_bound"), float(w_inv_scale),
                                                   _dev(out_bound, "out_bound") if out_bound is not None else None, _stream(x_cl)),
              "cds_deconv3d_zm_sf16_f32")
        return out
    check(_lib.load().cds_deconv3d_zm_f32(_dev(x_cl, "x"), wcls.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                          _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, 16, D, H, W,
                                          ACT_RELU if relu else ACT_NONE, _stream(x_cl)), "cds_deconv3d_zm_f32")
    return out


def split_pack_deconv_prob(w: Tensor, f16: bool = False):
    """Pack the (BN-folded) conv11 weight [16,8,3,3,3] for cds_deconv_prob_zm_f32: five matrix operands per 8-channel round,
    rows = (x parity, cout), K slot g = a cell offset (csrc/deconv_prob_zm.hip).  Output parity 0 of an axis takes kernel tap
    1 of cell a; parity 1 takes tap 2 of cell a and tap 0 of cell a + 1.  int16 [2][5][3][64][8]."""
    if tuple(w.shape) != (16, 8, 3, 3, 3):
        raise ValueError("split_pack_deconv_prob: ConvTranspose3d weight [16,8,3,3,3]")
    wf = w.detach().float().reshape(2, 8, 8, 3, 3, 3)                               # [rd][j][co][kz][ky][kx]
    a = torch.zeros((2, 5, 4, 16, 8), dtype=torch.float32, device=w.device)         # [rd][operand][g][row][j]
    # operand -> per slot group (g >> 1) the (kz, ky) it multiplies, or None; g & 1 = dx
    ops_tab = [
        [(1, 1), None],          # plane 2a,     y parity 0: slots (dy, dx); dy = 1 unused
        [(1, 2), (1, 0)],        # plane 2a,     y parity 1: dy = 0 -> ky 2, dy = 1 -> ky 0
        [(2, 1), (0, 1)],        # plane 2a + 1, y parity 0: slots (dz, dx) at dy = 0: dz = 0 -> kz 2, dz = 1 -> kz 0
        [(2, 2), (0, 2)],        # plane 2a + 1, y parity 1, dy = 0 (ky 2)
        [(2, 0), (0, 0)],        # plane 2a + 1, y parity 1, dy = 1 (ky 0)
    ]
    for k, groups in enumerate(ops_tab):
        for hi, kk in enumerate(groups):
            if kk is None:
                continue
            kz, ky = kk
            a[:, k, 2 * hi, 0:8] = wf[:, :, :, kz, ky, 1].permute(0, 2, 1)          # dx = 0, px = 0: tap 1 of cell a
            a[:, k, 2 * hi, 8:16] = wf[:, :, :, kz, ky, 2].permute(0, 2, 1)         # dx = 0, px = 1: tap 2 of cell a
            a[:, k, 2 * hi + 1, 8:16] = wf[:, :, :, kz, ky, 0].permute(0, 2, 1)     # dx = 1, px = 1: tap 0 of cell a + 1
    a = a.reshape(2, 5, 64, 8)
    return _split2_f16(a) if f16 else _split3(a)      # f16: (tensor, 1 / weight scale) for cds_deconv_prob_zm_sf16_f32


def pack_prob_table(w: Tensor) -> Tensor:
    """prob weight [1,8,3,3,3] -> float [3 kx][2 channel halves][3 ky][3 kz][4] for cds_deconv_prob_zm_f32."""
    if tuple(w.shape) != (1, 8, 3, 3, 3):
        raise ValueError("pack_prob_table: Conv3d weight [1,8,3,3,3]")
    return w.detach().float()[0].reshape(2, 4, 3, 3, 3).permute(4, 0, 3, 2, 1).contiguous()   # [h][i][kz][ky][kx] -> [kx][h][ky][kz][i]


def deconv_prob_zm(x_cl: Tensor, wsplit: Tensor, bias: Tensor, skip: Tensor, prob_table: Tensor, in_bound: Optional[Tensor] = None,
                   w_inv_scale: float = 1.0) -> Tensor:
    """conv11 + residual + prob in one launch: x_cl [D,H,W,16], skip [2D,2H,2W,8] channels-last -> [2D,2H,2W].  in_bound given: the
    transposed convolution in split-f16 arithmetic (wsplit / w_inv_scale from split_pack_deconv_prob(..., f16=True))."""
    D, H, W, Cin = x_cl.shape
    if Cin != 16 or tuple(skip.shape) != (2 * D, 2 * H, 2 * W, 8):
        raise ValueError("deconv_prob_zm: x [D,H,W,16] and skip [2D,2H,2W,8]")
    if wsplit.dtype != torch.int16 or not wsplit.is_cuda or not wsplit.is_contiguous() or wsplit.numel() != 2 * 5 * 3 * 64 * 8:
        raise ValueError("deconv_prob_zm: wsplit must be the contiguous int16 device tensor from split_pack_deconv_prob")
    if prob_table.numel() != 216 or bias.numel() != 8:
        raise ValueError("deconv_prob_zm: prob_table from pack_prob_table, bias [8]")
    out = torch.empty((2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x_cl.device)
    if in_bound is not None:
        check(_lib.load().cds_deconv_prob_zm_sf16_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias"), _dev(skip, "skip"),
                                                      _dev(prob_table, "prob_table"), out.data_ptr(), D, H, W, _dev(in_bound, "in_bound"),
                                                      float(w_inv_scale), _stream(x_cl)), "cds_deconv_prob_zm_sf16_f32")
        return out
    check(_lib.load().cds_deconv_prob_zm_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias"), _dev(skip, "skip"),
                                             _dev(prob_table, "prob_table"), out.data_ptr(), D, H, W, _stream(x_cl)),
          "cds_deconv_prob_zm_f32")
    return out


SBF_PAIR = 101   # CDS_SBF_PAIR: stride code of the pair-packed stride-1, Cout = 8 form
# split-f16 arithmetic (two fp16 terms, three products: half the matrix-pipe work of split-bf16 at fp32-class error) for the layers that
# have it (round 6: conv0 - conv3 of CostRegNet); CDS_SPLIT_F16=0: split-bf16 everywhere (A/B, and the arithmetic of rounds 2-5)
USE_SPLIT_F16 = os.environ.get("CDS_SPLIT_F16", "1") != "0"


def split_pack_conv3d_pair(w: Tensor, f16: bool = False):
    """Pair packing of a (BN-folded) Conv3d weight [8,Cin,3,3,3] for cds_conv3d_sbf_f32(stride=CDS_SBF_PAIR): an MFMA column
    is the voxel pair (2 j, 2 j + 1), row i = 8 p + co is output channel co of voxel 2 j + p, and the K window is 3 x 3 x 4
    taps (x' = 0..3 relative to the pair): row (p, co) multiplies w[co][ci][kz][ky][x' - p], or 0 outside 0..2.
    int16 [Cin/8][9 ksteps][1][3][64][8]; K-step t = kz * 3 + ky, lane l = 16 g + i holds x' = (0, 2, 1, 3)[g]."""
    Cout, Cin = w.shape[:2]
    if Cout != 8 or Cin % 8:
        raise ValueError("split_pack_conv3d_pair: Cout == 8 and Cin % 8 == 0")
    rounds = Cin // 8
    wf = w.detach().float().reshape(8, rounds, 8, 3, 3, 3)                          # [co][rd][j][kz][ky][kx]
    taps = torch.zeros((2, 8, rounds, 3, 3, 4, 8), dtype=torch.float32, device=w.device)   # [p][co][rd][kz][ky][x'][j]
    for p_ in (0, 1):
        taps[p_, :, :, :, :, p_:p_ + 3] = wf.permute(0, 1, 3, 4, 5, 2)
    taps = taps[:, :, :, :, :, [0, 2, 1, 3]]          # lane group g multiplies x' = (0, 2, 1, 3)[g] (LDS bank pairing)
    a = taps.reshape(16, rounds, 9, 4, 8).permute(1, 2, 3, 0, 4).reshape(rounds, 9, 1, 64, 8)   # [rd][t][mb][16 g + i][j]
    return _split2_f16(a) if f16 else _split3(a)


def conv3d_sf16_supported(cin: int, cout: int, stride: int) -> bool:
    """Shapes of the split-f16 z-marching kernels (cds_conv3d_sf16_f32)."""
    return USE_SPLIT_F16 and (cin, cout, stride) in ((8, 8, SBF_PAIR), (16, 8, SBF_PAIR), (32, 8, SBF_PAIR), (16, 16, 1), (8, 16, 2), (16, 32, 2),
                                                     (32, 32, 1), (32, 64, 2), (64, 64, 1))      # z-marching kernels | tiled kernels


def conv3d_sbf(x_cl: Tensor, wsplit: Tensor, bias: Optional[Tensor], cout: int, stride: int = 1, relu: bool = True,
               skip: Optional[Tensor] = None, in_bound: Optional[Tensor] = None, w_inv_scale: float = 1.0,
               out_bound: Optional[Tensor] = None) -> Tensor:
    """K4 in split-bf16 arithmetic (fp32-class error on the bf16 matrix cores) on channels-last volumes.
    x_cl [D,H,W,Cin] fp32, wsplit from split_pack_conv3d -> [Do,Ho,Wo,cout].
    in_bound given: SPLIT-F16 arithmetic instead (cds_conv3d_sf16_f32; conv3d_sf16_supported shapes): wsplit / w_inv_scale from the
    packers with f16=True, in_bound a 1-element device tensor >= max |x_cl|, out_bound a ZEROED 1-element device tensor that receives
    max |out| (the next layer's in_bound) or None."""
    D, H, W, Cin = x_cl.shape
    sg = 1 if stride == SBF_PAIR else stride        # geometric stride (SBF_PAIR: stride 1, pair-packed weights)
    Do, Ho, Wo = (D - 1) // sg + 1, (H - 1) // sg + 1, (W - 1) // sg + 1
    out = torch.empty((Do, Ho, Wo, cout), dtype=torch.float32, device=x_cl.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("conv3d_sbf: residual shape mismatch")
    if wsplit.dtype != torch.int16 or not wsplit.is_cuda or not wsplit.is_contiguous():
        raise ValueError("conv3d_sbf: wsplit must be the contiguous int16 device tensor from split_pack_conv3d")
    if stride == SBF_PAIR and cout != 8:
        raise ValueError("conv3d_sbf: the pair-packed form (stride=SBF_PAIR) exists for cout == 8 only")
    ksteps = 9 if stride == SBF_PAIR else 7                     # split_pack_conv3d_pair / split_pack_conv3d
    want = (Cin // 8) * ksteps * ((cout + 15) // 16) * 3 * 64 * 8
    if Cin % 8 or wsplit.numel() != want:
        raise ValueError(f"conv3d_sbf: wsplit has {wsplit.numel()} entries, the packer gives {want} for Cin={Cin}, cout={cout}, "
                         f"stride code {stride}")
    if in_bound is not None:
        if skip is not None or not conv3d_sf16_supported(Cin, cout, stride):
            raise ValueError(f"conv3d_sbf: no split-f16 kernel for Cin={Cin}, cout={cout}, stride code {stride}, skip={skip is not None}")
        check(_lib.load().cds_conv3d_sf16_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                              out.data_ptr(), Cin, cout, D, H, W, stride, ACT_RELU if relu else ACT_NONE,
                                              _dev(in_bound, "in_bound"), float(w_inv_scale),
                                              _dev(out_bound, "out_bound") if out_bound is not None else None, _stream(x_cl)),
              "cds_conv3d_sf16_f32")
        return out
    check(_lib.load().cds_conv3d_sbf_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                         _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, cout,
                                         D, H, W, stride, ACT_RELU if relu else ACT_NONE, _stream(x_cl)), "cds_conv3d_sbf_f32")
    return out


def deconv3d_k3s2(x: Tensor, wpk: Tensor, bias: Optional[Tensor], relu: bool = True,
                  skip: Optional[Tensor] = None) -> Tensor:
    """K4.  x [Cin,D,H,W], wpk packed [Cin,27,Cout] -> [Cout,2D,2H,2W]."""
    Cin, D, H, W = x.shape
    if wpk.shape[0] != Cin or wpk.shape[1] != 27:
        raise ValueError("deconv3d_k3s2: packed weight must be [Cin,27,Cout]")
    Cout = wpk.shape[2]
    out = torch.empty((Cout, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("deconv3d_k3s2: residual shape mismatch")
    check(_lib.load().cds_deconv3d_k3s2_f32(_dev(x, "x"), _dev(wpk, "weight"),
                                            _dev(bias, "bias") if bias is not None else None,
                                            _dev(skip, "skip") if skip is not None else None, out.data_ptr(), Cin, Cout,
                                            D, H, W, ACT_RELU if relu else ACT_NONE, _stream(x)),
          "cds_deconv3d_k3s2_f32")
    return out


def conv2d(x: Tensor, wpk: Tensor, bias: Optional[Tensor], cout: int, k: int, stride: int = 1, pad: int = 0,
           act: int = ACT_NONE, out: Optional[Tensor] = None, in_affine: Optional[Tensor] = None) -> Tensor:
    """x [N,Cin,H,W], wpk packed [Cin,k*k,CoutP] -> [N,cout,Ho,Wo] (written into `out` if given).
    in_affine [N,Cin,3] (alpha, beta, slope): the producing layer's InstanceNorm + LeakyReLU applied on load."""
    N, Cin, H, W = x.shape
    if in_affine is not None and tuple(in_affine.shape) != (N, Cin, 3):
        raise ValueError(f"conv2d: in_affine must be [{N},{Cin},3], got {tuple(in_affine.shape)}")
    coutp = (cout + 7) // 8 * 8
    if tuple(wpk.shape) != (Cin, k * k, coutp):
        raise ValueError(f"conv2d: packed weight must be [{Cin},{k * k},{coutp}], got {tuple(wpk.shape)}")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    elif out.numel() != N * cout * Ho * Wo:
        raise ValueError("conv2d: bad output buffer")
    check(_lib.load().cds_conv2d_affine_f32(_dev(x, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None,
                                            _dev(wpk, "weight"), _dev(bias, "bias") if bias is not None else None,
                                            _dev(out, "out"), N, Cin, cout, H, W, k, stride, pad, act, _stream(x)),
          "cds_conv2d_affine_f32")
    return out


USE_CONV2D_MFMA = os.environ.get("CDS_CONV2D_MFMA", "1") != "0"   # A/B knob
# DynamicConv branch convolutions on the bf16 matrix cores in split-bf16 arithmetic (CDS_CONV_EXACT=1: exact-fp32 VALU kernels)
USE_CONV2D_SBF = os.environ.get("CDS_CONV_EXACT", "0") != "1" and os.environ.get("CDS_CONV2D_SBF", "1") != "0"


def conv2d_c16_supported(x: Tensor) -> bool:
    """Shapes the matrix-core 16 -> 16 3x3 convolution covers (rows must be whole 16-byte groups)."""
    return USE_CONV2D_MFMA and x.dim() == 4 and x.shape[1] == 16 and x.shape[3] % 4 == 0 and x.shape[3] >= 4


def conv2d_k3_c16(x: Tensor, wcl: Tensor, bias: Optional[Tensor], act: int = ACT_RELU, head_w: Optional[Tensor] = None,
                  head_b: Optional[Tensor] = None) -> Tensor:
    """3x3, pad 1, 16 -> 16 channels on the matrix cores (visibility CNN, model.py:14).  x [N,16,H,W], wcl [9,16,16]
    = weight.permute(2,3,0,1) (tap, cout, cin), bias [16].  With head_w [16] / head_b [1] the 1x1 head + sigmoid is
    applied in the same kernel and the result is [N,H,W]."""
    N, C, H, W = x.shape
    if C != 16 or tuple(wcl.shape) != (9, 16, 16) or W % 4:
        raise ValueError(f"conv2d_k3_c16: need x [N,16,H,W%4==0] and wcl [9,16,16], got {tuple(x.shape)}, {tuple(wcl.shape)}")
    if (head_w is None) != (head_b is None) or (head_w is not None and (head_w.numel() != 16 or head_b.numel() != 1)):
        raise ValueError("conv2d_k3_c16: head_w [16] and head_b [1] go together")
    out = torch.empty((N, H, W) if head_w is not None else (N, 16, H, W), dtype=torch.float32, device=x.device)
    check(_lib.load().cds_conv2d_k3_c16_f32(_dev(x, "x"), _dev(wcl, "weight_cl"), _dev(bias, "bias") if bias is not None else None,
                                            _dev(head_w, "head_w") if head_w is not None else None,
                                            _dev(head_b, "head_b") if head_b is not None else None,
                                            _dev(out, "out"), N, H, W, act, _stream(x)), "cds_conv2d_k3_c16_f32")
    return out


def conv2d_k3_relu_sbf(x: Tensor, wsplit: Tensor, bias: Optional[Tensor], head_w: Optional[Tensor] = None,
                       head_b: Optional[Tensor] = None) -> Tensor:
    """3x3, pad 1, Cin -> 16 channels + bias + ReLU in split-bf16 arithmetic on the matrix cores (visibility CNN, model.py:14);
    with head_w [16] / head_b [1] the 1x1 head + sigmoid is applied in the same kernel and the result is [N,H,W].
    wsplit = split_pack_dynconv([w]) with w [16,Cin,3,3]."""
    N, C, H, W = x.shape
    if C % 8 or W % 4:
        raise ValueError(f"conv2d_k3_relu_sbf: need Cin % 8 == 0 and W % 4 == 0, got {tuple(x.shape)}")
    if (head_w is None) != (head_b is None) or (head_w is not None and (head_w.numel() != 16 or head_b.numel() != 1)):
        raise ValueError("conv2d_k3_relu_sbf: head_w [16] and head_b [1] go together")
    out = torch.empty((N, H, W) if head_w is not None else (N, 16, H, W), dtype=torch.float32, device=x.device)
    check(_lib.load().cds_conv2d_k3_relu_sbf_f32(_dev(x, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                                 _dev(head_w, "head_w") if head_w is not None else None,
                                                 _dev(head_b, "head_b") if head_b is not None else None,
                                                 _dev(out, "out"), N, C, H, W, _stream(x)), "cds_conv2d_k3_relu_sbf_f32")
    return out


def dynconv_sbf_supported(Cin: int, co3: int, ksizes, W: int, fused: bool = False) -> bool:
    """Shapes cds_dynconv_branches_sbf_f32 covers (everything in FeatureNet but conv00, whose 3 input channels and 11 x 11
    kernel would leave the matrix tiles mostly padding)."""
    nb, nblk = len(ksizes), (co3 + 15) // 16
    # measured at the 1600x1184 cascade shapes (scripts/time_dynconv_sbf.py, 8 images): conv01 3.02 -> 1.96 ms, conv10 1.21 ->
    # 0.75, conv20 0.42 -> 0.37; the (1, 3) branches of the 8- / 16-channel output layers have too few taps to amortise the
    # staged tile (out2 0.44 -> 0.50, out3 0.77 -> 0.88): those stay on the VALU kernels
    # (with the blend fused behind them - dynconv_fused_sbf - they pay as well: the VALU branches + the separate blend cost more)
    worth = fused or max(ksizes) >= 5 or Cin >= 32
    return (USE_CONV2D_SBF and worth and Cin % 8 == 0 and W % 4 == 0 and all(k in (1, 3, 5, 7) for k in ksizes)
            and (nb, nblk) in ((3, 1), (3, 2), (2, 1), (2, 2), (2, 3)))


def split_pack_dynconv(ws, f16: bool = False):
    """Pack the branch weights of one DynamicConv for cds_dynconv_branches_sbf_f32.  ws: list over kernel sizes of
    [Co3,Cin,k,k] (convs[k] and att_convs[k] concatenated).  int16 [Cin/8][nks][nblk][3][64][8]; within a round branch b
    owns K-steps ks0_b .. ks0_b + ceil(k_b^2 / 4); lane l = 16 g + n multiplies output channel 16 nb + n by tap 4 t + g
    (tap = ky * k + kx; taps beyond k^2 and channels beyond Co3 are zero) of input channels 8 rd + 0..7."""
    Co3, Cin = ws[0].shape[:2]
    rounds, nblk = Cin // 8, (Co3 + 15) // 16
    parts = []
    for w in ws:
        k = w.shape[-1]
        nks = (k * k + 3) // 4
        taps = torch.zeros((nblk * 16, rounds, nks * 4, 8), dtype=torch.float32, device=w.device)     # [co][rd][tap][j]
        taps[:Co3, :, :k * k] = w.detach().float().reshape(Co3, rounds, 8, k * k).permute(0, 1, 3, 2)
        # -> [rd][t][nb][g][n][j]
        parts.append(taps.reshape(nblk, 16, rounds, nks, 4, 8).permute(2, 3, 0, 4, 1, 5).reshape(rounds, nks, nblk, 64, 8))
    a = torch.cat(parts, dim=1)
    return _split2_f16(a) if f16 else _split3(a)      # f16: (tensor, 1 / weight scale) for cds_dynconv_cl_sf16_f32


def split_pack_conv00(ws, f16: bool = False):
    """Pack the branch weights of conv00 (DynamicConv 3 -> 8: ws = [Co3 = 11, 3, k, k] for k = 3, 7, 11: convs[k] and att_convs[k]
    concatenated) for cds_conv00_cl_f32: a K-step is 4 x-adjacent tap PAIRS x (2 taps x 4 channels, the 4th zero).  int16
    [sum_k ceil(k ceil(k/2) / 4) = 26][3][64][8]; lane l = 16 g + n multiplies output channel n by the pair 4 t + g = (ky, kxp), values
    (kx = 2 kxp: c0 c1 c2 0, kx = 2 kxp + 1: c0 c1 c2 0); pairs beyond k ceil(k/2), kx = k and channels beyond Co3 are zero."""
    parts = []
    for w in ws:
        co3, cin, k, _ = w.shape
        if cin != 3 or co3 > 16:
            raise ValueError("split_pack_conv00: weights must be [<=16, 3, k, k]")
        np_ = (k + 1) // 2
        npairs, nks = k * np_, (k * np_ + 3) // 4
        wp = torch.zeros((co3, 3, k, 2 * np_), dtype=torch.float32, device=w.device)
        wp[..., :k] = w.detach().float()
        t = torch.zeros((16, nks * 4, 2, 4), dtype=torch.float32, device=w.device)                  # [col][pair][tap of the pair][ch]
        t[:co3, :npairs, :, :3] = wp.reshape(co3, 3, k, np_, 2).permute(0, 2, 3, 4, 1).reshape(co3, npairs, 2, 3)
        parts.append(t.reshape(16, nks, 4, 8).permute(1, 2, 0, 3).reshape(nks, 64, 8))              # [t][16 g + n][8]
    a = torch.cat(parts, dim=0)
    return _split2_f16(a) if f16 else _split3(a)      # f16: (tensor, 1 / weight scale) for cds_conv00_cl_sf16_f32


def conv00_cl(imgs: Tensor, wsplit: Tensor, bias: Optional[Tensor], w1: Tensor, b1: Tensor, w2: Tensor, epipoles: Tensor,
              temperature: float, n_shared: int = 1, stats_slope: float = 0.1, in_bound: Optional[Tensor] = None,
              w_inv_scale: float = 1.0):
    """conv00 of FeatureNet in one kernel on the matrix cores.  imgs [S,3,H,W] planar (S = N - n_shared + 1 slots: slot 0 is shown by
    the first n_shared of the N output images), epipoles [N,2] (device) -> (out_cl [N,H,W,8], norm_curv [N,H,W], stats [N,8,2] float64,
    affine [N,8,3]).  in_bound (1-element device tensor >= max |imgs|) given: split-f16 arithmetic, wsplit / w_inv_scale from
    split_pack_conv00(..., f16=True)."""
    S, C, H, W = imgs.shape
    N = S + n_shared - 1
    if C != 3 or tuple(epipoles.shape) != (N, 2) or n_shared < 1:
        raise ValueError(f"conv00_cl: need imgs [S,3,H,W] and epipoles [S + n_shared - 1, 2], got {tuple(imgs.shape)}, {tuple(epipoles.shape)}")
    if wsplit.dtype != torch.int16 or wsplit.numel() != 26 * 3 * 64 * 8:
        raise ValueError("conv00_cl: wsplit must come from split_pack_conv00 (kernel sizes 3, 7, 11)")
    if bias is not None and tuple(bias.shape) != (3, 11):
        raise ValueError("conv00_cl: bias must be [3, 11]")
    dev = imgs.device
    epipoles = geo(epipoles, dev, "epipoles")          # device slice of the geometry block (a CPU tensor is uploaded)
    lib = _lib.load()
    out = torch.empty((N, H, W, 8), dtype=torch.float32, device=dev)
    nc = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    partial = torch.empty((N, lib.cds_dynconv_cl_parts(H, W), 8, 2), dtype=torch.float64, device=dev)
    if in_bound is not None:
        check(lib.cds_conv00_cl_sf16_f32(_dev(imgs, "imgs"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                         _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"), _dev(epipoles, "epipoles"), float(temperature),
                                         out.data_ptr(), nc.data_ptr(), partial.data_ptr(), N, n_shared, H, W, _dev(in_bound, "in_bound"),
                                         float(w_inv_scale), _stream(imgs)), "cds_conv00_cl_sf16_f32")
    else:
        check(lib.cds_conv00_cl_f32(_dev(imgs, "imgs"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None, _dev(w1, "w1"),
                                    _dev(b1, "b1"), _dev(w2, "w2"), _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(),
                                    nc.data_ptr(), partial.data_ptr(), N, n_shared, H, W, _stream(imgs)), "cds_conv00_cl_f32")
    stats, affine = _reduce_records(partial, N, 8, H, W, stats_slope)
    return out, nc, stats, affine


def dynconv_branches_sbf(x: Tensor, wsplit: Tensor, bias: Optional[Tensor], co3: int, ksizes, out: Optional[Tensor] = None,
                         in_affine: Optional[Tensor] = None) -> Tensor:
    """All branch convolutions of a DynamicConv on the matrix cores: x [N,Cin,H,W] -> branches [K,N,co3,H,W]."""
    N, Cin, H, W = x.shape
    K = len(ksizes)
    if in_affine is not None and tuple(in_affine.shape) != (N, Cin, 3):
        raise ValueError(f"dynconv_branches_sbf: in_affine must be [{N},{Cin},3]")
    if out is None:
        out = torch.empty((K, N, co3, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (K, N, co3, H, W):
        raise ValueError("dynconv_branches_sbf: bad output buffer")
    if bias is not None and tuple(bias.shape) != (K, co3):
        raise ValueError("dynconv_branches_sbf: bias must be [K, co3]")
    import ctypes
    ks = (ctypes.c_int * K)(*[int(k) for k in ksizes])
    check(_lib.load().cds_dynconv_branches_sbf_f32(_dev(x, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None,
                                                   wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                                   _dev(out, "out"), N, Cin, co3, H, W, ks, K, _stream(x)),
          "cds_dynconv_branches_sbf_f32")
    return out


def dynconv_fused_sbf(x: Tensor, wsplit: Tensor, bias: Optional[Tensor], cout: int, ksizes, w1: Tensor, b1: Tensor, w2: Tensor,
                      epipoles: Tensor, temperature: float, stats_slope: float, in_affine: Optional[Tensor] = None):
    """One DynamicConv in one kernel (branch convolutions on the matrix cores + the blend epilogue on their accumulators):
    x [N,Cin,H,W] (+ its pending affine) -> (out [N,cout,H,W], norm_curv [N,H,W], stats [N,cout,2] float64, affine [N,cout,3]),
    the same tuple as ``dynconv_blend(dynconv_branches_sbf(...), ..., stats_slope=...)`` without the branch tensor."""
    N, Cin, H, W = x.shape
    K = len(ksizes)
    if in_affine is not None and tuple(in_affine.shape) != (N, Cin, 3):
        raise ValueError(f"dynconv_fused_sbf: in_affine must be [{N},{Cin},3]")
    if bias is not None and tuple(bias.shape) != (K, cout + 3):
        raise ValueError("dynconv_fused_sbf: bias must be [K, cout + 3]")
    if tuple(epipoles.shape) != (N, 2):
        raise ValueError("dynconv_fused_sbf: epipoles must be [N,2]")
    dev = x.device
    epipoles = geo(epipoles, dev, "epipoles")          # device slice of the geometry block (a CPU tensor is uploaded)
    lib = _lib.load()
    out = torch.empty((N, cout, H, W), dtype=torch.float32, device=dev)
    nc = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    parts = lib.cds_dynconv_fused_parts(H, W)
    partial = torch.empty((N, parts, cout, 2), dtype=torch.float64, device=dev)
    stats = torch.empty((N, cout, 2), dtype=torch.float64, device=dev)
    affine = torch.empty((N, cout, 3), dtype=torch.float32, device=dev)
    import ctypes
    ks = (ctypes.c_int * K)(*[int(k) for k in ksizes])
    check(lib.cds_dynconv_fused_sbf_f32(_dev(x, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None,
                                        wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None, _dev(w1, "w1"),
                                        _dev(b1, "b1"), _dev(w2, "w2"), _dev(epipoles, "epipoles"), float(temperature),
                                        out.data_ptr(), nc.data_ptr(), partial.data_ptr(), N, Cin, cout, H, W, ks, K, _stream(x)),
          "cds_dynconv_fused_sbf_f32")
    check(lib.cds_instnorm_reduce_f32(partial.data_ptr(), parts, stats.data_ptr(), affine.data_ptr(), N, cout, H, W,
                                      float(stats_slope), _stream(out)), "cds_instnorm_reduce_f32")
    return out, nc, stats, affine


def conv2d_fpn(coarse: Tensor, skip: Tensor, wpk: Tensor, cout: int, coarse_affine: Optional[Tensor] = None,
               skip_affine: Optional[Tensor] = None, stats_slope: Optional[float] = None):
    """FPN lateral (module.py:253-254,260-261): 1x1 conv of cat(nearest2x(coarse), skip) without building either.
    coarse [N,Ca,H/2,W/2], skip [N,Cb,H,W], wpk packed [Ca+Cb,1,CoutP] -> [N,cout,H,W]; the affine tables are the
    sources' pending InstanceNorm + LeakyReLU (see conv2d).  stats_slope given: returns (out, affine [N,cout,3]) with the
    output's own InstanceNorm table (= instnorm_affine(out, stats_slope)) from statistics taken inside the kernel."""
    N, Ca, hc, wc = coarse.shape
    Nb, Cb, H, W = skip.shape
    if Nb != N or H != 2 * hc or W != 2 * wc:
        raise ValueError(f"conv2d_fpn: skip {tuple(skip.shape)} is not twice coarse {tuple(coarse.shape)}")
    coutp = (cout + 7) // 8 * 8
    if tuple(wpk.shape) != (Ca + Cb, 1, coutp):
        raise ValueError(f"conv2d_fpn: packed weight must be [{Ca + Cb},1,{coutp}], got {tuple(wpk.shape)}")
    for a, c, nm in ((coarse_affine, Ca, "coarse_affine"), (skip_affine, Cb, "skip_affine")):
        if a is not None and tuple(a.shape) != (N, c, 3):
            raise ValueError(f"conv2d_fpn: {nm} must be [{N},{c},3], got {tuple(a.shape)}")
    dev = skip.device
    lib = _lib.load()
    out = torch.empty((N, cout, H, W), dtype=torch.float32, device=dev)
    parts = lib.cds_fpn_stats_parts(H, W) if stats_slope is not None else 0
    partial = torch.empty((N, parts, cout, 2), dtype=torch.float64, device=dev) if parts else None
    check(lib.cds_conv2d_fpn_f32(_dev(coarse, "coarse"),
                                 _dev(coarse_affine, "coarse_affine") if coarse_affine is not None else None,
                                 _dev(skip, "skip"), _dev(skip_affine, "skip_affine") if skip_affine is not None else None,
                                 _dev(wpk, "weight"), _dev(out, "out"), partial.data_ptr() if parts else None,
                                 N, Ca, Cb, cout, H, W, _stream(skip)), "cds_conv2d_fpn_f32")
    if stats_slope is None:
        return out
    stats = torch.empty((N, cout, 2), dtype=torch.float64, device=dev)
    affine = torch.empty((N, cout, 3), dtype=torch.float32, device=dev)
    check(lib.cds_instnorm_reduce_f32(partial.data_ptr(), parts, stats.data_ptr(), affine.data_ptr(), N, cout, H, W,
                                      float(stats_slope), _stream(skip)), "cds_instnorm_reduce_f32")
    return out, affine


def instnorm_affine(x: Tensor, slope: float = 0.1) -> Tensor:
    """InstanceNorm statistics of x [N,C,H,W] as (alpha, beta, slope) rows [N,C,3] for a consumer that normalises on
    load (conv2d(in_affine=...)); same statistics and expression as instnorm_act."""
    N, C, H, W = x.shape
    aff = torch.empty((N, C, 3), dtype=torch.float32, device=x.device)
    stats = torch.empty((2 * N * C,), dtype=torch.float64, device=x.device)
    check(_lib.load().cds_instnorm_affine_f32(_dev(x, "x"), aff.data_ptr(), stats.data_ptr(), N, C, H, W, float(slope),
                                              _stream(x)), "cds_instnorm_affine_f32")
    return aff


def dynconv_blend(branches: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, epipoles: Tensor,
                  temperature: float, n_shared: int = 1, stats_slope: Optional[float] = None):
    """K7 epilogue.  branches [K,N - n_shared + 1,Cout+3,H,W] (the first n_shared images share slot 0), epipoles
    [N,2] (device) -> (out [N,Cout,H,W], norm_curv [N,H,W]).
    stats_slope given: the kernel also leaves the InstanceNorm statistics of `out` (no second pass over it) and the call
    returns (out, norm_curv, stats [N,Cout,2] float64 = (sum, sum of squares), affine [N,Cout,3]) - affine as from
    instnorm_affine(out, stats_slope), stats for instnorm_apply."""
    K, nslots, C3, H, W = branches.shape
    N = nslots + n_shared - 1
    cout = C3 - 3
    if tuple(epipoles.shape) != (N, 2) or n_shared < 1:
        raise ValueError("dynconv_blend: epipoles must be [N,2]")
    dev = branches.device
    epipoles = geo(epipoles, dev, "epipoles")          # device slice of the geometry block (a CPU tensor is uploaded)
    out = torch.empty((N, cout, H, W), dtype=torch.float32, device=dev)
    nc = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if stats_slope is None:
        check(lib.cds_dynconv_blend_shared_f32(_dev(branches, "branches"), _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"),
                                               _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(),
                                               nc.data_ptr(), N, K, cout, H, W, n_shared, _stream(out)),
              "cds_dynconv_blend_shared_f32")
        return out, nc
    parts = lib.cds_blend_stats_parts(H, W)
    partial = torch.empty((N, parts, cout, 2), dtype=torch.float64, device=dev)
    stats = torch.empty((N, cout, 2), dtype=torch.float64, device=dev)
    affine = torch.empty((N, cout, 3), dtype=torch.float32, device=dev)
    check(lib.cds_dynconv_blend_stats_f32(_dev(branches, "branches"), _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"),
                                          _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(), nc.data_ptr(),
                                          partial.data_ptr(), N, K, cout, H, W, n_shared, _stream(out)),
          "cds_dynconv_blend_stats_f32")
    check(lib.cds_instnorm_reduce_f32(partial.data_ptr(), parts, stats.data_ptr(), affine.data_ptr(), N, cout, H, W,
                                      float(stats_slope), _stream(out)), "cds_instnorm_reduce_f32")
    return out, nc, stats, affine


def instnorm_apply(x: Tensor, stats: Tensor, act: int, out_hwc: bool = False) -> Tensor:
    """Second half of K8: x [N,C,H,W] with its statistics [N,C,2] float64 (sum, sum of squares; from
    dynconv_blend(stats_slope=...)) -> InstanceNorm + activation; [N,H,W,C] if out_hwc."""
    N, C, H, W = x.shape
    if stats.dtype != torch.float64 or tuple(stats.shape) != (N, C, 2):
        raise ValueError(f"instnorm_apply: stats must be float64 [{N},{C},2], got {stats.dtype} {tuple(stats.shape)}")
    out = torch.empty((N, H, W, C) if out_hwc else (N, C, H, W), dtype=torch.float32, device=x.device)
    check(_lib.load().cds_instnorm_apply_f32(_dev(x, "x"), _dev64(stats, "stats"), out.data_ptr(), N, C, H, W, act,
                                             1 if out_hwc else 0, _stream(x)), "cds_instnorm_apply_f32")
    return out


def instnorm_act(x: Tensor, act: int, out_hwc: bool = False) -> Tensor:
    """K8.  x [N,C,H,W] -> InstanceNorm + activation; [N,H,W,C] if out_hwc."""
    N, C, H, W = x.shape
    out = torch.empty((N, H, W, C) if out_hwc else (N, C, H, W), dtype=torch.float32, device=x.device)
    stats = torch.empty((2 * N * C,), dtype=torch.float64, device=x.device)
    check(_lib.load().cds_instnorm_act_f32(_dev(x, "x"), out.data_ptr(), stats.data_ptr(), N, C, H, W, act,
                                           1 if out_hwc else 0, _stream(x)), "cds_instnorm_act_f32")
    return out


# ------------------------------------------------------------------------------------------------
# FeatureNet on channels-last activations (csrc/feat_cl.hip): x_cl [N,H,W,C]
# ------------------------------------------------------------------------------------------------
def _reduce_records(partial: Tensor, N: int, C: int, H: int, W: int, slope: float):
    """partial [N,parts,C,2] float64 -> (stats [N,C,2] float64, affine [N,C,3]) in a fixed summation order."""
    stats = torch.empty((N, C, 2), dtype=torch.float64, device=partial.device)
    affine = torch.empty((N, C, 3), dtype=torch.float32, device=partial.device)
    check(_lib.load().cds_instnorm_reduce_f32(partial.data_ptr(), partial.shape[1], stats.data_ptr(), affine.data_ptr(), N, C, H, W,
                                              float(slope), _stream(partial)), "cds_instnorm_reduce_f32")
    return stats, affine


DYNCONV_CL_SHAPES = ((8, (3, 5, 7)), (8, (1, 3)), (16, (3, 5)), (16, (1, 3)), (32, (1, 3)))


def dynconv_cl(x_cl: Tensor, wsplit: Tensor, bias: Optional[Tensor], ksizes, w1: Tensor, b1: Tensor, w2: Tensor, epipoles: Tensor,
               temperature: float, stats_slope: float = 0.1, in_affine: Optional[Tensor] = None, x_bound: Optional[float] = None,
               w_inv_scale: float = 1.0):
    """One DynamicConv (Cin == Cout == C) on channels-last activations in one kernel: x_cl [N,H,W,C] (+ its pending affine [N,C,3])
    -> (out_cl [N,H,W,C] before its InstanceNorm, norm_curv [N,H,W], stats [N,C,2] float64, affine [N,C,3]).
    x_bound given: SPLIT-F16 arithmetic (cds_dynconv_cl_sf16_f32): wsplit / w_inv_scale from split_pack_dynconv(..., f16=True), x_bound a
    number >= max |input after its affine| (sqrt(H W) for an InstanceNorm-ed input)."""
    N, H, W, C = x_cl.shape
    K = len(ksizes)
    if (C, tuple(int(k) for k in ksizes)) not in DYNCONV_CL_SHAPES:
        raise ValueError(f"dynconv_cl: (C, kernel sizes) = ({C}, {tuple(ksizes)}) not in {DYNCONV_CL_SHAPES}")
    if in_affine is not None and tuple(in_affine.shape) != (N, C, 3):
        raise ValueError(f"dynconv_cl: in_affine must be [{N},{C},3]")
    if bias is not None and tuple(bias.shape) != (K, C + 3):
        raise ValueError("dynconv_cl: bias must be [K, C + 3]")
    if tuple(epipoles.shape) != (N, 2):
        raise ValueError("dynconv_cl: epipoles must be [N,2]")
    nks = sum((int(k) * int(k) + 3) // 4 for k in ksizes)
    if wsplit.dtype != torch.int16 or not wsplit.is_cuda or wsplit.numel() != (C // 8) * nks * ((C + 3 + 15) // 16) * 3 * 64 * 8:
        raise ValueError("dynconv_cl: wsplit must be split_pack_dynconv's int16 device tensor for these kernel sizes")
    dev = x_cl.device
    epipoles = geo(epipoles, dev, "epipoles")          # device slice of the geometry block (a CPU tensor is uploaded)
    lib = _lib.load()
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=dev)
    nc = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    partial = torch.empty((N, lib.cds_dynconv_cl_parts(H, W), C, 2), dtype=torch.float64, device=dev)
    import ctypes
    ks = (ctypes.c_int * K)(*[int(k) for k in ksizes])
    if x_bound is not None:
        check(lib.cds_dynconv_cl_sf16_f32(_dev(x_cl, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None, wsplit.data_ptr(),
                                          _dev(bias, "bias") if bias is not None else None, _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"),
                                          _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(), nc.data_ptr(), partial.data_ptr(),
                                          N, C, H, W, ks, K, float(x_bound), float(w_inv_scale), _stream(x_cl)), "cds_dynconv_cl_sf16_f32")
    else:
        check(lib.cds_dynconv_cl_f32(_dev(x_cl, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None, wsplit.data_ptr(),
                                     _dev(bias, "bias") if bias is not None else None, _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"),
                                     _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(), nc.data_ptr(), partial.data_ptr(),
                                     N, C, H, W, ks, K, _stream(x_cl)), "cds_dynconv_cl_f32")
    stats, affine = _reduce_records(partial, N, C, H, W, stats_slope)
    return out, nc, stats, affine


def dynconv_blend_cl(branches: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, epipoles: Tensor, temperature: float, n_shared: int = 1,
                     stats_slope: float = 0.1):
    """DynamicConv epilogue over a PLANAR branch tensor [3,N - n_shared + 1,8 + 3,H,W] (conv00) with a channels-last result:
    -> (out_cl [N,H,W,8], norm_curv [N,H,W], stats [N,8,2] float64, affine [N,8,3])."""
    K, nslots, C3, H, W = branches.shape
    N = nslots + n_shared - 1
    cout = C3 - 3
    if tuple(epipoles.shape) != (N, 2) or n_shared < 1:
        raise ValueError("dynconv_blend_cl: epipoles must be [N,2]")
    dev = branches.device
    epipoles = geo(epipoles, dev, "epipoles")          # device slice of the geometry block (a CPU tensor is uploaded)
    lib = _lib.load()
    out = torch.empty((N, H, W, cout), dtype=torch.float32, device=dev)
    nc = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    partial = torch.empty((N, lib.cds_blend_cl_parts(H, W), cout, 2), dtype=torch.float64, device=dev)
    check(lib.cds_dynconv_blend_cl_f32(_dev(branches, "branches"), _dev(w1, "w1"), _dev(b1, "b1"), _dev(w2, "w2"),
                                       _dev(epipoles, "epipoles"), float(temperature), out.data_ptr(), nc.data_ptr(),
                                       partial.data_ptr(), N, K, cout, H, W, n_shared, _stream(out)), "cds_dynconv_blend_cl_f32")
    stats, affine = _reduce_records(partial, N, cout, H, W, stats_slope)
    return out, nc, stats, affine


def instnorm_stats_cl(x_cl: Tensor, slope: float = 0.1):
    """InstanceNorm statistics of x_cl [N,H,W,C] -> (stats [N,C,2] float64, affine [N,C,3] = (1/std, -mean/std, slope))."""
    N, H, W, C = x_cl.shape
    lib = _lib.load()
    partial = torch.empty((N, lib.cds_instnorm_stats_cl_parts(H, W), C, 2), dtype=torch.float64, device=x_cl.device)
    check(lib.cds_instnorm_stats_cl_f32(_dev(x_cl, "x"), partial.data_ptr(), N, C, H, W, _stream(x_cl)), "cds_instnorm_stats_cl_f32")
    return _reduce_records(partial, N, C, H, W, slope)


def conv2d_k3s2_cl(x_cl: Tensor, w9: Optional[Tensor], cout: int, in_affine: Optional[Tensor] = None, wsplit: Optional[Tensor] = None,
                   w_inv_scale: float = 1.0, x_bound: Optional[float] = None) -> Tensor:
    """3x3, stride 2, pad 1, no bias on channels-last activations: x_cl [N,H,W,Cin] (+ pending affine), w9 [9,Cin,cout] ->
    [N,Ho,Wo,cout].  wsplit / w_inv_scale / x_bound given: the matrix-core kernel in split-f16 arithmetic (cds_conv2d_k3s2_cl_sf16_f32;
    wsplit from split_pack_dynconv([w [cout,Cin,3,3]], f16=True), x_bound a number >= max |input after its affine|)."""
    N, H, W, Cin = x_cl.shape
    if wsplit is not None:
        if x_bound is None or (Cin, cout) not in ((8, 16), (16, 32)) or wsplit.dtype != torch.int16 or \
                wsplit.numel() != (Cin // 8) * 3 * (cout // 16) * 3 * 64 * 8:
            raise ValueError("conv2d_k3s2_cl: the split-f16 form needs x_bound, (Cin, cout) in ((8, 16), (16, 32)) and split_pack_dynconv's tensor")
        if in_affine is not None and tuple(in_affine.shape) != (N, Cin, 3):
            raise ValueError(f"conv2d_k3s2_cl: in_affine must be [{N},{Cin},3]")
        out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, cout), dtype=torch.float32, device=x_cl.device)
        check(_lib.load().cds_conv2d_k3s2_cl_sf16_f32(_dev(x_cl, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None,
                                                      wsplit.data_ptr(), out.data_ptr(), N, Cin, cout, H, W, float(x_bound),
                                                      float(w_inv_scale), _stream(x_cl)), "cds_conv2d_k3s2_cl_sf16_f32")
        return out
    if tuple(w9.shape) != (9, Cin, cout):
        raise ValueError(f"conv2d_k3s2_cl: weight must be [9,{Cin},{cout}], got {tuple(w9.shape)}")
    if in_affine is not None and tuple(in_affine.shape) != (N, Cin, 3):
        raise ValueError(f"conv2d_k3s2_cl: in_affine must be [{N},{Cin},3]")
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, cout), dtype=torch.float32, device=x_cl.device)
    check(_lib.load().cds_conv2d_k3s2_cl_f32(_dev(x_cl, "x"), _dev(in_affine, "in_affine") if in_affine is not None else None,
                                             _dev(w9, "weight"), out.data_ptr(), N, Cin, cout, H, W, _stream(x_cl)),
          "cds_conv2d_k3s2_cl_f32")
    return out


def conv2d_fpn_cl(coarse_cl: Tensor, skip_cl: Tensor, w: Tensor, cout: int, coarse_affine: Optional[Tensor] = None,
                  skip_affine: Optional[Tensor] = None, stats_slope: Optional[float] = 0.1):
    """FPN lateral on channels-last activations: coarse_cl [N,H/2,W/2,Ca], skip_cl [N,H,W,Cb], w [Ca+Cb,cout] ->
    out_cl [N,H,W,cout], or (out_cl, affine [N,cout,3]) with stats_slope."""
    N, hc, wc, Ca = coarse_cl.shape
    Nb, H, W, Cb = skip_cl.shape
    if Nb != N or H != 2 * hc or W != 2 * wc:
        raise ValueError(f"conv2d_fpn_cl: skip {tuple(skip_cl.shape)} is not twice coarse {tuple(coarse_cl.shape)}")
    if tuple(w.shape) != (Ca + Cb, cout):
        raise ValueError(f"conv2d_fpn_cl: weight must be [{Ca + Cb},{cout}], got {tuple(w.shape)}")
    for a, c, nm in ((coarse_affine, Ca, "coarse_affine"), (skip_affine, Cb, "skip_affine")):
        if a is not None and tuple(a.shape) != (N, c, 3):
            raise ValueError(f"conv2d_fpn_cl: {nm} must be [{N},{c},3], got {tuple(a.shape)}")
    dev = skip_cl.device
    lib = _lib.load()
    out = torch.empty((N, H, W, cout), dtype=torch.float32, device=dev)
    partial = (torch.empty((N, lib.cds_fpn_cl_parts(H, W), cout, 2), dtype=torch.float64, device=dev)
               if stats_slope is not None else None)
    check(lib.cds_conv2d_fpn_cl_f32(_dev(coarse_cl, "coarse"), _dev(coarse_affine, "coarse_affine") if coarse_affine is not None else None,
                                    _dev(skip_cl, "skip"), _dev(skip_affine, "skip_affine") if skip_affine is not None else None,
                                    _dev(w, "weight"), out.data_ptr(), partial.data_ptr() if partial is not None else None,
                                    N, Ca, Cb, cout, H, W, _stream(skip_cl)), "cds_conv2d_fpn_cl_f32")
    if stats_slope is None:
        return out
    return out, _reduce_records(partial, N, cout, H, W, stats_slope)[1]


def instnorm_apply_cl(x_cl: Tensor, stats: Tensor, act: int, n_chw: int = 0, cl_from: Optional[int] = 0):
    """InstanceNorm + activation of x_cl [N,H,W,C] for given statistics [N,C,2] float64 -> (out_cl [N - cl_from,H,W,C] for the images
    n >= cl_from, or None with cl_from=None; out_chw [n_chw,C,H,W] for the first n_chw images, or None)."""
    N, H, W, C = x_cl.shape
    if stats.dtype != torch.float64 or tuple(stats.shape) != (N, C, 2):
        raise ValueError(f"instnorm_apply_cl: stats must be float64 [{N},{C},2]")
    want_cl = cl_from is not None and cl_from < N
    if not want_cl and n_chw < 1:
        raise ValueError("instnorm_apply_cl: nothing to write")
    dev = x_cl.device
    out_cl = torch.empty((N - cl_from, H, W, C), dtype=torch.float32, device=dev) if want_cl else None
    out_chw = torch.empty((n_chw, C, H, W), dtype=torch.float32, device=dev) if n_chw > 0 else None
    check(_lib.load().cds_instnorm_apply_cl_f32(_dev(x_cl, "x"), _dev64(stats, "stats"), out_cl.data_ptr() if want_cl else None,
                                                out_chw.data_ptr() if n_chw > 0 else None, N, C, H, W, act, n_chw,
                                                cl_from if want_cl else 0, _stream(x_cl)), "cds_instnorm_apply_cl_f32")
    return out_cl, out_chw


def vis_layer1_cl(entropy: Tensor, ref_nc: Tensor, wpk: Tensor, bias: Tensor) -> Tensor:
    """Visibility CNN layer 1 (model.py:14,51): entropy, ref_nc [V,h,w] -> ReLU(conv3x3(cat) + bias) channels-last [V,h,w,16].
    wpk packed [2,9,16] (BatchNorm folded), bias [16]."""
    V, h, w = entropy.shape
    if tuple(ref_nc.shape) != (V, h, w) or tuple(wpk.shape) != (2, 9, 16) or bias.numel() != 16:
        raise ValueError(f"vis_layer1_cl: need entropy / ref_nc [V,h,w], weight [2,9,16], bias [16]; got {tuple(entropy.shape)}, "
                         f"{tuple(ref_nc.shape)}, {tuple(wpk.shape)}")
    out = torch.empty((V, h, w, 16), dtype=torch.float32, device=entropy.device)
    check(_lib.load().cds_vis_layer1_cl_f32(_dev(entropy, "entropy"), _dev(ref_nc, "ref_nc"), _dev(wpk, "weight"), _dev(bias, "bias"),
                                            out.data_ptr(), V, h, w, _stream(entropy)), "cds_vis_layer1_cl_f32")
    return out


def conv2d_k3_relu_cl(x_cl: Tensor, wsplit: Tensor, bias: Optional[Tensor], head_w: Optional[Tensor] = None,
                      head_b: Optional[Tensor] = None) -> Tensor:
    """3x3, pad 1, 16 -> 16 channels + bias + ReLU on channels-last activations in split-bf16 arithmetic on the matrix cores
    (visibility CNN layers 2 / 3); x_cl [N,H,W,16] -> [N,H,W,16], or with head_w [16] / head_b [1] the 1x1 head + sigmoid: [N,H,W]."""
    N, H, W, C = x_cl.shape
    if C != 16:
        raise ValueError(f"conv2d_k3_relu_cl: need [N,H,W,16], got {tuple(x_cl.shape)}")
    if (head_w is None) != (head_b is None) or (head_w is not None and (head_w.numel() != 16 or head_b.numel() != 1)):
        raise ValueError("conv2d_k3_relu_cl: head_w [16] and head_b [1] go together")
    if wsplit.dtype != torch.int16 or wsplit.numel() != 2 * 3 * 1 * 3 * 64 * 8:
        raise ValueError("conv2d_k3_relu_cl: wsplit must be split_pack_dynconv([w]) of a [16,16,3,3] weight")
    out = torch.empty((N, H, W) if head_w is not None else (N, H, W, 16), dtype=torch.float32, device=x_cl.device)
    check(_lib.load().cds_conv2d_k3_relu_cl_f32(_dev(x_cl, "x"), wsplit.data_ptr(), _dev(bias, "bias") if bias is not None else None,
                                                _dev(head_w, "head_w") if head_w is not None else None,
                                                _dev(head_b, "head_b") if head_b is not None else None,
                                                out.data_ptr(), N, C, H, W, _stream(x_cl)), "cds_conv2d_k3_relu_cl_f32")
    return out


def depth_fusion(ref_depth: Tensor, ref_conf: Tensor, src_depths: Tensor, src_confs: Tensor, cams: Tensor,
                 prob_thresh, dist_thresh: float, depth_thresh: float, view_thresh: float,
                 want_view_masks: bool = False):
    """Depth-map filtering + average fusion for one reference view (fusion.py:75-114, test.py:334-351).
    ref_depth [h,w], ref_conf [3,h,w], src_depths [V,h,w], src_confs [V,3,h,w], cams [V,100] (device, layout in
    include/cds_mvsnet_hip.h) -> (fused [h,w], mask [h,w] in {0,1}, points [3,h,w], view_masks [V,h,w] | None)."""
    V, h, w = src_depths.shape
    if tuple(ref_depth.shape) != (h, w) or tuple(ref_conf.shape) != (3, h, w) or tuple(src_confs.shape) != (V, 3, h, w) \
            or tuple(cams.shape) != (V, 100):
        raise ValueError("depth_fusion: inconsistent shapes")
    dev = ref_depth.device
    fused = torch.empty((h, w), dtype=torch.float32, device=dev)
    mask = torch.empty((h, w), dtype=torch.float32, device=dev)
    points = torch.empty((3, h, w), dtype=torch.float32, device=dev)
    vm = torch.empty((V, h, w), dtype=torch.float32, device=dev) if want_view_masks else None
    th = torch.tensor([float(p) for p in prob_thresh], dtype=torch.float32)
    if th.numel() != 3:
        raise ValueError("depth_fusion: three confidence thresholds expected")
    check(_lib.load().cds_depth_fusion_f32(_dev(ref_depth, "ref_depth"), _dev(ref_conf, "ref_conf"),
                                           _dev(src_depths, "src_depths"), _dev(src_confs, "src_confs"),
                                           _dev(cams, "cams"), fused.data_ptr(), mask.data_ptr(), points.data_ptr(),
                                           vm.data_ptr() if vm is not None else None, V, h, w, _host(th, "prob_thresh"),
                                           float(dist_thresh), float(depth_thresh), float(view_thresh),
                                           _stream(fused)), "cds_depth_fusion_f32")
    return fused, mask, points, vm


def _refine_range(lo, hi, device) -> Tensor:
    """(depth_min, depth_max, interval) of the Refinement kernels: a 3-element device tensor (geometry block) with hi None, or two
    Python floats = limits that are already in interval units (interval 1: module.py's network on its own)."""
    rng = geo(lo if hi is None else [lo, hi, 1.0], device, "depth_range")
    if rng.numel() < 3:
        raise ValueError("Refinement depth range: (depth_min, depth_max, interval) needs 3 values")
    return rng


def depth_affine(depth: Tensor, lo, hi=None) -> Tensor:
    """(depth / ival - lo') / (hi' - lo') * 10 with lo' = lo / ival, hi' = hi / ival (models/model.py:213-216 + the Refinement pre-scale,
    module.py:353-355).  (lo, hi, ival): see _refine_range."""
    out = torch.empty_like(depth)
    rng = _refine_range(lo, hi, depth.device)
    check(_lib.load().cds_depth_affine_f32(_dev(depth, "depth"), out.data_ptr(), depth.numel(), _dev(rng, "depth_range"),
                                           _stream(depth)), "cds_depth_affine_f32")
    return out


def deconv2d_k3s2(x: Tensor, wpk: Tensor, bias: Optional[Tensor], act: int = ACT_NONE, out: Optional[Tensor] = None) -> Tensor:
    """ConvTranspose2d k3 s2 p1 op1: x [Cin,H,W], wpk packed [Cin,9,8] -> [8,2H,2W] (written into `out` if given)."""
    Cin, H, W = x.shape
    if tuple(wpk.shape) != (Cin, 9, 8):
        raise ValueError(f"deconv2d_k3s2: packed weight must be [{Cin},9,8], got {tuple(wpk.shape)}")
    if out is None:
        out = torch.empty((8, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    elif out.numel() != 8 * 4 * H * W:
        raise ValueError("deconv2d_k3s2: bad output buffer")
    check(_lib.load().cds_deconv2d_k3s2_f32(_dev(x, "x"), _dev(wpk, "weight"), _dev(bias, "bias") if bias is not None else None,
                                            _dev(out, "out"), Cin, 8, H, W, act, _stream(x)), "cds_deconv2d_k3s2_f32")
    return out


def refine_finish(d_norm: Tensor, res: Tensor, lo, hi=None) -> Tensor:
    """(((bilinear x2, align_corners=True)(d_norm [h,w]) + res [2h,2w]) / 10 * (hi' - lo') + lo') * ival   (module.py:366-368,
    models/model.py:218).  (lo, hi, ival): see _refine_range."""
    h, w = d_norm.shape
    if tuple(res.shape) != (2 * h, 2 * w):
        raise ValueError("refine_finish: res must be [2h,2w]")
    out = torch.empty_like(res)
    rng = _refine_range(lo, hi, res.device)
    check(_lib.load().cds_refine_finish_f32(_dev(d_norm, "d_norm"), _dev(res, "res"), out.data_ptr(), h, w, _dev(rng, "depth_range"),
                                            _stream(res)), "cds_refine_finish_f32")
    return out


def curvature_stats(a: Tensor, b: Tensor, c: Tensor) -> Tuple[Tensor, Tensor]:
    """(a^2 + b^2 + c^2) / 3 and |c| of the three curvature maps of a FeatureNet level, one launch."""
    if a.shape != b.shape or a.shape != c.shape:
        raise ValueError("curvature_stats: shape mismatch")
    s, m = torch.empty_like(a), torch.empty_like(a)
    check(_lib.load().cds_curvature_stats_f32(_dev(a, "a"), _dev(b, "b"), _dev(c, "c"), s.data_ptr(), m.data_ptr(), a.numel(),
                                              _stream(a)), "cds_curvature_stats_f32")
    return s, m


def pair_mean(x: Tensor, V: int) -> Tensor:
    """x [2V,...] -> [V,...]: (x[v] + x[V+v]) / 2 (per-pair norm-curvature mean, model.py:59)."""
    if x.shape[0] != 2 * V:
        raise ValueError("pair_mean: leading dimension must be 2V")
    out = torch.empty((V,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(_lib.load().cds_pair_mean_f32(_dev(x, "x"), out.data_ptr(), V, out[0].numel(), _stream(x)), "cds_pair_mean_f32")
    return out


def view_mean(x: Tensor) -> Tensor:
    """x [V,...] -> mean over the leading (view) dimension, summed in view order (model.py:60,79)."""
    out = torch.empty(tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(_lib.load().cds_view_mean_f32(_dev(x, "x"), out.data_ptr(), x.shape[0], out.numel(), _stream(x)), "cds_view_mean_f32")
    return out


def feat_target(hyp: Tensor, gt: Tensor, interval: Tensor, scale: float, thresh: float) -> Tensor:
    """Targets of the feature-distance loss (models/model.py:202-207): hyp [B,D,h,w], gt [B,h,w], interval [B] (device) ->
    [B,D+1,h,w] = |hyp - gt| / (interval x scale) < thresh, last plane 1."""
    B, D, h, w = hyp.shape
    hyp, gt, interval = hyp.detach().float().contiguous(), gt.detach().float().contiguous(), interval.detach().float().contiguous()
    if tuple(gt.shape) != (B, h, w) or interval.numel() != B:
        raise ValueError(f"feat_target: gt {tuple(gt.shape)} / interval {tuple(interval.shape)} do not match hyp {tuple(hyp.shape)}")
    out = torch.empty((B, D + 1, h, w), dtype=torch.float32, device=hyp.device)
    check(_lib.load().cds_feat_target_f32(hyp.data_ptr(), gt.data_ptr(), interval.data_ptr(), float(scale), float(thresh), B, D, h * w,
                                          out.data_ptr(), _stream(hyp)), "cds_feat_target_f32")
    return out
