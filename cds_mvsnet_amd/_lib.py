"""ctypes binding of libcdsmvs_hip.so (the C ABI declared in include/cds_mvsnet_hip.h).

The shared object is built in-tree by ``cds_mvsnet_amd/csrc/Makefile`` (``__graft_entry__.build()``)
with hipcc for gfx950.  There is deliberately NO fallback: if the library is missing the import of
the ops fails loudly — the product path never routes through PyTorch or CPU code.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDS_MVSNET_LIB: developer knob to A/B a differently built copy of the same library (scripts/build_variant.sh)
LIB_PATH = os.environ.get("CDS_MVSNET_LIB") or os.path.join(_HERE, "libcdsmvs_hip.so")

# activation / flag codes (mirror include/cds_mvsnet_hip.h)
ACT_NONE, ACT_RELU, ACT_LEAKY01, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
ACT_ACCUM = 16
AGG_ACCUMULATE, AGG_NORMALIZE, AGG_CHANNELS_LAST, AGG_FAST_POSITIONS = 1, 2, 4, 8
WARP_FAST_POSITIONS = AGG_FAST_POSITIONS
MAX_VIEWS = 8
MAX_IMAGES = 16
STAGE_STATE_WORDS = 2080
EINVAL = -1000

P = c_void_p
I = c_int
F = c_float
DB = ctypes.c_double
L = c_longlong

# name -> argtypes; every function returns int
SIGNATURES = {
    "cds_version": [],
    "cds_chw_to_hwc_f32": [P, P, I, I, I, P],
    "cds_stage_inputs_f32": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "cds_homo_warp_f32": [P, P, P, P, I, I, I, I, I, P],
    "cds_warp_entropy_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_warp_entropy_flags_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_warp_aggregate_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_warp_entropy_window_f32": [P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "cds_warp_aggregate_window_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "cds_warp_aggregate_bwd_f32": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_volume_finish_f32": [P, P, P, I, I, I, I, P, P, P],
    "cds_volume_finish_bwd_f32": [P, P, P, P, P, I, I, I, I, P, P, P, P, P],
    "cds_volume_normalize_f32": [P, P, I, I, I, P],
    "cds_volume_normalize_cl_f32": [P, P, I, I, I, P],
    "cds_softargmin_conf_f32": [P, P, P, P, P, I, I, I, I, P],
    "cds_depth_hypotheses_f32": [P, P, I, I, I, I, I, I, P, P, P],
    "cds_depth_planes_f32": [P, I, I, I, P, P],
    "cds_conv3d_k3_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_conv3d_k3_cl_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_conv3d_sbf_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_conv3d_sf16_f32": [P, P, P, P, I, I, I, I, I, I, I, P, F, P, P],
    "cds_deconv3d_sbf_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_deconv3d_sf16_f32": [P, P, P, P, P, I, I, I, I, I, I, P, F, P, P],
    "cds_deconv3d_zm_sf16_f32": [P, P, P, P, P, I, I, I, I, I, I, P, F, P, P],
    "cds_deconv_prob_zm_f32": [P, P, P, P, P, P, I, I, I, P],
    "cds_deconv_prob_zm_sf16_f32": [P, P, P, P, P, P, I, I, I, P, F, P],
    "cds_deconv3d_zm_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_deconv3d_k3s2_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_conv2d_f32": [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "cds_conv2d_affine_f32": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "cds_instnorm_affine_f32": [P, P, P, I, I, I, I, F, P],
    "cds_fpn_stats_parts": [I, I],
    "cds_conv2d_fpn_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_conv2d_k3_c16_f32": [P, P, P, P, P, P, I, I, I, I, P],
    "cds_dynconv_branches_sbf_f32": [P, P, P, P, P, I, I, I, I, I, P, I, P],
    "cds_dynconv_fused_sbf_f32": [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, I, P, I, P],
    "cds_dynconv_fused_parts": [I, I],
    "cds_conv2d_k3_relu_sbf_f32": [P, P, P, P, P, P, I, I, I, I, P],
    "cds_dynconv_blend_f32": [P, P, P, P, P, F, P, P, I, I, I, I, I, P],
    "cds_dynconv_blend_shared_f32": [P, P, P, P, P, F, P, P, I, I, I, I, I, I, P],
    "cds_blend_stats_parts": [I, I],
    "cds_dynconv_blend_stats_f32": [P, P, P, P, P, F, P, P, P, I, I, I, I, I, I, P],
    "cds_instnorm_reduce_f32": [P, I, P, P, I, I, I, I, F, P],
    "cds_instnorm_apply_f32": [P, P, P, I, I, I, I, I, I, P],
    "cds_instnorm_act_f32": [P, P, P, I, I, I, I, I, I, P],
    "cds_debug_poison_lds": [ctypes.c_uint],
    "cds_dynconv_cl_parts": [I, I],
    "cds_dynconv_cl_f32": [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, P, I, P],
    "cds_dynconv_cl_sf16_f32": [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, P, I, F, F, P],
    "cds_conv00_cl_f32": [P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, P],
    "cds_conv00_cl_sf16_f32": [P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, P, F, P],
    "cds_blend_cl_parts": [I, I],
    "cds_dynconv_blend_cl_f32": [P, P, P, P, P, F, P, P, P, I, I, I, I, I, I, P],
    "cds_conv2d_k3s2_cl_f32": [P, P, P, P, I, I, I, I, I, P],
    "cds_conv2d_k3s2_cl_sf16_f32": [P, P, P, P, I, I, I, I, I, F, F, P],
    "cds_fpn_cl_parts": [I, I],
    "cds_conv2d_fpn_cl_f32": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_vis_layer1_cl_f32": [P, P, P, P, P, I, I, I, P],
    "cds_conv2d_k3_relu_cl_f32": [P, P, P, P, P, P, I, I, I, I, P],
    "cds_instnorm_stats_cl_parts": [I, I],
    "cds_instnorm_stats_cl_f32": [P, P, I, I, I, I, P],
    "cds_instnorm_apply_cl_f32": [P, P, P, P, I, I, I, I, I, I, I, P],
    "cds_curvature_stats_f32": [P, P, P, P, P, I, P],
    "cds_curvature_stats_bwd_f32": [P, P, P, P, P, P, P, P, I, P],
    "cds_pair_mean_f32": [P, P, I, I, P],
    "cds_view_mean_f32": [P, P, I, I, P],
    "cds_depth_affine_f32": [P, P, I, P, P],
    "cds_deconv2d_k3s2_f32": [P, P, P, P, I, I, I, I, I, P],
    "cds_refine_finish_f32": [P, P, P, I, I, P, P],
    "cds_bn3d_stats_f32": [P, P, I, I, L, P],
    "cds_bn3d_norm_f32": [P, P, P, P, DB, DB, F, P, P, P, P, P, P, P, P, I, I, L, I, P],
    "cds_bn3d_bwd_reduce_f32": [P, P, P, P, P, I, I, L, I, P],
    "cds_bn3d_bwd_norm_f32": [P, P, P, P, P, P, P, DB, P, P, P, I, I, L, I, P],
    "cds_conv3d_wgrad_f32": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cds_conv2d_wgrad_f32": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cds_conv2d_dgrad_s2_f32": [P, P, P, I, I, I, I, I, I, I, P],
    "cds_instnorm_bwd_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_dynconv_bn_stats_f32": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, I, I, P],
    "cds_dynconv_blend_train_f32": [P, P, P, P, P, P, P, P, F, P, P, I, I, I, I, I, I, P],
    "cds_dynconv_blend_bwd_f32": [P, P, P, P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "cds_f32_to_bf16": [P, P, L, P],
    "cds_instnorm_act_b16_f32": [P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_conv2d_wgrad_xb16_f32": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "cds_instnorm_bwd_yb16_f32": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cds_dynconv_bn_stats_b16_f32": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, I, I, P],
    "cds_dynconv_blend_train_b16_f32": [P, P, P, P, P, P, P, P, F, P, P, I, I, I, I, I, I, P],
    "cds_dynconv_blend_bwd_b16_f32": [P, P, P, P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    "cds_pack_conv2d_f32": [P, P, P, P, I, I, I, I, P],
    "cds_pack_conv3d_f32": [P, P, P, I, I, I, P],
    "cds_softargmin_bwd_f32": [P, P, P, P, I, I, I, I, P],
    "cds_dynconv_bwd_finish_f32": [P, P, I, I, P, P],
    "cds_bn_running_update_f32": [P, P, P, F, I, I, P, P, P],
    "cds_loss_records": [ctypes.c_longlong],
    "cds_loss_stage_f32": [P, P, P, P, P, P, P, I, I, I, P, P, P],
    "cds_loss_final_f32": [P, P, P, P, P, P, I, P, P, P, P],
    "cds_loss_stage_bwd_f32": [P, P, P, P, P, P, P, P, F, I, I, I, P, P, P, P],
    "cds_feat_target_f32": [P, P, P, F, F, I, I, I, P, P],
    "cds_depth_fusion_f32": [P, P, P, P, P, P, P, P, P, I, I, I, P, F, F, F, P],
}

_lib = None


class CdsLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises CdsLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CdsLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "cds_mvsnet_amd has no CPU / PyTorch fallback.")
    import torch  # noqa: F401  (ensures torch's libamdhip64.so.7 is the one the loader binds to)

    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise CdsLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == EINVAL:
        raise ValueError(f"{what}: invalid argument (CDS_EINVAL)")
    raise RuntimeError(f"{what}: HIP error {-rc}")
