/*
 * cds_mvsnet_hip.h — C ABI of libcdsmvs_hip.so, the MI355X (gfx950) implementation of the
 * CDS-MVSNet plane-sweep hot path.
 *
 * The reference (TruongKhang/cds-mvsnet) is pure Python/PyTorch and has no FFI of its own; its
 * boundary for this path is models.model.CDSMVSNet.forward (models/model.py:140).  Beneath that
 * boundary this library replaces the PyTorch op sequences listed below.  Each entry point cites
 * the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the name ends in _host.  Since round 6 that includes the
 *     per-call GEOMETRY: homographies (`mats`), epipoles and the depth-range scalars are device data - slices of one small
 *     "geometry block" the host writes with a single host -> device copy per forward (cds_mvsnet_amd/geometry.py) - and no
 *     longer by-value arguments, so that a captured hipGraph of the forward / training step is replayed for new cameras by
 *     rewriting that block;
 *   - the caller owns all memory; nothing is allocated, freed or synchronised inside;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are re-entrant across streams;
 *   - return value: 0 on success, a negative hipError_t on a launch failure,
 *     CDS_EINVAL (-1000) for invalid arguments;
 *   - batch: one call handles one batch item (the Python host loops over B).
 *
 * Layouts: feature maps NCHW without the N ([C][h][w]); channels-last copies [h][w][C] where
 * stated; cost volumes [C][D][h][w]; per-pixel hypotheses [D][h][w].
 */
#ifndef CDS_MVSNET_HIP_H
#define CDS_MVSNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CDS_EINVAL (-1000)
#define CDS_MAX_VIEWS 8 /* source views per launch; more are handled by chunked calls */
#define CDS_MAX_IMAGES 16 /* images per batched FeatureNet launch */
#define CDS_STAGE_STATE_WORDS 2080 /* workspace of cds_stage_inputs_f32: 1 + 2 x (<= 1024 + CDS_MAX_VIEWS workgroups), rounded up */

/* activation codes for the conv entry points */
#define CDS_ACT_NONE 0
#define CDS_ACT_RELU 1
#define CDS_ACT_LEAKY01 2 /* LeakyReLU(0.1) */
#define CDS_ACT_SIGMOID 3
#define CDS_ACT_ACCUM 16 /* OR-ed onto an activation code of cds_conv2d_*_f32: out += act(conv) (gradient accumulation) */
#define CDS_ACT_TANH 4

/* flags of cds_warp_aggregate_f32 */
#define CDS_AGG_ACCUMULATE 1 /* add onto the existing volume / vis_sum instead of overwriting */
#define CDS_AGG_NORMALIZE 2  /* divide by (vis_sum + 1e-6) before the store (model.py:74) */
#define CDS_AGG_CHANNELS_LAST 4 /* volume laid out [D][h][w][C] (a voxel's channels contiguous: one 32-byte store per 8 channels) */
/*
 * Sample-position arithmetic of the LDS-staged K1 / K3 kernels (flag of cds_warp_aggregate_f32 and cds_warp_entropy_flags_f32).
 * Absent (what the product path passes by default): the reference's fp32 operation order (correctly rounded divisions, ATen's
 * normalise / de-normalise round trip): sample positions bit-identical to F.grid_sample's.  Set (opt-in): (u, v) = p.xy *
 * rcp(p.z + 1e-6) directly, 5-11 % faster; the positions then differ by up to ~1e-4 px at w = 640 and the volume of sharp
 * feature maps moves by up to 8.4e-5 against the reference - OUTSIDE the 1e-5 parity tolerance (exact mode: 1.2e-7); the depth
 * mean-L1 is unaffected (tests/test_hip_parity.py::test_fast_positions_leave_the_parity_tolerance_at_full_width).  The direct
 * (non-LDS) fallback kernels always use the reference order.
 */
#define CDS_AGG_FAST_POSITIONS 8
#define CDS_WARP_FAST_POSITIONS CDS_AGG_FAST_POSITIONS

/* Library version (major*10000 + minor*100 + patch). */
int cds_version(void);

/* [C][h][w] -> [h][w][C] re-layout of one feature map (source maps are gathered channels-last). */
int cds_chw_to_hwc_f32(const float* src_chw, float* dst_hwc, int C, int h, int w, void* stream);

/*
 * The reference-signature boundary of one stage in one launch.  models/model.py:16-40 hands StageNet.forward a list over the
 * source views of {'ref': (fea, nc_sum, nc), 'src': (fea, nc_sum, _)}; this gathers it into the layouts K1 / K3 read:
 *   ref_feas, src_feas      HOST arrays of V DEVICE pointers, each one [C][h][w] feature map (contiguous fp32)
 *   ref_nc, ref_ncsum, src_ncsum   HOST arrays of V DEVICE pointers to [h][w] maps (model.py:51,59), or NULL with their outputs
 *   ref_chw_out [V][C][h][w]   the stacked reference copies
 *   src_hwc_out [V][h][w][C]   the source maps channels-last (cds_chw_to_hwc_f32 per view)
 *   ref_nc_out  [V][h][w] | NULL;  nc_mean_out [h][w] | NULL = sum_v ((ref_ncsum[v] + src_ncsum[v]) / 2) / V in view order
 *                              (= cds_pair_mean_f32 then cds_view_mean_f32, model.py:59-60)
 *   state [CDS_STAGE_STATE_WORDS] | NULL   device workspace (no initialisation needed); on completion state[0] = max |ref| *
 *                              max |src|: the bound of the normalised volume, the in_bound of cds_conv3d_sf16_f32's first layer
 *                              (NaN if a feature is NaN); the rest holds the per-workgroup maxima a second tiny launch merges
 * V <= CDS_MAX_VIEWS, C in {8, 16, 32}.
 */
int cds_stage_inputs_f32(const float* const* ref_feas, const float* const* src_feas, const float* const* ref_nc,
                         const float* const* ref_ncsum, const float* const* src_ncsum, float* ref_chw_out, float* src_hwc_out,
                         float* ref_nc_out, float* nc_mean_out, float* state, int V, int C, int h, int w, void* stream);

/*
 * homo_warping_3D (models/utils/warping.py:69-104): bilinear, zero padding, align_corners=True.
 *   src_hwc   [h][w][C]   source feature map, channels-last
 *   mat       12 floats   rows of (P_src * P_ref^-1)[:3,:3] then its [:3,3] (warping.py:80-82); DEVICE memory
 *   hyp       [D][h][w] if hyp_per_pixel else [D]
 *   out       [C][D][h][w]
 */
int cds_homo_warp_f32(const float* src_hwc, const float* mat, const float* hyp, float* out,
                      int C, int D, int h, int w, int hyp_per_pixel, void* stream);

/*
 * K1 "warp-correlate-entropy" (warping.py:79-102 + model.py:46-50): for every source view v and
 * pixel, the entropy over D of softmax_D( sum_C ref_v * warp(src_v) ).  Never materialises the
 * warped volume.
 *   ref_chw   [V][C][h][w]  per-pair reference features
 *   src_hwc   [V][h][w][C]  source features, channels-last
 *   mats      [V][12]       homographies as above, DEVICE memory (geometry block)
 *   entropy   [V][h][w]
 */
int cds_warp_entropy_f32(const float* ref_chw, const float* src_hwc, const float* mats,
                         const float* hyp, float* entropy, int V, int C, int D, int h, int w,
                         int hyp_per_pixel, void* stream);
/* the same with flags: CDS_WARP_FAST_POSITIONS (cds_warp_entropy_f32 == flags 0) */
int cds_warp_entropy_flags_f32(const float* ref_chw, const float* src_hwc, const float* mats,
                               const float* hyp, float* entropy, int V, int C, int D, int h, int w,
                               int hyp_per_pixel, int flags, void* stream);

/*
 * K3 "warp-aggregate" (model.py:44-47,57-60,74): volume = sum_v vis_v * (ref_v (x) warp(src_v)),
 * vis_sum = sum_v vis_v, optionally normalised by (vis_sum + 1e-6).  The volume is written once.
 *   vis_w     [V][h][w]
 *   volume    [C][D][h][w], or [D][h][w][C] with CDS_AGG_CHANNELS_LAST (what the split-bf16 CostRegNet kernels read)
 *   vis_sum   [h][w]
 *   flags     CDS_AGG_*
 * V <= CDS_MAX_VIEWS per call.
 */
int cds_warp_aggregate_f32(const float* ref_chw, const float* src_hwc, const float* vis_w,
                           const float* mats, const float* hyp, float* volume, float* vis_sum,
                           int V, int C, int D, int h, int w, int hyp_per_pixel, int flags,
                           void* stream);

/*
 * Row-window forms of K1 / K3 (pixel-slab sharding across GPUs): the reference-side tensors cover rows [y_off, y_off + h) of the
 * hs x w image grid -- ref_chw [V][C][h][w], hyp [D][h][w] (per pixel), vis_w / entropy [V][h][w], volume [C][D][h][w] or
 * [D][h][w][C], vis_sum [h][w] -- while src_hwc [V][hs][w][C] covers the whole grid.  Positions are computed from the global row
 * y_off + y: a window's result equals the same rows of the full-grid call bit for bit.  flags as in the full-grid calls.
 * Returns CDS_EINVAL for shapes outside the LDS-staged kernels (C not in {8, 16, 32}, w < 2).
 */
int cds_warp_entropy_window_f32(const float* ref_chw, const float* src_hwc, const float* mats, const float* hyp,
                                float* entropy, int V, int C, int D, int h, int w, int hs, int y_off, int flags, void* stream);
int cds_warp_aggregate_window_f32(const float* ref_chw, const float* src_hwc, const float* vis_w, const float* mats,
                                  const float* hyp, float* volume, float* vis_sum, int V, int C, int D, int h, int w, int hs,
                                  int y_off, int flags, void* stream);

/*
 * Backward of the un-normalised K3 (training step).  With volume = sum_v vis_v * ref_v (x) warp(src_v):
 *   grad_volume  [C][D][h][w]   incoming gradient
 *   grad_ref     [V][C][h][w]   ACCUMULATED with atomics (partial sums per depth segment): the caller zeroes it first
 *   grad_src_hwc [V][h][w][C]   ACCUMULATED with atomics (bilinear scatter; runs of planes in one texel cell are merged first)
 *   grad_vis     [V][h][w]      ACCUMULATED with atomics (partial sums per channel group and depth segment)
 * The sampling grid has no gradient (built under no_grad, warping.py:79): nothing flows to hypotheses / cameras.
 */
int cds_warp_aggregate_bwd_f32(const float* ref_chw, const float* src_hwc, const float* vis_w,
                               const float* mats, const float* hyp, const float* grad_volume,
                               float* grad_ref, float* grad_src_hwc, float* grad_vis, int V, int C, int D,
                               int h, int w, int hyp_per_pixel, void* stream);
/* The training step's epilogue of K3 (models/model.py:56-78) in one launch, and its backward in two: volume = volume_sum /
 * (sum_v vis + 1e-6), feat_distance[d] = sum_c volume_sum[c][d] / (sum_v vis + 1e-6), plane D from gt_sum [C][hw] (K3 at the
 * ground-truth depth) when given.  Backward: g_volume / g_feat_distance may be NULL (no gradient); scratch D x hw floats. */
int cds_volume_finish_f32(const float* volume_sum, const float* gt_sum, const float* vis, int V, int C, int D, int hw, float* volume,
                          float* feat_distance, void* stream);
int cds_volume_finish_bwd_f32(const float* g_volume, const float* g_feat_distance, const float* volume_sum, const float* gt_sum,
                              const float* vis, int V, int C, int D, int hw, float* g_volume_sum, float* g_gt_sum, float* g_vis,
                              float* scratch, void* stream);

/* volume[c][d][p] /= (vis_sum[p] + 1e-6)  (model.py:74) — the finalisation after a view-shard
 * all-reduce of partial sums. */
int cds_volume_normalize_f32(float* volume, const float* vis_sum, int C, int D, int hw, void* stream);

/* The same division for a channels-last partial volume [D][h][w][C] (C % 4 == 0): every channel of voxel (d, p) is divided
 * by (vis_sum[p] + 1e-6).  Used after the view-shard all-reduce (model.py:74). */
int cds_volume_normalize_cl_f32(float* volume, const float* vis_sum, int C, int D, int hw, void* stream);

/*
 * K5 (model.py:90-92, module.py:373-391): softmax over D, depth = sum p*hyp, confidence =
 * sum of p over [i-1,i+2] at i = clamp(trunc(sum p*index),0,D-1).  prob (may be NULL) receives
 * the softmax volume [D][h][w].
 */
int cds_softargmin_conf_f32(const float* prob_pre, const float* hyp, float* depth, float* conf,
                            float* prob, int D, int h, int w, int hyp_per_pixel, void* stream);

/*
 * K6 (module.py:394-439 + model.py:176-193): per-pixel hypotheses of a cascade stage.
 *   prev_depth [hp][wp]  previous stage depth map; it is bilinearly (align_corners=False)
 *                        upsampled to H x W, sampled at  up - ((D-1)/2)*interval + k*interval,
 *                        clamped to [dmin,dmax] and resized trilinearly to [D][H/scale][W/scale].
 *   out        [D][H/scale][W/scale]
 *   interval   1 float, DEVICE: the stage's hypothesis spacing (ratio x depth interval)
 *   depth_range 2 floats, DEVICE: (dmin, dmax)
 */
int cds_depth_hypotheses_f32(const float* prev_depth, float* out, int D, int hp, int wp, int H, int W,
                             int scale, const float* interval, const float* depth_range, void* stream);

/* First-stage planes: out[k][y][x] = lo + k*((hi-lo)/(D-1))  (module.py:425-433); depth_range = (lo, hi), DEVICE. */
int cds_depth_planes_f32(float* out, int D, int h, int w, const float* depth_range, void* stream);

/*
 * K4 (module.py:80-116,270-315): 3x3x3 convolution, padding 1, stride 1 or 2, with fused
 * per-channel bias (folded BatchNorm3d), optional ReLU and optional residual added AFTER the
 * activation (module.py:311-313).
 *   x [Cin][D][H][W]; bias [Cout] or NULL; skip like out or NULL
 *   weight PACKED [Cin][27][Cout] (tap = (kz*3+ky)*3+kx, cout fastest) — i.e. PyTorch's
 *          [Cout][Cin][3][3][3] permuted (1,2,3,4,0); Cout must be 1 or a multiple of 8
 *   out [Cout][Do][Ho][Wo], Xo = (X-1)/stride+1
 */
int cds_conv3d_k3_f32(const float* x, const float* weight, const float* bias, const float* skip,
                      float* out, int Cin, int Cout, int D, int H, int W, int stride, int act,
                      void* stream);

/*
 * Same convolution at stride 1 for Cin % 16 == 0, Cout % 16 == 0, W % 4 == 0 on the matrix cores with a
 * channels-last LDS tile (one 16-byte LDS read feeds four MFMAs).
 *   weight_cl PACKED [27][Cout][Cin] (cin fastest) = PyTorch's [Cout][Cin][3][3][3] permuted (2,3,4,0,1)
 * Returns CDS_EINVAL for shapes it does not cover (callers fall back to cds_conv3d_k3_f32).
 */
int cds_conv3d_k3_cl_f32(const float* x, const float* weight_cl, const float* bias, const float* skip,
                         float* out, int Cin, int Cout, int D, int H, int W, int act, void* stream);

/* K4, split-bf16 arithmetic on channels-last volumes (csrc/conv3d_sbf.hip): 3x3x3 convolution, pad 1, stride 1 | 2
 * (+bias +ReLU +residual).  Every fp32 operand is split exactly into three bf16 terms and multiplied as six error-compensated
 * partial products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32-class error (dropped terms <= 2^-23 |a b| per
 * product) at the bf16 matrix rate.  Replaces models/module.py:80-116 (Conv3d + BatchNorm3d(eval, folded) + ReLU) like
 * cds_conv3d_k3_f32, on x [D][H][W][Cin] -> out [Do][Ho][Wo][Cout] (a voxel's channels contiguous).
 * weight_split: int16 [Cin/8][7][ceil(Cout/16)][3][64][8] packed by the host (ops.split_pack_conv3d).
 * stride == CDS_SBF_PAIR: stride 1 with Cout == 8 and pair-packed weights [Cin/8][9][1][3][64][8] (ops.split_pack_conv3d_pair):
 * an MFMA column is a pair of x-adjacent voxels, its rows (x parity, cout), so no matrix row multiplies padding.
 * Needs Cin % 8 == 0, Cout % 4 == 0, Cout <= 64; CDS_EINVAL otherwise. */
#define CDS_SBF_PAIR 101
int cds_conv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                       int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream);
/* The same convolution in SPLIT-F16 arithmetic: every fp32 operand as TWO fp16 terms of (operand x a power-of-two tensor scale), three
 * partial products per K-step on v_mfma_f32_16x16x32_f16, fp32 accumulate, exact rescaling in the epilogue - fp32-class error (the
 * representation error is ~0.3x the rounding error of an fp32 convolution) at half the matrix-pipe work of split-bf16.  Shapes: those
 * of the z-marching kernels (Cin, Cout, stride) in {(8 | 16 | 32, 8, PAIR), (16, 16, 1), (8, 16, 2), (16, 32, 2)}; CDS_EINVAL otherwise.
 *   weight_split  the split-bf16 layout with fp16 terms (hi, lo, unused) of weight * w_scale (ops.split_pack_conv3d / _pair (f16=True))
 *   w_inv_scale   1 / w_scale, a power of two
 *   in_bound      DEVICE scalar >= max |x| (the producing kernel's out_bound, or any upper bound: it fixes the input's scale)
 *   out_bound     DEVICE scalar that receives max |out| by an atomic maximum (zero it before the call), or NULL */
int cds_conv3d_sf16_f32(const float* x, const void* weight_split, const float* bias, float* out, int Cin, int Cout, int D, int H, int W,
                        int stride, int act, const float* in_bound, float w_inv_scale, float* out_bound, void* stream);


/* ConvTranspose3d k3 s2 p1 op1 (+bias +ReLU +residual) in the same split-bf16 arithmetic, channels-last: x [D][H][W][Cin]
 * -> out [2D][2H][2W][Cout].  Replaces models/module.py:125-160 (Deconv3d + BatchNorm3d(eval, folded) + ReLU + the U-Net
 * skip addition of :310-312).  weight_split from ops.split_pack_deconv3d.  Cout in {8, 16, 32}, Cin % 8 == 0.
 * out_planar != 0: the output is written [Cout][2D][2H][2W] (for a planar consumer: conv11 -> prob); skip stays channels-last. */
int cds_deconv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                         int Cin, int Cout, int D, int H, int W, int act, int out_planar, void* stream);

/* ConvTranspose3d(32 -> 16, k3 s2 p1 op1) + folded BN shift + ReLU + residual (conv9 of CostRegNet: models/module.py:125-160, :496)
 * as a z-marching kernel with one consumer wave per output parity class and that class's weights resident in registers
 * (csrc/deconv3d_zm.hip).  Same arithmetic (split-bf16) and tensors as cds_deconv3d_sbf_f32: x [D][H][W][32] channels-last ->
 * out [2D][2H][2W][16], skip like out or NULL; weight_cls from ops.split_pack_deconv_cls (int16 [8][4][2][3][64][8]).
 * Cin == 32 and Cout == 16 only (CDS_EINVAL otherwise). */
int cds_deconv3d_zm_f32(const float* x, const void* weight_cls, const float* bias, const float* skip, float* out,
                        int Cin, int Cout, int D, int H, int W, int act, void* stream);
/* conv7 (Cout = 32) and conv9 (32 -> 16, z-marching) in SPLIT-F16 arithmetic (see cds_conv3d_sf16_f32 for the operands: weights from the
 * packers with f16=True, w_inv_scale, in_bound / out_bound device scalars). */
int cds_deconv3d_sf16_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out, int Cin, int Cout,
                          int D, int H, int W, int act, const float* in_bound, float w_inv_scale, float* out_bound, void* stream);
int cds_deconv3d_zm_sf16_f32(const float* x, const void* weight_cls, const float* bias, const float* skip, float* out, int Cin, int Cout,
                             int D, int H, int W, int act, const float* in_bound, float w_inv_scale, float* out_bound, void* stream);


/* The tail of CostRegNet in one launch: conv11 = ConvTranspose3d(16 -> 8, k3 s2 p1 op1) + BatchNorm3d(eval, folded) + ReLU
 * (models/module.py:125-160, :495), the residual `conv0 + conv11(x)` (:498) and prob = Conv3d(8 -> 1, k3, p1, bias=False)
 * (:499).  x [D][H][W][16] channels-last input cells, skip [2D][2H][2W][8] channels-last, out [2D][2H][2W] fp32.  The 8-channel
 * full-resolution volume between the two layers stays in LDS (z-marching workgroups, csrc/deconv_prob_zm.hip); the transposed
 * convolution runs in split-bf16 arithmetic on the matrix cores, prob in packed fp32 on the VALU.
 * weight_split from ops.split_pack_deconv_prob (int16 [2][5][3][64][8]), bias [8], prob_table from ops.pack_prob_table
 * (float [3 kx][2][3 ky][3 kz][4]). */
int cds_deconv_prob_zm_f32(const float* x, const void* weight_split, const float* bias, const float* skip,
                           const float* prob_table, float* out, int D, int H, int W, void* stream);
/* The fused tail with its transposed convolution in SPLIT-F16 arithmetic (weights from ops.split_pack_deconv_prob(..., f16=True);
 * in_bound: DEVICE scalar >= max |x|, conv9's out_bound). */
int cds_deconv_prob_zm_sf16_f32(const float* x, const void* weight_split, const float* bias, const float* skip, const float* prob_table,
                                float* out, int D, int H, int W, const float* in_bound, float w_inv_scale, void* stream);


/*
 * K4 (module.py:125-160): ConvTranspose3d k=3, stride 2, padding 1, output_padding 1 (doubles
 * D,H,W) + bias + activation + residual.
 *   weight PACKED [Cin][27][Cout] — PyTorch's transposed-conv layout [Cin][Cout][3][3][3]
 *          permuted (0,2,3,4,1) (no flip: out[2*zi-1+kz] += x[zi]*w[kz]); Cout multiple of 8
 *   x [Cin][D][H][W] -> out [Cout][2D][2H][2W]
 */
int cds_deconv3d_k3s2_f32(const float* x, const float* weight, const float* bias, const float* skip,
                          float* out, int Cin, int Cout, int D, int H, int W, int act, void* stream);

/*
 * Direct 2D convolution (vis CNN model.py:14; FeatureNet branches dynamic_conv.py:112,116;
 * module.py:28-71).  Square kernel k in {1,3,5,7,11} at stride 1, k = 3 at stride 2, zero padding.
 *   x [N][Cin][H][W]; bias [Cout] or NULL; out [N][Cout][Ho][Wo]
 *   weight PACKED [Cin][k*k][CoutP], CoutP = Cout rounded up to a multiple of 8 (zero padded),
 *          cout fastest — PyTorch's [Cout][Cin][k][k] permuted (1,2,3,0)
 */
int cds_conv2d_f32(const float* x, const float* weight, const float* bias, float* out, int N, int Cin,
                   int Cout, int H, int W, int k, int stride, int pad, int act, void* stream);
/*
 * Same convolution with the producing layer's InstanceNorm + LeakyReLU applied on load:
 *   in_affine [N][Cin][3] = (alpha, beta, slope) per image and input channel (device) or NULL; every in-bounds input
 *   value v is replaced by t = v * alpha + beta, t > 0 ? t : t * slope before it is multiplied (zero padding stays zero).
 *   (alpha, beta, slope) = (1, 0, 1) leaves a channel untouched.  Built by cds_instnorm_affine_f32.
 */
int cds_conv2d_affine_f32(const float* x, const float* in_affine, const float* weight, const float* bias, float* out,
                          int N, int Cin, int Cout, int H, int W, int k, int stride, int pad, int act, void* stream);

/*
 * 3x3 convolution, padding 1, 16 -> 16 channels, on the matrix cores: the inner ConvBn2d layers of the visibility CNN
 * (model.py:14, BN folded by the caller).  x [N][16][H][W], W % 4 == 0;
 *   weight_cl [9][16][16] = (tap ky*3+kx, cout, cin)  — PyTorch's [Cout][Cin][3][3] permuted (2,3,0,1);  bias [16] or NULL;
 *   act = CDS_ACT_NONE | CDS_ACT_RELU.
 * head_w / head_b NULL: out [N][16][H][W].  Otherwise (head_w [16], head_b [1], device) the CNN's 1x1 head follows in the
 * same kernel: out [N][H][W] = sigmoid(head_b + sum_c head_w[c] * act(conv + bias)[c])   (nn.Conv2d(16,1,1) + Sigmoid).
 * fp32 inputs and accumulation (v_mfma_f32_16x16x4_f32); the summation order differs from cds_conv2d_f32 (~1e-6).
 */
int cds_conv2d_k3_c16_f32(const float* x, const float* weight_cl, const float* bias, const float* head_w,
                          const float* head_b, float* out, int N, int H, int W, int act, void* stream);

/*
 * K7 on the matrix cores (csrc/conv2d_sbf.hip): ALL branch convolutions of one DynamicConv (models/dynamic_conv.py:112,116:
 * convs[k] and the 3-channel att_convs[k] of every kernel size k, concatenated to Co3 = Cout + 3 output channels per branch)
 * from one staged input tile, stride 1, "same" padding, in split-bf16 arithmetic (every fp32 operand split exactly into three
 * bf16 terms, six error-compensated partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulation: fp32-class error).
 *   x [N][Cin][H][W]; in_affine [N][Cin][3] or NULL (normalise-on-load like cds_conv2d_affine_f32); bias [nb][Co3] or NULL
 *   weight_split: int16 [Cin/8][sum_b ceil(k_b^2/4)][ceil(Co3/16)][3][64][8] (ops.split_pack_dynconv)
 *   out [nb][N][Co3][H][W] (the `branches` tensor of cds_dynconv_blend_*_f32); ksizes: HOST array of nb kernel sizes in {1,3,5,7}
 * Covers Cin % 8 == 0, Co3 <= 48, W % 4 == 0, (nb, ceil(Co3/16)) in {(3,1), (3,2), (2,1), (2,2), (2,3)}; CDS_EINVAL otherwise.
 */
int cds_dynconv_branches_sbf_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                                 float* out, int N, int Cin, int Co3, int H, int W, const int* ksizes, int nb, void* stream);

/* One DynamicConv (models/dynamic_conv.py:97-122) in ONE kernel: the branch convolutions of cds_dynconv_branches_sbf_f32 with
 * the epilogue of cds_dynconv_blend_stats_f32 (epipolar projection of the curvature responses, 1x1 MLP, softmax(./T), blend,
 * InstanceNorm records of the result) applied to the accumulators: the [K][N][Cout + 3] branch tensor never reaches HBM.
 *   out [N][Cout][H][W] (before its InstanceNorm), norm_curv [N][H][W]
 *   partial: 8-byte aligned scratch of 2 * N * parts * Cout doubles, parts = cds_dynconv_fused_parts(H, W); reduce with
 *            cds_instnorm_reduce_f32
 *   w1 [4][K], b1 [4], w2 [K][4]: the attention MLP with its BatchNorm folded in; epipoles [N][2] (DEVICE) (pixels, this resolution)
 * K in {2, 3}, kernel sizes in {1, 3, 5, 7}, Cin % 8 == 0, Cout + 3 <= 48, Cout % 16 <= 13, W % 4 == 0, N <= CDS_MAX_IMAGES. */
int cds_dynconv_fused_sbf_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                              const float* w1, const float* b1, const float* w2, const float* epipoles,
                              float temperature, float* out, float* norm_curv, double* partial, int N, int Cin, int Cout,
                              int H, int W, const int* ksizes, int nb, void* stream);
int cds_dynconv_fused_parts(int H, int W);

/* Visibility-CNN layers 2 and 3 (models/model.py:14: Conv2d 16 -> 16 k3 p1 + BatchNorm(folded) + ReLU; the last one followed by
 * the 1x1 head 16 -> 1 + sigmoid) in split-bf16 arithmetic on the bf16 matrix cores (the same kernel as the DynamicConv
 * branches, one branch).  x [N][Cin][H][W], weight_split = ops.split_pack_dynconv([w]) with w [16][Cin][3][3], bias [16];
 * head_w [16] / head_b [1] or both NULL; out [N][16][H][W], or [N][H][W] with the head.  Cin % 8 == 0, W % 4 == 0. */
int cds_conv2d_k3_relu_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* head_w,
                               const float* head_b, float* out, int N, int Cin, int H, int W, void* stream);

/*
 * FPN lateral connection (module.py:253-254, 260-261): the 1x1 convolution of
 *   cat(interpolate(coarse, scale_factor=2, mode="nearest"), skip)
 * without materialising the up-sampled tensor or the concatenation.
 *   coarse [N][Ca][H/2][W/2], skip [N][Cb][H][W] (H, W even), out [N][Cout][H][W]
 *   weight PACKED [Ca + Cb][CoutP] (coarse channels first, as in the concatenation)
 *   coarse_affine [N][Ca][3] / skip_affine [N][Cb][3]: normalise-on-load tables like cds_conv2d_affine_f32, or NULL.
 * Results are bit-identical to cds_conv2d_affine_f32 (k = 1) on the materialised concatenation.
 *   partial: NULL, or a scratch of 4 * N * cds_fpn_stats_parts(H, W) * Cout floats that receives the per-wave
 *   InstanceNorm records of `out` (as in cds_dynconv_blend_stats_f32; reduce with cds_instnorm_reduce_f32).
 */
int cds_fpn_stats_parts(int H, int W);
int cds_conv2d_fpn_f32(const float* coarse, const float* coarse_affine, const float* skip, const float* skip_affine,
                       const float* weight, float* out, float* partial, int N, int Ca, int Cb, int Cout, int H, int W,
                       void* stream);

/*
 * K7 epilogue of DynamicConv (dynamic_conv.py:97-122) for a batch of N images (each with its own epipole):
 * epipolar projection of the K 3-channel curvature responses, 1x1 MLP (K->4, folded BN, ReLU, 4->K),
 * softmax(./temperature), blend.
 *   branches [K][N][Cout+3][H][W]  per kernel size: the Cout responses of convs[k] followed by the 3 responses of
 *                                  att_convs[k] (one batched cds_conv2d_f32 per size)
 *   w1 [4][K], b1 [4] (BN folded), w2 [K][4]   (device pointers); K in {2,3}
 *   epipoles [N][2] (DEVICE)           (x, y) in pixels of this resolution; N <= CDS_MAX_IMAGES
 *   out [N][Cout][H][W]; norm_curv [N][H][W]
 */
int cds_dynconv_blend_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                          const float* epipoles, float temperature, float* out, float* norm_curv,
                          int N, int K, int Cout, int H, int W, void* stream);
/*
 * Same, when the first n_shared images of the batch are copies of ONE image (the reference image of every pair,
 * model.py:154-161) that only differ in their epipole: the responses of convs[k] / att_convs[k] do not depend on the
 * epipole (dynamic_conv.py:112,116), so they are computed once.
 *   branches [K][N - n_shared + 1][Cout+3][H][W]: slot 0 = the shared image, slot n - n_shared + 1 = image n >= n_shared
 */
int cds_dynconv_blend_shared_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                 const float* epipoles, float temperature, float* out, float* norm_curv, int N,
                                 int K, int Cout, int H, int W, int n_shared, void* stream);

/*
 * cds_dynconv_blend_shared_f32 that also leaves the InstanceNorm statistics of `out`, so that the normalisation that
 * follows every DynamicConv (module.py:53,66-69) needs no second pass over the tensor.  Every wave writes one
 * (sum, sum of squares) fp64 record per channel of the rounded fp32 outputs:
 *   partial: caller-provided, 8-byte aligned scratch of 4 * N * parts * Cout floats, parts = cds_blend_stats_parts(H, W)
 * cds_instnorm_reduce_f32 adds the records in a fixed order (no atomics: bit-reproducible) into
 *   stats  [N][C][2] doubles (sum, sum of squares) — stored in a float* scratch of 4*N*C floats like cds_instnorm_act_f32's
 *   affine [N][C][3] (alpha, beta, slope) as cds_instnorm_affine_f32 would produce, or NULL
 * cds_instnorm_apply_f32 is the second half of cds_instnorm_act_f32 for given statistics.
 */
int cds_blend_stats_parts(int H, int W);
int cds_dynconv_blend_stats_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                const float* epipoles, float temperature, float* out, float* norm_curv,
                                float* partial, int N, int K, int Cout, int H, int W, int n_shared, void* stream);
int cds_instnorm_reduce_f32(const float* partial, int parts, float* stats, float* affine, int N, int C, int H, int W,
                            float slope, void* stream);
int cds_instnorm_apply_f32(const float* x, const float* stats, float* out, int N, int C, int H, int W, int act,
                           int out_hwc, void* stream);

/*
 * K8 (module.py:53,66-69,223,230,232): InstanceNorm2d (no affine, eps 1e-5, biased variance) followed by
 * LeakyReLU(0.1) or tanh, for N images.  x [N][C][H][W]; `stats` is a caller-provided, 8-byte aligned scratch of
 * 4*N*C floats (per-(image,channel) fp64 sum and sum of squares).  If out_hwc is non-zero the result is written
 * channels-last [N][H][W][C].
 */
int cds_instnorm_act_f32(const float* x, float* out, float* stats, int N, int C, int H, int W, int act,
                         int out_hwc, void* stream);
/*
 * InstanceNorm statistics only (module.py:53,66-69): affine [N][C][3] = (1/sqrt(var + 1e-5), -mean/sqrt(var + 1e-5), slope)
 * for consumers that normalise on load (cds_conv2d_affine_f32); stats = scratch of 4*N*C floats like cds_instnorm_act_f32.
 */
int cds_instnorm_affine_f32(const float* x, float* affine, float* stats, int N, int C, int H, int W, float slope,
                            void* stream);

/*
 * FeatureNet on CHANNELS-LAST activations (csrc/feat_cl.hip, round 5; models/module.py:234-267, models/dynamic_conv.py:97-122).
 * Every activation is [N][H][W][C] fp32 (a pixel's channels contiguous); normalise-on-load tables [N][C][3] as above; InstanceNorm
 * records [N][parts][C][2] doubles (sum, sum of squares per record) reduced by cds_instnorm_reduce_f32.  The inference runner
 * (cds_mvsnet_amd.model._FeatureRunner) uses these; the planar entry points above remain for the training path and the
 * visibility CNN.
 *
 * cds_dynconv_cl_f32: one DynamicConv (dynamic_conv.py:97-122) in ONE kernel - all branch convolutions on the bf16 matrix cores in
 *   split-bf16 arithmetic as a transposed implicit GEMM (rows = output channels, columns = pixels) + the blend epilogue on the
 *   accumulators + the InstanceNorm records of the result.
 *     x [N][H][W][C], in_affine [N][C][3] or NULL, weight_split = ops.split_pack_dynconv (as cds_dynconv_branches_sbf_f32),
 *     bias [nb][C + 3] or NULL, w1 [4][nb], b1 [4], w2 [nb][4], epipoles [N][2] (DEVICE) (pixels at this resolution)
 *     out [N][H][W][C] (before its InstanceNorm), norm_curv [N][H][W], partial [N][cds_dynconv_cl_parts(H, W)][C][2]
 *   (C, ksizes) in {(8, 3-5-7), (8, 1-3), (16, 3-5), (16, 1-3), (32, 1-3)}, Cin == Cout == C, N <= CDS_MAX_IMAGES.
 * cds_dynconv_blend_cl_f32: the epilogue alone over a PLANAR branch tensor [3][N - n_shared + 1][8 + 3][H][W] (conv00: 3 input
 *   channels, the VALU branch kernels; the first n_shared images share slot 0) -> out [N][H][W][8], norm_curv, partial
 *   [N][cds_blend_cl_parts(H, W)][8][2].
 * cds_conv2d_k3s2_cl_f32: 3x3, stride 2, pad 1, no bias (downsample1 / downsample2): x [N][H][W][Cin] -> out [N][Ho][Wo][Cout],
 *   weight [9][Cin][Cout] (tap = ky * 3 + kx, cout fastest); (Cin, Cout) in {(8, 16), (16, 32)}.
 * cds_conv2d_fpn_cl_f32: FPN lateral, 1x1 convolution of cat(nearest2x(coarse [N][H/2][W/2][Ca]), skip [N][H][W][Cb]) ->
 *   out [N][H][W][Cout]; weight [Ca + Cb][Cout]; partial NULL or [N][cds_fpn_cl_parts(H, W)][Cout][2];
 *   (Ca, Cb, Cout) in {(32, 16, 16), (16, 8, 8)}.
 * cds_instnorm_stats_cl_f32: records of x [N][H][W][C] -> partial [N][cds_instnorm_stats_cl_parts(H, W)][C][2]; C in {8, 16, 32}.
 * cds_instnorm_apply_cl_f32: InstanceNorm + activation for given statistics [N][C][2] doubles: the images n >= cl_from channels-last
 *   into out_cl [N - cl_from][H][W][C] (or NULL), the first n_chw images planar into out_chw [n_chw][C][H][W] (the reference-view
 *   feature maps K1 / K3 read).
 */
/* Debug aid: synchronise the device and fill the LDS of every CU with `pattern` (see csrc/lib.hip; CDS_DEBUG_POISON_LDS=<hex> makes
 * every entry point do this after its launch). */
int cds_debug_poison_lds(unsigned pattern);

int cds_dynconv_cl_parts(int H, int W);
int cds_dynconv_cl_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias, const float* w1,
                       const float* b1, const float* w2, const float* epipoles, float temperature, float* out,
                       float* norm_curv, double* partial, int N, int C, int H, int W, const int* ksizes, int nb, void* stream);
/* The same DynamicConv in SPLIT-F16 arithmetic (two fp16 terms of value x power-of-two scale, three products per K-step; see
 * cds_conv3d_sf16_f32): weight_split from ops.split_pack_dynconv(..., f16=True), w_inv_scale = 1 / its weight scale; x_bound = a HOST
 * number >= max |input after its affine + LeakyReLU| (sqrt(H W) bounds any InstanceNorm-ed map: Samuelson's inequality). */
int cds_dynconv_cl_sf16_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias, const float* w1,
                            const float* b1, const float* w2, const float* epipoles, float temperature, float* out, float* norm_curv,
                            double* partial, int N, int C, int H, int W, const int* ksizes, int nb, float x_bound, float w_inv_scale,
                            void* stream);

/* conv00 of FeatureNet (module.py:209: DynamicConv 3 -> 8, kernel sizes 3 / 7 / 11) in ONE kernel on the matrix cores: x [S][3][H][W]
 * planar images in S = N - n_shared + 1 slots (slot 0 is shown by the first n_shared output images, each with its own epipole),
 * weight_split = ops.split_pack_conv00, bias [3][11] or NULL -> out [N][H][W][8], norm_curv [N][H][W],
 * partial [N][cds_dynconv_cl_parts(H, W)][8][2] doubles. */
int cds_conv00_cl_f32(const float* x, const void* weight_split, const float* bias, const float* w1, const float* b1, const float* w2,
                      const float* epipoles, float temperature, float* out, float* norm_curv, double* partial, int N, int n_shared,
                      int H, int W, void* stream);
/* conv00 in SPLIT-F16 arithmetic (see cds_conv3d_sf16_f32): weight_split from ops.split_pack_conv00(..., f16=True), w_inv_scale = 1 / its
 * weight scale, in_bound a DEVICE scalar >= max |x| (e.g. the images' amax). */
int cds_conv00_cl_sf16_f32(const float* x, const void* weight_split, const float* bias, const float* w1, const float* b1, const float* w2,
                           const float* epipoles, float temperature, float* out, float* norm_curv, double* partial, int N, int n_shared,
                           int H, int W, const float* in_bound, float w_inv_scale, void* stream);

int cds_blend_cl_parts(int H, int W);
int cds_dynconv_blend_cl_f32(const float* branches, const float* w1, const float* b1, const float* w2, const float* epipoles,
                             float temperature, float* out, float* norm_curv, double* partial, int N, int K, int Cout, int H, int W,
                             int n_shared, void* stream);
int cds_conv2d_k3s2_cl_f32(const float* x, const float* in_affine, const float* weight, float* out, int N, int Cin, int Cout, int H,
                           int W, void* stream);
/* The stride-2 units on the matrix cores in SPLIT-F16 arithmetic (round 6): weight_split from ops.split_pack_dynconv([w [Cout,Cin,3,3]],
 * f16=True), w_inv_scale = 1 / its weight scale, x_bound a HOST number >= max |input after its affine + LeakyReLU|. */
int cds_conv2d_k3s2_cl_sf16_f32(const float* x, const float* in_affine, const void* weight_split, float* out, int N, int Cin, int Cout,
                                int H, int W, float x_bound, float w_inv_scale, void* stream);

int cds_fpn_cl_parts(int H, int W);
int cds_conv2d_fpn_cl_f32(const float* coarse, const float* coarse_affine, const float* skip, const float* skip_affine,
                          const float* weight, float* out, double* partial, int N, int Ca, int Cb, int Cout, int H, int W,
                          void* stream);
/* Visibility CNN on channels-last activations (models/model.py:14,51; csrc/feat_cl.hip): layer 1 from the two maps
 * (entropy, ref_nc [V][h][w]; weight packed [2][9][16], bias [16], BatchNorm folded) to [V][h][w][16]; layers 2 / 3 (3x3, 16 -> 16,
 * bias + ReLU, split-bf16 on the matrix cores; weight_split = ops.split_pack_dynconv([w]); with head_w [16] / head_b [1] the 1x1 head
 * + sigmoid follows and out is [N][H][W], else [N][H][W][16]). */
int cds_vis_layer1_cl_f32(const float* entropy, const float* ref_nc, const float* weight, const float* bias, float* out, int V, int H,
                          int W, void* stream);
int cds_conv2d_k3_relu_cl_f32(const float* x, const void* weight_split, const float* bias, const float* head_w, const float* head_b,
                              float* out, int N, int Cin, int H, int W, void* stream);
int cds_instnorm_stats_cl_parts(int H, int W);
int cds_instnorm_stats_cl_f32(const float* x, double* partial, int N, int C, int H, int W, void* stream);
int cds_instnorm_apply_cl_f32(const float* x, const double* stats, float* out_cl, float* out_chw, int N, int C, int H, int W, int act,
                              int n_chw, int cl_from, void* stream);

/*
 * Depth-map filtering + fusion (fusion.py:7-114, test.py:334-351).  Per reference pixel and source
 * view: re-projection through the source depth map, pixel-distance / relative-depth / in-range tests, average fusion.
 *   ref_depth [h][w]; ref_conf [3][h][w]; src_depths [V][h][w]; src_confs [V][3][h][w] (probability filter
 *   conf_k > prob_thresh_host[k] applied to the source depths and to the final mask)
 *   cams [V][100]: per view Kinv_ref(9) Einv_ref(16) E_src(16) K_src(9) Kinv_src(9) Einv_src(16) E_ref(16) K_ref(9)
 *   fused [h][w], mask [h][w] (0/1), points [3][h][w] (world), view_masks [V][h][w] or NULL
 */
int cds_depth_fusion_f32(const float* ref_depth, const float* ref_conf, const float* src_depths,
                         const float* src_confs, const float* cams, float* fused, float* mask, float* points,
                         float* view_masks, int V, int h, int w, const float* prob_thresh_host,
                         float dist_thresh, float depth_thresh, float view_thresh, void* stream);

/*
 * Norm-curvature bookkeeping of one FeatureNet level (module.py:250-251,257-258,264-265) in one launch:
 *   nc_sum[i] = (a[i]^2 + b[i]^2 + c[i]^2) / 3,  nc_abs[i] = |c[i]|   for the three DynamicConv curvature maps of the level
 */
int cds_curvature_stats_f32(const float* a, const float* b, const float* c, float* nc_sum, float* nc_abs, int n,
                            void* stream);
/* Its backward (training): ga = (g_sum / 3) 2 a, gb likewise, gc = (g_sum / 3) 2 c + g_abs sgn(c); g_sum / g_abs may be NULL. */
int cds_curvature_stats_bwd_f32(const float* a, const float* b, const float* c, const float* g_sum, const float* g_abs, float* ga,
                                float* gb, float* gc, int n, void* stream);
/* out[v][i] = (x[v][i] + x[V+v][i]) / 2 for v < V (model.py:59); x [2V][n] */
int cds_pair_mean_f32(const float* x, float* out, int V, int n, void* stream);
/* out[i] = (sum over v of x[v][i]) / V (model.py:60,79); x [V][n] */
int cds_view_mean_f32(const float* x, float* out, int V, int n, void* stream);

/*
 * Refinement network (module.py:318-370) pieces besides its 3x3 Conv+BN+ReLU units (those run on cds_conv2d_f32):
 *   depth_range            3 floats, DEVICE (geometry block): (depth_min, depth_max, ival).  models/model.py:213-216 divides the depth
 *                          and both limits by the depth interval before the network and multiplies the result by it (:218): that is
 *                          folded into the two kernels - lo = depth_min / ival, hi = depth_max / ival (true divisions, as the
 *                          reference's tensor ops); ival = 1 gives module.py's network on its own
 *   cds_depth_affine_f32   out[i] = (depth[i] / ival - lo) / (hi - lo) * 10               (module.py:353-355)
 *   cds_deconv2d_k3s2_f32  ConvTranspose2d k=3, stride 2, padding 1, output_padding 1 (+ bias + activation);
 *                          x [Cin][H][W] -> out [Cout][2H][2W]; weight PACKED [Cin][9][Cout] = PyTorch's
 *                          [Cin][Cout][3][3] permuted (0,2,3,1) (BatchNorm folded in by the caller); Cout = 8
 *   cds_refine_finish_f32  out = (((bilinear x2, align_corners=True)(d_norm [h][w]) + res [2h][2w]) / 10 * (hi - lo) + lo) * ival
 */
int cds_depth_affine_f32(const float* depth, float* out, int n, const float* depth_range, void* stream);
int cds_deconv2d_k3s2_f32(const float* x, const float* weight, const float* bias, float* out, int Cin, int Cout,
                          int H, int W, int act, void* stream);
int cds_refine_finish_f32(const float* d_norm, const float* res, float* out, int h, int w, const float* depth_range,
                          void* stream);

/*
 * Training kernels of the CostRegNet stack (models/module.py:80-160 with BatchNorm3d in training mode).
 * Activations [B][C][D][H][W] fp32 (V = D*H*W), statistics fp64.  Forward convolutions and data gradients use
 * cds_conv3d_k3_f32 / cds_deconv3d_k3s2_f32 (a convolution's data gradient is the transposed convolution and vice versa).
 *   cds_bn3d_stats_f32:       sums[c] = (sum x, sum x^2) over batch and voxels, ADDED onto sums [C][2] (zero it first)
 *   cds_bn3d_norm_f32:        the per-channel step of nn.BatchNorm3d in training mode from those sums (fp64: mean, biased variance,
 *                             scale = gamma / sqrt(var + eps), shift = beta - mean scale; n = elements per channel) done by every
 *                             workgroup, then out = relu?(y * scale[c] + shift[c]) (+ skip): BatchNorm(train) + ReLU + U-Net skip in
 *                             one launch.  Writes scale / shift [C] floats and mean / invstd [C] doubles for the backward and updates
 *                             running_mean / running_var in place with `momentum` (unbiased variance; both may be NULL)
 *   cds_bn3d_bwd_reduce_f32:  sums[c] += (sum g, sum g*y), g = dout * [relu ? y*scale+shift > 0 : 1]
 *   cds_bn3d_bwd_norm_f32:    dgamma, dbeta [C] and dy = g * scale[c] + y * k1[c] + k0[c] (the BatchNorm backward in closed form; k1,
 *                             k0 derived from the sums of cds_bn3d_bwd_reduce_f32 by every workgroup) in one launch
 *   cds_conv3d_wgrad_f32:     dw[a][b][tap] += sum_{batch, o} g[a][o] * xin[b][stride * o - 1 + tap]   (k3, pad 1)
 *                             Conv3d: g = dy, xin = x -> dw [Cout][Cin][27]; ConvTranspose3d: g = x, xin = dy, stride 2 ->
 *                             dw [Cin][Cout][27] (PyTorch's layouts).  dw is accumulated onto: zero it first.
 */
int cds_bn3d_stats_f32(const float* x, double* sums, int B, int C, long long V, void* stream);
int cds_bn3d_norm_f32(const float* y, const double* sums, const float* gamma, const float* beta, double n, double eps, float momentum,
                      float* running_mean, float* running_var, const float* skip, float* out, float* scale, float* shift, double* mean,
                      double* invstd, int B, int C, long long V, int relu, void* stream);
int cds_bn3d_bwd_reduce_f32(const float* dout, const float* y, const float* scale, const float* shift, double* sums, int B,
                            int C, long long V, int relu, void* stream);
int cds_bn3d_bwd_norm_f32(const float* dout, const float* y, const float* scale, const float* shift, const double* sums,
                          const double* mean, const double* invstd, double n, float* dy, float* dgamma, float* dbeta, int B, int C,
                          long long V, int relu, void* stream);
int cds_conv3d_wgrad_f32(const float* g, const float* xin, float* dw, int B, int Ca, int Cb, int Do, int Ho, int Wo, int Di,
                         int Hi, int Wi, int stride, void* stream);

/* ---- training step, 2D stacks (csrc/train2d.hip) ---------------------------------------------------------------
 * Backward kernels of FeatureNet / DynamicConv (models/dynamic_conv.py:97-122, models/module.py:28-71,234-267), the visibility CNN
 * (models/model.py:14,51), Refinement (models/module.py:318-370) and depth_regression (models/module.py:373-379); they replace what
 * torch.autograd + cuDNN derive from those forwards in trainer/trainer.py:69-82.
 *   cds_conv2d_wgrad_f32:     dw[co][ci][ky][kx] += sum_{n,o} g[n][co][o] x[n][ci][stride * o - pad + k]; Conv2d: g = dy; a transposed
 *                             convolution swaps the roles (g = x, x = dy, stride 2) and receives [Cin][Cout][k][k].  dw is accumulated
 *                             onto (zero it first).  k in {1,3,5,7,11} at stride 1, k = 3 at stride 2.
 *   cds_conv2d_dgrad_s2_f32:  data gradient of Conv2d(k 3, stride 2, pad 1), weight [Co][Cin][3][3].  (Stride-1 data gradients are
 *                             cds_conv2d_f32 on flipped / transposed weights, CDS_ACT_ACCUM to sum the branches of a DynamicConv.)
 *   cds_instnorm_bwd_f32:     backward of InstanceNorm2d(eps 1e-5) + LeakyReLU(0.1) | tanh | nothing; stats = the forward's fp64
 *                             (sum, sum of squares) per (image, channel), sums = fp64 scratch of the same size.
 *   cds_dynconv_bn_stats_f32: (mean, rstd) [G][4] of the attention MLP's BatchNorm2d for G groups of N / G images - batch statistics
 *                             from the fp64 moments of the K curvature maps (use_batch), else the running ones; updates the running
 *                             statistics group after group.  mom: fp64 scratch [G][K + K (K + 1) / 2].  epipoles: DEVICE [N][2].
 *   cds_dynconv_blend_train_f32: the DynamicConv epilogue with those statistics: out [N][Cout][H][W], norm_curv [N][H][W].
 *   cds_dynconv_blend_bwd_f32:   its backward.  gbr [K][N][Cout+3][H][W] = gradient of the branch tensor; fp64: sums [G][8] =
 *                             per group (sum g_bn_j, sum g_bn_j xhat_j) (dbeta_j / dgamma_j summed over groups) followed by dw2 [K][4];
 *                             dw1 [4][K].  gnc may be NULL.
 *   cds_softargmin_bwd_f32:   gpre[d] = softmax(prob_pre)_d (hyp_d - depth) gdepth.
 * scratch_zeroed != 0: the fp64 accumulation buffers (sums / mom / dw1) arrive zero-filled (the training step zeroes ONE arena per step
 * instead of ~40 small memsets); 0: the function zeroes them itself. */
int cds_conv2d_wgrad_f32(const float* g, const float* x, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H, int W, int k,
                         int stride, int pad, void* stream);
int cds_conv2d_dgrad_s2_f32(const float* g, const float* w, float* gx, int N, int Co, int Cin, int Ho, int Wo, int H, int W,
                            void* stream);
int cds_instnorm_bwd_f32(const float* gz, const float* y, const double* stats, double* sums, float* gy, int N, int C, int H, int W,
                         int act, int scratch_zeroed, void* stream);
int cds_dynconv_bn_stats_f32(const float* branches, const float* epipoles, const float* w1, double* mom, float* mean, float* rstd,
                             float* running_mean, float* running_var, int N, int G, int K, int Cout, int H, int W, float eps,
                             float momentum, int use_batch, int scratch_zeroed, void* stream);
int cds_dynconv_blend_train_f32(const float* branches, const float* epipoles, const float* w1, const float* w2, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, float temperature, float* out,
                                float* norm_curv, int N, int G, int K, int Cout, int H, int W, void* stream);
int cds_dynconv_blend_bwd_f32(const float* branches, const float* epipoles, const float* w1, const float* w2, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, float temperature, const float* gy,
                              const float* gnc, float* gbr, double* sums, double* dw1, int N, int G, int K, int Cout, int H, int W,
                              int use_batch, int scratch_zeroed, void* stream);
/* bf16 STORAGE of the 2D activations with fp32 accumulation (BASELINE config 5 is labelled bf16; the reference itself trains in fp32,
 * trainer/trainer.py:69-82, so this is an opt-in policy of the training step, train2d_ops.activation_storage("bf16")).  Tensors that
 * live from the forward to the backward pass - layer inputs, DynamicConv branch responses, pre-normalisation maps - are raw bfloat16
 * (unsigned short, round to nearest even); every kernel widens on load and accumulates as its fp32 twin does.
 *   cds_f32_to_bf16:             dst = bfloat16(src), n elements.
 *   cds_instnorm_act_b16_f32:    InstanceNorm2d + activation of cds_instnorm_act_f32 that also STORES: y16 = bfloat16(y) (kept for the
 *                                backward), z = act((y - mean) rstd) as out32 (the transient the next layer's forward kernel reads)
 *                                and out16 = bfloat16(z) (kept); out16 == NULL: no stored output.  strict != 0: the forward continues on
 *                                the stored values too (statistics of y16, out32 = widened out16).
 *   *_xb16 / *_yb16 / *_b16:     the fp32 entry of the same name with that operand read from its bf16-stored form.
 *                                cds_instnorm_bwd_yb16_f32 also takes the stored LeakyReLU output z16 (or NULL): its sign bit is the
 *                                gate the forward took. */
int cds_f32_to_bf16(const float* src, unsigned short* dst, long long n, void* stream);
int cds_instnorm_act_b16_f32(const float* y, unsigned short* y16, float* out32, unsigned short* out16, double* stats, int N, int C,
                             int H, int W, int act, int strict, void* stream);
int cds_conv2d_wgrad_xb16_f32(const float* g, const unsigned short* x, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H, int W,
                              int k, int stride, int pad, void* stream);
int cds_instnorm_bwd_yb16_f32(const float* gz, const unsigned short* y, const unsigned short* z16, const double* stats, double* sums,
                              float* gy, int N, int C, int H, int W, int act, int scratch_zeroed, void* stream);
int cds_dynconv_bn_stats_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, double* mom, float* mean,
                                 float* rstd, float* running_mean, float* running_var, int N, int G, int K, int Cout, int H, int W,
                                 float eps, float momentum, int use_batch, int scratch_zeroed, void* stream);
int cds_dynconv_blend_train_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, const float* w2,
                                    const float* gamma, const float* beta, const float* mean, const float* rstd, float temperature,
                                    float* out, float* norm_curv, int N, int G, int K, int Cout, int H, int W, void* stream);
int cds_dynconv_blend_bwd_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, const float* w2,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd, float temperature,
                                  const float* gy, const float* gnc, float* gbr, double* sums, double* dw1, int N, int G, int K,
                                  int Cout, int H, int W, int use_batch, int scratch_zeroed, void* stream);
/* The weight layouts of cds_conv2d_f32 in one launch: fwd [Cin][k k][CoP] (forward) and / or dgrad [Ca+Cb][k k][CiP] (stride-1 data
 * gradient: taps flipped, channels swapped) from wa [Ca][Cin][k][k] and, optionally, wb [Cb][Cin][k][k] stacked behind it. */
int cds_pack_conv2d_f32(const float* wa, const float* wb, float* fwd, float* dgrad, int Ca, int Cb, int Cin, int k, void* stream);
/* The forward and data-gradient weight layouts of a 3x3x3 CostRegNet layer in one launch.  w [A][B][27]; mode 0: Conv3d stride 1
 * (A = Cout, B = Cin): fwd [Cin][27][Cout], dgrad [Cout][27][Cin] with flipped taps; mode 1: Conv3d stride 2: dgrad [Cout][27][Cin]
 * unflipped (cds_deconv3d_k3s2_f32's layout); mode 2: ConvTranspose3d (A = Cin, B = Cout): fwd [Cin][27][Cout], dgrad [Cout][27][Cin]. */
int cds_pack_conv3d_f32(const float* w, float* fwd, float* dgrad, int A, int B, int mode, void* stream);
int cds_softargmin_bwd_f32(const float* prob_pre, const float* hyp, const float* gdepth, float* gpre, int D, int h, int w,
                           int hyp_per_pixel, void* stream);
/* The attention-MLP gradients of a DynamicConv from cds_dynconv_blend_bwd_f32's fp64 accumulators, one launch:
 * out = [dbeta (4) | dgamma (4) | dw2 [K][4] | dw1 [4][K]] floats (the groups summed in a fixed order). */
int cds_dynconv_bwd_finish_f32(const double* sums, const double* dw1, int G, int K, float* out, void* stream);

/* final_loss on the device (reference: models/losses.py:6-48; replaces ~90 ATen launches forward and ~150 backward per step) and the
 * feature-distance targets (models/model.py:202-207).  Sums in fp64 through per-workgroup records reduced in a fixed order.
 *   cds_loss_records(n):       records a pass over n elements writes (recA: x 4 doubles over B h w pixels; recB: x 1 over B Dp h w).
 *   cds_loss_stage_f32:        passes over one stage: depth, gt, mask [B][hw] (mask > 0.5 selects), norm_curv [B][hw] or NULL, dist / target
 *                              [B][Dp][hw] or both NULL (no feature term: the refined-depth stage), interval [B] on the DEVICE.
 *   cds_loss_final_f32:        total = sum_s weight_s (depth_s + 5 feature_s + 0.1 curvature_s), depth_loss of the last stage (device
 *                              scalars) and scalars [n_stages][4] doubles for the backward; host arrays of n_stages <= 4 entries.
 *   cds_loss_stage_bwd_f32:    gdepth [B][hw], gnc [B][hw] or NULL, gdist [B][Dp][hw] or NULL from the device scalar gtotal.
 *   cds_feat_target_f32:       target [B][D+1][hw] = |hyp - gt| / (di[b] scale) < thresh (planes < D), 1 (plane D); di [B] on the DEVICE. */
int cds_loss_records(long long elements);
int cds_loss_stage_f32(const float* depth, const float* gt, const float* mask, const float* norm_curv, const float* dist,
                       const float* target, const float* interval, int B, int hw, int Dp, double* recA, double* recB, void* stream);
int cds_loss_final_f32(const double* const* recA, const double* const* recB, const long long* pixels, const int* Dp, const float* weight,
                       const int* has_curv, int n_stages, float* total, float* depth_loss, double* scalars, void* stream);
int cds_loss_stage_bwd_f32(const float* depth, const float* gt, const float* mask, const float* dist, const float* target,
                           const float* interval, const float* gtotal, const double* scalars, float weight, int B, int hw, int Dp,
                           float* gdepth, float* gnc, float* gdist, void* stream);
int cds_feat_target_f32(const float* hyp, const float* gt, const float* di, float scale, float thresh, int B, int D, int hw,
                        float* target, void* stream);

/* Running statistics of a BatchNorm call that stacked G groups: r = keep r + sum_g group_weights[g] stat[g][c] for the mean and the
 * variance in one launch; batch_mean / batch_var [G][C], group_weights [G] on the device. */
int cds_bn_running_update_f32(const float* batch_mean, const float* batch_var, const float* group_weights, float keep, int G, int C,
                              float* running_mean, float* running_var, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDS_MVSNET_HIP_H */
