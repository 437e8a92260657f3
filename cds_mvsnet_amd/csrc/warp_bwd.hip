// Backward of the (un-normalised) warp-aggregate, for the training step.
//   forward:  volume[c][d][p] = sum_v vis_v[p] * ref_v[c][p] * warp_v[c][d][p],  warp = bilinear sample of src_v
//   backward: grad_ref_v[c][p]   = vis_v[p] * sum_d g[c][d][p] * warp_v[c][d][p]
//             grad_vis_v[p]      = sum_{c,d} g[c][d][p] * ref_v[c][p] * warp_v[c][d][p]
//             grad_src_v[tap][c] += g[c][d][p] * ref_v[c][p] * vis_v[p] * w_tap      (scatter-add, atomics)
// The sampling grid carries no gradient (it is built under no_grad in the reference, models/utils/warping.py:79), so
// nothing flows to the hypotheses or the cameras.  Positions / weights are recomputed with the forward's arithmetic.
#include <stdlib.h>

#include "warp_common.hpp"

namespace {

__global__ __launch_bounds__(256) void warp_aggregate_bwd_kernel(const float* __restrict__ ref,
                                                                 const float* __restrict__ src,
                                                                 const float* __restrict__ vis, const float* __restrict__ mats_d,
                                                                 const float* __restrict__ hyp,
                                                                 const float* __restrict__ gvol, float* __restrict__ gref,
                                                                 float* __restrict__ gsrc, float* __restrict__ gvis, int V,
                                                                 int C, int D, int h, int w, int hyp_pp, int tiles_x,
                                                                 int ntiles, int nseg, int seg_planes) {
  constexpr int CG = 8;
  // one workgroup per (tile, view, group of 8 channels, depth segment): the coarse cascade stages have few pixels (72 x 96)
  // but 32 channels x 48 planes, a thread per (pixel, view) alone left most of the chip idle behind 192 serial iterations
  const int ngroups = C / CG;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * V * ngroups * nseg);
  const int seg = lin % nseg;
  lin /= nseg;
  const int c0 = (lin % ngroups) * CG;
  lin /= ngroups;
  const int v = lin % V;
  const int tile = lin / V;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  const int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)y * w + x;
  const float* __restrict__ srcv = src + (size_t)v * hw * C;
  float* __restrict__ gsrcv = gsrc + (size_t)v * hw * C;
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = mats_d[v * 12 + i];
  float r[3];
  cds_row_terms(m, (float)x, (float)y, r);
  const float vw = vis[(size_t)v * hw + pix];
  float gv = 0.f;
  const int d_lo = seg * seg_planes, d_hi = min(D, d_lo + seg_planes);
  {
    float rf[CG], gr[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      rf[c] = ref[((size_t)v * C + c0 + c) * hw + pix];
      gr[c] = 0.f;
    }
    // Scatter with run merging: consecutive planes of a pixel mostly fall into the SAME 2x2 texel cell (the sample moves
    // ~0.15 px per plane), so the 4 x 8 tap contributions are summed in registers while the cell stays and flushed with
    // atomics only when it changes: 4-6x fewer global atomics (they were the cost of this kernel: 32 per plane and pixel).
    int cell[4] = {-2, -2, -2, -2};      // offsets of the cell being accumulated (-2: none yet)
    float run[4][CG];
    auto flush = [&]() {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (cell[t] >= 0) {
          float* dst = gsrcv + (size_t)cell[t] * C + c0;
#pragma unroll
          for (int c = 0; c < CG; ++c) atomicAdd(dst + c, run[t][c]);
        }
      }
    };
    for (int d = d_lo; d < d_hi; ++d) {
      const float dv = hyp_pp ? hyp[(size_t)d * hw + pix] : hyp[d];
      const Taps tp = cds_taps(r, m + 9, dv, h, w, half_w, half_h);
      float g[CG], wv[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) g[c] = gvol[((size_t)(c0 + c) * D + d) * hw + pix];
#pragma unroll
      for (int q = 0; q < CG; q += 4) {
        const float4 a = cds_ld4(srcv, tp.off[0], C, c0 + q), b = cds_ld4(srcv, tp.off[1], C, c0 + q);
        const float4 cc = cds_ld4(srcv, tp.off[2], C, c0 + q), e = cds_ld4(srcv, tp.off[3], C, c0 + q);
        wv[q + 0] = cds_interp(a.x, b.x, cc.x, e.x, tp.wt);
        wv[q + 1] = cds_interp(a.y, b.y, cc.y, e.y, tp.wt);
        wv[q + 2] = cds_interp(a.z, b.z, cc.z, e.z, tp.wt);
        wv[q + 3] = cds_interp(a.w, b.w, cc.w, e.w, tp.wt);
      }
      float coef[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        gr[c] = fmaf(g[c], wv[c], gr[c]);
        gv = fmaf(g[c] * rf[c], wv[c], gv);
        coef[c] = g[c] * rf[c] * vw;
      }
      const bool same = tp.off[0] == cell[0] && tp.off[1] == cell[1] && tp.off[2] == cell[2] && tp.off[3] == cell[3];
      if (!same) {
        flush();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          cell[t] = tp.off[t];
#pragma unroll
          for (int c = 0; c < CG; ++c) run[t][c] = 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < CG; ++c) run[t][c] = fmaf(coef[c], tp.wt[t], run[t][c]);
    }
    flush();
    // partial sums of this (channel group, depth segment): gref / gvis are zero-initialised by the caller
#pragma unroll
    for (int c = 0; c < CG; ++c) atomicAdd(&gref[((size_t)v * C + c0 + c) * hw + pix], gr[c] * vw);
  }
  atomicAdd(&gvis[(size_t)v * hw + pix], gv);
}

// The same gradients with the scatter into grad_src privatised in LDS.  The samples of a (tile, view, depth segment) land in a
// bounded box of source texels (the warp is a homography: neighbouring pixels and planes map to neighbouring texels), so the
// workgroup first finds that box (a cheap pass over the sample positions), accumulates the 4 x 8 tap contributions of every
// sample with LDS atomics into [8 channels][box texel], and then adds the box to grad_src with ONE coalesced pass of global
// atomics (box-size many, on consecutive addresses) instead of 32 scattered ones per pixel and cell change.  Boxes that do
// not fit (BWD_BOX texels) fall back to the direct scatter above, decided per workgroup.
//
// The LDS accumulators are 64-bit FIXED POINT, not fp32: ds_add_f32 retires ~0.35 lanes per clock and CU on gfx950 whatever the
// address pattern (85 % of this kernel's time when the box was fp32: profiles/r05_k3_bwd.md), ds_add_u64 is ~30 x faster.  The first
// pass also finds M = max |g ref vis| over the workgroup's samples; a contribution coef x weight (<= M in magnitude, product taken in
// fp64 = exact) is scaled by 2^E with M 2^E < 2^45, rounded to an integer and added: 2^16 contributions cannot overflow 63 bits, the
// rounding is 2^-45 of M per contribution (fp32 atomics: 2^-24 of the running sum), and integer addition is associative, so the box
// is independent of the order the lanes arrive in.  Non-finite M (a NaN / inf in the gradient) takes the fp32 direct scatter, which
// propagates it.
#ifndef CDS_K3BWD_BOX
#define CDS_K3BWD_BOX 1024   // texels: (1024 + 1) x 8 channels x 8 B = 65 600 B of LDS (+ 20 B of limits) -> two workgroups per gfx950 CU (160 KB);
                             // over the 64 KB of gfx942 / gfx90a on purpose - this library is gfx950-only (static_assert below).  Build with a tiny value to force the fallback path
#endif
constexpr int BWD_BOX = CDS_K3BWD_BOX;

struct TapsXY {
  int x0, y0;          // top-left texel of the 2 x 2 cell (may be -1: left / top neighbour outside)
  bool ok[4];          // nw, ne, sw, se inside the map
  float wt[4];
  bool any;
};

// cds_taps with the cell coordinates kept (same arithmetic, same operation order)
__device__ __forceinline__ TapsXY cds_taps_xy(const float r[3], const float* __restrict__ t, float d, int h, int w, float half_w,
                                              float half_h) {
  float px = r[0] * d + t[0];
  float py = r[1] * d + t[1];
  float pz = r[2] * d + t[2];
  float z = pz + 1e-6f;
  float u = px / z;
  float v = py / z;
  float gx = u / half_w - 1.0f;
  float gy = v / half_h - 1.0f;
  float ix = (gx + 1.0f) * half_w;
  float iy = (gy + 1.0f) * half_h;
  float x0f = floorf(ix), y0f = floorf(iy);
  float wx = ix - x0f, ex = 1.0f - wx;
  float ny = iy - y0f, sy = 1.0f - ny;
  TapsXY tp;
  tp.wt[0] = sy * ex;
  tp.wt[1] = sy * wx;
  tp.wt[2] = ny * ex;
  tp.wt[3] = ny * wx;
  const bool x0ok = (x0f >= 0.0f) && (x0f <= (float)(w - 1));
  const bool x1ok = (x0f >= -1.0f) && (x0f <= (float)(w - 2));
  const bool y0ok = (y0f >= 0.0f) && (y0f <= (float)(h - 1));
  const bool y1ok = (y0f >= -1.0f) && (y0f <= (float)(h - 2));
  tp.x0 = (x0ok || x1ok) ? (int)x0f : 0;
  tp.y0 = (y0ok || y1ok) ? (int)y0f : 0;
  tp.ok[0] = x0ok && y0ok;
  tp.ok[1] = x1ok && y0ok;
  tp.ok[2] = x0ok && y1ok;
  tp.ok[3] = x1ok && y1ok;
  tp.any = tp.ok[0] || tp.ok[1] || tp.ok[2] || tp.ok[3];
  return tp;
}

__global__ __launch_bounds__(256) void warp_aggregate_bwd_box_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                                                                     const float* __restrict__ vis, const float* __restrict__ mats_d,
                                                                     const float* __restrict__ hyp,
                                                                     const float* __restrict__ gvol, float* __restrict__ gref,
                                                                     float* __restrict__ gsrc, float* __restrict__ gvis, int V,
                                                                     int C, int D, int h, int w, int hyp_pp, int tiles_x,
                                                                     int ntiles, int nseg, int seg_planes) {
  constexpr int CG = 8;
  constexpr int BOXP = BWD_BOX + 1;                         // channel-planar [c][texel]: the lanes of one ds_add (neighbouring pixels) hit neighbouring words
  __shared__ long long box[BOXP * CG];
  static_assert((size_t)BOXP * CG * 8 + 32 <= 80 * 1024, "K3 backward box: two workgroups must fit the 160 KB LDS of a gfx950 CU");
  __shared__ int lim[4];                                   // xmin, ymin, xmax, ymax of the touched texels
  __shared__ unsigned mbits;                               // bits of M = max |g ref vis| (non-negative floats order like their bit patterns; NaN on top)
  const int ngroups = C / CG;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * V * ngroups * nseg);
  const int seg = lin % nseg;
  lin /= nseg;
  const int c0 = (lin % ngroups) * CG;
  lin /= ngroups;
  const int v = lin % V;
  const int tile = lin / V;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  const int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  const bool live = x < w && y < h;
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = live ? (size_t)y * w + x : 0;
  const float* __restrict__ srcv = src + (size_t)v * hw * C;
  float* __restrict__ gsrcv = gsrc + (size_t)v * hw * C;
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = mats_d[v * 12 + i];
  float r[3];
  cds_row_terms(m, (float)x, (float)y, r);
  const int d_lo = seg * seg_planes, d_hi = min(D, d_lo + seg_planes);
  const float vw = live ? vis[(size_t)v * hw + pix] : 0.f;
  float rf[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) rf[c] = live ? ref[((size_t)v * C + c0 + c) * hw + pix] : 0.f;

  // ---- pass 1: the box and the magnitude bound ----
  if (threadIdx.x == 0) {
    lim[0] = 1 << 30; lim[1] = 1 << 30; lim[2] = -(1 << 30); lim[3] = -(1 << 30);
    mbits = 0u;
  }
  __syncthreads();
  {
    int xmin = 1 << 30, ymin = 1 << 30, xmax = -(1 << 30), ymax = -(1 << 30);
    float mx = 0.f;
    if (live)
      for (int d = d_lo; d < d_hi; ++d) {
        const float dv = hyp_pp ? hyp[(size_t)d * hw + pix] : hyp[d];
        const TapsXY tp = cds_taps_xy(r, m + 9, dv, h, w, half_w, half_h);
        if (tp.any) {
          xmin = min(xmin, max(tp.x0, 0)); ymin = min(ymin, max(tp.y0, 0));
          xmax = max(xmax, min(tp.x0 + 1, w - 1)); ymax = max(ymax, min(tp.y0 + 1, h - 1));
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const float a = fabsf(gvol[((size_t)(c0 + c) * D + d) * hw + pix] * rf[c] * vw);
            mx = (a > mx || a != a) ? a : mx;               // a NaN sticks
          }
        }
      }
    unsigned mb = __float_as_uint(mx) & 0x7fffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      xmin = min(xmin, __shfl_xor(xmin, o)); ymin = min(ymin, __shfl_xor(ymin, o));
      xmax = max(xmax, __shfl_xor(xmax, o)); ymax = max(ymax, __shfl_xor(ymax, o));
      mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&lim[0], xmin); atomicMin(&lim[1], ymin); atomicMax(&lim[2], xmax); atomicMax(&lim[3], ymax);
      atomicMax(&mbits, mb);
    }
  }
  __syncthreads();
  const int bx0 = lim[0], by0 = lim[1];
  const int bw = lim[2] - lim[0] + 1, bh = lim[3] - lim[1] + 1;
  const bool empty = bw <= 0 || bh <= 0;
  const int ef = (int)(mbits >> 23);                        // M < 2^(ef - 126)
  const bool boxed = !empty && bw * bh <= BWD_BOX && ef != 255;   // (workgroup-uniform)
  const bool scatter = mbits != 0u;                         // M == 0: every contribution to grad_src is zero
  // contributions per box cell <= 256 pixels x planes of the segment (one tap of a sample per cell): 2^16 up to 256 planes
  int extra = 0;
  while ((256 << extra) < d_hi - d_lo) ++extra;
  const int E = 45 - extra - (ef - 126);
  const double scale = __longlong_as_double((long long)(E + 1023) << 52), inv_scale = __longlong_as_double((long long)(1023 - E) << 52);
  if (boxed) {
    for (int i = threadIdx.x; i < bw * bh * CG; i += 256) box[(i % CG) * BOXP + i / CG] = 0;
    __syncthreads();
  }

  // ---- pass 2 ----
  constexpr double MAGIC = 6755399441055744.0;              // 1.5 x 2^52: (x + MAGIC) holds round(x) in its low mantissa bits, |x| < 2^51
  float gv = 0.f;
  float gr[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) gr[c] = 0.f;
  if (live && !empty) {
    for (int d = d_lo; d < d_hi; ++d) {
      const float dv = hyp_pp ? hyp[(size_t)d * hw + pix] : hyp[d];
      const TapsXY tp = cds_taps_xy(r, m + 9, dv, h, w, half_w, half_h);
      if (!tp.any) continue;                                // all four taps outside: the sample and its gradients are zero
      const int base = tp.y0 * w + tp.x0;
      const int off[4] = {tp.ok[0] ? base : -1, tp.ok[1] ? base + 1 : -1, tp.ok[2] ? base + w : -1, tp.ok[3] ? base + w + 1 : -1};
      float g[CG], wv[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) g[c] = gvol[((size_t)(c0 + c) * D + d) * hw + pix];
#pragma unroll
      for (int q = 0; q < CG; q += 4) {
        const float4 a = cds_ld4(srcv, off[0], C, c0 + q), b = cds_ld4(srcv, off[1], C, c0 + q);
        const float4 cc = cds_ld4(srcv, off[2], C, c0 + q), e = cds_ld4(srcv, off[3], C, c0 + q);
        wv[q + 0] = cds_interp(a.x, b.x, cc.x, e.x, tp.wt);
        wv[q + 1] = cds_interp(a.y, b.y, cc.y, e.y, tp.wt);
        wv[q + 2] = cds_interp(a.z, b.z, cc.z, e.z, tp.wt);
        wv[q + 3] = cds_interp(a.w, b.w, cc.w, e.w, tp.wt);
      }
      float coef[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        gr[c] = fmaf(g[c], wv[c], gr[c]);
        gv = fmaf(g[c] * rf[c], wv[c], gv);
        coef[c] = g[c] * rf[c] * vw;
      }
      if (!scatter) continue;
      if (boxed) {
        double cd[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) cd[c] = (double)coef[c];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (off[t] < 0) continue;
          const double wd = (double)tp.wt[t] * scale;
          unsigned long long* dst =
              reinterpret_cast<unsigned long long*>(box) + ((tp.y0 + (t >> 1) - by0) * bw + (tp.x0 + (t & 1) - bx0));
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const long long q = __double_as_longlong(fma(cd[c], wd, MAGIC)) - __double_as_longlong(MAGIC);
            atomicAdd(dst + c * BOXP, (unsigned long long)q);                            // ds_add_u64
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (off[t] < 0) continue;
          float* dst = gsrcv + (size_t)off[t] * C + c0;
#pragma unroll
          for (int c = 0; c < CG; ++c) atomicAdd(dst + c, coef[c] * tp.wt[t]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CG; ++c) atomicAdd(&gref[((size_t)v * C + c0 + c) * hw + pix], gr[c] * vw);
    atomicAdd(&gvis[(size_t)v * hw + pix], gv);
  }
  if (boxed && scatter) {
    __syncthreads();
    for (int i = threadIdx.x; i < bw * bh * CG; i += 256) {
      const int t = i / CG, c = i - t * CG;
      const long long q = box[c * BOXP + t];
      if (q != 0) {
        const int by = t / bw, bxx = t - by * bw;
        atomicAdd(&gsrcv[((size_t)(by0 + by) * w + bx0 + bxx) * C + c0 + c], (float)((double)q * inv_scale));
      }
    }
  }
}

}  // namespace

extern "C" int cds_warp_aggregate_bwd_f32(const float* ref_chw, const float* src_hwc, const float* vis_w,
                                          const float* mats, const float* hyp, const float* grad_volume,
                                          float* grad_ref, float* grad_src_hwc, float* grad_vis, int V, int C, int D, int h,
                                          int w, int hyp_per_pixel, void* stream) {
  if (!ref_chw || !src_hwc || !vis_w || !mats || !hyp || !grad_volume || !grad_ref || !grad_src_hwc || !grad_vis ||
      V < 1 || V > CDS_MAX_VIEWS || (C != 8 && C != 16 && C != 32) || D < 1 || h < 1 || w < 1)
    return CDS_EINVAL;
  const float* wm = mats;
  const int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  const int ntiles = tiles_x * tiles_y;
  const int base = ntiles * V * (C / 8);
  const bool direct = cds_env_set("CDS_K3BWD_DIRECT");   // A/B knob: the direct-scatter kernel
  if (!direct) {
    // LDS-privatised scatter: segments only until ~1024 workgroups (every segment pays one box flush), >= 8 planes each
    int nseg = 1;
    while (base * nseg < 1024 && D / (2 * nseg) >= 8) nseg *= 2;
    const int seg_planes = cds_ceil_div(D, nseg);
    nseg = cds_ceil_div(D, seg_planes);
    hipLaunchKernelGGL(warp_aggregate_bwd_box_kernel, dim3(base * nseg), dim3(256), 0, (hipStream_t)stream, ref_chw, src_hwc, vis_w,
                       wm, hyp, grad_volume, grad_ref, grad_src_hwc, grad_vis, V, C, D, h, w, hyp_per_pixel, tiles_x, ntiles, nseg,
                       seg_planes);
    return cds_launch_status();
  }
  // depth segments: aim at >= 4096 workgroups, at least 4 planes per segment
  int nseg = 1;
  while (base * nseg < 4096 && D / (2 * nseg) >= 4) nseg *= 2;
  const int seg_planes = cds_ceil_div(D, nseg);
  nseg = cds_ceil_div(D, seg_planes);
  hipLaunchKernelGGL(warp_aggregate_bwd_kernel, dim3(base * nseg), dim3(256), 0, (hipStream_t)stream, ref_chw, src_hwc,
                     vis_w, wm, hyp, grad_volume, grad_ref, grad_src_hwc, grad_vis, V, C, D, h, w, hyp_per_pixel, tiles_x,
                     ntiles, nseg, seg_planes);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The training step's epilogue of K3 (models/model.py:56-78): cost volume = sum / (sum_v vis + 1e-6), feature distance
// fd[d] = (sum_c volume_sum[c][d]) / (sum_v vis + 1e-6) and, with the ground-truth plane, fd[D] from gt_sum; one launch forward,
// two backward (the reduction over planes of d(denominator) goes through per-plane partial sums: no atomics, fixed order).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float vis_denominator(const float* __restrict__ vis, int V, size_t hw, size_t p) {
  float s = vis[p];
  for (int v = 1; v < V; ++v) s += vis[(size_t)v * hw + p];
  return s + 1e-6f;
}

__global__ __launch_bounds__(256) void volume_finish_kernel(const float* __restrict__ vsum, const float* __restrict__ gtsum,
                                                            const float* __restrict__ vis, int V, int C, int D, int hw,
                                                            float* __restrict__ vol, float* __restrict__ fd) {
  const int p = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
  if (p >= hw) return;
  const float den = vis_denominator(vis, V, hw, p);
  float s = 0.f;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const size_t i = ((size_t)c * D + d) * hw + p;
    const float x = vsum[i];
    vol[i] = x / den;
    s = c ? s + x : x;
  }
  fd[(size_t)d * hw + p] = s / den;
  if (gtsum && d == 0) {
    float g = gtsum[p];
    for (int c = 1; c < C; ++c) g += gtsum[(size_t)c * hw + p];
    fd[(size_t)D * hw + p] = g / den;
  }
}

__global__ __launch_bounds__(256) void volume_finish_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ gfd,
                                                                const float* __restrict__ vsum, const float* __restrict__ vis, int V,
                                                                int C, int D, int hw, float* __restrict__ gvsum,
                                                                float* __restrict__ part) {
  const int p = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
  if (p >= hw) return;
  const float den = vis_denominator(vis, V, hw, p);
  const float gf = gfd ? gfd[(size_t)d * hw + p] : 0.f;
  float acc = 0.f;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const size_t i = ((size_t)c * D + d) * hw + p;
    const float g = (gvol ? gvol[i] : 0.f) + gf;
    gvsum[i] = g / den;
    acc = fmaf(g, vsum[i], acc);
  }
  part[(size_t)d * hw + p] = acc;
}

__global__ __launch_bounds__(256) void volume_finish_bwd_vis_kernel(const float* __restrict__ part, const float* __restrict__ gfd,
                                                                    const float* __restrict__ gtsum, const float* __restrict__ vis, int V,
                                                                    int C, int D, int hw, float* __restrict__ ggt,
                                                                    float* __restrict__ gvis) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const float den = vis_denominator(vis, V, hw, p);
  float tot = 0.f;
#pragma unroll 4
  for (int d = 0; d < D; ++d) tot += part[(size_t)d * hw + p];
  if (gtsum) {
    const float gf = gfd ? gfd[(size_t)D * hw + p] : 0.f;
    for (int c = 0; c < C; ++c) {
      ggt[(size_t)c * hw + p] = gf / den;
      tot = fmaf(gf, gtsum[(size_t)c * hw + p], tot);
    }
  }
  const float gden = -tot / (den * den);
  for (int v = 0; v < V; ++v) gvis[(size_t)v * hw + p] = gden;
}

}  // namespace

// volume_sum [C][D][hw] (un-normalised K3), gt_sum [C][hw] (K3 at the ground-truth depth) or NULL, vis [V][hw] ->
// volume [C][D][hw], feat_distance [D (+ 1 with gt_sum)][hw].
extern "C" int cds_volume_finish_f32(const float* volume_sum, const float* gt_sum, const float* vis, int V, int C, int D, int hw,
                                     float* volume, float* feat_distance, void* stream) {
  if (!volume_sum || !vis || !volume || !feat_distance || V < 1 || C < 1 || D < 1 || D > 65535 || hw < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(volume_finish_kernel, dim3(cds_ceil_div(hw, 256), D), dim3(256), 0, (hipStream_t)stream, volume_sum, gt_sum, vis, V,
                     C, D, hw, volume, feat_distance);
  return cds_launch_status();
}

// Its backward.  g_volume [C][D][hw] / g_feat_distance [D (+ 1)][hw] (either may be NULL: no gradient) -> g_volume_sum [C][D][hw],
// g_gt_sum [C][hw] (with gt_sum), g_vis [V][hw]; scratch: D x hw floats.
extern "C" int cds_volume_finish_bwd_f32(const float* g_volume, const float* g_feat_distance, const float* volume_sum, const float* gt_sum,
                                         const float* vis, int V, int C, int D, int hw, float* g_volume_sum, float* g_gt_sum, float* g_vis,
                                         float* scratch, void* stream) {
  if (!volume_sum || !vis || !g_volume_sum || !g_vis || !scratch || (gt_sum != nullptr) != (g_gt_sum != nullptr) || V < 1 || C < 1 ||
      D < 1 || D > 65535 || hw < 1)
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(volume_finish_bwd_kernel, dim3(cds_ceil_div(hw, 256), D), dim3(256), 0, st, g_volume, g_feat_distance, volume_sum,
                     vis, V, C, D, hw, g_volume_sum, scratch);
  hipLaunchKernelGGL(volume_finish_bwd_vis_kernel, dim3(cds_ceil_div(hw, 256)), dim3(256), 0, st, scratch, g_feat_distance, gt_sum, vis, V,
                     C, D, hw, g_gt_sum, g_vis);
  return cds_launch_status();
}
