// Backward of the (un-normalised) warp-aggregate, for the training step.
//   forward:  volume[c][d][p] = sum_v vis_v[p] * ref_v[c][p] * warp_v[c][d][p],  warp = bilinear sample of src_v
//   backward: grad_ref_v[c][p]   = vis_v[p] * sum_d g[c][d][p] * warp_v[c][d][p]
//             grad_vis_v[p]      = sum_{c,d} g[c][d][p] * ref_v[c][p] * warp_v[c][d][p]
//             grad_src_v[tap][c] += g[c][d][p] * ref_v[c][p] * vis_v[p] * w_tap      (scatter-add, atomics)
// The sampling grid carries no gradient (it is built under no_grad in the reference, models/utils/warping.py:79), so
// nothing flows to the hypotheses or the cameras.  Positions / weights are recomputed with the forward's arithmetic.
#include "warp_common.hpp"

namespace {

__global__ __launch_bounds__(256) void warp_aggregate_bwd_kernel(const float* __restrict__ ref,
                                                                 const float* __restrict__ src,
                                                                 const float* __restrict__ vis, WarpMats mats,
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
  for (int i = 0; i < 12; ++i) m[i] = mats.m[v][i];
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

}  // namespace

extern "C" int cds_warp_aggregate_bwd_f32(const float* ref_chw, const float* src_hwc, const float* vis_w,
                                          const float* mats_host, const float* hyp, const float* grad_volume,
                                          float* grad_ref, float* grad_src_hwc, float* grad_vis, int V, int C, int D, int h,
                                          int w, int hyp_per_pixel, void* stream) {
  if (!ref_chw || !src_hwc || !vis_w || !mats_host || !hyp || !grad_volume || !grad_ref || !grad_src_hwc || !grad_vis ||
      V < 1 || V > CDS_MAX_VIEWS || (C != 8 && C != 16 && C != 32) || D < 1 || h < 1 || w < 1)
    return CDS_EINVAL;
  WarpMats wm;
  for (int v = 0; v < CDS_MAX_VIEWS; ++v)
    for (int i = 0; i < 12; ++i) wm.m[v][i] = v < V ? mats_host[v * 12 + i] : 0.f;
  const int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  const int ntiles = tiles_x * tiles_y;
  // depth segments: aim at >= 4096 workgroups, at least 4 planes per segment
  const int base = ntiles * V * (C / 8);
  int nseg = 1;
  while (base * nseg < 4096 && D / (2 * nseg) >= 4) nseg *= 2;
  const int seg_planes = cds_ceil_div(D, nseg);
  nseg = cds_ceil_div(D, seg_planes);
  hipLaunchKernelGGL(warp_aggregate_bwd_kernel, dim3(base * nseg), dim3(256), 0, (hipStream_t)stream, ref_chw, src_hwc,
                     vis_w, wm, hyp, grad_volume, grad_ref, grad_src_hwc, grad_vis, V, C, D, h, w, hyp_per_pixel, tiles_x,
                     ntiles, nseg, seg_planes);
  return cds_launch_status();
}
