// LDS-staged variants of K1 (warp-correlate-entropy) and K3 (warp-aggregate) for C = 8.
//
// The direct kernels in warp.hip gather 4 taps x 32 B per (voxel, view) through the vector L1 and are
// bound by its address/tag rate (measured 1.4-1.8 ms at M1 against 0.3-0.5 ms of HBM time).  Here a
// workgroup owns a 64x4 tile of reference pixels; for a chunk of DC consecutive depth planes the source
// footprint of the tile is a thin parallelogram along the epipolar line (~0.15 px per plane at M1), so its
// bounding box (tile + halo, ~70x7 texels = 16 KB per view) is staged once into LDS with coalesced 16-byte
// loads and reused by every plane of the chunk and every pixel of the tile; the taps are then ds_read_b128s.
//
// LDS image of a box: two planes [BH][BW] of float4 (channels 0-3 and 4-7) so that consecutive lanes, which
// sample (nearly) consecutive texels, read consecutive 16-byte slots (conflict-free ds_read_b128).
// Robustness: the box is the min/max over every lane's own first/last plane of the chunk; any tap that still
// falls outside it (non-monotone hypotheses, projective pole) and any box larger than the LDS budget take the
// original global-memory path per lane / per block, so results never depend on the geometry assumptions.
// Arithmetic and operation order are identical to warp.hip (same cds_taps / cds_interp).
#include "warp_common.hpp"

namespace {

constexpr int C8 = 8;
constexpr int BOX_CAP = 504;   // texels per view box (15.75 KB; 4 views + scratch stay below 64 KB)
constexpr int DC = 32;         // depth planes per staged chunk

struct Box {
  int x0, y0, bw, bh;  // origin, width, height in texels (block-uniform)
  bool staged;
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// integer cell (floor of the sample position) clamped to [-2, n+1]; NaN -> -2 (never in the image)
__device__ __forceinline__ void cell_of(const float r[3], const float* __restrict__ t, float d, int h, int w,
                                        float half_w, float half_h, int& cx, int& cy) {
  float px = r[0] * d + t[0];
  float py = r[1] * d + t[1];
  float pz = r[2] * d + t[2];
  float z = pz + 1e-6f;
  float ix = ((px / z) / half_w - 1.0f + 1.0f) * half_w;
  float iy = ((py / z) / half_h - 1.0f + 1.0f) * half_h;
  float fx = floorf(ix), fy = floorf(iy);
  fx = (fx >= -2.0f) ? fx : -2.0f;  // also catches NaN
  fy = (fy >= -2.0f) ? fy : -2.0f;
  fx = fminf(fx, (float)(w + 1));
  fy = fminf(fy, (float)(h + 1));
  cx = (int)fx;
  cy = (int)fy;
}

// Block-wide bounding boxes for NV views.  lo/hi: this thread's cells at the chunk's first and last plane.
// red: LDS scratch int[4 waves][NV][4].
template <int NV>
__device__ __forceinline__ void reduce_boxes(const int cx0[NV], const int cy0[NV], const int cx1[NV], const int cy1[NV],
                                             bool active, int nv, int h, int w, int* red, Box box[NV]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int xmin = active ? min(cx0[v], cx1[v]) : 0x7fffffff;
    int xmax = active ? max(cx0[v], cx1[v]) : -0x7fffffff;
    int ymin = active ? min(cy0[v], cy1[v]) : 0x7fffffff;
    int ymax = active ? max(cy0[v], cy1[v]) : -0x7fffffff;
    xmin = wave_min(xmin);
    xmax = wave_max(xmax);
    ymin = wave_min(ymin);
    ymax = wave_max(ymax);
    if (lane == 0) {
      int* p = red + (wave * NV + v) * 4;
      p[0] = xmin;
      p[1] = xmax;
      p[2] = ymin;
      p[3] = ymax;
    }
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int xmin = 0x7fffffff, xmax = -0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const int* p = red + (wv * NV + v) * 4;
      xmin = min(xmin, p[0]);
      xmax = max(xmax, p[1]);
      ymin = min(ymin, p[2]);
      ymax = max(ymax, p[3]);
    }
    // taps touch cells [xmin, xmax+1] x [ymin, ymax+1]; clip to the image
    int bx0 = max(xmin, 0), bx1 = min(xmax + 1, w - 1);
    int by0 = max(ymin, 0), by1 = min(ymax + 1, h - 1);
    int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    bool ok = (v < nv) && bw > 0 && bh > 0 && bw * bh <= BOX_CAP;
    box[v].x0 = __builtin_amdgcn_readfirstlane(bx0);
    box[v].y0 = __builtin_amdgcn_readfirstlane(by0);
    box[v].bw = __builtin_amdgcn_readfirstlane(ok ? bw : 0);
    box[v].bh = __builtin_amdgcn_readfirstlane(ok ? bh : 0);
    box[v].staged = __builtin_amdgcn_readfirstlane((int)ok) != 0;
  }
}

// Cooperative copy of one box into its two LDS planes.  Each wave takes rows wave, wave+4, ...
__device__ __forceinline__ void stage_box(const float* __restrict__ srcv, int w, const Box& b, float4* __restrict__ dst) {
  if (!b.staged) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n4 = 2 * b.bw;  // float4 per row
  for (int row = wave; row < b.bh; row += 4) {
    const float4* __restrict__ g = reinterpret_cast<const float4*>(srcv + ((size_t)(b.y0 + row) * w + b.x0) * C8);
    float4* lo = dst + row * b.bw;
    float4* hi = dst + BOX_CAP + row * b.bw;
    for (int i = lane; i < n4; i += 64) {
      const float4 v = g[i];
      ((i & 1) ? hi : lo)[i >> 1] = v;
    }
  }
}

struct Tex {
  float4 lo, hi;
};

// Taps with box-relative coordinates: same arithmetic as cds_taps, but keeps (x0,y0) instead of a linear offset.
struct Taps2 {
  int x0, y0;
  bool ok[4];
  float wt[4];
};
__device__ __forceinline__ Taps2 taps2(const float r[3], const float* __restrict__ t, float d, int h, int w, float half_w,
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
  Taps2 tp;
  tp.wt[0] = sy * ex;
  tp.wt[1] = sy * wx;
  tp.wt[2] = ny * ex;
  tp.wt[3] = ny * wx;
  bool x0ok = (x0f >= 0.0f) && (x0f <= (float)(w - 1));
  bool x1ok = (x0f >= -1.0f) && (x0f <= (float)(w - 2));
  bool y0ok = (y0f >= 0.0f) && (y0f <= (float)(h - 1));
  bool y1ok = (y0f >= -1.0f) && (y0f <= (float)(h - 2));
  tp.x0 = (x0ok || x1ok) ? (int)x0f : 0;
  tp.y0 = (y0ok || y1ok) ? (int)y0f : 0;
  tp.ok[0] = x0ok && y0ok;
  tp.ok[1] = x1ok && y0ok;
  tp.ok[2] = x0ok && y1ok;
  tp.ok[3] = x1ok && y1ok;
  return tp;
}

__device__ __forceinline__ Tex fetch2(int x, int y, bool ok, int w, const Box& b, const float4* __restrict__ lds,
                                      const float* __restrict__ srcv) {
  Tex t;
  t.lo = make_float4(0.f, 0.f, 0.f, 0.f);
  t.hi = t.lo;
  if (ok) {
    const int bx = x - b.x0, by = y - b.y0;
    if ((unsigned)bx < (unsigned)b.bw && (unsigned)by < (unsigned)b.bh) {  // bw = bh = 0 when the box is not staged
      const int ti = by * b.bw + bx;
      t.lo = lds[ti];
      t.hi = lds[BOX_CAP + ti];
    } else {
      const float4* g = reinterpret_cast<const float4*>(srcv + ((size_t)y * w + x) * C8);
      t.lo = g[0];
      t.hi = g[1];
    }
  }
  return t;
}

// ---------------------------------------------------------------------------------------------
// K3 with LDS-staged boxes, C = 8, V <= 4 (all views of a chunk resident: 4 x 16 KB)
// ---------------------------------------------------------------------------------------------
template <int VMAX>
__global__ __launch_bounds__(256) void warp_aggregate_lds_kernel(
    const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ vis, WarpMats mats,
    const float* __restrict__ hyp, float* __restrict__ volume, const float* __restrict__ vis_sum, int V, int D, int h,
    int w, int flags, int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];  // VMAX * 2*BOX_CAP float4, then int red[4*VMAX*4]
  int* red = reinterpret_cast<int*>(lds4 + VMAX * 2 * BOX_CAP);

  const int tile = cds_xcd_remap(blockIdx.x, ntiles);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  const int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  const bool active = x < w && y < h;
  const int xc = min(x, w - 1), yc = min(y, h - 1);  // inactive lanes shadow a valid pixel, never store
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)yc * w + xc;

  float rf[VMAX][C8];
  float vw[VMAX];
  float r[VMAX][3];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    if (v < V) {
      vw[v] = vis[(size_t)v * hw + pix];
#pragma unroll
      for (int c = 0; c < C8; ++c) rf[v][c] = ref[((size_t)v * C8 + c) * hw + pix];
      cds_row_terms(mats.m[v], (float)xc, (float)yc, r[v]);
    }
  }
  const bool accumulate = flags & CDS_AGG_ACCUMULATE;
  const bool normalize = flags & CDS_AGG_NORMALIZE;
  const float denom = (normalize ? vis_sum[pix] : 1.0f) + 1e-6f;

  for (int d0 = 0; d0 < D; d0 += DC) {
    const int d1 = min(D, d0 + DC);
    // ---- boxes of this chunk ----
    int cx0[VMAX], cy0[VMAX], cx1[VMAX], cy1[VMAX];
    const float dfirst = hyp[(size_t)d0 * hw + pix], dlast = hyp[(size_t)(d1 - 1) * hw + pix];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      cx0[v] = cy0[v] = cx1[v] = cy1[v] = 0;
      if (v < V) {
        cell_of(r[v], mats.m[v] + 9, dfirst, h, w, half_w, half_h, cx0[v], cy0[v]);
        cell_of(r[v], mats.m[v] + 9, dlast, h, w, half_w, half_h, cx1[v], cy1[v]);
      }
    }
    Box box[VMAX];
    __syncthreads();  // previous chunk's LDS reads are done (also protects `red`)
    reduce_boxes<VMAX>(cx0, cy0, cx1, cy1, active, V, h, w, red, box);
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < V) stage_box(src + (size_t)v * hw * C8, w, box[v], lds4 + v * 2 * BOX_CAP);
    __syncthreads();

    // ---- planes of the chunk ----
    float dnext = hyp[(size_t)d0 * hw + pix];
    for (int d = d0; d < d1; ++d) {
      const float dv = dnext;
      if (d + 1 < d1) dnext = hyp[(size_t)(d + 1) * hw + pix];
      float acc[C8];
#pragma unroll
      for (int c = 0; c < C8; ++c) acc[c] = accumulate ? volume[((size_t)c * D + d) * hw + pix] : 0.f;
#pragma unroll
      for (int v = 0; v < VMAX; ++v) {
        if (v < V) {
          const float* __restrict__ srcv = src + (size_t)v * hw * C8;
          const float4* lv = lds4 + v * 2 * BOX_CAP;
          const Taps2 tp = taps2(r[v], mats.m[v] + 9, dv, h, w, half_w, half_h);
          const Tex a = fetch2(tp.x0, tp.y0, tp.ok[0], w, box[v], lv, srcv);
          const Tex b = fetch2(tp.x0 + 1, tp.y0, tp.ok[1], w, box[v], lv, srcv);
          const Tex c = fetch2(tp.x0, tp.y0 + 1, tp.ok[2], w, box[v], lv, srcv);
          const Tex e = fetch2(tp.x0 + 1, tp.y0 + 1, tp.ok[3], w, box[v], lv, srcv);
          const float w0 = cds_interp(a.lo.x, b.lo.x, c.lo.x, e.lo.x, tp.wt);
          const float w1 = cds_interp(a.lo.y, b.lo.y, c.lo.y, e.lo.y, tp.wt);
          const float w2 = cds_interp(a.lo.z, b.lo.z, c.lo.z, e.lo.z, tp.wt);
          const float w3 = cds_interp(a.lo.w, b.lo.w, c.lo.w, e.lo.w, tp.wt);
          const float w4 = cds_interp(a.hi.x, b.hi.x, c.hi.x, e.hi.x, tp.wt);
          const float w5 = cds_interp(a.hi.y, b.hi.y, c.hi.y, e.hi.y, tp.wt);
          const float w6 = cds_interp(a.hi.z, b.hi.z, c.hi.z, e.hi.z, tp.wt);
          const float w7 = cds_interp(a.hi.w, b.hi.w, c.hi.w, e.hi.w, tp.wt);
          acc[0] = acc[0] + (rf[v][0] * w0) * vw[v];
          acc[1] = acc[1] + (rf[v][1] * w1) * vw[v];
          acc[2] = acc[2] + (rf[v][2] * w2) * vw[v];
          acc[3] = acc[3] + (rf[v][3] * w3) * vw[v];
          acc[4] = acc[4] + (rf[v][4] * w4) * vw[v];
          acc[5] = acc[5] + (rf[v][5] * w5) * vw[v];
          acc[6] = acc[6] + (rf[v][6] * w6) * vw[v];
          acc[7] = acc[7] + (rf[v][7] * w7) * vw[v];
        }
      }
      if (active) {
#pragma unroll
        for (int c = 0; c < C8; ++c) {
          const float o = normalize ? acc[c] / denom : acc[c];
          __builtin_nontemporal_store(o, &volume[((size_t)c * D + d) * hw + pix]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 with an LDS-staged box, C = 8, one (tile, view) per workgroup
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_entropy_lds_kernel(const float* __restrict__ ref,
                                                               const float* __restrict__ src, WarpMats mats,
                                                               const float* __restrict__ hyp,
                                                               float* __restrict__ entropy, int V, int D, int h, int w,
                                                               int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  int* red = reinterpret_cast<int*>(lds4 + 2 * BOX_CAP);
  const int lin = cds_xcd_remap(blockIdx.x, ntiles * V);
  const int v = lin % V;
  const int tile = lin / V;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  const int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  const bool active = x < w && y < h;
  const int xc = min(x, w - 1), yc = min(y, h - 1);
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)yc * w + xc;
  const float* __restrict__ srcv = src + (size_t)v * hw * C8;
  // the matrix of this block's view, copied out of the kernarg struct with a block-uniform index
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = mats.m[v][i];
  float rf[C8];
#pragma unroll
  for (int c = 0; c < C8; ++c) rf[c] = ref[((size_t)v * C8 + c) * hw + pix];
  float r[3];
  cds_row_terms(m, (float)xc, (float)yc, r);
  float mx = -INFINITY, Z = 0.f, T = 0.f;
  for (int d0 = 0; d0 < D; d0 += DC) {
    const int d1 = min(D, d0 + DC);
    int cx0[1], cy0[1], cx1[1], cy1[1];
    cell_of(r, m + 9, hyp[(size_t)d0 * hw + pix], h, w, half_w, half_h, cx0[0], cy0[0]);
    cell_of(r, m + 9, hyp[(size_t)(d1 - 1) * hw + pix], h, w, half_w, half_h, cx1[0], cy1[0]);
    Box box[1];
    __syncthreads();
    reduce_boxes<1>(cx0, cy0, cx1, cy1, active, 1, h, w, red, box);
    stage_box(srcv, w, box[0], lds4);
    __syncthreads();
    float dnext = hyp[(size_t)d0 * hw + pix];
    for (int d = d0; d < d1; ++d) {
      const float dv = dnext;
      if (d + 1 < d1) dnext = hyp[(size_t)(d + 1) * hw + pix];
      const Taps2 tp = taps2(r, m + 9, dv, h, w, half_w, half_h);
      const Tex a = fetch2(tp.x0, tp.y0, tp.ok[0], w, box[0], lds4, srcv);
      const Tex b = fetch2(tp.x0 + 1, tp.y0, tp.ok[1], w, box[0], lds4, srcv);
      const Tex c = fetch2(tp.x0, tp.y0 + 1, tp.ok[2], w, box[0], lds4, srcv);
      const Tex e = fetch2(tp.x0 + 1, tp.y0 + 1, tp.ok[3], w, box[0], lds4, srcv);
      float s = 0.f;
      s = s + rf[0] * cds_interp(a.lo.x, b.lo.x, c.lo.x, e.lo.x, tp.wt);
      s = s + rf[1] * cds_interp(a.lo.y, b.lo.y, c.lo.y, e.lo.y, tp.wt);
      s = s + rf[2] * cds_interp(a.lo.z, b.lo.z, c.lo.z, e.lo.z, tp.wt);
      s = s + rf[3] * cds_interp(a.lo.w, b.lo.w, c.lo.w, e.lo.w, tp.wt);
      s = s + rf[4] * cds_interp(a.hi.x, b.hi.x, c.hi.x, e.hi.x, tp.wt);
      s = s + rf[5] * cds_interp(a.hi.y, b.hi.y, c.hi.y, e.hi.y, tp.wt);
      s = s + rf[6] * cds_interp(a.hi.z, b.hi.z, c.hi.z, e.hi.z, tp.wt);
      s = s + rf[7] * cds_interp(a.hi.w, b.hi.w, c.hi.w, e.hi.w, tp.wt);
      s = 0.f + s;  // (level sum of ATen's cascade: one 16-row level for C = 8)
      if (s > mx) {
        const float sc = expf(mx - s);
        const float shift = (Z == 0.f) ? 0.f : (mx - s) * Z;
        T = sc * (T + shift);
        Z = Z * sc;
        mx = s;
      }
      const float dlt = s - mx;
      const float ev = expf(dlt);
      Z += ev;
      T = fmaf(dlt, ev, T);
    }
  }
  if (active) entropy[(size_t)v * hw + pix] = logf(Z) - T / Z;
}

}  // namespace

// Launchers used by the extern "C" entry points in warp.hip.  Return false if the shape is not covered.
bool cds_warp_aggregate_lds_launch(const float* ref, const float* src, const float* vis, const WarpMats& wm,
                                   const float* hyp, float* volume, const float* vis_sum, int V, int C, int D, int h,
                                   int w, int hyp_pp, int flags, hipStream_t st) {
  if (C != 8 || V > 4 || !hyp_pp) return false;
  const int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  const int ntiles = tiles_x * tiles_y;
#define LAUNCH(VM)                                                                                                     \
  hipLaunchKernelGGL(warp_aggregate_lds_kernel<VM>, dim3(ntiles), dim3(256),                                           \
                     (size_t)VM * 2 * BOX_CAP * sizeof(float4) + 4 * VM * 4 * sizeof(int), st, ref, src, vis, wm, hyp, \
                     volume, vis_sum, V, D, h, w, flags, tiles_x, ntiles)
  if (V <= 2) LAUNCH(2);
  else LAUNCH(4);
#undef LAUNCH
  return true;
}

bool cds_warp_entropy_lds_launch(const float* ref, const float* src, const WarpMats& wm, const float* hyp,
                                 float* entropy, int V, int C, int D, int h, int w, int hyp_pp, hipStream_t st) {
  if (C != 8 || !hyp_pp) return false;
  const int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  const int ntiles = tiles_x * tiles_y;
  hipLaunchKernelGGL(warp_entropy_lds_kernel, dim3(ntiles * V), dim3(256),
                     (size_t)2 * BOX_CAP * sizeof(float4) + 4 * 4 * sizeof(int), st, ref, src, wm, hyp, entropy, V, D, h,
                     w, tiles_x, ntiles);
  return true;
}
