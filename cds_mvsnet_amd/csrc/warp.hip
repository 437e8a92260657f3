// Plane-sweep warp kernels: homography warp, K1 warp-correlate-entropy, K3 warp-aggregate.
// Replaces models/utils/warping.py:69-104 + models/model.py:44-50,57-60,74 of the reference.
//
// fp32 operation order (pinned against ATen's CPU kernels, see oracle/cds_oracle.py):
//   r   = fma(R2, 1, fma(R1, y, R0*x))            (sgemm with K=3)
//   p   = r*d (rounded) + t (rounded)
//   u   = p.x / (p.z + 1e-6)   v = p.y / (p.z + 1e-6)        true IEEE division
//   ix  = ((u / ((w-1)/2) - 1) + 1) * ((w-1)/2)              normalise / un-normalise round trip
//   val = fma(v_se, w_se, fma(v_sw, w_sw, fma(v_ne, w_ne, v_nw*w_nw)))
//   in_prod = ref*val ; volume += in_prod*vis                (three separate roundings)
#include <stdlib.h>

#include "warp_common.hpp"

// ---------------------------------------------------------------------------------------------
// plain warp  -> out [C][D][h][w]
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, const float* __restrict__ mats_d,
                                                   const float* __restrict__ hyp, float* __restrict__ out,
                                                   int D, int h, int w, int hyp_pp, int tiles_x,
                                                   int ntiles) {
  int tile = cds_xcd_remap(blockIdx.x, ntiles);
  int tx = tile % tiles_x, ty = tile / tiles_x;
  int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const MatRegs<1> mats(mats_d);
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  float r[3];
  cds_row_terms(mats.m[0], (float)x, (float)y, r);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)y * w + x;
  for (int d = 0; d < D; ++d) {
    float dv = hyp_pp ? hyp[d * hw + pix] : hyp[d];
    Taps tp = cds_taps(r, mats.m[0] + 9, dv, h, w, half_w, half_h);
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 4) {
      float4 a = cds_ld4(src, tp.off[0], C, c0), b = cds_ld4(src, tp.off[1], C, c0);
      float4 c = cds_ld4(src, tp.off[2], C, c0), e = cds_ld4(src, tp.off[3], C, c0);
      out[((size_t)(c0 + 0) * D + d) * hw + pix] = cds_interp(a.x, b.x, c.x, e.x, tp.wt);
      out[((size_t)(c0 + 1) * D + d) * hw + pix] = cds_interp(a.y, b.y, c.y, e.y, tp.wt);
      out[((size_t)(c0 + 2) * D + d) * hw + pix] = cds_interp(a.z, b.z, c.z, e.z, tp.wt);
      out[((size_t)(c0 + 3) * D + d) * hw + pix] = cds_interp(a.w, b.w, c.w, e.w, tp.wt);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1: entropy over D of softmax(sum_C ref*warp), one thread per (pixel, view).
// Online softmax statistics: m (running max), Z = sum e^(s-m), T = sum (s-m) e^(s-m);
// entropy = log Z - T/Z   (algebraically -sum p log p, finite where the textbook form is 0*log 0).
// ---------------------------------------------------------------------------------------------
// DS = depth slices per pixel: DS = 1 -> 64x4 pixel tile, one thread per pixel over all planes; DS = 4 -> 64x1 pixel tile,
// the four waves of a workgroup take the planes d = s, s + 4, ... and their (m, Z, T) statistics are merged through LDS
// (small images: 4x the workgroups, 1/4 of the serial plane loop).
template <int C, int DS>
__global__ __launch_bounds__(256) void warp_entropy_kernel(const float* __restrict__ ref,
                                                           const float* __restrict__ src, const float* __restrict__ mats_d,
                                                           const float* __restrict__ hyp,
                                                           float* __restrict__ entropy, int V, int D, int h,
                                                           int w, int hyp_pp, int tiles_x, int ntiles) {
  // view is the fastest-varying logical index so the V blocks of one pixel tile are co-scheduled
  // and share the tile's hypotheses through L2.
  int lin = cds_xcd_remap(blockIdx.x, ntiles * V);
  int v = lin % V;
  int tile = lin / V;
  int tx = tile % tiles_x, ty = tile / tiles_x;
  const int slice = (DS == 1) ? 0 : (int)(threadIdx.x >> 6);
  int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  int y = (DS == 1) ? ty * CDS_TILE_Y + (int)(threadIdx.x >> 6) : ty;
  const bool inside = x < w && y < h;
  if (DS == 1 && !inside) return;
  x = min(x, w - 1);
  y = min(y, h - 1);
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)y * w + x;
  const float* __restrict__ srcv = src + (size_t)v * hw * C;
  float rf[C];
#pragma unroll
  for (int c = 0; c < C; ++c) rf[c] = ref[((size_t)v * C + c) * hw + pix];
  const MatRegs<1> mats(mats_d + v * 12);
  float r[3];
  cds_row_terms(mats.m[0], (float)x, (float)y, r);
  float m = -INFINITY, Z = 0.f, T = 0.f;
  for (int d = slice; d < D; d += DS) {
    float dv = hyp_pp ? hyp[d * hw + pix] : hyp[d];
    Taps tp = cds_taps(r, mats.m[0] + 9, dv, h, w, half_w, half_h);
    float s = 0.f;
    // ATen's outer-dim sum: sequential inside 16-row levels, then level sums added.
#pragma unroll
    for (int cb = 0; cb < C; cb += 16) {
      float part = 0.f;
#pragma unroll
      for (int c0 = cb; c0 < cb + 16 && c0 < C; c0 += 4) {
        float4 a = cds_ld4(srcv, tp.off[0], C, c0), b = cds_ld4(srcv, tp.off[1], C, c0);
        float4 c = cds_ld4(srcv, tp.off[2], C, c0), e = cds_ld4(srcv, tp.off[3], C, c0);
        part = part + rf[c0 + 0] * cds_interp(a.x, b.x, c.x, e.x, tp.wt);
        part = part + rf[c0 + 1] * cds_interp(a.y, b.y, c.y, e.y, tp.wt);
        part = part + rf[c0 + 2] * cds_interp(a.z, b.z, c.z, e.z, tp.wt);
        part = part + rf[c0 + 3] * cds_interp(a.w, b.w, c.w, e.w, tp.wt);
      }
      s = s + part;
    }
    if (s > m) {  // also taken on the first plane (m = -inf)
      float sc = expf(m - s);  // exp(-inf) = 0 on the first plane
      // T' = sum (s_i - m') e_i' = sc*(T + (m - m')*Z);  guard 0*inf on the first plane
      float shift = (Z == 0.f) ? 0.f : (m - s) * Z;
      T = sc * (T + shift);
      Z = Z * sc;
      m = s;
    }
    float dlt = s - m;
    float e = expf(dlt);
    Z += e;
    T = fmaf(dlt, e, T);
  }
  if (DS > 1) {
    __shared__ float red[3][DS][64];
    const int lane = threadIdx.x & 63;
    red[0][slice][lane] = m;
    red[1][slice][lane] = Z;
    red[2][slice][lane] = T;
    __syncthreads();
    if (slice != 0) return;
    float mm = -INFINITY;
#pragma unroll
    for (int i = 0; i < DS; ++i) mm = fmaxf(mm, red[0][i][lane]);
    float Zs = 0.f, Ts = 0.f;
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      const float mi = red[0][i][lane], Zi = red[1][i][lane], Ti = red[2][i][lane];
      if (Zi > 0.f) {            // a slice without planes (D < DS) contributes nothing
        const float sc = expf(mi - mm);
        Zs += Zi * sc;
        Ts += sc * (Ti + (mi - mm) * Zi);
      }
    }
    Z = Zs;
    T = Ts;
    if (!inside) return;
  }
  entropy[(size_t)v * hw + pix] = logf(Z) - T / Z;
}

// ---------------------------------------------------------------------------------------------
// K3: volume[c][d][p] = sum_v vis_v[p] * ref_v[c][p] * warp_v[c][d][p]  (/ (vis_sum+1e-6)).
// One thread per (pixel, group of CG=8 channels); all views in the inner loop so every volume
// element is produced in registers and stored exactly once.
// ---------------------------------------------------------------------------------------------
template <int VMAX>
__global__ __launch_bounds__(256) void warp_aggregate_kernel(
    const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ vis, const float* __restrict__ mats_d,
    const float* __restrict__ hyp, float* __restrict__ volume, const float* __restrict__ vis_sum, int V, int C,
    int D, int h, int w, int hyp_pp, int flags, int tiles_x, int ntiles, int nseg, int seg_planes) {
  constexpr int CG = 8;
  const MatRegs<VMAX> mats(mats_d, V);
  const int ngroups = C / CG;
  // depth segment fastest, then channel group: the blocks of one pixel tile run together (same features in L2).
  // Small images (cascade stage 1: 160x128) would otherwise give 1-2 workgroups per CU looping over all planes.
  int lin = cds_xcd_remap(blockIdx.x, ntiles * ngroups * nseg);
  const int seg = lin % nseg;
  lin /= nseg;
  int g = lin % ngroups;
  int tile = lin / ngroups;
  int tx = tile % tiles_x, ty = tile / tiles_x;
  int x = tx * CDS_TILE_X + (threadIdx.x & 63);
  int y = ty * CDS_TILE_Y + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float half_w = (float)((w - 1) / 2.0), half_h = (float)((h - 1) / 2.0);
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)y * w + x;
  const int c_base = g * CG;

  float rf[VMAX][CG];
  float vw[VMAX];
  float r[VMAX][3];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    if (v < V) {
      vw[v] = vis[(size_t)v * hw + pix];
#pragma unroll
      for (int c = 0; c < CG; ++c) rf[v][c] = ref[((size_t)v * C + c_base + c) * hw + pix];
      cds_row_terms(mats.m[v], (float)x, (float)y, r[v]);
    }
  }
  const bool accumulate = flags & CDS_AGG_ACCUMULATE;
  const bool normalize = flags & CDS_AGG_NORMALIZE;
  const bool cl = flags & CDS_AGG_CHANNELS_LAST;                  // volume [D][h][w][C] instead of [C][D][h][w]
  const float denom = (normalize ? vis_sum[pix] : 1.0f) + 1e-6f;  // vis_sum finalised by vis_sum_kernel

  const int dbeg = seg * seg_planes, dend = min(D, dbeg + seg_planes);
  if (dbeg >= dend) return;
  float dnext = hyp_pp ? hyp[(size_t)dbeg * hw + pix] : hyp[dbeg];
  for (int d = dbeg; d < dend; ++d) {
    float dv = dnext;
    if (d + 1 < dend) dnext = hyp_pp ? hyp[(size_t)(d + 1) * hw + pix] : hyp[d + 1];
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c)
      acc[c] = accumulate ? (cl ? volume[((size_t)d * hw + pix) * C + c_base + c] : volume[((size_t)(c_base + c) * D + d) * hw + pix]) : 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      if (v < V) {
        const float* __restrict__ srcv = src + (size_t)v * hw * C;
        Taps tp = cds_taps(r[v], mats.m[v] + 9, dv, h, w, half_w, half_h);
#pragma unroll
        for (int c0 = 0; c0 < CG; c0 += 4) {
          float4 a = cds_ld4(srcv, tp.off[0], C, c_base + c0), b = cds_ld4(srcv, tp.off[1], C, c_base + c0);
          float4 c = cds_ld4(srcv, tp.off[2], C, c_base + c0), e = cds_ld4(srcv, tp.off[3], C, c_base + c0);
          float w0 = cds_interp(a.x, b.x, c.x, e.x, tp.wt);
          float w1 = cds_interp(a.y, b.y, c.y, e.y, tp.wt);
          float w2 = cds_interp(a.z, b.z, c.z, e.z, tp.wt);
          float w3 = cds_interp(a.w, b.w, c.w, e.w, tp.wt);
          acc[c0 + 0] = acc[c0 + 0] + (rf[v][c0 + 0] * w0) * vw[v];
          acc[c0 + 1] = acc[c0 + 1] + (rf[v][c0 + 1] * w1) * vw[v];
          acc[c0 + 2] = acc[c0 + 2] + (rf[v][c0 + 2] * w2) * vw[v];
          acc[c0 + 3] = acc[c0 + 3] + (rf[v][c0 + 3] * w3) * vw[v];
        }
      }
    }
    if (cl) {
      float o[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) o[c] = normalize ? acc[c] / denom : acc[c];
      float4* dst = reinterpret_cast<float4*>(volume + ((size_t)d * hw + pix) * C + c_base);
      dst[0] = make_float4(o[0], o[1], o[2], o[3]);
      dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        float o = normalize ? acc[c] / denom : acc[c];
        __builtin_nontemporal_store(o, &volume[((size_t)(c_base + c) * D + d) * hw + pix]);
      }
    }
  }
}

// vis_sum[p] = (accumulate ? vis_sum[p] : 0) + vis_0[p] + vis_1[p] + ...   (view order, model.py:58)
__global__ void vis_sum_kernel(const float* __restrict__ vis, float* __restrict__ vis_sum, int V, int hw,
                               int accumulate) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  float s = accumulate ? vis_sum[p] : 0.f;
  for (int v = 0; v < V; ++v) s = s + vis[(size_t)v * hw + p];
  vis_sum[p] = s;
}

__global__ void volume_normalize_kernel(float* __restrict__ vol, const float* __restrict__ vis_sum, size_t hw,
                                        size_t planes) {
  size_t n4 = (hw % 4 == 0) ? hw / 4 : 0;  // rows are 16-byte aligned only when hw % 4 == 0
  for (size_t pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    float4* row = reinterpret_cast<float4*>(vol + pl * hw);
    const float4* vs = reinterpret_cast<const float4*>(vis_sum);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      float4 a = row[i], s = vs[i];
      a.x = a.x / (s.x + 1e-6f);
      a.y = a.y / (s.y + 1e-6f);
      a.z = a.z / (s.z + 1e-6f);
      a.w = a.w / (s.w + 1e-6f);
      row[i] = a;
    }
    if (blockIdx.x == 0) {
      for (size_t i = n4 * 4 + threadIdx.x; i < hw; i += blockDim.x) vol[pl * hw + i] = vol[pl * hw + i] / (vis_sum[i] + 1e-6f);
    }
  }
}

// channels-last volume [n pixels x planes][C]: every voxel's C channels scaled by 1 / (vis_sum[pixel] + 1e-6)
__global__ void volume_normalize_cl_kernel(float* __restrict__ vol, const float* __restrict__ vis_sum, size_t hw, size_t nvox,
                                           int c4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvox * c4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t vox = i / c4;
    const float s = vis_sum[vox % hw] + 1e-6f;
    float4 a = reinterpret_cast<float4*>(vol)[i];
    a.x = a.x / s; a.y = a.y / s; a.z = a.z / s; a.w = a.w / s;
    reinterpret_cast<float4*>(vol)[i] = a;
  }
}

__global__ void chw_to_hwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int hw) {
  // one thread per (pixel, 4 channels): coalesced plane reads, 16-byte texel stores
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  for (int c0 = 0; c0 < C; c0 += 4) {
    float4 v;
    v.x = src[(size_t)(c0 + 0) * hw + p];
    v.y = src[(size_t)(c0 + 1) * hw + p];
    v.z = src[(size_t)(c0 + 2) * hw + p];
    v.w = src[(size_t)(c0 + 3) * hw + p];
    *reinterpret_cast<float4*>(dst + (size_t)p * C + c0) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// The reference-signature boundary of one stage in ONE launch (models/model.py:16-40: `features` is a list over the source
// views of {'ref': (fea, nc_sum, nc), 'src': (fea, nc_sum, _)}): stack the reference copies' features [V][C][h][w], transpose the
// source features to the channels-last maps K1 / K3 gather from [V][h][w][C], stack the reference curvature maps, average the
// per-pair curvature sums over the views (model.py:59-60, same operation order as pair_mean + view_mean), and leave
// max |ref| x max |src| - the bound of the normalised volume CostRegNet's split-f16 layers scale by - in state[0].
// Replaces 4 + V launches of the harness (cat, V transposes, 2 abs, 2 amax, mul) and their 4 extra passes over the features.
// ---------------------------------------------------------------------------------------------
struct StagePtrs {
  const float* ref[CDS_MAX_VIEWS];
  const float* src[CDS_MAX_VIEWS];
  const float* ref_nc[CDS_MAX_VIEWS];
  const float* ref_ncsum[CDS_MAX_VIEWS];
  const float* src_ncsum[CDS_MAX_VIEWS];
};

__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
  return v;
}

template <int C>
__global__ __launch_bounds__(256) void stage_inputs_kernel(StagePtrs ptrs, float* __restrict__ ref_out, float* __restrict__ src_out,
                                                           float* __restrict__ ref_nc_out, float* __restrict__ nc_mean_out,
                                                           unsigned* __restrict__ state, int V, int hw) {
  const int v = blockIdx.y;
  // |x| as an integer: the order of non-negative floats, with every NaN above +inf (a NaN feature makes the bound NaN, like amax)
  unsigned mref = 0u, msrc = 0u;
  const float* __restrict__ rf = ptrs.ref[v];
  const float* __restrict__ sf = ptrs.src[v];
  float* __restrict__ ro = ref_out + (size_t)v * C * hw;
  // ~1024 workgroups walk the pixels with a grid stride and leave one pair of maxima each
  for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
    float a[C], b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {       // all loads first: 2 C independent coalesced plane reads in flight
      a[c] = rf[(size_t)c * hw + p];
      b[c] = sf[(size_t)c * hw + p];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      ro[(size_t)c * hw + p] = a[c];
      mref = max(mref, __float_as_uint(a[c]) & 0x7fffffffu);
      msrc = max(msrc, __float_as_uint(b[c]) & 0x7fffffffu);
    }
    float* __restrict__ so = src_out + ((size_t)v * hw + p) * C;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 4)
      *reinterpret_cast<float4*>(so + c0) = make_float4(b[c0], b[c0 + 1], b[c0 + 2], b[c0 + 3]);
    if (ref_nc_out) ref_nc_out[(size_t)v * hw + p] = ptrs.ref_nc[v][p];
    if (nc_mean_out && v == 0) {
      float s = (ptrs.ref_ncsum[0][p] + ptrs.src_ncsum[0][p]) / 2.0f;
      for (int u = 1; u < V; ++u) s = s + (ptrs.ref_ncsum[u][p] + ptrs.src_ncsum[u][p]) / 2.0f;
      nc_mean_out[p] = s / (float)V;
    }
  }
  if (!state) return;
  __shared__ unsigned red[2][4];
  mref = wave_umax(mref);
  msrc = wave_umax(msrc);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mref;
    red[1][threadIdx.x >> 6] = msrc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // one plain store per workgroup, merged by stage_bound_kernel: same-address device-scope atomics serialise at ~18 ns each on
    // this part (one per wave of a one-pixel-per-thread grid: 46 000 of them, 0.85 ms at 640x512; one per workgroup of this grid
    // with a completion counter: still ~0.1 ms)
    const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    state[1 + b] = max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]));
    state[1 + nb + b] = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
  }
}

__global__ __launch_bounds__(256) void stage_bound_kernel(unsigned* __restrict__ state, int nb) {
  unsigned mref = 0u, msrc = 0u;
  for (int i = threadIdx.x; i < nb; i += 256) {
    mref = max(mref, state[1 + i]);
    msrc = max(msrc, state[1 + nb + i]);
  }
  __shared__ unsigned red[2][4];
  mref = wave_umax(mref);
  msrc = wave_umax(msrc);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = mref;
    red[1][threadIdx.x >> 6] = msrc;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    reinterpret_cast<float*>(state)[0] = __uint_as_float(max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3]))) *
                                         __uint_as_float(max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3])));
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
// LDS-staged fast paths (warp_lds.hip); return false when the shape is not covered.
bool cds_warp_aggregate_lds_launch(const float* ref, const float* src, const float* vis, const float* wm,
                                   const float* hyp, float* volume, const float* vis_sum, int V, int C, int D, int h,
                                   int w, int hyp_pp, int flags, hipStream_t st, int hs = 0, int y_off = 0);
bool cds_warp_entropy_lds_launch(const float* ref, const float* src, const float* wm, const float* hyp,
                                 float* entropy, int V, int C, int D, int h, int w, int hyp_pp, bool fast, hipStream_t st, int hs = 0,
                                 int y_off = 0);
static bool cds_use_lds_path() {
  return !cds_env_is("CDS_WARP_DIRECT", '1');   // CDS_WARP_DIRECT=1 forces the direct (L1 gather) kernels
}

static bool cds_warp_args_ok(int V, int C, int D, int h, int w) {
  return V >= 1 && V <= CDS_MAX_VIEWS && (C == 8 || C == 16 || C == 32) && D >= 1 && h >= 1 && w >= 1;
}

// Shapes the LDS-staged row-window kernels cover (the conditions cds_warp_*_lds_launch rejects, checked BEFORE anything is launched:
// the window entry points have no direct-kernel fallback and must not leave side effects behind an error return).
static bool cds_warp_window_ok(int V, int C, int D, int h, int w, int hs, int y_off) {
  return cds_warp_args_ok(V, C, D, h, w) && w >= 2 && hs >= 2 && hs >= h && y_off >= 0 && y_off + h <= hs &&
         (size_t)D * h * w * 4 < ((size_t)1 << 32);
}

extern "C" int cds_chw_to_hwc_f32(const float* src_chw, float* dst_hwc, int C, int h, int w, void* stream) {
  if (!src_chw || !dst_hwc || C < 4 || (C % 4) || h < 1 || w < 1) return CDS_EINVAL;
  int hw = h * w;
  hipLaunchKernelGGL(chw_to_hwc_kernel, dim3(cds_ceil_div(hw, 256)), dim3(256), 0, (hipStream_t)stream, src_chw,
                     dst_hwc, C, hw);
  return cds_launch_status();
}

extern "C" int cds_stage_inputs_f32(const float* const* ref_feas, const float* const* src_feas, const float* const* ref_nc,
                                    const float* const* ref_ncsum, const float* const* src_ncsum, float* ref_chw_out,
                                    float* src_hwc_out, float* ref_nc_out, float* nc_mean_out, float* state, int V, int C, int h,
                                    int w, void* stream) {
  if (!ref_feas || !src_feas || !ref_chw_out || !src_hwc_out || V < 1 || V > CDS_MAX_VIEWS || (C != 8 && C != 16 && C != 32) ||
      h < 1 || w < 1 || (size_t)h * w > 0x7fffffffu / 32u)
    return CDS_EINVAL;
  if ((ref_nc_out && !ref_nc) || (nc_mean_out && (!ref_ncsum || !src_ncsum))) return CDS_EINVAL;
  StagePtrs ptrs = {};
  for (int v = 0; v < V; ++v) {
    if (!ref_feas[v] || !src_feas[v]) return CDS_EINVAL;
    ptrs.ref[v] = ref_feas[v];
    ptrs.src[v] = src_feas[v];
    if (ref_nc_out) { if (!ref_nc[v]) return CDS_EINVAL; ptrs.ref_nc[v] = ref_nc[v]; }
    if (nc_mean_out) {
      if (!ref_ncsum[v] || !src_ncsum[v]) return CDS_EINVAL;
      ptrs.ref_ncsum[v] = ref_ncsum[v];
      ptrs.src_ncsum[v] = src_ncsum[v];
    }
  }
  const int hw = h * w;
  const dim3 grid(min(cds_ceil_div(hw, 256), cds_ceil_div(1024, V)), V), block(256);   // <= ~1024 workgroups
  hipStream_t st = (hipStream_t)stream;
  unsigned* su = reinterpret_cast<unsigned*>(state);
#define LAUNCH(CC) \
  hipLaunchKernelGGL(stage_inputs_kernel<CC>, grid, block, 0, st, ptrs, ref_chw_out, src_hwc_out, ref_nc_out, nc_mean_out, su, V, hw)
  if (C == 8) LAUNCH(8);
  else if (C == 16) LAUNCH(16);
  else LAUNCH(32);
#undef LAUNCH
  if (state) hipLaunchKernelGGL(stage_bound_kernel, dim3(1), block, 0, st, su, (int)(grid.x * grid.y));
  return cds_launch_status();
}

extern "C" int cds_homo_warp_f32(const float* src_hwc, const float* mat, const float* hyp, float* out, int C,
                                 int D, int h, int w, int hyp_per_pixel, void* stream) {
  if (!src_hwc || !mat || !hyp || !out || !cds_warp_args_ok(1, C, D, h, w)) return CDS_EINVAL;
  const float* wm = mat;
  int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  int ntiles = tiles_x * tiles_y;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(CC)                                                                                              \
  hipLaunchKernelGGL(warp_kernel<CC>, dim3(ntiles), dim3(256), 0, st, src_hwc, wm, hyp, out, D, h, w, hyp_per_pixel, \
                     tiles_x, ntiles)
  if (C == 8) LAUNCH(8);
  else if (C == 16) LAUNCH(16);
  else LAUNCH(32);
#undef LAUNCH
  return cds_launch_status();
}

extern "C" int cds_warp_entropy_f32(const float* ref_chw, const float* src_hwc, const float* mats,
                                    const float* hyp, float* entropy, int V, int C, int D, int h, int w,
                                    int hyp_per_pixel, void* stream) {
  return cds_warp_entropy_flags_f32(ref_chw, src_hwc, mats, hyp, entropy, V, C, D, h, w, hyp_per_pixel, 0, stream);
}

extern "C" int cds_warp_entropy_flags_f32(const float* ref_chw, const float* src_hwc, const float* mats,
                                          const float* hyp, float* entropy, int V, int C, int D, int h, int w,
                                          int hyp_per_pixel, int flags, void* stream) {
  if (!ref_chw || !src_hwc || !mats || !hyp || !entropy || !cds_warp_args_ok(V, C, D, h, w)) return CDS_EINVAL;
  const float* wm = mats;
  int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  int ntiles = tiles_x * tiles_y;
  hipStream_t st = (hipStream_t)stream;
  if (cds_use_lds_path() && cds_warp_entropy_lds_launch(ref_chw, src_hwc, wm, hyp, entropy, V, C, D, h, w, hyp_per_pixel,
                                                          (flags & CDS_WARP_FAST_POSITIONS) != 0, st))
    return cds_launch_status();
  // small images: one pixel row of 64 per workgroup, planes split over its four waves (>= ~6 workgroups per CU otherwise)
  const bool dsplit = (long)ntiles * V < 6L * 256 && D >= 8;
  const int tiles_y1 = dsplit ? h : tiles_y, nt = tiles_x * tiles_y1;
#define LAUNCH(CC)                                                                                                 \
  do {                                                                                                             \
    if (dsplit)                                                                                                    \
      hipLaunchKernelGGL((warp_entropy_kernel<CC, 4>), dim3(nt * V), dim3(256), 0, st, ref_chw, src_hwc, wm, hyp,  \
                         entropy, V, D, h, w, hyp_per_pixel, tiles_x, nt);                                         \
    else                                                                                                           \
      hipLaunchKernelGGL((warp_entropy_kernel<CC, 1>), dim3(nt * V), dim3(256), 0, st, ref_chw, src_hwc, wm, hyp,  \
                         entropy, V, D, h, w, hyp_per_pixel, tiles_x, nt);                                         \
  } while (0)
  if (C == 8) LAUNCH(8);
  else if (C == 16) LAUNCH(16);
  else LAUNCH(32);
#undef LAUNCH
  return cds_launch_status();
}

// Row-window forms (pixel-slab sharding, cds_mvsnet_amd/distributed.py): the reference-side tensors (ref features, hypotheses,
// weights, entropy / volume outputs) cover rows [y_off, y_off + h) of the hs x w image grid, the source maps the whole grid.
// Sample positions are computed from the GLOBAL pixel row, so the window of a result is bit-identical to the same rows of the
// full-grid call.  LDS-staged kernels only (per-pixel hypotheses, C in {8, 16, 32}): anything else is CDS_EINVAL.
extern "C" int cds_warp_entropy_window_f32(const float* ref_chw, const float* src_hwc, const float* mats, const float* hyp,
                                           float* entropy, int V, int C, int D, int h, int w, int hs, int y_off, int flags,
                                           void* stream) {
  if (!ref_chw || !src_hwc || !mats || !hyp || !entropy || !cds_warp_window_ok(V, C, D, h, w, hs, y_off)) return CDS_EINVAL;
  const float* wm = mats;
  if (!cds_warp_entropy_lds_launch(ref_chw, src_hwc, wm, hyp, entropy, V, C, D, h, w, 1, (flags & CDS_WARP_FAST_POSITIONS) != 0,
                                   (hipStream_t)stream, hs, y_off))
    return CDS_EINVAL;
  return cds_launch_status();
}

extern "C" int cds_warp_aggregate_window_f32(const float* ref_chw, const float* src_hwc, const float* vis_w, const float* mats,
                                             const float* hyp, float* volume, float* vis_sum, int V, int C, int D, int h, int w,
                                             int hs, int y_off, int flags, void* stream) {
  if (!ref_chw || !src_hwc || !vis_w || !mats || !hyp || !volume || !vis_sum || !cds_warp_window_ok(V, C, D, h, w, hs, y_off))
    return CDS_EINVAL;
  const float* wm = mats;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vis_sum_kernel, dim3(cds_ceil_div(h * w, 256)), dim3(256), 0, st, vis_w, vis_sum, V, h * w,
                     flags & CDS_AGG_ACCUMULATE);
  if (!cds_warp_aggregate_lds_launch(ref_chw, src_hwc, vis_w, wm, hyp, volume, vis_sum, V, C, D, h, w, 1, flags, st, hs, y_off))
    return CDS_EINVAL;
  return cds_launch_status();
}

extern "C" int cds_warp_aggregate_f32(const float* ref_chw, const float* src_hwc, const float* vis_w,
                                      const float* mats, const float* hyp, float* volume, float* vis_sum, int V,
                                      int C, int D, int h, int w, int hyp_per_pixel, int flags, void* stream) {
  if (!ref_chw || !src_hwc || !vis_w || !mats || !hyp || !volume || !vis_sum || !cds_warp_args_ok(V, C, D, h, w))
    return CDS_EINVAL;
  const float* wm = mats;
  int tiles_x = cds_ceil_div(w, CDS_TILE_X), tiles_y = cds_ceil_div(h, CDS_TILE_Y);
  int ntiles = tiles_x * tiles_y;
  int ngroups = C / 8;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vis_sum_kernel, dim3(cds_ceil_div(h * w, 256)), dim3(256), 0, st, vis_w, vis_sum, V, h * w,
                     flags & CDS_AGG_ACCUMULATE);
  if (cds_use_lds_path() &&
      cds_warp_aggregate_lds_launch(ref_chw, src_hwc, vis_w, wm, hyp, volume, vis_sum, V, C, D, h, w, hyp_per_pixel, flags, st))
    return cds_launch_status();
  // depth segments: aim at >= ~6 workgroups per CU, at least 8 planes per segment
  int nseg = 1;
  while ((long)ntiles * ngroups * nseg < 6L * 256 && D / (2 * nseg) >= 8) nseg *= 2;
  const int seg_planes = cds_ceil_div(D, nseg);
  nseg = cds_ceil_div(D, seg_planes);
#define LAUNCH(VM)                                                                                                  \
  hipLaunchKernelGGL(warp_aggregate_kernel<VM>, dim3(ntiles * ngroups * nseg), dim3(256), 0, st, ref_chw, src_hwc, vis_w, wm, \
                     hyp, volume, vis_sum, V, C, D, h, w, hyp_per_pixel, flags, tiles_x, ntiles, nseg, seg_planes)
  if (V <= 2) LAUNCH(2);
  else if (V <= 4) LAUNCH(4);
  else if (V <= 6) LAUNCH(6);
  else LAUNCH(8);
#undef LAUNCH
  return cds_launch_status();
}

extern "C" int cds_volume_normalize_cl_f32(float* volume, const float* vis_sum, int C, int D, int hw, void* stream) {
  if (!volume || !vis_sum || C < 4 || (C % 4) || D < 1 || hw < 1) return CDS_EINVAL;
  const size_t nvox = (size_t)D * hw;
  size_t blocks = (nvox * (C / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(volume_normalize_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, volume, vis_sum,
                     (size_t)hw, nvox, C / 4);
  return cds_launch_status();
}

extern "C" int cds_volume_normalize_f32(float* volume, const float* vis_sum, int C, int D, int hw, void* stream) {
  if (!volume || !vis_sum || C < 1 || D < 1 || hw < 1) return CDS_EINVAL;
  size_t planes = (size_t)C * D;
  int gx = cds_ceil_div(cds_ceil_div(hw, 4), 256);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  int gy = planes > 4096 ? 4096 : (int)planes;
  hipLaunchKernelGGL(volume_normalize_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, volume, vis_sum,
                     (size_t)hw, planes);
  return cds_launch_status();
}
