// Shared device helpers of the plane-sweep warp kernels (warp.hip, warp_lds.hip).
#pragma once
#include "cds_common.hpp"

// The homographies of a call live in DEVICE memory: mats_d = [V][12] floats (rows of (P_src P_ref^-1)[:3,:3], then its [:3,3]), part of the
// call's geometry block (cds_mvsnet_amd/geometry.py: one small host -> device copy per forward).  They were by-value kernel
// arguments until round 5; as device data a captured hipGraph of the forward / the training step is re-targeted at new cameras by
// rewriting the block before the replay.  The values are wave-uniform: the loads are scalar (s_load) and what a
// kernel keeps of them stays in SGPRs.
template <int NV>
struct MatRegs {
  float m[NV][12];
  __device__ __forceinline__ explicit MatRegs(const float* __restrict__ mats_d, int V = NV) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 12; ++i) m[v][i] = v < V ? mats_d[v * 12 + i] : 0.f;
  }
};

struct Taps {
  int off[4];   // element offset of the texel (pixel index, not yet multiplied by C); -1 if outside
  float wt[4];  // nw, ne, sw, se
};

__device__ __forceinline__ void cds_row_terms(const float* __restrict__ m, float x, float y, float r[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = m[3 * i + 0] * x;
    a = fmaf(m[3 * i + 1], y, a);
    a = fmaf(m[3 * i + 2], 1.0f, a);
    r[i] = a;
  }
}

__device__ __forceinline__ Taps cds_taps(const float r[3], const float* __restrict__ t, float d, int h,
                                         int w, float half_w, float half_h) {
  float px = r[0] * d + t[0];
  float py = r[1] * d + t[1];
  float pz = r[2] * d + t[2];
  float z = pz + 1e-6f;
#ifdef CDS_WARP_RELAXED   // A/B build only, see positions2() in warp_lds.hip
  float ry = __builtin_amdgcn_rcpf(z);
  float ix = px * ry, iy = py * ry;
#else
  float u = px / z;
  float v = py / z;
  float gx = u / half_w - 1.0f;
  float gy = v / half_h - 1.0f;
  float ix = (gx + 1.0f) * half_w;
  float iy = (gy + 1.0f) * half_h;
#endif
  float x0f = floorf(ix), y0f = floorf(iy);
  float wx = ix - x0f, ex = 1.0f - wx;
  float ny = iy - y0f, sy = 1.0f - ny;
  Taps tp;
  tp.wt[0] = sy * ex;
  tp.wt[1] = sy * wx;
  tp.wt[2] = ny * ex;
  tp.wt[3] = ny * wx;
  // Range test in float first: NaN / inf / huge coordinates fail every comparison -> all taps "outside".
  bool x0ok = (x0f >= 0.0f) && (x0f <= (float)(w - 1));
  bool x1ok = (x0f >= -1.0f) && (x0f <= (float)(w - 2));
  bool y0ok = (y0f >= 0.0f) && (y0f <= (float)(h - 1));
  bool y1ok = (y0f >= -1.0f) && (y0f <= (float)(h - 2));
  int x0 = (x0ok || x1ok) ? (int)x0f : 0;
  int y0 = (y0ok || y1ok) ? (int)y0f : 0;
  int base = y0 * w + x0;
  tp.off[0] = (x0ok && y0ok) ? base : -1;
  tp.off[1] = (x1ok && y0ok) ? base + 1 : -1;
  tp.off[2] = (x0ok && y1ok) ? base + w : -1;
  tp.off[3] = (x1ok && y1ok) ? base + w + 1 : -1;
  return tp;
}

__device__ __forceinline__ float4 cds_ld4(const float* __restrict__ p, int off, int C, int c0) {
  if (off < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
  return *reinterpret_cast<const float4*>(p + (size_t)off * C + c0);
}

__device__ __forceinline__ float cds_interp(float a, float b, float c, float d, const float wt[4]) {
  float o = a * wt[0];
  o = fmaf(b, wt[1], o);
  o = fmaf(c, wt[2], o);
  o = fmaf(d, wt[3], o);
  return o;
}

#define CDS_TILE_X 64
#define CDS_TILE_Y 4

