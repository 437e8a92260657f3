// Shared device helpers of the 2D feature path (conv2d.hip: planar activations; feat_cl.hip: channels-last activations):
// fp64 wave reductions for the InstanceNorm records and the per-pixel part of the DynamicConv epilogue.
#pragma once
#include "cds_common.hpp"

namespace {

// fp64 sum over the 64 lanes of a wave without the LDS crossbar: four DPP steps inside each row of 16 lanes, then the
// four row totals through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Epipoles of the images of a call: DEVICE [N][2] (x, y in pixels of the layer's resolution), part of the call's geometry block like the
// homographies (warp_common.hpp): by-value kernel arguments until round 5, device data since, so that a captured graph can be replayed
// with new cameras.
struct EpiBatch {
  const float* p;
  __device__ __forceinline__ float x(int n) const { return p[2 * n]; }
  __device__ __forceinline__ float y(int n) const { return p[2 * n + 1]; }
};

// Per-pixel part of the DynamicConv epilogue (dynamic_conv.py:100-121) from the K x 3 curvature responses att[k][0..2] of pixel
// (x, y): unit epipolar direction, projection onto [u^2, 2uv, v^2], 1x1 MLP (BatchNorm folded), softmax(./T).  Returns the blend
// weights in logit[] and the weighted curvature.  ONE definition of this arithmetic for every kernel that applies it.
template <int K>
__device__ __forceinline__ float blend_from_att(const float att[K][3], int x, int y, float epi_x, float epi_y,
                                                const float* __restrict__ w1, const float* __restrict__ b1,
                                                const float* __restrict__ w2, float temperature, float logit[K]) {
  float u = (float)x - epi_x, v = (float)y - epi_y;
  const float nrm = sqrtf(u * u + v * v);
  u = u / (nrm + 1e-6f);
  v = v / (nrm + 1e-6f);
  const float b0 = u * u, b1v = 2.0f * u * v, b2 = v * v;
  float curv[K];
#pragma unroll
  for (int k = 0; k < K; ++k) curv[k] = att[k][0] * b0 + att[k][1] * b1v + att[k][2] * b2;
  float hid[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s = fmaf(w1[j * K + k], curv[k], s);
    hid[j] = fmaxf(s + b1[j], 0.f);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s = fmaf(w2[k * 4 + j], hid[j], s);
    logit[k] = s / temperature;
    mx = fmaxf(mx, logit[k]);
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    logit[k] = expf(logit[k] - mx);
    den += logit[k];
  }
  float nc = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    logit[k] = logit[k] / den;
    nc = nc + curv[k] * logit[k];
  }
  return nc;
}

// The same with the curvature responses read from a planar branch tensor [K][..][Cout+3][hw] (bstride = stride between kernel sizes).
template <int K>
__device__ __forceinline__ float blend_weights(const float* __restrict__ branch, size_t bstride, int Cout, int hw, int p,
                                               int W, float epi_x, float epi_y, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                               float temperature, float logit[K]) {
  float att[K][3];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float* a = branch + k * bstride + (size_t)Cout * hw + p;
    att[k][0] = a[0];
    att[k][1] = a[hw];
    att[k][2] = a[2 * (size_t)hw];
  }
  return blend_from_att<K>(att, p % W, p / W, epi_x, epi_y, w1, b1, w2, temperature, logit);
}

}  // namespace
