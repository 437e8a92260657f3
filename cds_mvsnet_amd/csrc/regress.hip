// K5: softmax over D + soft-argmin depth + 4-window confidence   (models/model.py:90-92,
//     models/module.py:373-391)
// K6: per-pixel depth hypotheses of a cascade stage                (models/module.py:394-439,
//     models/model.py:176-193)
#include "cds_common.hpp"

// ---------------------------------------------------------------------------------------------
// K5.  A wave covers 16 consecutive pixels x 4 depth slices: lanes l and l^16, l^32 hold the same
// pixel and disjoint quarter-ranges of D, so every load instruction still touches 64-byte
// contiguous segments while D is split across lanes; the four partial (max, Z, sum) tuples are
// combined with wave shuffles (no LDS, no atomics).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cds_slice_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float cds_slice_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

__global__ __launch_bounds__(256) void softargmin_conf_kernel(const float* __restrict__ pre,
                                                              const float* __restrict__ hyp, float* __restrict__ depth,
                                                              float* __restrict__ conf, float* __restrict__ prob, int D,
                                                              int hw, int hyp_pp) {
  const int t = threadIdx.x;
  const int slice = (t >> 4) & 3;
  const int p_raw = blockIdx.x * 64 + (t & 15) + 16 * (t >> 6);
  const bool live = p_raw < hw;
  const size_t p = live ? p_raw : hw - 1;  // clamp: every lane stays active for the shuffles
  const int chunk = (D + 3) / 4;
  const int d0 = slice * chunk;
  const int d1 = min(D, d0 + chunk);

  // one pass over the logits (online softmax: the running maximum rescales the partial sums when it moves), so the
  // volume is read once instead of twice; the four slices of a pixel are merged with their own rescale factors
  float m = -INFINITY, Z = 0.f, Sd = 0.f, Si = 0.f;
  // eight planes per round: their loads are issued together (clamped plane index), the arithmetic runs in plane order as before
  // (one load + s_waitcnt vmcnt(0) per plane made the 48-plane walk a chain of memory round trips)
  for (int db = d0; db < d1; db += 8) {
    float xs[8], hs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int d = min(db + k, d1 - 1);
      xs[k] = pre[(size_t)d * hw + p];
      hs[k] = hyp_pp ? hyp[(size_t)d * hw + p] : hyp[d];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int d = db + k;
      if (d >= d1) break;
      const float x = xs[k], hv = hs[k];
      if (x > m) {                      // also taken on the first plane (m = -inf: the sums are still zero)
        const float sc = expf(m - x);   // exp(-inf) = 0
        Z *= sc;
        Sd *= sc;
        Si *= sc;
        m = x;
      }
      const float e = expf(x - m);
      Z += e;
      Sd = fmaf(e, hv, Sd);
      Si = fmaf(e, (float)d, Si);
    }
  }
  {
    const float M = cds_slice_max(m);
    const float sc = (d0 < d1) ? expf(m - M) : 0.f;   // a slice without planes (D < 4) contributes nothing
    Z = cds_slice_sum(Z * sc);
    Sd = cds_slice_sum(Sd * sc);
    Si = cds_slice_sum(Si * sc);
    m = M;
  }
  const float inv = 1.0f / Z;

  // confidence: slice k contributes probability at index i-1+k (zero outside [0,D))
  int i = (int)(Si / Z);  // trunc == floor, the value is >= 0
  i = max(0, min(D - 1, i));
  int j = i - 1 + slice;
  float pj = (j >= 0 && j < D) ? expf(pre[(size_t)j * hw + p] - m) / Z : 0.f;
  float c = cds_slice_sum(pj);

  if (live && slice == 0) {
    depth[p] = Sd / Z;
    conf[p] = c;
  }
  if (prob != nullptr && live) {
    for (int d = d0; d < d1; ++d) prob[(size_t)d * hw + p] = expf(pre[(size_t)d * hw + p] - m) * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// K6.  ATen's linear resize (align_corners=False): src = scale*(dst+0.5)-0.5 clamped at 0,
// i0 = trunc(src), i1 = i0 + (i0 < in-1), l1 = src-i0, l0 = 1-l1,
// value = fma(v0, l0, v1*l1)      (operation order measured against F.interpolate on CPU).
// ---------------------------------------------------------------------------------------------
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp cds_lerp_index(int dst, int n_in, float scale) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  Lerp L;
  L.i0 = (int)s;
  L.i1 = L.i0 + (L.i0 < n_in - 1 ? 1 : 0);
  L.l1 = s - (float)L.i0;
  L.l0 = 1.0f - L.l1;
  return L;
}
__device__ __forceinline__ float cds_lerp(float a, float la, float b, float lb) { return fmaf(a, la, b * lb); }

__device__ __forceinline__ float cds_upsampled_depth(const float* __restrict__ prev, int hp, int wp, int Y, int X,
                                                     float sy, float sx) {
  Lerp ly = cds_lerp_index(Y, hp, sy), lx = cds_lerp_index(X, wp, sx);
  float top = cds_lerp(prev[ly.i0 * wp + lx.i0], lx.l0, prev[ly.i0 * wp + lx.i1], lx.l1);
  float bot = cds_lerp(prev[ly.i1 * wp + lx.i0], lx.l0, prev[ly.i1 * wp + lx.i1], lx.l1);
  return cds_lerp(top, ly.l0, bot, ly.l1);
}

__device__ __forceinline__ float cds_clamped_sample(float first, float k, float interval, float dmin, float dmax) {
  float s = first + k * interval;      // arange*interval rounded, then added (module.py:407-411)
  float a = dmin + fmaxf(s - dmin, 0.f);  // module.py:413-414
  return dmax + fminf(a - dmax, 0.f);     // module.py:415-416
}

__global__ __launch_bounds__(256) void depth_hypotheses_kernel(const float* __restrict__ prev, float* __restrict__ out,
                                                               int D, int hp, int wp, int H, int W, int h, int w,
                                                               const float* __restrict__ interval_d,
                                                               const float* __restrict__ range_d) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= h * w) return;
  const float interval = interval_d[0], dmin = range_d[0], dmax = range_d[1];   // the call's geometry block (device scalars)
  int y = p / w, x = p % w;
  const float up_sy = (float)hp / (float)H, up_sx = (float)wp / (float)W;
  const float dn_sy = (float)H / (float)h, dn_sx = (float)W / (float)w;
  Lerp ry = cds_lerp_index(y, H, dn_sy), rx = cds_lerp_index(x, W, dn_sx);
  const float nl = (float)((D - 1) / 2);
  const float back = nl * interval;  // rounded product (module.py:400)
  float f00 = cds_upsampled_depth(prev, hp, wp, ry.i0, rx.i0, up_sy, up_sx) - back;
  float f01 = cds_upsampled_depth(prev, hp, wp, ry.i0, rx.i1, up_sy, up_sx) - back;
  float f10 = cds_upsampled_depth(prev, hp, wp, ry.i1, rx.i0, up_sy, up_sx) - back;
  float f11 = cds_upsampled_depth(prev, hp, wp, ry.i1, rx.i1, up_sy, up_sx) - back;
  const size_t hw = (size_t)h * w;
  for (int k = 0; k < D; ++k) {
    float kf = (float)k;
    float a = cds_clamped_sample(f00, kf, interval, dmin, dmax);
    float b = cds_clamped_sample(f01, kf, interval, dmin, dmax);
    float c = cds_clamped_sample(f10, kf, interval, dmin, dmax);
    float e = cds_clamped_sample(f11, kf, interval, dmin, dmax);
    float top = cds_lerp(a, rx.l0, b, rx.l1);
    float bot = cds_lerp(c, rx.l0, e, rx.l1);
    out[(size_t)k * hw + p] = cds_lerp(top, ry.l0, bot, ry.l1);  // depth axis: weight (1,0), exact
  }
}

__global__ void depth_planes_kernel(float* __restrict__ out, int D, size_t hw, const float* __restrict__ range_d) {
  const float lo = range_d[0];
  const float step = (range_d[1] - lo) / (float)(D - 1);
  size_t n = (size_t)D * hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float k = (float)(i / hw);
    out[i] = lo + k * step;
  }
}

extern "C" int cds_softargmin_conf_f32(const float* prob_pre, const float* hyp, float* depth, float* conf, float* prob,
                                       int D, int h, int w, int hyp_per_pixel, void* stream) {
  if (!prob_pre || !hyp || !depth || !conf || D < 1 || h < 1 || w < 1) return CDS_EINVAL;
  int hw = h * w;
  hipLaunchKernelGGL(softargmin_conf_kernel, dim3(cds_ceil_div(hw, 64)), dim3(256), 0, (hipStream_t)stream, prob_pre,
                     hyp, depth, conf, prob, D, hw, hyp_per_pixel);
  return cds_launch_status();
}

extern "C" int cds_depth_hypotheses_f32(const float* prev_depth, float* out, int D, int hp, int wp, int H, int W,
                                        int scale, const float* interval, const float* depth_range, void* stream) {
  if (!prev_depth || !out || !interval || !depth_range || D < 1 || hp < 1 || wp < 1 || H < 1 || W < 1 || scale < 1 || (H % scale) || (W % scale))
    return CDS_EINVAL;
  int h = H / scale, w = W / scale;
  hipLaunchKernelGGL(depth_hypotheses_kernel, dim3(cds_ceil_div(h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                     prev_depth, out, D, hp, wp, H, W, h, w, interval, depth_range);
  return cds_launch_status();
}

extern "C" int cds_depth_planes_f32(float* out, int D, int h, int w, const float* depth_range, void* stream) {
  if (!out || !depth_range || D < 2 || h < 1 || w < 1) return CDS_EINVAL;
  size_t n = (size_t)D * h * w;
  int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(depth_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, D, (size_t)h * w,
                     depth_range);
  return cds_launch_status();
}
