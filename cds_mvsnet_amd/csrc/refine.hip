// Refinement network pieces that are not plain 3x3 convolutions (module.py:318-370):
//   cds_depth_affine_f32     d = (depth / ival - lo) / (hi - lo) * 10,  lo = depth_min / ival, hi = depth_max / ival   (model.py:213-216, module.py:353-355)
//   cds_deconv2d_k3s2_f32    ConvTranspose2d k3 s2 p1 op1 + folded BN + activation    (module.py:331-335,359)
//   cds_refine_finish_f32    (((bilinear x2, align_corners=True)(d) + res) / 10 * (hi - lo) + lo) * ival   (module.py:366-368, model.py:218)
// (depth_min, depth_max, ival) are DEVICE scalars of the call's geometry block; ival = 1 gives the network of module.py on its own)
// The 3x3 Conv+BN+ReLU units run on cds_conv2d_f32 (conv2d.hip).  All of it is a few hundred microseconds of
// HBM-bound work at 640x512; the kernels are written for clarity, one thread per input cell / output pixel.
#include "cds_common.hpp"

namespace {

__global__ __launch_bounds__(256) void depth_affine_kernel(const float* __restrict__ x, float* __restrict__ out, int n,
                                                           const float* __restrict__ range_d) {
  // range_d = (depth_min, depth_max, depth_interval) in DEVICE memory (the call's geometry block): models/model.py:213-216 divides the
  // depth and both limits by the interval (true divisions), module.py:353-355 normalises
  const float ival = range_d[2];
  const float lo = range_d[0] / ival, hi = range_d[1] / ival;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (x[i] / ival - lo) / (hi - lo) * 10.0f;
}

// thread = one input cell (y, x) -> the 2x2 output block (2y.., 2x..) for CO output channels.
// out[2y-1+ky][2x-1+kx] += in[y][x] * w[ci][ky][kx][co]  =>  with i00 = in[y][x], i01 = in[y][x+1], i10 = in[y+1][x]:
//   o(2y  ,2x  ) = i00 w11
//   o(2y  ,2x+1) = i01 w10 + i00 w12
//   o(2y+1,2x  ) = i10 w01 + i00 w21
//   o(2y+1,2x+1) = i11 w00 + i10 w02 + i01 w20 + i00 w22
template <int CO>
__global__ __launch_bounds__(256) void deconv2d_k3s2_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Cin, int H, int W, int act) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, xx = p - y * W;
  const bool xr = xx + 1 < W, yd = y + 1 < H;
  float acc[4][CO];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[q][c] = 0.f;
  const size_t plane = (size_t)H * W;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* __restrict__ xc = x + (size_t)ci * plane + p;
    const float i00 = xc[0];
    const float i01 = xr ? xc[1] : 0.f;
    const float i10 = yd ? xc[W] : 0.f;
    const float i11 = (xr && yd) ? xc[W + 1] : 0.f;
    const float* __restrict__ w = wpk + (size_t)ci * 9 * CO;   // [ky*3+kx][co]
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      acc[0][c] = fmaf(i00, w[4 * CO + c], acc[0][c]);
      acc[1][c] = fmaf(i01, w[3 * CO + c], acc[1][c]);
      acc[1][c] = fmaf(i00, w[5 * CO + c], acc[1][c]);
      acc[2][c] = fmaf(i10, w[1 * CO + c], acc[2][c]);
      acc[2][c] = fmaf(i00, w[7 * CO + c], acc[2][c]);
      acc[3][c] = fmaf(i11, w[0 * CO + c], acc[3][c]);
      acc[3][c] = fmaf(i10, w[2 * CO + c], acc[3][c]);
      acc[3][c] = fmaf(i01, w[6 * CO + c], acc[3][c]);
      acc[3][c] = fmaf(i00, w[8 * CO + c], acc[3][c]);
    }
  }
  const int Wo = 2 * W;
  const size_t oplane = 4 * plane;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float b = bias ? bias[c] : 0.f;
    float* o = out + (size_t)c * oplane + (size_t)(2 * y) * Wo + 2 * xx;
    *reinterpret_cast<float2*>(o) = make_float2(cds_apply_act(acc[0][c] + b, act), cds_apply_act(acc[1][c] + b, act));
    *reinterpret_cast<float2*>(o + Wo) = make_float2(cds_apply_act(acc[2][c] + b, act), cds_apply_act(acc[3][c] + b, act));
  }
}

// bilinear x2 with align_corners=True: src = dst * (in - 1) / (out - 1); ATen's order l0*v0 + l1*v1 per axis
__global__ __launch_bounds__(256) void refine_finish_kernel(const float* __restrict__ d, const float* __restrict__ res,
                                                            float* __restrict__ out, int h, int w,
                                                            const float* __restrict__ range_d) {
  const float ival = range_d[2];
  const float lo = range_d[0] / ival, hi = range_d[1] / ival;
  const int H = 2 * h, W = 2 * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float fy = sy * (float)y, fx = sx * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
  const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
  const float top = lx0 * d[y0 * w + x0] + lx1 * d[y0 * w + x1];
  const float bot = lx0 * d[y1 * w + x0] + lx1 * d[y1 * w + x1];
  const float up = ly0 * top + ly1 * bot;
  const float v = (up + res[p]) / 10.0f;
  out[p] = (v * (hi - lo) + lo) * ival;     // module.py:368, then models/model.py:218
}

}  // namespace

extern "C" int cds_depth_affine_f32(const float* depth, float* out, int n, const float* depth_range, void* stream) {
  if (!depth || !out || !depth_range || n < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(depth_affine_kernel, dim3(cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, out, n,
                     depth_range);
  return cds_launch_status();
}

extern "C" int cds_deconv2d_k3s2_f32(const float* x, const float* weight, const float* bias, float* out, int Cin, int Cout,
                                     int H, int W, int act, void* stream) {
  if (!x || !weight || !out || Cin < 1 || Cout != 8 || H < 1 || W < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(deconv2d_k3s2_kernel<8>, dim3(cds_ceil_div(H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, weight,
                     bias, out, Cin, H, W, act);
  return cds_launch_status();
}

extern "C" int cds_refine_finish_f32(const float* d_norm, const float* res, float* out, int h, int w, const float* depth_range,
                                     void* stream) {
  if (!d_norm || !res || !out || !depth_range || h < 1 || w < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(refine_finish_kernel, dim3(cds_ceil_div(4 * h * w, 256)), dim3(256), 0, (hipStream_t)stream, d_norm,
                     res, out, h, w, depth_range);
  return cds_launch_status();
}
