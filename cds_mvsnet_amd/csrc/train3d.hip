// Training kernels of the CostRegNet stack (reference: models/module.py:80-160 Conv3d / Deconv3d with
// BatchNorm3d in training mode, trainer/trainer.py:78-82).  The forward convolutions and the data gradients reuse the
// inference kernels (cds_conv3d_k3_f32 / cds_deconv3d_k3s2_f32: the data gradient of a convolution is the transposed
// convolution with the same weights and vice versa; a stride-1 convolution's is the convolution with flipped, transposed
// weights).  New here:
//   cds_bn3d_stats_f32        per-channel sum / sum of squares over (batch, voxels), fp64 accumulation
//   cds_bn3d_norm_f32         per-channel step + y -> relu(y * scale[c] + shift[c]) (+ residual): BatchNorm(train) + ReLU + U-Net skip, fused
//   cds_bn3d_bwd_reduce_f32   sum g and sum g*y per channel, g = dout * [relu argument > 0]
//   cds_bn3d_bwd_norm_f32     dgamma, dbeta, dy = g * scale[c] + y * k1[c] + k0[c]   (the BatchNorm backward in closed form)
//   cds_conv3d_wgrad_f32      dw[a][b][tap] = sum_o g[a][o] * xin[b][S*o - 1 + tap]   (conv: g = dy, xin = x, S = stride;
//                             transposed conv: g = x, xin = dy, S = 2: the same sum with the roles swapped)
// Layouts: activations [B][C][D][H][W] fp32 planar (PyTorch's), statistics fp64 [C].
#include <stdlib.h>

#include "cds_common.hpp"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// grid (chunks, C, B); sums [C][2] fp64 (zeroed by the caller)
// V a multiple of 4 and every pointer 16-byte aligned: the elementwise / reduction loops below run on float4
__device__ __forceinline__ bool bn_vec4(size_t V, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
  return ((V & 3) | ((uintptr_t)a & 15) | ((uintptr_t)b & 15) | ((uintptr_t)c & 15) | ((uintptr_t)d & 15)) == 0;
}

__global__ __launch_bounds__(256) void bn3d_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int C, size_t V) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* __restrict__ p = x + ((size_t)b * C + c) * V;
  double s = 0.0, q = 0.0;
  if (bn_vec4(V, p)) {               // 16-byte loads, two in flight per thread (the scalar loop is one 4-byte load per memory round trip)
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
    const size_t V4 = V >> 2, step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step < V4; i += 2 * step) {
      const float4 a = p4[i], b4 = p4[i + step];
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w) + ((double)b4.x + (double)b4.y) + ((double)b4.z + (double)b4.w);
      q += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w) + ((double)b4.x * b4.x + (double)b4.y * b4.y) +
           ((double)b4.z * b4.z + (double)b4.w * b4.w);
    }
    if (i < V4) {
      const float4 a = p4[i];
      s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
      q += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
      const double v = p[i];
      s += v;
      q += v * v;
    }
  }
  __shared__ double red[2][4];
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[2 * c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[2 * c + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// BatchNorm (training) normalisation pass: every workgroup redoes the per-channel step from the fp64 sums (sum y, sum y^2) of the
// statistics pass - mean, biased variance, scale = gamma invstd, shift = beta - mean scale: a dozen fp64 operations - and applies
// out = relu(y * scale[c] + shift[c]) (+ skip); the first workgroup of a channel also writes scale / shift / mean / invstd for the
// backward and updates the running statistics.  (The per-channel step used to be its own one-workgroup launch between the two passes:
// 2 x 44 launches of a launch-bound training step.)  grid (chunks, C, B)
__global__ __launch_bounds__(256) void bn3d_norm_kernel(const float* __restrict__ y, const double* __restrict__ sums,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, double n,
                                                        double eps, float momentum, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, const float* __restrict__ skip,
                                                        float* __restrict__ out, float* __restrict__ scale_out,
                                                        float* __restrict__ shift_out, double* __restrict__ mean_out,
                                                        double* __restrict__ invstd_out, int C, size_t V, int relu) {
  const int c = blockIdx.y, b = blockIdx.z;
  const double mean = sums[2 * c] / n;
  double var = sums[2 * c + 1] / n - mean * mean;                  // biased, like F.batch_norm in training
  var = var < 0.0 ? 0.0 : var;
  const double invstd = 1.0 / sqrt(var + eps);
  const double gm = (double)gamma[c];
  const float sc = (float)(gm * invstd), sh = (float)((double)beta[c] - mean * gm * invstd);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    scale_out[c] = sc;
    shift_out[c] = sh;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    if (running_mean) {
      running_mean[c] = running_mean[c] * (1.0f - momentum) + momentum * (float)mean;
      running_var[c] = running_var[c] * (1.0f - momentum) + momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
    }
  }
  const size_t base = ((size_t)b * C + c) * V;
  if (bn_vec4(V, y + base, out + base, skip ? skip + base : nullptr)) {
    const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y + base);
    const float4* __restrict__ k4 = skip ? reinterpret_cast<const float4*>(skip + base) : nullptr;
    float4* __restrict__ o4 = reinterpret_cast<float4*>(out + base);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (V >> 2); i += (size_t)gridDim.x * 256) {
      const float4 a = y4[i];
      float4 v = make_float4(fmaf(a.x, sc, sh), fmaf(a.y, sc, sh), fmaf(a.z, sc, sh), fmaf(a.w, sc, sh));
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (k4) {
        const float4 k = k4[i];
        v = make_float4(k.x + v.x, k.y + v.y, k.z + v.z, k.w + v.w);
      }
      o4[i] = v;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
    float v = fmaf(y[base + i], sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    if (skip) v = skip[base + i] + v;
    out[base + i] = v;
  }
}

// sums[c] = (sum g, sum g * y), g = dout * [y * scale + shift > 0]
__global__ __launch_bounds__(256) void bn3d_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              double* __restrict__ sums, int C, size_t V, int relu) {
  const int c = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * C + c) * V;
  const float sc = scale[c], sh = shift[c];
  double s = 0.0, q = 0.0;
  if (bn_vec4(V, y + base, dout + base)) {
    const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y + base);
    const float4* __restrict__ d4 = reinterpret_cast<const float4*>(dout + base);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (V >> 2); i += (size_t)gridDim.x * 256) {
      const float4 yv = y4[i], dv = d4[i];
      const float g0 = (!relu || fmaf(yv.x, sc, sh) > 0.f) ? dv.x : 0.f, g1 = (!relu || fmaf(yv.y, sc, sh) > 0.f) ? dv.y : 0.f;
      const float g2 = (!relu || fmaf(yv.z, sc, sh) > 0.f) ? dv.z : 0.f, g3 = (!relu || fmaf(yv.w, sc, sh) > 0.f) ? dv.w : 0.f;
      s += ((double)g0 + (double)g1) + ((double)g2 + (double)g3);
      q += ((double)g0 * (double)yv.x + (double)g1 * (double)yv.y) + ((double)g2 * (double)yv.z + (double)g3 * (double)yv.w);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
      const float yv = y[base + i];
      const float g = (!relu || fmaf(yv, sc, sh) > 0.f) ? dout[base + i] : 0.f;
      s += (double)g;
      q += (double)g * (double)yv;
    }
  }
  __shared__ double red[2][4];
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[2 * c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[2 * c + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// BatchNorm backward, second pass: from the fp64 sums (sum g, sum g y) of the reduction pass every workgroup derives dgamma, dbeta
// and the (k1, k0) of dy = g scale[c] + y k1[c] + k0[c] itself; the first workgroup of a channel writes dgamma / dbeta.
__global__ __launch_bounds__(256) void bn3d_bwd_norm_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const double* __restrict__ sums, const double* __restrict__ mean,
                                                            const double* __restrict__ invstd, double n, float* __restrict__ dy,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int C, size_t V,
                                                            int relu) {
  const int c = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * C + c) * V;
  const float sc = scale[c], sh = shift[c];
  const double db = sums[2 * c];
  const double dg = invstd[c] * (sums[2 * c + 1] - mean[c] * sums[2 * c]);      // sum g * xhat
  const double scd = (double)sc;
  const float a1 = (float)(-scd * dg * invstd[c] / n), a0 = (float)(-scd * db / n + scd * dg * invstd[c] * mean[c] / n);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
  }
  if (bn_vec4(V, y + base, dout + base, dy + base)) {
    const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y + base);
    const float4* __restrict__ d4 = reinterpret_cast<const float4*>(dout + base);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(dy + base);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (V >> 2); i += (size_t)gridDim.x * 256) {
      const float4 yv = y4[i], dv = d4[i];
      const float g0 = (!relu || fmaf(yv.x, sc, sh) > 0.f) ? dv.x : 0.f, g1 = (!relu || fmaf(yv.y, sc, sh) > 0.f) ? dv.y : 0.f;
      const float g2 = (!relu || fmaf(yv.z, sc, sh) > 0.f) ? dv.z : 0.f, g3 = (!relu || fmaf(yv.w, sc, sh) > 0.f) ? dv.w : 0.f;
      o4[i] = make_float4(fmaf(g0, sc, fmaf(yv.x, a1, a0)), fmaf(g1, sc, fmaf(yv.y, a1, a0)), fmaf(g2, sc, fmaf(yv.z, a1, a0)),
                          fmaf(g3, sc, fmaf(yv.w, a1, a0)));
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
    const float yv = y[base + i];
    const float g = (!relu || fmaf(yv, sc, sh) > 0.f) ? dout[base + i] : 0.f;
    dy[base + i] = fmaf(g, sc, fmaf(yv, a1, a0));
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  dw[a][b][tap] += sum over the o-tiles of this workgroup of g[a][o] * xin[b][S o - 1 + tap].
// Workgroup = (group of o-tiles) x (8 a-channels) x (8 b-channels); thread = (a, b, tap group of 7): 7 accumulators.
// o-tile 8 x 4 x 4 voxels of g; the matching xin tile ((8-1) S + 3) x ((4-1) S + 3)^2 with zero padding, both in LDS.
// ---------------------------------------------------------------------------------------------
template <int S>
struct WCfg {
  static constexpr int OX = 8, OY = 4, OZ = 4, NO = OX * OY * OZ;
  static constexpr int IX = (OX - 1) * S + 3, IY = (OY - 1) * S + 3, IZ = (OZ - 1) * S + 3;
  static constexpr int NI = IX * IY * IZ;
#ifdef CDS_WGRAD3D_NOPAD
  static constexpr int GS = NO;
#else
  static constexpr int GS = NO + 4;      // row stride of the g tile in the MFMA kernel: the 16 channel rows a K-step reads fall on different banks
#endif
};

template <int S>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ xin,
                                                           float* __restrict__ dw, int B, int Ca, int Cb, int Do, int Ho, int Wo,
                                                           int Di, int Hi, int Wi, int tiles_x, int tiles_y, int ntiles,
                                                           int tiles_per_wg) {
  using Cfg = WCfg<S>;
  __shared__ float lg[8][Cfg::NO];
  __shared__ float lx[8][Cfg::NI];
  const int tid = threadIdx.x;
  const int a = tid >> 5, b = (tid >> 2) & 7, kg = tid & 3;       // tap group kg: taps 7 kg .. 7 kg + 6 (27 taps, last group 6)
  const int a0 = blockIdx.y * 8, b0 = blockIdx.z * 8;
  const size_t vo = (size_t)Do * Ho * Wo, vi = (size_t)Di * Hi * Wi;
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int k = min(7 * kg + t, 26);
    toff[t] = ((k / 9) * Cfg::IY + (k / 3) % 3) * Cfg::IX + k % 3;
  }
  const int t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles * B, t0 + tiles_per_wg);
  for (int tile = t0; tile < t1; ++tile) {
    const int bi = tile / ntiles;
    int r = tile - bi * ntiles;
    const int tx_i = r % tiles_x;
    r /= tiles_x;
    const int ty_i = r % tiles_y, tz_i = r / tiles_y;
    const int ox0 = tx_i * Cfg::OX, oy0 = ty_i * Cfg::OY, oz0 = tz_i * Cfg::OZ;
    __syncthreads();
    for (int i = tid; i < 8 * Cfg::NO; i += 256) {
      const int ch = i / Cfg::NO, p = i - ch * Cfg::NO;
      const int px = p % Cfg::OX, py = (p / Cfg::OX) % Cfg::OY, pz = p / (Cfg::OX * Cfg::OY);
      const int ox = ox0 + px, oy = oy0 + py, oz = oz0 + pz;
      const bool ok = a0 + ch < Ca && ox < Wo && oy < Ho && oz < Do;
      lg[ch][p] = ok ? g[((size_t)bi * Ca + a0 + ch) * vo + ((size_t)oz * Ho + oy) * Wo + ox] : 0.f;
    }
    for (int i = tid; i < 8 * Cfg::NI; i += 256) {
      const int ch = i / Cfg::NI, p = i - ch * Cfg::NI;
      const int px = p % Cfg::IX, py = (p / Cfg::IX) % Cfg::IY, pz = p / (Cfg::IX * Cfg::IY);
      const int ix = ox0 * S - 1 + px, iy = oy0 * S - 1 + py, iz = oz0 * S - 1 + pz;
      const bool ok = b0 + ch < Cb && (unsigned)ix < (unsigned)Wi && (unsigned)iy < (unsigned)Hi && (unsigned)iz < (unsigned)Di;
      lx[ch][p] = ok ? xin[((size_t)bi * Cb + b0 + ch) * vi + ((size_t)iz * Hi + iy) * Wi + ix] : 0.f;
    }
    __syncthreads();
    const float* __restrict__ ga = lg[a];
    const float* __restrict__ xb = lx[b];
#pragma unroll 4
    for (int p = 0; p < Cfg::NO; ++p) {
      const int px = p % Cfg::OX, py = (p / Cfg::OX) % Cfg::OY, pz = p / (Cfg::OX * Cfg::OY);
      const float gv = ga[p];
      const int base = ((pz * S) * Cfg::IY + py * S) * Cfg::IX + px * S;
#pragma unroll
      for (int t = 0; t < 7; ++t) acc[t] = fmaf(gv, xb[base + toff[t]], acc[t]);
    }
  }
  if (a0 + a < Ca && b0 + b < Cb) {
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int k = 7 * kg + t;
      if (k < 27) atomicAdd(&dw[((size_t)(a0 + a) * Cb + b0 + b) * 27 + k], acc[t]);
    }
  }
}

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

// The K-steps of one staged tile for a wave that owns NQ column blocks (branch-free: NQ is a template parameter).
template <typename Cfg, int S, int NQ>
__device__ __forceinline__ void wgrad3d_ksteps(const float* __restrict__ lg, const float* __restrict__ lx, int j, int kq, const int* colofs,
                                               wg_f32x4* acc) {
#pragma unroll 4
  for (int ks = 0; ks < Cfg::NO / 4; ++ks) {
    const int p = 4 * ks + kq;                         // this lane's voxel of the K-step
    const int px = p % Cfg::OX, py = (p / Cfg::OX) % Cfg::OY, pz = p / (Cfg::OX * Cfg::OY);
    const int base = ((pz * S) * Cfg::IY + py * S) * Cfg::IX + px * S;
    const float av = lg[j * Cfg::GS + p];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, lx[base + colofs[q]], acc[q], 0, 0, 0);
  }
}

// The same sums on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation):
//   D[a][n] += A[a][k] B[k][n],  a = 16 channels of g, n = (b, tap) column (8 b-channels x 27 taps = 216 of 224), k = voxel.
// A lane reads ONE float of g and one float of the xin tile per MFMA (the VALU kernel above read 8 per 7 FMAs and was
// LDS-bound at ~12 TFLOP/s); the four waves own 3-4 of the 14 column blocks each and all walk the tile's 128 voxels.
template <int S>
__global__ __launch_bounds__(256) void conv3d_wgrad_mfma_kernel(const float* __restrict__ g, const float* __restrict__ xin,
                                                                float* __restrict__ dw, int B, int Ca, int Cb, int Do, int Ho,
                                                                int Wo, int Di, int Hi, int Wi, int tiles_x, int tiles_y, int ntiles,
                                                                int tiles_per_wg) {
  using Cfg = WCfg<S>;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  float* lg = wlds;                       // [16][GS]
  float* lx = wlds + 16 * Cfg::GS;        // [8][NI]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int a0 = blockIdx.y * 16, b0 = blockIdx.z * 8;
  const size_t vo = (size_t)Do * Ho * Wo, vi = (size_t)Di * Hi * Wi;
  constexpr int NBLK = 14, QW = 4;        // column blocks; per wave: blocks wave, wave + 4, ...
  int colofs[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int n = min((wave + 4 * q) * 16 + j, 8 * 27 - 1);
    const int b = n / 27, tap = n - b * 27;
    colofs[q] = b * Cfg::NI + ((tap / 9) * Cfg::IY + (tap / 3) % 3) * Cfg::IX + tap % 3;
  }
  f32x4 acc[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles * B, t0 + tiles_per_wg);
  // Staging through registers: all loads of a tile are issued back to back (unconditional loads from clamped addresses, zeros selected
  // afterwards: a guarded load inside a loop compiled to ONE load per iteration followed by s_waitcnt vmcnt(0), ~20 serial round trips
  // per tile), and the loads of tile t + 1 are issued before the K-steps of tile t.
  constexpr int NG = (16 * Cfg::NO + 255) / 256, NX = (8 * Cfg::NI + 255) / 256;
  float rg[NG], rx[NX];
  // per element of this thread, fixed for all tiles: offset from the tile's origin and (x, y, z) inside the tile for the bounds tests
  // (the div / mod chains per element and tile were ~1 200 VALU instructions per thread and tile: more than the K-steps' matrix time)
  int relg[NG], pkg[NG], relx[NX], pkx[NX];
#pragma unroll
  for (int e = 0; e < NG; ++e) {
    const int i = tid + 256 * e;
    const int ch = i / Cfg::NO, p = i - ch * Cfg::NO;
    const int px = p % Cfg::OX, py = (p / Cfg::OX) % Cfg::OY, pz = p / (Cfg::OX * Cfg::OY);
    relg[e] = ch * (int)vo + (pz * Ho + py) * Wo + px;
    pkg[e] = (i < 16 * Cfg::NO && a0 + ch < Ca) ? (px | (py << 8) | (pz << 16)) : -1;
  }
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    const int i = tid + 256 * e;
    const int ch = i / Cfg::NI, p = i - ch * Cfg::NI;
    const int px = p % Cfg::IX, py = (p / Cfg::IX) % Cfg::IY, pz = p / (Cfg::IX * Cfg::IY);
    relx[e] = ch * (int)vi + (pz * Hi + py) * Wi + px;
    pkx[e] = (i < 8 * Cfg::NI && b0 + ch < Cb) ? (px | (py << 8) | (pz << 16)) : -1;
  }
  auto fetch = [&](int tile) {
    const int bi = tile / ntiles;
    int r = tile - bi * ntiles;
    const int tx_i = r % tiles_x;
    r /= tiles_x;
    const int ty_i = r % tiles_y, tz_i = r / tiles_y;
    const int ox0 = tx_i * Cfg::OX, oy0 = ty_i * Cfg::OY, oz0 = tz_i * Cfg::OZ;
    const float* __restrict__ gt = g + ((size_t)bi * Ca + a0) * vo + ((size_t)oz0 * Ho + oy0) * Wo + ox0;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
    const long long xbase = (long long)(((size_t)bi * Cb + b0) * vi) + ((long long)iz0 * Hi + iy0) * Wi + ix0;
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int pk = pkg[e];
      const bool ok = pk >= 0 && ox0 + (pk & 255) < Wo && oy0 + ((pk >> 8) & 255) < Ho && oz0 + (pk >> 16) < Do;
      const float v = gt[ok ? relg[e] : 0];
      rg[e] = ok ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int pk = pkx[e];
      const bool ok = pk >= 0 && (unsigned)(ix0 + (pk & 255)) < (unsigned)Wi && (unsigned)(iy0 + ((pk >> 8) & 255)) < (unsigned)Hi &&
                      (unsigned)(iz0 + (pk >> 16)) < (unsigned)Di;
      const float v = xin[ok ? xbase + relx[e] : 0];
      rx[e] = ok ? v : 0.f;
    }
  };
  if (t0 < t1) fetch(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const int i = tid + 256 * e;
      if (i < 16 * Cfg::NO) lg[(i / Cfg::NO) * Cfg::GS + i % Cfg::NO] = rg[e];
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int i = tid + 256 * e;
      if (i < 8 * Cfg::NI) lx[i] = rx[e];
    }
    __syncthreads();
    if (tile + 1 < t1) fetch(tile + 1);
    // waves 0 / 1 own four column blocks, waves 2 / 3 three: decided ONCE per tile.  With the test inside the K loop the compiler
    // wrapped every MFMA in a branch and copied the accumulators in and out of the AGPRs around it (s_nop + v_accvgpr_read after
    // each MFMA: the matrix pipe ran at 14 %)
    if (wave + 4 * (QW - 1) < NBLK)
      wgrad3d_ksteps<Cfg, S, QW>(lg, lx, j, kq, colofs, acc);
    else
      wgrad3d_ksteps<Cfg, S, QW - 1>(lg, lx, j, kq, colofs, acc);
  }
  // D: lane holds rows a = 4 kq + 0..3 of column n = 16 (wave + 4 q) + j
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int n = (wave + 4 * q) * 16 + j;
    if (wave + 4 * q >= NBLK || n >= 8 * 27) continue;
    const int b = n / 27, tap = n - b * 27;
    if (b0 + b >= Cb) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = a0 + 4 * kq + r;
      if (a < Ca) atomicAdd(&dw[((size_t)a * Cb + b0 + b) * 27 + tap], r == 0 ? acc[q].x : r == 1 ? acc[q].y : r == 2 ? acc[q].z : acc[q].w);
    }
  }
}

inline dim3 ew_grid(size_t V, int C, int B, int cap = 512) {
  size_t chunks = (V + 256 * 8 - 1) / (256 * 8);
  if (chunks > (size_t)cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  return dim3((unsigned)chunks, C, B);
}

}  // namespace

extern "C" int cds_bn3d_stats_f32(const float* x, double* sums, int B, int C, long long V, void* stream) {
  if (!x || !sums || B < 1 || C < 1 || V < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(bn3d_stats_kernel, ew_grid((size_t)V, C, B, cds_env_int("CDS_BN_RED_CHUNKS", 128)), dim3(256), 0, (hipStream_t)stream, x, sums, C, (size_t)V);
  return cds_launch_status();
}

extern "C" int cds_bn3d_norm_f32(const float* y, const double* sums, const float* gamma, const float* beta, double n, double eps,
                                 float momentum, float* running_mean, float* running_var, const float* skip, float* out, float* scale,
                                 float* shift, double* mean, double* invstd, int B, int C, long long V, int relu, void* stream) {
  if (!y || !sums || !gamma || !beta || !out || !scale || !shift || !mean || !invstd || B < 1 || C < 1 || V < 1 || n < 1.0 ||
      (running_mean != nullptr) != (running_var != nullptr))
    return CDS_EINVAL;
  hipLaunchKernelGGL(bn3d_norm_kernel, ew_grid((size_t)V, C, B), dim3(256), 0, (hipStream_t)stream, y, sums, gamma, beta, n, eps,
                     momentum, running_mean, running_var, skip, out, scale, shift, mean, invstd, C, (size_t)V, relu);
  return cds_launch_status();
}

extern "C" int cds_bn3d_bwd_reduce_f32(const float* dout, const float* y, const float* scale, const float* shift, double* sums,
                                       int B, int C, long long V, int relu, void* stream) {
  if (!dout || !y || !scale || !shift || !sums || B < 1 || C < 1 || V < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(bn3d_bwd_reduce_kernel, ew_grid((size_t)V, C, B, cds_env_int("CDS_BN_RED_CHUNKS", 128)), dim3(256), 0, (hipStream_t)stream, dout, y, scale, shift,
                     sums, C, (size_t)V, relu);
  return cds_launch_status();
}

extern "C" int cds_bn3d_bwd_norm_f32(const float* dout, const float* y, const float* scale, const float* shift, const double* sums,
                                     const double* mean, const double* invstd, double n, float* dy, float* dgamma, float* dbeta, int B,
                                     int C, long long V, int relu, void* stream) {
  if (!dout || !y || !scale || !shift || !sums || !mean || !invstd || !dy || !dgamma || !dbeta || B < 1 || C < 1 || V < 1 || n < 1.0)
    return CDS_EINVAL;
  hipLaunchKernelGGL(bn3d_bwd_norm_kernel, ew_grid((size_t)V, C, B), dim3(256), 0, (hipStream_t)stream, dout, y, scale, shift, sums,
                     mean, invstd, n, dy, dgamma, dbeta, C, (size_t)V, relu);
  return cds_launch_status();
}

// dw [Ca][Cb][27] (accumulated onto: zero it first).  g [B][Ca][Do][Ho][Wo], xin [B][Cb][Di][Hi][Wi], stride S in {1, 2},
// pad 1: dw[a][b][(kz*3+ky)*3+kx] += sum_{b', o} g[a][o] * xin[b][S o - 1 + k].
extern "C" int cds_conv3d_wgrad_f32(const float* g, const float* xin, float* dw, int B, int Ca, int Cb, int Do, int Ho, int Wo,
                                    int Di, int Hi, int Wi, int stride, void* stream) {
  if (!g || !xin || !dw || B < 1 || Ca < 1 || Cb < 1 || Do < 1 || Ho < 1 || Wo < 1 || (stride != 1 && stride != 2)) return CDS_EINVAL;
  const int tx = cds_ceil_div(Wo, 8), ty = cds_ceil_div(Ho, 4), tz = cds_ceil_div(Do, 4);
  const int ntiles = tx * ty * tz;
  // tiles per workgroup: every workgroup ends with one atomic per weight of its (16 x 8-channel) block, all workgroups on the same few
  // thousand addresses (a quarter of the kernel's time at ~1024 tile groups per channel block), so aim at CDS_WG3_WGS workgroups in ALL
  int per = cds_ceil_div(ntiles * B * cds_ceil_div(Ca, 16) * cds_ceil_div(Cb, 8), cds_env_int("CDS_WG3_WGS", 512));
  if (per < 1) per = 1;
  const bool valu = cds_env_set("CDS_WGRAD_VALU");   // A/B knob: the VALU kernel
  if (!valu) {
    const dim3 gm(cds_ceil_div(ntiles * B, per), cds_ceil_div(Ca, 16), cds_ceil_div(Cb, 8));
    if (stride == 1) {
      const size_t ldsb = (16 * WCfg<1>::GS + 8 * WCfg<1>::NI) * sizeof(float);
      hipLaunchKernelGGL(conv3d_wgrad_mfma_kernel<1>, gm, dim3(256), ldsb, (hipStream_t)stream, g, xin, dw, B, Ca, Cb, Do, Ho, Wo, Di,
                         Hi, Wi, tx, ty, ntiles, per);
    } else {
      const size_t ldsb = (16 * WCfg<2>::GS + 8 * WCfg<2>::NI) * sizeof(float);
      hipLaunchKernelGGL(conv3d_wgrad_mfma_kernel<2>, gm, dim3(256), ldsb, (hipStream_t)stream, g, xin, dw, B, Ca, Cb, Do, Ho, Wo, Di,
                         Hi, Wi, tx, ty, ntiles, per);
    }
    return cds_launch_status();
  }
  const dim3 grid(cds_ceil_div(ntiles * B, per), cds_ceil_div(Ca, 8), cds_ceil_div(Cb, 8));
  if (stride == 1)
    hipLaunchKernelGGL(conv3d_wgrad_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, g, xin, dw, B, Ca, Cb, Do, Ho, Wo, Di, Hi, Wi,
                       tx, ty, ntiles, per);
  else
    hipLaunchKernelGGL(conv3d_wgrad_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, g, xin, dw, B, Ca, Cb, Do, Ho, Wo, Di, Hi, Wi,
                       tx, ty, ntiles, per);
  return cds_launch_status();
}


// Running statistics of a BatchNorm whose call stacked G groups (train2d_ops.bn_relu2d): r = keep r + sum_g wts[g] stat[g][c] for the mean
// and the variance in one launch (the groups in call order; wts[g] = m (1 - m)^(G-1-g), keep = (1 - m)^G).
namespace {
__global__ void bn_running_update_kernel(const float* __restrict__ tm, const float* __restrict__ tv, const float* __restrict__ wts,
                                         float keep, int G, int C, float* __restrict__ rmean, float* __restrict__ rvar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C) return;
  const int c = i < C ? i : i - C;
  const float* __restrict__ st = i < C ? tm : tv;
  float* __restrict__ r = i < C ? rmean : rvar;
  float acc = 0.f;
  for (int g = 0; g < G; ++g) acc = fmaf(wts[g], st[g * C + c], acc);
  r[c] = keep * r[c] + acc;
}
}  // namespace

extern "C" int cds_bn_running_update_f32(const float* batch_mean, const float* batch_var, const float* group_weights, float keep, int G,
                                         int C, float* running_mean, float* running_var, void* stream) {
  if (!batch_mean || !batch_var || !group_weights || !running_mean || !running_var || G < 1 || C < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(cds_ceil_div(2 * C, 128)), dim3(128), 0, (hipStream_t)stream, batch_mean, batch_var,
                     group_weights, keep, G, C, running_mean, running_var);
  return cds_launch_status();
}
