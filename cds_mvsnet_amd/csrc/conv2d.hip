// 2D feature path:
//   cds_conv2d_f32 / cds_conv2d_affine_f32   direct LDS-tiled k x k convolution (k in 1,3,5,7,11; stride 1|2), optionally
//                          with the producing layer's InstanceNorm + LeakyReLU applied while the tile is loaded
//                          models/module.py:28-71, models/dynamic_conv.py:86-87,112,116, model.py:14
//       conv2d_kernel        row-per-wave staging (any width; the 7x7 / 11x11 layers, 1x1 straight from global memory)
//       conv2d_pipe_kernel   aligned 16-byte staging with register prefetch (3x3 / 5x5, W % 4 == 0)
//   cds_conv2d_fpn_f32     FPN lateral: 1x1 convolution over the virtual nearest-2x up-sample + concatenation
//                          models/module.py:253-254,260-261
//   cds_dynconv_blend*_f32 DynamicConv epilogue: epipolar curvature projection, 1x1 MLP, softmax(./T), blend; the
//                          _stats variant also leaves the InstanceNorm records of its output   models/dynamic_conv.py:97-122
//   cds_instnorm_*_f32     InstanceNorm2d (+LeakyReLU(0.1) | tanh): two-pass (act), statistics only (affine), fixed-order
//                          reduction of in-kernel records (reduce), normalisation for given statistics (apply)
//                          models/module.py:53,66-69,223
//
// Convolution scheme (same as conv3d.hip): input tile + halo of CI_CHUNK channels in LDS, PX
// x-adjacent outputs x 8 output channels of accumulators per thread, weights packed
// [Cin][k*k][CoutP] (cout fastest, CoutP = Cout rounded up to 8, zero padded) and fetched as
// wave-uniform scalar loads.
#include <stdlib.h>

#include "cds_common.hpp"
#include "feat_common.hpp"

#ifndef CDS_C2_CHUNK3
#define CDS_C2_CHUNK3 4     // input channels per staged chunk of the 3x3 layers (A/B: 2 / 4 / 8)
#endif
#ifndef CDS_C2_CHUNK3S2
#define CDS_C2_CHUNK3S2 2   // stride 2: 242 vs 254 us (8->16, 1600x1184), 176 vs 190 us (16->32, 800x592)
#endif

#ifdef CDS_CONV2D_NO_ACCUM          // A/B knob: the epilogues without the CDS_ACT_ACCUM path (gradient accumulation of train2d_ops.py)
#define C2_ACCUM 0
#else
#define C2_ACCUM CDS_ACT_ACCUM
#endif

namespace {

constexpr int CO = 8;

template <int K, int S, int PX, int CI_CHUNK>
struct C2Cfg {
  static constexpr int LX = 16, LY = 16;
  static constexpr int TX = LX * PX, TY = LY;
  static constexpr int IX = (TX - 1) * S + K, IY = (TY - 1) * S + K;
  static constexpr int IXP = (IX + 3) & ~3;
  static constexpr int TILE = IY * IXP;
  static constexpr int NIN = (PX - 1) * S + K;
};

// NCB = number of 8-wide output-channel blocks a workgroup produces from ONE staged input tile (the launcher only
// picks values that divide CoutP / 8, so every group is full width).  The DynamicConv branches have Cout + 3 =
// 11 / 19 / 35 output channels (2 / 3 / 5 blocks): with one block per workgroup the input tile is staged 2-5 times.
// K = 1 needs no halo and reads its inputs straight from global memory (the 1x1 convolutions ran at 1/7 of the HBM
// rate through the LDS path).
// CWE = output channels actually computed (<= 8 NCB): the DynamicConv widths 11 / 19 / 35 are not multiples of 8, and a
// workgroup that owns all of them skips the 5 zero-padded columns (31 / 21 / 12 % of the multiply-adds).
template <int K, int S, int PX, int CI_CHUNK, int NCB, int CWE = 8 * NCB>
__global__ __launch_bounds__(256) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ in_affine,
                                                     const float* __restrict__ wpk,
                                                     const float* __restrict__ bias, float* __restrict__ out, int N,
                                                     int Cin, int Cout, int CoutP, int H, int W, int Ho, int Wo, int pad,
                                                     int act, int tiles_x, int tiles_y) {
  using Cfg = C2Cfg<K, S, PX, CI_CHUNK>;
  constexpr int CW = CWE;        // output channels per workgroup
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_groups = CoutP / (CO * NCB);
  const int ntiles = tiles_x * tiles_y;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_groups * N);
  const int cog = lin % co_groups;
  lin /= co_groups;
  const int tile = lin % ntiles;
  const int n = lin / ntiles;
  const int tx_i = tile % tiles_x, ty_i = tile / tiles_x;
  const int co0 = cog * (CO * NCB);
  const int tid = threadIdx.x;
  const int lx = tid % Cfg::LX, ly = tid / Cfg::LX;
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY;
  const int gx0 = ox0 * S - pad, gy0 = oy0 * S - pad;
  const size_t plane = (size_t)H * W;
  const float* __restrict__ xn = x + (size_t)n * Cin * plane;
  // optional per-(image, channel) affine + leaky slope applied to every in-bounds input value on load: the lazily applied
  // InstanceNorm + LeakyReLU of the producing layer (zero padding stays zero, as padding follows the normalisation)
  const float* __restrict__ aff = in_affine ? in_affine + (size_t)n * Cin * 3 : nullptr;

  float acc[PX][CW];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[p][c] = 0.f;

  if constexpr (K == 1 && S == 1) {
    // pointwise (pad = 0): out[co][p] = sum_ci w[ci][co] * in[ci][p]
    const int oyp = oy0 + ly, oxp = ox0 + lx * PX;
    const bool rowok = oyp < H;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* __restrict__ src = xn + (size_t)ci * plane + (size_t)(rowok ? oyp : 0) * W;
      float in[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) in[p] = (rowok && oxp + p < W) ? src[oxp + p] : 0.f;
      if (aff) {
        const float al = aff[3 * ci], be = aff[3 * ci + 1], sl = aff[3 * ci + 2];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const float t = in[p] * al + be;
          in[p] = (rowok && oxp + p < W) ? (t > 0.f ? t : t * sl) : 0.f;
        }
      }
      const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(ci * CoutP + co0);
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const float wv = wc[c];
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p][c] = fmaf(in[p], wv, acc[p][c]);
      }
    }
  } else {
    for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
      __syncthreads();
      const int nrows = CI_CHUNK * Cfg::IY;
      for (int row = tid / 64; row < nrows; row += 4) {
        const int ci = row / Cfg::IY, ry = row % Cfg::IY;
        const int gy = gy0 + ry;
        const bool row_ok = (ci0 + ci < Cin) && gy >= 0 && gy < H;
        const float* __restrict__ src = xn + (size_t)(ci0 + ci) * plane + (size_t)gy * W;
        float* dst = lds + ci * Cfg::TILE + ry * Cfg::IXP;
        float al = 1.f, be = 0.f, sl = 1.f;
        if (aff && ci0 + ci < Cin) {
          al = aff[3 * (ci0 + ci)]; be = aff[3 * (ci0 + ci) + 1]; sl = aff[3 * (ci0 + ci) + 2];
        }
        for (int i = tid & 63; i < Cfg::IXP; i += 64) {
          const int gx = gx0 + i;
          float v = 0.f;
          if (row_ok && gx >= 0 && gx < W && i < Cfg::IX) {
            v = src[gx];
            if (aff) {
              const float t = v * al + be;
              v = t > 0.f ? t : t * sl;
            }
          }
          dst[i] = v;
        }
      }
      __syncthreads();
      const int cmax = min(CI_CHUNK, Cin - ci0);
      for (int ci = 0; ci < cmax; ++ci) {
        const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(((ci0 + ci) * K * K) * CoutP + co0);
        const float* tile_ci = lds + ci * Cfg::TILE;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const float* rowp = tile_ci + (ly * S + ky) * Cfg::IXP + lx * PX * S;
          float in[Cfg::NIN];
#pragma unroll
          for (int i = 0; i < Cfg::NIN; ++i) in[i] = rowp[i];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
              const float wv = wc[(ky * K + kx) * CoutP + c];
#pragma unroll
              for (int p = 0; p < PX; ++p) acc[p][c] = fmaf(in[p * S + kx], wv, acc[p][c]);
            }
          }
        }
      }
    }
  }

  const int oy = oy0 + ly, oxb = ox0 + lx * PX;
  if (oy >= Ho) return;
  const size_t oplane = (size_t)Ho * Wo;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    if (co0 + c < Cout) {
      const float b = bias ? bias[co0 + c] : 0.f;
      const size_t base = ((size_t)n * Cout + co0 + c) * oplane + (size_t)oy * Wo + oxb;
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (oxb + p < Wo) out[base + p] = cds_act_conv(acc[p][c] + b, act & 15) + ((act & C2_ACCUM) ? out[base + p] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "pipe" variant (W % 4 == 0, k = 3 / 5): as conv3d_k3_pipe_kernel.  The input tile starts at an x that is a multiple of 4
// (column c <-> x = S*ox0 - 4 + c), so it is staged with aligned 16-byte loads, a handful per thread instead of one dword
// per lane and row; the loads of chunk k+1 are issued before the FMAs of chunk k and land in registers meanwhile.  The
// row-per-wave staging of conv2d_kernel costs ~540 instructions per thread and chunk against the 1584 FMAs of a 3x3
// chunk: 3x3 layers ran at 30-45 TF where the 7x7 layers (5x the FMAs per staged byte) reach 80.
// ---------------------------------------------------------------------------------------------
template <int K, int S, int PX, int CI_CHUNK>
struct C2PipeCfg {
  static constexpr int LX = 16, LY = 16;
  static constexpr int TX = LX * PX, TY = LY;
  static constexpr int OFFX = 4 - (K - 1) / 2;                     // first needed column of a thread's window
  static constexpr int IY = (TY - 1) * S + K;
  static constexpr int IXP = (OFFX + (TX - 1) * S + K + 3) & ~3;
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IY * Q;                                // float4 per input channel
  static constexpr int TILE = NS * 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
  static constexpr int NIN = (PX - 1) * S + K;
  static constexpr int NV4 = (OFFX + NIN + 3) / 4;                 // aligned float4 reads covering the window (PX = 4)
};

template <int K, int S, int PX, int CI_CHUNK, int NCB, int CWE = 8 * NCB>
__global__ __launch_bounds__(256) void conv2d_pipe_kernel(const float* __restrict__ x, const float* __restrict__ in_affine,
                                                          const float* __restrict__ wpk, const float* __restrict__ bias,
                                                          float* __restrict__ out, int N, int Cin, int Cout, int CoutP,
                                                          int H, int W, int Ho, int Wo, int pad, int act, int tiles_x,
                                                          int tiles_y) {
  using Cfg = C2PipeCfg<K, S, PX, CI_CHUNK>;
  constexpr int CW = CWE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_groups = CoutP / (CO * NCB);
  const int ntiles = tiles_x * tiles_y;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_groups * N);
  const int cog = lin % co_groups;
  lin /= co_groups;
  const int tile = lin % ntiles;
  const int n = lin / ntiles;
  const int tx_i = tile % tiles_x, ty_i = tile / tiles_x;
  const int co0 = cog * (CO * NCB);
  const int tid = threadIdx.x;
  const int lx = tid % Cfg::LX, ly = tid / Cfg::LX;
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY;
  const int gx0 = ox0 * S - 4, gy0 = oy0 * S - pad;
  const size_t plane = (size_t)H * W;
  const float* __restrict__ xn = x + (size_t)n * Cin * plane;
  const float* __restrict__ aff = in_affine ? in_affine + (size_t)n * Cin * 3 : nullptr;

  int goff[Cfg::NSLOT];   // element offset of the slot's float4 inside channel ci0 (+ ci * plane), -1 = outside / unused
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    const int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int gy = gy0 + row, gx = gx0 + 4 * c4;
    const bool ok = (s < CI_CHUNK * Cfg::NS) && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;   // W % 4 == 0: whole groups
    goff[j] = ok ? (int)((size_t)ci * plane + (size_t)gy * W + gx) : -1;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = xn + (size_t)ci0 * plane;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int ci = (tid + 256 * j) / Cfg::NS;
      const bool ok = goff[j] >= 0 && ci0 + ci < Cin;
      pre[j] = *reinterpret_cast<const float4*>(ok ? xb + goff[j] : x);   // branch-free; masked when deposited
    }
  };

  float acc[PX][CW];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[p][c] = 0.f;

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      const int ci = s / Cfg::NS;
      const bool ok = goff[j] >= 0 && ci0 + ci < Cin;
      if (s < CI_CHUNK * Cfg::NS) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          v = pre[j];
          if (aff) {   // InstanceNorm + LeakyReLU of the producing layer, in-bounds values only (padding stays zero)
            const float al = aff[3 * (ci0 + ci)], be = aff[3 * (ci0 + ci) + 1], sl = aff[3 * (ci0 + ci) + 2];
            float t;
            t = v.x * al + be; v.x = t > 0.f ? t : t * sl;
            t = v.y * al + be; v.y = t > 0.f ? t : t * sl;
            t = v.z * al + be; v.z = t > 0.f ? t : t * sl;
            t = v.w * al + be; v.w = t > 0.f ? t : t * sl;
          }
        }
        *reinterpret_cast<float4*>(lds + 4 * s) = v;
      }
    }
    __syncthreads();
    if (ci0 + CI_CHUNK < Cin) issue(ci0 + CI_CHUNK);
    const int cmax = min(CI_CHUNK, Cin - ci0);
#pragma unroll 1
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(((ci0 + ci) * K * K) * CoutP + co0);
      const float* tile_ci = lds + ci * Cfg::TILE;
#pragma unroll 1
      for (int ky = 0; ky < K; ++ky) {   // not unrolled: one tap row of weights (K x CW scalars) in SGPRs at a time
        const float* rowp = tile_ci + (ly * S + ky) * Cfg::IXP + lx * PX * S;
        float in[Cfg::NIN];
        if constexpr (S == 1 && PX == 4) {
          float win[4 * Cfg::NV4];   // aligned 16-byte reads covering columns OFFX .. OFFX + NIN - 1 of the window
#pragma unroll
          for (int q = 0; q < Cfg::NV4; ++q) {
            const cds_f4 b = *reinterpret_cast<const cds_f4*>(rowp + 4 * q);
            win[4 * q] = b.x; win[4 * q + 1] = b.y; win[4 * q + 2] = b.z; win[4 * q + 3] = b.w;
          }
#pragma unroll
          for (int i = 0; i < Cfg::NIN; ++i) in[i] = win[Cfg::OFFX + i];
        } else {
#pragma unroll
          for (int i = 0; i < Cfg::NIN; ++i) in[i] = rowp[Cfg::OFFX + i];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            const float wv = wc[(ky * K + kx) * CoutP + c];
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[p][c] = fmaf(in[p * S + kx], wv, acc[p][c]);
          }
        }
      }
    }
  }

  const int oy = oy0 + ly, oxb = ox0 + lx * PX;
  if (oy >= Ho || oxb >= Wo) return;
  const size_t oplane = (size_t)Ho * Wo;
  const bool vec = (PX == 4) && ((Wo & 3) == 0);
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    if (co0 + c < Cout) {
      const float b = bias ? bias[co0 + c] : 0.f;
      const size_t base = ((size_t)n * Cout + co0 + c) * oplane + (size_t)oy * Wo + oxb;
      if (vec) {
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act & C2_ACCUM) prev = *reinterpret_cast<const float4*>(out + base);
        *reinterpret_cast<float4*>(out + base) =
            make_float4(cds_act_conv(acc[0][c] + b, act & 15) + prev.x, cds_act_conv(acc[1 % PX][c] + b, act & 15) + prev.y,
                        cds_act_conv(acc[2 % PX][c] + b, act & 15) + prev.z, cds_act_conv(acc[3 % PX][c] + b, act & 15) + prev.w);
      } else {
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (oxb + p < Wo) out[base + p] = cds_act_conv(acc[p][c] + b, act & 15) + ((act & C2_ACCUM) ? out[base + p] : 0.f);
      }
    }
  }
}

template <int K, int S, int PX, int CI_CHUNK, int NCB, int CWE = 8 * NCB>
int launch_conv2d_n(const float* x, const float* aff, const float* w, const float* b, float* out, int N, int Cin, int Cout,
                    int H, int W, int pad, int act, hipStream_t st) {
  using Cfg = C2Cfg<K, S, PX, CI_CHUNK>;
  const int Ho = (H + 2 * pad - K) / S + 1, Wo = (W + 2 * pad - K) / S + 1;
  const int CoutP = (Cout + CO - 1) / CO * CO;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY);
  const int co_groups = CoutP / (CO * NCB);
  const size_t lds_bytes = (K == 1 && S == 1) ? 0 : (size_t)Cfg::TILE * CI_CHUNK * sizeof(float);
  if constexpr (K >= 3 && K <= 5) {   // 7x7 / 11x11 amortise their staging already (80+ TF); measured slower here
    const bool pipe = !cds_env_is("CDS_CONV2D_PIPE", '0');   // A/B knob
    if (pipe && W % 4 == 0 && pad == (K - 1) / 2) {
      using PCfg = C2PipeCfg<K, S, PX, CI_CHUNK>;
      auto pk = conv2d_pipe_kernel<K, S, PX, CI_CHUNK, NCB, CWE>;
      hipLaunchKernelGGL(pk, dim3(tx * ty * co_groups * N), dim3(256), (size_t)PCfg::TILE * CI_CHUNK * sizeof(float), st, x, aff,
                         w, b, out, N, Cin, Cout, CoutP, H, W, Ho, Wo, pad, act, tx, ty);
      return cds_launch_status();
    }
  }
  auto kern = conv2d_kernel<K, S, PX, CI_CHUNK, NCB, CWE>;
  hipLaunchKernelGGL(kern, dim3(tx * ty * co_groups * N), dim3(256), lds_bytes, st, x, aff, w, b, out, N, Cin, Cout, CoutP, H,
                     W, Ho, Wo, pad, act, tx, ty);
  return cds_launch_status();
}

// All output-channel blocks of a layer from one staged tile where a variant exists (against one 8-wide block per workgroup:
// cascade forward 640x512 11.0 -> 9.8 ms, 1600x1184 46.9 -> 41.4 ms; the A/B knob is closed).
template <int K, int S, int CI_CHUNK>
int launch_conv2d(const float* x, const float* aff, const float* w, const float* b, float* out, int N, int Cin, int Cout,
                  int H, int W, int pad, int act, hipStream_t st) {
  const int blocks = (Cout + CO - 1) / CO;
  constexpr int PXW = (S == 2) ? 2 : 4;
  {
    const bool exact = !cds_env_is("CDS_CONV2D_EXACT", '0');
    if (exact && S == 1) {   // DynamicConv branch widths: one workgroup owns every output channel, no padded columns
      if (Cout == 11) return launch_conv2d_n<K, S, PXW, CI_CHUNK, 2, 11>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
      if constexpr (K <= 5) {
        if (Cout == 19) return launch_conv2d_n<K, S, 2, CI_CHUNK, 3, 19>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
        if (K <= 3 && Cout == 35) return launch_conv2d_n<K, S, 2, CI_CHUNK, 5, 35>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
      }
    }
    if (blocks % 2 == 0 && (blocks == 2 || K > 5))
      return launch_conv2d_n<K, S, PXW, CI_CHUNK, 2>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
    if constexpr (K <= 5) {
      if (K <= 3 && blocks % 5 == 0) return launch_conv2d_n<K, S, 2, CI_CHUNK, 5>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
      if (blocks % 4 == 0) return launch_conv2d_n<K, S, 2, CI_CHUNK, 4>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
      if (blocks % 3 == 0) return launch_conv2d_n<K, S, 2, CI_CHUNK, 3>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
      if (blocks % 2 == 0) return launch_conv2d_n<K, S, PXW, CI_CHUNK, 2>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
    }
  }
  return launch_conv2d_n<K, S, PXW, CI_CHUNK, 1>(x, aff, w, b, out, N, Cin, Cout, H, W, pad, act, st);
}

// ---------------------------------------------------------------------------------------------
// FPN lateral (module.py:253-254,260-261): 1x1 convolution over cat(nearest2x(coarse), skip) without building either
// the up-sampled tensor or the concatenation.  coarse [N][Ca][H/2][W/2], skip [N][Cb][H][W], weights packed
// [Ca + Cb][CoutP]; each source has its own optional normalise-on-load table.  Channel order and fp32 operation
// order are those of conv2d_kernel<1,...> run on the materialised concatenation (bit-identical results).
// ---------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256) void fpn_lateral_kernel(const float* __restrict__ xa, const float* __restrict__ affa,
                                                          const float* __restrict__ xb, const float* __restrict__ affb,
                                                          const float* __restrict__ wpk, float* __restrict__ out,
                                                          double* __restrict__ partial, int N, int Ca, int Cb, int Cout,
                                                          int CoutP, int H, int W, int tiles_x, int tiles_y) {
  constexpr int PX = 4, LX = 16, CW = CO * NCB;
  const int co_groups = CoutP / CW;
  const int ntiles = tiles_x * tiles_y;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_groups * N);
  const int cog = lin % co_groups;
  lin /= co_groups;
  const int tile = lin % ntiles;
  const int n = lin / ntiles;
  const int co0 = cog * CW;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  int oy = (tile / tiles_x) * 16 + ly, ox = ((tile % tiles_x) * LX + lx) * PX;
  const bool valid = oy < H && ox < W;
  if (!partial && !valid) return;
  if (!valid) {   // statistics: the wave reduction needs every lane; out-of-image lanes shadow pixel (0, 0), store nothing
    oy = 0;
    ox = 0;
  }
  const int Hc = H >> 1, Wc = W >> 1;
  const size_t plane = (size_t)H * W, cplane = (size_t)Hc * Wc;
  float acc[PX][CW];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[p][c] = 0.f;

  auto mac = [&](const float in[PX], int ci) {
    const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(ci * CoutP + co0);
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const float wv = wc[c];
#pragma unroll
      for (int p = 0; p < PX; ++p) acc[p][c] = fmaf(in[p], wv, acc[p][c]);
    }
  };
  // coarse source: output pixels ox..ox+3 (ox % 4 == 0) read the two coarse pixels ox/2, ox/2 + 1 of row oy/2
  {
    const float* __restrict__ src = xa + (size_t)n * Ca * cplane + (size_t)(oy >> 1) * Wc + (ox >> 1);
    const float* __restrict__ aff = affa ? affa + (size_t)n * Ca * 3 : nullptr;
    const bool second = (ox >> 1) + 1 < Wc;
    for (int ci = 0; ci < Ca; ++ci, src += cplane) {
      float v0 = src[0], v1 = second ? src[1] : 0.f;
      if (aff) {
        const float al = aff[3 * ci], be = aff[3 * ci + 1], sl = aff[3 * ci + 2];
        const float t0 = v0 * al + be, t1 = v1 * al + be;
        v0 = t0 > 0.f ? t0 : t0 * sl;
        v1 = t1 > 0.f ? t1 : t1 * sl;
      }
      const float in[PX] = {v0, v0, v1, v1};
      mac(in, ci);
    }
  }
  {
    const float* __restrict__ src = xb + (size_t)n * Cb * plane + (size_t)oy * W + ox;
    const float* __restrict__ aff = affb ? affb + (size_t)n * Cb * 3 : nullptr;
    for (int ci = 0; ci < Cb; ++ci, src += plane) {
      float in[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) in[p] = (ox + p < W) ? src[p] : 0.f;
      if (aff) {
        const float al = aff[3 * ci], be = aff[3 * ci + 1], sl = aff[3 * ci + 2];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const float t = in[p] * al + be;
          in[p] = t > 0.f ? t : t * sl;
        }
      }
      mac(in, Ca + ci);
    }
  }
  // InstanceNorm statistics of the output (see dynconv_blend_stats_kernel): one record per (tile, wave) and channel
  double* rec = partial ? partial + (((size_t)n * (ntiles * 4) + tile * 4 + (threadIdx.x >> 6)) * Cout) * 2 : nullptr;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    if (co0 + c < Cout) {
      float* o = out + ((size_t)n * Cout + co0 + c) * plane + (size_t)oy * W + ox;
      double ds = 0.0, dq = 0.0;
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (valid && ox + p < W) {
          o[p] = acc[p][c];
          const double v = (double)acc[p][c];
          ds += v;
          dq += v * v;
        }
      if (partial) {
        ds = wave_sum_f64(ds);
        dq = wave_sum_f64(dq);
        if ((threadIdx.x & 63) == 0) {
          rec[2 * (co0 + c)] = ds;
          rec[2 * (co0 + c) + 1] = dq;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// DynamicConv epilogue.  `branch` holds, for each kernel size k, the Cout conv responses followed
// by the 3 curvature responses: [K][Cout+3][H][W].
// ---------------------------------------------------------------------------------------------
// branch: [K][N][Cout+3][H][W]; out: [N][Cout][H][W]; norm_curv: [N][H][W]; image n = blockIdx.y
template <int K>
__global__ __launch_bounds__(256) void dynconv_blend_kernel(const float* __restrict__ branch,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, EpiBatch epi, float temperature,
                                                            float* __restrict__ out, float* __restrict__ norm_curv,
                                                            int N, int Cout, int H, int W, int n_shared) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int hw = H * W;
  if (p >= hw) return;
  const int n = blockIdx.y;
  // the first n_shared images (copies of the reference image, each with its own epipole) share branch slot 0
  const int slot = n < n_shared ? 0 : n - n_shared + 1;
  const int nslots = N - n_shared + 1;
  branch += (size_t)slot * (Cout + 3) * hw;
  out += (size_t)n * Cout * hw;
  const size_t bstride = (size_t)nslots * (Cout + 3) * hw;  // stride between kernel sizes
  float logit[K];
  norm_curv[(size_t)n * hw + p] = blend_weights<K>(branch, bstride, Cout, hw, p, W, epi.x(n), epi.y(n), w1, b1, w2,
                                                   temperature, logit);
  for (int c = 0; c < Cout; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s = s + branch[k * bstride + (size_t)c * hw + p] * logit[k];
    out[(size_t)c * hw + p] = s;
  }
}

// The same epilogue, PXT pixels per thread, that also leaves the InstanceNorm statistics of its output: every wave
// writes one (sum, sum of squares) record per channel (fp64, of the rounded fp32 outputs = what a separate statistics
// pass would read back), reduced in a fixed order by instnorm_reduce_kernel: no atomics, bit-reproducible.
// partial: [N][parts][Cout][2], parts = 4 * gridDim.x.
constexpr int BLEND_PXT = 4;
template <int K>
__global__ __launch_bounds__(256) void dynconv_blend_stats_kernel(const float* __restrict__ branch,
                                                                  const float* __restrict__ w1,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ w2, EpiBatch epi,
                                                                  float temperature, float* __restrict__ out,
                                                                  float* __restrict__ norm_curv,
                                                                  double* __restrict__ partial, int N, int Cout, int H,
                                                                  int W, int n_shared) {
  constexpr int PXT = BLEND_PXT;
  const int hw = H * W;
  const int n = blockIdx.y;
  const int slot = n < n_shared ? 0 : n - n_shared + 1;
  const int nslots = N - n_shared + 1;
  branch += (size_t)slot * (Cout + 3) * hw;
  out += (size_t)n * Cout * hw;
  const size_t bstride = (size_t)nslots * (Cout + 3) * hw;
  const int base = blockIdx.x * (256 * PXT) + threadIdx.x;
  float lg[PXT][K];
  int px[PXT];
  bool ok[PXT];
#pragma unroll
  for (int j = 0; j < PXT; ++j) {
    const int p = base + 256 * j;
    ok[j] = p < hw;
    px[j] = ok[j] ? p : hw - 1;
    const float nc = blend_weights<K>(branch, bstride, Cout, hw, px[j], W, epi.x(n), epi.y(n), w1, b1, w2, temperature, lg[j]);
    if (ok[j]) norm_curv[(size_t)n * hw + p] = nc;
  }
  const int wave = threadIdx.x >> 6;
  const int parts = 4 * gridDim.x;
  double* rec = partial + (((size_t)n * parts + blockIdx.x * 4 + wave) * Cout) * 2;
  for (int c = 0; c < Cout; ++c) {
    double ds = 0.0, dq = 0.0;
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) s = s + branch[k * bstride + (size_t)c * hw + px[j]] * lg[j][k];
      if (ok[j]) {
        out[(size_t)c * hw + px[j]] = s;
        const double v = (double)s;
        ds += v;
        dq += v * v;
      }
    }
    ds = wave_sum_f64(ds);
    dq = wave_sum_f64(dq);
    if ((threadIdx.x & 63) == 0) {
      rec[2 * c] = ds;
      rec[2 * c + 1] = dq;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm: per-channel sum / sum of squares in fp64 (ATen's CPU kernel accumulates float
// statistics in double), wave-shuffle + one atomic per block; then y = act(x*alpha + beta).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                             int hw, int blocks_per_c) {
  const int c = blockIdx.x / blocks_per_c, b = blockIdx.x % blocks_per_c;
  const float* __restrict__ xc = x + (size_t)c * hw;
  double s = 0.0, q = 0.0;
#pragma unroll 4
  for (int i = b * 256 + threadIdx.x; i < hw; i += blocks_per_c * 256) {
    double v = (double)xc[i];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  __shared__ double red[2][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[2 * c + 0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&stats[2 * c + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// x: [N][C][hw]; stats per (n,c); out: [N][C][hw] or channels-last [N][hw][C]; image n = blockIdx.y
// CT > 0: C == CT known at compile time (8 / 16 / 32: the FeatureNet stage outputs): the per-channel (alpha, beta) are
// computed once per workgroup (they were two fp64 divisions and a square root per thread and channel) and a channels-last
// pixel leaves as CT / 4 16-byte stores on CT * 4 contiguous bytes instead of CT scattered dwords.
template <int CT>
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x,
                                                             const double* __restrict__ stats, float* __restrict__ out,
                                                             int C, int hw, int act, int out_hwc) {
  __shared__ float ab[2][64];
  const int n = blockIdx.y;
  stats += (size_t)n * 2 * C;
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    const double mean = stats[2 * c] / hw;
    double var = stats[2 * c + 1] / hw - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    ab[0][c] = invstd;
    ab[1][c] = -(float)mean * invstd;
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  x += (size_t)n * C * hw;
  out += (size_t)n * C * hw;
  if (CT > 0) {
    float v[CT > 0 ? CT : 1];
#pragma unroll
    for (int c = 0; c < CT; ++c) v[c] = cds_apply_act(x[(size_t)c * hw + p] * ab[0][c] + ab[1][c], act);
    if (out_hwc) {
      float4* o4 = reinterpret_cast<float4*>(out + (size_t)p * CT);
#pragma unroll
      for (int c = 0; c < CT; c += 4) o4[c >> 2] = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) out[(size_t)c * hw + p] = v[c];
    }
    return;
  }
  for (int c = 0; c < C; ++c) {
    float v = cds_apply_act(x[(size_t)c * hw + p] * ab[0][c] + ab[1][c], act);
    if (out_hwc)
      out[(size_t)p * C + c] = v;
    else
      out[(size_t)c * hw + p] = v;
  }
}

// (sum, sum of squares) -> (alpha, beta, slope) so that a consumer applies y = leaky(x * alpha + beta, slope) on load
__global__ void instnorm_affine_kernel(const double* __restrict__ stats, float* __restrict__ affine, int nc, int hw,
                                       float slope) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc) return;
  const double mean = stats[2 * i] / hw;
  double var = stats[2 * i + 1] / hw - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  affine[3 * i] = invstd;
  affine[3 * i + 1] = -(float)mean * invstd;
  affine[3 * i + 2] = slope;
}

// partial [N][parts][C][2] -> stats [N][C][2] (sum, sum of squares) and, if asked, the (alpha, beta, slope) rows.
// One workgroup per (image, channel); fixed summation order.
__global__ __launch_bounds__(256) void instnorm_reduce_kernel(const double* __restrict__ partial, int parts, int C, int hw,
                                                              float slope, double* __restrict__ stats,
                                                              float* __restrict__ affine) {
  const int nc = blockIdx.x, n = nc / C, c = nc % C;
  const double* __restrict__ p = partial + ((size_t)n * parts * C + c) * 2;
  double s = 0.0, q = 0.0;
  // Eight records of a thread in flight per trip (indices clamped to the last record, the repeats are not added): the rolled loop -
  // one record, then a wait, per trip - made this launch a chain of ~parts / 256 L2 round trips (25 us for the 7 920 records of a
  // 1920x1056 image; a cascade forward runs 13 of these between its layers).  The additions keep their order: bit-identical sums.
#ifdef CDS_REDUCE_SERIAL   // the rolled loop (A/B)
  for (int i = threadIdx.x; i < parts; i += 256) {
    s += p[(size_t)i * C * 2];
    q += p[(size_t)i * C * 2 + 1];
  }
#else
  for (int i0 = threadIdx.x; i0 < parts; i0 += 256 * 8) {
    double a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = min(i0 + 256 * k, parts - 1);
      a[k] = p[(size_t)i * C * 2];
      b[k] = p[(size_t)i * C * 2 + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + 256 * k < parts) {
        s += a[k];
        q += b[k];
      }
  }
#endif
  s = wave_sum_f64(s);
  q = wave_sum_f64(q);
  __shared__ double red[2][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    stats[2 * nc] = s;
    stats[2 * nc + 1] = q;
    if (affine) {
      const double mean = s / hw;
      double var = q / hw - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float invstd = (float)(1.0 / sqrt(var + 1e-5));
      affine[3 * nc] = invstd;
      affine[3 * nc + 1] = -(float)mean * invstd;
      affine[3 * nc + 2] = slope;
    }
  }
}

// norm-curvature bookkeeping of a FeatureNet level (module.py:250-251,257-258,264-265): (a^2 + b^2 + c^2) / 3 and |c|
__global__ __launch_bounds__(256) void curvature_stats_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ c, float* __restrict__ nc_sum,
                                                              float* __restrict__ nc_abs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float va = a[i], vb = b[i], vc = c[i];
  nc_sum[i] = ((va * va + vb * vb) + vc * vc) / 3.0f;
  nc_abs[i] = fabsf(vc);
}

// backward of curvature_stats_kernel: ga = (g_sum / 3) 2 a, ..., gc += g_abs sgn(c); g_sum / g_abs may be NULL (no gradient)
__global__ __launch_bounds__(256) void curvature_stats_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  const float* __restrict__ c, const float* __restrict__ g_sum,
                                                                  const float* __restrict__ g_abs, float* __restrict__ ga,
                                                                  float* __restrict__ gb, float* __restrict__ gc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gs = g_sum ? g_sum[i] / 3.0f : 0.f;
  const float vc = c[i];
  ga[i] = gs * (2.0f * a[i]);
  gb[i] = gs * (2.0f * b[i]);
  gc[i] = gs * (2.0f * vc) + (g_abs ? g_abs[i] * (vc > 0.f ? 1.f : (vc < 0.f ? -1.f : 0.f)) : 0.f);
}

// out[v][i] = (x[v][i] + x[V+v][i]) / 2: per pair (ref_nc_sum + src_nc_sum) / 2 (model.py:59)
__global__ __launch_bounds__(256) void pair_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int V, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int v = 0; v < V; ++v) out[(size_t)v * n + i] = (x[(size_t)v * n + i] + x[(size_t)(V + v) * n + i]) / 2.0f;
}

// out[i] = (sum_v x[v][i]) / V, summed in view order (model.py:60,79)
__global__ __launch_bounds__(256) void view_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int V, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = x[i];
  for (int v = 1; v < V; ++v) s = s + x[(size_t)v * n + i];
  out[i] = s / (float)V;
}

}  // namespace

extern "C" int cds_pair_mean_f32(const float* x, float* out, int V, int n, void* stream) {
  if (!x || !out || V < 1 || n < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(pair_mean_kernel, dim3(cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, V, n);
  return cds_launch_status();
}

extern "C" int cds_view_mean_f32(const float* x, float* out, int V, int n, void* stream) {
  if (!x || !out || V < 1 || n < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(view_mean_kernel, dim3(cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, V, n);
  return cds_launch_status();
}

extern "C" int cds_curvature_stats_f32(const float* a, const float* b, const float* c, float* nc_sum, float* nc_abs, int n,
                                       void* stream) {
  if (!a || !b || !c || !nc_sum || !nc_abs || n < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(curvature_stats_kernel, dim3(cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c,
                     nc_sum, nc_abs, n);
  return cds_launch_status();
}

extern "C" int cds_curvature_stats_bwd_f32(const float* a, const float* b, const float* c, const float* g_sum, const float* g_abs,
                                           float* ga, float* gb, float* gc, int n, void* stream) {
  if (!a || !b || !c || !ga || !gb || !gc || n < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(curvature_stats_bwd_kernel, dim3(cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, g_sum, g_abs,
                     ga, gb, gc, n);
  return cds_launch_status();
}

extern "C" int cds_conv2d_affine_f32(const float* x, const float* in_affine, const float* weight, const float* bias,
                                     float* out, int N, int Cin, int Cout, int H, int W, int k, int stride, int pad,
                                     int act, void* stream) {
  if (!x || !weight || !out || N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || pad < 0) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float* a = in_affine;
  if (stride == 1) {
    switch (k) {
      case 1: return pad == 0 ? launch_conv2d<1, 1, 8>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st) : CDS_EINVAL;
      case 3: return launch_conv2d<3, 1, CDS_C2_CHUNK3>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st);
      case 5: return launch_conv2d<5, 1, 4>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st);
      case 7: return launch_conv2d<7, 1, 4>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st);
      case 11: return launch_conv2d<11, 1, 4>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st);
      default: return CDS_EINVAL;
    }
  }
  if (stride == 2 && k == 3) return launch_conv2d<3, 2, CDS_C2_CHUNK3S2>(x, a, weight, bias, out, N, Cin, Cout, H, W, pad, act, st);
  return CDS_EINVAL;
}

extern "C" int cds_fpn_stats_parts(int H, int W) { return 4 * cds_ceil_div(W, 64) * cds_ceil_div(H, 16); }

extern "C" int cds_conv2d_fpn_f32(const float* coarse, const float* coarse_affine, const float* skip,
                                  const float* skip_affine, const float* weight, float* out, float* partial, int N,
                                  int Ca, int Cb, int Cout, int H, int W, void* stream) {
  double* dpart = reinterpret_cast<double*>(partial);
  if (!coarse || !skip || !weight || !out || N < 1 || Ca < 1 || Cb < 1 || Cout < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
    return CDS_EINVAL;
  const int CoutP = (Cout + CO - 1) / CO * CO;
  const int tx = cds_ceil_div(W, 64), ty = cds_ceil_div(H, 16);
  hipStream_t st = (hipStream_t)stream;
  if ((CoutP / CO) % 2 == 0)
    hipLaunchKernelGGL(fpn_lateral_kernel<2>, dim3(tx * ty * (CoutP / 16) * N), dim3(256), 0, st, coarse, coarse_affine, skip,
                       skip_affine, weight, out, dpart, N, Ca, Cb, Cout, CoutP, H, W, tx, ty);
  else
    hipLaunchKernelGGL(fpn_lateral_kernel<1>, dim3(tx * ty * (CoutP / 8) * N), dim3(256), 0, st, coarse, coarse_affine, skip,
                       skip_affine, weight, out, dpart, N, Ca, Cb, Cout, CoutP, H, W, tx, ty);
  return cds_launch_status();
}

extern "C" int cds_conv2d_f32(const float* x, const float* weight, const float* bias, float* out, int N, int Cin,
                              int Cout, int H, int W, int k, int stride, int pad, int act, void* stream) {
  return cds_conv2d_affine_f32(x, nullptr, weight, bias, out, N, Cin, Cout, H, W, k, stride, pad, act, stream);
}

extern "C" int cds_dynconv_blend_shared_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                            const float* epipoles, float temperature, float* out, float* norm_curv,
                                            int N, int K, int Cout, int H, int W, int n_shared, void* stream) {
  if (!branches || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || N < 1 || N > CDS_MAX_IMAGES ||
      Cout < 1 || H < 1 || W < 1 || n_shared < 1 || n_shared > N)
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const EpiBatch epi{epipoles};
  dim3 grid(cds_ceil_div(H * W, 256), N), block(256);
  if (K == 2)
    hipLaunchKernelGGL(dynconv_blend_kernel<2>, grid, block, 0, st, branches, w1, b1, w2, epi, temperature, out,
                       norm_curv, N, Cout, H, W, n_shared);
  else if (K == 3)
    hipLaunchKernelGGL(dynconv_blend_kernel<3>, grid, block, 0, st, branches, w1, b1, w2, epi, temperature, out,
                       norm_curv, N, Cout, H, W, n_shared);
  else
    return CDS_EINVAL;
  return cds_launch_status();
}

extern "C" int cds_blend_stats_parts(int H, int W) { return 4 * cds_ceil_div(H * W, 256 * BLEND_PXT); }

extern "C" int cds_dynconv_blend_stats_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                           const float* epipoles, float temperature, float* out, float* norm_curv,
                                           float* partial, int N, int K, int Cout, int H, int W, int n_shared,
                                           void* stream) {
  if (!branches || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || N < 1 || N > CDS_MAX_IMAGES ||
      Cout < 1 || H < 1 || W < 1 || n_shared < 1 || n_shared > N)
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const EpiBatch epi{epipoles};
  dim3 grid(cds_ceil_div(H * W, 256 * BLEND_PXT), N), block(256);
  double* dpart = reinterpret_cast<double*>(partial);
  if (K == 2)
    hipLaunchKernelGGL(dynconv_blend_stats_kernel<2>, grid, block, 0, st, branches, w1, b1, w2, epi, temperature, out,
                       norm_curv, dpart, N, Cout, H, W, n_shared);
  else if (K == 3)
    hipLaunchKernelGGL(dynconv_blend_stats_kernel<3>, grid, block, 0, st, branches, w1, b1, w2, epi, temperature, out,
                       norm_curv, dpart, N, Cout, H, W, n_shared);
  else
    return CDS_EINVAL;
  return cds_launch_status();
}

extern "C" int cds_instnorm_reduce_f32(const float* partial, int parts, float* stats, float* affine, int N, int C, int H,
                                       int W, float slope, void* stream) {
  if (!partial || !stats || parts < 1 || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(instnorm_reduce_kernel, dim3(N * C), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const double*>(partial), parts, C, H * W, slope, reinterpret_cast<double*>(stats),
                     affine);
  return cds_launch_status();
}

extern "C" int cds_instnorm_apply_f32(const float* x, const float* stats, float* out, int N, int C, int H, int W, int act,
                                      int out_hwc, void* stream) {
  if (!x || !out || !stats || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  const int hw = H * W;
  if (C > 64) return CDS_EINVAL;
  const dim3 grid(cds_ceil_div(hw, 256), N), block(256);
  const double* st64 = reinterpret_cast<const double*>(stats);
  hipStream_t st = (hipStream_t)stream;
  if (C == 8) hipLaunchKernelGGL(instnorm_apply_kernel<8>, grid, block, 0, st, x, st64, out, C, hw, act, out_hwc);
  else if (C == 16) hipLaunchKernelGGL(instnorm_apply_kernel<16>, grid, block, 0, st, x, st64, out, C, hw, act, out_hwc);
  else if (C == 32) hipLaunchKernelGGL(instnorm_apply_kernel<32>, grid, block, 0, st, x, st64, out, C, hw, act, out_hwc);
  else hipLaunchKernelGGL(instnorm_apply_kernel<0>, grid, block, 0, st, x, st64, out, C, hw, act, out_hwc);
  return cds_launch_status();
}

extern "C" int cds_dynconv_blend_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                     const float* epipoles, float temperature, float* out, float* norm_curv, int N,
                                     int K, int Cout, int H, int W, void* stream) {
  return cds_dynconv_blend_shared_f32(branches, w1, b1, w2, epipoles, temperature, out, norm_curv, N, K, Cout, H, W, 1,
                                      stream);
}

// InstanceNorm statistics only: affine[n][c] = (1/std, -mean/std, slope) for a consumer that normalises on load
// (same fp64 statistics and the same x * alpha + beta expression as cds_instnorm_act_f32: bit-identical values)
extern "C" int cds_instnorm_affine_f32(const float* x, float* affine, float* stats, int N, int C, int H, int W, float slope,
                                       void* stream) {
  if (!x || !affine || !stats || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hw = H * W;
  double* dstats = reinterpret_cast<double*>(stats);  // scratch: 2*N*C doubles
  hipError_t e = hipMemsetAsync(dstats, 0, sizeof(double) * 2 * N * C, st);
  if (e != hipSuccess) return -(int)e;
  int bpc = cds_ceil_div(hw, 256 * 16);
  if (bpc < 1) bpc = 1;
  hipLaunchKernelGGL(instnorm_stats_kernel, dim3(N * C * bpc), dim3(256), 0, st, x, dstats, hw, bpc);
  hipLaunchKernelGGL(instnorm_affine_kernel, dim3(cds_ceil_div(N * C, 256)), dim3(256), 0, st, dstats, affine, N * C, hw,
                     slope);
  return cds_launch_status();
}

extern "C" int cds_instnorm_act_f32(const float* x, float* out, float* stats, int N, int C, int H, int W, int act,
                                    int out_hwc, void* stream) {
  if (!x || !out || !stats || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hw = H * W;
  double* dstats = reinterpret_cast<double*>(stats);  // scratch: 2*N*C doubles
  hipError_t e = hipMemsetAsync(dstats, 0, sizeof(double) * 2 * N * C, st);
  if (e != hipSuccess) return -(int)e;
  int bpc = cds_ceil_div(hw, 256 * 16);
  if (bpc < 1) bpc = 1;
  hipLaunchKernelGGL(instnorm_stats_kernel, dim3(N * C * bpc), dim3(256), 0, st, x, dstats, hw, bpc);
  if (C > 64) return CDS_EINVAL;
  hipLaunchKernelGGL(instnorm_apply_kernel<0>, dim3(cds_ceil_div(hw, 256), N), dim3(256), 0, st, x, dstats, out, C, hw, act,
                     out_hwc);
  return cds_launch_status();
}
