// K4 on the matrix cores: 3x3x3 convolution as an implicit GEMM on v_mfma_f32_16x16x4_f32 (fp32 in, fp32
// accumulate, bit-exact fmaf chains, 157 TF peak = the packed-VALU peak but with 1 instruction per 2048 FLOP, so the
// issue slots stay free for LDS reads / staging).  Used for the layers with Cout % 16 == 0 (N = 16 output channels
// per MFMA); the Cout = 8 / 1 layers stay on the packed-VALU kernels of conv3d.hip (an N = 16 tile would idle half).
//
//   D[m = voxel][n = cout] += A[m][k] * B[k][n],  k = 4 consecutive input channels of ONE tap per MFMA
//   A: lane l -> voxel (l & 15) of a 16-voxel x-run, input channel (l >> 4): one ds_read_b32 from the LDS tile
//   B: lane l -> weight [ci0 + (l >> 4)][tap][co0 + (l & 15)]: one cached global load per (tap, chunk), reused by
//      every M-tile of the wave
//   C/D: lane l holds cout (l & 15), voxels (l >> 4) * 4 + 0..3 of the run -> one 16-byte store per M-tile
#include <stdlib.h>

#include "cds_common.hpp"

#ifndef CDS_MFMA_TZ1
#define CDS_MFMA_TZ1 2   // z planes per stride-1 tile: 2 (28 KB of LDS, 32 accumulators) measured 4-5 % faster than 4
#endif
#ifndef CDS_MFMA_TZ2
#define CDS_MFMA_TZ2 2
#endif
#ifndef CDS_MFMA_MINW
#define CDS_MFMA_MINW 3   // 3 waves per SIMD (<= 168 VGPRs): measured 5 % faster than 2 despite a few spilled staging registers
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// VAR = 0: wave = one y row of the tile, its M-tiles cover XT runs x TZ planes.
// VAR = 1 (stride 2): 64 x 2 x 2 output tile, wave = one (y, z) row with 4 runs: the staged input rows are 132 floats
// (528 B) instead of 72, which the stride-2 layers (whole input volume streamed once, little reuse) read faster.
template <int S, int VAR>
struct MCfg {
  static constexpr int TX = VAR ? 64 : 64 / S, TY = VAR ? 2 : 4, TZ = VAR ? 2 : (S == 1 ? CDS_MFMA_TZ1 : CDS_MFMA_TZ2);   // output tile
  static constexpr int XT = TX / 16;                                           // 16-voxel runs per row
  static constexpr int NT = VAR ? XT : XT * TZ;                               // M-tiles per wave
  static constexpr int IY = (TY - 1) * S + 3, IZ = (TZ - 1) * S + 3;
  static constexpr int IXP = ((TX - 1) * S + 6 + 3) & ~3;                     // tile x origin = S*ox0 - 4
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IZ * IY * Q;                                      // float4 per channel
  static constexpr int SLAB = NS * 4 + 16;                                    // +16 floats: channel k lands 16 banks apart
  static constexpr int CI_CHUNK = 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
};

// UNAL: W % 4 != 0 (e.g. the 50-wide deepest level of a 1600-wide scene): rows are not 16-byte aligned, the tile is staged
// with four bounds-checked dword loads per slot instead of one dwordx4 (small layers; the point is to stay on the MFMA path).
template <int S, int VAR, bool UNAL = false>
__global__ __launch_bounds__(256, CDS_MFMA_MINW) void conv3d_k3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ out,
                                                             int Cin, int Cout, int D, int H, int W, int Do, int Ho,
                                                             int Wo, int act, int tiles_x, int tiles_y, int tiles_z,
                                                             int ntiles) {
  using Cfg = MCfg<S, VAR>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / 16;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = output row y of this wave inside the tile
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY, oz0 = tz_i * Cfg::TZ;
  const int gx0 = ox0 * S - 4, gy0 = oy0 * S - 1, gz0 = oz0 * S - 1;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  // ---- staging slots (same scheme as conv3d_k3_pipe_kernel: aligned float4, one chunk ahead) ----
  int goff[Cfg::NSLOT];
  int loff[Cfg::NSLOT];
  int gxs[UNAL ? Cfg::NSLOT : 1];   // UNAL: goff is the offset of the row start, the columns are tested one by one
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = gz0 + rz, gy = gy0 + ry, gx = gx0 + 4 * c4;
    const bool row_ok = (s < Cfg::CI_CHUNK * Cfg::NS) && gz >= 0 && gz < D && gy >= 0 && gy < H;
    if constexpr (UNAL) {
      goff[j] = row_ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W) : -1;
      gxs[j] = gx;
    } else {
      const bool ok = row_ok && gx >= 0 && gx + 3 < W;
      goff[j] = ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
    }
    loff[j] = ci * Cfg::SLAB + 4 * r;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const bool ok = goff[j] >= 0;
      if constexpr (UNAL) {
        const float* __restrict__ rowp = xb + (ok ? goff[j] : 0);
        const int g = gxs[j];
        pre[j].x = (ok && (unsigned)(g + 0) < (unsigned)W) ? rowp[g + 0] : 0.f;
        pre[j].y = (ok && (unsigned)(g + 1) < (unsigned)W) ? rowp[g + 1] : 0.f;
        pre[j].z = (ok && (unsigned)(g + 2) < (unsigned)W) ? rowp[g + 2] : 0.f;
        pre[j].w = (ok && (unsigned)(g + 3) < (unsigned)W) ? rowp[g + 3] : 0.f;
      } else {
        pre[j] = *reinterpret_cast<const float4*>(ok ? xb + goff[j] : x);
      }
    }
  };

  f32x4 acc[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // lane-constant part of the A address: channel slab (l >> 4), voxel (l & 15) of the run, this wave's row
  const int wy = VAR ? (wave & 1) : wave, wz = VAR ? (wave >> 1) : 0;
  const float* a_base = lds + (lane >> 4) * Cfg::SLAB + ((wz * S) * Cfg::IY + wy * S) * Cfg::IXP + (lane & 15) * S + 3;
  // lane-constant part of the B address: [ci0 + (l >> 4)][tap][co0 + (l & 15)]
  const float* __restrict__ b_base = wpk + (size_t)(lane >> 4) * 27 * Cout + co0 + (lane & 15);

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += Cfg::CI_CHUNK) {   // host guarantees Cin % 4 == 0
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      if (s < Cfg::CI_CHUNK * Cfg::NS)
        *reinterpret_cast<float4*>(lds + loff[j]) = goff[j] >= 0 ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ci0 + Cfg::CI_CHUNK < Cin) issue(ci0 + Cfg::CI_CHUNK);
    const float* __restrict__ bw = b_base + (size_t)ci0 * 27 * Cout;
#pragma unroll 1
    for (int kz = 0; kz < 3; ++kz) {
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {
        float bv[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) bv[kx] = bw[((kz * 3 + ky) * 3 + kx) * Cout];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
          for (int t = 0; t < Cfg::NT; ++t) {
            const int tz = VAR ? 0 : t / Cfg::XT, txr = VAR ? t : t % Cfg::XT;
            const float a = a_base[((tz * S + kz) * Cfg::IY + ky) * Cfg::IXP + txr * 16 * S + kx];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[kx], acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: lane -> cout (l & 15), voxels x = run*16 + (l >> 4)*4 + 0..3 ----
  const int oy = oy0 + wy;
  if (oy >= Ho) return;
  const int co = co0 + (lane & 15);
  const float b = bias ? bias[co] : 0.f;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const bool vec = (Wo & 3) == 0;
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) {
    const int tz = VAR ? wz : t / Cfg::XT, txr = VAR ? t : t % Cfg::XT;
    const int oz = oz0 + tz, oxb = ox0 + txr * 16 + (lane >> 4) * 4;
    if (oz >= Do || oxb >= Wo) continue;
    const size_t base = (size_t)co * ovol + (size_t)oz * oplane + (size_t)oy * Wo + oxb;
    float v[4] = {acc[t].x + b, acc[t].y + b, acc[t].z + b, acc[t].w + b};
    if (act == CDS_ACT_RELU) {
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = fmaxf(v[p], 0.f);
    }
    if (vec) {
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (skip) {
        const float4 s4 = *reinterpret_cast<const float4*>(skip + base);
        o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
      }
      *reinterpret_cast<float4*>(out + base) = o;
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (oxb + p < Wo) out[base + p] = skip ? skip[base + p] + v[p] : v[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Channels-last LDS variant (stride 1, Cin % 16 == 0, Cout % 16 == 0).  Micro-benchmark (scripts/ubench/mfma_rate.hip):
// back-to-back v_mfma_f32_16x16x4_f32 reach 148 TF, but with ONE ds_read_b32 per MFMA (the kernel above) only 86 TF at
// 2 waves/SIMD — an LDS read costs ~10 ns of its SIMD regardless of width.  Here the LDS tile is [z][y][x][16 ci]
// (64 B per position, 16-byte slots XOR-swizzled by (x >> 1) & 3 so the 16 lanes of an M-run are conflict-free) and
// lane (m, k) fetches ci = 4k..4k+3 of its voxel with one ds_read_b128 = the A operands of FOUR MFMAs (K-step c
// multiplies ci = 4k + c); the matching B operands w[tap][co][4k..4k+3] are one global_load_dwordx4 from a
// ci-fastest weight copy.  5 operand fetches per 16 MFMAs instead of 17.
//   tile 32 x 4 x 2 outputs (wave = y row, 4 M-tiles of 16 voxels), LDS 4*6*40 positions * 64 B = 60 KB
// ---------------------------------------------------------------------------------------------
struct MClCfg {
  static constexpr int TX = 32, TY = 4, TZ = 2, XT = TX / 16, NT = XT * TZ;
  static constexpr int IY = TY + 2, IZ = TZ + 2;
  static constexpr int IXP = TX + 8;                 // column c <-> x = ox0 - 4 + c (16-byte aligned global rows)
  static constexpr int Q = IXP / 4;
  static constexpr int NPOS = IZ * IY * IXP;          // 960
  static constexpr int CI = 16;                       // input channels resident per chunk
  static constexpr int NSLOTS = IZ * IY * Q * 4;      // (row, x-group, k-group) staging slots
  static constexpr int NSLOT = (NSLOTS + 255) / 256;  // per thread
};

__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma_cl_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ wcl,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ skip,
                                                                    float* __restrict__ out, int Cin, int Cout, int D,
                                                                    int H, int W, int act, int tiles_x, int tiles_y,
                                                                    int tiles_z, int ntiles) {
  using Cfg = MClCfg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / 16;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = output row y of this wave inside the tile
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY, oz0 = tz_i * Cfg::TZ;
  const int gx0 = ox0 - 4, gy0 = oy0 - 1, gz0 = oz0 - 1;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  // ---- staging: slot = (row, x-group q, k-group kg): 4 channels x 4 x-positions, transposed in registers ----
  int goff[Cfg::NSLOT];    // element offset of (ci = 4 kg, row, 4q) inside the chunk, -1 = outside / unused
  int loff[Cfg::NSLOT][2]; // LDS float offsets of x = 4q (+1 shares the swizzle) and x = 4q + 2 (+3)
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int kg = s & 3;
    const int q = (s >> 2) % Cfg::Q;
    const int row = (s >> 2) / Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = gz0 + rz, gy = gy0 + ry, gx = gx0 + 4 * q;
    const bool ok = (s < Cfg::NSLOTS) && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
    goff[j] = ok ? (int)((size_t)(4 * kg) * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
    const int pos = row * Cfg::IXP + 4 * q;
    loff[j][0] = pos * 16 + ((kg ^ ((2 * q) & 3)) << 2);
    loff[j][1] = (pos + 2) * 16 + ((kg ^ ((2 * q + 1) & 3)) << 2);
  }
  float4 pre[Cfg::NSLOT][4];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const bool ok = goff[j] >= 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[j][i] = *reinterpret_cast<const float4*>(ok ? xb + goff[j] + (size_t)i * vol : x);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      if (s < Cfg::NSLOTS) {
        const bool ok = goff[j] >= 0;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 c0 = ok ? make_float4(pre[j][0].x, pre[j][1].x, pre[j][2].x, pre[j][3].x) : z4;
        const float4 c1 = ok ? make_float4(pre[j][0].y, pre[j][1].y, pre[j][2].y, pre[j][3].y) : z4;
        const float4 c2 = ok ? make_float4(pre[j][0].z, pre[j][1].z, pre[j][2].z, pre[j][3].z) : z4;
        const float4 c3 = ok ? make_float4(pre[j][0].w, pre[j][1].w, pre[j][2].w, pre[j][3].w) : z4;
        *reinterpret_cast<float4*>(lds + loff[j][0]) = c0;
        *reinterpret_cast<float4*>(lds + loff[j][0] + 16) = c1;
        *reinterpret_cast<float4*>(lds + loff[j][1]) = c2;
        *reinterpret_cast<float4*>(lds + loff[j][1] + 16) = c3;
      }
    }
  };

  // ---- lane-constant A addresses: voxel m = l & 15 of run txr, tap column kx; row / plane offsets are immediates ----
  const int m = lane & 15, kq = lane >> 4;
  const float* a_ptr[3][Cfg::XT];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int txr = 0; txr < Cfg::XT; ++txr) {
      const int col = 3 + txr * 16 + m + kx;
      a_ptr[kx][txr] = lds + (wave * Cfg::IXP + col) * 16 + ((kq ^ ((col >> 1) & 3)) << 2);
    }
  // B: w_cl[tap][co][ci], lane (n = l & 15, k) reads ci = ci0 + 4k .. +3 of output channel co0 + n
  const float* __restrict__ b_lane = wcl + (size_t)(co0 + m) * Cin + 4 * kq;
  const size_t tap_stride = (size_t)Cout * Cin;

  f32x4 acc[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += Cfg::CI) {
    __syncthreads();
    deposit();
    __syncthreads();
    if (ci0 + Cfg::CI < Cin) issue(ci0 + Cfg::CI);
    const float* __restrict__ bw = b_lane + ci0;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const cds_f4 bv = *reinterpret_cast<const cds_f4*>(bw + (size_t)((kz * 3 + ky) * 3 + kx) * tap_stride);
#pragma unroll
          for (int t = 0; t < Cfg::NT; ++t) {
            const int tz = t / Cfg::XT, txr = t % Cfg::XT;
            const cds_f4 av = *reinterpret_cast<const cds_f4*>(a_ptr[kx][txr] + ((tz + kz) * Cfg::IY + ky) * Cfg::IXP * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: lane -> cout (l & 15), voxels x = run*16 + (l >> 4)*4 + 0..3 ----
  const int oy = oy0 + wave;
  if (oy >= H) return;
  const int co = co0 + m;
  const float b = bias ? bias[co] : 0.f;
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) {
    const int tz = t / Cfg::XT, txr = t % Cfg::XT;
    const int oz = oz0 + tz, oxb = ox0 + txr * 16 + kq * 4;
    if (oz >= D || oxb >= W) continue;          // W % 4 == 0: the four voxels are inside together
    const size_t base = (size_t)co * vol + (size_t)oz * plane + (size_t)oy * W + oxb;
    float4 o = make_float4(acc[t].x + b, acc[t].y + b, acc[t].z + b, acc[t].w + b);
    if (act == CDS_ACT_RELU) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (skip) {
      const float4 s4 = *reinterpret_cast<const float4*>(skip + base);
      o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
    }
    *reinterpret_cast<float4*>(out + base) = o;
  }
}

template <int S, int VAR, bool UNAL = false>
int launch_mfma(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout, int D,
                int H, int W, int act, hipStream_t st) {
  using Cfg = MCfg<S, VAR>;
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY), tz = cds_ceil_div(Do, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::SLAB * Cfg::CI_CHUNK * sizeof(float);
  static_assert(Cfg::SLAB * Cfg::CI_CHUNK * sizeof(float) <= 65536, "LDS tile above 64 KB");
  hipLaunchKernelGGL((conv3d_k3_mfma_kernel<S, VAR, UNAL>), dim3(ntiles * (Cout / 16)), dim3(256), lds_bytes, st, x, w, b, skip, out,
                     Cin, Cout, D, H, W, Do, Ho, Wo, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}


// ---------------------------------------------------------------------------------------------
// Transposed conv (k3, s2, p1, op1) on the matrix cores.  A workgroup owns 64x4x2 input cells and one (z,y) output
// parity class; M = 16 x-consecutive cells, N = 16 couts, K = 4 input channels of one tap.  Both x parities are
// accumulated in the same lane (acc0: x = 2a, acc1: x = 2a+1) so the epilogue stores 8 consecutive outputs.
// ---------------------------------------------------------------------------------------------
struct MDCfg {
  static constexpr int CX = 64, CY = 4, CZ = 2;                // input cells per workgroup
  static constexpr int XT = CX / 16, NT = XT * CZ;             // M-tiles per wave (wave = one y row of cells)
  static constexpr int IY = CY + 1, IZ = CZ + 1;
  static constexpr int IXP = (CX + 1 + 3) & ~3;                // 68
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IZ * IY * Q;
  static constexpr int SLAB = NS * 4 + 16;
  static constexpr int CI_CHUNK = 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
};

// NCO = 16: N = 16 output channels, x parities in two accumulators.  NCO = 8: N = (cout, x parity) = 8 x 2 in ONE
// accumulator (2 MFMAs per tap row instead of 3; the B operand of the second one is zero for the even parity).
// UNAL (NCO = 16 only): W even but not a multiple of 4 — bounds-checked dword staging, output stored per pair of cells.
template <int NCO, bool UNAL = false>
__global__ __launch_bounds__(256) void deconv3d_k3s2_mfma_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ wpk,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ skip, float* __restrict__ out,
                                                                 int Cin, int Cout, int D, int H, int W, int act,
                                                                 int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  using Cfg = MDCfg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / NCO;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks * 4);
  const int cls = lin & 3;
  lin >>= 2;
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * NCO;
  const int pz = cls >> 1, py = cls & 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ax0 = tx_i * Cfg::CX, ay0 = ty_i * Cfg::CY, az0 = tz_i * Cfg::CZ;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  int goff[Cfg::NSLOT];
  int loff[Cfg::NSLOT];
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = az0 + rz, gy = ay0 + ry, gx = ax0 + 4 * c4;
    const bool ok = (s < Cfg::CI_CHUNK * Cfg::NS) && gz < D && gy < H && (UNAL ? gx < W : gx + 3 < W);
    goff[j] = ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
    loff[j] = ci * Cfg::SLAB + 4 * r;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      if constexpr (UNAL) {
        const bool ok = goff[j] >= 0;
        const float* __restrict__ p = xb + (ok ? goff[j] : 0);
        const int gx = ax0 + 4 * (((tid + 256 * j) % Cfg::NS) % Cfg::Q);
        pre[j].x = ok ? p[0] : 0.f;                      // gx < W by construction of goff
        pre[j].y = (ok && gx + 1 < W) ? p[1] : 0.f;
        pre[j].z = (ok && gx + 2 < W) ? p[2] : 0.f;
        pre[j].w = (ok && gx + 3 < W) ? p[3] : 0.f;
      } else {
        pre[j] = *reinterpret_cast<const float4*>(goff[j] >= 0 ? xb + goff[j] : x);
      }
    }
  };

  f32x4 acc0[Cfg::NT], acc1[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc0[t] = acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* a_base = lds + (lane >> 4) * Cfg::SLAB + wave * Cfg::IXP + (lane & 15);
  const int nco = NCO == 16 ? (lane & 15) : ((lane & 15) >> 1);   // output channel of this lane's column
  const int npx = (lane & 1);                                      // NCO = 8: x parity of this lane's column
  const float* __restrict__ b_base = wpk + (size_t)(lane >> 4) * 27 * Cout + co0 + nco;

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += Cfg::CI_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      if (s < Cfg::CI_CHUNK * Cfg::NS)
        *reinterpret_cast<float4*>(lds + loff[j]) = goff[j] >= 0 ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ci0 + Cfg::CI_CHUNK < Cin) issue(ci0 + Cfg::CI_CHUNK);
    const float* __restrict__ bw = b_base + (size_t)ci0 * 27 * Cout;
    for (int sz = 0; sz <= pz; ++sz) {
      const int iz = pz ? 1 - sz : 0, kz = pz ? 2 * sz : 1;
      for (int sy = 0; sy <= py; ++sy) {
        const int iy = py ? 1 - sy : 0, ky = py ? 2 * sy : 1;
        const float* __restrict__ bt = bw + ((kz * 3 + ky) * 3) * Cout;
        const float b0 = bt[0], b1 = bt[Cout], b2 = bt[2 * Cout];   // taps kx = 0, 1, 2
        const float* arow = a_base + (iz * Cfg::IY + iy) * Cfg::IXP;
#pragma unroll
        for (int t = 0; t < Cfg::NT; ++t) {
          const int tz = t / Cfg::XT, txr = t % Cfg::XT;
          const float* ap = arow + (tz * Cfg::IY) * Cfg::IXP + txr * 16;
          const float a0 = ap[0], a1 = ap[1];
          if constexpr (NCO == 16) {
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc0[t], 0, 0, 0);  // x = 2a   : (cell a,   tap 1)
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc1[t], 0, 0, 0);  // x = 2a+1 : (cell a+1, tap 0)
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2, acc1[t], 0, 0, 0);  //            (cell a,   tap 2)
          } else {
            // column (cout, parity): cell a feeds tap 1 (even) / tap 2 (odd); cell a+1 feeds tap 0 (odd only)
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, npx ? b2 : b1, acc0[t], 0, 0, 0);
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, npx ? b0 : 0.f, acc0[t], 0, 0, 0);
          }
        }
      }
    }
  }

  const int ay = ay0 + wave;
  if (ay >= H) return;
  const int co = co0 + nco;
  const float b = bias ? bias[co] : 0.f;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) {
    const int tz = t / Cfg::XT, txr = t % Cfg::XT;
    const int az = az0 + tz, ax = ax0 + txr * 16 + (lane >> 4) * 4;   // first of this lane's 4 cells
    const bool inside = az < D && ax < W;                              // W % 4 == 0: the 4 cells are all in or all out
    const size_t base = (size_t)co * ovol + (size_t)(2 * az + pz) * oplane + (size_t)(2 * ay + py) * Wo + 2 * ax;
    if constexpr (NCO == 16) {
      if (!inside) continue;
      float v[8] = {acc0[t].x, acc1[t].x, acc0[t].y, acc1[t].y, acc0[t].z, acc1[t].z, acc0[t].w, acc1[t].w};
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        v[p] += b;
        if (act == CDS_ACT_RELU) v[p] = fmaxf(v[p], 0.f);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (UNAL && ax + 2 * half >= W) continue;   // W even: cells come in pairs, a pair = 4 consecutive outputs
        float4 o = make_float4(v[4 * half], v[4 * half + 1], v[4 * half + 2], v[4 * half + 3]);
        if (skip) {
          const float4 s4 = *reinterpret_cast<const float4*>(skip + base + 4 * half);
          o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
        }
        *reinterpret_cast<float4*>(out + base + 4 * half) = o;
      }
    } else {
      // this lane: parity npx of cells c0..c3; the neighbouring lane (l ^ 1): the other parity of the same cells.
      float mine[4] = {acc0[t].x + b, acc0[t].y + b, acc0[t].z + b, acc0[t].w + b};
      float other[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (act == CDS_ACT_RELU) mine[p] = fmaxf(mine[p], 0.f);
        other[p] = __shfl_xor(mine[p], 1);
      }
      if (!inside) continue;
      // even lane stores x = 2c0 .. 2c0+3 (cells c0, c1), odd lane x = 2c2 .. 2c2+3 (cells c2, c3)
      float4 o = npx ? make_float4(other[2], mine[2], other[3], mine[3]) : make_float4(mine[0], other[0], mine[1], other[1]);
      const size_t addr = base + (npx ? 4 : 0);
      if (skip) {
        const float4 s4 = *reinterpret_cast<const float4*>(skip + addr);
        o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
      }
      *reinterpret_cast<float4*>(out + addr) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Transposed conv, all four (z, y) output parity classes in ONE workgroup.  The kernel above launches a workgroup per
// class and each stages the same input tile (PMC: 28-33 % MFMA-pipe utilisation, the rest is staging); here the 27 taps
// of a staged chunk go to their class accumulators (tap (kz, ky) belongs to class (kz != 1, ky != 1)), so the tile is
// staged once for 27 MFMAs per M-tile instead of four times for 3 / 6 / 6 / 12.  64 x 4 x 1 input cells per workgroup
// (4 M-tiles per wave x 4 classes x 2 x-parities = 32 accumulators).  Cout % 16 == 0; UNAL as above.
// ---------------------------------------------------------------------------------------------
struct MD4Cfg {
  static constexpr int CX = 64, CY = 4;                        // input cells per workgroup (one z plane of cells)
  static constexpr int XT = CX / 16;                           // M-tiles per wave (wave = one y row of cells)
  static constexpr int IY = CY + 1, IZ = 2;
  static constexpr int IXP = (CX + 1 + 3) & ~3;                // 68
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IZ * IY * Q;
  static constexpr int SLAB = NS * 4 + 16;
  static constexpr int CI_CHUNK = 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
};

template <bool UNAL>
__global__ __launch_bounds__(256, 2) void deconv3d_k3s2_mfma4_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ skip, float* __restrict__ out,
                                                                  int Cin, int Cout, int D, int H, int W, int act,
                                                                  int tiles_x, int tiles_y, int ntiles) {
  using Cfg = MD4Cfg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / 16;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int az = tile / tiles_y;                               // one plane of cells per workgroup
  const int co0 = cob * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ax0 = tx_i * Cfg::CX, ay0 = ty_i * Cfg::CY;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  int goff[Cfg::NSLOT];
  int loff[Cfg::NSLOT];
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = az + rz, gy = ay0 + ry, gx = ax0 + 4 * c4;
    const bool ok = (s < Cfg::CI_CHUNK * Cfg::NS) && gz < D && gy < H && (UNAL ? gx < W : gx + 3 < W);
    goff[j] = ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
    loff[j] = ci * Cfg::SLAB + 4 * r;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      if constexpr (UNAL) {
        const bool ok = goff[j] >= 0;
        const float* __restrict__ p = xb + (ok ? goff[j] : 0);
        const int gx = ax0 + 4 * (((tid + 256 * j) % Cfg::NS) % Cfg::Q);
        pre[j].x = ok ? p[0] : 0.f;
        pre[j].y = (ok && gx + 1 < W) ? p[1] : 0.f;
        pre[j].z = (ok && gx + 2 < W) ? p[2] : 0.f;
        pre[j].w = (ok && gx + 3 < W) ? p[3] : 0.f;
      } else {
        pre[j] = *reinterpret_cast<const float4*>(goff[j] >= 0 ? xb + goff[j] : x);
      }
    }
  };

  // [class = 2 pz + py][M-tile]: acc0 -> x = 2a, acc1 -> x = 2a + 1
  f32x4 acc0[4][Cfg::XT], acc1[4][Cfg::XT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < Cfg::XT; ++t) acc0[c][t] = acc1[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* a_base = lds + (lane >> 4) * Cfg::SLAB + wave * Cfg::IXP + (lane & 15);
  const float* __restrict__ b_base = wpk + (size_t)(lane >> 4) * 27 * Cout + co0 + (lane & 15);

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += Cfg::CI_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      if (s < Cfg::CI_CHUNK * Cfg::NS)
        *reinterpret_cast<float4*>(lds + loff[j]) = goff[j] >= 0 ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ci0 + Cfg::CI_CHUNK < Cin) issue(ci0 + Cfg::CI_CHUNK);
    const float* __restrict__ bw = b_base + (size_t)ci0 * 27 * Cout;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // output (2a + p) takes tap k from input cell a + (p + 1 - k) / 2: k = 1 -> parity 0, cell a; k = 2 -> parity 1,
        // cell a; k = 0 -> parity 1, cell a + 1
        const int cls = (kz != 1 ? 2 : 0) + (ky != 1 ? 1 : 0);
        const int iz = kz == 0 ? 1 : 0, iy = ky == 0 ? 1 : 0;
        const float* __restrict__ bt = bw + ((kz * 3 + ky) * 3) * Cout;
        const float b0 = bt[0], b1 = bt[Cout], b2 = bt[2 * Cout];
        const float* arow = a_base + (iz * Cfg::IY + iy) * Cfg::IXP;
#pragma unroll
        for (int t = 0; t < Cfg::XT; ++t) {
          const float a0 = arow[t * 16], a1 = arow[t * 16 + 1];
          acc0[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc0[cls][t], 0, 0, 0);
          acc1[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc1[cls][t], 0, 0, 0);
          acc1[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2, acc1[cls][t], 0, 0, 0);
        }
      }
    }
  }

  const int ay = ay0 + wave;
  if (ay >= H) return;
  const int co = co0 + (lane & 15);
  const float b = bias ? bias[co] : 0.f;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int pz = cls >> 1, py = cls & 1;
#pragma unroll
    for (int t = 0; t < Cfg::XT; ++t) {
      const int ax = ax0 + t * 16 + (lane >> 4) * 4;   // first of this lane's 4 cells
      if (ax >= W) continue;
      const size_t base = (size_t)co * ovol + (size_t)(2 * az + pz) * oplane + (size_t)(2 * ay + py) * Wo + 2 * ax;
      float v[8] = {acc0[cls][t].x, acc1[cls][t].x, acc0[cls][t].y, acc1[cls][t].y,
                    acc0[cls][t].z, acc1[cls][t].z, acc0[cls][t].w, acc1[cls][t].w};
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        v[p] += b;
        if (act == CDS_ACT_RELU) v[p] = fmaxf(v[p], 0.f);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (UNAL && ax + 2 * half >= W) continue;   // W even: cells come in pairs, a pair = 4 consecutive outputs
        float4 o = make_float4(v[4 * half], v[4 * half + 1], v[4 * half + 2], v[4 * half + 3]);
        if (skip) {
          const float4 s4 = *reinterpret_cast<const float4*>(skip + base + 4 * half);
          o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
        }
        *reinterpret_cast<float4*>(out + base + 4 * half) = o;
      }
    }
  }
}

}  // namespace

// Returns false when the shape is not covered (caller falls back to the VALU kernels).
bool cds_conv3d_mfma_launch(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin,
                            int Cout, int D, int H, int W, int stride, int act, hipStream_t st, int* rc) {
  const int Wo = (W - 1) / stride + 1;
  if ((Cout % 16) || (Cin % 4) || Wo < 8 || (size_t)Cin * D * H * W >= (size_t)0x7fffffff) return false;
  if (W % 4) {   // unaligned rows: same kernels with dword staging
    if (stride == 1) *rc = launch_mfma<1, 0, true>(x, w, b, skip, out, Cin, Cout, D, H, W, act, st);
    else *rc = launch_mfma<2, 0, true>(x, w, b, skip, out, Cin, Cout, D, H, W, act, st);
    return true;
  }
  if (stride == 1) *rc = launch_mfma<1, 0>(x, w, b, skip, out, Cin, Cout, D, H, W, act, st);
  else if (Wo >= 256 && Wo % 64 == 0) *rc = launch_mfma<2, 1>(x, w, b, skip, out, Cin, Cout, D, H, W, act, st);
  else *rc = launch_mfma<2, 0>(x, w, b, skip, out, Cin, Cout, D, H, W, act, st);
  return true;
}

bool cds_deconv3d_mfma_launch(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin,
                              int Cout, int D, int H, int W, int act, hipStream_t st, int* rc) {
  // Cout % 16 != 0 (conv11, 16 -> 8) stays on the packed-VALU kernels: a (cout, parity) MFMA variant measured slower on this
  // memory-bound layer (2.2 vs 1.5 ms at M1, round 1) and was removed in round 4
  if ((Cout % 16) || (Cin % 4) || (W % 2) || W < 8 ||
      (size_t)Cin * D * H * W >= (size_t)0x7fffffff)
    return false;
  using Cfg = MDCfg;
  const int tx = cds_ceil_div(W, Cfg::CX), ty = cds_ceil_div(H, Cfg::CY), tz = cds_ceil_div(D, Cfg::CZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::SLAB * Cfg::CI_CHUNK * sizeof(float);
  using C4 = MD4Cfg;
  const int tx4 = cds_ceil_div(W, C4::CX), ty4 = cds_ceil_div(H, C4::CY);
  const int nt4 = tx4 * ty4 * D;
  // all four parity classes per workgroup (the input tile is staged once, not four times): 64->32 at 80x64x24 399 -> 339 us,
  // 32->16 at 160x128x48 695 -> 635 us; levels too small to give every CU a workgroup keep the one-class-per-workgroup kernel
  // (50x37x6: 63 vs 81 us)
  if ((long)nt4 * (Cout / 16) >= 256) {
    const size_t lds4 = (size_t)C4::SLAB * C4::CI_CHUNK * sizeof(float);
    if (W % 4)
      hipLaunchKernelGGL(deconv3d_k3s2_mfma4_kernel<true>, dim3(nt4 * (Cout / 16)), dim3(256), lds4, st, x, w, b, skip, out, Cin,
                         Cout, D, H, W, act, tx4, ty4, nt4);
    else
      hipLaunchKernelGGL(deconv3d_k3s2_mfma4_kernel<false>, dim3(nt4 * (Cout / 16)), dim3(256), lds4, st, x, w, b, skip, out,
                         Cin, Cout, D, H, W, act, tx4, ty4, nt4);
    *rc = cds_launch_status();
    return true;
  }
  if (W % 4)
    hipLaunchKernelGGL((deconv3d_k3s2_mfma_kernel<16, true>), dim3(ntiles * (Cout / 16) * 4), dim3(256), lds_bytes, st, x, w, b,
                       skip, out, Cin, Cout, D, H, W, act, tx, ty, tz, ntiles);
  else
    hipLaunchKernelGGL(deconv3d_k3s2_mfma_kernel<16>, dim3(ntiles * (Cout / 16) * 4), dim3(256), lds_bytes, st, x, w, b,
                       skip, out, Cin, Cout, D, H, W, act, tx, ty, tz, ntiles);
  *rc = cds_launch_status();
  return true;
}

// stride-1 3x3x3 convolution with the ci-fastest weight copy (see conv3d_k3_mfma_cl_kernel)
extern "C" int cds_conv3d_k3_cl_f32(const float* x, const float* weight_cl, const float* bias, const float* skip,
                                    float* out, int Cin, int Cout, int D, int H, int W, int act, void* stream) {
  if (!x || !weight_cl || !out || Cin < 16 || (Cin % 16) || Cout < 16 || (Cout % 16) || D < 1 || H < 1 || W < 4 || (W % 4) ||
      (size_t)Cin * D * H * W >= (size_t)0x7fffffff)
    return CDS_EINVAL;
  using Cfg = MClCfg;
  const int tx = cds_ceil_div(W, Cfg::TX), ty = cds_ceil_div(H, Cfg::TY), tz = cds_ceil_div(D, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::NPOS * 16 * sizeof(float);
  static_assert(Cfg::NPOS * 16 * sizeof(float) <= 65536, "LDS tile above 64 KB");
  hipLaunchKernelGGL(conv3d_k3_mfma_cl_kernel, dim3(ntiles * (Cout / 16)), dim3(256), lds_bytes, (hipStream_t)stream, x,
                     weight_cl, bias, skip, out, Cin, Cout, D, H, W, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}

