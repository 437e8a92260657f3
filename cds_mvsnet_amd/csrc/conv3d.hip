// K4: CostRegNet building blocks as direct LDS-tiled 3x3x3 kernels (no im2col).
//   cds_conv3d_k3_f32      Conv3d k3 p1 stride 1|2 (+bias +ReLU +residual)   module.py:80-116
//   cds_deconv3d_k3s2_f32  ConvTranspose3d k3 s2 p1 op1 (+bias +ReLU +residual) module.py:125-160
//
// Scheme: a workgroup owns an output tile and all of a block of CO output channels.  The input
// tile (with halo) of CI_CHUNK input channels is staged in LDS; every thread keeps PX x-adjacent
// voxels x CO channels of accumulators in registers.  Weights are pre-packed [Cin][27][Cout]
// (cout fastest) so the 3*CO weights of one (cin,kz,ky) row are contiguous and wave-uniform: they
// are fetched through the scalar cache into SGPRs and feed v_fmac directly (no LDS / VGPR cost).
// fp32 accumulate with explicit fmaf, order: cin-major, then kz, ky, kx.
#include <stdlib.h>

#include "cds_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// forward conv, stride S in {1,2}
// thread layout LX x LY x LZ (=256 threads), each thread PX outputs along x.
// ---------------------------------------------------------------------------------------------
template <int S, int LX, int LY, int LZ, int PX, int CO, int CI_CHUNK>
struct ConvCfg {
  static constexpr int TX = LX * PX, TY = LY, TZ = LZ;                    // output tile
  static constexpr int IX = (TX - 1) * S + 3, IY = (TY - 1) * S + 3, IZ = (TZ - 1) * S + 3;  // input tile
  static constexpr int IXP = (IX + 3) & ~3;                                // row stride (16-byte rows)
  static constexpr int TILE = IZ * IY * IXP;
  static constexpr int LDS_FLOATS = TILE * CI_CHUNK;
  static constexpr int NIN = (PX - 1) * S + 3;                             // inputs per row per thread
};

template <int S, int LX, int LY, int LZ, int PX, int CO, int CI_CHUNK>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                        const float* __restrict__ bias, const float* __restrict__ skip,
                                                        float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                        int Do, int Ho, int Wo, int act, int tiles_x, int tiles_y,
                                                        int tiles_z, int ntiles) {
  using Cfg = ConvCfg<S, LX, LY, LZ, PX, CO, CI_CHUNK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int co_blocks = Cout / CO;  // host guarantees divisibility (or CO == Cout)
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * CO;

  const int tid = threadIdx.x;
  const int lx = tid % LX, ly = (tid / LX) % LY, lz = tid / (LX * LY);
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY, oz0 = tz_i * Cfg::TZ;
  // input-space origin of the tile (padding 1)
  const int gx0 = ox0 * S - 1, gy0 = oy0 * S - 1, gz0 = oz0 * S - 1;

  float acc[PX][CO];
#pragma unroll
  for (int p = 0; p < PX; ++p)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[p][c] = 0.f;

  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
    __syncthreads();
    // ---- stage: rows of the input tile, one (ci,z,y) row at a time per LX*? threads ----
    const int nrows = CI_CHUNK * Cfg::IZ * Cfg::IY;
    for (int row = tid / 64; row < nrows; row += 4) {  // each wave stages whole rows
      const int ci = row / (Cfg::IZ * Cfg::IY);
      const int rz = (row / Cfg::IY) % Cfg::IZ;
      const int ry = row % Cfg::IY;
      const int gz = gz0 + rz, gy = gy0 + ry;
      const bool row_ok = (ci0 + ci < Cin) && gz >= 0 && gz < D && gy >= 0 && gy < H;
      const float* __restrict__ src = x + (size_t)(ci0 + ci) * vol + (size_t)gz * plane + (size_t)gy * W;
      float* dst = lds + (size_t)ci * Cfg::TILE + (rz * Cfg::IY + ry) * Cfg::IXP;
      for (int i = tid & 63; i < Cfg::IXP; i += 64) {
        const int gx = gx0 + i;
        float v = 0.f;
        if (row_ok && gx >= 0 && gx < W && i < Cfg::IX) v = src[gx];
        dst[i] = v;
      }
    }
    __syncthreads();
    // ---- compute ----
    const int cmax = min(CI_CHUNK, Cin - ci0);
#pragma unroll 1
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wpk + ((size_t)(ci0 + ci) * 27) * Cout + co0;
      const float* tile_ci = lds + (size_t)ci * Cfg::TILE;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* rowp = tile_ci + ((lz * S + kz) * Cfg::IY + (ly * S + ky)) * Cfg::IXP + lx * PX * S;
          float in[Cfg::NIN];
#pragma unroll
          for (int i = 0; i < Cfg::NIN; ++i) in[i] = rowp[i];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
              const float wv = wc[((kz * 3 + ky) * 3 + kx) * Cout + c];
#pragma unroll
              for (int p = 0; p < PX; ++p) acc[p][c] = fmaf(in[p * S + kx], wv, acc[p][c]);
            }
          }
        }
      }
    }
  }

  // ---- epilogue ----
  const int oz = oz0 + lz, oy = oy0 + ly, oxb = ox0 + lx * PX;
  if (oz >= Do || oy >= Ho) return;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float b = bias ? bias[co0 + c] : 0.f;
    const size_t base = (size_t)(co0 + c) * ovol + (size_t)oz * oplane + (size_t)oy * Wo + oxb;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      if (oxb + p < Wo) {
        float v = (act == CDS_ACT_RELU ? fmaxf(acc[p][c] + b, 0.f) : acc[p][c] + b);
        if (skip) v = skip[base + p] + v;
        out[base + p] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// transposed conv k3 s2 p1 op1.  A thread owns one input-space cell a=(az,ay,ax) and produces the
// 2x2x2 output voxels (2a+{0,1}) for CO channels.  Along one axis, output 2a uses (input a, tap 1)
// and output 2a+1 uses (input a+1, tap 0) and (input a, tap 2).
// wpk layout [Cin][27][Cout], tap index (kz*3+ky)*3+kx, taken from PyTorch's [Cin][Cout][3][3][3].
// ---------------------------------------------------------------------------------------------
template <int LX, int LY, int LZ, int CO, int CI_CHUNK>
__global__ __launch_bounds__(256) void deconv3d_k3s2_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ skip, float* __restrict__ out,
                                                            int Cin, int Cout, int D, int H, int W, int act, int tiles_x,
                                                            int tiles_y, int tiles_z, int ntiles) {
  constexpr int IX = LX + 1, IY = LY + 1, IZ = LZ + 1;
  constexpr int TILE = IZ * IY * IX;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int co_blocks = Cout / CO;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * CO;

  const int tid = threadIdx.x;
  const int lx = tid % LX, ly = (tid / LX) % LY, lz = tid / (LX * LY);
  const int ax0 = tx_i * LX, ay0 = ty_i * LY, az0 = tz_i * LZ;

  float acc[8][CO];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[q][c] = 0.f;

  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
    __syncthreads();
    for (int i = tid; i < TILE * CI_CHUNK; i += 256) {
      const int ci = i / TILE;
      int r = i % TILE;
      const int rx = r % IX;
      r /= IX;
      const int ry = r % IY, rz = r / IY;
      const int gz = az0 + rz, gy = ay0 + ry, gx = ax0 + rx;
      float v = 0.f;
      if (ci0 + ci < Cin && gz < D && gy < H && gx < W) v = x[(size_t)(ci0 + ci) * vol + (size_t)gz * plane + (size_t)gy * W + gx];
      lds[i] = v;
    }
    __syncthreads();
    const int cmax = min(CI_CHUNK, Cin - ci0);
#pragma unroll 1
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wpk + ((size_t)(ci0 + ci) * 27) * Cout + co0;
      const float* t = lds + (size_t)ci * TILE + (lz * IY + ly) * IX + lx;
      float in[2][2][2];
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) in[dz][dy][dx] = t[(dz * IY + dy) * IX + dx];
      // per axis: parity 0 -> {(in 0, tap 1)}; parity 1 -> {(in 1, tap 0), (in 0, tap 2)}
#pragma unroll
      for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int q = (pz * 2 + py) * 2 + px;
#pragma unroll
            for (int sz = 0; sz <= pz; ++sz)
#pragma unroll
              for (int sy = 0; sy <= py; ++sy)
#pragma unroll
                for (int sx = 0; sx <= px; ++sx) {
                  const int iz = pz ? 1 - sz : 0, kz = pz ? 2 * sz : 1;
                  const int iy = py ? 1 - sy : 0, ky = py ? 2 * sy : 1;
                  const int ix = px ? 1 - sx : 0, kx = px ? 2 * sx : 1;
                  const float v = in[iz][iy][ix];
#pragma unroll
                  for (int c = 0; c < CO; ++c)
                    acc[q][c] = fmaf(v, wc[((kz * 3 + ky) * 3 + kx) * Cout + c], acc[q][c]);
                }
          }
    }
  }

  const int az = az0 + lz, ay = ay0 + ly, ax = ax0 + lx;
  if (az >= D || ay >= H || ax >= W) return;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float b = bias ? bias[co0 + c] : 0.f;
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const size_t base = (size_t)(co0 + c) * ovol + (size_t)(2 * az + pz) * oplane + (size_t)(2 * ay + py) * Wo + 2 * ax;
        float2 v;
        v.x = acc[(pz * 2 + py) * 2 + 0][c] + b;
        v.y = acc[(pz * 2 + py) * 2 + 1][c] + b;
        if (act == CDS_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
        }
        if (skip) {
          const float2 s = *reinterpret_cast<const float2*>(skip + base);
          v.x = s.x + v.x;
          v.y = s.y + v.y;
        }
        *reinterpret_cast<float2*>(out + base) = v;
      }
  }
}

template <int S, int LX, int LY, int LZ, int PX, int CO, int CI_CHUNK>
int launch_conv(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout, int D,
                int H, int W, int act, hipStream_t st) {
  using Cfg = ConvCfg<S, LX, LY, LZ, PX, CO, CI_CHUNK>;
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY), tz = cds_ceil_div(Do, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::LDS_FLOATS * sizeof(float);
  auto kern = conv3d_k3_kernel<S, LX, LY, LZ, PX, CO, CI_CHUNK>;
  hipLaunchKernelGGL(kern, dim3(ntiles * (Cout / CO)), dim3(256), lds_bytes, st, x, w, b, skip, out, Cin, Cout, D, H, W,
                     Do, Ho, Wo, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}

template <int LX, int LY, int LZ, int CO, int CI_CHUNK>
int launch_deconv(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout,
                  int D, int H, int W, int act, hipStream_t st) {
  const int tx = cds_ceil_div(W, LX), ty = cds_ceil_div(H, LY), tz = cds_ceil_div(D, LZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)(LX + 1) * (LY + 1) * (LZ + 1) * CI_CHUNK * sizeof(float);
  auto kern = deconv3d_k3s2_kernel<LX, LY, LZ, CO, CI_CHUNK>;
  hipLaunchKernelGGL(kern, dim3(ntiles * (Cout / CO)), dim3(256), lds_bytes, st, x, w, b, skip, out, Cin, Cout, D, H, W,
                     act, tx, ty, tz, ntiles);
  return cds_launch_status();
}


// ---------------------------------------------------------------------------------------------
// v2 "pipe" kernels (wide volumes, W % 4 == 0): the input tile starts at an x that is a multiple of 4
// so it is staged with aligned 16-byte loads; every thread owns NSLOT float4 slots of the tile, the
// loads of chunk k+1 are issued before the FMAs of chunk k and land in registers while the chunk is
// computed (global latency hidden behind ~7k cycles of FMAs instead of being serialised per row).
// ---------------------------------------------------------------------------------------------
template <int S, int LX, int LY, int LZ, int PX, int PZ, int CO, int CI_CHUNK>
struct PipeCfg {
  static constexpr int TX = LX * PX, TY = LY, TZ = LZ * PZ;
  static constexpr int OFFX = 3;                                   // tile x origin = S*ox0 - 4, first needed col = 3
  static constexpr int IY = (TY - 1) * S + 3, IZ = (TZ - 1) * S + 3;
  static constexpr int IXP = ((TX - 1) * S + 6 + 3) & ~3;
  static constexpr int Q = IXP / 4;                                // float4 per row
  static constexpr int NS = IZ * IY * Q;                           // float4 per input channel
  static constexpr int TILE = NS * 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
  static constexpr int NIN = (PX - 1) * S + 3;
  static constexpr int NZ = (PZ - 1) * S + 3;                      // input rows along z per thread
};

// PZ = outputs per thread along z (register blocking: an input row read from LDS feeds up to 3 z-outputs).
template <int S, int LX, int LY, int LZ, int PX, int PZ, int CO, int CI_CHUNK>
__global__ __launch_bounds__(256) void conv3d_k3_pipe_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ out,
                                                             int Cin, int Cout, int D, int H, int W, int Do, int Ho,
                                                             int Wo, int act, int tiles_x, int tiles_y, int tiles_z,
                                                             int ntiles) {
  using Cfg = PipeCfg<S, LX, LY, LZ, PX, PZ, CO, CI_CHUNK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / CO;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks);
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * CO;
  const int tid = threadIdx.x;
  const int lx = tid % LX, ly = (tid / LX) % LY, lz = tid / (LX * LY);
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY, oz0 = tz_i * Cfg::TZ;
  const int gx0 = ox0 * S - 4, gy0 = oy0 * S - 1, gz0 = oz0 * S - 1;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  // per-thread staging slots: global element offset (within channel ci0) or -1
  int goff[Cfg::NSLOT];
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = gz0 + rz, gy = gy0 + ry, gx = gx0 + 4 * c4;
    const bool ok = (s < CI_CHUNK * Cfg::NS) && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
    goff[j] = ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      const int ci = s / Cfg::NS;
      // branch-free: always load (from element 0 when the slot is outside); masked when written to LDS
      const bool ok = goff[j] >= 0 && ci0 + ci < Cin;
      pre[j] = *reinterpret_cast<const float4*>(ok ? xb + goff[j] : x);
    }
  };

  float acc[PZ][PX][CO];
#pragma unroll
  for (int z = 0; z < PZ; ++z)
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[z][p][c] = 0.f;

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      const bool ok = goff[j] >= 0 && ci0 + s / Cfg::NS < Cin;
      if (s < CI_CHUNK * Cfg::NS)
        *reinterpret_cast<float4*>(lds + 4 * s) = ok ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ci0 + CI_CHUNK < Cin) issue(ci0 + CI_CHUNK);
    const int cmax = min(CI_CHUNK, Cin - ci0);
#pragma unroll 1
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(((ci0 + ci) * 27) * Cout + co0);
      const float* tile_ci = lds + ci * Cfg::TILE;
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int iz = 0; iz < Cfg::NZ; ++iz) {
          const float* rowp = tile_ci + ((lz * PZ * S + iz) * Cfg::IY + (ly * S + ky)) * Cfg::IXP + lx * PX * S;
          float in[Cfg::NIN];
          if constexpr (S == 1 && PX == 4) {
            // cols 4lx+3 .. 4lx+8: one aligned 16-byte read for the middle four, two dword reads for the ends
            const cds_f4 b = *reinterpret_cast<const cds_f4*>(rowp + 4);
            in[0] = rowp[3]; in[1] = b.x; in[2] = b.y; in[3] = b.z; in[4] = b.w; in[5] = rowp[8];
          } else if constexpr (S == 2 && PX == 2) {
            const cds_f4 b = *reinterpret_cast<const cds_f4*>(rowp + 4);
            in[0] = rowp[3]; in[1] = b.x; in[2] = b.y; in[3] = b.z; in[4] = b.w;
          } else {
#pragma unroll
            for (int i = 0; i < Cfg::NIN; ++i) in[i] = rowp[Cfg::OFFX + i];
          }
#pragma unroll
          for (int z = 0; z < PZ; ++z) {
            const int kz = iz - z * S;  // compile-time after unrolling
            if (kz >= 0 && kz < 3) {
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                  const float wv = wc[((kz * 3 + ky) * 3 + kx) * Cout + c];
#pragma unroll
                  for (int p = 0; p < PX; ++p) acc[z][p][c] = fmaf(in[p * S + kx], wv, acc[z][p][c]);
                }
              }
            }
          }
        }
      }
    }
  }

  const int oy = oy0 + ly, oxb = ox0 + lx * PX;
  if (oy >= Ho || oxb >= Wo) return;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const bool vec = (PX == 4) && ((Wo & 3) == 0);  // oxb is a multiple of 4 -> 16-byte aligned rows
#pragma unroll
  for (int z = 0; z < PZ; ++z) {
    const int oz = oz0 + lz * PZ + z;
    if (oz >= Do) break;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float b = bias ? bias[co0 + c] : 0.f;
      const size_t base = (size_t)(co0 + c) * ovol + (size_t)oz * oplane + (size_t)oy * Wo + oxb;
      float v[PX];
#pragma unroll
      for (int p = 0; p < PX; ++p) v[p] = (act == CDS_ACT_RELU) ? fmaxf(acc[z][p][c] + b, 0.f) : acc[z][p][c] + b;
      if (vec) {
        if constexpr (PX == 4) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (skip) {
            const float4 s4 = *reinterpret_cast<const float4*>(skip + base);
            o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
          }
          *reinterpret_cast<float4*>(out + base) = o;
        }
      } else {
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (oxb + p < Wo) out[base + p] = skip ? skip[base + p] + v[p] : v[p];
      }
    }
  }
}

template <int S, int LX, int LY, int LZ, int PX, int PZ, int CO, int CI_CHUNK>
int launch_conv_pipe(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout,
                     int D, int H, int W, int act, hipStream_t st) {
  using Cfg = PipeCfg<S, LX, LY, LZ, PX, PZ, CO, CI_CHUNK>;
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY), tz = cds_ceil_div(Do, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::TILE * CI_CHUNK * sizeof(float);
  static_assert(Cfg::TILE * CI_CHUNK * sizeof(float) <= 65536, "dynamic LDS above 64 KB needs a function attribute");
  auto kern = conv3d_k3_pipe_kernel<S, LX, LY, LZ, PX, PZ, CO, CI_CHUNK>;
  hipLaunchKernelGGL(kern, dim3(ntiles * (Cout / CO)), dim3(256), lds_bytes, st, x, w, b, skip, out, Cin, Cout, D, H, W,
                     Do, Ho, Wo, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}

template <int CO, int CI_CHUNK>
struct D2Cfg {
  static constexpr int LX = 16, LY = 4, LZ = 4, PC = 4;   // PC cells per thread
  static constexpr int IY = LY + 1, IZ = LZ + 1;
  static constexpr int IXP = (LX * PC + 1 + 3) & ~3;        // 68
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IZ * IY * Q;
  static constexpr int TILE = NS * 4;
  static constexpr int NSLOT = (CI_CHUNK * NS + 255) / 256;
};

template <int CO, int CI_CHUNK>
__global__ __launch_bounds__(256) void deconv3d_k3s2_v2_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ skip, float* __restrict__ out,
                                                               int Cin, int Cout, int D, int H, int W, int act,
                                                               int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  using Cfg = D2Cfg<CO, CI_CHUNK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int co_blocks = Cout / CO;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * co_blocks * 4);
  const int cls = lin & 3;          // parity class: the 4 classes of a tile run together (same input tile in L2)
  lin >>= 2;
  const int cob = lin % co_blocks;
  int tile = lin / co_blocks;
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int co0 = cob * CO;
  const int pz = cls >> 1, py = cls & 1;
  const int tid = threadIdx.x;
  const int lx = tid % Cfg::LX, ly = (tid / Cfg::LX) % Cfg::LY, lz = tid / (Cfg::LX * Cfg::LY);
  const int ax0 = tx_i * Cfg::LX * Cfg::PC, ay0 = ty_i * Cfg::LY, az0 = tz_i * Cfg::LZ;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  int goff[Cfg::NSLOT];
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    const int ci = s / Cfg::NS;
    int r = s - ci * Cfg::NS;
    const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = az0 + rz, gy = ay0 + ry, gx = ax0 + 4 * c4;
    const bool ok = (s < CI_CHUNK * Cfg::NS) && gz < D && gy < H && gx + 3 < W;
    goff[j] = ok ? (int)((size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx) : -1;
  }
  float4 pre[Cfg::NSLOT];
  auto issue = [&](int ci0) {
    const float* __restrict__ xb = x + (size_t)ci0 * vol;
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      const int ci = s / Cfg::NS;
      const bool ok = goff[j] >= 0 && ci0 + ci < Cin;
      pre[j] = *reinterpret_cast<const float4*>(ok ? xb + goff[j] : x);
    }
  };

  float acc[2 * Cfg::PC][CO];  // [output x: 2*cell + px][co]
#pragma unroll
  for (int q = 0; q < 2 * Cfg::PC; ++q)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[q][c] = 0.f;

  issue(0);
  for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Cfg::NSLOT; ++j) {
      const int s = tid + 256 * j;
      const bool ok = goff[j] >= 0 && ci0 + s / Cfg::NS < Cin;
      if (s < CI_CHUNK * Cfg::NS)
        *reinterpret_cast<float4*>(lds + 4 * s) = ok ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (ci0 + CI_CHUNK < Cin) issue(ci0 + CI_CHUNK);
    const int cmax = min(CI_CHUNK, Cin - ci0);
#pragma unroll 1
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wpk + __builtin_amdgcn_readfirstlane(((ci0 + ci) * 27) * Cout + co0);
      const float* t = lds + ci * Cfg::TILE + (lz * Cfg::IY + ly) * Cfg::IXP + lx * Cfg::PC;
      // along z (and y): parity 0 -> (input +0, tap 1); parity 1 -> (input +1, tap 0) and (input +0, tap 2)
      for (int sz = 0; sz <= pz; ++sz) {
        const int iz = pz ? 1 - sz : 0, kz = pz ? 2 * sz : 1;
        for (int sy = 0; sy <= py; ++sy) {
          const int iy = py ? 1 - sy : 0, ky = py ? 2 * sy : 1;
          const float* rowp = t + (iz * Cfg::IY + iy) * Cfg::IXP;
          const cds_f4 b = *reinterpret_cast<const cds_f4*>(rowp);
          const float in[5] = {b.x, b.y, b.z, b.w, rowp[4]};
          const float* __restrict__ wrow = wc + ((kz * 3 + ky) * 3) * Cout;  // taps kx = 0,1,2 of this (kz,ky)
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            const float w0 = wrow[c], w1 = wrow[Cout + c], w2 = wrow[2 * Cout + c];
#pragma unroll
            for (int p = 0; p < Cfg::PC; ++p) {
              acc[2 * p][c] = fmaf(in[p], w1, acc[2 * p][c]);              // x = 2a   : (input a,   tap 1)
              acc[2 * p + 1][c] = fmaf(in[p + 1], w0, acc[2 * p + 1][c]);  // x = 2a+1 : (input a+1, tap 0)
              acc[2 * p + 1][c] = fmaf(in[p], w2, acc[2 * p + 1][c]);      //            (input a,   tap 2)
            }
          }
        }
      }
    }
  }

  const int az = az0 + lz, ay = ay0 + ly, ax = ax0 + lx * Cfg::PC;
  if (az >= D || ay >= H || ax >= W) return;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float b = bias ? bias[co0 + c] : 0.f;
    const size_t base = (size_t)(co0 + c) * ovol + (size_t)(2 * az + pz) * oplane + (size_t)(2 * ay + py) * Wo + 2 * ax;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // W % 4 == 0 and ax % 4 == 0: both halves are fully inside or outside
      if (ax + 2 * half < W) {
        float4 o;
        o.x = acc[4 * half + 0][c] + b;
        o.y = acc[4 * half + 1][c] + b;
        o.z = acc[4 * half + 2][c] + b;
        o.w = acc[4 * half + 3][c] + b;
        if (act == CDS_ACT_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (skip) {
          const float4 s4 = *reinterpret_cast<const float4*>(skip + base + 4 * half);
          o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
        }
        *reinterpret_cast<float4*>(out + base + 4 * half) = o;
      }
    }
  }
}

template <int CO, int CI_CHUNK>
int launch_deconv_v2(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout,
                     int D, int H, int W, int act, hipStream_t st) {
  using Cfg = D2Cfg<CO, CI_CHUNK>;
  const int tx = cds_ceil_div(W, Cfg::LX * Cfg::PC), ty = cds_ceil_div(H, Cfg::LY), tz = cds_ceil_div(D, Cfg::LZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::TILE * CI_CHUNK * sizeof(float);
  auto kern = deconv3d_k3s2_v2_kernel<CO, CI_CHUNK>;
  hipLaunchKernelGGL(kern, dim3(ntiles * (Cout / CO) * 4), dim3(256), lds_bytes, st, x, w, b, skip, out, Cin, Cout, D, H,
                     W, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------
// transposed conv v3 (Cout = 8, Cin <= 16, wide volumes).  The v2 kernel stages the same input tile once per
// (z,y) parity class and per channel chunk (6.6x the input bytes through L2 -> LDS at M1) and leaves the skip-tensor
// reads of the epilogue exposed.  Here a workgroup stages its (16*PC)x4x2 input cells (+1 halo) for ALL input
// channels once, then waves 0-1 produce the parity classes (0,0) and (1,1) (1 + 4 tap pairs) and waves 2-3 the
// classes (0,1) and (1,0) (2 + 2) from the resident tile.  Thread = PC x-adjacent cells -> 2*PC x-adjacent outputs
// x 8 channels.  The skip values of a class are requested before its FMA loop and consumed after it (that alone is
// 1.55 -> 1.2 ms at M1; with 5 GB of compulsory HBM traffic the layer then sits at ~4.3 TB/s).
// PC = 4: 63.75 KB of LDS for Cin = 16 (2 workgroups per CU); PC = 2: 34.5 KB (4 per CU), same speed at M1.
// ---------------------------------------------------------------------------------------------
template <int PC_>
struct D3Cfg {
  static constexpr int LX = 16, LY = 4, LZ = 2, PC = PC_;
  static constexpr int IY = LY + 1, IZ = LZ + 1;
  static constexpr int IXP = (LX * PC + 1 + 3) & ~3;   // 68 / 36
  static constexpr int Q = IXP / 4;
  static constexpr int NS = IZ * IY * Q;               // float4 per channel
  static constexpr int TILE = NS * 4;
  static constexpr int NB = (16 * NS + 255) / 256 > 8 ? 8 : (16 * NS + 255) / 256;   // staging loads in flight
};
constexpr int D3_MAX_CIN = 16;

template <int CO, int PC>
__global__ __launch_bounds__(256) void deconv3d_k3s2_v3_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ skip, float* __restrict__ out,
                                                               int Cin, int Cout, int D, int H, int W, int act,
                                                               int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  using Cfg = D3Cfg<PC>;
  static_assert(PC == 2 || PC == 4, "PC");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int tile = cds_xcd_remap(blockIdx.x, ntiles);
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y;
  const int tz_i = tile / tiles_y;
  const int tid = threadIdx.x;
  const int ax0 = tx_i * Cfg::LX * PC, ay0 = ty_i * Cfg::LY, az0 = tz_i * Cfg::LZ;
  const size_t plane = (size_t)H * W, vol = (size_t)D * plane;

  // ---- stage every input channel of the tile: Cin * NS float4 slots, 256 at a time, loads batched by NB ----
  const int nslots = Cin * Cfg::NS;
  for (int s0 = 0; s0 < nslots; s0 += 256 * Cfg::NB) {
    float4 pre[Cfg::NB];
    bool okv[Cfg::NB];
#pragma unroll
    for (int j = 0; j < Cfg::NB; ++j) {
      const int s = s0 + 256 * j + tid;
      const int ci = s / Cfg::NS;
      const int r = s - ci * Cfg::NS;
      const int row = r / Cfg::Q, c4 = r - row * Cfg::Q;
      const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
      const int gz = az0 + rz, gy = ay0 + ry, gx = ax0 + 4 * c4;
      okv[j] = s < nslots && gz < D && gy < H && gx + 3 < W;
      const size_t off = okv[j] ? (size_t)ci * vol + (size_t)gz * plane + (size_t)gy * W + gx : 0;
      pre[j] = *reinterpret_cast<const float4*>(x + off);
    }
#pragma unroll
    for (int j = 0; j < Cfg::NB; ++j) {
      const int s = s0 + 256 * j + tid;
      if (s < nslots) *reinterpret_cast<float4*>(lds + 4 * s) = okv[j] ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();

  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);   // wave pair
  const int t128 = tid & 127;
  const int lx = t128 % Cfg::LX, ly = (t128 / Cfg::LX) % Cfg::LY, lz = t128 / (Cfg::LX * Cfg::LY);
  const int az = az0 + lz, ay = ay0 + ly, ax = ax0 + lx * PC;
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const float* tbase = lds + (lz * Cfg::IY + ly) * Cfg::IXP + lx * PC;
  const bool inside = az < D && ay < H && ax < W;   // W % 4 == 0 and ax % PC == 0: all 2*PC outputs inside or outside
  constexpr int NV = PC / 2;                           // float4 stores per (channel, thread)

#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    // half 0: (pz,py) = (0,0) then (1,1); half 1: (0,1) then (1,0)
    const int pz = pass, py = half ? 1 - pass : pass;
    float acc[2 * PC][CO];
#pragma unroll
    for (int q = 0; q < 2 * PC; ++q)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[q][c] = 0.f;
    const size_t obase = (size_t)(2 * az + pz) * oplane + (size_t)(2 * ay + py) * Wo + 2 * ax;
    float4 sk[CO][NV];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
      for (int hf = 0; hf < NV; ++hf) {
        const bool ok = skip && inside;
        sk[c][hf] = *reinterpret_cast<const float4*>(ok ? skip + (size_t)c * ovol + obase + 4 * hf : x);
      }
    // Steps = (input channel, tap pair of this parity class), flattened.  (Software-pipelining the LDS row and the
    // 3x8 scalar weight loads one step ahead was measured and does not help: 2+ waves per SIMD already cover them.)
    // along z (and y): parity 0 -> (input +0, tap 1); parity 1 -> (input +1, tap 0) and (input +0, tap 2)
    const int ncombo = (1 << pz) << py;
    const int nsteps = Cin * ncombo;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      const int ci = step / ncombo, combo = step - ci * ncombo;
      const int sy = py ? (combo & 1) : 0, sz = pz ? ((combo >> py) & 1) : 0;
      const int iz = pz ? 1 - sz : 0, kz = pz ? 2 * sz : 1;
      const int iy = py ? 1 - sy : 0, ky = py ? 2 * sy : 1;
      const float* rowp = tbase + ci * Cfg::TILE + (iz * Cfg::IY + iy) * Cfg::IXP;
      float in[PC + 1];
      if constexpr (PC == 4) {
        const cds_f4 b = *reinterpret_cast<const cds_f4*>(rowp);
        in[0] = b.x; in[1] = b.y; in[2] = b.z; in[3] = b.w;
      } else {
        const float2 b = *reinterpret_cast<const float2*>(rowp);
        in[0] = b.x; in[1] = b.y;
      }
      in[PC] = rowp[PC];
      const float* __restrict__ wrow = wpk + __builtin_amdgcn_readfirstlane(((ci * 9 + kz * 3 + ky) * 3) * Cout);
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const float w0 = wrow[c], w1 = wrow[Cout + c], w2 = wrow[2 * Cout + c];
#pragma unroll
        for (int p = 0; p < PC; ++p) {
          acc[2 * p][c] = fmaf(in[p], w1, acc[2 * p][c]);              // x = 2a   : (input a,   tap 1)
          acc[2 * p + 1][c] = fmaf(in[p + 1], w0, acc[2 * p + 1][c]);  // x = 2a+1 : (input a+1, tap 0)
          acc[2 * p + 1][c] = fmaf(in[p], w2, acc[2 * p + 1][c]);      //            (input a,   tap 2)
        }
      }
    }
    if (inside) {
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const float b = bias ? bias[c] : 0.f;
        const size_t base = (size_t)c * ovol + obase;
#pragma unroll
        for (int hf = 0; hf < NV; ++hf) {
          float4 o;
          o.x = acc[4 * hf + 0][c] + b;
          o.y = acc[4 * hf + 1][c] + b;
          o.z = acc[4 * hf + 2][c] + b;
          o.w = acc[4 * hf + 3][c] + b;
          if (act == CDS_ACT_RELU) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
          }
          if (skip) {
            const float4 s4 = sk[c][hf];
            o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
          }
          *reinterpret_cast<float4*>(out + base + 4 * hf) = o;
        }
      }
    }
  }
}

template <int PC>
int launch_deconv_v3(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin, int Cout,
                     int D, int H, int W, int act, hipStream_t st) {
  using Cfg = D3Cfg<PC>;
  const int tx = cds_ceil_div(W, Cfg::LX * PC), ty = cds_ceil_div(H, Cfg::LY), tz = cds_ceil_div(D, Cfg::LZ);
  const int ntiles = tx * ty * tz;
  const size_t lds_bytes = (size_t)Cfg::TILE * Cin * sizeof(float);
  hipLaunchKernelGGL((deconv3d_k3s2_v3_kernel<8, PC>), dim3(ntiles), dim3(256), lds_bytes, st, x, w, b, skip, out, Cin,
                     Cout, D, H, W, act, tx, ty, tz, ntiles);
  return cds_launch_status();
}

}  // namespace

bool cds_conv3d_mfma_launch(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin,
                            int Cout, int D, int H, int W, int stride, int act, hipStream_t st, int* rc);

bool cds_deconv3d_mfma_launch(const float* x, const float* w, const float* b, const float* skip, float* out, int Cin,
                              int Cout, int D, int H, int W, int act, hipStream_t st, int* rc);

extern "C" int cds_conv3d_k3_f32(const float* x, const float* weight, const float* bias, const float* skip, float* out,
                                 int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream) {
  if (!x || !weight || !out || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2))
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int Wo = (W - 1) / stride + 1;
  const bool wide = Wo >= 48;
  // total elements must fit the 32-bit staging offsets of the pipe kernels
  const bool pipe_ok = (W % 4 == 0) && Wo >= 32 && ((size_t)Cin * D * H * W < (size_t)0x7fffffff);
  const bool no_mfma = cds_env_is("CDS_CONV_NO_MFMA", '1');
  if (!no_mfma) {
    int rc = 0;
    if (cds_conv3d_mfma_launch(x, weight, bias, skip, out, Cin, Cout, D, H, W, stride, act, st, &rc)) return rc;
  }
  if (pipe_ok && Cout % 8 == 0) {
    if (stride == 1) {
      // one input channel per staged chunk (10 KB of LDS, fewer staging registers -> more resident waves): conv0 at M1
      // 2810 us with 4 channels per chunk, 2610 with 2, 2370 with 1 (round-1 A/B; the other blockings were removed in round 4)
      return launch_conv_pipe<1, 16, 4, 4, 4, 1, 8, 1>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
    }
    return launch_conv_pipe<2, 16, 4, 4, 2, 1, 8, 2>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  if (pipe_ok && Cout == 1 && stride == 1) {
    // Cout = 1 (prob): 4 z-outputs per thread so every LDS row read feeds up to 3 outputs (LDS-bound otherwise)
    // one input channel per staged chunk: 31 KB of LDS -> 5 workgroups per CU (2 with two channels); 898 -> 627 us at M1
    return launch_conv_pipe<1, 16, 4, 4, 4, 4, 1, 1>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  if (Cout == 1) {
    if (stride != 1) return CDS_EINVAL;
    return wide ? launch_conv<1, 16, 4, 4, 4, 1, 4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st)
                : launch_conv<1, 4, 8, 8, 4, 1, 4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  if (Cout % 8) return CDS_EINVAL;
  if (stride == 1) {
    return wide ? launch_conv<1, 16, 4, 4, 4, 8, 4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st)
                : launch_conv<1, 4, 8, 8, 4, 8, 4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  return wide ? launch_conv<2, 16, 4, 4, 2, 8, 2>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st)
              : launch_conv<2, 4, 8, 8, 2, 8, 2>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
}

extern "C" int cds_deconv3d_k3s2_f32(const float* x, const float* weight, const float* bias, const float* skip,
                                     float* out, int Cin, int Cout, int D, int H, int W, int act, void* stream) {
  if (!x || !weight || !out || Cin < 1 || Cout < 1 || (Cout % 8) || D < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  {
    const bool no_mfma = cds_env_is("CDS_CONV_NO_MFMA", '1');
    int rc = 0;
    if (!no_mfma && cds_deconv3d_mfma_launch(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st, &rc)) return rc;
  }
  if ((W % 4 == 0) && W >= 32 && ((size_t)Cin * D * H * W < (size_t)0x7fffffff)) {
    if (Cout == 8 && Cin <= D3_MAX_CIN) return launch_deconv_v3<4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
    return launch_deconv_v2<8, 4>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  return W >= 48 ? launch_deconv<64, 2, 2, 8, 8>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st)
                 : launch_deconv<16, 4, 4, 8, 8>(x, weight, bias, skip, out, Cin, Cout, D, H, W, act, st);
}
