// 3x3 convolution 16 -> 16 channels on the matrix cores (the two inner layers of the visibility CNN, model.py:14;
// SURVEY K2), optionally followed in the same kernel by the CNN's 1x1 head (16 -> 1, sigmoid).
//
// The VALU kernel of conv2d.hip runs these layers at ~37 TF (144 scalar weight loads per input channel do not fit the
// SGPR file); here one workgroup stages its 64 x 8 pixel tile + halo once, channels-last ([pos][16 ci], 16-byte slots
// XOR-swizzled like conv3d_k3_mfma_cl_kernel), holds all 9 x 16 x 16 weights in 36 VGPRs per lane, and each lane fetches
// ci = 4k..4k+3 of its pixel with one ds_read_b128 = the A operands of four v_mfma_f32_16x16x4_f32:
//   D[m = pixel][n = cout] += A[m][k] * B[k][n];   72 LDS reads per 288 MFMAs per wave.
// C/D layout: lane l holds cout (l & 15), pixels 4 (l >> 4) + 0..3 of the 16-pixel run.
#include "cds_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CDS_V16_RW
#define CDS_V16_RW 2   // tile rows per wave (A/B: 1 -> 64x4 tile, 28 KB of LDS; 4 -> 64x16 tile, 83 KB)
#endif
struct V16Cfg {
  static constexpr int RW = CDS_V16_RW;              // tile rows per wave
  static constexpr int TX = 64, TY = 4 * RW, XT = TX / 16;
  static constexpr int NT = XT * RW;                 // M-tiles per wave
  static constexpr int IY = TY + 2;
  static constexpr int IXP = TX + 8;                 // column c <-> x = ox0 - 4 + c (16-byte aligned global rows)
  static constexpr int Q = IXP / 4;
  static constexpr int NPOS = IY * IXP;              // 720 positions x 64 B = 45 KB
  static constexpr int NSLOTS = IY * Q * 4;          // (row, x-group, k-group) staging slots
  static constexpr int NSLOT = (NSLOTS + 255) / 256;
};

// sum over the 16 lanes of a DPP row (all lanes end up with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// HEAD = false: out [N][16][H][W] = act(conv + bias).  HEAD = true: out [N][H][W] = sigmoid(head_b + sum_c head_w[c] *
// act(conv + bias)[c]).
template <bool HEAD>
__global__ __launch_bounds__(256, 2) void conv2d_k3_c16_mfma_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ wcl,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ head_w,
                                                                     const float* __restrict__ head_b,
                                                                     float* __restrict__ out, int N, int H, int W, int act,
                                                                     int tiles_x, int ntiles) {
  using Cfg = V16Cfg;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lin = cds_xcd_remap(blockIdx.x, ntiles * N);
  const int tile = lin % ntiles, n_img = lin / ntiles;
  const int tx_i = tile % tiles_x, ty_i = tile / tiles_x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY;
  const int gx0 = ox0 - 4, gy0 = oy0 - 1;
  const size_t plane = (size_t)H * W;
  const float* __restrict__ xn = x + (size_t)n_img * 16 * plane;

  // ---- staging: slot = (row, x-group q, k-group kg): 4 channels x 4 x-positions, transposed in registers ----
#pragma unroll
  for (int j = 0; j < Cfg::NSLOT; ++j) {
    const int s = tid + 256 * j;
    if (s < Cfg::NSLOTS) {
      const int kg = s & 3;
      const int q = (s >> 2) % Cfg::Q;
      const int row = (s >> 2) / Cfg::Q;
      const int gy = gy0 + row, gx = gx0 + 4 * q;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;   // W % 4 == 0: a group is inside or outside as a whole
      float4 pre[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) pre[i] = *reinterpret_cast<const float4*>(xn + (size_t)(4 * kg + i) * plane + (size_t)gy * W + gx);
      }
      const int pos = row * Cfg::IXP + 4 * q;
      const int l0 = pos * 16 + ((kg ^ ((2 * q) & 3)) << 2);
      const int l1 = (pos + 2) * 16 + ((kg ^ ((2 * q + 1) & 3)) << 2);
      *reinterpret_cast<float4*>(lds + l0) = make_float4(pre[0].x, pre[1].x, pre[2].x, pre[3].x);
      *reinterpret_cast<float4*>(lds + l0 + 16) = make_float4(pre[0].y, pre[1].y, pre[2].y, pre[3].y);
      *reinterpret_cast<float4*>(lds + l1) = make_float4(pre[0].z, pre[1].z, pre[2].z, pre[3].z);
      *reinterpret_cast<float4*>(lds + l1 + 16) = make_float4(pre[0].w, pre[1].w, pre[2].w, pre[3].w);
    }
  }

  // ---- B: w_cl[tap][co][ci], lane (n = l & 15, k = l >> 4) holds ci = 4k .. 4k+3 of output channel n, all 9 taps ----
  const int m = lane & 15, kq = lane >> 4;
  f32x4 bw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) bw[t] = *reinterpret_cast<const f32x4*>(wcl + ((size_t)t * 16 + m) * 16 + 4 * kq);

  // ---- lane-constant A addresses: pixel m of run txr, tap column kx; row offsets are immediates ----
  const float* a_ptr[3][Cfg::XT];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int txr = 0; txr < Cfg::XT; ++txr) {
      const int col = 3 + txr * 16 + m + kx;
      a_ptr[kx][txr] = lds + ((wave * Cfg::RW) * Cfg::IXP + col) * 16 + ((kq ^ ((col >> 1) & 3)) << 2);
    }

  f32x4 acc[Cfg::NT];
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f32x4 bv = bw[ky * 3 + kx];
#pragma unroll
      for (int t = 0; t < Cfg::NT; ++t) {
        const int ry = t / Cfg::XT, txr = t % Cfg::XT;
        const f32x4 av = *reinterpret_cast<const f32x4*>(a_ptr[kx][txr] + (ry + ky) * Cfg::IXP * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue ----
  const float bn = bias ? bias[m] : 0.f;
  const float hw_n = HEAD ? head_w[m] : 0.f;
  const float hb = HEAD ? head_b[0] : 0.f;
#pragma unroll
  for (int t = 0; t < Cfg::NT; ++t) {
    const int ry = t / Cfg::XT, txr = t % Cfg::XT;
    const int oy = oy0 + wave * Cfg::RW + ry, ox = ox0 + txr * 16 + 4 * kq;
    float v[4] = {acc[t].x + bn, acc[t].y + bn, acc[t].z + bn, acc[t].w + bn};
    if (act == CDS_ACT_RELU) {
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = fmaxf(v[p], 0.f);
    }
    if (HEAD) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float s = row16_sum(v[p] * hw_n) + hb;
        v[p] = 1.0f / (1.0f + expf(-s));
      }
      if (m == 0 && oy < H && ox < W)
        *reinterpret_cast<float4*>(out + (size_t)n_img * plane + (size_t)oy * W + ox) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (oy < H && ox < W) {
      *reinterpret_cast<float4*>(out + ((size_t)n_img * 16 + m) * plane + (size_t)oy * W + ox) =
          make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace

extern "C" int cds_conv2d_k3_c16_f32(const float* x, const float* weight_cl, const float* bias, const float* head_w,
                                     const float* head_b, float* out, int N, int H, int W, int act, void* stream) {
  if (!x || !weight_cl || !out || N < 1 || H < 1 || W < 4 || (W % 4) || (act != CDS_ACT_NONE && act != CDS_ACT_RELU) ||
      (head_w != nullptr) != (head_b != nullptr))
    return CDS_EINVAL;
  using Cfg = V16Cfg;
  const int tx = cds_ceil_div(W, Cfg::TX), ty = cds_ceil_div(H, Cfg::TY);
  const int ntiles = tx * ty;
  const size_t lds_bytes = (size_t)Cfg::NPOS * 16 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (head_w)
    hipLaunchKernelGGL(conv2d_k3_c16_mfma_kernel<true>, dim3(ntiles * N), dim3(256), lds_bytes, st, x, weight_cl, bias, head_w,
                       head_b, out, N, H, W, act, tx, ntiles);
  else
    hipLaunchKernelGGL(conv2d_k3_c16_mfma_kernel<false>, dim3(ntiles * N), dim3(256), lds_bytes, st, x, weight_cl, bias, head_w,
                       head_b, out, N, H, W, act, tx, ntiles);
  return cds_launch_status();
}
