// FeatureNet on CHANNELS-LAST activations (round 5; models/module.py:234-267, models/dynamic_conv.py:97-122).
//
// Every FeatureNet activation is [N][H][W][C] fp32: a pixel's channels are contiguous (32 / 64 / 128-byte texels for C = 8 / 16 /
// 32), a staged tile row is ONE contiguous, 32-byte aligned run of (tile width + 2 halo) * C floats instead of C row segments of
// 160 bytes from C channel planes (what the planar kernels of conv2d_sbf.hip waited for: memory-system throughput on short
// segments, profiles/r04_experiments.md).  K1 / K3 gather source features channels-last already, so the stage outputs need no
// transposition pass.
//
//   cds_dynconv_cl_f32        one DynamicConv in one kernel: all branch convolutions (conv_k and the 3 curvature channels att_k
//                             of every kernel size k) as implicit GEMMs on the bf16 matrix cores in split-bf16 arithmetic
//                             (sbf_common.hpp: 3 bf16 terms per fp32 operand, 6 partial products, fp32 accumulate) + the blend
//                             epilogue (epipolar projection, 1x1 MLP, softmax(./T), blend) on the accumulators + the InstanceNorm
//                             records of the result; the K-loop and epilogue of the planar kernel (conv2d_sbf.hip), the blended
//                             16 x 16 tiles transposed through LDS for 16-byte channels-last stores (why not the transposed GEMM:
//                             see the kernel's comment - it corrupts other waves).
//   cds_dynconv_blend_cl_f32  the same epilogue over a planar branch tensor (conv00: 3 input channels, VALU branch kernels),
//                             channels-last output
//   cds_conv2d_k3s2_cl_f32    the two down-sampling units (3x3, stride 2, pad 1), VALU, texel loads straight from global memory
//   cds_conv2d_fpn_cl_f32     FPN lateral: 1x1 convolution over cat(nearest2x(coarse), skip), neither materialised
//   cds_instnorm_stats_cl_f32 / cds_instnorm_apply_cl_f32   InstanceNorm statistics records / normalise + activation, with the
//                             reference-view rows also written planar [C][h][w] (what K1 / K3 read per reference pixel)
// Normalise-on-load everywhere: a layer output travels as (raw, affine [N][C][3] = 1/std, -mean/std, leaky slope).
#include "sbf_common.hpp"
#include "feat_common.hpp"

namespace {

constexpr int TX = 32, TY = 8;   // output tile of a workgroup: wave w owns rows 2w, 2w + 1 = four 16-pixel MFMA column tiles

struct DynEpi {
  const float* w1;      // [4][K]
  const float* b1;      // [4]
  const float* w2;      // [K][4]
  float* out;           // [N][H][W][C]
  float* norm_curv;     // [N][H][W]
  double* partial;      // [N][tiles][C][2]
  float temperature;
  const float* epi;     // [N][2] DEVICE: epipoles in pixels of this resolution (feat_common.hpp)
  float xs, omul;       // split-f16 kernels: power-of-two scale of the (normalised) input, accumulator multiplier 1 / (xs w_scale)
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int nks_of(int k) { return (k * k + 3) / 4; }

// MODE 1: one DynamicConv (NBR kernel sizes, CIN + 3 columns each, blend epilogue); MODE 2: ONE 3 x 3 convolution CIN -> 16 with bias +
// ReLU, optionally followed by a 1x1 head 16 -> 1 + sigmoid (the visibility CNN's layers 2 / 3, models/model.py:14)
template <int CIN, int K0, int K1, int K2, int MODE = 1>
struct DCfg {
  static constexpr int NBR = MODE == 2 ? 1 : (K2 > 0 ? 3 : 2);
  static constexpr int R = (cmax(K0, cmax(K1, K2)) - 1) / 2;
  static constexpr int IXP = TX + 2 * R, IY = TY + 2 * R, NPOS = IXP * IY;
  static constexpr int ROUNDS = CIN / 8, PLANE = NPOS * POSB;
  static constexpr int COUT = MODE == 2 ? 16 : CIN;
  static constexpr int CO3 = MODE == 2 ? 16 : CIN + 3, NBLK = (CO3 + 15) / 16;
  static constexpr int NCB = COUT >= 16 ? COUT / 16 : 1;      // blocks that hold output channels
  static constexpr int C4 = CIN / 4;                           // float4 chunks per texel
  static constexpr int NCH = NPOS * C4, NIT = (NCH + 255) / 256, PSTEP = 256 / C4;
  static constexpr int NKS = nks_of(K0) + (K1 > 0 ? nks_of(K1) : 0) + (K2 > 0 ? nks_of(K2) : 0);
  // epilogue areas (they reuse the staged tile): curvature responses [b][j][256 px] and blend weights [b][256 px] floats, then the
  // transposition tiles [wave][q][block][16 px][TP] floats and the statistics [wave][NBLK * 16][2] doubles
  static constexpr int TP = 20;                                // row pitch of a transposition tile: conflict-free writes and 16-byte reads
  static constexpr int WREG = cmax(NBR * 4 * 64 * 4, 4 * NCB * 16 * TP * 4);   // per-wave epilogue region: curvature + weights | tiles
  static constexpr int TRB = 4 * 4 * NCB * 16 * TP * 4;
  static constexpr int REDB = 4 * NBLK * 16 * 2 * 8;
  static constexpr int LDSB = cmax(ROUNDS * PLANE, 4 * WREG + REDB);
};

// One DynamicConv on channels-last activations.  Staging: channels-last (contiguous tile rows).  K-loop: exactly the planar kernel's
// (conv2d_sbf.hip): A operand = data from LDS (rows = 16 pixels of an x-run), B operand = weights (columns = output channels), so a
// lane ends with ONE output channel of 4 consecutive pixels.  Epilogue: the planar kernel's (curvature columns and per-pixel blend
// weights through LDS, statistics per channel column) - then each wave transposes its blended 16 x 16 tiles through LDS so that a lane
// stores 4 consecutive channels of one pixel (16 bytes, channels-last).
//
// Why not the transposed GEMM (rows = output channels, weights as the A operand), which needs no transposition and keeps a pixel's three
// curvature responses in one lane?  It was built first and is ~5 % faster - and it CORRUPTS OTHER WAVES: with the weights (loaded from
// global memory) as SrcA and the LDS-loaded data as SrcB of v_mfma_f32_16x16x32_bf16, waves of ANY kernel sharing the CU (other
// waves of this kernel, K1 on another stream) occasionally get wrong values in lanes 48-63 (about one 16-pixel tile in 10^4); with the
// two source operands exchanged - nothing else changed - the effect is gone.  Measured by an aggressor / victim experiment and bisected to
// the K-loop (profiles/r05_experiments.md, scripts/ab/r05_aggressor.py); operand data, zero padding, wait states after the MFMAs and the
// distance to the next loads make no difference.  The operand roles below are the ones that have been bit-stable for three rounds.
// The DynamicConv epilogue on the accumulators of one 32 x 8 tile (the planar kernel's, conv2d_sbf.hip MODE 1).  Accumulator layout: lane
// (m, g) holds column 16 nb + m of pixels x = (q & 1) 16 + 4 g + i, y = 2 wave + (q >> 1).  (1) the lanes of the three curvature columns
// leave them in LDS; (2) lane m of a 16-lane group owns pixel (q, i) = (m >> 2, m & 3) of its group: projection, MLP, softmax -> K weights
// into LDS, norm_curv to memory; (3) every lane reads the weights of its 16 pixels and blends its column; (4) the blended 16 x 16 tiles are
// transposed through LDS: a lane stores 4 consecutive channels of one pixel (16 bytes, channels-last).  InstanceNorm records per (wave,
// channel) as the planar kernel leaves them.  `img` = the OUTPUT image (its epipole, its rows of out / norm_curv / partial): conv00 calls
// this once per reference copy on the same accumulators.  Starts with a workgroup barrier (the LDS areas alias the staged tile).
// order this wave's LDS writes before its later LDS reads for the compiler (the LDS executes one wave's instructions in order)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class C>
__device__ __forceinline__ void dyn_epilogue(f32x4 (&acc)[C::NBR][C::NBLK][4], unsigned char* lds, const float* __restrict__ bias,
                                             const DynEpi& ep, int img, int H, int W, int ox0, int oy0, int wave, int m, int g, int tid,
                                             int tile, int parts, const float* mlp_w1, const float* mlp_b1, const float* mlp_w2,
                                             float epi_x, float epi_y) {
  // mlp_* / epi_*: the attention MLP and this image's epipole, handed in by the caller: conv00 runs this epilogue once per reference copy
  // and keeps them in REGISTERS across the copies - read from memory inside, every later epilogue's loads were vector loads behind the
  // previous one's output stores, and on gfx9's single vmcnt the wait for them drained those stores (round 6: ~4 000 cycles per epilogue)
  constexpr int NBR = C::NBR, NBLK = C::NBLK, NCB = C::NCB, Cout = C::COUT, Co3 = C::CO3, CIN = C::COUT;
  // every exchange below stays inside ONE wave (a wave blends the 64 pixels it convolved): each wave has its own LDS region and orders
  // its writes and reads with wave-level fences; only the tile hand-over and the four waves' statistics need workgroup barriers
  float* attL = reinterpret_cast<float*>(lds + wave * C::WREG);    // [b][j][64 pixels of the wave]
  float* wL = attL + NBR * 3 * 64;                                 // [b][64]
  __syncthreads();                                             // every wave is done with the staged input tile
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) {
    const int jj = nb * 16 + m - Cout;
    if (jj < 0 || jj > 2) continue;
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
      const float bv = bias ? bias[b * Co3 + nb * 16 + m] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a = acc[b][nb][q];
        *reinterpret_cast<float4*>(attL + (b * 3 + jj) * 64 + q * 16 + g * 4) =
            make_float4(a.x + bv, a.y + bv, a.z + bv, a.w + bv);
      }
    }
  }
  wave_lds_fence();
  {
    const int wp = (m >> 2) * 16 + g * 4 + (m & 3);
    const int px = ox0 + ((m >> 2) & 1) * 16 + g * 4 + (m & 3), py = oy0 + wave * 2 + (m >> 3);
    float att[NBR][3], logit[NBR];
#pragma unroll
    for (int b = 0; b < NBR; ++b)
#pragma unroll
      for (int j = 0; j < 3; ++j) att[b][j] = attL[(b * 3 + j) * 64 + wp];
#ifndef CDS_PROBE_EPI
#define CDS_PROBE_EPI 0      // probe builds of the epilogue (wrong results, right timing): 1 = no fp64 statistics, 2 = no per-pixel blend weights
#endif
#if CDS_PROBE_EPI == 2
    float nc = att[0][0];
#pragma unroll
    for (int b = 0; b < NBR; ++b) logit[b] = att[b][1];
#else
    const float nc = blend_from_att<NBR>(att, px, py, epi_x, epi_y, mlp_w1, mlp_b1, mlp_w2, ep.temperature, logit);
#endif
#pragma unroll
    for (int b = 0; b < NBR; ++b) wL[b * 64 + wp] = logit[b];
    if (px < W && py < H && CDS_PROBE_EPI != 3) ep.norm_curv[((size_t)img * H + py) * W + px] = nc;
  }
  wave_lds_fence();
  float4 wq[NBR][4];
#pragma unroll
  for (int b = 0; b < NBR; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[b][q] = *reinterpret_cast<const float4*>(wL + b * 64 + q * 16 + g * 4);
  wave_lds_fence();                                            // the weights are in registers: the wave's region becomes its transposition tiles
  float* trL = reinterpret_cast<float*>(lds + wave * C::WREG);                  // [q][block][16 pixels][TP]
  double* red = reinterpret_cast<double*>(lds + 4 * C::WREG);  // [wave][NBLK * 16][2]: the four waves' sums of a tile, added below
#pragma unroll
  for (int nb = 0; nb < NCB; ++nb) {
    const int co = nb * 16 + m;
    const bool col = co < Cout;
    float bvb[NBR];
#pragma unroll
    for (int b = 0; b < NBR; ++b) bvb[b] = (bias && col) ? bias[b * Co3 + co] : 0.f;
    double ds = 0.0, dq = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + g * 4;
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sacc = 0.f;
#pragma unroll
        for (int b = 0; b < NBR; ++b) {
          const f32x4 a = acc[b][nb][q];
          const float av = (i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w) + bvb[b];
          const float wv = i == 0 ? wq[b][q].x : i == 1 ? wq[b][q].y : i == 2 ? wq[b][q].z : wq[b][q].w;
          sacc = sacc + av * wv;
        }
        o[i] = sacc;
        trL[((q * NCB + nb) * 16 + g * 4 + i) * C::TP + m] = sacc;      // pixel row 4 g + i, channel column m
      }
      if (col && oy < H && CDS_PROBE_EPI != 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ox + i < W) {
            const double dv = (double)o[i];
            ds += dv;
            dq += dv * dv;
          }
      }
    }
    // the wave's 64 pixels of this channel: lanes (m, g = 0..3)
    ds += __shfl_xor(ds, 16);
    dq += __shfl_xor(dq, 16);
    ds += __shfl_xor(ds, 32);
    dq += __shfl_xor(dq, 32);
    if (g == 0) {
      red[(wave * NBLK * 16 + co) * 2] = ds;
      red[(wave * NBLK * 16 + co) * 2 + 1] = dq;
    }
  }
  wave_lds_fence();
  // channels-last store: lane (m, g) takes pixel m of each x-run and channels 4 g .. 4 g + 3 of each block: 16 bytes
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + m;
    if (oy >= H || ox >= W || CDS_PROBE_EPI == 3) continue;
    float* __restrict__ op = ep.out + (((size_t)img * H + oy) * W + ox) * CIN;
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) {
      if (nb * 16 + 4 * g < Cout)
        *reinterpret_cast<float4*>(op + nb * 16 + 4 * g) = *reinterpret_cast<const float4*>(trL + ((q * NCB + nb) * 16 + m) * C::TP + 4 * g);
    }
  }
  __syncthreads();
  // one record per (tile, channel): the four waves in a fixed order
  if (tid < NBLK * 16 && tid < Cout) {
    double* rec = ep.partial + (((size_t)img * parts + (size_t)tile) * Cout + tid) * 2;
    rec[0] = (red[tid * 2] + red[(NBLK * 16 + tid) * 2]) + (red[(2 * NBLK * 16 + tid) * 2] + red[(3 * NBLK * 16 + tid) * 2]);
    rec[1] = (red[tid * 2 + 1] + red[(NBLK * 16 + tid) * 2 + 1]) + (red[(2 * NBLK * 16 + tid) * 2 + 1] + red[(3 * NBLK * 16 + tid) * 2 + 1]);
  }
}

// sum over the 16 lanes of a DPP row (all lanes end up with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// F16: split-f16 arithmetic (sbf_common.hpp: two fp16 terms of value x power-of-two scale, three products per K-step instead of six).
// The input scale needs no measured bound here: a DynamicConv's input is InstanceNorm-ed on load, and n samples normalised by their own
// mean and standard deviation are bounded by sqrt(n - 1) (Samuelson), LeakyReLU only shrinks; a bound that loose (~100x the real
// maximum) moves the absolute error floor of the low term to B 2^-40 = 1e-9 of a unit-scale activation - far below an fp32 ulp.
template <int CIN, int K0, int K1, int K2, int MODE, bool F16 = false>
__global__ __launch_bounds__(256, 2) void dynconv_cl_kernel(const float* __restrict__ x, const float* __restrict__ affine,
                                                            const uint4* __restrict__ wsp, const float* __restrict__ bias,
                                                            DynEpi ep, int N, int H, int W, int tiles_x, int tiles_y) {
  using C = DCfg<CIN, K0, K1, K2, MODE>;
  constexpr int NBR = C::NBR, NBLK = C::NBLK, R = C::R, IXP = C::IXP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int lin = cds_xcd_remap(blockIdx.x, tiles_x * tiles_y * N);
  const int tx_i = lin % tiles_x;
  lin /= tiles_x;
  const int ty_i = lin % tiles_y, img = lin / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int ox0 = tx_i * TX, oy0 = ty_i * TY;

  // ---- stage the (TY + 2R) x (TX + 2R) texel tile, all channels: thread = float4 chunk of the tile's contiguous rows, so the
  // 64 lanes of a load instruction read 1 KB of consecutive bytes.  A thread's channel quad is the same in every iteration
  // (256 % C4 == 0): its 12 normalise-on-load constants are loaded once.  Exact 3-way bf16 split in registers. ----
  {
    const int c4 = tid % C::C4;
    float al[4], be[4], sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      al[j] = 1.f; be[j] = 0.f; sl[j] = 1.f;
    }
    if (affine) {
      const float* __restrict__ af = affine + ((size_t)img * CIN + c4 * 4) * 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        al[j] = af[3 * j]; be[j] = af[3 * j + 1]; sl[j] = af[3 * j + 2];
      }
    }
    float4 v[C::NIT];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
      const int pos = tid / C::C4 + i * C::PSTEP;
      const int row = pos / IXP, col = pos - row * IXP;
      const int gy = oy0 - R + row, gx = ox0 - R + col;
      const bool ok = pos < C::NPOS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        v[i] = *reinterpret_cast<const float4*>(x + (((size_t)img * H + gy) * W + gx) * CIN + c4 * 4);
        okmask |= 1u << i;
      }
    }
    unsigned char* dst0 = lds + (c4 >> 1) * C::PLANE + (c4 & 1) * 8;
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
      const int pos = tid / C::C4 + i * C::PSTEP;
      if (pos >= C::NPOS) break;
      float4 t = v[i];
      if (okmask & (1u << i)) {            // zero padding follows the normalisation: out-of-image texels stay zero
        float a;
        a = fmaf(t.x, al[0], be[0]); t.x = a > 0.f ? a : a * sl[0];
        a = fmaf(t.y, al[1], be[1]); t.y = a > 0.f ? a : a * sl[1];
        a = fmaf(t.z, al[2], be[2]); t.z = a > 0.f ? a : a * sl[2];
        a = fmaf(t.w, al[3], be[3]); t.w = a > 0.f ? a : a * sl[3];
      }
      uint32_t h0, m0, l0, h1, m1, l1;
      unsigned char* d = dst0 + pos * POSB;
      if (F16) {
        split2_f16(t.x, t.y, ep.xs, h0, m0);
        split2_f16(t.z, t.w, ep.xs, h1, m1);
      } else {
        split2(t.x, t.y, h0, m0, l0);
        split2(t.z, t.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(l0, l1);
      }
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + 16) = make_uint2(m0, m1);
    }
  }
  __syncthreads();

  f32x4 acc[NBR][NBLK][4];       // M-tiles: (row 0 | 1 of the wave) x (x-run 0 | 1)
#pragma unroll
  for (int b = 0; b < NBR; ++b)
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[b][nb][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A operand (data): lane (m, g) supplies the 8 channels of tap 4 t + g for pixel m of the x-run = one ds_read_b128 per term
  const int a_base = ((wave * 2) * IXP + m) * POSB;
  const uint4* __restrict__ wl = wsp + lane;
  constexpr int KS[3] = {K0, K1 > 0 ? K1 : 1, K2 > 0 ? K2 : 1};
#pragma unroll 1
  for (int rd = 0; rd < C::ROUNDS; ++rd) {
    const uint4* __restrict__ wr = wl + (size_t)rd * C::NKS * NBLK * 3 * 64;
    const unsigned char* __restrict__ lp = lds + rd * C::PLANE + a_base;
    int ks0 = 0;
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
      const int k = KS[b], kk = k * k, rk = (k - 1) >> 1, nks = (kk + 3) >> 2;
      const uint4* __restrict__ wb = wr + (size_t)ks0 * NBLK * 3 * 64;
#pragma unroll 1
      for (int t = 0; t < nks; ++t) {
        int tap = 4 * t + g;
        if (tap >= kk) tap = kk - 1;                    // padded tap: zero weights, any in-tile data
        const int ky = tap / k, kx = tap - ky * k;
        const unsigned char* ap = lp + ((ky + R - rk) * IXP + (kx + R - rk)) * POSB;
        BV wh[NBLK], wm[NBLK], wlo[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
          const uint4* p = wb + (size_t)((t * NBLK + nb) * 3) * 64;
          wh[nb].u = p[0];
          wm[nb].u = p[64];
          if (!F16) wlo[nb].u = p[128];
        }
        BV ah[4], am[4], al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned char* a = ap + ((q >> 1) * IXP + (q & 1) * 16) * POSB;
          ah[q].u = *reinterpret_cast<const uint4*>(a);
          am[q].u = *reinterpret_cast<const uint4*>(a + 16);
          if (!F16) al[q].u = *reinterpret_cast<const uint4*>(a + 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F16) {
#pragma unroll
          for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][nb][q], am[q], wh[nb]);   // lo x hi
#pragma unroll
            for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][nb][q], ah[q], wm[nb]);   // hi x lo
#pragma unroll
            for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][nb][q], ah[q], wh[nb]);   // hi x hi
          }
          continue;
        }
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], al[q], wh[nb]);    // order 2^-16 terms first
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], am[q], wm[nb]);
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], ah[q], wlo[nb]);
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], am[q], wh[nb]);    // 2^-8
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], ah[q], wm[nb]);
#pragma unroll
          for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][nb][q], ah[q], wh[nb]);    // leading term
        }
      }
      ks0 += nks;
    }
  }

  if constexpr (F16) {         // back to the convolution's scale (exact: powers of two)
#pragma unroll
    for (int b = 0; b < NBR; ++b)
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][nb][q] = acc[b][nb][q] * ep.omul;
  }
  if constexpr (MODE == 2) {
    // visibility-CNN layer: ReLU(conv + folded BatchNorm); with a head (ep.w1 = head weights [16], ep.b1 = head bias [1]) the 1x1
    // convolution 16 -> 1 + sigmoid follows and the result is one value per pixel, ep.out [N][H][W]; else ep.out [N][H][W][16]
    const float bv = bias ? bias[m] : 0.f;
    const bool head = ep.w1 != nullptr;
    if (head) {
      const float hw_n = ep.w1[m], hb = ep.b1[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + g * 4;
        const f32x4 a = acc[0][0][q];
        float v[4] = {fmaxf(a.x + bv, 0.f), fmaxf(a.y + bv, 0.f), fmaxf(a.z + bv, 0.f), fmaxf(a.w + bv, 0.f)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sacc = row16_sum(v[i] * hw_n) + hb;
          v[i] = 1.0f / (1.0f + expf(-sacc));
        }
        if (m == 0 && oy < H) {
          float* __restrict__ o = ep.out + ((size_t)img * H + oy) * W + ox;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (ox + i < W) o[i] = v[i];
        }
      }
      return;
    }
    __syncthreads();                                           // every wave is done with the staged input tile
    float* trL = reinterpret_cast<float*>(lds) + wave * (4 * 16 * C::TP);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a = acc[0][0][q];
      const float v[4] = {fmaxf(a.x + bv, 0.f), fmaxf(a.y + bv, 0.f), fmaxf(a.z + bv, 0.f), fmaxf(a.w + bv, 0.f)};
#pragma unroll
      for (int i = 0; i < 4; ++i) trL[(q * 16 + g * 4 + i) * C::TP + m] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + m;
      if (oy < H && ox < W)
        *reinterpret_cast<float4*>(ep.out + (((size_t)img * H + oy) * W + ox) * 16 + 4 * g) =
            *reinterpret_cast<const float4*>(trL + (q * 16 + m) * C::TP + 4 * g);
    }
    return;
  }
  dyn_epilogue<C>(acc, lds, bias, ep, img, H, W, ox0, oy0, wave, m, g, tid, ty_i * tiles_x + tx_i, tiles_x * tiles_y, ep.w1, ep.b1, ep.w2,
                  ep.epi[2 * img], ep.epi[2 * img + 1]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv00 (module.py:209: DynamicConv 3 -> 8, kernel sizes 3 / 7 / 11) on the matrix cores.  Three input channels would leave 5/8 of a
// K-step's 4 taps x 8 channels padding; here a K-step is 8 x-adjacent tap PAIRS... 4 k-groups x (2 taps x 4 channels, the 4th zero):
// k = 3 / 7 / 11 take 2 + 7 + 17 = 26 K-steps (the 8-channel scheme: 41).  The image planes [S][3][H][W] are staged once per tile as
// [term][row][x][4 ch] bf16 (8 bytes per position and term: a tap pair = 16 contiguous bytes = one LDS read per term), the A operand
// of lane (m, g) is pixel m of its x-run at the pair 4 t + g; weights from ops.split_pack_conv00.  Same operand roles, accumulator
// layout and epilogue as dynconv_cl_kernel.  An image SLOT is convolved once and blended for every output image that shows it: the
// V copies of the reference image of a FeatureNet batch share slot 0, each with its own epipole.
// ---------------------------------------------------------------------------------------------------------------------------
struct C00 {
  static constexpr int NBR = 3, NBLK = 1, NCB = 1, COUT = 8, CO3 = 11, TP = 20;
  static constexpr int R = 5, HL = 8;                          // halo rows / staged halo columns (16-byte aligned image rows)
  static constexpr int IXP = TX + 2 * HL, IY = TY + 2 * R, NPOS = IXP * IY, PLANE = NPOS * 8;
  static constexpr int WREG = cmax(NBR * 4 * 64 * 4, 4 * NCB * 16 * TP * 4);
  static constexpr int REDB = 4 * NBLK * 16 * 2 * 8;
  static constexpr int LDSB = cmax(3 * PLANE, 4 * WREG + REDB);
  static constexpr int KS[3] = {3, 7, 11};
  static constexpr int pairs(int k) { return (k + 1) / 2; }
  static constexpr int nks(int k) { return (k * pairs(k) + 3) / 4; }
};

#ifndef CDS_PROBE_C00
#define CDS_PROBE_C00 0      // probe builds (wrong results, right timing; profiles/r06_experiments.md): 1 = one epilogue per slot, 2 = no K-loop
#endif
template <bool F16>      // split-f16 arithmetic: the images' scale from a device bound (in_bound >= max |x|), weights x w_scale
__global__ __launch_bounds__(256, 2) void conv00_cl_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                           const float* __restrict__ bias, DynEpi ep, int S, int n_shared, int H, int W,
                                                           int tiles_x, int tiles_y, const float* __restrict__ in_bound, float w_inv) {
  using C = C00;
  const float xs = F16 ? sf16_scale(in_bound[0]) : 1.0f;
  constexpr int NBR = C::NBR, IXP = C::IXP, R = C::R, HL = C::HL;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int lin = cds_xcd_remap(blockIdx.x, tiles_x * tiles_y * S);
  const int tx_i = lin % tiles_x;
  lin /= tiles_x;
  const int ty_i = lin % tiles_y, slot = lin / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int ox0 = tx_i * TX, oy0 = ty_i * TY;
  const size_t plane = (size_t)H * W;

  // ---- stage: unit = (row, x-quad): three 16-byte loads (one per colour plane), four positions of 4 channels (the 4th zero),
  // exact 3-way bf16 split, per term four positions = 32 contiguous bytes ----
  if (tid < C::IY * (IXP / 4)) {
    const int row = tid / (IXP / 4), q4 = tid - row * (IXP / 4);
    const int gy = oy0 - R + row, gx = ox0 - HL + 4 * q4;
    float4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gy < (unsigned)H) {
        const float* __restrict__ src = x + ((size_t)slot * 3 + c) * plane + (size_t)gy * W;
        if (gx >= 0 && gx + 3 < W && (W & 3) == 0) {
          v[c] = *reinterpret_cast<const float4*>(src + gx);
        } else {
          if ((unsigned)gx < (unsigned)W) v[c].x = src[gx];
          if ((unsigned)(gx + 1) < (unsigned)W) v[c].y = src[gx + 1];
          if ((unsigned)(gx + 2) < (unsigned)W) v[c].z = src[gx + 2];
          if ((unsigned)(gx + 3) < (unsigned)W) v[c].w = src[gx + 3];
        }
      }
    }
    uint32_t h[4][2], md[4][2], lo[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float c0 = p == 0 ? v[0].x : p == 1 ? v[0].y : p == 2 ? v[0].z : v[0].w;
      const float c1 = p == 0 ? v[1].x : p == 1 ? v[1].y : p == 2 ? v[1].z : v[1].w;
      const float c2 = p == 0 ? v[2].x : p == 1 ? v[2].y : p == 2 ? v[2].z : v[2].w;
      if (F16) {
        split2_f16(c0, c1, xs, h[p][0], md[p][0]);
        split2_f16(c2, 0.f, xs, h[p][1], md[p][1]);
      } else {
        split2(c0, c1, h[p][0], md[p][0], lo[p][0]);
        split2(c2, 0.f, h[p][1], md[p][1], lo[p][1]);
      }
    }
    unsigned char* d = lds + (row * IXP + 4 * q4) * 8;
    uint4* d4;
    d4 = reinterpret_cast<uint4*>(d);
    d4[0] = make_uint4(h[0][0], h[0][1], h[1][0], h[1][1]);
    d4[1] = make_uint4(h[2][0], h[2][1], h[3][0], h[3][1]);
    d4 = reinterpret_cast<uint4*>(d + C::PLANE);
    d4[0] = make_uint4(md[0][0], md[0][1], md[1][0], md[1][1]);
    d4[1] = make_uint4(md[2][0], md[2][1], md[3][0], md[3][1]);
    if (!F16) {
      d4 = reinterpret_cast<uint4*>(d + 2 * C::PLANE);
      d4[0] = make_uint4(lo[0][0], lo[0][1], lo[1][0], lo[1][1]);
      d4[1] = make_uint4(lo[2][0], lo[2][1], lo[3][0], lo[3][1]);
    }
  }
  __syncthreads();

  f32x4 acc[NBR][1][4];
#pragma unroll
  for (int b = 0; b < NBR; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[b][0][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint4* __restrict__ wl = wsp + lane;
  int ks0 = 0;
#pragma unroll
  for (int b = 0; b < NBR; ++b) {
    const int k = C::KS[b], rk = (k - 1) >> 1, np = C::pairs(k), npairs = k * np, nks = C::nks(k);
    const uint4* __restrict__ wb = wl + (size_t)ks0 * 3 * 64;
#pragma unroll 1
    for (int t = 0; t < (CDS_PROBE_C00 == 2 ? 1 : nks); ++t) {
      int pi = 4 * t + g;
      if (pi >= npairs) pi = npairs - 1;                        // padded pair: zero weights, any in-tile data
      const int ky = pi / np, kxp = pi - ky * np;
      // pixel (row 2 wave + (q >> 1), x-run (q & 1)) x m: first tap of the pair at staged column HL + x - rk + 2 kxp
      const unsigned char* ap = lds + (((wave * 2 + ky + R - rk) * IXP) + (HL - rk + 2 * kxp + m)) * 8;
      BV wh, wm, wlo;
      {
        const uint4* p = wb + (size_t)(t * 3) * 64;
        wh.u = p[0];
        wm.u = p[64];
        if (!F16) wlo.u = p[128];
      }
      BV ah[4], am[4], al[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned char* a = ap + ((q >> 1) * IXP + (q & 1) * 16) * 8;
        const uint2 h0 = *reinterpret_cast<const uint2*>(a), h1 = *reinterpret_cast<const uint2*>(a + 8);
        const uint2 m0 = *reinterpret_cast<const uint2*>(a + C::PLANE), m1 = *reinterpret_cast<const uint2*>(a + C::PLANE + 8);
        ah[q].u = make_uint4(h0.x, h0.y, h1.x, h1.y);
        am[q].u = make_uint4(m0.x, m0.y, m1.x, m1.y);
        if (!F16) {
          const uint2 l0 = *reinterpret_cast<const uint2*>(a + 2 * C::PLANE), l1 = *reinterpret_cast<const uint2*>(a + 2 * C::PLANE + 8);
          al[q].u = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (F16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][0][q], am[q], wh);   // lo x hi
#pragma unroll
        for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][0][q], ah[q], wm);   // hi x lo
#pragma unroll
        for (int q = 0; q < 4; ++q) SF16_MFMA(acc[b][0][q], ah[q], wh);   // hi x hi
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], al[q], wh);    // order 2^-16 terms first
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], am[q], wm);
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], ah[q], wlo);
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], am[q], wh);    // 2^-8
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], ah[q], wm);
#pragma unroll
      for (int q = 0; q < 4; ++q) SBF_MFMA(acc[b][0][q], ah[q], wh);    // leading term
    }
    ks0 += nks;
  }
  if constexpr (F16) {
    const float omul = w_inv / xs;                              // exact: powers of two
#pragma unroll
    for (int b = 0; b < NBR; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[b][0][q] = acc[b][0][q] * omul;
  }
  // output images of this slot: the n_shared reference copies for slot 0, one image otherwise
  const int first = slot == 0 ? 0 : slot + n_shared - 1, count = (slot == 0 && CDS_PROBE_C00 != 1) ? n_shared : 1;
  // the attention MLP (28 floats) and the epipoles of this slot's output images: loaded ONCE, before the first epilogue's stores
  float mw1[4 * NBR], mb1[4], mw2[NBR * 4];
#pragma unroll
  for (int q = 0; q < 4 * NBR; ++q) {
    mw1[q] = ep.w1[q];
    mw2[q] = ep.w2[q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) mb1[q] = ep.b1[q];
  const int eimg = first + min(lane, count - 1);               // lane i holds the epipole of output image first + i (count <= 64)
  const float exl = ep.epi[2 * eimg], eyl = ep.epi[2 * eimg + 1];
  for (int i = 0; i < count; ++i) {
    const float ex = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(exl), i));
    const float ey = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eyl), i));
    dyn_epilogue<C>(acc, lds, bias, ep, first + i, H, W, ox0, oy0, wave, m, g, tid, ty_i * tiles_x + tx_i, tiles_x * tiles_y, mw1, mb1, mw2,
                    ex, ey);
  }
}

template <int CIN, int K0, int K1, int K2, int MODE = 1, bool F16 = false>
int launch_dynconv_cl(const float* x, const float* aff, const void* wsp, const float* bias, const DynEpi& ep, int N, int H, int W,
                      hipStream_t st) {
  using C = DCfg<CIN, K0, K1, K2, MODE>;
  const int tx = cds_ceil_div(W, TX), ty = cds_ceil_div(H, TY);
  auto kern = dynconv_cl_kernel<CIN, K0, K1, K2, MODE, F16>;
  if (C::LDSB > 64 * 1024) {
    static std::atomic<unsigned long long> lds_ok{0};   // per instantiation
    if (int e = cds_allow_lds(reinterpret_cast<const void*>(kern), C::LDSB, lds_ok)) return e;
  }
  hipLaunchKernelGGL(kern, dim3(tx * ty * N), dim3(256), (size_t)C::LDSB, st, x, aff, reinterpret_cast<const uint4*>(wsp), bias, ep, N,
                     H, W, tx, ty);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// DynamicConv epilogue over a PLANAR branch tensor [K][slots][Cout + 3][H][W] (conv00: the VALU branch kernels of conv2d.hip),
// channels-last output [N][H][W][8] + InstanceNorm records; the first n_shared images share branch slot 0.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BCL_PXT = 4;
template <int K>
__global__ __launch_bounds__(256) void dynconv_blend_cl_kernel(const float* __restrict__ branch, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               EpiBatch epi, float temperature, float* __restrict__ out,
                                                               float* __restrict__ norm_curv, double* __restrict__ partial, int N,
                                                               int H, int W, int n_shared) {
  constexpr int PXT = BCL_PXT, Cout = 8;
  const int hw = H * W;
  const int n = blockIdx.y;
  const int slot = n < n_shared ? 0 : n - n_shared + 1;
  const int nslots = N - n_shared + 1;
  branch += (size_t)slot * (Cout + 3) * hw;
  out += (size_t)n * Cout * hw;
  const size_t bstride = (size_t)nslots * (Cout + 3) * hw;
  const int base = blockIdx.x * (256 * PXT) + threadIdx.x;
  double ds[Cout], dq[Cout];
#pragma unroll
  for (int c = 0; c < Cout; ++c) ds[c] = dq[c] = 0.0;
#pragma unroll
  for (int j = 0; j < PXT; ++j) {
    const int p = base + 256 * j;
    if (p >= hw) continue;
    float lg[K];
    norm_curv[(size_t)n * hw + p] = blend_weights<K>(branch, bstride, Cout, hw, p, W, epi.x(n), epi.y(n), w1, b1, w2, temperature, lg);
    float o[Cout];
#pragma unroll
    for (int c = 0; c < Cout; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) s = s + branch[k * bstride + (size_t)c * hw + p] * lg[k];
      o[c] = s;
      const double v = (double)s;
      ds[c] += v;
      dq[c] += v * v;
    }
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)p * Cout);
    o4[0] = make_float4(o[0], o[1], o[2], o[3]);
    o4[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
  const int wave = threadIdx.x >> 6;
  const int parts = 4 * gridDim.x;
  double* rec = partial + (((size_t)n * parts + blockIdx.x * 4 + wave) * Cout) * 2;
#pragma unroll
  for (int c = 0; c < Cout; ++c) {
    const double s = wave_sum_f64(ds[c]), s2 = wave_sum_f64(dq[c]);
    if ((threadIdx.x & 63) == 0) {
      rec[2 * c] = s;
      rec[2 * c + 1] = s2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3x3, stride 2, pad 1 convolution CIN -> COUT (downsample1 8 -> 16, downsample2 16 -> 32; module.py:236,240), channels-last in
// and out, normalise-on-load.  Thread = TWO x-adjacent output pixels: per tap the two texels (columns kx and kx + 2 of the shared
// 3 x 5 window) are loaded straight from global memory (consecutive lanes read consecutive texel pairs; the overlaps are L1 hits)
// and every wave-uniform weight (scalar cache, [tap][cin][cout]) feeds two multiply-adds.  At most 64 weights are in flight (SGPRs).
// Summation order per output: taps outer, input channels inner (fp32 fmaf chain).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float affine_leaky(float v, const float* __restrict__ aff, int ci) {
  const float t = v * aff[3 * ci] + aff[3 * ci + 1];
  return t > 0.f ? t : t * aff[3 * ci + 2];
}

// opaque zero (asm volatile): pointer arithmetic with it cannot be hoisted above this point, so the scalar weight loads of one group
// of input channels are issued here and not all at once (hundreds of SGPRs: spills)
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}
// the same, ordered AFTER the computation of `dep` (a data dependency the scheduler must respect): the loads behind it cannot be
// clustered with those of the previous channel group
__device__ __forceinline__ int opaque_zero_after(float dep) {
  int z = 0;
  asm volatile("" : "+s"(z) : "v"(dep));
  return z;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv2d_k3s2_cl_kernel(const float* __restrict__ x, const float* __restrict__ affine,
                                                             const float* __restrict__ wpk, float* __restrict__ out, int H, int W,
                                                             int Ho, int Wo) {
  constexpr int C4 = CIN / 4;
  const int n = blockIdx.y;
  const int Wp = (Wo + 1) >> 1;                               // output pixel pairs per row
  const int pp = blockIdx.x * 256 + threadIdx.x;
  if (pp >= Ho * Wp) return;
  const int oy = pp / Wp, ox = (pp - oy * Wp) * 2;
  const bool has2 = ox + 1 < Wo;
  const float* __restrict__ aff = affine ? affine + (size_t)n * CIN * 3 : nullptr;
  const float* __restrict__ xn = x + (size_t)n * H * W * CIN;
  float acc0[COUT], acc1[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc0[c] = acc1[c] = 0.f;
  // one input row of the 3 x 5 window at a time: its five texels are requested together (5 C4 16-byte loads in flight per thread),
  // then consumed: column kx feeds the first pixel, column kx + 2 the second
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    const int gy = 2 * oy - 1 + ky;
    const bool oky = (unsigned)gy < (unsigned)H;
    float4 t[5][C4];
    bool okc[5];
#pragma unroll
    for (int cx = 0; cx < 5; ++cx) {
      const int gx = 2 * ox - 1 + cx;
      okc[cx] = oky && (unsigned)gx < (unsigned)W && (cx < 3 || has2);
      const float4* __restrict__ sp = reinterpret_cast<const float4*>(xn + ((size_t)(okc[cx] ? gy : 0) * W + (okc[cx] ? gx : 0)) * CIN);
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) t[cx][c4] = sp[c4];
    }
    if (aff) {                                                // normalise-on-load; zero padding follows the normalisation
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        const float* __restrict__ af = aff + opaque_zero() + c4 * 12;
#pragma unroll
        for (int cx = 0; cx < 5; ++cx) {
          t[cx][c4].x = affine_leaky(t[cx][c4].x, af, 0);
          t[cx][c4].y = affine_leaky(t[cx][c4].y, af, 1);
          t[cx][c4].z = affine_leaky(t[cx][c4].z, af, 2);
          t[cx][c4].w = affine_leaky(t[cx][c4].w, af, 3);
        }
      }
    }
#pragma unroll
    for (int cx = 0; cx < 5; ++cx)
      if (!okc[cx]) {
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) t[cx][c4] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int c4 = 0; c4 < C4; ++c4) {
        const float* __restrict__ wt = wpk + opaque_zero() + ((ky * 3 + kx) * CIN + c4 * 4) * COUT;
        const float va[4] = {t[kx][c4].x, t[kx][c4].y, t[kx][c4].z, t[kx][c4].w};
        const float vb[4] = {t[kx + 2][c4].x, t[kx + 2][c4].y, t[kx + 2][c4].z, t[kx + 2][c4].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (COUT >= 32 && j == 2) wt += opaque_zero();     // at most 2 x 32 weights in flight for the wide layer
#pragma unroll
          for (int c = 0; c < COUT; ++c) {
            const float wv = wt[j * COUT + c];
            acc0[c] = fmaf(va[j], wv, acc0[c]);
            acc1[c] = fmaf(vb[j], wv, acc1[c]);
          }
        }
      }
    }
  }
  float4* o4 = reinterpret_cast<float4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * COUT);
#pragma unroll
  for (int c = 0; c < COUT; c += 4) o4[c >> 2] = make_float4(acc0[c], acc0[c + 1], acc0[c + 2], acc0[c + 3]);
  if (has2) {
#pragma unroll
    for (int c = 0; c < COUT; c += 4) o4[(COUT + c) >> 2] = make_float4(acc1[c], acc1[c + 1], acc1[c + 2], acc1[c + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// FPN lateral (module.py:253-254,260-261): 1x1 convolution over cat(nearest2x(coarse), skip), channels-last, neither tensor
// materialised; each source has its own optional normalise-on-load table.  Thread = one 2 x 2 block of output pixels = ONE coarse
// texel: the coarse half of the fmaf chain (the first Ca channels of the concatenation) is the same for the four pixels and is
// computed once; the skip half continues it per pixel - the chain order of conv2d_kernel<1,...> on the materialised concatenation,
// bit-identical results.  ALL texels of a block (one coarse, four skip: 2 rows of 2 Cb contiguous floats) are requested before the
// first multiply (Ca / 4 + Cb 16-byte loads in flight per thread: the kernel is memory-bound); every wave-uniform weight
// ([Ca + Cb][COUT], scalar cache) of the skip half feeds four multiply-adds.  InstanceNorm records: fp64 sums per thread over its
// FPN_NB blocks, one wave reduction per channel at the end.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int FPN_NB = 2;
template <int CA, int CB, int COUT, bool HAS_A, bool HAS_B>
__global__ __launch_bounds__(256) void fpn_lateral_cl_kernel(const float* __restrict__ xa, const float* __restrict__ affa,
                                                             const float* __restrict__ xb, const float* __restrict__ affb,
                                                             const float* __restrict__ wpk, float* __restrict__ out,
                                                             double* __restrict__ partial, int H, int W) {
  const int n = blockIdx.y;
  const int Hc = H >> 1, Wc = W >> 1, nblk = Hc * Wc;
  const float* __restrict__ fa = affa + (HAS_A ? (size_t)n * CA * 3 : 0);     // compile-time switches: no per-element pointer tests
  const float* __restrict__ fb = affb + (HAS_B ? (size_t)n * CB * 3 : 0);
  xa += (size_t)n * nblk * CA;
  xb += (size_t)n * H * W * CB;
  out += (size_t)n * H * W * COUT;
  double ds[COUT], dq[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) ds[c] = dq[c] = 0.0;
#pragma unroll 1
  for (int it = 0; it < FPN_NB; ++it) {
    const int bi = (blockIdx.x * FPN_NB + it) * 256 + threadIdx.x;
    if (bi >= nblk) break;
    const int by = bi / Wc, bx = bi - by * Wc;
    const size_t p0 = (size_t)(2 * by) * W + 2 * bx;           // pixels p0, p0 + 1 and p0 + W, p0 + W + 1
    float4 ta[CA / 4], tb[2][2 * CB / 4];
    {
      const float4* __restrict__ sa = reinterpret_cast<const float4*>(xa + (size_t)bi * CA);
#pragma unroll
      for (int c4 = 0; c4 < CA / 4; ++c4) ta[c4] = sa[c4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float4* __restrict__ sb = reinterpret_cast<const float4*>(xb + (p0 + (size_t)r * W) * CB);
#pragma unroll
        for (int c4 = 0; c4 < 2 * CB / 4; ++c4) tb[r][c4] = sb[c4];
      }
    }
    // normalise-on-load of every texel first (12 wave-uniform constants per channel quad at a time), then the multiply-adds see
    // plain values and only the weights travel through scalar registers
    if (HAS_A) {
#pragma unroll
      for (int c4 = 0; c4 < CA / 4; ++c4) {
        const float* __restrict__ af = fa + opaque_zero_after(ta[c4 ? c4 - 1 : 0].x) + c4 * 12;
        ta[c4].x = affine_leaky(ta[c4].x, af, 0);
        ta[c4].y = affine_leaky(ta[c4].y, af, 1);
        ta[c4].z = affine_leaky(ta[c4].z, af, 2);
        ta[c4].w = affine_leaky(ta[c4].w, af, 3);
      }
    }
    if (HAS_B) {
#pragma unroll
      for (int c4 = 0; c4 < CB / 4; ++c4) {
        const float* __restrict__ af = fb + opaque_zero_after(tb[0][c4 ? c4 - 1 : 0].x) + c4 * 12;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            float4& t4 = tb[r][sx * (CB / 4) + c4];
            t4.x = affine_leaky(t4.x, af, 0);
            t4.y = affine_leaky(t4.y, af, 1);
            t4.z = affine_leaky(t4.z, af, 2);
            t4.w = affine_leaky(t4.w, af, 3);
          }
      }
    }
    float base[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) base[c] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < CA / 4; ++c4) {
      const float* __restrict__ wc = wpk + opaque_zero_after(base[0]) + c4 * 4 * COUT;
      const float v[4] = {ta[c4].x, ta[c4].y, ta[c4].z, ta[c4].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) base[c] = fmaf(v[j], wc[j * COUT + c], base[c]);
      }
    }
    float acc[4][COUT];                                        // pixel 2 r + s of the block
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[q][c] = base[c];
#pragma unroll
    for (int c4 = 0; c4 < CB / 4; ++c4) {
      const float* __restrict__ wc = wpk + opaque_zero_after(acc[0][0]) + (CA + c4 * 4) * COUT;
      float v[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t4 = tb[q >> 1][(q & 1) * (CB / 4) + c4];
        v[q][0] = t4.x; v[q][1] = t4.y; v[q][2] = t4.z; v[q][3] = t4.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const float wv = wc[j * COUT + c];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][c] = fmaf(v[q][j], wv, acc[q][c]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float4* o4 = reinterpret_cast<float4*>(out + (p0 + (size_t)r * W) * COUT);
#pragma unroll
      for (int c = 0; c < COUT; c += 4) {
        o4[c >> 2] = make_float4(acc[2 * r][c], acc[2 * r][c + 1], acc[2 * r][c + 2], acc[2 * r][c + 3]);
        o4[(COUT + c) >> 2] = make_float4(acc[2 * r + 1][c], acc[2 * r + 1][c + 1], acc[2 * r + 1][c + 2], acc[2 * r + 1][c + 3]);
      }
    }
    if (partial) {
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        const double d0 = (double)acc[0][c], d1 = (double)acc[1][c], d2 = (double)acc[2][c], d3 = (double)acc[3][c];
        ds[c] += (d0 + d1) + (d2 + d3);
        dq[c] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
  }
  if (partial) {
    const int wave = threadIdx.x >> 6;
    double* rec = partial + (((size_t)n * (4 * gridDim.x) + blockIdx.x * 4 + wave) * COUT) * 2;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      const double s = wave_sum_f64(ds[c]), s2 = wave_sum_f64(dq[c]);
      if ((threadIdx.x & 63) == 0) {
        rec[2 * c] = s;
        rec[2 * c + 1] = s2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// InstanceNorm records of a channels-last tensor [N][hw][C]: thread = float4 chunk of the image's contiguous bytes, so its channel
// quad is fixed (grid stride % (C / 4) == 0): 8 fp64 sums in registers; lanes with the same quad are combined by cross-lane adds,
// waves through LDS; one record per (workgroup, channel).  partial [N][gridDim.x][C][2].
// ---------------------------------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256) void instnorm_stats_cl_kernel(const float* __restrict__ x, double* __restrict__ partial, int hw) {
  constexpr int C4 = CT / 4;
  __shared__ double red[4][CT][2];
  const int n = blockIdx.y;
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + (size_t)n * hw * CT);
  const size_t total = (size_t)hw * C4;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float4 v = x4[i];
    const double a = v.x, b = v.y, c = v.z, d = v.w;
    s[0] += a; q[0] += a * a;
    s[1] += b; q[1] += b * b;
    s[2] += c; q[2] += c * c;
    s[3] += d; q[3] += d * d;
  }
  // lanes l and l' hold the same channel quad iff l % C4 == l' % C4: fold the 64 lanes down to C4 (xor offsets 32 .. C4)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 32; o >= C4; o >>= 1) {
      s[j] += __shfl_xor(s[j], o);
      q[j] += __shfl_xor(q[j], o);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < C4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][lane * 4 + j][0] = s[j];
      red[wave][lane * 4 + j][1] = q[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < CT) {
    const int c = threadIdx.x;
    double* rec = partial + (((size_t)n * gridDim.x + blockIdx.x) * CT + c) * 2;
    rec[0] = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
    rec[1] = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
  }
}

// InstanceNorm + activation of a channels-last tensor for given statistics: out_cl [N - cl_from][hw][CT] for the images
// n >= cl_from (may be NULL) and out_chw [n_chw][CT][hw] for the first n_chw images (the reference-view feature maps K1 / K3 read per
// pixel and channel plane).
template <int CT>
__global__ __launch_bounds__(256) void instnorm_apply_cl_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                                float* __restrict__ out_cl, float* __restrict__ out_chw, int hw,
                                                                int act, int n_chw, int cl_from) {
  __shared__ float ab[2][CT];
  const int n = blockIdx.y;
  stats += (size_t)n * 2 * CT;
  if ((int)threadIdx.x < CT) {
    const int c = threadIdx.x;
    const double mean = stats[2 * c] / hw;
    double var = stats[2 * c + 1] / hw - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + 1e-5));
    ab[0][c] = invstd;
    ab[1][c] = -(float)mean * invstd;
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool want_cl = out_cl && n >= cl_from;
  if (p >= hw || (!want_cl && n >= n_chw)) return;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(x + ((size_t)n * hw + p) * CT);
  float v[CT];
#pragma unroll
  for (int c4 = 0; c4 < CT / 4; ++c4) {
    const float4 t = src[c4];
    v[4 * c4] = t.x; v[4 * c4 + 1] = t.y; v[4 * c4 + 2] = t.z; v[4 * c4 + 3] = t.w;
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) v[c] = cds_apply_act(v[c] * ab[0][c] + ab[1][c], act);
  if (want_cl) {
    float4* o4 = reinterpret_cast<float4*>(out_cl + ((size_t)(n - cl_from) * hw + p) * CT);
#pragma unroll
    for (int c = 0; c < CT; c += 4) o4[c >> 2] = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
  }
  if (n < n_chw) {
    float* o = out_chw + (size_t)n * CT * hw + p;
#pragma unroll
    for (int c = 0; c < CT; ++c) o[(size_t)c * hw] = v[c];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Visibility CNN, layer 1 (models/model.py:14,51): cat(entropy, ref_nc) -> 3x3 convolution 2 -> 16 (BatchNorm folded) + ReLU, straight
// from the two maps [V][h][w] to the channels-last activation [V][h][w][16] the matrix-core layers 2 / 3 stage from.  Memory-bound
// (8 bytes in, 64 bytes out per pixel): one thread per pixel, the 2 x 9 taps through L1, wave-uniform weights [2][9][16].
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vis_layer1_cl_kernel(const float* __restrict__ ent, const float* __restrict__ nc,
                                                            const float* __restrict__ wpk, const float* __restrict__ bias,
                                                            float* __restrict__ out, int H, int W) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const float* __restrict__ src = (ci ? nc : ent) + (size_t)n * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int gy = y - 1 + ky, gx = x - 1 + kx;
        const float v = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? src[(size_t)gy * W + gx] : 0.f;
        const float* __restrict__ wc = wpk + (ci * 9 + ky * 3 + kx) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wc[c], acc[c]);
      }
  }
  float4* o4 = reinterpret_cast<float4*>(out + ((size_t)n * H * W + p) * 16);
#pragma unroll
  for (int c = 0; c < 16; c += 4)
    o4[c >> 2] = make_float4(fmaxf(acc[c] + bias[c], 0.f), fmaxf(acc[c + 1] + bias[c + 1], 0.f), fmaxf(acc[c + 2] + bias[c + 2], 0.f),
                             fmaxf(acc[c + 3] + bias[c + 3], 0.f));
}

// ---------------------------------------------------------------------------------------------------------------------------
// The stride-2 units (downsample1 8 -> 16, downsample2 16 -> 32; module.py:213,217) on the MATRIX CORES in split-f16 arithmetic (round
// 6; the VALU kernel above was 0.63 ms per 1600x1184 forward at 25 TFLOP/s).  A workgroup owns 32 x 4 OUTPUT pixels: wave = output row,
// two 16-pixel x-runs per wave; it stages the 65 x 9 input texels (+ 1 pad column) of all channels once - normalise-on-load (the
// producer's InstanceNorm + LeakyReLU), two fp16 terms of value x xs (xs from the bound sqrt(H W) of an InstanceNorm-ed map: host
// number), 48-byte positions as everywhere (term slot 2 unused): the A operand of lane (m, g) is the input texel 2 (16 q + m) + kx of row
// 2 wave + ky for tap 4 t + g - a 96-byte lane stride, conflict-free for the eight lanes an LDS cycle serves.  B = weights from
// ops.split_pack_dynconv([w], f16=True) (tap-major K-steps: 3 per 8-channel round, 3 of 12 tap slots zero).  Epilogue: exact rescaling,
// transposition through LDS, 16-byte channels-last stores (raw convolution result: its InstanceNorm statistics are the next launch's).
// ---------------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct S2M {
  static constexpr int TXO = 32, TYO = 4;
  static constexpr int IX = 2 * TXO + 1, IY = 2 * TYO + 1, IXP = IX + 1, NPOS = IXP * IY;
  static constexpr int ROUNDS = CIN / 8, NBLK = COUT / 16, C4 = CIN / 4;
  static constexpr int PLANE = NPOS * POSB;
  static constexpr int NCH = NPOS * C4, NIT = (NCH + 255) / 256, PSTEP = 256 / C4;
  static constexpr int TP = 20;
  static constexpr int TRB = 4 * 2 * NBLK * 16 * TP * 4;      // [wave][q][block][16 px][TP] floats
  static constexpr int LDSB = cmax(ROUNDS * PLANE, TRB);
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv2d_k3s2_mfma_cl_kernel(const float* __restrict__ x, const float* __restrict__ affine,
                                                                     const uint4* __restrict__ wsp, float* __restrict__ out, int H, int W,
                                                                     int Ho, int Wo, int tiles_x, int tiles_y, int N, float xs, float omul) {
  using C = S2M<CIN, COUT>;
  constexpr int IXP = C::IXP, NBLK = C::NBLK;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int lin = cds_xcd_remap(blockIdx.x, tiles_x * tiles_y * N);
  const int tx_i = lin % tiles_x;
  lin /= tiles_x;
  const int ty_i = lin % tiles_y, img = lin / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int ox0 = tx_i * C::TXO, oy0 = ty_i * C::TYO;
  const int ix0 = 2 * ox0 - 1, iy0 = 2 * oy0 - 1;             // input texel of staged position (0, 0)
  {
    const int c4 = tid % C::C4;
    float al[4], be[4], sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      al[j] = 1.f; be[j] = 0.f; sl[j] = 1.f;
    }
    if (affine) {
      const float* __restrict__ af = affine + ((size_t)img * CIN + c4 * 4) * 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        al[j] = af[3 * j]; be[j] = af[3 * j + 1]; sl[j] = af[3 * j + 2];
      }
    }
    float4 v[C::NIT];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
      const int pos = tid / C::C4 + i * C::PSTEP;
      const int row = pos / IXP, col = pos - row * IXP;
      const int gy = iy0 + row, gx = ix0 + col;
      const bool ok = pos < C::NPOS && col < C::IX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        v[i] = *reinterpret_cast<const float4*>(x + (((size_t)img * H + gy) * W + gx) * CIN + c4 * 4);
        okmask |= 1u << i;
      }
    }
    unsigned char* dst0 = lds + (c4 >> 1) * C::PLANE + (c4 & 1) * 8;
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
      const int pos = tid / C::C4 + i * C::PSTEP;
      if (pos >= C::NPOS) break;
      float4 t = v[i];
      if (okmask & (1u << i)) {            // zero padding follows the normalisation
        float a;
        a = fmaf(t.x, al[0], be[0]); t.x = a > 0.f ? a : a * sl[0];
        a = fmaf(t.y, al[1], be[1]); t.y = a > 0.f ? a : a * sl[1];
        a = fmaf(t.z, al[2], be[2]); t.z = a > 0.f ? a : a * sl[2];
        a = fmaf(t.w, al[3], be[3]); t.w = a > 0.f ? a : a * sl[3];
      }
      uint32_t h0, l0, h1, l1;
      split2_f16(t.x, t.y, xs, h0, l0);
      split2_f16(t.z, t.w, xs, h1, l1);
      unsigned char* d = dst0 + pos * POSB;
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + 16) = make_uint2(l0, l1);
    }
  }
  __syncthreads();

  f32x4 acc[NBLK][2];
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[nb][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4* __restrict__ wl = wsp + lane;
#pragma unroll
  for (int rd = 0; rd < C::ROUNDS; ++rd) {
    const uint4* __restrict__ wr = wl + (size_t)rd * 3 * NBLK * 3 * 64;
    const unsigned char* __restrict__ lp = lds + rd * C::PLANE;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      int tap = 4 * t + g;
      if (tap > 8) tap = 8;                               // padded tap slots carry zero weights: any staged position
      const int ky = tap / 3, kx = tap - ky * 3;
      const unsigned char* ap = lp + ((2 * wave + ky) * IXP + 2 * m + kx) * POSB;
      BV wh[NBLK], wlo[NBLK], ah[2], alo[2];
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) {
        const uint4* p = wr + (size_t)((t * NBLK + nb) * 3) * 64;
        wh[nb].u = p[0];
        wlo[nb].u = p[64];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned char* a = ap + q * 32 * POSB;
        ah[q].u = *reinterpret_cast<const uint4*>(a);
        alo[q].u = *reinterpret_cast<const uint4*>(a + 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) SF16_MFMA(acc[nb][q], alo[q], wh[nb]);    // lo x hi
#pragma unroll
        for (int q = 0; q < 2; ++q) SF16_MFMA(acc[nb][q], ah[q], wlo[nb]);    // hi x lo
#pragma unroll
        for (int q = 0; q < 2; ++q) SF16_MFMA(acc[nb][q], ah[q], wh[nb]);     // hi x hi
      }
    }
  }
  __syncthreads();                                             // every wave is done with the staged tile: it becomes the transposition area
  // lane (m, g) holds output channel 16 nb + m of pixels x = 16 q + 4 g + i: through LDS to 4 consecutive channels of one pixel per lane
  float* trL = reinterpret_cast<float*>(lds) + wave * (2 * NBLK * 16 * C::TP);
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 a = acc[nb][q] * omul;
      const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) trL[((q * NBLK + nb) * 16 + g * 4 + i) * C::TP + m] = v[i];
    }
  wave_lds_fence();
  const int oy = oy0 + wave;
  if (oy < Ho) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ox = ox0 + q * 16 + m;
      if (ox >= Wo) continue;
      float* __restrict__ op = out + (((size_t)img * Ho + oy) * Wo + ox) * COUT;
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
        *reinterpret_cast<float4*>(op + nb * 16 + 4 * g) = *reinterpret_cast<const float4*>(trL + ((q * NBLK + nb) * 16 + m) * C::TP + 4 * g);
    }
  }
}

}  // namespace

// Visibility CNN on channels-last activations (models/model.py:14,51).
// cds_vis_layer1_cl_f32: entropy, ref_nc [V][h][w] -> ReLU(conv3x3(cat) + bias) as [V][h][w][16]; weight packed [2][9][16] (cout fastest),
//   bias [16] (BatchNorm folded by the caller).
// cds_conv2d_k3_relu_cl_f32: 3x3, 16 -> 16, bias + ReLU in split-bf16 arithmetic on the matrix cores (layers 2 / 3); x [N][H][W][16],
//   weight_split = ops.split_pack_dynconv([w]) with w [16][16][3][3], bias [16]; head_w [16] / head_b [1] or both NULL:
//   out [N][H][W][16], or with the head sigmoid(head_b + sum_c head_w[c] relu(..)[c]) as [N][H][W].
extern "C" int cds_vis_layer1_cl_f32(const float* entropy, const float* ref_nc, const float* weight, const float* bias, float* out, int V,
                                     int H, int W, void* stream) {
  if (!entropy || !ref_nc || !weight || !bias || !out || V < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(vis_layer1_cl_kernel, dim3(cds_ceil_div(H * W, 256), V), dim3(256), 0, (hipStream_t)stream, entropy, ref_nc, weight,
                     bias, out, H, W);
  return cds_launch_status();
}

extern "C" int cds_conv2d_k3_relu_cl_f32(const float* x, const void* weight_split, const float* bias, const float* head_w,
                                         const float* head_b, float* out, int N, int Cin, int H, int W, void* stream) {
  if (!x || !weight_split || !out || N < 1 || Cin != 16 || H < 1 || W < 1 || (head_w != nullptr) != (head_b != nullptr)) return CDS_EINVAL;
  DynEpi ep{};
  ep.w1 = head_w;
  ep.b1 = head_b;
  ep.out = out;
  return launch_dynconv_cl<16, 3, 0, 0, 2>(x, nullptr, weight_split, bias, ep, N, H, W, (hipStream_t)stream);
}

// Records per image that cds_dynconv_cl_f32 leaves for cds_instnorm_reduce_f32: one per 32 x 8 tile.
extern "C" int cds_dynconv_cl_parts(int H, int W) { return cds_ceil_div(W, TX) * cds_ceil_div(H, TY); }

// One DynamicConv (dynamic_conv.py:97-122) on channels-last activations in ONE kernel.  x [N][H][W][C] (+ in_affine [N][C][3] or
// NULL), weight_split from ops.split_pack_dynconv (the packing of cds_dynconv_branches_sbf_f32), bias [nb][C + 3] or NULL,
// w1 [4][nb], b1 [4], w2 [nb][4] the attention MLP with its BatchNorm folded in, epipoles [N][2] (pixels at this resolution)
// -> out [N][H][W][C] (before its InstanceNorm), norm_curv [N][H][W], partial [N][parts][C][2] doubles with
// parts = cds_dynconv_cl_parts(H, W) (reduce with cds_instnorm_reduce_f32).  (C, ksizes) in {(8, 3-5-7), (8, 1-3), (16, 3-5),
// (16, 1-3), (32, 1-3)}: the DynamicConv layers of FeatureNet with Cin == Cout; N <= CDS_MAX_IMAGES; CDS_EINVAL otherwise.
extern "C" int cds_dynconv_cl_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                                  const float* w1, const float* b1, const float* w2, const float* epipoles, float temperature,
                                  float* out, float* norm_curv, double* partial, int N, int C, int H, int W, const int* ksizes, int nb,
                                  void* stream) {
  if (!x || !weight_split || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || !ksizes || N < 1 ||
      N > CDS_MAX_IMAGES || H < 1 || W < 1 || nb < 2 || nb > 3)
    return CDS_EINVAL;
  DynEpi ep{};
  ep.w1 = w1; ep.b1 = b1; ep.w2 = w2; ep.out = out; ep.norm_curv = norm_curv; ep.partial = partial; ep.temperature = temperature;
  ep.epi = epipoles;
  hipStream_t st = (hipStream_t)stream;
  const int k0 = ksizes[0], k1 = ksizes[1], k2 = nb == 3 ? ksizes[2] : 0;
  if (C == 8 && k0 == 3 && k1 == 5 && k2 == 7) return launch_dynconv_cl<8, 3, 5, 7>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 8 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<8, 1, 3, 0>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 16 && k0 == 3 && k1 == 5 && k2 == 0) return launch_dynconv_cl<16, 3, 5, 0>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 16 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<16, 1, 3, 0>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 32 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<32, 1, 3, 0>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  return CDS_EINVAL;
}

// The same DynamicConv in SPLIT-F16 arithmetic (sbf_common.hpp; half the matrix-pipe work, fp32-class error).  weight_split: the same
// layout with fp16 terms (hi, lo, unused) of weight x w_scale (ops.split_pack_dynconv(..., f16=True)), w_inv_scale = 1 / w_scale;
// x_bound >= max |input after its affine + LeakyReLU| (a host number: sqrt(H W) bounds any InstanceNorm-ed map; 1 a tanh output).
extern "C" int cds_dynconv_cl_sf16_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                                       const float* w1, const float* b1, const float* w2, const float* epipoles, float temperature,
                                       float* out, float* norm_curv, double* partial, int N, int C, int H, int W, const int* ksizes,
                                       int nb, float x_bound, float w_inv_scale, void* stream) {
  if (!x || !weight_split || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || !ksizes || N < 1 ||
      N > CDS_MAX_IMAGES || H < 1 || W < 1 || nb < 2 || nb > 3 || !(x_bound > 0.f) || !(w_inv_scale > 0.f))
    return CDS_EINVAL;
  DynEpi ep{};
  ep.w1 = w1; ep.b1 = b1; ep.w2 = w2; ep.out = out; ep.norm_curv = norm_curv; ep.partial = partial; ep.temperature = temperature;
  ep.epi = epipoles;
  int e = 0;
  (void)frexpf(x_bound, &e);                           // x_bound = m 2^e, m in [0.5, 1): x_bound xs <= 2^15
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  ep.xs = ldexpf(1.0f, 15 - e);
  ep.omul = w_inv_scale / ep.xs;
  hipStream_t st = (hipStream_t)stream;
  const int k0 = ksizes[0], k1 = ksizes[1], k2 = nb == 3 ? ksizes[2] : 0;
  if (C == 8 && k0 == 3 && k1 == 5 && k2 == 7) return launch_dynconv_cl<8, 3, 5, 7, 1, true>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 8 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<8, 1, 3, 0, 1, true>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 16 && k0 == 3 && k1 == 5 && k2 == 0) return launch_dynconv_cl<16, 3, 5, 0, 1, true>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 16 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<16, 1, 3, 0, 1, true>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  if (C == 32 && k0 == 1 && k1 == 3 && k2 == 0) return launch_dynconv_cl<32, 1, 3, 0, 1, true>(x, in_affine, weight_split, bias, ep, N, H, W, st);
  return CDS_EINVAL;
}

// conv00 of FeatureNet (DynamicConv 3 -> 8, kernel sizes 3 / 7 / 11) in ONE kernel on the matrix cores, channels-last result.
// x [S][3][H][W] planar images, S = N - n_shared + 1 slots: slot 0 is shown by the first n_shared output images (the reference copies of a
// FeatureNet batch, each with its own epipole), slot s > 0 by image n_shared - 1 + s.  weight_split = ops.split_pack_conv00, bias
// [3][11] or NULL; w1 [4][3], b1 [4], w2 [3][4] the attention MLP; epipoles [N][2] -> out [N][H][W][8], norm_curv [N][H][W],
// partial [N][cds_dynconv_cl_parts(H, W)][8][2] doubles.
static int conv00_entry(const float* x, const void* weight_split, const float* bias, const float* w1, const float* b1, const float* w2,
                        const float* epipoles, float temperature, float* out, float* norm_curv, double* partial, int N, int n_shared,
                        int H, int W, const float* in_bound, float w_inv_scale, void* stream) {
  if (!x || !weight_split || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || N < 1 || N > CDS_MAX_IMAGES ||
      n_shared < 1 || n_shared > N || H < 1 || W < 1)
    return CDS_EINVAL;
  DynEpi ep{};
  ep.w1 = w1; ep.b1 = b1; ep.w2 = w2; ep.out = out; ep.norm_curv = norm_curv; ep.partial = partial; ep.temperature = temperature;
  ep.epi = epipoles;
  const int S = N - n_shared + 1;
  const int tx = cds_ceil_div(W, TX), ty = cds_ceil_div(H, TY);
  if (in_bound)
    hipLaunchKernelGGL(conv00_cl_kernel<true>, dim3(tx * ty * S), dim3(256), (size_t)C00::LDSB, (hipStream_t)stream, x,
                       reinterpret_cast<const uint4*>(weight_split), bias, ep, S, n_shared, H, W, tx, ty, in_bound, w_inv_scale);
  else
    hipLaunchKernelGGL(conv00_cl_kernel<false>, dim3(tx * ty * S), dim3(256), (size_t)C00::LDSB, (hipStream_t)stream, x,
                       reinterpret_cast<const uint4*>(weight_split), bias, ep, S, n_shared, H, W, tx, ty, nullptr, 1.0f);
  return cds_launch_status();
}

extern "C" int cds_conv00_cl_f32(const float* x, const void* weight_split, const float* bias, const float* w1, const float* b1,
                                 const float* w2, const float* epipoles, float temperature, float* out, float* norm_curv,
                                 double* partial, int N, int n_shared, int H, int W, void* stream) {
  return conv00_entry(x, weight_split, bias, w1, b1, w2, epipoles, temperature, out, norm_curv, partial, N, n_shared, H, W, nullptr, 1.0f,
                      stream);
}

// conv00 in SPLIT-F16 arithmetic: weight_split from ops.split_pack_conv00(..., f16=True), w_inv_scale = 1 / its weight scale; in_bound: a
// DEVICE scalar >= max |x| (the images: e.g. their amax; it fixes their scale).
extern "C" int cds_conv00_cl_sf16_f32(const float* x, const void* weight_split, const float* bias, const float* w1, const float* b1,
                                      const float* w2, const float* epipoles, float temperature, float* out, float* norm_curv,
                                      double* partial, int N, int n_shared, int H, int W, const float* in_bound, float w_inv_scale,
                                      void* stream) {
  if (!in_bound || !(w_inv_scale > 0.f)) return CDS_EINVAL;
  return conv00_entry(x, weight_split, bias, w1, b1, w2, epipoles, temperature, out, norm_curv, partial, N, n_shared, H, W, in_bound,
                      w_inv_scale, stream);
}

extern "C" int cds_blend_cl_parts(int H, int W) { return 4 * cds_ceil_div(H * W, 256 * BCL_PXT); }

// The DynamicConv epilogue of cds_dynconv_blend_stats_f32 with a channels-last result: branches [K][N - n_shared + 1][8 + 3][H][W]
// planar (the first n_shared images share slot 0) -> out [N][H][W][8], norm_curv [N][H][W], partial [N][cds_blend_cl_parts][8][2].
extern "C" int cds_dynconv_blend_cl_f32(const float* branches, const float* w1, const float* b1, const float* w2,
                                        const float* epipoles, float temperature, float* out, float* norm_curv, double* partial,
                                        int N, int K, int Cout, int H, int W, int n_shared, void* stream) {
  if (!branches || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || N < 1 || N > CDS_MAX_IMAGES || K != 3 ||
      Cout != 8 || H < 1 || W < 1 || n_shared < 1 || n_shared > N)
    return CDS_EINVAL;
  const EpiBatch epi{epipoles};
  const dim3 grid(cds_ceil_div(H * W, 256 * BCL_PXT), N);
  hipLaunchKernelGGL(dynconv_blend_cl_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, branches, w1, b1, w2, epi, temperature, out,
                     norm_curv, partial, N, H, W, n_shared);
  return cds_launch_status();
}

// 3x3 stride-2 pad-1 convolution on channels-last activations (downsample1 / downsample2): x [N][H][W][Cin] (+ in_affine) ->
// out [N][Ho][Wo][Cout], Ho = (H - 1) / 2 + 1; weight [9][Cin][Cout] (tap = ky * 3 + kx; cout fastest), no bias.
// (Cin, Cout) in {(8, 16), (16, 32)}.
extern "C" int cds_conv2d_k3s2_cl_f32(const float* x, const float* in_affine, const float* weight, float* out, int N, int Cin,
                                      int Cout, int H, int W, void* stream) {
  if (!x || !weight || !out || N < 1 || H < 1 || W < 1) return CDS_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const dim3 grid(cds_ceil_div(Ho * ((Wo + 1) / 2), 256), N);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 8 && Cout == 16)
    hipLaunchKernelGGL((conv2d_k3s2_cl_kernel<8, 16>), grid, dim3(256), 0, st, x, in_affine, weight, out, H, W, Ho, Wo);
  else if (Cin == 16 && Cout == 32)
    hipLaunchKernelGGL((conv2d_k3s2_cl_kernel<16, 32>), grid, dim3(256), 0, st, x, in_affine, weight, out, H, W, Ho, Wo);
  else
    return CDS_EINVAL;
  return cds_launch_status();
}

// The same layer on the matrix cores in split-f16 arithmetic: weight_split from ops.split_pack_dynconv([w [Cout,Cin,3,3]], f16=True),
// w_inv_scale = 1 / its weight scale, x_bound a HOST number >= max |input after its affine + LeakyReLU| (sqrt(H W) for an InstanceNorm-ed map).
extern "C" int cds_conv2d_k3s2_cl_sf16_f32(const float* x, const float* in_affine, const void* weight_split, float* out, int N, int Cin,
                                           int Cout, int H, int W, float x_bound, float w_inv_scale, void* stream) {
  if (!x || !weight_split || !out || N < 1 || H < 1 || W < 1 || !(x_bound > 0.f) || !(w_inv_scale > 0.f)) return CDS_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  int e = 0;
  (void)frexpf(x_bound, &e);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  const float xs = ldexpf(1.0f, 15 - e), omul = w_inv_scale / xs;
  hipStream_t st = (hipStream_t)stream;
  const uint4* wsp = reinterpret_cast<const uint4*>(weight_split);
  if (Cin == 8 && Cout == 16) {
    using C = S2M<8, 16>;
    const int tx = cds_ceil_div(Wo, C::TXO), ty = cds_ceil_div(Ho, C::TYO);
    hipLaunchKernelGGL((conv2d_k3s2_mfma_cl_kernel<8, 16>), dim3(tx * ty * N), dim3(256), (size_t)C::LDSB, st, x, in_affine, wsp, out, H, W,
                       Ho, Wo, tx, ty, N, xs, omul);
  } else if (Cin == 16 && Cout == 32) {
    using C = S2M<16, 32>;
    const int tx = cds_ceil_div(Wo, C::TXO), ty = cds_ceil_div(Ho, C::TYO);
    hipLaunchKernelGGL((conv2d_k3s2_mfma_cl_kernel<16, 32>), dim3(tx * ty * N), dim3(256), (size_t)C::LDSB, st, x, in_affine, wsp, out, H, W,
                       Ho, Wo, tx, ty, N, xs, omul);
  } else {
    return CDS_EINVAL;
  }
  return cds_launch_status();
}

extern "C" int cds_fpn_cl_parts(int H, int W) { return 4 * cds_ceil_div((H / 2) * (W / 2), 256 * FPN_NB); }

// FPN lateral on channels-last activations: out [N][H][W][Cout] = 1x1 conv of cat(nearest2x(coarse [N][H/2][W/2][Ca]), skip
// [N][H][W][Cb]); weight [Ca + Cb][Cout] (coarse channels first); affine tables [N][Ca][3] / [N][Cb][3] or NULL; partial NULL or
// [N][cds_fpn_cl_parts(H, W)][Cout][2] doubles (InstanceNorm records of `out`).  (Ca, Cb, Cout) in {(32, 16, 16), (16, 8, 8)}.
extern "C" int cds_conv2d_fpn_cl_f32(const float* coarse, const float* coarse_affine, const float* skip, const float* skip_affine,
                                     const float* weight, float* out, double* partial, int N, int Ca, int Cb, int Cout, int H, int W,
                                     void* stream) {
  if (!coarse || !skip || !weight || !out || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return CDS_EINVAL;
  const dim3 grid(cds_ceil_div((H / 2) * (W / 2), 256 * FPN_NB), N);
  hipStream_t st = (hipStream_t)stream;
#define FPN_LAUNCH(CA_, CB_, CO_, HA_, HB_)                                                                                       \
  hipLaunchKernelGGL((fpn_lateral_cl_kernel<CA_, CB_, CO_, HA_, HB_>), grid, dim3(256), 0, st, coarse, coarse_affine, skip, skip_affine, \
                     weight, out, partial, H, W)
#define FPN_PICK(CA_, CB_, CO_)                                  \
  do {                                                           \
    if (coarse_affine && skip_affine) FPN_LAUNCH(CA_, CB_, CO_, true, true);     \
    else if (coarse_affine) FPN_LAUNCH(CA_, CB_, CO_, true, false);              \
    else if (skip_affine) FPN_LAUNCH(CA_, CB_, CO_, false, true);                \
    else FPN_LAUNCH(CA_, CB_, CO_, false, false);                                \
  } while (0)
  if (Ca == 32 && Cb == 16 && Cout == 16) FPN_PICK(32, 16, 16);
  else if (Ca == 16 && Cb == 8 && Cout == 8) FPN_PICK(16, 8, 8);
  else return CDS_EINVAL;
#undef FPN_PICK
#undef FPN_LAUNCH
  return cds_launch_status();
}

extern "C" int cds_instnorm_stats_cl_parts(int H, int W) {
  const int wg = cds_ceil_div(H * W, 256 * 8);
  return wg < 1 ? 1 : (wg > 512 ? 512 : wg);
}

// InstanceNorm records of x [N][H][W][C] (C in {8, 16, 32}): partial [N][cds_instnorm_stats_cl_parts(H, W)][C][2] doubles; reduce
// with cds_instnorm_reduce_f32.
extern "C" int cds_instnorm_stats_cl_f32(const float* x, double* partial, int N, int C, int H, int W, void* stream) {
  if (!x || !partial || N < 1 || H < 1 || W < 1) return CDS_EINVAL;
  const dim3 grid(cds_instnorm_stats_cl_parts(H, W), N);
  hipStream_t st = (hipStream_t)stream;
  if (C == 8) hipLaunchKernelGGL(instnorm_stats_cl_kernel<8>, grid, dim3(256), 0, st, x, partial, H * W);
  else if (C == 16) hipLaunchKernelGGL(instnorm_stats_cl_kernel<16>, grid, dim3(256), 0, st, x, partial, H * W);
  else if (C == 32) hipLaunchKernelGGL(instnorm_stats_cl_kernel<32>, grid, dim3(256), 0, st, x, partial, H * W);
  else return CDS_EINVAL;
  return cds_launch_status();
}

// InstanceNorm + activation for given statistics [N][C][2] doubles (sum, sum of squares) on x [N][H][W][C]: the images n >= cl_from
// channels-last into out_cl [N - cl_from][H][W][C] (or NULL), the first n_chw images planar into out_chw [n_chw][C][H][W] (or NULL
// with n_chw = 0).  C in {8, 16, 32}.
extern "C" int cds_instnorm_apply_cl_f32(const float* x, const double* stats, float* out_cl, float* out_chw, int N, int C, int H, int W,
                                         int act, int n_chw, int cl_from, void* stream) {
  if (!x || !stats || N < 1 || H < 1 || W < 1 || n_chw < 0 || n_chw > N || (n_chw > 0 && !out_chw) || (!out_cl && n_chw < 1) ||
      cl_from < 0 || (out_cl && cl_from >= N))
    return CDS_EINVAL;
  const dim3 grid(cds_ceil_div(H * W, 256), N);
  hipStream_t st = (hipStream_t)stream;
  if (C == 8) hipLaunchKernelGGL(instnorm_apply_cl_kernel<8>, grid, dim3(256), 0, st, x, stats, out_cl, out_chw, H * W, act, n_chw, cl_from);
  else if (C == 16) hipLaunchKernelGGL(instnorm_apply_cl_kernel<16>, grid, dim3(256), 0, st, x, stats, out_cl, out_chw, H * W, act, n_chw, cl_from);
  else if (C == 32) hipLaunchKernelGGL(instnorm_apply_cl_kernel<32>, grid, dim3(256), 0, st, x, stats, out_cl, out_chw, H * W, act, n_chw, cl_from);
  else return CDS_EINVAL;
  return cds_launch_status();
}

