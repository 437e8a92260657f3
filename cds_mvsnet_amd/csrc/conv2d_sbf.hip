// K7 on the matrix cores: all branch convolutions of ONE DynamicConv (models/dynamic_conv.py:112,116 — for every kernel
// size k the feature branch convs[k] and the 3-channel curvature branch att_convs[k], concatenated to Cout + 3 output
// channels) as one implicit GEMM per branch from ONE staged input tile, in the split-bf16 arithmetic of conv3d_sbf.hip:
// every fp32 operand is split exactly into three bf16 terms and a product is the six error-compensated partial products
// of order <= 2^-16 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (fp32-class error; the dense 2D feature
// convolutions are where north_star wants the matrix cores).
//
//   D[m = pixel][n = cout] += A[m][k] * B[k][n],  K-step = 32 = 4 taps x 8 input channels
//   input  x [N][Cin][H][W] fp32 planar, with the producing layer's InstanceNorm + LeakyReLU applied on load
//          (in_affine [N][Cin][3], like cds_conv2d_affine_f32); staged 8 channels per round, channels-last in LDS:
//          [position][term 0..2][8 ch] bf16, 48 B per position
//   A (data): lane l -> pixel (l & 15) of a 16-pixel x-run, tap (l >> 4) of the K-step: one ds_read_b128 per term
//   B (weights): host-split [branch][round][kstep][nblock][term][lane][8]: one coalesced 16-byte load per lane and term
//   C/D: lane l holds cout (l & 15), pixels (l >> 4) * 4 + 0..3 -> one 16-byte planar store per M-tile
//   output branches [K][N][Cout + 3][H][W] (what cds_dynconv_blend_*_f32 reads)
// Tile 32 x 8 pixels, wave = two rows, halo = the largest kernel's radius (<= 3: k in {1, 3, 5, 7}).
#include <stdlib.h>

#include "cds_common.hpp"
#include "feat_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

union BV {
  uint4 u;
  bf16x8 v;
};

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  f32x2 r = v - __builtin_convertvector(h, f32x2);
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  f32x2 r2 = r - __builtin_convertvector(m, f32x2);
  bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = *reinterpret_cast<uint32_t*>(&h);
  mid = *reinterpret_cast<uint32_t*>(&m);
  lo = *reinterpret_cast<uint32_t*>(&l);
}

#define SBF_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16((a).v, (b).v, acc, 0, 0, 0)

constexpr int POSB = 48;

// sum over the 16 lanes of a DPP row (all lanes end up with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}
constexpr int TX = 32, TY = 8, R = 3;                 // tile, halo
constexpr int IXP = TX + 8, IY = TY + 2 * R;          // column c <-> x = ox0 - 4 + c (16-byte aligned global rows)
constexpr int NPOS = IY * IXP;                        // 560 positions = 26.9 KB
constexpr int MAXB = 3;                               // branches (kernel sizes) per DynamicConv

struct Branches {
  int nb;
  int k[MAXB];        // kernel size
  int ks0[MAXB];      // first K-step of the branch inside a round's weight block
  int nks;            // K-steps per round over all branches
};

// The DynamicConv epilogue of cds_dynconv_blend_stats_f32 (dynamic_conv.py:113-122: epipolar projection of the curvature
// responses, 1x1 MLP with the BatchNorm folded in, softmax(./T), blend; plus the InstanceNorm records of the result), fused
// behind the branch convolutions: the [K][N][Cout + 3] branch tensor is never written or read back.
struct Blend {
  const float* w1;      // [4][K]
  const float* b1;      // [4]
  const float* w2;      // [K][4]
  float* out;           // [N][Cout][H][W]
  float* norm_curv;     // [N][H][W]
  double* partial;      // [N][parts = tiles][Cout][2]
  float temperature;
  const float* epi;     // [N][2] DEVICE: epipoles in pixels of this resolution (feat_common.hpp)
};

// NBR branches, NBLK 16-cout blocks (Cout + 3 <= 16 NBLK)
template <int NBR, int NBLK, int MODE>   // MODE 0: branch tensor; 1: blend epilogue fused; 2: ReLU (+ 1x1 head + sigmoid), one branch
__global__ __launch_bounds__(256, 2) void dynconv_branches_sbf_kernel(const float* __restrict__ x, const float* __restrict__ affine,
                                                                      const uint4* __restrict__ wsp, const float* __restrict__ bias,
                                                                      float* __restrict__ out, Branches br, int N, int Cin, int Co3,
                                                                      int H, int W, int tiles_x, int tiles_y, Blend bl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int lin = cds_xcd_remap(blockIdx.x, tiles_x * tiles_y * N);
  const int tx_i = lin % tiles_x;
  lin /= tiles_x;
  const int ty_i = lin % tiles_y, img = lin / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int ox0 = tx_i * TX, oy0 = ty_i * TY;
  const size_t plane = (size_t)H * W;
  const int rounds = Cin >> 3;

  f32x4 acc[NBR][NBLK][4];       // M-tiles: (row 0 | 1 of the wave) x (x-run 0 | 1)
#pragma unroll
  for (int b = 0; b < NBR; ++b)
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[b][nb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // lane-constant A base: pixel m of run 0, first row of this wave; column = 4 + x - R + kx, row = wave * 2 + ky
  const int a_base = ((wave * 2) * IXP + (4 - R) + m) * POSB;
  const uint4* __restrict__ wl = wsp + lane;

  for (int rd = 0; rd < rounds; ++rd) {
    if (rd) __syncthreads();
    // ---- stage 8 channels: unit = (row, x-quad): 8 float4 (one per channel), normalise-on-load, split, 4 positions ----
    // only the rows the largest kernel of this DynamicConv reads are staged (mode 2 is always 3x3): a (1, 3) layer stages 10 of the
    // 14 rows, a (3, 5) layer 12 - the load + normalise + 3-way split of a row is what these layers cost
#ifdef CDS_DYNCONV_FULL_HALO
    const int rmax = R;
#else
    int rmax = 0;
#pragma unroll
    for (int b = 0; b < NBR; ++b) rmax = max(rmax, (br.k[b] - 1) >> 1);
#endif
    const int ROW0 = MODE == 2 ? R - 1 : R - rmax, ROWS = MODE == 2 ? TY + 2 : TY + 2 * rmax;
    for (int u = tid; u < ROWS * (IXP / 4); u += 256) {
      const int row = ROW0 + u / (IXP / 4), q = u - (row - ROW0) * (IXP / 4);
      const int gy = oy0 - R + row, gx = ox0 - 4 + 4 * q;
      const bool ok = (unsigned)gy < (unsigned)H && gx >= 0 && gx + 3 < W;
      const float* __restrict__ src = x + ((size_t)img * Cin + rd * 8) * plane + (size_t)gy * W + gx;
      float4 v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v[c] = ok ? *reinterpret_cast<const float4*>(src + (size_t)c * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (affine && ok) {
          const float* __restrict__ af = affine + ((size_t)img * Cin + rd * 8 + c) * 3;
          const float al = af[0], be = af[1], sl = af[2];
          float t;
          t = fmaf(v[c].x, al, be); v[c].x = t > 0.f ? t : t * sl;
          t = fmaf(v[c].y, al, be); v[c].y = t > 0.f ? t : t * sl;
          t = fmaf(v[c].z, al, be); v[c].z = t > 0.f ? t : t * sl;
          t = fmaf(v[c].w, al, be); v[c].w = t > 0.f ? t : t * sl;
        }
      }
      unsigned char* dst = lds + (row * IXP + 4 * q) * POSB;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        uint32_t hi[4], mid[4], lo[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          const float a = p == 0 ? v[2 * c2].x : p == 1 ? v[2 * c2].y : p == 2 ? v[2 * c2].z : v[2 * c2].w;
          const float b = p == 0 ? v[2 * c2 + 1].x : p == 1 ? v[2 * c2 + 1].y : p == 2 ? v[2 * c2 + 1].z : v[2 * c2 + 1].w;
          split2(a, b, hi[c2], mid[c2], lo[c2]);
        }
        uint4* d4 = reinterpret_cast<uint4*>(dst + p * POSB);
        d4[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        d4[1] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        d4[2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
    __syncthreads();

    const uint4* __restrict__ wr = wl + (size_t)rd * br.nks * NBLK * 3 * 64;
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
      const int k = br.k[b], kk = k * k, rk = (k - 1) >> 1;
      const int nks = (kk + 3) >> 2;
      const uint4* __restrict__ wb = wr + (size_t)br.ks0[b] * NBLK * 3 * 64;
#pragma unroll 1
      for (int t = 0; t < nks; ++t) {
        int tap = 4 * t + g;
        if (tap >= kk) tap = kk - 1;                    // padded tap: zero weights, any in-tile data
        const int ky = tap / k, kx = tap - ky * k;
        const unsigned char* ap = lds + a_base + ((ky + R - rk) * IXP + (kx + R - rk)) * POSB;
        BV wh[NBLK], wm[NBLK], wlo[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
          const uint4* p = wb + (size_t)((t * NBLK + nb) * 3) * 64;
          wh[nb].u = p[0];
          wm[nb].u = p[64];
          wlo[nb].u = p[128];
        }
        BV ah[4], am[4], al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned char* a = ap + ((q >> 1) * IXP + (q & 1) * 16) * POSB;
          ah[q].u = *reinterpret_cast<const uint4*>(a);
          am[q].u = *reinterpret_cast<const uint4*>(a + 16);
          al[q].u = *reinterpret_cast<const uint4*>(a + 32);
        }
        __builtin_amdgcn_sched_barrier(0);
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
    }
  }

  if (MODE == 2) {
    // visibility CNN layer (model.py:14): ReLU(conv + folded BatchNorm), and for the last layer the 1x1 head + sigmoid
    // (bl.w1 = head weights [16], bl.b1 = head bias [1], bl.out = [N][H][W]); else out [N][16][H][W]
    const float bv = bias ? bias[m] : 0.f;
    const bool head = bl.w1 != nullptr;
    const float hw_n = head ? bl.w1[m] : 0.f, hb = head ? bl.b1[0] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + g * 4;
      const f32x4 a = acc[0][0][q];
      float v[4] = {fmaxf(a.x + bv, 0.f), fmaxf(a.y + bv, 0.f), fmaxf(a.z + bv, 0.f), fmaxf(a.w + bv, 0.f)};
      if (head) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sacc = row16_sum(v[i] * hw_n) + hb;
          v[i] = 1.0f / (1.0f + expf(-sacc));
        }
        if (m == 0 && oy < H && ox < W)
          *reinterpret_cast<float4*>(bl.out + (size_t)img * plane + (size_t)oy * W + ox) = make_float4(v[0], v[1], v[2], v[3]);
      } else if (m < Co3 && oy < H && ox < W) {
        *reinterpret_cast<float4*>(bl.out + ((size_t)img * Co3 + m) * plane + (size_t)oy * W + ox) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    return;
  }
  if (MODE == 1) {
    // Accumulator layout: lane (m, g) holds column 16 nb + m of pixels x = (q & 1) 16 + 4 g + i, y = 2 wave + (q >> 1).
    // (1) the lanes of the three curvature columns leave them in LDS; (2) lane m of a 16-lane group owns pixel (q, i) = (m >> 2,
    // m & 3) of its group: projection, MLP, softmax -> K weights into LDS, norm_curv to memory; (3) every lane reads the weights
    // of its 16 pixels and blends its column; InstanceNorm records per (wave, channel) as the separate kernel leaves them.
    const int Cout = Co3 - 3;
    float* attL = reinterpret_cast<float*>(lds);               // [b][j][256 pixels of the tile]
    float* wL = attL + NBR * 3 * 256;                           // [b][256]
    __syncthreads();                                            // every wave is done with the staged input tile
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
          *reinterpret_cast<float4*>(attL + (b * 3 + jj) * 256 + wave * 64 + q * 16 + g * 4) =
              make_float4(a.x + bv, a.y + bv, a.z + bv, a.w + bv);
        }
      }
    }
    {
      const int wp = wave * 64 + (m >> 2) * 16 + g * 4 + (m & 3);
      const int px = ox0 + ((m >> 2) & 1) * 16 + g * 4 + (m & 3), py = oy0 + wave * 2 + (m >> 3);
      float att[NBR][3], logit[NBR];
#pragma unroll
      for (int b = 0; b < NBR; ++b)
#pragma unroll
        for (int j = 0; j < 3; ++j) att[b][j] = attL[(b * 3 + j) * 256 + wp];
      const float nc = blend_from_att<NBR>(att, px, py, bl.epi[2 * img], bl.epi[2 * img + 1], bl.w1, bl.b1, bl.w2, bl.temperature, logit);
#pragma unroll
      for (int b = 0; b < NBR; ++b) wL[b * 256 + wp] = logit[b];
      if (px < W && py < H) bl.norm_curv[(size_t)img * plane + (size_t)py * W + px] = nc;
    }
    float4 wq[NBR][4];
#pragma unroll
    for (int b = 0; b < NBR; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[b][q] = *reinterpret_cast<const float4*>(wL + b * 256 + wave * 64 + q * 16 + g * 4);
    double* red = reinterpret_cast<double*>(lds + 16384);      // [wave][NBLK * 16][2]: the four waves' sums of a tile, added below
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      const int co = nb * 16 + m;
      const bool col = co < Cout;
      float bvb[NBR];
#pragma unroll
      for (int b = 0; b < NBR; ++b) bvb[b] = (bias && col) ? bias[b * Co3 + co] : 0.f;
      float* __restrict__ ob = bl.out + ((size_t)img * Cout + (col ? co : 0)) * plane;
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
        }
        if (col && oy < H && ox < W) {                           // W % 4 == 0
          *reinterpret_cast<float4*>(ob + (size_t)oy * W + ox) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
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
    __syncthreads();
    // one record per (tile, channel): the four waves in a fixed order
    if (tid < NBLK * 16 && tid < Cout) {
      const int parts = tiles_x * tiles_y;
      double* rec = bl.partial + (((size_t)img * parts + (size_t)(ty_i * tiles_x + tx_i)) * Cout + tid) * 2;
      rec[0] = (red[tid * 2] + red[(NBLK * 16 + tid) * 2]) + (red[(2 * NBLK * 16 + tid) * 2] + red[(3 * NBLK * 16 + tid) * 2]);
      rec[1] = (red[tid * 2 + 1] + red[(NBLK * 16 + tid) * 2 + 1]) + (red[(2 * NBLK * 16 + tid) * 2 + 1] + red[(3 * NBLK * 16 + tid) * 2 + 1]);
    }
    return;
  }
  // ---- epilogue: lane -> cout (l & 15) of block nb, pixels x = run * 16 + (l >> 4) * 4 + 0..3 ----
#pragma unroll
  for (int b = 0; b < NBR; ++b) {
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      const int co = nb * 16 + m;
      if (co >= Co3) continue;
      const float bv = bias ? bias[b * Co3 + co] : 0.f;
      float* __restrict__ ob = out + (((size_t)b * N + img) * Co3 + co) * plane;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int oy = oy0 + wave * 2 + (q >> 1), ox = ox0 + (q & 1) * 16 + g * 4;
        if (oy >= H || ox >= W) continue;              // W % 4 == 0
        const f32x4 a = acc[b][nb][q];
        *reinterpret_cast<float4*>(ob + (size_t)oy * W + ox) = make_float4(a.x + bv, a.y + bv, a.z + bv, a.w + bv);
      }
    }
  }
}

}  // namespace

// All branch convolutions of one DynamicConv (dynamic_conv.py:112,116), stride 1, "same" padding, in split-bf16 arithmetic.
// x [N][Cin][H][W] (+ in_affine [N][Cin][3] or NULL), weight_split from ops.split_pack_dynconv, bias [nb][Co3] or NULL,
// out [nb][N][Co3][H][W] with Co3 = Cout + 3.  ksizes: nb kernel sizes, each in {1, 3, 5, 7}.  Cin % 8 == 0, Co3 <= 48, W % 4 == 0.
extern "C" int cds_dynconv_branches_sbf_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                                            float* out, int N, int Cin, int Co3, int H, int W, const int* ksizes, int nb,
                                            void* stream) {
  if (!x || !weight_split || !out || !ksizes || N < 1 || Cin < 8 || (Cin % 8) || Co3 < 1 || Co3 > 48 || H < 1 || W < 4 || (W % 4) ||
      nb < 1 || nb > MAXB)
    return CDS_EINVAL;
  Branches br;
  br.nb = nb;
  int ks = 0;
  for (int b = 0; b < MAXB; ++b) {
    br.k[b] = b < nb ? ksizes[b] : 1;
    br.ks0[b] = ks;
    if (b < nb) {
      if (br.k[b] != 1 && br.k[b] != 3 && br.k[b] != 5 && br.k[b] != 7) return CDS_EINVAL;
      ks += (br.k[b] * br.k[b] + 3) / 4;
    }
  }
  br.nks = ks;
  const int nblk = (Co3 + 15) / 16;
  const int tx = cds_ceil_div(W, TX), ty = cds_ceil_div(H, TY);
  const dim3 grid(tx * ty * N), block(256);
  const size_t ldsb = (size_t)NPOS * POSB;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(NBR, NBLK)                                                                                                    \
  hipLaunchKernelGGL((dynconv_branches_sbf_kernel<NBR, NBLK, 0>), grid, block, ldsb, st, x, in_affine,                   \
                     reinterpret_cast<const uint4*>(weight_split), bias, out, br, N, Cin, Co3, H, W, tx, ty, Blend{})
  if (nb == 3 && nblk == 1) LAUNCH(3, 1);
  else if (nb == 2 && nblk == 1) LAUNCH(2, 1);
  else if (nb == 2 && nblk == 2) LAUNCH(2, 2);
  else if (nb == 2 && nblk == 3) LAUNCH(2, 3);
  else if (nb == 3 && nblk == 2) LAUNCH(3, 2);
  else return CDS_EINVAL;
#undef LAUNCH
  return cds_launch_status();
}

// Records per image that cds_dynconv_fused_sbf_f32 leaves for cds_instnorm_reduce_f32: one per 32 x 8 tile.
extern "C" int cds_dynconv_fused_parts(int H, int W) { return cds_ceil_div(W, TX) * cds_ceil_div(H, TY); }

// One DynamicConv (dynamic_conv.py:97-122) in ONE kernel: the branch convolutions of cds_dynconv_branches_sbf_f32 with the
// epilogue of cds_dynconv_blend_stats_f32 applied to the accumulators.  out [N][Cout][H][W] (before its InstanceNorm),
// norm_curv [N][H][W], partial: 8-byte aligned scratch of 2 * N * parts * Cout doubles, parts = cds_dynconv_fused_parts(H, W)
// (reduce with cds_instnorm_reduce_f32).  w1 [4][K], b1 [4], w2 [K][4]: the attention MLP with its BatchNorm folded in;
// epipoles [N][2] pixels at this resolution.  Same shape limits as cds_dynconv_branches_sbf_f32; N <= CDS_MAX_IMAGES.
extern "C" int cds_dynconv_fused_sbf_f32(const float* x, const float* in_affine, const void* weight_split, const float* bias,
                                         const float* w1, const float* b1, const float* w2, const float* epipoles,
                                         float temperature, float* out, float* norm_curv, double* partial, int N, int Cin,
                                         int Cout, int H, int W, const int* ksizes, int nb, void* stream) {
  const int Co3 = Cout + 3;
  if (!x || !weight_split || !w1 || !b1 || !w2 || !epipoles || !out || !norm_curv || !partial || !ksizes || N < 1 ||
      N > CDS_MAX_IMAGES || Cin < 8 || (Cin % 8) || Cout < 1 || Co3 > 48 || (Cout % 16) + 2 > 15 || H < 1 || W < 4 || (W % 4) ||
      nb < 2 || nb > MAXB)
    return CDS_EINVAL;
  Branches br;
  br.nb = nb;
  int ks = 0;
  for (int b = 0; b < MAXB; ++b) {
    br.k[b] = b < nb ? ksizes[b] : 1;
    br.ks0[b] = ks;
    if (b < nb) {
      if (br.k[b] != 1 && br.k[b] != 3 && br.k[b] != 5 && br.k[b] != 7) return CDS_EINVAL;
      ks += (br.k[b] * br.k[b] + 3) / 4;
    }
  }
  br.nks = ks;
  Blend bl;
  bl.w1 = w1; bl.b1 = b1; bl.w2 = w2; bl.out = out; bl.norm_curv = norm_curv; bl.partial = partial; bl.temperature = temperature;
  bl.epi = epipoles;
  const int nblk = (Co3 + 15) / 16;
  const int tx = cds_ceil_div(W, TX), ty = cds_ceil_div(H, TY);
  const dim3 grid(tx * ty * N), block(256);
  const size_t ldsb = (size_t)NPOS * POSB;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCHF(NBR, NBLK)                                                                                                   \
  hipLaunchKernelGGL((dynconv_branches_sbf_kernel<NBR, NBLK, 1>), grid, block, ldsb, st, x, in_affine,                    \
                     reinterpret_cast<const uint4*>(weight_split), bias, nullptr, br, N, Cin, Co3, H, W, tx, ty, bl)
  if (nb == 3 && nblk == 1) LAUNCHF(3, 1);
  else if (nb == 2 && nblk == 1) LAUNCHF(2, 1);
  else if (nb == 2 && nblk == 2) LAUNCHF(2, 2);
  else if (nb == 2 && nblk == 3) LAUNCHF(2, 3);
  else if (nb == 3 && nblk == 2) LAUNCHF(3, 2);
  else return CDS_EINVAL;
#undef LAUNCHF
  return cds_launch_status();
}


// 3x3 convolution 8k -> 16 channels (pad 1) + bias + ReLU in split-bf16 arithmetic on the matrix cores, optionally followed by
// a 1x1 head (16 -> 1) + sigmoid: the visibility CNN's layers 2 and 3 + head (models/model.py:14; BatchNorm folded by the
// caller).  x [N][Cin][H][W], weight_split from ops.split_pack_dynconv([w]) with w [16][Cin][3][3], bias [16];
// head_w [16] / head_b [1] or both NULL; out [N][16][H][W], or [N][H][W] with the head.  Cin % 8 == 0, W % 4 == 0.
extern "C" int cds_conv2d_k3_relu_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* head_w,
                                          const float* head_b, float* out, int N, int Cin, int H, int W, void* stream) {
  if (!x || !weight_split || !out || N < 1 || Cin < 8 || (Cin % 8) || H < 1 || W < 4 || (W % 4) ||
      (head_w != nullptr) != (head_b != nullptr))
    return CDS_EINVAL;
  Branches br;
  br.nb = 1;
  for (int b = 0; b < MAXB; ++b) {
    br.k[b] = b == 0 ? 3 : 1;
    br.ks0[b] = b == 0 ? 0 : 3;
  }
  br.nks = 3;
  Blend bl{};
  bl.w1 = head_w;
  bl.b1 = head_b;
  bl.out = out;
  const int tx = cds_ceil_div(W, TX), ty = cds_ceil_div(H, TY);
  hipLaunchKernelGGL((dynconv_branches_sbf_kernel<1, 1, 2>), dim3(tx * ty * N), dim3(256), (size_t)NPOS * POSB, (hipStream_t)stream, x,
                     nullptr, reinterpret_cast<const uint4*>(weight_split), bias, nullptr, br, N, Cin, 16, H, W, tx, ty, bl);
  return cds_launch_status();
}
