// Training kernels of the 2D stacks: FeatureNet / DynamicConv, visibility CNN, Refinement, soft-argmin.
//   reference (forward; the backward is what torch.autograd derives from it):
//     models/dynamic_conv.py:97-122   DynamicConv.forward
//     models/module.py:28-71          Conv2dUnit (conv -> InstanceNorm2d -> LeakyReLU(0.1))
//     models/module.py:234-267        FeatureNet.forward
//     models/model.py:14,51           visibility CNN
//     models/module.py:318-370        Refinement
//     models/module.py:373-379        depth_regression
//
//   cds_conv2d_wgrad_f32        dw[co][ci][ky][kx] += sum_{n,o} g[n][co][o] x[n][ci][S o - pad + k]   (fp32 matrix pipe)
//   cds_conv2d_dgrad_s2_f32     data gradient of a 3x3 stride-2 pad-1 convolution (the two down-sampling units)
//   cds_instnorm_bwd_f32        InstanceNorm2d + LeakyReLU(0.1) | tanh backward: statistics pass + apply pass
//   cds_dynconv_bn_stats_f32 / cds_dynconv_blend_train_f32
//                               DynamicConv epilogue with the BATCH statistics of the attention MLP's BatchNorm2d
//   cds_dynconv_blend_bwd_f32   its backward: two passes (BatchNorm backward sums, then the branch gradients)
//   cds_softargmin_bwd_f32      backward of softmax over hypotheses + expectation
//
// The stride-1 data gradients are the forward kernels of conv2d.hip on flipped, transposed weights (CDS_ACT_ACCUM adds onto the
// output, so the K branches of a DynamicConv accumulate into one gradient tensor).
#include <stdlib.h>

#include "cds_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 STORAGE of the 2D activations (the "bf16-storage / f32-accumulate" policy of the training step): tensors that live from the forward to
// the backward pass (layer inputs, DynamicConv branch responses, pre-normalisation maps) are raw bfloat16; every kernel widens on load
// and accumulates in fp32 / fp64 exactly as on the fp32 path.  Rounding is round-to-nearest-even (what torch's .bfloat16() does).
typedef unsigned short b16;
__device__ __forceinline__ float ldf(const float* __restrict__ p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const b16* __restrict__ p, size_t i) { return __uint_as_float((unsigned)p[i] << 16); }
__device__ __forceinline__ b16 f2b(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (b16)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (b16)(u >> 16);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), the 2D form of
// conv3d_wgrad_mfma_kernel:  D[a][n] += A[a][p] B[p][n],  a = 16 output channels of g, n = (input channel, tap) column of a chunk of
// CB input channels, p = output pixel.  A workgroup stages a 32 x 8 tile of g (16 channels) and the matching input tile with its
// halo; the four waves own a quarter of the tile's pixels each and ALL column blocks (any number of blocks is balanced), a lane
// reads one float of g per K-step and one float of the input tile per MFMA.  The waves' partial tiles are summed through LDS and
// leave as one atomic per weight and workgroup.
// ---------------------------------------------------------------------------------------------------------------------------
template <int K, int S, int CB>
struct W2Cfg {
  static constexpr int OX = 32, OY = 8, NO = OX * OY;
  static constexpr int GS = NO + 4;                      // row stride of the g tile (bank spread of the 16 channel rows)
  static constexpr int IX = (OX - 1) * S + K, IY = (OY - 1) * S + K, NI = IX * IY;
  static constexpr int NCOL = CB * K * K;
  static constexpr int NBLK = (NCOL + 15) / 16;
  static constexpr int LDS_IN = 16 * GS + CB * NI;
  static constexpr int LDS_RED = 4 * NBLK * 256;         // the four waves' accumulators for the final reduction
  static constexpr int LDS_FLOATS = LDS_IN > LDS_RED ? LDS_IN : LDS_RED;
};

template <int K, int S, int CB, typename TX>
__global__ __launch_bounds__(256) void conv2d_wgrad_mfma_kernel(const float* __restrict__ g, const TX* __restrict__ xin,
                                                                float* __restrict__ dw, int N, int Co, int Cin, int Ho, int Wo,
                                                                int H, int W, int pad, int tiles_x, int ntiles, int tiles_per_wg) {
  using Cfg = W2Cfg<K, S, CB>;
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  float* lg = wlds;                        // [16][GS]
  float* lx = wlds + 16 * Cfg::GS;         // [CB][NI]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int a0 = blockIdx.y * 16, b0 = blockIdx.z * CB;
  const size_t po = (size_t)Ho * Wo, pi = (size_t)H * W;
  int colofs[Cfg::NBLK];
#pragma unroll
  for (int q = 0; q < Cfg::NBLK; ++q) {
    const int n = min(q * 16 + j, Cfg::NCOL - 1);
    const int b = n / (K * K), tap = n - b * (K * K);
    colofs[q] = b * Cfg::NI + (tap / K) * Cfg::IX + tap % K;
  }
  f32x4 acc[Cfg::NBLK];
#pragma unroll
  for (int q = 0; q < Cfg::NBLK; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles * N, t0 + tiles_per_wg);
  // Staging through registers (see conv3d_wgrad_mfma_kernel): per-thread element offsets computed once, all loads of a tile issued back
  // to back from clamped addresses, and the loads of tile t + 1 in flight during the K-steps of tile t.  The guarded scalar loop this
  // replaces compiled to one load + s_waitcnt vmcnt(0) per element: 80 % of the kernel's time.
  constexpr int NG = 16 * Cfg::NO / 256, NX = (CB * Cfg::NI + 255) / 256;
  float rg[NG], rx[NX];
  int relx[NX], pkx[NX];
  const int gpx = tid % Cfg::OX, gpy = tid / Cfg::OX;      // NO == 256: element e of the g tile is channel e, pixel tid
  static_assert(Cfg::NO == 256, "the g tile is one pixel per thread and channel");
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    const int i = tid + 256 * e;
    const int ch = i / Cfg::NI, p = i - ch * Cfg::NI;
    const int px = p % Cfg::IX, py = p / Cfg::IX;
    relx[e] = ch * (int)pi + py * W + px;
    pkx[e] = (i < CB * Cfg::NI && b0 + ch < Cin) ? (px | (py << 16)) : -1;
  }
  auto fetch = [&](int tile) {
    const int n = tile / ntiles;
    const int r = tile - n * ntiles;
    const int ox0 = (r % tiles_x) * Cfg::OX, oy0 = (r / tiles_x) * Cfg::OY;
    const bool gok = ox0 + gpx < Wo && oy0 + gpy < Ho;
    const float* __restrict__ gt = g + ((size_t)n * Co + a0) * po + (size_t)(oy0 + gpy) * Wo + ox0 + gpx;
#pragma unroll
    for (int e = 0; e < NG; ++e) {
      const bool ok = gok && a0 + e < Co;
      const float v = *(ok ? gt + (size_t)e * po : g);
      rg[e] = ok ? v : 0.f;
    }
    const int ix0 = ox0 * S - pad, iy0 = oy0 * S - pad;
    const long long xbase = (long long)(((size_t)n * Cin + b0) * pi) + (long long)iy0 * W + ix0;
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int pk = pkx[e];
      const bool ok = pk >= 0 && (unsigned)(ix0 + (pk & 0xffff)) < (unsigned)W && (unsigned)(iy0 + (pk >> 16)) < (unsigned)H;
      const float v = ldf(xin, (size_t)(ok ? xbase + relx[e] : 0));
      rx[e] = ok ? v : 0.f;
    }
  };
  if (t0 < t1) fetch(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NG; ++e) lg[e * Cfg::GS + tid] = rg[e];
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int i = tid + 256 * e;
      if (i < CB * Cfg::NI) lx[i] = rx[e];
    }
    __syncthreads();
    if (tile + 1 < t1) fetch(tile + 1);
    const int pw = wave * (Cfg::NO / 4);
#pragma unroll 2
    for (int ks = 0; ks < Cfg::NO / 16; ++ks) {
      const int p = pw + 4 * ks + kq;                    // this lane's pixel of the K-step
      const int base = ((p / Cfg::OX) * S) * Cfg::IX + (p % Cfg::OX) * S;
      const float av = lg[j * Cfg::GS + p];
#pragma unroll
      for (int q = 0; q < Cfg::NBLK; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, lx[base + colofs[q]], acc[q], 0, 0, 0);
    }
  }
  // sum the four waves' tiles: red[wave][q][r][lane]
  __syncthreads();
  float* red = wlds;
#pragma unroll
  for (int q = 0; q < Cfg::NBLK; ++q) {
    red[((wave * Cfg::NBLK + q) * 4 + 0) * 64 + lane] = acc[q].x;
    red[((wave * Cfg::NBLK + q) * 4 + 1) * 64 + lane] = acc[q].y;
    red[((wave * Cfg::NBLK + q) * 4 + 2) * 64 + lane] = acc[q].z;
    red[((wave * Cfg::NBLK + q) * 4 + 3) * 64 + lane] = acc[q].w;
  }
  __syncthreads();
  // D layout: lane (j, kq) holds rows a = 4 kq + r of column n = 16 q + j
  for (int i = tid; i < Cfg::NBLK * 256; i += 256) {
    const int l = i & 63, r = (i >> 6) & 3, q = i >> 8;
    const float v = red[i] + red[Cfg::NBLK * 256 + i] + red[2 * Cfg::NBLK * 256 + i] + red[3 * Cfg::NBLK * 256 + i];
    const int n = q * 16 + (l & 15), a = a0 + 4 * (l >> 4) + r;
    if (n >= Cfg::NCOL || a >= Co) continue;
    const int b = n / (K * K), tap = n - b * (K * K);
    if (b0 + b >= Cin) continue;
    atomicAdd(&dw[((size_t)a * Cin + b0 + b) * (K * K) + tap], v);
  }
}

template <int K, int S, int CB, typename TX>
int launch_wgrad2d(const float* g, const TX* xin, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H, int W, int pad,
                   hipStream_t st) {
  using Cfg = W2Cfg<K, S, CB>;
  const int tx = cds_ceil_div(Wo, Cfg::OX), ty = cds_ceil_div(Ho, Cfg::OY);
  const int ntiles = tx * ty;
  const int ab = cds_ceil_div(Co, 16) * cds_ceil_div(Cin, CB);
  int per = cds_ceil_div(ntiles * N * ab, cds_env_int("CDS_WG2_WGS", 512));         // ~512 workgroups in all: each ends with one atomic per weight of its block
  if (per < 1) per = 1;
  const dim3 grid(cds_ceil_div(ntiles * N, per), cds_ceil_div(Co, 16), cds_ceil_div(Cin, CB));
  const int ldsb = Cfg::LDS_FLOATS * (int)sizeof(float);
  static std::atomic<unsigned long long> ok{0};
  if (ldsb > 64 * 1024)
    if (int e = cds_allow_lds((const void*)conv2d_wgrad_mfma_kernel<K, S, CB, TX>, ldsb, ok)) return e;
  hipLaunchKernelGGL((conv2d_wgrad_mfma_kernel<K, S, CB, TX>), grid, dim3(256), ldsb, st, g, xin, dw, N, Co, Cin, Ho, Wo, H, W, pad, tx,
                     ntiles, per);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Data gradient of conv 3x3, stride 2, pad 1:  gx[n][ci][y][x] = sum_co sum_{ky,kx : (y+1-ky), (x+1-kx) even} g[n][co][(y+1-ky)/2][(x+1-kx)/2]
// w[co][ci][ky][kx].  One thread per input pixel and CI input channels; weights [Co][Cin][3][3] through the scalar cache.
// ---------------------------------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(256) void conv2d_dgrad_s2_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                              float* __restrict__ gx, int Co, int Cin, int Ho, int Wo, int H, int W) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z / (Cin / CI), c0 = (blockIdx.z % (Cin / CI)) * CI;
  if (x >= W || y >= H) return;
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.f;
  const size_t po = (size_t)Ho * Wo;
  const float* __restrict__ gn = g + (size_t)n * Co * po;
  // rows: y even -> ky = 1; y odd -> ky = 0 (oy = (y + 1) / 2) and ky = 2 (oy = (y - 1) / 2); same along x
  for (int co = 0; co < Co; ++co) {
    const float* __restrict__ wc = w + ((size_t)co * Cin + c0) * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (((y + 1 - ky) & 1) != 0) continue;
      const int oy = (y + 1 - ky) >> 1;
      if (oy < 0 || oy >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int t = x + 1 - kx;
        const int ox = t >> 1;
        const bool ok = (t & 1) == 0 && ox >= 0 && ox < Wo;
        const float gv = ok ? gn[(size_t)co * po + (size_t)oy * Wo + ox] : 0.f;
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[c] = fmaf(gv, wc[c * 9 + ky * 3 + kx], acc[c]);
      }
    }
  }
  const size_t pi = (size_t)H * W;
#pragma unroll
  for (int c = 0; c < CI; ++c) gx[((size_t)n * Cin + c0 + c) * pi + (size_t)y * W + x] = acc[c];
}

// ---------------------------------------------------------------------------------------------------------------------------
// InstanceNorm2d + activation backward.  z = act(xhat), xhat = (y - mean) rstd;  gh = gz act'(xhat);
//   gy = rstd (gh - mean(gh) - xhat mean(gh xhat))     (means over the H W pixels of one (image, channel))
// stats: the forward's fp64 (sum, sum of squares) per (image, channel).  Pass 1 leaves (sum gh, sum gh xhat) in fp64.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void in_stats(const double* __restrict__ st, int hw, float& mean, float& rstd) {
  const double m = st[0] / hw;
  double var = st[1] / hw - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + 1e-5));
}

__device__ __forceinline__ float act_grad(float xhat, int act) {
  if (act == CDS_ACT_LEAKY01) return xhat > 0.f ? 1.f : 0.1f;
  if (act == CDS_ACT_TANH) {
    const float t = tanhf(xhat);
    return 1.f - t * t;
  }
  return 1.f;
}

template <typename TY>
__global__ __launch_bounds__(256) void instnorm_bwd_reduce_kernel(const float* __restrict__ gz, const TY* __restrict__ y,
                                                                  const double* __restrict__ stats, double* __restrict__ sums,
                                                                  int hw, int act, int blocks_per_c, const b16* __restrict__ zs) {
  const int c = blockIdx.x / blocks_per_c, b = blockIdx.x % blocks_per_c;   // c = image * C + channel
  float mean, rstd;
  in_stats(stats + 2 * c, hw, mean, rstd);
  const TY* __restrict__ yc = y + (size_t)c * hw;
  const float* __restrict__ gc = gz + (size_t)c * hw;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
  for (int i = b * 256 + threadIdx.x; i < hw; i += blocks_per_c * 256) {
    const float xh = (ldf(yc, i) - mean) * rstd;
    const float gh = gc[i] * (zs ? ((zs[(size_t)c * hw + i] & 0x8000u) ? 0.1f : 1.f) : act_grad(xh, act));
    s0 += (double)gh;
    s1 += (double)gh * (double)xh;
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  __shared__ double red[2][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[2 * c + 0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[2 * c + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

template <typename TY>
__global__ __launch_bounds__(256) void instnorm_bwd_apply_kernel(const float* __restrict__ gz, const TY* __restrict__ y,
                                                                 const double* __restrict__ stats, const double* __restrict__ sums,
                                                                 float* __restrict__ gy, int hw, int act, const b16* __restrict__ zs) {
  const int c = blockIdx.y;
  float mean, rstd;
  in_stats(stats + 2 * c, hw, mean, rstd);
  const float m0 = (float)(sums[2 * c] / hw), m1 = (float)(sums[2 * c + 1] / hw);
  const size_t o = (size_t)c * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
    const float xh = (ldf(y, o + i) - mean) * rstd;
    const float gh = gz[o + i] * (zs ? ((zs[o + i] & 0x8000u) ? 0.1f : 1.f) : act_grad(xh, act));
    gy[o + i] = rstd * (gh - m0 - xh * m1);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// DynamicConv epilogue in training mode (dynamic_conv.py:97-122).  branch [K][N][Cout + 3][H][W]: per kernel size the Cout
// convolution responses followed by the 3 curvature responses.  Per pixel:
//   basis = (u^2, 2 u v, v^2), (u, v) = unit vector from the epipole;   curv_k = <att_k, basis>
//   h_j = sum_k W1[j][k] curv_k;  hbn_j = (h_j - mean_j) rstd_j gamma_j + beta_j;  r_j = max(hbn_j, 0)        (BatchNorm2d of the MLP)
//   aw_k = sum_j W2[k][j] r_j;  wts = softmax(aw / T);   y_c = sum_k res_kc wts_k;   nc = sum_k curv_k wts_k
// (mean_j, rstd_j) are BATCH statistics over the (images of a group) x H x W values of h_j, one set per group of N / G images (a
// group = what one call of the reference's FeatureNet sees).  h is linear in curv, so the statistics follow from the first and
// second moments of curv (K + K (K + 1) / 2 sums in fp64) - one pass over the curvature channels, no pass over h.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int HID = 4;                 // hidden width of the attention MLP (dynamic_conv.py:84)
constexpr int TPX = 8;                 // pixels per thread of the epilogue kernels

template <typename TB>
struct BlendArgsT {
  const TB* branch;                    // [K][N][Cout + 3][hw], fp32 or raw bf16
  const float* epi;                    // [N][2] device
  const float* w1;                     // [HID][K]
  const float* w2;                     // [K][HID]
  const float* gamma;                  // [HID]
  const float* beta;                   // [HID]
  const float* mean;                   // [G][HID]   (batch or running statistics, as cds_dynconv_bn_fold_f32 left them)
  const float* rstd;                   // [G][HID]
  int N, G, Cout, H, W;
  float inv_T;
};

template <int K>
struct PixelState {
  float basis[3], curv[K], hhat[HID], r[HID], wts[K];
};

template <int K, typename TB>
__device__ __forceinline__ void pixel_basis_curv(const BlendArgsT<TB>& a, int n, int p, PixelState<K>& s) {
  const int hw = a.H * a.W;
  const int y = p / a.W, x = p - y * a.W;
  float u = (float)x - a.epi[2 * n], v = (float)y - a.epi[2 * n + 1];
  const float nrm = sqrtf(u * u + v * v);
  u = u / (nrm + 1e-6f);
  v = v / (nrm + 1e-6f);
  s.basis[0] = u * u;
  s.basis[1] = 2.0f * u * v;
  s.basis[2] = v * v;
  const size_t bstride = (size_t)a.N * (a.Cout + 3) * hw;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const TB* c = a.branch + k * bstride + ((size_t)n * (a.Cout + 3) + a.Cout) * hw + p;
    s.curv[k] = ldf(c, 0) * s.basis[0] + ldf(c, hw) * s.basis[1] + ldf(c, 2 * (size_t)hw) * s.basis[2];
  }
}

template <int K, typename TB>
__device__ __forceinline__ void pixel_weights(const BlendArgsT<TB>& a, int g, PixelState<K>& s) {
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    float h = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) h = fmaf(a.w1[j * K + k], s.curv[k], h);
    s.hhat[j] = (h - a.mean[g * HID + j]) * a.rstd[g * HID + j];
    s.r[j] = fmaxf(s.hhat[j] * a.gamma[j] + a.beta[j], 0.f);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < HID; ++j) t = fmaf(a.w2[k * HID + j], s.r[j], t);
    s.wts[k] = t * a.inv_T;
    mx = fmaxf(mx, s.wts[k]);
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s.wts[k] = expf(s.wts[k] - mx);
    den += s.wts[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) s.wts[k] = s.wts[k] / den;
}

// block-level sum of NV per-thread doubles, then one atomic per value: dst[i] += sum
template <int NV>
__device__ __forceinline__ void block_atomic_sum(double (&v)[NV], double* __restrict__ dst) {
  __shared__ double red[4][NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double t = wave_sum_d(v[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) atomicAdd(&dst[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// mom [G][K + K (K + 1) / 2] fp64 (zeroed): sums of curv_k, then of curv_k curv_l (k <= l, row-major upper triangle)
template <int K, typename TB>
__global__ __launch_bounds__(256) void dynconv_moments_kernel(BlendArgsT<TB> a, double* __restrict__ mom) {
  constexpr int NM = K + K * (K + 1) / 2;
  const int hw = a.H * a.W, n = blockIdx.y, g = n / (a.N / a.G);
  double acc[NM];
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i] = 0.0;
  for (int t = 0; t < TPX; ++t) {
    const int p = (blockIdx.x * TPX + t) * 256 + threadIdx.x;
    if (p >= hw) break;
    PixelState<K> s;
    pixel_basis_curv<K, TB>(a, n, p, s);
    int i = K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      acc[k] += (double)s.curv[k];
#pragma unroll
      for (int l = k; l < K; ++l) acc[i++] += (double)s.curv[k] * (double)s.curv[l];
    }
  }
  block_atomic_sum<NM>(acc, mom + (size_t)g * NM);
}

// One thread per hidden channel j: batch statistics of h_j for every group (in call order, for the running statistics).
__global__ void dynconv_bn_fold_kernel(const double* __restrict__ mom, const float* __restrict__ w1, int K, int G, double count,
                                       double eps, float momentum, int use_batch, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ rstd) {
  const int j = threadIdx.x;
  if (j >= HID) return;
  const int NM = K + K * (K + 1) / 2;
  for (int g = 0; g < G; ++g) {
    if (!use_batch) {
      mean[g * HID + j] = running_mean[j];
      rstd[g * HID + j] = (float)(1.0 / sqrt((double)running_var[j] + eps));
      continue;
    }
    const double* m = mom + (size_t)g * NM;
    double mu = 0.0, e2 = 0.0;
    int i = K;
    for (int k = 0; k < K; ++k) {
      mu += (double)w1[j * K + k] * m[k];
      for (int l = k; l < K; ++l, ++i) e2 += (k == l ? 1.0 : 2.0) * (double)w1[j * K + k] * (double)w1[j * K + l] * m[i];
    }
    mu /= count;
    double var = e2 / count - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    mean[g * HID + j] = (float)mu;
    rstd[g * HID + j] = (float)(1.0 / sqrt(var + eps));
    if (running_mean) {
      running_mean[j] = running_mean[j] * (1.0f - momentum) + momentum * (float)mu;
      running_var[j] = running_var[j] * (1.0f - momentum) + momentum * (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
    }
  }
}

// out [N][Cout][hw], norm_curv [N][hw]
template <int K, typename TB>
__global__ __launch_bounds__(256) void dynconv_blend_train_kernel(BlendArgsT<TB> a, float* __restrict__ out, float* __restrict__ norm_curv) {
  const int hw = a.H * a.W, n = blockIdx.y, g = n / (a.N / a.G);
  const size_t bstride = (size_t)a.N * (a.Cout + 3) * hw;
  for (int t = 0; t < TPX; ++t) {
    const int p = (blockIdx.x * TPX + t) * 256 + threadIdx.x;
    if (p >= hw) break;
    PixelState<K> s;
    pixel_basis_curv<K, TB>(a, n, p, s);
    pixel_weights<K, TB>(a, g, s);
    float nc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) nc = nc + s.curv[k] * s.wts[k];
    norm_curv[(size_t)n * hw + p] = nc;
    const TB* __restrict__ res = a.branch + (size_t)n * (a.Cout + 3) * hw + p;
#pragma unroll 4                                   // Cout is a multiple of 8: four channels of loads in flight
    for (int c = 0; c < a.Cout; ++c) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) v = v + ldf(res, k * bstride + (size_t)c * hw) * s.wts[k];
      out[((size_t)n * a.Cout + c) * hw + p] = v;
    }
  }
}

// The per-pixel backward up to the BatchNorm: g_aw, g_hbn (already masked by the ReLU).
template <int K, typename TB>
__device__ __forceinline__ void pixel_backward(const BlendArgsT<TB>& a, int n, int p, const PixelState<K>& s, const float* __restrict__ gy,
                                               float gnc, float (&g_aw)[K], float (&g_hbn)[HID]) {
  const int hw = a.H * a.W;
  const size_t bstride = (size_t)a.N * (a.Cout + 3) * hw;
  const TB* __restrict__ res = a.branch + (size_t)n * (a.Cout + 3) * hw + p;
  float g_w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) g_w[k] = gnc * s.curv[k];
#pragma unroll 4                                   // Cout is a multiple of 8: four channels of loads in flight
  for (int c = 0; c < a.Cout; ++c) {
    const float gv = gy[((size_t)n * a.Cout + c) * hw + p];
#pragma unroll
    for (int k = 0; k < K; ++k) g_w[k] = fmaf(gv, ldf(res, k * bstride + (size_t)c * hw), g_w[k]);
  }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) dot = fmaf(s.wts[k], g_w[k], dot);
#pragma unroll
  for (int k = 0; k < K; ++k) g_aw[k] = s.wts[k] * (g_w[k] - dot) * a.inv_T;
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) t = fmaf(a.w2[k * HID + j], g_aw[k], t);
    g_hbn[j] = s.r[j] > 0.f ? t : 0.f;
  }
}

// Pass 1.  sums (fp64, zeroed): [G][2 HID] = per group (sum g_hbn_j, sum g_hbn_j hhat_j), then dW2 [K][HID] over all images.
template <int K, typename TB>
__global__ __launch_bounds__(256) void dynconv_blend_bwd_reduce_kernel(BlendArgsT<TB> a, const float* __restrict__ gy,
                                                                      const float* __restrict__ gnc, double* __restrict__ sums) {
  const int hw = a.H * a.W, n = blockIdx.y, g = n / (a.N / a.G);
  double acc[2 * HID], accw[K * HID];
#pragma unroll
  for (int i = 0; i < 2 * HID; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < K * HID; ++i) accw[i] = 0.0;
  for (int t = 0; t < TPX; ++t) {
    const int p = (blockIdx.x * TPX + t) * 256 + threadIdx.x;
    if (p >= hw) break;
    PixelState<K> s;
    pixel_basis_curv<K, TB>(a, n, p, s);
    pixel_weights<K, TB>(a, g, s);
    float g_aw[K], g_hbn[HID];
    pixel_backward<K, TB>(a, n, p, s, gy, gnc ? gnc[(size_t)n * hw + p] : 0.f, g_aw, g_hbn);
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      acc[j] += (double)g_hbn[j];
      acc[HID + j] += (double)g_hbn[j] * (double)s.hhat[j];
#pragma unroll
      for (int k = 0; k < K; ++k) accw[k * HID + j] += (double)g_aw[k] * (double)s.r[j];
    }
  }
  block_atomic_sum<2 * HID>(acc, sums + (size_t)g * 2 * HID);
  __syncthreads();
  block_atomic_sum<K * HID>(accw, sums + (size_t)a.G * 2 * HID);
}

// Pass 2.  gbr [K][N][Cout + 3][hw] (overwritten); dw1 [HID][K] fp64 (zeroed).  use_batch = 0: BatchNorm in eval mode.
template <int K, typename TB>
__global__ __launch_bounds__(256) void dynconv_blend_bwd_apply_kernel(BlendArgsT<TB> a, const float* __restrict__ gy,
                                                                     const float* __restrict__ gnc, const double* __restrict__ sums,
                                                                     double count, int use_batch, float* __restrict__ gbr,
                                                                     double* __restrict__ dw1) {
  const int hw = a.H * a.W, n = blockIdx.y, g = n / (a.N / a.G);
  const size_t bstride = (size_t)a.N * (a.Cout + 3) * hw;
  float m0[HID], m1[HID];
#pragma unroll
  for (int j = 0; j < HID; ++j) {
    m0[j] = use_batch ? (float)(sums[(size_t)g * 2 * HID + j] / count) : 0.f;
    m1[j] = use_batch ? (float)(sums[(size_t)g * 2 * HID + HID + j] / count) : 0.f;
  }
  double accw[HID * K];
#pragma unroll
  for (int i = 0; i < HID * K; ++i) accw[i] = 0.0;
  for (int t = 0; t < TPX; ++t) {
    const int p = (blockIdx.x * TPX + t) * 256 + threadIdx.x;
    if (p >= hw) break;
    PixelState<K> s;
    pixel_basis_curv<K, TB>(a, n, p, s);
    pixel_weights<K, TB>(a, g, s);
    float g_aw[K], g_hbn[HID];
    const float gn = gnc ? gnc[(size_t)n * hw + p] : 0.f;
    pixel_backward<K, TB>(a, n, p, s, gy, gn, g_aw, g_hbn);
    float g_curv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) g_curv[k] = gn * s.wts[k];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
      const float gh = a.gamma[j] * a.rstd[g * HID + j] * (g_hbn[j] - m0[j] - s.hhat[j] * m1[j]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        g_curv[k] = fmaf(a.w1[j * K + k], gh, g_curv[k]);
        accw[j * K + k] += (double)gh * (double)s.curv[k];
      }
    }
    float* __restrict__ o = gbr + (size_t)n * (a.Cout + 3) * hw + p;
#pragma unroll 4                                   // Cout is a multiple of 8: four channels of loads in flight
    for (int c = 0; c < a.Cout; ++c) {
      const float gv = gy[((size_t)n * a.Cout + c) * hw + p];
#pragma unroll
      for (int k = 0; k < K; ++k) o[k * bstride + (size_t)c * hw] = s.wts[k] * gv;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int m = 0; m < 3; ++m) o[k * bstride + (size_t)(a.Cout + m) * hw] = g_curv[k] * s.basis[m];
  }
  block_atomic_sum<HID * K>(accw, dw1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// depth = sum_d softmax(prob_pre)_d hyp_d (module.py:373-379) and its backward  g_pre_d = p_d (hyp_d - depth) g_depth.
// prob_pre, hyp [D][hw] (hyp per pixel, or [D] if !hyp_per_pixel).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softargmin_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ hyp,
                                                             const float* __restrict__ gdepth, float* __restrict__ gpre, int D, int hw,
                                                             int hyp_per_pixel) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  float mx = -INFINITY;
#pragma unroll 8
  for (int d = 0; d < D; ++d) mx = fmaxf(mx, pre[(size_t)d * hw + p]);
  float den = 0.f, num = 0.f;
#pragma unroll 8
  for (int d = 0; d < D; ++d) {
    const float e = expf(pre[(size_t)d * hw + p] - mx);
    den += e;
    num = fmaf(e, hyp_per_pixel ? hyp[(size_t)d * hw + p] : hyp[d], num);
  }
  const float depth = num / den, gd = gdepth[p];
#pragma unroll 8
  for (int d = 0; d < D; ++d) {
    const float pd = expf(pre[(size_t)d * hw + p] - mx) / den;
    gpre[(size_t)d * hw + p] = pd * ((hyp_per_pixel ? hyp[(size_t)d * hw + p] : hyp[d]) - depth) * gd;
  }
}

// Weight layouts of cds_conv2d_f32 for a convolution weight w [Ca + Cb][Cin][k][k] given as two tensors (wa = the DynamicConv branch
// convolution, wb = its 3-channel attention convolution, or NULL): fwd [Cin][k k][CoP] for y = conv(x, w), dgrad [Ca + Cb][k k][CiP] for
// dx = conv(dy, flipped / transposed w); CoP / CiP = channel counts rounded up to 8, zero padded.  One launch instead of a dozen ATen ops.
__global__ void pack_conv2d_kernel(const float* __restrict__ wa, const float* __restrict__ wb, float* __restrict__ fwd,
                                   float* __restrict__ dgrad, int Ca, int Cb, int Cin, int kk) {
  const int Co = Ca + Cb, CoP = (Co + 7) & ~7, CiP = (Cin + 7) & ~7;
  const int nf = fwd ? Cin * kk * CoP : 0, nd = dgrad ? Co * kk * CiP : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += gridDim.x * blockDim.x) {
    if (i < nf) {
      const int co = i % CoP, tap = (i / CoP) % kk, ci = i / (CoP * kk);
      float v = 0.f;
      if (co < Ca) v = wa[((size_t)co * Cin + ci) * kk + tap];
      else if (co < Co) v = wb[((size_t)(co - Ca) * Cin + ci) * kk + tap];
      fwd[i] = v;
    } else {
      const int jd = i - nf;
      const int ci = jd % CiP, tap = (jd / CiP) % kk, co = jd / (CiP * kk);
      float v = 0.f;
      if (ci < Cin) v = co < Ca ? wa[((size_t)co * Cin + ci) * kk + (kk - 1 - tap)] : wb[((size_t)(co - Ca) * Cin + ci) * kk + (kk - 1 - tap)];
      dgrad[jd] = v;
    }
  }
}

// The two weight layouts of a 3x3x3 CostRegNet layer (forward kernel, data-gradient kernel) in one launch.  a = outer, b = inner
// channel count of w [a][b][27]; mode 0: Conv3d stride 1 (w [Co][Cin]: fwd [Cin][27][Co], dgrad [Co][27][Cin] with flipped taps),
// mode 1: Conv3d stride 2 (dgrad = the transposed convolution's layout, taps as they are), mode 2: ConvTranspose3d (w [Cin][Cout]:
// fwd [Cin][27][Cout], dgrad [Cout][27][Cin] = the stride-2 convolution it is the transpose of).
__global__ void pack_conv3d_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgrad, int A, int B, int mode) {
  const int n = A * B * 27;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int t = i % 27, b = (i / 27) % B, a = i / (27 * B);
    const float v = w[i];
    if (mode == 2) {                                    // a = ci, b = co
      if (fwd) fwd[((size_t)a * 27 + t) * B + b] = v;
      if (dgrad) dgrad[((size_t)b * 27 + t) * A + a] = v;
    } else {                                            // a = co, b = ci
      if (fwd) fwd[((size_t)b * 27 + t) * A + a] = v;
      if (dgrad) dgrad[((size_t)a * 27 + (mode == 0 ? 26 - t : t)) * B + b] = v;
    }
  }
}

template <int K, typename TB>
int blend_dispatch(int what, const BlendArgsT<TB>& a, float* out, float* nc, const float* gy, const float* gnc, double* sums, double count,
                   int use_batch, float* gbr, double* dw1, hipStream_t st) {
  const dim3 grid(cds_ceil_div(a.H * a.W, 256 * TPX), a.N), block(256);
  switch (what) {
    case 0: hipLaunchKernelGGL((dynconv_moments_kernel<K, TB>), grid, block, 0, st, a, sums); break;
    case 1: hipLaunchKernelGGL((dynconv_blend_train_kernel<K, TB>), grid, block, 0, st, a, out, nc); break;
    case 2: hipLaunchKernelGGL((dynconv_blend_bwd_reduce_kernel<K, TB>), grid, block, 0, st, a, gy, gnc, sums); break;
    default: hipLaunchKernelGGL((dynconv_blend_bwd_apply_kernel<K, TB>), grid, block, 0, st, a, gy, gnc, sums, count, use_batch, gbr, dw1);
  }
  return cds_launch_status();
}

template <typename TB>
int blend_dispatch_k(int K, int what, const BlendArgsT<TB>& a, float* out, float* nc, const float* gy, const float* gnc, double* sums,
                     double count, int use_batch, float* gbr, double* dw1, hipStream_t st) {
  if (K == 2) return blend_dispatch<2, TB>(what, a, out, nc, gy, gnc, sums, count, use_batch, gbr, dw1, st);
  if (K == 3) return blend_dispatch<3, TB>(what, a, out, nc, gy, gnc, sums, count, use_batch, gbr, dw1, st);
  return CDS_EINVAL;
}

}  // namespace

// dw [Co][Cin][k][k] is ACCUMULATED onto (zero it first).  g [N][Co][Ho][Wo], x [N][Cin][H][W]; k in {1,3,5,7,11} with stride 1, or
// k = 3 with stride 2.
template <typename TX>
static int wgrad2d_entry(const float* g, const TX* x, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H, int W, int k, int stride,
                         int pad, void* stream) {
  if (!g || !x || !dw || N < 1 || Co < 1 || Cin < 1 || Ho < 1 || Wo < 1 || H < 1 || W < 1 || pad < 0) return CDS_EINVAL;
  if (Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    switch (k) {
      case 1: return launch_wgrad2d<1, 1, 16>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
      case 3: return Cin >= 12 ? launch_wgrad2d<3, 1, 16>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st)
                               : launch_wgrad2d<3, 1, 8>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
      case 5: return launch_wgrad2d<5, 1, 8>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
      case 7: return launch_wgrad2d<7, 1, 4>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
      case 11: return launch_wgrad2d<11, 1, 1>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
      default: return CDS_EINVAL;
    }
  }
  if (stride == 2 && k == 3) return launch_wgrad2d<3, 2, 8>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, pad, st);
  return CDS_EINVAL;
}

extern "C" int cds_conv2d_wgrad_f32(const float* g, const float* x, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H, int W,
                                    int k, int stride, int pad, void* stream) {
  return wgrad2d_entry<float>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, k, stride, pad, stream);
}

// The same with the layer input x stored as raw bfloat16 (widened on load, fp32 matrix pipe, fp32 accumulation).
extern "C" int cds_conv2d_wgrad_xb16_f32(const float* g, const unsigned short* x, float* dw, int N, int Co, int Cin, int Ho, int Wo, int H,
                                         int W, int k, int stride, int pad, void* stream) {
  return wgrad2d_entry<b16>(g, x, dw, N, Co, Cin, Ho, Wo, H, W, k, stride, pad, stream);
}

// gx [N][Cin][H][W] (overwritten) for conv 3x3 stride 2 pad 1 with weight w [Co][Cin][3][3]; g [N][Co][Ho][Wo], Cin % 8 == 0.
extern "C" int cds_conv2d_dgrad_s2_f32(const float* g, const float* w, float* gx, int N, int Co, int Cin, int Ho, int Wo, int H,
                                       int W, void* stream) {
  if (!g || !w || !gx || N < 1 || Co < 1 || Cin < 1 || H < 1 || W < 1) return CDS_EINVAL;
  if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (Cin % 8 == 0) {
    const dim3 grid(cds_ceil_div(W, 64), cds_ceil_div(H, 4), N * (Cin / 8));
    hipLaunchKernelGGL(conv2d_dgrad_s2_kernel<8>, grid, dim3(256), 0, st, g, w, gx, Co, Cin, Ho, Wo, H, W);
  } else {
    const dim3 grid(cds_ceil_div(W, 64), cds_ceil_div(H, 4), N * Cin);
    hipLaunchKernelGGL(conv2d_dgrad_s2_kernel<1>, grid, dim3(256), 0, st, g, w, gx, Co, Cin, Ho, Wo, H, W);
  }
  return cds_launch_status();
}

// InstanceNorm2d(eps 1e-5, no affine) + activation backward.  gz, y, gy [N][C][H][W]; stats [N][C][2] fp64 (sum, sum of squares of y:
// what cds_instnorm_act_f32 leaves); sums [N][C][2] fp64 scratch (zeroed here unless scratch_zeroed says the caller did).  act: CDS_ACT_LEAKY01 | CDS_ACT_TANH | CDS_ACT_NONE.
template <typename TY>
static int instnorm_bwd_entry(const float* gz, const TY* y, const b16* zs, const double* stats, double* sums, float* gy, int N, int C, int H,
                              int W, int act, int scratch_zeroed, void* stream) {
  if (!gz || !y || !stats || !sums || !gy || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hw = H * W;
  if (!scratch_zeroed && hipMemsetAsync(sums, 0, sizeof(double) * 2 * N * C, st) != hipSuccess) return cds_launch_status();
  int bpc = cds_ceil_div(hw, 256 * 16);
  if (bpc > 64) bpc = 64;
  hipLaunchKernelGGL(instnorm_bwd_reduce_kernel<TY>, dim3(N * C * bpc), dim3(256), 0, st, gz, y, stats, sums, hw, act, bpc, zs);
  int gx = cds_ceil_div(hw, 256 * 4);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(instnorm_bwd_apply_kernel<TY>, dim3(gx, N * C), dim3(256), 0, st, gz, y, stats, sums, gy, hw, act, zs);
  return cds_launch_status();
}

extern "C" int cds_instnorm_bwd_f32(const float* gz, const float* y, const double* stats, double* sums, float* gy, int N, int C, int H,
                                    int W, int act, int scratch_zeroed, void* stream) {
  return instnorm_bwd_entry<float>(gz, y, nullptr, stats, sums, gy, N, C, H, W, act, scratch_zeroed, stream);
}

// The same with the pre-normalisation map y stored as raw bfloat16 (what cds_instnorm_act_b16_f32 left).  z16 (optional, LeakyReLU
// only): the stored OUTPUT of the forward; its sign bit is the LeakyReLU decision the forward took (xhat rebuilt from the rounded y can
// fall on the other side of zero: a gradient factor of 0.1 instead of 1 on ~0.3 % of the elements).
extern "C" int cds_instnorm_bwd_yb16_f32(const float* gz, const unsigned short* y, const unsigned short* z16, const double* stats,
                                         double* sums, float* gy, int N, int C, int H, int W, int act, int scratch_zeroed, void* stream) {
  if (z16 && act != CDS_ACT_LEAKY01) return CDS_EINVAL;
  return instnorm_bwd_entry<b16>(gz, y, z16, stats, sums, gy, N, C, H, W, act, scratch_zeroed, stream);
}

// ---- DynamicConv epilogue, training mode --------------------------------------------------------------------------------------
// Statistics of the attention MLP's BatchNorm2d for G groups of N / G images.  branches [K][N][Cout+3][H][W], epipoles DEVICE [N][2],
// w1 [4][K].  mom: fp64 scratch [G][K + K(K+1)/2] (overwritten).  Leaves mean / rstd [G][4] (batch statistics if use_batch, else the
// running ones) and, if use_batch and running_mean != NULL, updates the running statistics group after group (momentum as given).
template <typename TB>
static int bn_stats_entry(const TB* branches, const float* epipoles, const float* w1, double* mom, float* mean, float* rstd,
                          float* running_mean, float* running_var, int N, int G, int K, int Cout, int H, int W, float eps, float momentum,
                          int use_batch, int scratch_zeroed, void* stream) {
  if (!branches || !epipoles || !w1 || !mom || !mean || !rstd || N < 1 || G < 1 || (N % G) || Cout < 1 || H < 1 || W < 1 || (K != 2 && K != 3))
    return CDS_EINVAL;
  if (!use_batch && (!running_mean || !running_var)) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  BlendArgsT<TB> a{branches, epipoles, w1, nullptr, nullptr, nullptr, nullptr, nullptr, N, G, Cout, H, W, 1.f};
  const int NM = K + K * (K + 1) / 2;
  if (use_batch) {
    if (!scratch_zeroed && hipMemsetAsync(mom, 0, sizeof(double) * G * NM, st) != hipSuccess) return cds_launch_status();
    if (int e = blend_dispatch_k(K, 0, a, nullptr, nullptr, nullptr, nullptr, mom, 0.0, 1, nullptr, nullptr, st)) return e;
  }
  const double count = (double)(N / G) * H * W;
  hipLaunchKernelGGL(dynconv_bn_fold_kernel, dim3(1), dim3(64), 0, st, mom, w1, K, G, count, (double)eps, momentum, use_batch,
                     running_mean, running_var, mean, rstd);
  return cds_launch_status();
}

extern "C" int cds_dynconv_bn_stats_f32(const float* branches, const float* epipoles, const float* w1, double* mom, float* mean,
                                        float* rstd, float* running_mean, float* running_var, int N, int G, int K, int Cout, int H,
                                        int W, float eps, float momentum, int use_batch, int scratch_zeroed, void* stream) {
  return bn_stats_entry<float>(branches, epipoles, w1, mom, mean, rstd, running_mean, running_var, N, G, K, Cout, H, W, eps, momentum,
                               use_batch, scratch_zeroed, stream);
}

// The three DynamicConv epilogue entries with the branch tensor stored as raw bfloat16 (cds_f32_to_bf16 of the convolution responses).
extern "C" int cds_dynconv_bn_stats_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, double* mom,
                                            float* mean, float* rstd, float* running_mean, float* running_var, int N, int G, int K,
                                            int Cout, int H, int W, float eps, float momentum, int use_batch, int scratch_zeroed,
                                            void* stream) {
  return bn_stats_entry<b16>(branches, epipoles, w1, mom, mean, rstd, running_mean, running_var, N, G, K, Cout, H, W, eps, momentum,
                             use_batch, scratch_zeroed, stream);
}

// out [N][Cout][H][W], norm_curv [N][H][W] from the branches and the statistics of cds_dynconv_bn_stats_f32.
template <typename TB>
static int blend_train_entry(const TB* branches, const float* epipoles, const float* w1, const float* w2, const float* gamma,
                             const float* beta, const float* mean, const float* rstd, float temperature, float* out, float* norm_curv,
                             int N, int G, int K, int Cout, int H, int W, void* stream) {
  if (!branches || !epipoles || !w1 || !w2 || !gamma || !beta || !mean || !rstd || !out || !norm_curv || N < 1 || G < 1 || (N % G) ||
      Cout < 1 || H < 1 || W < 1 || !(temperature > 0.f))
    return CDS_EINVAL;
  BlendArgsT<TB> a{branches, epipoles, w1, w2, gamma, beta, mean, rstd, N, G, Cout, H, W, 1.0f / temperature};
  return blend_dispatch_k(K, 1, a, out, norm_curv, nullptr, nullptr, nullptr, 0.0, 1, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int cds_dynconv_blend_train_f32(const float* branches, const float* epipoles, const float* w1, const float* w2,
                                           const float* gamma, const float* beta, const float* mean, const float* rstd,
                                           float temperature, float* out, float* norm_curv, int N, int G, int K, int Cout, int H, int W,
                                           void* stream) {
  return blend_train_entry<float>(branches, epipoles, w1, w2, gamma, beta, mean, rstd, temperature, out, norm_curv, N, G, K, Cout, H, W,
                                  stream);
}

extern "C" int cds_dynconv_blend_train_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, const float* w2,
                                               const float* gamma, const float* beta, const float* mean, const float* rstd,
                                               float temperature, float* out, float* norm_curv, int N, int G, int K, int Cout, int H,
                                               int W, void* stream) {
  return blend_train_entry<b16>(branches, epipoles, w1, w2, gamma, beta, mean, rstd, temperature, out, norm_curv, N, G, K, Cout, H, W,
                                stream);
}

// Backward of cds_dynconv_blend_train_f32 (+ the BatchNorm, batch statistics if use_batch).  gy [N][Cout][H][W], gnc [N][H][W] or NULL.
// Leaves gbr [K][N][Cout+3][H][W] (gradient of the branch tensor) and, in fp64, sums [G][8] scratch followed by dw2 [K][4], and
// dw1 [4][K]; dgamma_j = sum_g sums[g][4 + j], dbeta_j = sum_g sums[g][j].  sums: G * 8 + K * 4 doubles, dw1: 4 K doubles.
template <typename TB>
static int blend_bwd_entry(const TB* branches, const float* epipoles, const float* w1, const float* w2, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, float temperature, const float* gy, const float* gnc,
                           float* gbr, double* sums, double* dw1, int N, int G, int K, int Cout, int H, int W, int use_batch,
                           int scratch_zeroed, void* stream) {
  if (!branches || !epipoles || !w1 || !w2 || !gamma || !beta || !mean || !rstd || !gy || !gbr || !sums || !dw1 || N < 1 || G < 1 ||
      (N % G) || Cout < 1 || H < 1 || W < 1 || !(temperature > 0.f))
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  BlendArgsT<TB> a{branches, epipoles, w1, w2, gamma, beta, mean, rstd, N, G, Cout, H, W, 1.0f / temperature};
  if (!scratch_zeroed) {
    if (hipMemsetAsync(sums, 0, sizeof(double) * (G * 2 * HID + K * HID), st) != hipSuccess) return cds_launch_status();
    if (hipMemsetAsync(dw1, 0, sizeof(double) * HID * K, st) != hipSuccess) return cds_launch_status();
  }
  const double count = (double)(N / G) * H * W;
  if (int e = blend_dispatch_k(K, 2, a, nullptr, nullptr, gy, gnc, sums, count, use_batch, nullptr, nullptr, st)) return e;
  return blend_dispatch_k(K, 3, a, nullptr, nullptr, gy, gnc, sums, count, use_batch, gbr, dw1, st);
}

extern "C" int cds_dynconv_blend_bwd_f32(const float* branches, const float* epipoles, const float* w1, const float* w2,
                                         const float* gamma, const float* beta, const float* mean, const float* rstd,
                                         float temperature, const float* gy, const float* gnc, float* gbr, double* sums, double* dw1,
                                         int N, int G, int K, int Cout, int H, int W, int use_batch, int scratch_zeroed, void* stream) {
  return blend_bwd_entry<float>(branches, epipoles, w1, w2, gamma, beta, mean, rstd, temperature, gy, gnc, gbr, sums, dw1, N, G, K, Cout,
                                H, W, use_batch, scratch_zeroed, stream);
}

extern "C" int cds_dynconv_blend_bwd_b16_f32(const unsigned short* branches, const float* epipoles, const float* w1, const float* w2,
                                             const float* gamma, const float* beta, const float* mean, const float* rstd,
                                             float temperature, const float* gy, const float* gnc, float* gbr, double* sums,
                                             double* dw1, int N, int G, int K, int Cout, int H, int W, int use_batch,
                                             int scratch_zeroed, void* stream) {
  return blend_bwd_entry<b16>(branches, epipoles, w1, w2, gamma, beta, mean, rstd, temperature, gy, gnc, gbr, sums, dw1, N, G, K, Cout, H,
                              W, use_batch, scratch_zeroed, stream);
}

namespace {
__global__ void dynconv_bwd_finish_kernel(const double* __restrict__ sums, const double* __restrict__ dw1, int G, int K,
                                          float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < 2 * HID) {                                        // dbeta_j (i < 4), dgamma_j: the groups in a fixed order
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += sums[g * 2 * HID + i];
    out[i] = (float)s;
  } else if (i < 2 * HID + K * HID) {
    out[i] = (float)sums[G * 2 * HID + (i - 2 * HID)];      // dw2 [K][4]
  } else if (i < 2 * HID + 2 * K * HID) {
    out[i] = (float)dw1[i - 2 * HID - K * HID];             // dw1 [4][K]
  }
}
}  // namespace

// The small gradients of a DynamicConv's attention MLP from the fp64 accumulators of cds_dynconv_blend_bwd_f32 in one launch:
// out = [dbeta (4) | dgamma (4) | dw2 [K][4] | dw1 [4][K]] floats.
extern "C" int cds_dynconv_bwd_finish_f32(const double* sums, const double* dw1, int G, int K, float* out, void* stream) {
  if (!sums || !dw1 || !out || G < 1 || K < 1 || K > 3) return CDS_EINVAL;
  hipLaunchKernelGGL(dynconv_bwd_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, dw1, G, K, out);
  return cds_launch_status();
}

// gpre [D][h][w] = d (sum_d softmax(prob_pre)_d hyp_d) / d prob_pre * gdepth [h][w]
extern "C" int cds_softargmin_bwd_f32(const float* prob_pre, const float* hyp, const float* gdepth, float* gpre, int D, int h, int w,
                                      int hyp_per_pixel, void* stream) {
  if (!prob_pre || !hyp || !gdepth || !gpre || D < 1 || h < 1 || w < 1) return CDS_EINVAL;
  hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(cds_ceil_div(h * w, 256)), dim3(256), 0, (hipStream_t)stream, prob_pre, hyp, gdepth,
                     gpre, D, h * w, hyp_per_pixel);
  return cds_launch_status();
}

// fwd [Cin][k k][CoP] and / or dgrad [Ca + Cb][k k][CiP] (either may be NULL) from wa [Ca][Cin][k][k] and wb [Cb][Cin][k][k] (NULL if Cb = 0).
extern "C" int cds_pack_conv2d_f32(const float* wa, const float* wb, float* fwd, float* dgrad, int Ca, int Cb, int Cin, int k,
                                   void* stream) {
  if (!wa || (Cb > 0 && !wb) || (!fwd && !dgrad) || Ca < 1 || Cb < 0 || Cin < 1 || k < 1) return CDS_EINVAL;
  const int Co = Ca + Cb, kk = k * k;
  const int n = (fwd ? Cin * kk * ((Co + 7) & ~7) : 0) + (dgrad ? Co * kk * ((Cin + 7) & ~7) : 0);
  hipLaunchKernelGGL(pack_conv2d_kernel, dim3(cds_ceil_div(n, 256) > 64 ? 64 : cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     wa, wb, fwd, dgrad, Ca, Cb, Cin, kk);
  return cds_launch_status();
}

// w [A][B][27] -> fwd / dgrad (either may be NULL), see pack_conv3d_kernel.
extern "C" int cds_pack_conv3d_f32(const float* w, float* fwd, float* dgrad, int A, int B, int mode, void* stream) {
  if (!w || (!fwd && !dgrad) || A < 1 || B < 1 || mode < 0 || mode > 2) return CDS_EINVAL;
  const int n = A * B * 27;
  hipLaunchKernelGGL(pack_conv3d_kernel, dim3(cds_ceil_div(n, 256) > 128 ? 128 : cds_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     fwd, dgrad, A, B, mode);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// bf16 storage of the 2D activations: the forward-side kernels of the policy (the backward-side ones are the *_b16 / *_xb16 / *_yb16
// entries above).  Rounding happens ONCE, where a tensor is stored.  Two forms: the default rounds only what is KEPT for the backward
// (the forward pass continues on the unrounded fp32 transients: loss and depth maps are those of the fp32 step); `strict` also feeds the
// forward with the stored values (statistics of y16, out32 = widened out16), so both passes see one and the same activation.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, b16* __restrict__ dst, size_t n4, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    const unsigned lo = (unsigned)f2b(v.x) | ((unsigned)f2b(v.y) << 16), hi = (unsigned)f2b(v.z) | ((unsigned)f2b(v.w) << 16);
    reinterpret_cast<uint2*>(dst)[i] = make_uint2(lo, hi);
  }
  if (blockIdx.x == 0)
    for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) dst[i] = f2b(src[i]);
}

// Pass 1 of InstanceNorm on a bf16-stored map: y16 = bf16(y) is written, the fp64 (sum, sum of squares) are those of y (strict: of y16).
__global__ __launch_bounds__(256) void instnorm_stats_store_b16_kernel(const float* __restrict__ y, b16* __restrict__ y16,
                                                                       double* __restrict__ stats, int hw, int blocks_per_c, int strict) {
  const int c = blockIdx.x / blocks_per_c, b = blockIdx.x % blocks_per_c;
  const float* __restrict__ yc = y + (size_t)c * hw;
  b16* __restrict__ oc = y16 + (size_t)c * hw;
  double s = 0.0, q = 0.0;
#pragma unroll 4
  for (int i = b * 256 + threadIdx.x; i < hw; i += blocks_per_c * 256) {
    const float yv = yc[i];
    const b16 h = f2b(yv);
    oc[i] = h;
    const double v = (double)(strict ? __uint_as_float((unsigned)h << 16) : yv);
    s += v;
    q += v * v;
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  __shared__ double red[2][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[2 * c + 0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&stats[2 * c + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// Pass 2: z = act(y * alpha + beta) (the expression of instnorm_apply_kernel; strict: of y16).  out16 != NULL: z is stored as bf16 and
// out32 (the transient fp32 copy the next layer's forward kernel reads) receives z, or under strict the widened stored value;
// out16 == NULL: out32 = z in fp32 (the tanh stage outputs, which leave the 2D stack for the cost volume).
__global__ __launch_bounds__(256) void instnorm_apply_b16_kernel(const float* __restrict__ y, const b16* __restrict__ y16,
                                                                 const double* __restrict__ stats, float* __restrict__ out32,
                                                                 b16* __restrict__ out16, int hw, int act, int strict) {
  const int c = blockIdx.y;
  const double mean = stats[2 * c] / hw;
  double var = stats[2 * c + 1] / hw - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float al = (float)(1.0 / sqrt(var + 1e-5));
  const float be = -(float)mean * al;
  const size_t o = (size_t)c * hw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
    float z = cds_apply_act((strict ? ldf(y16, o + i) : y[o + i]) * al + be, act);
    if (out16) {
      const b16 h = f2b(z);
      out16[o + i] = h;
      if (strict) z = __uint_as_float((unsigned)h << 16);
    }
    out32[o + i] = z;
  }
}

}  // namespace

// dst[i] = bfloat16(src[i]) (round to nearest even), n elements; src 16-byte aligned.
extern "C" int cds_f32_to_bf16(const float* src, unsigned short* dst, long long n, void* stream) {
  if (!src || !dst || n < 1) return CDS_EINVAL;
  const size_t n4 = (size_t)n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n4, (size_t)n);
  return cds_launch_status();
}

// InstanceNorm2d (eps 1e-5, no affine) + activation with bf16 storage: y [N][C][H][W] fp32 (the transient convolution output) ->
// y16 (its stored form, what cds_instnorm_bwd_yb16_f32 reads), stats [N][C][2] fp64, out32 / out16 as described at
// instnorm_apply_b16_kernel (out16 may be NULL); strict as described above.
extern "C" int cds_instnorm_act_b16_f32(const float* y, unsigned short* y16, float* out32, unsigned short* out16, double* stats, int N,
                                        int C, int H, int W, int act, int strict, void* stream) {
  if (!y || !y16 || !out32 || !stats || N < 1 || C < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int hw = H * W;
  if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * N * C, st) != hipSuccess) return cds_launch_status();
  int bpc = cds_ceil_div(hw, 256 * 16);
  if (bpc < 1) bpc = 1;
  hipLaunchKernelGGL(instnorm_stats_store_b16_kernel, dim3(N * C * bpc), dim3(256), 0, st, y, y16, stats, hw, bpc, strict);
  int gx = cds_ceil_div(hw, 256 * 4);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(instnorm_apply_b16_kernel, dim3(gx, N * C), dim3(256), 0, st, y, y16, stats, out32, out16, hw, act, strict);
  return cds_launch_status();
}
