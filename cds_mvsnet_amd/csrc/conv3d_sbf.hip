// K4 on the bf16 matrix cores with fp32-equivalent arithmetic ("split-bf16", 3xBF16 error-compensated products).
//
// Every fp32 operand is split exactly into three bf16 terms, a = a1 + a2 + a3 (a1 = RN_bf16(a), a2 = RN_bf16(a - a1),
// a3 = a - a1 - a2: 3 x 8 significand bits = the 24 bits of fp32, the residuals are exact), and a product a*b is
// evaluated as the six partial products of order <= 2^-16,
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// each exact in the MFMA (bf16 x bf16 fits fp32) and accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The dropped
// terms (a2 b3, a3 b2, a3 b3) are <= 2^-23 |a b|: the size of the ONE rounding an fp32 multiply makes, so the result
// carries fp32-class error (tests/test_hip_parity.py::test_conv3d_split_bf16_*: measured against float64, inside 1.5x
// the error of PyTorch's own fp32 convolution) while the matrix pipe runs the bf16 instruction at 16x the fp32-MFMA
// rate: 6 bf16 MFMAs (K = 32 each) replace 16 fp32 ones (K = 4 each), and ONE ds_read_b128 feeds a 16-voxel x 8-channel
// operand instead of one ds_read_b32 per MFMA.
//
// Activations are channels-last fp32 in HBM, [D][H][W][C] (a voxel's C channels are contiguous: 32 B at C = 8), which is
// what the fused warp-aggregate kernel stores (one 32-byte vector store per voxel) and what makes the staging loads and
// the epilogue stores of every layer full 16-byte lanes on contiguous runs.
//
//   transposed implicit GEMM:  D[i = cout][j = voxel] += A[i][k] * B[k][j],  K-step = 32 = 4 taps x 8 channels
//   LDS tile: [position][term 0..2][8 channels] bf16 (48 B per position: 16 consecutive positions hit all 64 banks once),
//             staged 8 input channels per round from the fp32 volume with the split done in registers
//   B (data): lane l -> voxel (l & 15) of a 16-voxel x-run, tap (l >> 4) of the K-step: one ds_read_b128 per term
//   A (weights): host-split [round][kstep][mblock][term][lane][8]: one coalesced 16-byte load per lane, term, K-step
//   C/D: lane l holds voxel (l & 15), couts 16 mb + 4 (l >> 4) + 0..3 -> one 16-byte channels-last store per N-tile
#include <stdlib.h>

#include "cds_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

union BV {
  uint4 u;
  bf16x8 v;
};

// exact three-way split of two floats: packed (hi0,hi1), (mid0,mid1), (lo0,lo1)
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

// split the 8 channels of one position (two float4) and store them as [term][8] bf16 (48 B)
__device__ __forceinline__ void split_store8(unsigned char* dst, const float4& a, const float4& b) {
  uint32_t h[4], m[4], l[4];
  split2(a.x, a.y, h[0], m[0], l[0]);
  split2(a.z, a.w, h[1], m[1], l[1]);
  split2(b.x, b.y, h[2], m[2], l[2]);
  split2(b.z, b.w, h[3], m[3], l[3]);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  d4[0] = make_uint4(h[0], h[1], h[2], h[3]);
  d4[1] = make_uint4(m[0], m[1], m[2], m[3]);
  d4[2] = make_uint4(l[0], l[1], l[2], l[3]);
}

#define SBF_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16((a).v, (b).v, acc, 0, 0, 0)

// The six partial products of one K-step for NQ independent accumulators, smallest terms first; W[term] = weights
// (hi, mid, lo), X[q][term] = data of N-tile q.
#define SBF_TERMS(ACC, T0, NQ, W, X)                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[2], (X)[q_][0]); /* lo x hi  */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[1], (X)[q_][1]); /* mid x mid */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[0], (X)[q_][2]); /* hi x lo  */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[1], (X)[q_][0]); /* mid x hi */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[0], (X)[q_][1]); /* hi x mid */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SBF_MFMA(ACC[(T0) + q_], (W)[0], (X)[q_][0]); /* hi x hi  */ \
  __builtin_amdgcn_sched_barrier(0);

constexpr int POSB = 48;   // bytes per LDS position

// A/B build knobs (scripts/build_variant.sh): CDS_SBF_PRIO = s_setprio level of the consumer (MFMA) waves, the producers stay at 0;
// CDS_SBF_NTSTORE = nontemporal epilogue stores (the activations are far larger than L2 + MALL and are read back a layer later).
#ifdef CDS_SBF_PRIO
#define SBF_CONSUMER_PRIO() __builtin_amdgcn_s_setprio(CDS_SBF_PRIO)
#else
#define SBF_CONSUMER_PRIO()
#endif
__device__ __forceinline__ void sbf_store4(float* p, const float4& o) {
#ifdef CDS_SBF_NTSTORE
  __builtin_nontemporal_store((f32x4){o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<float4*>(p) = o;
#endif
}

// Tile order: z fastest, then x, then y.  A workgroup walks consecutive tiles, i.e. a column of z-adjacent tiles: the halo
// planes it shares with the tile before are still in its XCD's L2.  (Measured, FETCH_SIZE: with x fastest the halo re-reads
// of conv0 went to the fabric almost entirely, 4.5 GB fetched for a 2.0 GB input, and the layer ran at the fabric's mixed
// read/write rate instead of the matrix pipe's.)
#ifdef CDS_SBF_XFAST
#define SBF_TILE(tile, A, B, C) const int A = (tile) % tiles_x, B = ((tile) / tiles_x) % tiles_y, C = (tile) / (tiles_x * tiles_y)
#else
#define SBF_TILE(tile, A, B, C) \
  const int tiles_z_ = ntiles / (tiles_x * tiles_y); \
  const int C = (tile) % tiles_z_, A = ((tile) / tiles_z_) % tiles_x, B = (tile) / (tiles_z_ * tiles_x)
#endif

// ---------------------------------------------------------------------------------------------
// forward convolution, stride S in {1, 2}, pad 1.  MB: 16-cout blocks; output tile TX x 4 x TZ (wave = y row).
// Stride 2 de-interleaves the x parities of the staged tile (position = row * IXP + parity * IXH + (col >> 1)), so that the
// 16 voxels of an N-tile read 16 consecutive positions for every tap.
// ---------------------------------------------------------------------------------------------
// PAIR (stride 1, Cout == 8): an MFMA column is a PAIR of x-adjacent voxels and its 16 rows are (x parity, cout): both
// halves of the matrix tile carry real outputs (with rows = 16 couts, half of every MFMA would multiply zero padding).  The
// K window is 3 x 3 x 4 taps (x' = 0..3 relative to the pair; weight row (p, co) is w[x' - p] or 0): 9 K-steps per 32 voxels
// instead of 14, and as many fewer LDS operand reads.  The x parities of the staged tile are de-interleaved as for stride 2.
template <int S, int MB, int TX_, int TZ_, bool PAIR = false, bool WRES_ = false>
struct FCfg {
  static constexpr int TX = TX_, TY = 4, TZ = TZ_;
  static constexpr bool DEINT = S == 2 || PAIR;
  static constexpr int XT = TX / (PAIR ? 32 : 16), NT = XT * TZ;   // N-tiles per y row of the tile
  static constexpr int IX = (TX - 1) * S + 3, IY = (TY - 1) * S + 3, IZ = (TZ - 1) * S + 3;
  static constexpr int IXH = DEINT ? (IX + 1) / 2 : 0;              // positions per parity half-row
  static constexpr int IXP = DEINT ? 2 * IXH : (IX + 7) / 8 * 8;    // positions per row
  static constexpr int NPOS = IZ * IY * IXP;
  static constexpr int LDSB = NPOS * POSB;
  static constexpr int KW = PAIR ? 4 : 3;                            // taps along x
  static constexpr int KSTEPS = (9 * KW + 3) / 4;                    // 27 taps + 1 zero tap (7) | 36 taps (9), 4 per K-step
  // consumer waves: two per SIMD when a y row has >= 2 N-tiles to split between them (one waits for LDS, the other issues)
  // (MB = 4: the accumulators need the 256-register budget; stride 2 stages 8 input voxels per output: the extra waves go
  // to the producers instead)
  // (PAIR: the 27 weight vectors of the layer stay in registers (WRES) -> one consumer wave per SIMD with the 256-register budget)
  static constexpr bool WRES = WRES_;                                // Cin == 8: weights do not change between stages
#ifndef CDS_SBF_V
#define CDS_SBF_V 1
#endif
  static constexpr int CW = (WRES || MB == 2 || (CDS_SBF_V == 2 && MB == 1 && S == 1)) ? 4 : ((NT >= 2 && MB < 4 && S == 1) ? 8 : 4);
  static constexpr int PW = (S == 2 && MB < 2) ? 8 : 4;           // producer waves (MB >= 2 keeps the 256-register budget)
  static constexpr int NTW = NT / (CW / 4);                          // N-tiles per consumer wave
  static constexpr int NG = (MB == 1 && (CW == 4 || CDS_SBF_V != 1)) ? (NTW < 4 ? NTW : 4) : (NTW < 2 ? NTW : 2);   // N-tiles whose operands are in registers together
  static constexpr bool WDB = MB < 4;                                // weights double-buffered across K-steps (register budget)
  static constexpr int THREADS = (CW + PW) * 64;
#ifndef CDS_SBF_NSETS
#define CDS_SBF_NSETS 2
#endif
  // register sets of the producers = stages between a stage's loads and its split (even: buffer parity == set parity)
  static constexpr int NSETS = CDS_SBF_NSETS;
};

// Warp-specialised, persistent over TPW consecutive tiles (and the Cin / 8 channel rounds of each): 512 threads = 4 consumer
// waves (one per SIMD; wave = output row y: LDS reads + MFMAs + epilogue stores, nothing else) and 4 producer waves (global
// loads -> exact bf16 split in registers -> LDS writes), double-buffered LDS tile, ONE workgroup barrier per stage.  The
// producers run a stage ahead in LDS and another one ahead in registers, so the matrix pipe never waits for staging: with
// the staging in the same waves as the MFMAs, the two workgroups of a CU fell into lock-step and the pipe idled half the time.
template <int S, int MB, int TX_, int TZ_, bool PAIR = false, bool WRES_ = false>
__global__ __launch_bounds__((FCfg<S, MB, TX_, TZ_, PAIR, WRES_>::THREADS)) void conv3d_sbf_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                            const float* __restrict__ bias, const float* __restrict__ skip,
                                                            float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                            int Do, int Ho, int Wo, int act, int tiles_x, int tiles_y,
                                                            int ntiles, int tpw) {
  using Cfg = FCfg<S, MB, TX_, TZ_, PAIR, WRES_>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  const int wg = cds_xcd_remap(blockIdx.x, nwg);
  const int tile0 = wg * tpw, tile1 = min(ntiles, tile0 + tpw);
  const int rounds = Cin >> 3;
  const int nstages = (tile1 - tile0) * rounds;
  if (nstages <= 0) return;

  if (wave >= Cfg::CW) {
    // ============================== producers ==============================
    const int ptid = tid - Cfg::CW * 64;
    constexpr int NP = Cfg::IZ * Cfg::IY * Cfg::IX, PT = Cfg::PW * 64;
    constexpr int PPT = (NP + PT - 1) / PT;
    int s_rel[PPT];   // packed (rz << 20) | (ry << 10) | c, -1 = no position
    int s_dst[PPT];   // LDS byte offset inside a buffer
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * PT + ptid;
      const int row = p / Cfg::IX, q = p - row * Cfg::IX;
      // de-interleaved tiles: consecutive threads fill consecutive LDS positions of ONE parity half-row (conflict-free
      // 16-byte stores); their global columns are then 2 apart
      const int c = Cfg::DEINT ? (2 * (q % Cfg::IXH) + q / Cfg::IXH) : q;
      const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
      s_rel[h] = (p < NP && c < Cfg::IX) ? ((rz << 20) | (ry << 10) | c) : -1;
      s_dst[h] = (row * Cfg::IXP + q) * POSB;
    }
    // Two register sets: the loads of stage st + 2 are in flight while stage st + 1 is split and written, so a stage's loads
    // have two stage times to arrive (the stride-2 and deep layers have stages of a few microseconds, about one HBM latency).
    float4 va[Cfg::NSETS][PPT], vb[Cfg::NSETS][PPT];
    auto issue = [&](int st, int set) {
      const int tile = tile0 + st / rounds, rd = st % rounds;
      SBF_TILE(tile, tx_i, ty_i, tz_i);
      const int gx0 = tx_i * Cfg::TX * S - 1, gy0 = ty_i * Cfg::TY * S - 1, gz0 = tz_i * Cfg::TZ * S - 1;
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        const int gz = gz0 + (s_rel[h] >> 20), gy = gy0 + ((s_rel[h] >> 10) & 1023), gx = gx0 + (s_rel[h] & 1023);
        const bool ok = s_rel[h] >= 0 && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const float* __restrict__ src = x + ((size_t)((size_t)gz * H + gy) * W + gx) * Cin + rd * 8;
        va[set][h] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        vb[set][h] = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto deposit = [&](int buf, int set) {
      unsigned char* base = lds + buf * Cfg::LDSB;
#pragma unroll
      for (int h = 0; h < PPT; ++h)
        if (s_rel[h] >= 0) split_store8(base + s_dst[h], va[set][h], vb[set][h]);
    };
    // Stage s travels in register set s % NSETS and lands in LDS buffer s & 1; its loads are issued NSETS stages before it is
    // split and written.  (NSETS = 4 measured no faster than 2 on any layer: the staging loads are not latency-bound.)
#pragma unroll
    for (int u = 0; u < Cfg::NSETS; ++u)
      if (u < nstages) issue(u, u);
    deposit(0, 0);
    if (Cfg::NSETS < nstages) issue(Cfg::NSETS, 0);
    __syncthreads();                                   // #0: buffer 0 holds stage 0
    for (int st = 0; st < nstages; st += Cfg::NSETS) {
      bool done = false;
#pragma unroll
      for (int u = 1; u <= Cfg::NSETS; ++u) {          // during stage st + u - 1: stage st + u -> buffer u & 1, set u % NSETS
        if (st + u < nstages) {
          deposit(u & 1, u % Cfg::NSETS);
          if (st + u + Cfg::NSETS < nstages) issue(st + u + Cfg::NSETS, u % Cfg::NSETS);
        }
        __syncthreads();                               // #(st + u): stage st + u - 1 consumed, stage st + u staged
        if (st + u >= nstages) { done = true; break; }
      }
      if (done) break;
    }
    return;
  }

  // ============================== consumers ==============================
  SBF_CONSUMER_PRIO();
  const int j = lane & 15, g = lane >> 4;
  // per-lane byte offset of the tap this lane group multiplies in K-step t (tap 27 = zero weights -> any in-tile data)
  int toff[Cfg::KSTEPS];
#pragma unroll
  for (int t = 0; t < Cfg::KSTEPS; ++t) {
    // PAIR: K-step t is the (kz, ky) row t, and the lane groups take x' = 0, 2, 1, 3: the two groups that share an LDS
    // service group (g = 0, 1 and g = 2, 3) then read the SAME parity plane one position apart (overlapping addresses
    // broadcast) instead of two planes whose bank windows collide (PMC: half of the LDS cycles were conflicts).
    int tap = PAIR ? 4 * t + ((g & 1) * 2 + (g >> 1)) : 4 * t + g;
    if (tap > 9 * Cfg::KW - 1) tap = 9 * Cfg::KW - 1;
    const int kz = tap / (3 * Cfg::KW), ky = (tap / Cfg::KW) % 3, kx = tap % Cfg::KW;
    const int xoff = Cfg::DEINT ? ((kx & 1) * Cfg::IXH + (kx >> 1)) : kx;
    toff[t] = ((kz * Cfg::IY + ky) * Cfg::IXP + xoff) * POSB;
  }
  const int wy = wave & 3, wh = wave >> 2;             // output row y, share of the row's N-tiles
  const int b_base = (wy * S * Cfg::IXP + j) * POSB;

  f32x4 acc[MB][Cfg::NTW];
  const uint4* __restrict__ wl = wsp + lane;
  BV wa[Cfg::WDB ? 2 : 1][MB][3];                      // weights of the current / next K-step (live across stages)
  auto load_w_from = [&](const uint4* __restrict__ wrp, int buf, int t) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const uint4* p = wrp + (size_t)((t * MB + mb) * 3) * 64;
      wa[buf][mb][0].u = p[0];
      wa[buf][mb][1].u = p[64];
      wa[buf][mb][2].u = p[128];
    }
  };
  // WDB (streamed weights, double-buffered): K-step 0 of a stage multiplies from its own registers w0, requested during the
  // stage BEFORE (at its K-step 1) together with K-step 1 (-> wa[1], at its last K-step): both are issued ahead of that
  // stage's epilogue stores, so the first MFMAs of a stage never wait for those stores to be acknowledged (vmcnt is in-order).
  BV w0[Cfg::WDB && !Cfg::WRES ? MB : 1][3];
  auto load_w0 = [&](const uint4* __restrict__ wrp) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      w0[mb][0].u = wrp[(size_t)(mb * 3) * 64];
      w0[mb][1].u = wrp[(size_t)(mb * 3 + 1) * 64];
      w0[mb][2].u = wrp[(size_t)(mb * 3 + 2) * 64];
    }
  };
  // WRES: every K-step's weights are loaded ONCE.  The consumer waves then issue no vector-memory loads in the stage loop,
  // so nothing ever waits (s_waitcnt vmcnt counts loads and stores in order) for the epilogue stores of the stage before.
  BV wres[Cfg::WRES ? Cfg::KSTEPS : 1][3];
  if (Cfg::WRES) {
#pragma unroll
    for (int t = 0; t < Cfg::KSTEPS; ++t) {
      wres[t][0].u = wl[(size_t)(t * 3) * 64];
      wres[t][1].u = wl[(size_t)(t * 3 + 1) * 64];
      wres[t][2].u = wl[(size_t)(t * 3 + 2) * 64];
    }
  } else if (Cfg::WDB) {
    load_w0(wl);                                       // first stage: K-steps 0 and 1 requested before the barrier
    load_w_from(wl, 1, 1);
  } else {
    load_w_from(wl, 0, 0);
  }
  float4 bvr[MB];                                      // bias of this lane's four couts per 16-cout block
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int co = PAIR ? 4 * (g & 1) : mb * 16 + 4 * g;
    bvr[mb] = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();                                     // #0
  int st = 0;
  for (int tile = tile0; tile < tile1; ++tile) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int t = 0; t < Cfg::NTW; ++t) acc[mb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int rd = 0; rd < rounds; ++rd, ++st) {
      const unsigned char* tbuf = lds + (st & 1) * Cfg::LDSB;
      // Flat software pipeline over the NS = KSTEPS x (NT / NG) stages of the round: the operands of step s + 1 (data
      // from LDS; the weights of the next K-step from L1 / L2) are requested BEFORE the 6 NG MB MFMAs of step s.
      const uint4* __restrict__ wr = wl + (size_t)rd * Cfg::KSTEPS * MB * 3 * 64;
      constexpr int NGRP = Cfg::NTW / Cfg::NG, NS = Cfg::KSTEPS * NGRP;
      BV bd[2][Cfg::NG][3];
      auto load_w = [&](int buf, int t) { load_w_from(wr, buf, t); };
      const uint4* __restrict__ wnext = wl + (size_t)(rd + 1 < rounds ? rd + 1 : 0) * Cfg::KSTEPS * MB * 3 * 64;
      auto load_b = [&](int buf, int t, int grp) {
        const unsigned char* bp = tbuf + b_base + toff[t];
#pragma unroll
        for (int q = 0; q < Cfg::NG; ++q) {
          const int ti = wh * Cfg::NTW + grp * Cfg::NG + q, tz = ti / Cfg::XT, txr = ti % Cfg::XT;
          const unsigned char* b = bp + ((tz * S * Cfg::IY) * Cfg::IXP + txr * 16) * POSB;
          bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
          bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
          bd[buf][q][2].u = *reinterpret_cast<const uint4*>(b + 32);
        }
      };
      load_b(0, 0, 0);                                 // (the K-step-0 weights were requested before the stage barrier)
#pragma unroll
      for (int ss = 0; ss < NS; ++ss) {
        const int t = ss / NGRP, grp = ss % NGRP;
        const int wb = Cfg::WDB ? (t & 1) : 0, db = ss & 1;
        if (ss + 1 < NS) load_b(db ^ 1, (ss + 1) / NGRP, (ss + 1) % NGRP);
        if (Cfg::WRES) {
        } else if (Cfg::WDB) {
          if (grp == 0) {
            if (t == 0) load_w(0, 2);                                   // K-step 2 -> wa[0] (K-step 0 multiplies from w0)
            else if (t >= 2 && t + 1 < Cfg::KSTEPS) load_w(wb ^ 1, t + 1);
            if (t == 1) load_w0(wnext);                                 // next stage: K-step 0 -> w0 (free since K-step 0)
            if (t == Cfg::KSTEPS - 1) load_w_from(wnext, 1, 1);        // next stage: K-step 1 -> wa[1] (free since K-step L - 1)
          }
        } else if (grp == 0 && t > 0) {
          load_w(0, t);
        }
        const int t0 = grp * Cfg::NG;
        // Term-major over independent accumulators; the sched_barriers pin that order and keep the operand requests above
        // ahead of the MFMAs (left alone, the machine scheduler sinks the loads to their first use).
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (Cfg::WRES) {
            SBF_TERMS(acc[mb], t0, Cfg::NG, wres[t], bd[db]);
          } else if (Cfg::WDB && t == 0) {
            SBF_TERMS(acc[mb], t0, Cfg::NG, w0[mb], bd[db]);
          } else {
            SBF_TERMS(acc[mb], t0, Cfg::NG, wa[wb][mb], bd[db]);
          }
        }
      }
      if (rd + 1 == rounds) {
        // ---- epilogue: lane -> voxel j of the run, couts 16 mb + 4 g + 0..3: one 16-byte channels-last store ----
        SBF_TILE(tile, tx_i, ty_i, tz_i);
        const int ox0 = tx_i * Cfg::TX, oy = ty_i * Cfg::TY + wy, oz0 = tz_i * Cfg::TZ;
        if (oy < Ho) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int co = PAIR ? 4 * (g & 1) : mb * 16 + 4 * g;   // PAIR: rows = (x parity g >> 1, cout)
            if (co >= Cout) continue;                          // Cout % 4 == 0 (host)
            const float4 bv = bvr[mb];
#pragma unroll
            for (int tl = 0; tl < Cfg::NTW; ++tl) {
              const int ti = wh * Cfg::NTW + tl, tz = ti / Cfg::XT, txr = ti % Cfg::XT;
              const int oz = oz0 + tz, ox = PAIR ? ox0 + txr * 32 + 2 * j + (g >> 1) : ox0 + txr * 16 + j;
              if (oz >= Do || ox >= Wo) continue;
              const size_t base = ((size_t)((size_t)oz * Ho + oy) * Wo + ox) * Cout + co;
              float4 o = make_float4(acc[mb][tl].x + bv.x, acc[mb][tl].y + bv.y, acc[mb][tl].z + bv.z, acc[mb][tl].w + bv.w);
              if (act == CDS_ACT_RELU) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
              }
              if (skip) {
                const float4 s4 = *reinterpret_cast<const float4*>(skip + base);
                o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
              }
#ifdef CDS_EXP_SBF_NOSTORE
              if (o.x == 1234.5f)
#endif
              sbf_store4(out + base, o);
            }
          }
        }
      }
      // the first K-step's weights of the next stage travel during the epilogue stores and the barrier wait
      if (!Cfg::WDB && st + 1 < nstages) load_w_from(wnext, 0, 0);
      __syncthreads();                                 // #(st + 1)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// z-marching variant of the Cin = 8 pair layer (conv0: the largest kernel of the network).
// What bounds the tiled kernel above on this layer (measured): its K-loop runs at the matrix pipe's peak, but a 32 x 4 x 4
// tile stages 34 x 6 x 6 input positions, 2.4 per output voxel, and those halo re-reads mostly miss L2 (FETCH_SIZE: 3.7-4.5 GB
// for a 2.0 GB input) while the four producer waves split every one of them again.  Here a workgroup owns a 32 x 8 column
// and marches along z: the staged input lives in a ring of 8 z-planes (34 x 10 positions each) in LDS, a stage computes three
// output planes from five resident input planes while the producers stage the next three: 1.33 input positions per output
// voxel, no re-read along z at all, and 1.5x the MFMAs per workgroup barrier.  Same operand layout, weights resident in
// registers, consumer waves free of vector-memory loads as above.
// ---------------------------------------------------------------------------------------------
struct ZCfg {
  static constexpr int TX = 32, TY = 8, G = 3, R = 2 * G + 2;
  static constexpr int IX = TX + 2, IY = TY + 2, IXH = IX / 2, IXP = IX;
  static constexpr int PPOS = IY * IXP, PLANEB = PPOS * POSB;       // one z-plane of the ring: 340 positions, 16,320 B
  static constexpr int LDSB = R * PLANEB;                           // 130,560 B
  static constexpr int CW = 4, PW = 4, THREADS = (CW + PW) * 64;
  static constexpr int KSTEPS = 9;
  static constexpr int NTW = 2 * G;                                 // N-tiles per consumer wave and stage: 2 rows x G planes
  static constexpr int NG = G;                                      // operands of one row's G planes in registers together
};

__global__ __launch_bounds__(ZCfg::THREADS) void conv3d_sbf_zm_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int D, int H, int W, int act, int tiles_x, int tiles_y,
                                                                     int zseg) {
  using Cfg = ZCfg;
  constexpr int Cin = 8, Cout = 8, G = Cfg::G, R = Cfg::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // unit = (column tx, ty; z segment): z segments of a column are consecutive workgroups
  const int nseg = (D + zseg - 1) / zseg;
  int unit = cds_xcd_remap(blockIdx.x, gridDim.x);
  const int seg = unit % nseg;
  unit /= nseg;
  const int tx_i = unit % tiles_x, ty_i = unit / tiles_x;
  const int z0 = seg * zseg, z1 = min(D, z0 + zseg);
  const int nstages = (z1 - z0 + G - 1) / G;
  const int gx0 = tx_i * Cfg::TX - 1, gy0 = ty_i * Cfg::TY - 1;
  // input plane p (-1 <= p - z0, any p) lives in ring slot (p - z0 + 1) % R

  if (wave >= Cfg::CW) {
    // ============================== producers ==============================
    constexpr int PT = Cfg::PW * 64;
    const int ptid = tid - Cfg::CW * 64;
    constexpr int NP = G * Cfg::PPOS, PPT = (NP + PT - 1) / PT;     // positions of the G planes of a stage
    int s_pl[PPT], s_yx[PPT], s_dst[PPT];
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * PT + ptid;
      const int pl = p / Cfg::PPOS, pp = p - pl * Cfg::PPOS;
      const int row = pp / Cfg::IXP, q = pp - row * Cfg::IXP;
      const int c = 2 * (q % Cfg::IXH) + q / Cfg::IXH;            // de-interleaved x parities (as the tiled pair kernel)
      s_pl[h] = p < NP ? pl : -1;
      s_yx[h] = (row << 10) | c;
      s_dst[h] = (row * Cfg::IXP + q) * POSB;
    }
    float4 va[2][PPT], vb[2][PPT];
    auto load_pos = [&](int z, int yx, float4& a, float4& b) {
      const int gy = gy0 + (yx >> 10), gx = gx0 + (yx & 1023);
      const bool ok = (unsigned)z < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float* __restrict__ src = x + ((size_t)((size_t)z * H + gy) * W + gx) * Cin;
      a = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      b = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // stage st needs input planes z0 + st G - 1 .. z0 + st G + G; its NEW planes are the last G of them
    auto issue = [&](int st, int set) {
#pragma unroll
      for (int h = 0; h < PPT; ++h)
        if (s_pl[h] >= 0) load_pos(z0 + st * G + 1 + s_pl[h], s_yx[h], va[set][h], vb[set][h]);
    };
    auto deposit = [&](int st, int set) {
      const int slot0 = (st * G + 2) % R;                            // slot of plane z0 + st G + 1
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        if (s_pl[h] < 0) continue;
        int slot = slot0 + s_pl[h];
        slot = slot >= R ? slot - R : slot;
        split_store8(lds + slot * Cfg::PLANEB + s_dst[h], va[set][h], vb[set][h]);
      }
    };
    // the two lowest planes of the segment (z0 - 1, z0): loaded, split and stored directly
    for (int p = ptid; p < 2 * Cfg::PPOS; p += PT) {
      const int pl = p / Cfg::PPOS, pp = p - pl * Cfg::PPOS;
      const int row = pp / Cfg::IXP, q = pp - row * Cfg::IXP;
      const int c = 2 * (q % Cfg::IXH) + q / Cfg::IXH;
      float4 a, b;
      load_pos(z0 - 1 + pl, (row << 10) | c, a, b);
      split_store8(lds + pl * Cfg::PLANEB + (row * Cfg::IXP + q) * POSB, a, b);
    }
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
    deposit(0, 0);
    if (nstages > 2) issue(2, 0);
    __syncthreads();                                   // #0: stage 0 staged
    for (int st = 0; st < nstages; st += 2) {
      if (st + 1 < nstages) {
        deposit(st + 1, 1);
        if (st + 3 < nstages) issue(st + 3, 1);
      }
      __syncthreads();                                 // #(st + 1): stage st consumed, stage st + 1 staged
      if (st + 1 >= nstages) break;
      if (st + 2 < nstages) {
        deposit(st + 2, 0);
        if (st + 4 < nstages) issue(st + 4, 0);
      }
      __syncthreads();                                 // #(st + 2)
    }
    return;
  }

  // ============================== consumers: wave = rows wave and wave + 4 ==============================
  SBF_CONSUMER_PRIO();
  const int j = lane & 15, g = lane >> 4;
  const int kxl = (g & 1) * 2 + (g >> 1);              // lane groups take x' = 0, 2, 1, 3 (see the tiled kernel)
  const int lane_base = (wave * Cfg::IXP + j + (kxl & 1) * Cfg::IXH + (kxl >> 1)) * POSB;
  const uint4* __restrict__ wl = wsp + lane;
  BV wres[Cfg::KSTEPS][3];
#pragma unroll
  for (int t = 0; t < Cfg::KSTEPS; ++t) {
    wres[t][0].u = wl[(size_t)(t * 3) * 64];
    wres[t][1].u = wl[(size_t)(t * 3 + 1) * 64];
    wres[t][2].u = wl[(size_t)(t * 3 + 2) * 64];
  }
  const int co = 4 * (g & 1);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int ox = tx_i * Cfg::TX + 2 * j + (g >> 1);
  f32x4 acc[Cfg::NTW];                                 // [row r][plane i] -> r * G + i
  __syncthreads();                                     // #0
  for (int st = 0; st < nstages; ++st) {
#pragma unroll
    for (int t = 0; t < Cfg::NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LDS address of input plane z0 + st G - 1 + u (u = 0 .. G + 1) for this lane
    const unsigned char* vpl[G + 2];
    {
      int slot = (st * G) % R;
#pragma unroll
      for (int u = 0; u < G + 2; ++u) {
        vpl[u] = lds + slot * Cfg::PLANEB + lane_base;
        slot = slot + 1 >= R ? slot + 1 - R : slot + 1;
      }
    }
    // steps: K-step t = (kz, ky) x row r; the G planes of a row are one operand group
    constexpr int NS = Cfg::KSTEPS * 2;
    BV bd[2][G][3];
    auto load_b = [&](int buf, int ss) {
      const int t = ss >> 1, r = ss & 1;
      const int kz = t / 3, ky = t - 3 * kz;
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const unsigned char* b = vpl[i + kz] + ((r * 4 + ky) * Cfg::IXP) * POSB;
        bd[buf][i][0].u = *reinterpret_cast<const uint4*>(b);
        bd[buf][i][1].u = *reinterpret_cast<const uint4*>(b + 16);
        bd[buf][i][2].u = *reinterpret_cast<const uint4*>(b + 32);
      }
    };
    load_b(0, 0);
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      const int t = ss >> 1, r = ss & 1, db = ss & 1;
      if (ss + 1 < NS) load_b(db ^ 1, ss + 1);
      __builtin_amdgcn_sched_barrier(0);
      SBF_TERMS(acc, r * G, G, wres[t], bd[db]);
    }
    // ---- epilogue ----
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = ty_i * Cfg::TY + wave + 4 * r;
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int oz = z0 + st * G + i;
        if (oz >= z1 || oy >= H || ox >= W) continue;
        const f32x4 a = acc[r * G + i];
        float4 o = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
        if (act == CDS_ACT_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        sbf_store4(out + ((size_t)((size_t)oz * H + oy) * W + ox) * Cout + co, o);
      }
    }
    __syncthreads();                                   // #(st + 1)
  }
}

// Round 3 variant: EIGHT consumer waves (two per SIMD, one output row each) + four producers = 12 waves, three on every SIMD; the 27
// weight vectors move from registers (108 VGPRs, one consumer wave per SIMD with the 256-register budget) into LDS (27 KB next to the
// 130 KB ring).  Why: one wave per SIMD issues v_mfma_f32_16x16x32_bf16 every 10.0 ns, two waves every 8.1 ns (scripts/ubench), and the
// second wave covers the first one's epilogue stores and barrier waits.
struct ZCfg8 : ZCfg {
  static constexpr int CW = 8, THREADS = (CW + PW) * 64;
  static constexpr int WBYTES = KSTEPS * 3 * 1024;
  static constexpr int LDSB8 = LDSB + WBYTES;                       // 158,208 B
};

__global__ __launch_bounds__(ZCfg8::THREADS, 3) void conv3d_sbf_zm8_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int D, int H, int W, int act, int tiles_x, int tiles_y,
                                                                     int zseg) {
  using Cfg = ZCfg8;
  constexpr int Cin = 8, Cout = 8, G = Cfg::G, R = Cfg::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // unit = (column tx, ty; z segment): z segments of a column are consecutive workgroups
  const int nseg = (D + zseg - 1) / zseg;
  int unit = cds_xcd_remap(blockIdx.x, gridDim.x);
  const int seg = unit % nseg;
  unit /= nseg;
  const int tx_i = unit % tiles_x, ty_i = unit / tiles_x;
  const int z0 = seg * zseg, z1 = min(D, z0 + zseg);
  const int nstages = (z1 - z0 + G - 1) / G;
  const int gx0 = tx_i * Cfg::TX - 1, gy0 = ty_i * Cfg::TY - 1;
  // input plane p (-1 <= p - z0, any p) lives in ring slot (p - z0 + 1) % R
  {
    uint4* wdst = reinterpret_cast<uint4*>(lds + Cfg::LDSB);
    for (int i = tid; i < Cfg::WBYTES / 16; i += Cfg::THREADS) wdst[i] = wsp[i];
  }

  if (wave >= Cfg::CW) {
    // ============================== producers ==============================
    constexpr int PT = Cfg::PW * 64;
    const int ptid = tid - Cfg::CW * 64;
    constexpr int NP = G * Cfg::PPOS, PPT = (NP + PT - 1) / PT;     // positions of the G planes of a stage
    int s_pl[PPT], s_yx[PPT], s_dst[PPT];
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * PT + ptid;
      const int pl = p / Cfg::PPOS, pp = p - pl * Cfg::PPOS;
      const int row = pp / Cfg::IXP, q = pp - row * Cfg::IXP;
      const int c = 2 * (q % Cfg::IXH) + q / Cfg::IXH;            // de-interleaved x parities (as the tiled pair kernel)
      s_pl[h] = p < NP ? pl : -1;
      s_yx[h] = (row << 10) | c;
      s_dst[h] = (row * Cfg::IXP + q) * POSB;
    }
    float4 va[2][PPT], vb[2][PPT];
    auto load_pos = [&](int z, int yx, float4& a, float4& b) {
      const int gy = gy0 + (yx >> 10), gx = gx0 + (yx & 1023);
      const bool ok = (unsigned)z < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float* __restrict__ src = x + ((size_t)((size_t)z * H + gy) * W + gx) * Cin;
      a = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      b = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // stage st needs input planes z0 + st G - 1 .. z0 + st G + G; its NEW planes are the last G of them
    auto issue = [&](int st, int set) {
#pragma unroll
      for (int h = 0; h < PPT; ++h)
        if (s_pl[h] >= 0) load_pos(z0 + st * G + 1 + s_pl[h], s_yx[h], va[set][h], vb[set][h]);
    };
    auto deposit = [&](int st, int set) {
      const int slot0 = (st * G + 2) % R;                            // slot of plane z0 + st G + 1
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        if (s_pl[h] < 0) continue;
        int slot = slot0 + s_pl[h];
        slot = slot >= R ? slot - R : slot;
        split_store8(lds + slot * Cfg::PLANEB + s_dst[h], va[set][h], vb[set][h]);
      }
    };
    // the two lowest planes of the segment (z0 - 1, z0): loaded, split and stored directly
    for (int p = ptid; p < 2 * Cfg::PPOS; p += PT) {
      const int pl = p / Cfg::PPOS, pp = p - pl * Cfg::PPOS;
      const int row = pp / Cfg::IXP, q = pp - row * Cfg::IXP;
      const int c = 2 * (q % Cfg::IXH) + q / Cfg::IXH;
      float4 a, b;
      load_pos(z0 - 1 + pl, (row << 10) | c, a, b);
      split_store8(lds + pl * Cfg::PLANEB + (row * Cfg::IXP + q) * POSB, a, b);
    }
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
    deposit(0, 0);
    if (nstages > 2) issue(2, 0);
    __syncthreads();                                   // #0: stage 0 staged
    for (int st = 0; st < nstages; st += 2) {
      if (st + 1 < nstages) {
        deposit(st + 1, 1);
        if (st + 3 < nstages) issue(st + 3, 1);
      }
      __syncthreads();                                 // #(st + 1): stage st consumed, stage st + 1 staged
      if (st + 1 >= nstages) break;
      if (st + 2 < nstages) {
        deposit(st + 2, 0);
        if (st + 4 < nstages) issue(st + 4, 0);
      }
      __syncthreads();                                 // #(st + 2)
    }
    return;
  }

  // ============================== consumers: wave = output row ==============================
  SBF_CONSUMER_PRIO();
  const int j = lane & 15, g = lane >> 4;
  const int kxl = (g & 1) * 2 + (g >> 1);              // lane groups take x' = 0, 2, 1, 3 (see the tiled kernel)
  const int lane_base = (wave * Cfg::IXP + j + (kxl & 1) * Cfg::IXH + (kxl >> 1)) * POSB;
  const unsigned char* wlds = lds + Cfg::LDSB + lane * 16;
  const int co = 4 * (g & 1);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int ox = tx_i * Cfg::TX + 2 * j + (g >> 1);
  f32x4 acc[G];                                        // [plane i]
  __syncthreads();                                     // #0
  for (int st = 0; st < nstages; ++st) {
#pragma unroll
    for (int t = 0; t < G; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // LDS address of input plane z0 + st G - 1 + u (u = 0 .. G + 1) for this lane
    const unsigned char* vpl[G + 2];
    {
      int slot = (st * G) % R;
#pragma unroll
      for (int u = 0; u < G + 2; ++u) {
        vpl[u] = lds + slot * Cfg::PLANEB + lane_base;
        slot = slot + 1 >= R ? slot + 1 - R : slot + 1;
      }
    }
    // steps: K-step t = (kz, ky); the G planes of this wave's row are one operand group; weights of the step from LDS
    constexpr int NS = Cfg::KSTEPS;
    BV bd[2][G][3];
    BV wa[2][3];
    auto load_b = [&](int buf, int t) {
      const int kz = t / 3, ky = t - 3 * kz;
      wa[buf][0].u = *reinterpret_cast<const uint4*>(wlds + (t * 3) * 1024);
      wa[buf][1].u = *reinterpret_cast<const uint4*>(wlds + (t * 3 + 1) * 1024);
      wa[buf][2].u = *reinterpret_cast<const uint4*>(wlds + (t * 3 + 2) * 1024);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const unsigned char* b = vpl[i + kz] + (ky * Cfg::IXP) * POSB;
        bd[buf][i][0].u = *reinterpret_cast<const uint4*>(b);
        bd[buf][i][1].u = *reinterpret_cast<const uint4*>(b + 16);
        bd[buf][i][2].u = *reinterpret_cast<const uint4*>(b + 32);
      }
    };
    load_b(0, 0);
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      const int db = ss & 1;
      if (ss + 1 < NS) load_b(db ^ 1, ss + 1);
      __builtin_amdgcn_sched_barrier(0);
      SBF_TERMS(acc, 0, G, wa[db], bd[db]);
    }
    // ---- epilogue ----
    {
      const int oy = ty_i * Cfg::TY + wave;
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int oz = z0 + st * G + i;
        if (oz >= z1 || oy >= H || ox >= W) continue;
        const f32x4 a = acc[i];
        float4 o = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
        if (act == CDS_ACT_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        sbf_store4(out + ((size_t)((size_t)oz * H + oy) * W + ox) * Cout + co, o);
      }
    }
    __syncthreads();                                   // #(st + 1)
  }
}

#ifndef CDS_ZM8_DEFAULT
#define CDS_ZM8_DEFAULT false
#endif
int launch_fwd_zm(const float* x, const void* wsp, const float* b, float* out, int D, int H, int W, int act, hipStream_t st) {
  using Cfg = ZCfg;
  const int tx = cds_ceil_div(W, Cfg::TX), ty = cds_ceil_div(H, Cfg::TY);
  // z segments: whole columns when there are enough of them for ~4 rounds of workgroups (every segment re-stages two halo planes and
  // refills the pipeline: M1 1.56 / 1.59 / 1.63 / 1.72 ms with 1 / 2 / 4 / 8 segments), else segments of >= 8 stages
  int nseg = 1;
  while (tx * ty * nseg < 256 * 4 && cds_ceil_div(D, nseg * 2) >= 8 * Cfg::G) nseg *= 2;
  static const int nseg_env = []() { const char* e = getenv("CDS_ZM_NSEG"); return e ? atoi(e) : 0; }();   // A/B knob
  if (nseg_env > 0) nseg = nseg_env;
  int zseg = cds_ceil_div(cds_ceil_div(D, nseg), Cfg::G) * Cfg::G;
  nseg = cds_ceil_div(D, zseg);
  static const bool zm8 = []() { const char* e = getenv("CDS_ZM8"); return e ? e[0] == '1' : CDS_ZM8_DEFAULT; }();   // A/B knob
  if (zm8) {
    static std::atomic<unsigned long long> lds_ok8{0};
    if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(conv3d_sbf_zm8_kernel), ZCfg8::LDSB8, lds_ok8)) return e_lds;
    hipLaunchKernelGGL(conv3d_sbf_zm8_kernel, dim3(tx * ty * nseg), dim3(ZCfg8::THREADS), ZCfg8::LDSB8, st, x,
                       reinterpret_cast<const uint4*>(wsp), b, out, D, H, W, act, tx, ty, zseg);
    return cds_launch_status();
  }
  static std::atomic<unsigned long long> lds_ok{0};
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(conv3d_sbf_zm_kernel), Cfg::LDSB, lds_ok)) return e_lds;
  hipLaunchKernelGGL(conv3d_sbf_zm_kernel, dim3(tx * ty * nseg), dim3(Cfg::THREADS), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(wsp), b,
                     out, D, H, W, act, tx, ty, zseg);
  return cds_launch_status();
}

template <int S, int MB, int TX, int TZ, bool PAIR = false, bool WRES_ = false>
int launch_fwd(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
               int W, int act, hipStream_t st) {
  using Cfg = FCfg<S, MB, TX, TZ, PAIR, WRES_>;
  static_assert(Cfg::KSTEPS % 2 == 1, "the weight ring assumes an even last K-step");
  static_assert(2 * Cfg::LDSB <= 160 * 1024, "two LDS tile buffers above 160 KB");
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY), tz = cds_ceil_div(Do, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  // tiles per workgroup: enough workgroups for ~6 rounds of one per CU (two where the LDS allows), at most 32 tiles each
  // (a sweep of 4 / 8 / 16 / 32 / 64 at M1 moves single layers by a few percent either way: conv4 likes 8-16, conv6 likes 1)
  static const int tpw_env = []() { const char* e = getenv("CDS_SBF_TPW"); return e ? atoi(e) : 0; }();   // A/B knob
  int tpw = tpw_env > 0 ? tpw_env : max(1, min(32, ntiles / (256 * 6)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  auto kern = conv3d_sbf_kernel<S, MB, TX, TZ, PAIR, WRES_>;
  constexpr int lds_bytes = 2 * Cfg::LDSB;
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> lds_ok{0};   // per instantiation
    if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok)) return e_lds;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(Cfg::THREADS), lds_bytes, st, x, reinterpret_cast<const uint4*>(wsp), b, skip, out, Cin,
                     Cout, D, H, W, Do, Ho, Wo, act, tx, ty, ntiles, tpw);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Transposed convolution k3 s2 p1 op1 (ConvTranspose3d, models/module.py:125-160), channels-last, split-bf16.
// Output voxel o = 2 a + p per axis: parity 0 takes tap k = 1 of input cell a; parity 1 takes tap 2 of cell a and tap 0 of
// cell a + 1.  A workgroup stages 32 x 4 x 1 input cells (+1 halo on the high sides) and produces all 8 parity classes:
// class c = (pz, py, px) is a small convolution over (1 + pz)(1 + py)(1 + px) taps.
//   MERGE = false (Cout % 16 == 0): one accumulator set per class, M rows = 16 couts; K-steps per round: 1,1,1,1,1,1,1,2.
//   MERGE = true  (Cout == 8): the two x parities of a (pz, py) class share an MFMA: M rows = (px, cout), K = its taps x
//     {cell a, cell a + 1}; K-steps per round: 1,1,1,2; a lane's four rows are 4 couts of ONE output voxel, and the two
//     voxels of a cell are adjacent in memory: each wave store covers 1 KB of contiguous channels-last output.
// ---------------------------------------------------------------------------------------------
template <bool MERGE>
struct DTab {
  static constexpr int NCLS = MERGE ? 4 : 8;
  static constexpr int NKS = MERGE ? 5 : 9;                 // K-steps per round over all classes
  // K-step index -> class, first tap slot
  __host__ __device__ static constexpr int cls_of(int ks) { return MERGE ? (ks < 3 ? ks : 3) : (ks < 7 ? ks : 7); }
  __host__ __device__ static constexpr int slot0_of(int ks) { return MERGE ? (ks == 4 ? 4 : 0) : (ks == 8 ? 4 : 0); }
  // tap slot s of class c -> (dz, dy, dx) cell offsets, or -1 when the slot is padding
  __host__ __device__ static constexpr int ntaps(int c) {
    return MERGE ? (1 + (c >> 1)) * (1 + (c & 1)) * 2 : (1 + (c >> 2)) * (1 + ((c >> 1) & 1)) * (1 + (c & 1));
  }
  __host__ __device__ static constexpr int tap_d(int c, int s, int axis) {   // axis 0 = z, 1 = y, 2 = x
    const int pz = MERGE ? (c >> 1) : (c >> 2), py = MERGE ? (c & 1) : ((c >> 1) & 1), px = MERGE ? 1 : (c & 1);
    const int nz = 1 + pz, ny = 1 + py, nx = 1 + px;
    if (s >= nz * ny * nx) return -1;
    const int iz = s / (ny * nx), iy = (s / nx) % ny, ix = s % nx;
    return axis == 0 ? iz : axis == 1 ? iy : ix;
  }
};

struct DCfg {
  static constexpr int CX = 32, CY = 4;
  static constexpr int XT = CX / 16, NT = XT;
  static constexpr int IX = CX + 1, IY = CY + 1, IZ = 2;
  static constexpr int IXP = 40;
  static constexpr int NPOS = IZ * IY * IXP;
  static constexpr int LDSB = NPOS * POSB;
};

#ifndef CDS_DECONV_SBF_MINW
#define CDS_DECONV_SBF_MINW 2   // minimum waves per SIMD of the Cout = 8 (memory-bound) variant: A/B build knob
#endif
template <bool MERGE, int MB>
#ifndef CDS_DECONV_NM_MINW
#define CDS_DECONV_NM_MINW 2   // waves per SIMD of the Cout = 16 / 32 variants (A/B build knob)
#endif
__global__ __launch_bounds__(256, (MERGE ? CDS_DECONV_SBF_MINW : (MB == 1 ? CDS_DECONV_NM_MINW : 2))) void deconv3d_sbf_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                              const float* __restrict__ bias, const float* __restrict__ skip,
                                                              float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                              int act, int out_planar, int tiles_x, int tiles_y, int ntiles,
                                                              int tpw) {
  using Cfg = DCfg;
  using Tab = DTab<MERGE>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = cell row y of this wave inside the tile
  const int j = lane & 15, g = lane >> 4;
  const int nwg = gridDim.x;
  const int wg = cds_xcd_remap(blockIdx.x, nwg);
  const int tile0 = wg * tpw, tile1 = min(ntiles, tile0 + tpw);
  const int rounds = Cin >> 3;

  int toff[Tab::NKS];
#pragma unroll
  for (int ks = 0; ks < Tab::NKS; ++ks) {
    const int c = Tab::cls_of(ks), s0 = Tab::slot0_of(ks);
    int off = 0;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int dz = Tab::tap_d(c, s0 + gg, 0), dy = Tab::tap_d(c, s0 + gg, 1), dx = Tab::tap_d(c, s0 + gg, 2);
      const int o = dz < 0 ? 0 : ((dz * Cfg::IY + dy) * Cfg::IXP + dx) * POSB;
      off = g == gg ? o : off;
    }
    toff[ks] = off;
  }
  const int b_base = (wave * Cfg::IXP + j) * POSB;

  constexpr int NP = Cfg::IZ * Cfg::IY * Cfg::IX;
  constexpr int PPT = (NP + 255) / 256;
  int s_rel[PPT], s_dst[PPT];
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    const int p = h * 256 + tid;
    const int row = p / Cfg::IX, c = p - row * Cfg::IX;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    s_rel[h] = p < NP ? ((rz << 20) | (ry << 10) | c) : -1;
    s_dst[h] = (row * Cfg::IXP + c) * POSB;
  }
  float4 va[PPT], vb[PPT];
  auto issue = [&](int tile, int rd) {
    SBF_TILE(tile, tx_i, ty_i, az);
    const int gx0 = tx_i * Cfg::CX, gy0 = ty_i * Cfg::CY;
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int gz = az + (s_rel[h] >> 20), gy = gy0 + ((s_rel[h] >> 10) & 1023), gx = gx0 + (s_rel[h] & 1023);
      const bool ok = s_rel[h] >= 0 && gz < D && gy < H && gx < W;
      const float* __restrict__ src = x + ((size_t)((size_t)gz * H + gy) * W + gx) * Cin + rd * 8;
      va[h] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      vb[h] = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int h = 0; h < PPT; ++h)
      if (s_rel[h] >= 0) split_store8(lds + s_dst[h], va[h], vb[h]);
  };

  f32x4 acc[Tab::NCLS][MB][Cfg::NT];
  constexpr bool SKIP_PF = MERGE;                      // residual prefetched ahead of the last round's MFMAs (register budget)
  constexpr bool WDB = MB == 1;                        // weights double-buffered across K-steps (register budget)
  float4 skv[SKIP_PF ? Tab::NCLS : 1][Cfg::NT];
  const uint4* __restrict__ wl = wsp + lane;
  const int Ho = 2 * H, Wo = 2 * W;
  if (tile0 < tile1) issue(tile0, 0);
  for (int tile = tile0; tile < tile1; ++tile) {
    SBF_TILE(tile, tx_i, ty_i, az);
    const int ay = ty_i * Cfg::CY + wave;
#pragma unroll
    for (int c = 0; c < Tab::NCLS; ++c)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < Cfg::NT; ++t) acc[c][mb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int rd = 0; rd < rounds; ++rd) {
      __syncthreads();
      deposit();
      __syncthreads();
      if (rd + 1 < rounds) issue(tile, rd + 1);
      else if (tile + 1 < tile1) issue(tile + 1, 0);
      if (SKIP_PF && skip && rd + 1 == rounds && ay < H) {
        // the U-Net skip tensor of this tile's outputs: requested now, added after the MFMAs
#pragma unroll
        for (int c = 0; c < Tab::NCLS; ++c) {
          const int pz = MERGE ? (c >> 1) : (c >> 2), py = MERGE ? (c & 1) : ((c >> 1) & 1);
          const int px = MERGE ? (g >> 1) : (c & 1);
          const int co = MERGE ? 4 * (g & 1) : 4 * g;
          const size_t rowbase = ((size_t)(2 * az + pz) * Ho + (2 * ay + py)) * Wo;
#pragma unroll
          for (int q = 0; q < Cfg::NT; ++q) {
            const int ax = min(tx_i * Cfg::CX + q * 16 + j, W - 1);
            skv[c][q] = *reinterpret_cast<const float4*>(skip + (rowbase + 2 * ax + px) * Cout + co);
          }
        }
      }
      // K-steps of all classes, software-pipelined: the operands of K-step ks + 1 are requested before the MFMAs of ks
      const uint4* __restrict__ wr = wl + (size_t)rd * Tab::NKS * MB * 3 * 64;
      BV wa[2][MB][3];
      BV bd[2][Cfg::NT][3];
      auto load_w = [&](int buf, int ks) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const uint4* p = wr + (size_t)((ks * MB + mb) * 3) * 64;
          wa[buf][mb][0].u = p[0];
          wa[buf][mb][1].u = p[64];
          wa[buf][mb][2].u = p[128];
        }
      };
      auto load_b = [&](int buf, int ks) {
        const unsigned char* bp = lds + b_base + toff[ks];
#pragma unroll
        for (int q = 0; q < Cfg::NT; ++q) {
          const unsigned char* b = bp + q * 16 * POSB;
          bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
          bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
          bd[buf][q][2].u = *reinterpret_cast<const uint4*>(b + 32);
        }
      };
      load_w(0, 0);
      load_b(0, 0);
#pragma unroll
      for (int ks = 0; ks < Tab::NKS; ++ks) {
        const int c = Tab::cls_of(ks), cur = ks & 1, wcur = WDB ? cur : 0;
        if (!WDB && ks > 0) load_w(0, ks);
        if (ks + 1 < Tab::NKS) {
          load_b(cur ^ 1, ks + 1);
          if (WDB) load_w(cur ^ 1, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          SBF_TERMS(acc[c][mb], 0, Cfg::NT, wa[wcur][mb], bd[cur]);
        }
      }
    }

    // ---- epilogue ----
    if (ay < H) {
#pragma unroll
      for (int c = 0; c < Tab::NCLS; ++c) {
        const int pz = MERGE ? (c >> 1) : (c >> 2), py = MERGE ? (c & 1) : ((c >> 1) & 1);
        const int px = MERGE ? (g >> 1) : (c & 1);
        const size_t rowbase = ((size_t)(2 * az + pz) * Ho + (2 * ay + py)) * Wo;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int co = MERGE ? 4 * (g & 1) : mb * 16 + 4 * g;
          const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < Cfg::NT; ++q) {
            const int ax = tx_i * Cfg::CX + q * 16 + j;
            if (ax >= W) continue;
            const size_t base = (rowbase + 2 * ax + px) * Cout + co;
            const f32x4 a = acc[c][mb][q];
            float4 o = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
            if (act == CDS_ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (skip) {
              const float4 s4 = SKIP_PF ? skv[SKIP_PF ? c : 0][q] : *reinterpret_cast<const float4*>(skip + base);
              o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
            }
            if (out_planar) {
              // [Cout][2D][2H][2W] for a planar consumer (the prob layer): per component the lanes of a store cover runs of
              // 32 consecutive x (both x parities of 16 cells) -> whole 128-byte segments
              const size_t ovol = (size_t)(2 * D) * Ho * Wo;
              float* po = out + (size_t)co * ovol + (rowbase + 2 * ax + px);
#ifdef CDS_SBF_NTSTORE
              __builtin_nontemporal_store(o.x, po); __builtin_nontemporal_store(o.y, po + ovol);
              __builtin_nontemporal_store(o.z, po + 2 * ovol); __builtin_nontemporal_store(o.w, po + 3 * ovol);
#else
              po[0] = o.x; po[ovol] = o.y; po[2 * ovol] = o.z; po[3 * ovol] = o.w;
#endif
            } else {
              sbf_store4(out + base, o);
            }
          }
        }
      }
    }
  }
}

template <bool MERGE, int MB>
int launch_deconv(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
                  int W, int act, int out_planar, hipStream_t st) {
  using Cfg = DCfg;
  const int tx = cds_ceil_div(W, Cfg::CX), ty = cds_ceil_div(H, Cfg::CY);
  const int ntiles = tx * ty * D;
  static const int tpw_env = []() { const char* e = getenv("CDS_SBF_TPW"); return e ? atoi(e) : 0; }();   // A/B knob
  int tpw = tpw_env > 0 ? tpw_env : max(1, min(16, ntiles / (256 * 2 * 8)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  hipLaunchKernelGGL((deconv3d_sbf_kernel<MERGE, MB>), dim3(nwg), dim3(256), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(wsp),
                     b, skip, out, Cin, Cout, D, H, W, act, out_planar, tx, ty, ntiles, tpw);
  return cds_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Transposed convolution to Cout = 8 (conv11: the full-resolution, memory-bound layer), warp-specialised.
// In the kernel above every wave loads, splits, multiplies and stores; s_waitcnt vmcnt counts loads and stores in
// order, so each weight wait of the K-loop also waited for the HBM loads of the next tile and the stores of the tile
// before (16 % matrix-pipe utilisation, 3.7 TB/s).  Here 4 consumer waves (cell row y) touch vector memory only for the
// residual prefetch (one tile ahead of its use) and the fire-and-forget output stores: the split weights of ALL rounds sit
// in LDS (rounds x 5 K-steps x 3 KB), the staged input tile is double-buffered and filled by 2 producer waves that run
// two stages ahead.  One workgroup barrier per stage (tile, 8-channel round); 2-3 workgroups per CU.
// ---------------------------------------------------------------------------------------------
// Round 3: the 6-wave (4 + 2) workgroups of round 2 never paired up on a CU (PMC: 5.9 resident waves per CU, SQ_WAVE_CYCLES x 4 /
// (GRBM_GUI_ACTIVE / 8 x 256): a 6-wave workgroup occupies the SIMDs 2-2-1-1 and a second one does not fit the 3-waves-per-SIMD register
// budget on the two fuller SIMDs).  CYT = 8: ONE 12-wave workgroup per CU (8 consumer waves = 8 cell rows, 4 producers): 3 waves on every
// SIMD, twice the loads in flight, two consumer waves per SIMD issuing MFMAs.  CYT = 4 keeps the old shape (A/B, small volumes).
template <int CYT>
struct DWSCfg {
  static constexpr int CX = 32, CY = CYT;
  static constexpr int XT = CX / 16, NT = XT;
  static constexpr int IX = CX + 1, IY = CY + 1, IZ = 2;
  static constexpr int IXP = 40;
  static constexpr int NPOS = IZ * IY * IXP;
  static constexpr int LDSB = NPOS * POSB;
  static constexpr int CW = CYT, PW = CYT / 2, THREADS = (CW + PW) * 64;
};

#ifndef CDS_DWS_MINW
#define CDS_DWS_MINW 3
#endif
template <int CYT>
__global__ __launch_bounds__((DWSCfg<CYT>::THREADS), CDS_DWS_MINW) void deconv3d_sbf_ws_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                                       const float* __restrict__ bias, const float* __restrict__ skip,
                                                                       float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                                       int act, int out_planar, int tiles_x, int tiles_y, int ntiles,
                                                                       int tpw) {
  using Cfg = DWSCfg<CYT>;
  constexpr int DWS_CW = Cfg::CW, DWS_PW = Cfg::PW, DWS_THREADS = Cfg::THREADS;
  using Tab = DTab<true>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  const int wg = cds_xcd_remap(blockIdx.x, nwg);
  const int tile0 = wg * tpw, tile1 = min(ntiles, tile0 + tpw);
  const int rounds = Cin >> 3;
  const int nstages = (tile1 - tile0) * rounds;
  if (nstages <= 0) return;
  const int wbytes = rounds * Tab::NKS * 3 * 1024;          // the layer's split weights: [round][K-step][term][lane] x 16 B
  unsigned char* tiles = lds + wbytes;
  {
    uint4* wdst = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < wbytes / 16; i += DWS_THREADS) wdst[i] = wsp[i];
  }

  if (wave >= DWS_CW) {
    // ============================== producers ==============================
    const int ptid = tid - DWS_CW * 64;
    constexpr int NP = Cfg::IZ * Cfg::IY * Cfg::IX;
    constexpr int PPT = (NP + DWS_PW * 64 - 1) / (DWS_PW * 64);
    int s_rel[PPT], s_dst[PPT];
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * DWS_PW * 64 + ptid;
      const int row = p / Cfg::IX, c = p - row * Cfg::IX;
      const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
      s_rel[h] = p < NP ? ((rz << 20) | (ry << 10) | c) : -1;
      s_dst[h] = (row * Cfg::IXP + c) * POSB;
    }
    float4 va[2][PPT], vb[2][PPT];
    auto issue = [&](int st, int set) {
      const int tile = tile0 + st / rounds, rd = st % rounds;
      SBF_TILE(tile, tx_i, ty_i, az);
      const int gx0 = tx_i * Cfg::CX, gy0 = ty_i * Cfg::CY;
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        const int gz = az + (s_rel[h] >> 20), gy = gy0 + ((s_rel[h] >> 10) & 1023), gx = gx0 + (s_rel[h] & 1023);
        const bool ok = s_rel[h] >= 0 && gz < D && gy < H && gx < W;
        const float* __restrict__ src = x + ((size_t)((size_t)gz * H + gy) * W + gx) * Cin + rd * 8;
        va[set][h] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        vb[set][h] = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto deposit = [&](int buf, int set) {
      unsigned char* base = tiles + buf * Cfg::LDSB;
#pragma unroll
      for (int h = 0; h < PPT; ++h)
        if (s_rel[h] >= 0) split_store8(base + s_dst[h], va[set][h], vb[set][h]);
    };
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
    deposit(0, 0);
    if (nstages > 2) issue(2, 0);
    __syncthreads();                                   // #0: weights + stage 0 staged
    for (int st = 0; st < nstages; st += 2) {
      if (st + 1 < nstages) {
        deposit(1, 1);
        if (st + 3 < nstages) issue(st + 3, 1);
      }
      __syncthreads();
      if (st + 1 >= nstages) break;
      if (st + 2 < nstages) {
        deposit(0, 0);
        if (st + 4 < nstages) issue(st + 4, 0);
      }
      __syncthreads();
    }
    return;
  }

  // ============================== consumers ==============================
  SBF_CONSUMER_PRIO();
  const int j = lane & 15, g = lane >> 4;
  int toff[Tab::NKS];
#pragma unroll
  for (int ks = 0; ks < Tab::NKS; ++ks) {
    const int c = Tab::cls_of(ks), s0 = Tab::slot0_of(ks);
    int off = 0;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int dz = Tab::tap_d(c, s0 + gg, 0), dy = Tab::tap_d(c, s0 + gg, 1), dx = Tab::tap_d(c, s0 + gg, 2);
      const int o = dz < 0 ? 0 : ((dz * Cfg::IY + dy) * Cfg::IXP + dx) * POSB;
      off = g == gg ? o : off;
    }
    toff[ks] = off;
  }
  const int b_base = (wave * Cfg::IXP + j) * POSB;
  const unsigned char* wlds = lds + lane * 16;
  const int co = 4 * (g & 1), px = g >> 1;             // rows of the matrix tile = (x parity, cout)
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int Ho = 2 * H, Wo = 2 * W;
  f32x4 acc[Tab::NCLS][Cfg::NT];
  float4 skv[Tab::NCLS][Cfg::NT];
  __syncthreads();                                     // #0
  int st = 0;
  for (int tile = tile0; tile < tile1; ++tile) {
    SBF_TILE(tile, tx_i, ty_i, az);
    const int ay = ty_i * Cfg::CY + wave;
#pragma unroll
    for (int c = 0; c < Tab::NCLS; ++c)
#pragma unroll
      for (int t = 0; t < Cfg::NT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (skip && ay < H) {
      // the U-Net skip rows of this tile: requested a whole tile ahead of the epilogue that adds them
#pragma unroll
      for (int c = 0; c < Tab::NCLS; ++c) {
        const size_t rowbase = ((size_t)(2 * az + (c >> 1)) * Ho + (2 * ay + (c & 1))) * Wo;
#pragma unroll
        for (int q = 0; q < Cfg::NT; ++q) {
          const int ax = min(tx_i * Cfg::CX + q * 16 + j, W - 1);
          skv[c][q] = *reinterpret_cast<const float4*>(skip + (rowbase + 2 * ax + px) * Cout + co);
        }
      }
    }
    for (int rd = 0; rd < rounds; ++rd, ++st) {
      const unsigned char* tbuf = tiles + (st & 1) * Cfg::LDSB;
      const unsigned char* wr = wlds + rd * Tab::NKS * 3 * 1024;
      BV wa[2][3];
      BV bd[2][Cfg::NT][3];
      auto load_w = [&](int buf, int ks) {
        wa[buf][0].u = *reinterpret_cast<const uint4*>(wr + (ks * 3) * 1024);
        wa[buf][1].u = *reinterpret_cast<const uint4*>(wr + (ks * 3 + 1) * 1024);
        wa[buf][2].u = *reinterpret_cast<const uint4*>(wr + (ks * 3 + 2) * 1024);
      };
      auto load_b = [&](int buf, int ks) {
        const unsigned char* bp = tbuf + b_base + toff[ks];
#pragma unroll
        for (int q = 0; q < Cfg::NT; ++q) {
          const unsigned char* b = bp + q * 16 * POSB;
          bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
          bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
          bd[buf][q][2].u = *reinterpret_cast<const uint4*>(b + 32);
        }
      };
      load_w(0, 0);
      load_b(0, 0);
#pragma unroll
      for (int ks = 0; ks < Tab::NKS; ++ks) {
        const int c = Tab::cls_of(ks), cur = ks & 1;
        if (ks + 1 < Tab::NKS) {
          load_b(cur ^ 1, ks + 1);
          load_w(cur ^ 1, ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        SBF_TERMS(acc[c], 0, Cfg::NT, wa[cur], bd[cur]);
      }
      if (rd + 1 == rounds && ay < H) {
        // ---- epilogue ----
#pragma unroll
        for (int c = 0; c < Tab::NCLS; ++c) {
          const size_t rowbase = ((size_t)(2 * az + (c >> 1)) * Ho + (2 * ay + (c & 1))) * Wo;
#pragma unroll
          for (int q = 0; q < Cfg::NT; ++q) {
            const int ax = tx_i * Cfg::CX + q * 16 + j;
            if (ax >= W) continue;
            const size_t base = (rowbase + 2 * ax + px) * Cout + co;
            const f32x4 a = acc[c][q];
            float4 o = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
            if (act == CDS_ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (skip) {
              const float4 s4 = skv[c][q];
              o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
            }
            if (out_planar) {
              const size_t ovol = (size_t)(2 * D) * Ho * Wo;
              float* po = out + (size_t)co * ovol + (rowbase + 2 * ax + px);
#ifdef CDS_SBF_NTSTORE
              __builtin_nontemporal_store(o.x, po); __builtin_nontemporal_store(o.y, po + ovol);
              __builtin_nontemporal_store(o.z, po + 2 * ovol); __builtin_nontemporal_store(o.w, po + 3 * ovol);
#else
              po[0] = o.x; po[ovol] = o.y; po[2 * ovol] = o.z; po[3 * ovol] = o.w;
#endif
            } else {
              sbf_store4(out + base, o);
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

template <int CYT>
int launch_deconv_ws_t(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
                       int W, int act, int out_planar, hipStream_t st) {
  using Cfg = DWSCfg<CYT>;
  const int tx = cds_ceil_div(W, Cfg::CX), ty = cds_ceil_div(H, Cfg::CY);
  const int ntiles = tx * ty * D;
  static const int tpw_env = []() { const char* e = getenv("CDS_SBF_TPW"); return e ? atoi(e) : 0; }();   // A/B knob
  int tpw = tpw_env > 0 ? tpw_env : max(1, min(16, ntiles / (256 * (CYT == 8 ? 1 : 2) * 8)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  const int lds_bytes = (Cin >> 3) * DTab<true>::NKS * 3 * 1024 + 2 * Cfg::LDSB;
  static std::atomic<unsigned long long> lds_ok{0};
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(deconv3d_sbf_ws_kernel<CYT>), 160 * 1024, lds_ok)) return e_lds;
  hipLaunchKernelGGL(deconv3d_sbf_ws_kernel<CYT>, dim3(nwg), dim3(Cfg::THREADS), lds_bytes, st, x, reinterpret_cast<const uint4*>(wsp), b,
                     skip, out, Cin, Cout, D, H, W, act, out_planar, tx, ty, ntiles, tpw);
  return cds_launch_status();
}

int launch_deconv_ws(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
                     int W, int act, int out_planar, hipStream_t st) {
  static const int cyt_env = []() { const char* e = getenv("CDS_DWS_CYT"); return e ? atoi(e) : 0; }();   // A/B knob: 4 | 8
  const bool big = cyt_env ? cyt_env == 8 : (H >= 16 && (long)cds_ceil_div(W, 32) * cds_ceil_div(H, 8) * D >= 2 * 256);
  if (big) return launch_deconv_ws_t<8>(x, wsp, b, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  return launch_deconv_ws_t<4>(x, wsp, b, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
}

// ---------------------------------------------------------------------------------------------
// prob layer: Conv3d(8 -> 1, k3, p1, no bias / BN / ReLU; module.py:303) on a channels-last input, plain fp32 FMAs
// (one output channel cannot fill a matrix tile), planar output [D][H][W] for the soft-argmin.
// Tile 64 x 4 x 4 outputs: lane = x, wave = y, four z outputs per thread.  The input tile (66 x 6 x 6 positions) sits in
// LDS as two channel-half planes of 16-byte slots (lane stride 16 B: conflict-free ds_read_b128); every position read
// feeds up to three outputs (27 x 8 x 4 = 864 FMAs for 108 reads).  The 24 weights of a (ky, kx) column of taps are
// wave-uniform scalars (s_load).  Two workgroups per CU overlap each other's staging.
// ---------------------------------------------------------------------------------------------
struct PCfg {
  static constexpr int TX = 64, TY = 4, TZ = 4;
  static constexpr int IX = TX + 2, IY = TY + 2, IZ = TZ + 2;
  static constexpr int NPOS = IZ * IY * IX;
  static constexpr int LDSB = 2 * NPOS * 16;
};

__global__ __launch_bounds__(256, 2) void prob_cl8_kernel(const float* __restrict__ x, const float* __restrict__ wtap,
                                                          float* __restrict__ out, int D, int H, int W, int tiles_x,
                                                          int tiles_y, int ntiles) {
  using Cfg = PCfg;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  cds_f4* l0 = reinterpret_cast<cds_f4*>(lds);
  cds_f4* l1 = l0 + Cfg::NPOS;
  int tile = cds_xcd_remap(blockIdx.x, ntiles);
  const int tx_i = tile % tiles_x;
  tile /= tiles_x;
  const int ty_i = tile % tiles_y, tz_i = tile / tiles_y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ox0 = tx_i * Cfg::TX, oy0 = ty_i * Cfg::TY, oz0 = tz_i * Cfg::TZ;
  for (int p = tid; p < Cfg::NPOS; p += 256) {
    const int row = p / Cfg::IX, c = p - row * Cfg::IX;
    const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
    const int gz = oz0 - 1 + rz, gy = oy0 - 1 + ry, gx = ox0 - 1 + c;
    const bool ok = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const float* __restrict__ src = x + ((size_t)((size_t)gz * H + gy) * W + gx) * 8;
    const cds_f4 z4 = {0.f, 0.f, 0.f, 0.f};
    l0[p] = ok ? *reinterpret_cast<const cds_f4*>(src) : z4;
    l1[p] = ok ? *reinterpret_cast<const cds_f4*>(src + 4) : z4;
  }
  __syncthreads();
  float acc[Cfg::TZ] = {0.f, 0.f, 0.f, 0.f};
  const cds_f4* p0 = l0 + wave * Cfg::IX + lane;
  const cds_f4* p1 = l1 + wave * Cfg::IX + lane;
#pragma unroll 1
  for (int kyx = 0; kyx < 9; ++kyx) {
    // the 24 weights [kz][ci] of this (ky, kx) column of taps: wave-uniform -> scalar loads into SGPRs
    const float* __restrict__ wk = wtap + __builtin_amdgcn_readfirstlane(kyx * 24);
    const int ky = kyx / 3, kx = kyx - 3 * ky;
    const int off = ky * Cfg::IX + kx;
#pragma unroll
    for (int zp = 0; zp < Cfg::IZ; ++zp) {
      const cds_f4 a = p0[off + zp * Cfg::IY * Cfg::IX], b = p1[off + zp * Cfg::IY * Cfg::IX];
#pragma unroll
      for (int dz = 0; dz < Cfg::TZ; ++dz) {
        const int kz = zp - dz;
        if (kz < 0 || kz > 2) continue;
        float v = acc[dz];
        v = fmaf(a.x, wk[kz * 8 + 0], v); v = fmaf(a.y, wk[kz * 8 + 1], v); v = fmaf(a.z, wk[kz * 8 + 2], v); v = fmaf(a.w, wk[kz * 8 + 3], v);
        v = fmaf(b.x, wk[kz * 8 + 4], v); v = fmaf(b.y, wk[kz * 8 + 5], v); v = fmaf(b.z, wk[kz * 8 + 6], v); v = fmaf(b.w, wk[kz * 8 + 7], v);
        acc[dz] = v;
      }
    }
  }
  const int ox = ox0 + lane, oy = oy0 + wave;
  if (ox < W && oy < H) {
#pragma unroll
    for (int dz = 0; dz < Cfg::TZ; ++dz)
      if (oz0 + dz < D) out[((size_t)(oz0 + dz) * H + oy) * W + ox] = acc[dz];
  }
}

// ---------------------------------------------------------------------------------------------
// conv11 FUSED with the prob layer: ConvTranspose3d(16 -> 8) + BN + ReLU + skip c0 (module.py:299-301,313) and the 3x3 in-plane
// part of Conv3d(8 -> 1) (module.py:303,314).  The 8-channel full-resolution tensor y = c0 + relu(deconv(x)) (2 GB at 640x512x192)
// is never written: a workgroup owns a 32 x 8 fine tile of TWO fine planes (one cell plane az), computes y on the tile plus a
// one-voxel ring (recomputed, 34 x 10), keeps both planes of it in LDS already split into their three bf16 terms, and emits for
// every fine voxel the three in-plane sums
//       P_kz[z][y][x] = sum_{ky, kx, c} w_prob[c][kz][ky][kx] * y[c][z][y + ky - 1][x + kx - 1]          (kz = 0, 1, 2)
// -- the prob layer is separable along z as prob[z] = P_0[z - 1] + P_1[z] + P_2[z + 1], which the soft-argmin kernel adds while it
// reads (cds_softargmin_conf_p3_f32).  12 B per voxel leave the kernel instead of 32 B out + 32 B back in + 4 B out.
//
// ONE 12-wave workgroup per CU (three waves on every SIMD; 6-wave workgroups do not pair up on a CU): 8 compute waves + 4 staging waves.
// Everything a compute wave touches is in LDS -- conv11's split weights (30 KB), the prob weights in Toeplitz split form (15 KB), the
// input tile of both 8-channel rounds (double-buffered across tiles), the skip tile c0 (fp32) -- so the compute waves issue no
// vector-memory load at all (the first version loaded skip and prob weights from global memory in the compute waves: every HBM / L2
// latency ended up exposed behind an s_waitcnt vmcnt(0), 3.2-3.5 ms).  Per tile, two workgroup barriers:
//   A  transposed convolution, MERGE form of deconv3d_sbf_ws_kernel (rows = (x parity, cout), columns = 16 cells, classes (pz, py)):
//      seven column groups -- the cell rows cy = -1 .. 4 and one group with the 12 ring cells (cx = -1 | 16 of every row) -- one per
//      wave (wave 7 idles), both rounds, 60 MFMAs;
//   E  + bias, ReLU, + skip from LDS, zero outside the volume (the prob layer's zero padding), exact 3-way bf16 split, 8-byte LDS
//      writes into the fine planes F[pz][row][x de-interleaved mod 4][term][8 ch];
//   P  in-plane 3x3 part of the prob layer: matrix rows = (kz, x offset 0..3), columns = 8 quads x 2 rows, K = 3 x 6 positions x 8
//      channels (5 K-steps, 30 MFMAs): one unit (plane, row pair) per wave; lane group kz stores P_kz of four consecutive x.
// The staging waves deposit the NEXT tile's input during A and its skip tile during P / the next A, loads a tile ahead in registers.
// ---------------------------------------------------------------------------------------------
#ifndef CDS_FP_IXP
#define CDS_FP_IXP 20
#endif
struct FPCfg {
  static constexpr int TX = 32, TY = 8;
  static constexpr int CXM = 16, CYM = 4;                   // main cells of a tile
  static constexpr int IX = CXM + 3, IY = CYM + 3, IZ = 2;   // staged input cells: cx -1 .. 17, cy -1 .. 5, cz az .. az + 1
  static constexpr int IXP = CDS_FP_IXP;
  static constexpr int INB1 = IZ * IY * IXP * POSB;          // one round of a tile: 13,440 B
  static constexpr int INB = 2 * INB1;                       // both rounds
  static constexpr int FW = 36, FH = TY + 2;                 // fine plane incl. ring: 34 columns, stored as 4 residue runs of 9
  static constexpr int FB1 = FH * FW * POSB;                 // 17,280 B
  static constexpr int SKW = TX + 2, SKH = TY + 2;
  static constexpr int SKB = 2 * SKH * SKW * 32;             // skip tile, two planes, fp32: 21,760 B
  static constexpr int CW = 8, PW = 4, THREADS = (CW + PW) * 64;
  static constexpr int WB = 2 * DTab<true>::NKS * 3 * 1024;  // conv11's split weights: 30,720 B
  static constexpr int PKS = 5;                              // K-steps of the P phase (18 positions + 2 zero slots)
  static constexpr int PWB = PKS * 3 * 1024;                 // prob weights: 15,360 B
  static constexpr int LDSB = WB + PWB + 2 * INB + 2 * FB1 + SKB;   // 156,160 B
};

__global__ __launch_bounds__(FPCfg::THREADS, 3) void deconv_prob_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                                        const float* __restrict__ bias, const float* __restrict__ skip,
                                                                        const uint4* __restrict__ pw, float* __restrict__ out, int Da,
                                                                        int Ha, int Wa, int tiles_x, int tiles_y, int ntiles, int tpw) {
  using Cfg = FPCfg;
  using Tab = DTab<true>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  const int wg = cds_xcd_remap(blockIdx.x, nwg);
  const int tile0 = wg * tpw, tile1 = min(ntiles, tile0 + tpw);
  if (tile0 >= tile1) return;
  const int D = 2 * Da, H = 2 * Ha, W = 2 * Wa;
  constexpr int Cin = 16;
  unsigned char* pwb = lds + Cfg::WB;
  unsigned char* inb = pwb + Cfg::PWB;
  unsigned char* fpl = inb + 2 * Cfg::INB;
  unsigned char* skb = fpl + 2 * Cfg::FB1;
  {
    uint4* wdst = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < Cfg::WB / 16; i += Cfg::THREADS) wdst[i] = wsp[i];
    uint4* pdst = reinterpret_cast<uint4*>(pwb);
    for (int i = tid; i < Cfg::PWB / 16; i += Cfg::THREADS) pdst[i] = pw[i];
  }

  if (wave >= Cfg::CW) {
    // ============================== staging waves ==============================
    const int ptid = tid - Cfg::CW * 64;
    constexpr int PT = Cfg::PW * 64;
    constexpr int NP = Cfg::IZ * Cfg::IY * Cfg::IX;
    constexpr int PPT = (NP + PT - 1) / PT;
    constexpr int NSK = 2 * Cfg::SKH * Cfg::SKW;
    constexpr int SPT = (NSK + PT - 1) / PT;
    int s_rel[PPT], s_dst[PPT];
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * PT + ptid;
      const int row = p / Cfg::IX, c = p - row * Cfg::IX;
      const int rz = row / Cfg::IY, ry = row - rz * Cfg::IY;
      s_rel[h] = p < NP ? ((rz << 20) | (ry << 10) | c) : -1;
      s_dst[h] = (row * Cfg::IXP + c) * POSB;
    }
    float4 va[2][PPT], vb[2][PPT];                      // [round][position]: the input of one tile in flight
    float4 ka[SPT], kb[SPT];                            // the skip tile of one tile in flight
    auto issue_in = [&](int tile) {
      SBF_TILE(tile, tx_i, ty_i, az);
      const int gx0 = tx_i * Cfg::CXM - 1, gy0 = ty_i * Cfg::CYM - 1;
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        const int gz = az + (s_rel[h] >> 20), gy = gy0 + ((s_rel[h] >> 10) & 1023), gx = gx0 + (s_rel[h] & 1023);
        const bool ok = s_rel[h] >= 0 && gz < Da && (unsigned)gy < (unsigned)Ha && (unsigned)gx < (unsigned)Wa;
        const float* __restrict__ src = x + ((size_t)((size_t)gz * Ha + gy) * Wa + gx) * Cin;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          va[rd][h] = ok ? *reinterpret_cast<const float4*>(src + rd * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
          vb[rd][h] = ok ? *reinterpret_cast<const float4*>(src + rd * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
    auto deposit_in = [&](int buf) {
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        unsigned char* base = inb + buf * Cfg::INB + rd * Cfg::INB1;
#pragma unroll
        for (int h = 0; h < PPT; ++h)
          if (s_rel[h] >= 0) split_store8(base + s_dst[h], va[rd][h], vb[rd][h]);
      }
    };
    auto issue_sk = [&](int tile) {
      SBF_TILE(tile, tx_i, ty_i, az);
      const int X0 = tx_i * Cfg::TX, Y0 = ty_i * Cfg::TY;
#pragma unroll
      for (int h = 0; h < SPT; ++h) {
        const int p = h * PT + ptid;
        const int pz = p / (Cfg::SKH * Cfg::SKW), r = p - pz * (Cfg::SKH * Cfg::SKW);
        const int ry = r / Cfg::SKW, rx = r - ry * Cfg::SKW;
        const int gz = 2 * az + pz, gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
        const bool ok = p < NSK && gz < D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const float* __restrict__ src = skip + ((size_t)((size_t)gz * H + gy) * W + gx) * 8;
        ka[h] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        kb[h] = ok ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto deposit_sk = [&]() {
#pragma unroll
      for (int h = 0; h < SPT; ++h) {
        const int p = h * PT + ptid;
        if (p < NSK) {
          float4* d = reinterpret_cast<float4*>(skb + p * 32);
          d[0] = ka[h];
          d[1] = kb[h];
        }
      }
    };
    issue_in(tile0);
    issue_sk(tile0);
    deposit_in(0);
    deposit_sk();
    if (tile0 + 1 < tile1) {
      issue_in(tile0 + 1);
      issue_sk(tile0 + 1);
    }
    __syncthreads();                                    // B0: weights, input and skip of the first tile staged
    for (int tile = tile0; tile < tile1; ++tile) {
      if (tile + 1 < tile1) deposit_in((tile + 1 - tile0) & 1);       // during A(tile): the other input buffer
      if (tile + 2 < tile1) issue_in(tile + 2);
      __syncthreads();                                  // B1
      __syncthreads();                                  // B2: E(tile) has read the skip tile
      if (tile + 1 < tile1) deposit_sk();
      if (tile + 2 < tile1) issue_sk(tile + 2);
    }
    return;
  }

  // ============================== compute waves ==============================
  SBF_CONSUMER_PRIO();
  const int j = lane & 15, g = lane >> 4;
  int toff[Tab::NKS];
#pragma unroll
  for (int ks = 0; ks < Tab::NKS; ++ks) {
    const int c = Tab::cls_of(ks), s0 = Tab::slot0_of(ks);
    int off = 0;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int dz = Tab::tap_d(c, s0 + gg, 0), dy = Tab::tap_d(c, s0 + gg, 1), dx = Tab::tap_d(c, s0 + gg, 2);
      const int o = dz < 0 ? 0 : ((dz * Cfg::IY + dy) * Cfg::IXP + dx) * POSB;
      off = g == gg ? o : off;
    }
    toff[ks] = off;
  }
  // phase A / E: wave = column group: cell row cy = wave - 1 (waves 0..5), the ring cells (wave 6: lane j -> cy = (j >> 1) - 1,
  // cx = -1 | 16), nothing (wave 7)
  const bool has_group = wave < 7;
  const bool ring = wave == 6;
  const int cyq = ring ? (j < 12 ? (j >> 1) - 1 : -1) : wave - 1;
  const int cxq = ring ? ((j & 1) ? Cfg::CXM : -1) : j;
  const bool colq = has_group && (!ring || j < 12);
  const int b_base = has_group ? ((cyq + 1) * Cfg::IXP + (cxq + 1)) * POSB : 0;
  const unsigned char* wlds = lds + lane * 16;
  const int co = 4 * (g & 1), px = g >> 1;              // rows of the transposed-convolution tile = (x parity, cout)
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  // phase P: wave = unit (plane pzp, row pair): lane -> column (quad qd, fine row rr); K-step t multiplies position s = 4 t + g
  const int pzp = wave >> 2;
  const int qd = j & 7, rr = 2 * (wave & 3) + (j >> 3);
  int poff[Cfg::PKS];
#pragma unroll
  for (int t = 0; t < Cfg::PKS; ++t) {
    const int s = min(4 * t + g, 17);                   // slots 18, 19: zero weights, any valid address
    const int ky = s / 6, dx = s - 6 * ky;
    poff[t] = pzp * Cfg::FB1 + ((rr + ky) * Cfg::FW + (dx & 3) * 9 + qd + (dx >> 2)) * POSB;
  }
  const unsigned char* pwl = pwb + lane * 16;
  const size_t planeHW = (size_t)H * W;

  __syncthreads();                                      // B0
  for (int tile = tile0; tile < tile1; ++tile) {
    SBF_TILE(tile, tx_i, ty_i, az);
    const int X0 = tx_i * Cfg::TX, Y0 = ty_i * Cfg::TY;
    // ---------------- phase A: transposed convolution of this wave's column group, both rounds ----------------
    f32x4 acc[Tab::NCLS][1];
#pragma unroll
    for (int c = 0; c < Tab::NCLS; ++c) acc[c][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_group) {
      const unsigned char* tile_in = inb + ((tile - tile0) & 1) * Cfg::INB + b_base;
      BV wa[2][3];
      BV bd[2][1][3];
      auto load_ab = [&](int buf, int s) {               // s = round * NKS + K-step
        const int rd = s / Tab::NKS, ks = s - rd * Tab::NKS;
        const unsigned char* wr = wlds + (rd * Tab::NKS + ks) * 3 * 1024;
        wa[buf][0].u = *reinterpret_cast<const uint4*>(wr);
        wa[buf][1].u = *reinterpret_cast<const uint4*>(wr + 1024);
        wa[buf][2].u = *reinterpret_cast<const uint4*>(wr + 2048);
        const unsigned char* b = tile_in + rd * Cfg::INB1 + toff[ks];
        bd[buf][0][0].u = *reinterpret_cast<const uint4*>(b);
        bd[buf][0][1].u = *reinterpret_cast<const uint4*>(b + 16);
        bd[buf][0][2].u = *reinterpret_cast<const uint4*>(b + 32);
      };
      load_ab(0, 0);
#pragma unroll
      for (int s = 0; s < 2 * Tab::NKS; ++s) {
        const int c = Tab::cls_of(s % Tab::NKS), cur = s & 1;
        if (s + 1 < 2 * Tab::NKS) load_ab(cur ^ 1, s + 1);
        __builtin_amdgcn_sched_barrier(0);
        SBF_TERMS(acc[c], 0, 1, wa[cur], bd[cur]);
      }
    }
    __syncthreads();                                    // B1: F and the skip tile are free / staged
    // ---------------- phase E: y = skip + relu(acc + bias), split, into the fine planes ----------------
    if (has_group) {
#pragma unroll
      for (int c = 0; c < Tab::NCLS; ++c) {
        const int pz = c >> 1;
        const int fy = 2 * cyq + (c & 1), fx = 2 * cxq + px;
        const bool in_tile = colq && fy >= -1 && fy <= Cfg::TY && fx >= -1 && fx <= Cfg::TX;
        if (!in_tile) continue;
        const int gy = Y0 + fy, gx = X0 + fx;
        const bool in_vol = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const float4 s4 = *reinterpret_cast<const float4*>(skb + ((pz * Cfg::SKH + fy + 1) * Cfg::SKW + fx + 1) * 32 + co * 4);
        const f32x4 a = acc[c][0];
        float o0 = fmaxf(a.x + bv.x, 0.f), o1 = fmaxf(a.y + bv.y, 0.f), o2 = fmaxf(a.z + bv.z, 0.f), o3 = fmaxf(a.w + bv.w, 0.f);
        o0 = in_vol ? s4.x + o0 : 0.f; o1 = in_vol ? s4.y + o1 : 0.f; o2 = in_vol ? s4.z + o2 : 0.f; o3 = in_vol ? s4.w + o3 : 0.f;
        uint32_t h0, m0, l0, h1, m1, l1;
        split2(o0, o1, h0, m0, l0);
        split2(o2, o3, h1, m1, l1);
        const int xs = fx + 1;
        unsigned char* d = fpl + pz * Cfg::FB1 + ((fy + 1) * Cfg::FW + (xs & 3) * 9 + (xs >> 2)) * POSB + co * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + 16) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(l0, l1);
      }
    }
    __syncthreads();                                    // B2: both fine planes complete
    // ---------------- phase P: in-plane 3x3 part of the prob layer, one (plane, row pair) unit per wave ----------------
    {
      f32x4 pacc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
      BV pb[2][1][3], pwr[2][3];
      auto load_p = [&](int buf, int t) {
        const unsigned char* b = fpl + poff[t];
        pb[buf][0][0].u = *reinterpret_cast<const uint4*>(b);
        pb[buf][0][1].u = *reinterpret_cast<const uint4*>(b + 16);
        pb[buf][0][2].u = *reinterpret_cast<const uint4*>(b + 32);
        const unsigned char* wq = pwl + t * 3 * 1024;
        pwr[buf][0].u = *reinterpret_cast<const uint4*>(wq);
        pwr[buf][1].u = *reinterpret_cast<const uint4*>(wq + 1024);
        pwr[buf][2].u = *reinterpret_cast<const uint4*>(wq + 2048);
      };
      load_p(0, 0);
#pragma unroll
      for (int t = 0; t < Cfg::PKS; ++t) {
        if (t + 1 < Cfg::PKS) load_p((t & 1) ^ 1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        SBF_TERMS(pacc, 0, 1, pwr[t & 1], pb[t & 1]);
      }
      const int gz = 2 * az + pzp, gy = Y0 + rr, gx = X0 + 4 * qd;
      if (g < 3 && gy < H && gx < W) {
        float* po = out + ((size_t)g * D + gz) * planeHW + (size_t)gy * W + gx;
        sbf_store4(po, make_float4(pacc[0].x, pacc[0].y, pacc[0].z, pacc[0].w));
      }
    }
  }
}

int launch_deconv_prob(const float* x, const void* wsp, const float* b, const float* skip, const void* pw, float* out, int Da, int Ha,
                       int Wa, hipStream_t st) {
  using Cfg = FPCfg;
  const int tx = cds_ceil_div(2 * Wa, Cfg::TX), ty = cds_ceil_div(2 * Ha, Cfg::TY);
  const int ntiles = tx * ty * Da;
  static const int tpw_env = []() { const char* e = getenv("CDS_FP_TPW"); return e ? atoi(e) : 0; }();   // A/B knob
  int tpw = tpw_env > 0 ? tpw_env : max(1, min(32, ntiles / (256 * 6)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  static std::atomic<unsigned long long> lds_ok{0};
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(deconv_prob_kernel), Cfg::LDSB, lds_ok)) return e_lds;
  hipLaunchKernelGGL(deconv_prob_kernel, dim3(nwg), dim3(Cfg::THREADS), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(wsp), b, skip,
                     reinterpret_cast<const uint4*>(pw), out, Da, Ha, Wa, tx, ty, ntiles, tpw);
  return cds_launch_status();
}

}  // namespace

// 3x3x3 convolution (pad 1, stride 1 | 2) in split-bf16 arithmetic on channels-last volumes.  x [D][H][W][Cin],
// out [Do][Ho][Wo][Cout]; weight_split from the host packer (ops.split_pack_conv3d): int16 [Cin/8][7][ceil(Cout/16)][3][64][8].
extern "C" int cds_conv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                                  int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream) {
  if (!x || !weight_split || !out || Cin < 8 || (Cin % 8) || Cout < 4 || (Cout % 4) || Cout > 64 || D < 1 || H < 1 || W < 1 ||
      (stride != 1 && stride != 2 && stride != CDS_SBF_PAIR))
    return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int mb = (Cout + 15) / 16;
  if (stride == CDS_SBF_PAIR) {   // stride 1, Cout == 8, pair-packed weights
    if (Cout != 8) return CDS_EINVAL;
    if (Cin == 8) {
      static const bool tiled = getenv("CDS_SBF_NOZM") != nullptr;   // A/B knob: the tiled kernel
      if (!skip && !tiled) return launch_fwd_zm(x, weight_split, bias, out, D, H, W, act, st);
      return launch_fwd<1, 1, 32, 4, true, true>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
    }
    return launch_fwd<1, 1, 32, 4, true>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
  }
  if (stride == 1) {
    if (mb == 1) return launch_fwd<1, 1, 32, 4>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
    if (mb == 2) return launch_fwd<1, 2, 32, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
    if (mb == 4) return launch_fwd<1, 4, 32, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
    return CDS_EINVAL;
  }
  if (mb == 1) return launch_fwd<2, 1, 16, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
  if (mb == 2) return launch_fwd<2, 2, 16, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
  if (mb == 4) return launch_fwd<2, 4, 16, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
  return CDS_EINVAL;
}

// ConvTranspose3d k3 s2 p1 op1 (+bias +ReLU +residual) in split-bf16 arithmetic on channels-last volumes.  x [D][H][W][Cin]
// -> out [2D][2H][2W][Cout]; weight_split from ops.split_pack_deconv3d (class / K-step tables: DTab above).  Cout == 8 or
// Cout in {16, 32}; Cin % 8 == 0.
extern "C" int cds_deconv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                                    int Cin, int Cout, int D, int H, int W, int act, int out_planar, void* stream) {
  if (!x || !weight_split || !out || Cin < 8 || (Cin % 8) || D < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 8) {
    static const bool old = getenv("CDS_DECONV_OLD") != nullptr;   // A/B knob
    if (Cin <= 32 && !old) return launch_deconv_ws(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
    return launch_deconv<true, 1>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  }
  if (Cout == 16) return launch_deconv<false, 1>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  if (Cout == 32) return launch_deconv<false, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  return CDS_EINVAL;
}

// prob layer (Conv3d 8 -> 1, no bias / activation; models/module.py:303) on a channels-last input: x [D][H][W][8] -> out
// [D][H][W].  weight_tap: fp32 [3 ky][3 kx][3 kz][8 ci] (ops.pack_prob_cl).
extern "C" int cds_conv3d_prob_cl8_f32(const float* x, const float* weight_tap, float* out, int D, int H, int W, void* stream) {
  if (!x || !weight_tap || !out || D < 1 || H < 1 || W < 1) return CDS_EINVAL;
  using Cfg = PCfg;
  const int tx = cds_ceil_div(W, Cfg::TX), ty = cds_ceil_div(H, Cfg::TY), tz = cds_ceil_div(D, Cfg::TZ);
  static std::atomic<unsigned long long> lds_ok{0};
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(prob_cl8_kernel), Cfg::LDSB, lds_ok)) return e_lds;
  hipLaunchKernelGGL(prob_cl8_kernel, dim3(tx * ty * tz), dim3(256), Cfg::LDSB, (hipStream_t)stream, x, weight_tap, out, D, H, W,
                     tx, ty, tx * ty * tz);
  return cds_launch_status();
}

// conv11 + prob fused (deconv_prob_kernel above): x [Da][Ha][Wa][16] channels-last, skip = c0 [2Da][2Ha][2Wa][8], weight_split =
// conv11's split weights (ops.split_pack_deconv3d), prob_split = ops.split_pack_prob_toeplitz(prob.weight); out = the three
// in-plane maps P_kz [3][2Da][2Ha][2Wa] of the prob layer (prob[z] = P_0[z-1] + P_1[z] + P_2[z+1]: cds_softargmin_conf_p3_f32).
extern "C" int cds_deconv3d_prob_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip,
                                         const void* prob_split, float* out_p3, int Da, int Ha, int Wa, void* stream) {
  if (!x || !weight_split || !skip || !prob_split || !out_p3 || Da < 1 || Ha < 1 || Wa < 2 || (Wa & 1)) return CDS_EINVAL;
  return launch_deconv_prob(x, weight_split, bias, skip, prob_split, out_p3, Da, Ha, Wa, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// The prob layer alone on the matrix cores ("a separate light kernel rather than the fused one").
// Conv3d(8 -> 1, k3, p1, no bias) is separable along z:  prob[z] = P_0[z-1] + P_1[z] + P_2[z+1],  P_kz = the in-plane 3x3 8 -> 1
// convolution with the kz slice of the weights - the P phase of deconv_prob_kernel: matrix rows = (kz, x offset 0..3), columns =
// 8 quads x 2 rows, K = 18 in-plane positions x 8 channels (5 K-steps x 6 split-bf16 MFMAs).
// A workgroup owns a 32 x TY pixel column and MARCHES along z through a chunk of planes: every input plane is read once (plus the
// one-voxel ring of the tile: 34 x (TY + 2) positions), split into its three bf16 terms on the way into LDS (double-buffered, one
// barrier per plane), one wave per row pair multiplies it by the Toeplitz weights it keeps in REGISTERS, and the three z-taps meet in
// registers: a lane keeps its P of the last two planes and the kz = 0 lanes gather P_1 / P_2 of the neighbouring lane groups with two
// cross-lane reads - one 4-byte store per voxel, nothing else written.  Algorithmic bytes: 32 B in + 4 B out per voxel.
// ---------------------------------------------------------------------------------------------
namespace {

template <int TY>
struct PMCfg {
  static constexpr int TX = 32;
#ifndef CDS_PROB_FW
#define CDS_PROB_FW 40
#endif
  // fine plane incl. ring: 34 columns stored as 4 residue runs of 9 positions.  Row pitch 40 positions = 1920 B = 128 (mod 256): the two
  // rows a wave's 16-lane group reads (8 quads each, 48 B apart = bank quads {0,3,6,9,12,15,2,5}) fall on disjoint bank quads; with the
  // natural pitch of 36 every ds_read_b128 was a 2-way conflict and the kernel LDS-bound (compute-only 599 us at M1)
  static constexpr int FW = CDS_PROB_FW, FH = TY + 2;
#ifndef CDS_PROB_RUN
#define CDS_PROB_RUN 10
#endif
  // pitch of a residue run: with 10 positions (480 B) the eight positions that consecutive lanes STORE (x, x+1, .. x+7 = residues
  // 0 1 2 3 0 1 2 3 of two run slots) fall on eight different 16-byte bank groups; with the natural 9 every staging store was a 2-way
  // conflict (SQ_LDS_BANK_CONFLICT 45 % of the LDS-active cycles, the LDS 61 % busy: the kernel's bottleneck)
  static constexpr int RUN = CDS_PROB_RUN;
  static constexpr int NPOS = FH * 34;                        // positions staged per plane
  static constexpr int FB1 = FH * FW * POSB;
  static constexpr int NW = TY / 2, THREADS = NW * 64;        // one wave per row pair
  static constexpr int PKS = 5;
  static constexpr int LDSB = 2 * FB1;
  static constexpr int NLD = (NPOS + THREADS - 1) / THREADS;  // positions a thread stages per plane
};

#ifndef CDS_PROB_MINW
#define CDS_PROB_MINW 2
#endif
template <int TY>
__global__ __launch_bounds__(PMCfg<TY>::THREADS, CDS_PROB_MINW) void prob_sbf_kernel(const float* __restrict__ x, const uint4* __restrict__ pw,
                                                                       float* __restrict__ out, int D, int H, int W, int tiles_x,
                                                                       int tiles_y, int zchunk, int nwg) {
  using Cfg = PMCfg<TY>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  int lin = cds_xcd_remap(blockIdx.x, nwg);
  const int tx_i = lin % tiles_x;
  lin /= tiles_x;
  const int ty_i = lin % tiles_y, zc = lin / tiles_y;
  const int X0 = tx_i * Cfg::TX, Y0 = ty_i * TY;
  const int z0 = zc * zchunk, z1 = min(D, z0 + zchunk);

  // the Toeplitz weights of this lane, all K-steps, in registers (loop invariant along the march)
  BV pwr[Cfg::PKS][3];
#pragma unroll
  for (int t = 0; t < Cfg::PKS; ++t)
#pragma unroll
    for (int k = 0; k < 3; ++k) pwr[t][k].u = pw[(t * 3 + k) * 64 + lane];

  const int qd = j & 7, rr = 2 * wave + (j >> 3);
  int poff[Cfg::PKS];
#pragma unroll
  for (int t = 0; t < Cfg::PKS; ++t) {
    const int s = min(4 * t + g, 17);                         // slots 18, 19: zero weights, any valid address
    const int ky = s / 6, dx = s - 6 * ky;
    poff[t] = ((rr + ky) * Cfg::FW + (dx & 3) * Cfg::RUN + qd + (dx >> 2)) * POSB;
  }
  // staging: position p -> (row, xs) of the ringed tile
  int s_dst[Cfg::NLD];
  unsigned s_src[Cfg::NLD];                                   // byte offset inside a plane (H W 32 B < 4 GB): scalar base + 32-bit lane offset
  bool s_ok[Cfg::NLD];
#pragma unroll
  for (int k = 0; k < Cfg::NLD; ++k) {
    const int p = tid + k * Cfg::THREADS;
    const int row = min(p, Cfg::NPOS - 1) / 34, xs = min(p, Cfg::NPOS - 1) % 34;
    const int gy = Y0 + row - 1, gx = X0 + xs - 1;
    s_ok[k] = p < Cfg::NPOS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    s_src[k] = s_ok[k] ? (unsigned)(gy * W + gx) * 32u : 0u;
    s_dst[k] = p < Cfg::NPOS ? (row * Cfg::FW + (xs & 3) * Cfg::RUN + (xs >> 2)) * POSB : -1;
  }
  // tiles whose ring lies inside the image (all but the border tiles) load without the zero-padding selects (workgroup-uniform branch)
  const bool interior = X0 >= 1 && X0 + Cfg::TX + 1 <= W && Y0 >= 1 && Y0 + TY + 1 <= H;
  const size_t plane = (size_t)H * W * 32;                    // bytes
  // two register sets: the loads of plane zp + 2 are issued before the MFMAs of plane zp and written to LDS an iteration later (an HBM
  // round trip is ~4x the 30 MFMAs of a plane: with one plane in flight per workgroup the march waited for memory every plane)
  struct Regs { float4 a[Cfg::NLD], b[Cfg::NLD]; };
  Regs r0, r1;
  auto load_plane = [&](int z, Regs& r) {
#ifdef CDS_PROB_NOLOAD
    z = -1;
#endif
    if ((unsigned)z >= (unsigned)D) {                          // (uniform) the zero plane beyond either end of the volume
#pragma unroll
      for (int k = 0; k < Cfg::NLD; ++k) r.a[k] = r.b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    const unsigned char* __restrict__ xz = reinterpret_cast<const unsigned char*>(x) + (size_t)z * plane;
    if (interior) {
#pragma unroll
      for (int k = 0; k < Cfg::NLD; ++k) {
        const float4* s4 = reinterpret_cast<const float4*>(xz + s_src[k]);
        r.a[k] = s4[0];
        r.b[k] = s4[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < Cfg::NLD; ++k) {
        const float4* s4 = reinterpret_cast<const float4*>(xz + s_src[k]);
        r.a[k] = s_ok[k] ? s4[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        r.b[k] = s_ok[k] ? s4[1] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_plane = [&](int buf, const Regs& r) {
#pragma unroll
    for (int k = 0; k < Cfg::NLD; ++k)
      if (s_dst[k] >= 0) split_store8(lds + buf * Cfg::FB1 + s_dst[k], r.a[k], r.b[k]);
  };

  load_plane(z0 - 1, r0);
  load_plane(z0, r1);
  store_plane(0, r0);
  __syncthreads();
  f32x4 pb1 = (f32x4){0.f, 0.f, 0.f, 0.f}, pb2 = pb1;          // this lane's P of the planes zp - 1 and zp - 2
  const size_t planeHW = (size_t)H * W;
  const int gy = Y0 + rr, gx = X0 + 4 * qd;
  const bool st_ok = g == 0 && gy < H && gx < W;
  int cur = 0;
  // one plane of the march; rn = the register set that holds plane zp + 1 (stored to LDS at the end), rl = the set that is free
  // (plane zp + 2 is loaded into it)
  auto step = [&](int zp, Regs& rl, const Regs& rn) {
    if (zp + 1 < z1) load_plane(zp + 2, rl);                  // in flight during the MFMAs of this plane and of the next one
    f32x4 pacc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
#ifndef CDS_PROB_NOMFMA
    if ((unsigned)zp < (unsigned)D) {
#else
    if (zp == -12345) {
#endif
      const unsigned char* fp = lds + cur * Cfg::FB1;
      // all five K-steps' operands requested up front (60 registers): the LDS round trip is paid once per plane, not once per K-step
      // (a K-step's six dependent MFMAs are shorter than an LDS read under load)
      BV pbv[Cfg::PKS][1][3];
#pragma unroll
      for (int t = 0; t < Cfg::PKS; ++t) {
        const unsigned char* b = fp + poff[t];
        pbv[t][0][0].u = *reinterpret_cast<const uint4*>(b);
        pbv[t][0][1].u = *reinterpret_cast<const uint4*>(b + 16);
        pbv[t][0][2].u = *reinterpret_cast<const uint4*>(b + 32);
      }
#pragma unroll
      for (int t = 0; t < Cfg::PKS; ++t) {
        SBF_MFMA(pacc[0], pwr[t][2], pbv[t][0][0]);
        SBF_MFMA(pacc[0], pwr[t][1], pbv[t][0][1]);
        SBF_MFMA(pacc[0], pwr[t][0], pbv[t][0][2]);
        SBF_MFMA(pacc[0], pwr[t][1], pbv[t][0][0]);
        SBF_MFMA(pacc[0], pwr[t][0], pbv[t][0][1]);
        SBF_MFMA(pacc[0], pwr[t][0], pbv[t][0][0]);
      }
    }
    // prob[zp - 1] = P_0[zp - 2] (kz = 0 lanes: pb2) + P_1[zp - 1] (kz = 1 lanes: pb1) + P_2[zp] (kz = 2 lanes: pacc)
    const int zo = zp - 1;
    f32x4 o;
    o.x = pb2.x + __shfl(pb1.x, j + 16) + __shfl(pacc[0].x, j + 32);
    o.y = pb2.y + __shfl(pb1.y, j + 16) + __shfl(pacc[0].y, j + 32);
    o.z = pb2.z + __shfl(pb1.z, j + 16) + __shfl(pacc[0].z, j + 32);
    o.w = pb2.w + __shfl(pb1.w, j + 16) + __shfl(pacc[0].w, j + 32);
    if (st_ok && zo >= z0 && zo < z1) sbf_store4(out + (size_t)zo * planeHW + (size_t)gy * W + gx, make_float4(o.x, o.y, o.z, o.w));
    pb2 = pb1;
    pb1 = pacc[0];
    if (zp < z1) store_plane(cur ^ 1, rn);
    __syncthreads();
    cur ^= 1;
  };
  for (int zp = z0 - 1; zp <= z1; zp += 2) {
    step(zp, r0, r1);                                         // r0 held plane zp (already in LDS): free; r1 holds plane zp + 1
    if (zp + 1 <= z1) step(zp + 1, r1, r0);
  }
}

template <int TY>
int launch_prob_sbf(const float* x, const void* pw, float* out, int D, int H, int W, hipStream_t st) {
  using Cfg = PMCfg<TY>;
  const int tx = cds_ceil_div(W, Cfg::TX), ty = cds_ceil_div(H, TY);
  static const int zc_env = []() { const char* e = getenv("CDS_PROB_ZCHUNK"); return e ? atoi(e) : 0; }();   // A/B knob
  int zchunk = zc_env > 0 ? zc_env : 48;
  // enough workgroups for 256 CUs: shorter chunks on small grids (2 extra planes per chunk are the price)
  while (zchunk > 8 && (long long)tx * ty * cds_ceil_div(D, zchunk) < 2048) zchunk /= 2;
  if (zchunk > D) zchunk = D;
  const int nz = cds_ceil_div(D, zchunk);
  const int nwg = tx * ty * nz;
  static std::atomic<unsigned long long> lds_ok{0};
  if (Cfg::LDSB > 64 * 1024)
    if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(prob_sbf_kernel<TY>), Cfg::LDSB, lds_ok)) return e_lds;
  hipLaunchKernelGGL(prob_sbf_kernel<TY>, dim3(nwg), dim3(Cfg::THREADS), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(pw), out, D, H,
                     W, tx, ty, zchunk, nwg);
  return cds_launch_status();
}

}  // namespace

// prob layer (Conv3d 8 -> 1, k3, p1; models/module.py:303) on the matrix cores in split-bf16 arithmetic: x [D][H][W][8] channels-last
// -> out [D][H][W].  prob_split = ops.split_pack_prob_toeplitz(prob.weight).  W % 4 == 0.
extern "C" int cds_conv3d_prob_sbf_f32(const float* x, const void* prob_split, float* out, int D, int H, int W, void* stream) {
  if (!x || !prob_split || !out || D < 1 || H < 1 || W < 4 || (W & 3)) return CDS_EINVAL;
  static const int ty_env = []() { const char* e = getenv("CDS_PROB_TY"); return e ? atoi(e) : 0; }();       // A/B knob
  const int ty = ty_env ? ty_env : (H >= 64 ? 16 : 8);
  if (ty == 16) return launch_prob_sbf<16>(x, prob_split, out, D, H, W, (hipStream_t)stream);
  return launch_prob_sbf<8>(x, prob_split, out, D, H, W, (hipStream_t)stream);
}
