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
#include "sbf_common.hpp"

namespace {

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
// F16: split-f16 arithmetic (sbf_common.hpp: two fp16 terms, three products per K-step, tensor scales from device bounds) - same tile
// geometry and weight layout, term 2 unused.
template <int S, int MB, int TX_, int TZ_, bool PAIR = false, bool WRES_ = false, bool F16 = false>
__global__ __launch_bounds__((FCfg<S, MB, TX_, TZ_, PAIR, WRES_>::THREADS)) void conv3d_sbf_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                            const float* __restrict__ bias, const float* __restrict__ skip,
                                                            float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                            int Do, int Ho, int Wo, int act, int tiles_x, int tiles_y,
                                                            int ntiles, int tpw, const float* __restrict__ in_bound, float w_inv,
                                                            float* __restrict__ out_bound) {
  using Cfg = FCfg<S, MB, TX_, TZ_, PAIR, WRES_>;
  constexpr int NT = F16 ? 2 : 3;                       // terms per operand
  // cout split (round 6): gridDim.y workgroups share a tile, each computing MB of the layer's mbtot = MB gridDim.y 16-cout blocks.
  // On the SMALL volumes of the cascade's deep layers a layer is a handful of tiles, each a serial chain of rounds x K-steps x MB
  // MFMA groups (conv6 at 6 x 16 x 20 voxels: 52 us whatever the size): more, shorter workgroups.  gridDim.y == 1: as before.
  const int mbtot = MB * (int)gridDim.y, mb0 = MB * (int)blockIdx.y;
  const float xs = F16 ? sf16_scale(in_bound[0]) : 1.0f;
  const float out_mul = F16 ? w_inv / xs : 1.0f;
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
        if (s_rel[h] >= 0) {
          if (F16) split_store8_f16(base + s_dst[h], va[set][h], vb[set][h], xs);
          else split_store8(base + s_dst[h], va[set][h], vb[set][h]);
        }
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
      const uint4* p = wrp + (size_t)((t * mbtot + mb0 + mb) * 3) * 64;
      wa[buf][mb][0].u = p[0];
      wa[buf][mb][1].u = p[64];
      if (!F16) wa[buf][mb][2].u = p[128];
    }
  };
  // WDB (streamed weights, double-buffered): K-step 0 of a stage multiplies from its own registers w0, requested during the
  // stage BEFORE (at its K-step 1) together with K-step 1 (-> wa[1], at its last K-step): both are issued ahead of that
  // stage's epilogue stores, so the first MFMAs of a stage never wait for those stores to be acknowledged (vmcnt is in-order).
  BV w0[Cfg::WDB && !Cfg::WRES ? MB : 1][3];
  auto load_w0 = [&](const uint4* __restrict__ wrp) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      w0[mb][0].u = wrp[(size_t)((mb0 + mb) * 3) * 64];
      w0[mb][1].u = wrp[(size_t)((mb0 + mb) * 3 + 1) * 64];
      if (!F16) w0[mb][2].u = wrp[(size_t)((mb0 + mb) * 3 + 2) * 64];
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
      if (!F16) wres[t][2].u = wl[(size_t)(t * 3 + 2) * 64];
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
    const int co = PAIR ? 4 * (g & 1) : (mb0 + mb) * 16 + 4 * g;
    bvr[mb] = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float amax = 0.f;                                    // split-f16: running maximum of the magnitudes this lane stores
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
      const uint4* __restrict__ wr = wl + (size_t)rd * Cfg::KSTEPS * mbtot * 3 * 64;
      constexpr int NGRP = Cfg::NTW / Cfg::NG, NS = Cfg::KSTEPS * NGRP;
      BV bd[2][Cfg::NG][3];
      auto load_w = [&](int buf, int t) { load_w_from(wr, buf, t); };
      const uint4* __restrict__ wnext = wl + (size_t)(rd + 1 < rounds ? rd + 1 : 0) * Cfg::KSTEPS * mbtot * 3 * 64;
      auto load_b = [&](int buf, int t, int grp) {
        const unsigned char* bp = tbuf + b_base + toff[t];
#pragma unroll
        for (int q = 0; q < Cfg::NG; ++q) {
          const int ti = wh * Cfg::NTW + grp * Cfg::NG + q, tz = ti / Cfg::XT, txr = ti % Cfg::XT;
          const unsigned char* b = bp + ((tz * S * Cfg::IY) * Cfg::IXP + txr * 16) * POSB;
          bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
          bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
          if (!F16) bd[buf][q][2].u = *reinterpret_cast<const uint4*>(b + 32);
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
          if constexpr (F16) {
            if (Cfg::WRES) {
              SF16_TERMS(acc[mb], t0, Cfg::NG, wres[t], bd[db]);
            } else if (Cfg::WDB && t == 0) {
              SF16_TERMS(acc[mb], t0, Cfg::NG, w0[mb], bd[db]);
            } else {
              SF16_TERMS(acc[mb], t0, Cfg::NG, wa[wb][mb], bd[db]);
            }
          } else {
            if (Cfg::WRES) {
              SBF_TERMS(acc[mb], t0, Cfg::NG, wres[t], bd[db]);
            } else if (Cfg::WDB && t == 0) {
              SBF_TERMS(acc[mb], t0, Cfg::NG, w0[mb], bd[db]);
            } else {
              SBF_TERMS(acc[mb], t0, Cfg::NG, wa[wb][mb], bd[db]);
            }
          }
        }
      }
      if (rd + 1 == rounds) {
        // Take delivery of the next stage's first weights (requested at K-steps 1 and L - 1 above) BEFORE the epilogue stores are
        // issued: gfx9 has ONE vmcnt for loads and stores and they complete out of order with respect to each other, so the compiler
        // turns a wait for a load that has younger stores in flight into vmcnt(0) - the first MFMAs of the next stage then waited for
        // every store of this epilogue to be acknowledged (round 5, scripts/isa_scan.py; the same fix as in K3's plane loop).
        if (Cfg::WDB && !Cfg::WRES) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) {
              asm volatile("" ::"v"(w0[mb][tm].u.x), "v"(w0[mb][tm].u.y), "v"(w0[mb][tm].u.z), "v"(w0[mb][tm].u.w));
              asm volatile("" ::"v"(wa[1][mb][tm].u.x), "v"(wa[1][mb][tm].u.y), "v"(wa[1][mb][tm].u.z), "v"(wa[1][mb][tm].u.w));
            }
        }
        // ---- epilogue: lane -> voxel j of the run, couts 16 mb + 4 g + 0..3: one 16-byte channels-last store ----
        SBF_TILE(tile, tx_i, ty_i, tz_i);
        const int ox0 = tx_i * Cfg::TX, oy = ty_i * Cfg::TY + wy, oz0 = tz_i * Cfg::TZ;
        if (oy < Ho) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int co = PAIR ? 4 * (g & 1) : (mb0 + mb) * 16 + 4 * g;   // PAIR: rows = (x parity g >> 1, cout)
            if (co >= Cout) continue;                          // Cout % 4 == 0 (host)
            const float4 bv = bvr[mb];
#pragma unroll
            for (int tl = 0; tl < Cfg::NTW; ++tl) {
              const int ti = wh * Cfg::NTW + tl, tz = ti / Cfg::XT, txr = ti % Cfg::XT;
              const int oz = oz0 + tz, ox = PAIR ? ox0 + txr * 32 + 2 * j + (g >> 1) : ox0 + txr * 16 + j;
              if (oz >= Do || ox >= Wo) continue;
              const size_t base = ((size_t)((size_t)oz * Ho + oy) * Wo + ox) * Cout + co;
              float4 o = F16 ? make_float4(acc[mb][tl].x * out_mul + bv.x, acc[mb][tl].y * out_mul + bv.y, acc[mb][tl].z * out_mul + bv.z,
                                           acc[mb][tl].w * out_mul + bv.w)
                             : make_float4(acc[mb][tl].x + bv.x, acc[mb][tl].y + bv.y, acc[mb][tl].z + bv.z, acc[mb][tl].w + bv.w);
              if (act == CDS_ACT_RELU) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
              }
              if (skip) {
                const float4 s4 = *reinterpret_cast<const float4*>(skip + base);
                o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
              }
              if (F16) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
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
  if (F16) sf16_publish_bound(amax, out_bound);
}

template <int S, int MB, int TX, int TZ, bool PAIR = false, bool WRES_ = false, bool F16 = false>
int launch_fwd(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
               int W, int act, hipStream_t st, const float* in_bound = nullptr, float w_inv = 1.f, float* out_bound = nullptr,
               int ysplit = 1) {
  using Cfg = FCfg<S, MB, TX, TZ, PAIR, WRES_>;
  static_assert(Cfg::KSTEPS % 2 == 1, "the weight ring assumes an even last K-step");
  static_assert(2 * Cfg::LDSB <= 160 * 1024, "two LDS tile buffers above 160 KB");
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TX), ty = cds_ceil_div(Ho, Cfg::TY), tz = cds_ceil_div(Do, Cfg::TZ);
  const int ntiles = tx * ty * tz;
  // tiles per workgroup: enough workgroups for ~6 rounds of one per CU (two where the LDS allows), at most 32 tiles each
  // (a sweep of 4 / 8 / 16 / 32 / 64 at M1 moves single layers by a few percent either way: conv4 likes 8-16, conv6 likes 1)
  int tpw = max(1, min(32, ntiles / (256 * 6)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  auto kern = conv3d_sbf_kernel<S, MB, TX, TZ, PAIR, WRES_, F16>;
  constexpr int lds_bytes = 2 * Cfg::LDSB;
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> lds_ok{0};   // per instantiation
    if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(kern), lds_bytes, lds_ok)) return e_lds;
  }
  hipLaunchKernelGGL(kern, dim3(nwg, ysplit), dim3(Cfg::THREADS), lds_bytes, st, x, reinterpret_cast<const uint4*>(wsp), b, skip, out, Cin,
                     Cout, D, H, W, Do, Ho, Wo, act, tx, ty, ntiles, tpw, in_bound, w_inv, out_bound);
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
template <bool MERGE, int MB, bool F16 = false>     // F16: split-f16 arithmetic (sbf_common.hpp), scales from device bounds
#ifndef CDS_DECONV_NM_MINW
#define CDS_DECONV_NM_MINW 2   // waves per SIMD of the Cout = 16 / 32 variants (A/B build knob)
#endif
__global__ __launch_bounds__(256, (MERGE ? CDS_DECONV_SBF_MINW : (MB == 1 ? CDS_DECONV_NM_MINW : 2))) void deconv3d_sbf_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                              const float* __restrict__ bias, const float* __restrict__ skip,
                                                              float* __restrict__ out, int Cin, int Cout, int D, int H, int W,
                                                              int act, int out_planar, int tiles_x, int tiles_y, int ntiles,
                                                              int tpw, const float* __restrict__ in_bound, float w_inv,
                                                              float* __restrict__ out_bound) {
  using Cfg = DCfg;
  using Tab = DTab<MERGE>;
  const float xs = F16 ? sf16_scale(in_bound[0]) : 1.0f;
  const float out_mul = F16 ? w_inv / xs : 1.0f;
  float amax = 0.f;
  const int mbtot = MB * (int)gridDim.y, mb0 = MB * (int)blockIdx.y;   // cout split over gridDim.y workgroups per tile (see conv3d_sbf_kernel)
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
      if (s_rel[h] >= 0) {
        if (F16) split_store8_f16(lds + s_dst[h], va[h], vb[h], xs);
        else split_store8(lds + s_dst[h], va[h], vb[h]);
      }
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
      const uint4* __restrict__ wr = wl + (size_t)rd * Tab::NKS * mbtot * 3 * 64;
      BV wa[2][MB][3];
      BV bd[2][Cfg::NT][3];
      auto load_w = [&](int buf, int ks) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const uint4* p = wr + (size_t)((ks * mbtot + mb0 + mb) * 3) * 64;
          wa[buf][mb][0].u = p[0];
          wa[buf][mb][1].u = p[64];
          if (!F16) wa[buf][mb][2].u = p[128];
        }
      };
      auto load_b = [&](int buf, int ks) {
        const unsigned char* bp = lds + b_base + toff[ks];
#pragma unroll
        for (int q = 0; q < Cfg::NT; ++q) {
          const unsigned char* b = bp + q * 16 * POSB;
          bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
          bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
          if (!F16) bd[buf][q][2].u = *reinterpret_cast<const uint4*>(b + 32);
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
          if constexpr (F16) {
            SF16_TERMS(acc[c][mb], 0, Cfg::NT, wa[wcur][mb], bd[cur]);
          } else {
            SBF_TERMS(acc[c][mb], 0, Cfg::NT, wa[wcur][mb], bd[cur]);
          }
        }
      }
    }

    // ---- epilogue ----
    if (ay < H) {
      // Two passes: first every output value is finished in its accumulator registers (bias, ReLU, + skip: all skip loads of the tile
      // in flight together), then all stores are issued.  One pass interleaved load - store - load: on gfx9's single vmcnt every skip
      // wait then drained the store before it (73 full drains per tile in the 64 -> 32 layer, scripts/isa_scan.py).
      float4 bvr[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int co = MERGE ? 4 * (g & 1) : (mb0 + mb) * 16 + 4 * g;
        bvr[mb] = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int c = 0; c < Tab::NCLS; ++c) {
        const int pz = MERGE ? (c >> 1) : (c >> 2), py = MERGE ? (c & 1) : ((c >> 1) & 1);
        const int px = MERGE ? (g >> 1) : (c & 1);
        const size_t rowbase = ((size_t)(2 * az + pz) * Ho + (2 * ay + py)) * Wo;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int co = MERGE ? 4 * (g & 1) : (mb0 + mb) * 16 + 4 * g;
          const float4 bv = bvr[mb];
#pragma unroll
          for (int q = 0; q < Cfg::NT; ++q) {
            const int ax = tx_i * Cfg::CX + q * 16 + j;
            if (ax >= W) continue;
            const size_t base = (rowbase + 2 * ax + px) * Cout + co;
            const f32x4 a = acc[c][mb][q];
            float4 o = F16 ? make_float4(a.x * out_mul + bv.x, a.y * out_mul + bv.y, a.z * out_mul + bv.z, a.w * out_mul + bv.w)
                           : make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
            if (act == CDS_ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (skip) {
              const float4 s4 = SKIP_PF ? skv[SKIP_PF ? c : 0][q] : *reinterpret_cast<const float4*>(skip + base);
              o.x = s4.x + o.x; o.y = s4.y + o.y; o.z = s4.z + o.z; o.w = s4.w + o.w;
            }
            if (F16) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            acc[c][mb][q] = (f32x4){o.x, o.y, o.z, o.w};
          }
        }
      }
#pragma unroll
      for (int c = 0; c < Tab::NCLS; ++c) {
        const int pz = MERGE ? (c >> 1) : (c >> 2), py = MERGE ? (c & 1) : ((c >> 1) & 1);
        const int px = MERGE ? (g >> 1) : (c & 1);
        const size_t rowbase = ((size_t)(2 * az + pz) * Ho + (2 * ay + py)) * Wo;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int co = MERGE ? 4 * (g & 1) : (mb0 + mb) * 16 + 4 * g;
#pragma unroll
          for (int q = 0; q < Cfg::NT; ++q) {
            const int ax = tx_i * Cfg::CX + q * 16 + j;
            if (ax >= W) continue;
            const size_t base = (rowbase + 2 * ax + px) * Cout + co;
            const f32x4 a = acc[c][mb][q];
            const float4 o = make_float4(a.x, a.y, a.z, a.w);
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
  if (F16) sf16_publish_bound(amax, out_bound);
}

template <bool MERGE, int MB, bool F16 = false>
int launch_deconv(const float* x, const void* wsp, const float* b, const float* skip, float* out, int Cin, int Cout, int D, int H,
                  int W, int act, int out_planar, hipStream_t st, const float* in_bound = nullptr, float w_inv = 1.f,
                  float* out_bound = nullptr, int ysplit = 1) {
  using Cfg = DCfg;
  const int tx = cds_ceil_div(W, Cfg::CX), ty = cds_ceil_div(H, Cfg::CY);
  const int ntiles = tx * ty * D;
  int tpw = max(1, min(16, ntiles / (256 * 2 * 8)));
  const int nwg = cds_ceil_div(ntiles, tpw);
  hipLaunchKernelGGL((deconv3d_sbf_kernel<MERGE, MB, F16>), dim3(nwg, ysplit), dim3(256), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(wsp),
                     b, skip, out, Cin, Cout, D, H, W, act, out_planar, tx, ty, ntiles, tpw, in_bound, w_inv, out_bound);
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
  int tpw = max(1, min(16, ntiles / (256 * (CYT == 8 ? 1 : 2) * 8)));
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
  const bool big = (H >= 16 && (long)cds_ceil_div(W, 32) * cds_ceil_div(H, 8) * D >= 2 * 256);
  if (big) return launch_deconv_ws_t<8>(x, wsp, b, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  return launch_deconv_ws_t<4>(x, wsp, b, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
}

}  // namespace

// 3x3x3 convolution (pad 1, stride 1 | 2) in split-bf16 arithmetic on channels-last volumes.  x [D][H][W][Cin],
// out [Do][Ho][Wo][Cout]; weight_split from the host packer (ops.split_pack_conv3d): int16 [Cin/8][7][ceil(Cout/16)][3][64][8].
extern "C" int cds_conv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                                  int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream) {
  if (!x || !weight_split || !out || Cin < 8 || (Cin % 8) || Cout < 4 || (Cout % 4) || Cout > 64 || D < 1 || H < 1 || W < 1 ||
      (stride != 1 && stride != 2 && stride != CDS_SBF_PAIR))
    return CDS_EINVAL;
  if (stride == CDS_SBF_PAIR && Cout != 8) return CDS_EINVAL;   // pair-packed weights exist for Cout == 8 only
  hipStream_t st = (hipStream_t)stream;
  const int mb = (Cout + 15) / 16;
  if (!skip) {   // z-marching kernels (conv3d_zmg.hip) where they cover the shape
    const int r = cds_conv3d_zmg_dispatch(x, weight_split, bias, out, Cin, Cout, D, H, W, stride == CDS_SBF_PAIR ? 1 : stride,
                                          stride == CDS_SBF_PAIR, act, st);
    if (r != CDS_ZMG_UNSUPPORTED) return r;
  }
  if (stride == CDS_SBF_PAIR) {   // stride 1, Cout == 8, pair-packed weights
    if (Cout != 8) return CDS_EINVAL;
    if (Cin == 8) return launch_fwd<1, 1, 32, 4, true, true>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, st);
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

// The same convolution in SPLIT-F16 arithmetic (sbf_common.hpp: two fp16 terms per operand, three products per K-step, fp32-class
// error at half the matrix-pipe work) for the shapes the z-marching kernels cover (CDS_EINVAL otherwise: the caller stays on
// cds_conv3d_sbf_f32).  weight_split: the split-bf16 layouts with fp16 terms (hi, lo, unused) of w * w_scale (ops.split_pack_conv3d*(...,
// f16=True)); w_inv_scale = 1 / w_scale (a power of two); in_bound: DEVICE scalar >= max |x| (the producer's out_bound, or any upper
// bound); out_bound: DEVICE scalar that receives max |out| by atomic maximum (zero it first) or NULL.
extern "C" int cds_conv3d_sf16_f32(const float* x, const void* weight_split, const float* bias, float* out, int Cin, int Cout, int D, int H,
                                   int W, int stride, int act, const float* in_bound, float w_inv_scale, float* out_bound, void* stream) {
  if (!x || !weight_split || !out || !in_bound || Cin < 8 || (Cin % 8) || Cout < 4 || (Cout % 4) || Cout > 64 || D < 1 || H < 1 || W < 1 ||
      (stride != 1 && stride != 2 && stride != CDS_SBF_PAIR) || !(w_inv_scale > 0.f))
    return CDS_EINVAL;
  if (stride == CDS_SBF_PAIR && Cout != 8) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int r = cds_conv3d_zmg_dispatch(x, weight_split, bias, out, Cin, Cout, D, H, W, stride == CDS_SBF_PAIR ? 1 : stride,
                                        stride == CDS_SBF_PAIR, act, st, in_bound, w_inv_scale, out_bound);
  if (r != CDS_ZMG_UNSUPPORTED) return r;
  // the tiled kernels in split-f16: the deep layers (conv4 32 -> 32, conv5 32 -> 64 stride 2, conv6 64 -> 64)
  const int mb = (Cout + 15) / 16;
  // few tiles (the cascade stages' deep layers): one 16-cout block per workgroup, mb workgroups per tile (CDS_SBF_YSPLIT=0: never)
  const int s_ = stride, Do = (D - 1) / s_ + 1, Ho = (H - 1) / s_ + 1, Wo = (W - 1) / s_ + 1;
  const long tiles = (long)cds_ceil_div(Wo, s_ == 1 ? 32 : 16) * cds_ceil_div(Ho, 4) * cds_ceil_div(Do, 2);
  const bool split = tiles <= cds_env_int("CDS_SBF_YSPLIT_TILES", 256) && !cds_env_is("CDS_SBF_YSPLIT", '0');
  if (split && stride == 1 && (mb == 2 || mb == 4))
    return launch_fwd<1, 1, 32, 2, false, false, true>(x, weight_split, bias, nullptr, out, Cin, Cout, D, H, W, act, st, in_bound, w_inv_scale, out_bound, mb);
  if (split && stride == 2 && mb == 4)
    return launch_fwd<2, 1, 16, 2, false, false, true>(x, weight_split, bias, nullptr, out, Cin, Cout, D, H, W, act, st, in_bound, w_inv_scale, out_bound, mb);
  if (stride == 1 && mb == 2) return launch_fwd<1, 2, 32, 2, false, false, true>(x, weight_split, bias, nullptr, out, Cin, Cout, D, H, W, act, st, in_bound, w_inv_scale, out_bound);
  if (stride == 1 && mb == 4) return launch_fwd<1, 4, 32, 2, false, false, true>(x, weight_split, bias, nullptr, out, Cin, Cout, D, H, W, act, st, in_bound, w_inv_scale, out_bound);
  if (stride == 2 && mb == 4) return launch_fwd<2, 4, 16, 2, false, false, true>(x, weight_split, bias, nullptr, out, Cin, Cout, D, H, W, act, st, in_bound, w_inv_scale, out_bound);
  return CDS_EINVAL;
}

// The transposed convolution to Cout = 32 (conv7 of CostRegNet) in SPLIT-F16 arithmetic: weight_split from
// ops.split_pack_deconv3d(..., f16=True), w_inv_scale = 1 / its weight scale, in_bound / out_bound as in cds_conv3d_sf16_f32.
extern "C" int cds_deconv3d_sf16_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out, int Cin,
                                     int Cout, int D, int H, int W, int act, const float* in_bound, float w_inv_scale, float* out_bound,
                                     void* stream) {
  if (!x || !weight_split || !out || !in_bound || Cin < 8 || (Cin % 8) || Cout != 32 || D < 1 || H < 1 || W < 1 || !(w_inv_scale > 0.f))
    return CDS_EINVAL;
  // few tiles (the cascade stages): one 16-cout block per workgroup, two workgroups per tile
  const long tiles = (long)cds_ceil_div(W, 32) * cds_ceil_div(H, 4) * D;
  if (tiles <= cds_env_int("CDS_SBF_YSPLIT_TILES", 256) && !cds_env_is("CDS_SBF_YSPLIT", '0'))
    return launch_deconv<false, 1, true>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, 0, (hipStream_t)stream, in_bound,
                                         w_inv_scale, out_bound, 2);
  return launch_deconv<false, 2, true>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, 0, (hipStream_t)stream, in_bound,
                                       w_inv_scale, out_bound);
}

// ConvTranspose3d k3 s2 p1 op1 (+bias +ReLU +residual) in split-bf16 arithmetic on channels-last volumes.  x [D][H][W][Cin]
// -> out [2D][2H][2W][Cout]; weight_split from ops.split_pack_deconv3d (class / K-step tables: DTab above).  Cout == 8 or
// Cout in {16, 32}; Cin % 8 == 0.
extern "C" int cds_deconv3d_sbf_f32(const float* x, const void* weight_split, const float* bias, const float* skip, float* out,
                                    int Cin, int Cout, int D, int H, int W, int act, int out_planar, void* stream) {
  if (!x || !weight_split || !out || Cin < 8 || (Cin % 8) || D < 1 || H < 1 || W < 1) return CDS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 8) {
    if (Cin <= 32) return launch_deconv_ws(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
    return launch_deconv<true, 1>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  }
  if (Cout == 16) return launch_deconv<false, 1>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  if (Cout == 32) return launch_deconv<false, 2>(x, weight_split, bias, skip, out, Cin, Cout, D, H, W, act, out_planar, st);
  return CDS_EINVAL;
}
