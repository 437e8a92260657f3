// K4, z-marching form of the split-bf16 3x3x3 convolutions for Cin = 8 / 16 and stride 1 / 2.
//
// What the tiled kernels of conv3d_sbf.hip pay (measured in rounds 2 / 3): a TX x 4 x TZ tile stages 2.4 input positions per
// output voxel at stride 1 and 11.6 at stride 2 (8 is the minimum), every 8-channel round of a tile is its own barrier stage, and
// the weights of every K-step stream through the consumer waves' vector-memory queue.  Here a workgroup owns a column of
// TXO x TYO output voxels and MARCHES along z (the layout conv0's conv3d_sbf_zm_kernel introduced, generalised):
//   * the staged input is a RING of z-planes in LDS ([plane][8-channel slice][row][x] x 48 B, the [term][8] bf16 position
//     format of sbf_common.hpp); every input plane is loaded, split and stored ONCE per column: 1.33 (stride 1, 32 x 8) or
//     9.1-9.3 (stride 2) staged positions per output voxel and no re-read along z;
//   * ONE workgroup barrier per G output planes whatever the channel count: all slices of a plane are resident together;
//   * the weights never move: the four consumer waves split the K dimension (8-channel rounds) and the 16-cout blocks between
//     them, so every wave keeps the <= 27 weight vectors of ITS (round, cout block) in registers for the whole kernel
//     (84 / 108 VGPRs) and its stage loop consists of LDS operand reads and MFMAs only.  Waves that split K exchange the
//     partial sums of half of their N-tiles through a double-buffered LDS area across the stage barrier and add them in a
//     fixed order (round 0 + round 1), so results do not depend on timing;
//   * the four producer waves issue unconditional loads: an out-of-volume position reads a 32-byte zero block in global memory
//     (a two-instruction address select) instead of the exec-masked branch per position of the first z-marching kernel.
// Same arithmetic, operand layout and weight packing (ops.split_pack_conv3d / split_pack_conv3d_pair) as the tiled kernels;
// reached through cds_conv3d_sbf_f32 (conv3d_sbf.hip), which keeps the tiled kernels for the shapes not covered here.
// Reference: models/module.py:80-116 (Conv3d + BatchNorm3d + ReLU), :270-315 (CostRegNet).
#include <stdlib.h>

#include "cds_common.hpp"
#include "sbf_common.hpp"

namespace {

__device__ __attribute__((aligned(32))) float g_zmg_zeros[8];   // what an out-of-volume position loads

// CW_ = 4: one consumer wave per SIMD (256-register budget); CW_ = 8: two per SIMD, 12 waves per workgroup, 168 registers: one
// MFMA-issuing wave per SIMD sustains one v_mfma_f32_16x16x32_bf16 per 10.0 ns, two sustain one per 8.1 ns (scripts/ubench), and the
// second wave's K-loop covers the first one's exchange / epilogue / barrier time.
// F16_: split-f16 arithmetic (two fp16 terms per operand, three products per K-step, tensor scales from device bounds: sbf_common.hpp)
// instead of split-bf16 (three bf16 terms, six products); same staging geometry, term 2 of a position / weight vector unused.
template <int S_, int RD_, int MB_, bool PAIR_, int TXO_, int TYO_, int G_, int NG_, int CW_ = 4, bool F16_ = false>
struct ZG {
  static constexpr int S = S_, RD = RD_, MB = MB_, TXO = TXO_, TYO = TYO_, G = G_, NG = NG_;
  static constexpr bool F16 = F16_;
  static constexpr int NT = F16_ ? 2 : 3;                          // terms per operand
  static constexpr bool PAIR = PAIR_, DEINT = PAIR_ || S_ == 2;
  static constexpr int CW = CW_, PW = 4, THREADS = (CW + PW) * 64;
  static constexpr int KSPL = RD, MBS = MB;                       // wave = (round, cout block, row part)
  static_assert(CW % (KSPL * MBS) == 0 && (KSPL == 1 || KSPL == 2 || KSPL == 4), "consumer waves");
  static constexpr int PARTS = CW / (KSPL * MBS);
  static constexpr int ROWS = TYO / PARTS;                        // output rows per consumer wave
  static constexpr int XT = TXO / (PAIR ? 32 : 16);
  static constexpr int NTW = ROWS * XT * G;                       // N-tiles per consumer wave and stage
  static_assert(TYO % PARTS == 0 && NTW % NG == 0 && (KSPL != 2 || NTW % 2 == 0), "tile split");
  static constexpr int KW = PAIR ? 4 : 3, KS = (9 * KW + 3) / 4;  // 7 K-steps of 4 taps (27 + 1 zero) | 9 of (kz, ky) x 4 x'
  static constexpr int IX = (TXO - 1) * S + 3, IY = (TYO - 1) * S + 3;
  static constexpr int IXH = DEINT ? (IX + 1) / 2 : 0;            // x parities de-interleaved (stride 2, pair columns)
  static constexpr int IXP = DEINT ? 2 * IXH : IX;
  static constexpr int SLICEB = IY * IXP * POSB, PLANEB = RD * SLICEB;
  static constexpr int NINIT = 3 - S;                             // planes below the first stage's new ones
  static constexpr int NRES = S * (G - 1) + 3, NNEW = S * G, R = NRES + NNEW;
  static constexpr int RINGB = R * PLANEB;
  // one exchange buffer.  K split in two: every wave sends NTW / 2 N-tiles to its partner.  In four (the 32 -> 8 pair layer: the
  // ring leaves no room for more than three output rows per stage): a slot per (N-tile, source round); N-tile n is finalised by the
  // wave of round n % 4, which adds the four partial sums in the order of the rounds.
  static constexpr int XCH1 = KSPL == 2 ? CW * (NTW / 2) * 1024 : (KSPL == 4 ? CW * NTW * 1024 : 0);
  static constexpr int NFIN = KSPL == 4 ? (NTW + 3) / 4 : 1;          // N-tiles a wave finalises (K split in four)
  // tap of (K-step t, lane group gg): PAIR: (kz, ky) row t, x' = 0, 2, 1, 3 (the two groups of an LDS service group read the same
  // parity plane one position apart); else tap 4 t + gg, tap 27 = zero weights -> any resident position
  static constexpr int tap_of(int t, int gg) {
    const int tap = PAIR ? 4 * t + ((gg & 1) * 2 + (gg >> 1)) : 4 * t + gg;
    return tap > 9 * KW - 1 ? 9 * KW - 1 : tap;
  }
  static constexpr int tap_kz(int t, int gg) { return tap_of(t, gg) / (3 * KW); }
  static constexpr int tap_off(int t, int gg) {                   // byte offset of the tap inside a slice
    const int tap = tap_of(t, gg), ky = (tap / KW) % 3, kx = tap % KW;
    return (ky * IXP + (DEINT ? ((kx & 1) * IXH + (kx >> 1)) : kx)) * POSB;
  }
  static constexpr int LDSB = RINGB + 2 * XCH1;
  static_assert(LDSB <= 160 * 1024, "LDS budget");
  static constexpr int NITEM = NNEW * IY * RD * IXP;              // staged positions per stage (incl. the de-interleave pad)
  static constexpr int PT = PW * 64, PPT = (NITEM + PT - 1) / PT;
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 1) void conv3d_zmg_kernel(const float* __restrict__ x, const uint4* __restrict__ wsp,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int Cout,
                                                                  int D, int H, int W, int Do, int Ho, int Wo, int act,
                                                                  int tiles_x, int tiles_y, int zseg,
                                                                  const float* __restrict__ in_bound, float w_inv,
                                                                  float* __restrict__ out_bound) {
  constexpr int S = Cfg::S, RD = Cfg::RD, G = Cfg::G, R = Cfg::R, Cin = 8 * RD, KS = Cfg::KS, NTW = Cfg::NTW, NG = Cfg::NG;
  constexpr bool PAIR = Cfg::PAIR, F16 = Cfg::F16;
  constexpr int NT = Cfg::NT;
  // split-f16: the input's scale from the bound its producer left, and what the accumulators are multiplied by at the end (exact)
  const float xs = F16 ? sf16_scale(in_bound[0]) : 1.0f;
  const float out_mul = F16 ? w_inv / xs : 1.0f;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // unit = (column tx, ty; z segment): the z segments of a column are consecutive workgroups
  const int nseg = (Do + zseg - 1) / zseg;
  int unit = cds_xcd_remap(blockIdx.x, gridDim.x);
  const int seg = unit % nseg;
  unit /= nseg;
  const int tx_i = unit % tiles_x, ty_i = unit / tiles_x;
  const int z0 = seg * zseg, z1 = min(Do, z0 + zseg);             // output planes of this unit
  const int nstages = (z1 - z0 + G - 1) / G;
  const int gx0 = tx_i * Cfg::TXO * S - 1, gy0 = ty_i * Cfg::TYO * S - 1;
  const int zin0 = S * z0 - 1;                                    // input plane in ring slot 0; plane p lives in slot (p - zin0) % R

  if (wave >= Cfg::CW) {
    // ============================== producers ==============================
    // The producers are the critical path of a stage (s_memtime probe, scripts/ubench/zmg_timeline_*: their split + LDS stores took
    // ~7000 of a conv0 stage's ~8400 cycles next to two MFMA-issuing waves per SIMD, and the consumers then waited for them at the
    // barrier): they issue ahead of the consumer waves whenever they have work.
#ifndef CDS_ZMG_PPRIO
#define CDS_ZMG_PPRIO 3
#endif
    __builtin_amdgcn_s_setprio(CDS_ZMG_PPRIO);
    constexpr int PT = Cfg::PT, PPT = Cfg::PPT;
    const int ptid = tid - Cfg::CW * 64;
    // item = one 8-channel position of the stage's NNEW new planes: x fastest, then slice, row, plane
    int s_pl[PPT], s_dst[PPT];
    long long s_off[PPT];                                         // element offset in x[] for plane index 0 of the volume
    unsigned okmask = 0;                                          // bit h: the item exists and its (y, x) is inside the volume
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const int p = h * PT + ptid;
      const int q = p % Cfg::IXP, t1 = p / Cfg::IXP;
      const int rd = t1 % RD, t2 = t1 / RD;
      const int row = t2 % Cfg::IY, pl = t2 / Cfg::IY;
      const int c = Cfg::DEINT ? (2 * (q % Cfg::IXH) + q / Cfg::IXH) : q;
      const int gy = gy0 + row, gx = gx0 + c;
      const bool exists = p < Cfg::NITEM && c < Cfg::IX;
      s_pl[h] = pl;
      s_dst[h] = rd * Cfg::SLICEB + (row * Cfg::IXP + q) * POSB;
      s_off[h] = ((long long)gy * W + gx) * Cin + rd * 8;
      if (exists && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) okmask |= 1u << h;
      if (!exists) s_pl[h] = -1;
    }
    const long long plane_elems = (long long)H * W * Cin;
    float4 va[2][PPT], vb[2][PPT];
    // stage st: new input planes zin0 + st S G + NINIT + (0 .. NNEW - 1)
    auto issue = [&](int st, int set) {
      const int zn = zin0 + st * S * G + Cfg::NINIT;
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        const int z = zn + s_pl[h];
        const bool ok = ((okmask >> h) & 1u) && (unsigned)z < (unsigned)D;
        const float* __restrict__ src = ok ? x + ((long long)z * plane_elems + s_off[h]) : g_zmg_zeros;
        va[set][h] = *reinterpret_cast<const float4*>(src);
        vb[set][h] = *reinterpret_cast<const float4*>(src + 4);
      }
    };
    auto deposit = [&](int st, int set) {
      const int slot0 = (st * S * G + Cfg::NINIT) % R;
#pragma unroll
      for (int h = 0; h < PPT; ++h) {
        if (h + 1 == PPT && s_pl[h] < 0) continue;                 // only the last item of a thread can be missing
        int slot = slot0 + (s_pl[h] < 0 ? 0 : s_pl[h]);
        slot = slot >= R ? slot - R : slot;
        if (F16) split_store8_f16(lds + slot * Cfg::PLANEB + s_dst[h], va[set][h], vb[set][h], xs);
        else split_store8(lds + slot * Cfg::PLANEB + s_dst[h], va[set][h], vb[set][h]);
      }
    };
    // the NINIT lowest planes of the segment: loaded, split and stored directly
    for (int p = ptid; p < Cfg::NINIT * Cfg::IY * RD * Cfg::IXP; p += PT) {
      const int q = p % Cfg::IXP, t1 = p / Cfg::IXP;
      const int rd = t1 % RD, t2 = t1 / RD;
      const int row = t2 % Cfg::IY, pl = t2 / Cfg::IY;
      const int c = Cfg::DEINT ? (2 * (q % Cfg::IXH) + q / Cfg::IXH) : q;
      const int gz = zin0 + pl, gy = gy0 + row, gx = gx0 + c;
      const bool ok = c < Cfg::IX && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float* __restrict__ src = ok ? x + (((long long)gz * H + gy) * W + gx) * Cin + rd * 8 : g_zmg_zeros;
      if (F16)
        split_store8_f16(lds + pl * Cfg::PLANEB + rd * Cfg::SLICEB + (row * Cfg::IXP + q) * POSB, *reinterpret_cast<const float4*>(src),
                         *reinterpret_cast<const float4*>(src + 4), xs);
      else
        split_store8(lds + pl * Cfg::PLANEB + rd * Cfg::SLICEB + (row * Cfg::IXP + q) * POSB, *reinterpret_cast<const float4*>(src),
                     *reinterpret_cast<const float4*>(src + 4));
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

  // ============================== consumers: wave = (round k, cout block mbi, row part) ==============================
#ifndef CDS_ZMG_CPRIO
#define CDS_ZMG_CPRIO 2
#endif
  __builtin_amdgcn_s_setprio(CDS_ZMG_CPRIO);
  const int k = wave % Cfg::KSPL, mbi = (wave / Cfg::KSPL) % Cfg::MBS, part = wave / (Cfg::KSPL * Cfg::MBS);
  const int j = lane & 15, g = lane >> 4;
  // Two consumer waves per SIMD (CW = 8: waves w and w + 4 share one): the second one runs its epilogue AFTER the stage barrier
  // instead of before it, so each wave's address set-up / exchange / bias / ReLU / stores overlap with the other one's MFMAs instead
  // of both doing them in lock-step on either side of the barrier (which idled the matrix pipe for a third of every stage).
#ifdef CDS_ZMG_NOSKEW
  const bool late = false;                             // A/B knob
#else
  const bool late = Cfg::CW == 8 && wave >= 4;
#endif
  // byte offset of this lane inside a slice: its slice (round k), first row of its part, voxel j; PAIR: + the x' of its lane group
  // (x' = 0, 2, 1, 3: de-interleaved parity plane + half index), the (kz, ky) of a K-step being uniform
  const int lane_base = k * Cfg::SLICEB + (part * Cfg::ROWS * S * Cfg::IXP + j) * POSB +
                        (PAIR ? (((g & 1) * 2 + (g >> 1)) & 1) * Cfg::IXH * POSB + ((((g & 1) * 2 + (g >> 1))) >> 1) * POSB : 0);
  // this wave's weights: [round k][K-step][cout block mbi][term][lane]
  BV wres[KS][NT];
  {
    const uint4* __restrict__ wl = wsp + lane;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const size_t o = (size_t)(((k * KS + t) * Cfg::MB + mbi) * 3) * 64;
#pragma unroll
      for (int tm = 0; tm < NT; ++tm) wres[t][tm].u = wl[o + 64 * tm];
    }
  }
  float amax = 0.f;                                    // running maximum of the magnitudes this lane stores (split-f16: the output's bound)
  const int co = PAIR ? 4 * (g & 1) : mbi * 16 + 4 * g;
  const float4 bv = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  // N-tile n of this wave = (row r, x run xt, plane i): n = (r XT + xt) G + i
  constexpr int XW = PAIR ? 32 : 16;
  const int oy0 = ty_i * Cfg::TYO + part * Cfg::ROWS;
  const int ox0 = tx_i * Cfg::TXO + (PAIR ? 2 * j + (g >> 1) : j);
  float* const obase = out + ((size_t)((size_t)z0 * Ho + oy0) * Wo + ox0) * Cout + co;   // this lane's voxel of N-tile 0, stage 0
  bool lane_ok[Cfg::XT];
#pragma unroll
  for (int xt = 0; xt < Cfg::XT; ++xt) lane_ok[xt] = ox0 + xt * XW < Wo && co < Cout;
  auto store_tile = [&](int n, int st, const f32x4& a) {
    const int i = n % G, rx = n / G, xt = rx % Cfg::XT, r = rx / Cfg::XT;
    if (z0 + st * G + i >= z1 || oy0 + r >= Ho) return;            // wave-uniform
    if (!lane_ok[xt]) return;
    float4 o = F16 ? make_float4(a.x * out_mul + bv.x, a.y * out_mul + bv.y, a.z * out_mul + bv.z, a.w * out_mul + bv.w)
                   : make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
    if (act == CDS_ACT_RELU) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (F16) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    sbf_store4(obase + ((size_t)((size_t)(st * G + i) * Ho + r) * Wo + xt * XW) * Cout, o);
  };
  unsigned char* xch = lds + Cfg::RINGB;                 // [buffer][sender wave][N-tile of its partner's half][lane] x 16 B
  constexpr int NH = NTW / 2;
  f32x4 acc[NTW];
  f32x4 keep[Cfg::KSPL == 2 ? NH : Cfg::NFIN];          // this wave's partial sums of the N-tiles it finalises, across the barrier
  // K split in four: N-tile n known only at run time (n = 4 j + k)
  auto store_tile_rt = [&](int n, int st, const f32x4& a) {
    const int i = n % G, rx = n / G, xt = rx % Cfg::XT, r = rx / Cfg::XT;
    if (z0 + st * G + i >= z1 || oy0 + r >= Ho) return;            // wave-uniform
    if (!(ox0 + xt * XW < Wo && co < Cout)) return;
    float4 o = F16 ? make_float4(a.x * out_mul + bv.x, a.y * out_mul + bv.y, a.z * out_mul + bv.z, a.w * out_mul + bv.w)
                   : make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
    if (act == CDS_ACT_RELU) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (F16) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    sbf_store4(obase + ((size_t)((size_t)(st * G + i) * Ho + r) * Wo + xt * XW) * Cout, o);
  };
  const int xgrp = (wave / Cfg::KSPL) * NTW;            // first exchange slot row of this wave's (cout block, row part)
  // finish stage sp: bias / ReLU / store of this wave's outputs.  K split: N-tiles [k NH, (k + 1) NH) = round 0's partial sum +
  // round 1's (the partner's half arrived through exchange buffer sp & 1 before the stage barrier)
  auto finish = [&](int sp) {
    if (Cfg::KSPL == 2) {
      const unsigned char* src = xch + (sp & 1) * Cfg::XCH1 + ((wave ^ 1) * NH) * 1024 + lane * 16;
#pragma unroll
      for (int m = 0; m < NH; ++m) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(src + m * 1024);
        store_tile(k * NH + m, sp, k == 0 ? keep[m] + o : o + keep[m]);
      }
    } else if (Cfg::KSPL == 4) {
#pragma unroll
      for (int jf = 0; jf < Cfg::NFIN; ++jf) {
        const int n = jf * 4 + k;
        if (n >= NTW) continue;                                      // wave-uniform
        const unsigned char* src = xch + (sp & 1) * Cfg::XCH1 + ((xgrp + n) * 4) * 1024 + lane * 16;
        f32x4 total = k == 0 ? keep[jf] : *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) total = total + (kk == k ? keep[jf] : *reinterpret_cast<const f32x4*>(src + kk * 1024));
        store_tile_rt(n, sp, total);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NTW; ++n) store_tile(n, sp, acc[n]);
    }
  };
  __syncthreads();                                     // #0
  for (int st = 0; st < nstages; ++st) {
    if (late && st > 0) finish(st - 1);
    // LDS byte offsets of this lane's operands.  PAIR: one per resident input plane (the tap row of a K-step is an immediate offset);
    // else one per (K-step, output plane): the four lane groups of a K-step sit in up to two planes (compile-time constants per lane
    // group: three selects per entry, no tables in registers)
    int vpl[PAIR ? Cfg::NRES : 1];
    int vaddr[PAIR ? 1 : KS][PAIR ? 1 : G];
    {
      int sl[Cfg::NRES];
      int slot = (st * S * G) % R;
#pragma unroll
      for (int u = 0; u < Cfg::NRES; ++u) {
        sl[u] = slot * Cfg::PLANEB;
        slot = slot + 1 >= R ? slot + 1 - R : slot + 1;
      }
      if (PAIR) {
#pragma unroll
        for (int u = 0; u < Cfg::NRES; ++u) vpl[u] = lane_base + sl[u];
      } else {
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
          for (int i = 0; i < G; ++i) {
            const int o0 = sl[S * i + Cfg::tap_kz(t, 0)] + Cfg::tap_off(t, 0), o1 = sl[S * i + Cfg::tap_kz(t, 1)] + Cfg::tap_off(t, 1);
            const int o2 = sl[S * i + Cfg::tap_kz(t, 2)] + Cfg::tap_off(t, 2), o3 = sl[S * i + Cfg::tap_kz(t, 3)] + Cfg::tap_off(t, 3);
            vaddr[t][i] = lane_base + (g == 0 ? o0 : (g == 1 ? o1 : (g == 2 ? o2 : o3)));
          }
      }
    }
    constexpr int NGRP = NTW / NG, NS = KS * NGRP;
    BV bd[2][NG][NT];
    auto load_b = [&](int buf, int ss) {
      const int t = ss / NGRP, grp = ss % NGRP;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int n = grp * NG + q, i = n % G, rx = n / G, xt = rx % Cfg::XT, r = rx / Cfg::XT;
        // PAIR: K-step t = (kz, ky) = (t / 3, t % 3)
        const unsigned char* b = lds + (PAIR ? vpl[PAIR ? S * i + t / 3 : 0] + (t % 3) * Cfg::IXP * POSB : vaddr[PAIR ? 0 : t][PAIR ? 0 : i]) +
                                 (r * S * Cfg::IXP + xt * 16) * POSB;
        bd[buf][q][0].u = *reinterpret_cast<const uint4*>(b);
        bd[buf][q][1].u = *reinterpret_cast<const uint4*>(b + 16);
        if (!F16) bd[buf][q][NT - 1].u = *reinterpret_cast<const uint4*>(b + 32);
      }
    };
    load_b(0, 0);
    // a wave that issues MFMAs back to back starves the other waves of its SIMD of issue slots (the producers' deposit went from
    // ~7000 to ~1100 cycles per stage once they outranked the consumers): the K-loop runs at the lowest priority, a consumer's
    // address set-up / exchange / epilogue above it, the producers above both
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
      const int t = ss / NGRP, grp = ss % NGRP, db = ss & 1;
      if (ss + 1 < NS) load_b(db ^ 1, ss + 1);
      __builtin_amdgcn_sched_barrier(0);
#ifndef CDS_ZMG_LAZYWAIT
      // ONE wait for all operands of this step (they were requested a whole step ago) instead of the compiler's lgkmcnt waits
      // BETWEEN the dependent MFMAs: an issue slot between two MFMAs on the same accumulator costs ~40 cycles of matrix pipe
      if (ss + 1 < NS) __builtin_amdgcn_s_waitcnt(0xC07F | ((NT * NG) << 8));   // lgkmcnt(NT NG): the next step's requests stay in flight
      else __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (F16) {
        SF16_TERMS(acc, grp * NG, NG, wres[t], bd[db]);
      } else {
        SBF_TERMS(acc, grp * NG, NG, wres[t], bd[db]);
      }
    }
    __builtin_amdgcn_s_setprio(CDS_ZMG_CPRIO);
    if (Cfg::KSPL == 2) {
      if (!late && st > 0) finish(st - 1);
      // partner's half -> exchange buffer (st & 1); own half stays in registers
      unsigned char* dst = xch + (st & 1) * Cfg::XCH1 + (wave * NH) * 1024 + lane * 16;
#pragma unroll
      for (int m = 0; m < NH; ++m) {
        *reinterpret_cast<f32x4*>(dst + m * 1024) = acc[(k ^ 1) * NH + m];
        keep[m] = acc[k * NH + m];
      }
    } else if (Cfg::KSPL == 4) {
      if (st > 0) finish(st - 1);
      unsigned char* dst = xch + (st & 1) * Cfg::XCH1 + (xgrp * 4 + k) * 1024 + lane * 16;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        if ((n & 3) == k) keep[n >> 2] = acc[n];                     // wave-uniform
        else *reinterpret_cast<f32x4*>(dst + n * 4096) = acc[n];
      }
    } else if (!late) {
      finish(st);
    }
    __syncthreads();                                   // #(st + 1)
  }
  if (Cfg::KSPL >= 2 || late) finish(nstages - 1);
  if (F16) sf16_publish_bound(amax, out_bound);       // this wave's largest stored magnitude -> the bound the next layer scales by
}

// z segments per column: enough workgroups for the 256 CUs (one workgroup per CU: the ring takes most of the LDS) with full
// rounds of them, against the two or three stage times a segment spends filling its pipeline.
inline int zmg_pick_nseg(int cols, int Do, int G) {
  const int nseg_env = cds_env_int("CDS_ZMG_NSEG", 0);   // A/B knob
  if (nseg_env > 0) return nseg_env;
  const int stages = cds_ceil_div(Do, G);
  int best = 1;
  double best_cost = 1e30;
  for (int n = 1; n <= 32 && n <= stages; ++n) {
    const int per = cds_ceil_div(stages, n);
    if (n > 1 && per < 6) break;
    const double cost = (double)cds_ceil_div(cols * cds_ceil_div(stages, per), 256) * (per + 2.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
  }
  return best;
}

template <class Cfg>
int launch_zmg(const float* x, const void* wsp, const float* b, float* out, int Cout, int D, int H, int W, int act, hipStream_t st,
               const float* in_bound = nullptr, float w_inv = 1.f, float* out_bound = nullptr) {
  constexpr int S = Cfg::S;
  const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
  const int tx = cds_ceil_div(Wo, Cfg::TXO), ty = cds_ceil_div(Ho, Cfg::TYO);
  int nseg = zmg_pick_nseg(tx * ty, Do, Cfg::G);
  const int zseg = cds_ceil_div(cds_ceil_div(Do, nseg), Cfg::G) * Cfg::G;
  nseg = cds_ceil_div(Do, zseg);
  auto kern = conv3d_zmg_kernel<Cfg>;
  static std::atomic<unsigned long long> lds_ok{0};     // per instantiation
  if (int e_lds = cds_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDSB, lds_ok)) return e_lds;
  hipLaunchKernelGGL(kern, dim3(tx * ty * nseg), dim3(Cfg::THREADS), Cfg::LDSB, st, x, reinterpret_cast<const uint4*>(wsp), b, out,
                     Cout, D, H, W, Do, Ho, Wo, act, tx, ty, zseg, in_bound, w_inv, out_bound);
  return cds_launch_status();
}

}  // namespace

// Dispatch for cds_conv3d_sbf_f32 (conv3d_sbf.hip): returns CDS_ZMG_UNSUPPORTED when the shape stays on the tiled kernels.
// pair != 0: Cout == 8 with the pair-packed weights (ops.split_pack_conv3d_pair), stride 1.
int cds_conv3d_zmg_dispatch(const float* x, const void* wsp, const float* bias, float* out, int Cin, int Cout, int D, int H, int W,
                            int stride, int pair, int act, hipStream_t st, const float* in_bound, float w_inv, float* out_bound) {
  if (in_bound) {   // split-f16 arithmetic: the same six shapes, scales from device bounds (cds_conv3d_sf16_f32)
    if (pair) {
      if (Cin == 8) return launch_zmg<ZG<1, 1, 1, true, 32, 8, 3, 1, 8, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
      if (Cin == 16) return launch_zmg<ZG<1, 2, 1, true, 32, 8, 1, 4, 4, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
      if (Cin == 32) return launch_zmg<ZG<1, 4, 1, true, 32, 3, 1, 3, 4, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
      return CDS_ZMG_UNSUPPORTED;
    }
    if (stride == 1 && Cin == 16 && Cout == 16)
      return launch_zmg<ZG<1, 2, 1, false, 32, 8, 1, 1, 8, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
    if (stride == 2 && Cin == 8 && Cout == 16)
      return launch_zmg<ZG<2, 1, 1, false, 16, 8, 1, 1, 8, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
    if (stride == 2 && Cin == 16 && Cout == 32)
      return launch_zmg<ZG<2, 2, 2, false, 16, 4, 1, 1, 8, true>>(x, wsp, bias, out, Cout, D, H, W, act, st, in_bound, w_inv, out_bound);
    return CDS_ZMG_UNSUPPORTED;
  }
  const bool off = cds_env_is("CDS_ZMG", '0');   // A/B knob: 0 = tiled kernels only
  if (off) return CDS_ZMG_UNSUPPORTED;
  // Consumer waves per SIMD as measured at the M1 / cascade shapes (profiles/r04_zmarch.md): two for everything but the 16 -> 8 pair layer
  if (pair) {
    if (Cin == 8) return launch_zmg<ZG<1, 1, 1, true, 32, 8, 3, 1, 8>>(x, wsp, bias, out, Cout, D, H, W, act, st);
    if (Cin == 16) return launch_zmg<ZG<1, 2, 1, true, 32, 8, 1, 4>>(x, wsp, bias, out, Cout, D, H, W, act, st);
    if (Cin == 32) return launch_zmg<ZG<1, 4, 1, true, 32, 3, 1, 3>>(x, wsp, bias, out, Cout, D, H, W, act, st);
    return CDS_ZMG_UNSUPPORTED;
  }
  if (stride == 1) {
    if (Cin == 16 && Cout == 16) return launch_zmg<ZG<1, 2, 1, false, 32, 8, 1, 1, 8>>(x, wsp, bias, out, Cout, D, H, W, act, st);
    return CDS_ZMG_UNSUPPORTED;
  }
  if (Cin == 8 && Cout == 16) return launch_zmg<ZG<2, 1, 1, false, 16, 8, 1, 1, 8>>(x, wsp, bias, out, Cout, D, H, W, act, st);
  if (Cin == 16 && Cout == 32) return launch_zmg<ZG<2, 2, 2, false, 16, 4, 1, 1, 8>>(x, wsp, bias, out, Cout, D, H, W, act, st);
  return CDS_ZMG_UNSUPPORTED;
}
