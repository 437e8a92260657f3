// LDS-staged variants of K1 (warp-correlate-entropy) and K3 (warp-aggregate) for C = 8.
//
// The direct kernels in warp.hip gather 4 taps x 32 B per (voxel, view) through the vector L1 and are
// bound by its address/tag rate (measured 1.4-1.8 ms at M1 against 0.3-0.5 ms of HBM time).  Here a
// workgroup owns a 32x8 tile of reference pixels; for a chunk of DC consecutive depth planes the source
// footprint of the tile is a thin parallelogram along the epipolar line (~0.15 px per plane at M1), so its
// bounding box (tile + halo, ~40x10 texels = 13 KB per view) is staged once into LDS with coalesced 16-byte
// loads and reused by every plane of the chunk and every pixel of the tile; the taps are then ds_read_b128s.
//
// LDS image of a box: two planes [BH][BW] of float4 (channels 0-3 and 4-7) so that consecutive lanes, which
// sample (nearly) consecutive texels, read consecutive 16-byte slots (conflict-free ds_read_b128).
// Robustness: the box is the min/max over every lane's cells at the chunk's smallest/largest depth; any tap that still
// falls outside it (non-monotone hypotheses, projective pole) and any box larger than the LDS budget take the
// original global-memory path per lane / per block, so results never depend on the geometry assumptions.
// Arithmetic and operation order are identical to warp.hip (same cds_taps / cds_interp).
#include <stdlib.h>

#include <type_traits>

#include "warp_common.hpp"

namespace {

// Tile and chunk geometry (A/B through scripts/build_variant.sh).  A 32x8 tile has a smaller source footprint than 64x4
// for the same 256 pixels ((34 + parallax) x 10 against (66 + parallax) x 6 texels), so fewer chunks are halved and less
// is staged: K3 at M1 1.07 -> 0.96 ms, cascade stage 2 0.60 -> 0.55 ms; 48-plane chunks make the D = 48 stage a single
// chunk (stage 1 of the 1600x1184 cascade: 0.77 -> 0.48 ms) and change nothing at D = 192; 64-plane chunks overflow the
// box too often (1.19 ms at M1); 16x16 tiles and 3 waves per SIMD (spills) are slower.  Box budget 632 texels per view
// (4 views: 79 KB, two workgroups per CU still fit the 160 KB): fewer halved chunks than with 504 (0.96 -> 0.90-0.93 ms at M1,
// stage 1 of the cascade 0.48 -> 0.42 ms); 568 is in between, 440 clearly worse (1.12 ms).
#ifndef CDS_K3_TW
#define CDS_K3_TW 32
#define CDS_K3_TH 8
#define CDS_K3_BOX 632
#define CDS_K3_DC 48
#define CDS_K3_MINW 2
#endif
#ifndef CDS_PROBE_K3
#define CDS_PROBE_K3 0
#endif
constexpr int C8 = 8;
constexpr int TW = CDS_K3_TW, TH = CDS_K3_TH;  // reference-pixel tile of a workgroup (TW*TH = 256)
constexpr int BOX_CAP = CDS_K3_BOX;   // texels per view box (x 32 B; 4 views + scratch must fit the LDS budget)
constexpr int DC = CDS_K3_DC;         // depth planes per staged chunk
#ifndef CDS_K1_DC
#define CDS_K1_DC 64
#define CDS_K1_BOX 1016
#endif
constexpr int DC1 = CDS_K1_DC;        // K1 stages one view per workgroup: longer chunks, bigger box budget
constexpr int BOX1 = CDS_K1_BOX;

struct Box {
  int x0, y0, bw, bh;  // origin, width, height in texels (block-uniform)
  bool staged;
};
// What the branch-free path keeps live in the plane loop: cells (x0f, y0f) with fx0 <= x0f <= fx1 and
// fy0 <= y0f <= fy1 have all four texels in the box; org = y0 * bw + x0 is the linear index of the box origin.
// The cell is first clamped to [-2, n] (everything further out is zero padding, like the border cells), then into the
// box; the second clamp must be a no-op for the fast path to be valid.
struct FastBox {
  float fx0, fx1, fy0, fy1;   // cells of the box whose four texels are all staged
  float bwf, orgf;            // row pitch and linear index of the box origin as floats (exact: both < 2^24)
  int bw, org;
};
__device__ __forceinline__ FastBox fast_box(const Box& b, int h, int w) {
  FastBox f;
  f.fx0 = (float)b.x0;
  f.fx1 = (float)(b.x0 + b.bw - 2);
  f.fy0 = (float)b.y0;
  f.fy1 = (float)(b.y0 + b.bh - 2);
  (void)h; (void)w;
  f.bw = b.bw;
  f.bwf = (float)b.bw;
  f.org = b.y0 * b.bw + b.x0;
  f.orgf = (float)f.org;
  return f;
}
// The generic path re-reads the box from LDS (int[4] per view, written by reduce_boxes) instead of pinning SGPRs.
__device__ __forceinline__ Box load_box(const int* p) {
  Box b;
  b.x0 = p[0]; b.y0 = p[1]; b.bw = p[2]; b.bh = p[3];
  b.staged = b.bw > 0;
  return b;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// integer cell (floor of the sample position) clamped to [-2, n]; NaN -> -2 (never in the image)
__device__ __forceinline__ void cell_of(const float r[3], const float* __restrict__ t, float d, int h, int w,
                                        float half_w, float half_h, int& cx, int& cy) {
  float px = r[0] * d + t[0];
  float py = r[1] * d + t[1];
  float pz = r[2] * d + t[2];
  float z = pz + 1e-6f;
  float ix = ((px / z) / half_w - 1.0f + 1.0f) * half_w;
  float iy = ((py / z) / half_h - 1.0f + 1.0f) * half_h;
  float fx = floorf(ix), fy = floorf(iy);
  fx = (fx >= -2.0f) ? fx : -2.0f;  // also catches NaN
  fy = (fy >= -2.0f) ? fy : -2.0f;
  fx = fminf(fx, (float)w);
  fy = fminf(fy, (float)h);
  cx = (int)fx;
  cy = (int)fy;
}

// Smallest and largest hypothesis of a chunk for this pixel.  The sample position is a Moebius function of the depth,
// so every plane of the chunk lands between the positions of these two (up to fp32 rounding and a projective pole
// inside the interval, both caught by the fast path's acceptance test); first/last plane alone is not enough because
// per-pixel hypotheses need not be monotone.
template <int NMAX>
__device__ __forceinline__ void chunk_depth_range(const float* __restrict__ hyp, unsigned hw, unsigned pix, int d0, int d1,
                                                  float& dlo, float& dhi) {
  dlo = INFINITY;
  dhi = -INFINITY;
  const float* p = hyp + (size_t)d0 * hw + pix;
#ifdef CDS_RANGE_SERIAL   // the rolled loop of rounds 2-6 (A/B: scripts/ab/r06_k3_preamble_ab.sh)
#pragma unroll 8
  for (int d = d0; d < d1; ++d, p += hw) {
    const float v = *p;
    dlo = fminf(dlo, v);
    dhi = fmaxf(dhi, v);
  }
#else
  // The planes of the chunk in flight together, 24 per batch (8 for a chunk of at most 8 planes: the D = 8 stage): plane indices are
  // clamped to the chunk's last plane, so the loads are unconditional and a repeated plane changes neither extreme.  The rolled loop
  // (eight loads, then a wait, per trip) was six serial memory round trips per workgroup and 48-plane chunk in front of the staging:
  // K3 -3 % at M1, -3.4 / -1.3 / -2.7 % at the 1600x1184 cascade's stage shapes (profiles/r06_experiments.md).
  const int nlast = d1 - d0 - 1;
  auto batch = [&](int k0, auto nb) {
    constexpr int NB = decltype(nb)::value;
    float v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = p[(size_t)min(k0 + k, nlast) * hw];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      dlo = fminf(dlo, v[k]);
      dhi = fmaxf(dhi, v[k]);
    }
  };
  if (nlast < 8) {             // block-uniform
    batch(0, std::integral_constant<int, 8>());
    return;
  }
#pragma unroll
  for (int k0 = 0; k0 < NMAX; k0 += 24) {
    batch(k0, std::integral_constant<int, 24>());
    if (k0 + 24 > nlast) break;   // block-uniform: the rest would repeat the last plane
  }
#endif
}

// Block-wide bounding boxes for NV views.  lo/hi: this thread's cells at the chunk's first and last plane.
// red: LDS scratch int[4 waves][NV][4].
template <int NV, int CAP>
__device__ __forceinline__ void reduce_boxes(const int cx0[NV], const int cy0[NV], const int cx1[NV], const int cy1[NV],
                                             bool active, int nv, int h, int w, int* red, Box box[NV]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int xmin = active ? min(cx0[v], cx1[v]) : 0x7fffffff;
    int xmax = active ? max(cx0[v], cx1[v]) : -0x7fffffff;
    int ymin = active ? min(cy0[v], cy1[v]) : 0x7fffffff;
    int ymax = active ? max(cy0[v], cy1[v]) : -0x7fffffff;
    xmin = wave_min(xmin);
    xmax = wave_max(xmax);
    ymin = wave_min(ymin);
    ymax = wave_max(ymax);
    if (lane == 0) {
      int* p = red + (wave * NV + v) * 4;
      p[0] = xmin;
      p[1] = xmax;
      p[2] = ymin;
      p[3] = ymax;
    }
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int xmin = 0x7fffffff, xmax = -0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const int* p = red + (wv * NV + v) * 4;
      xmin = min(xmin, p[0]);
      xmax = max(xmax, p[1]);
      ymin = min(ymin, p[2]);
      ymax = max(ymax, p[3]);
    }
    // taps touch cells [xmin, xmax+1] x [ymin, ymax+1] (cells are clamped to [-2, n], so the box reaches at most
    // two texels outside the image; that border is staged as zeros = grid_sample's zero padding)
    const int bw = xmax + 2 - xmin, bh = ymax + 2 - ymin;
    const bool ok = (v < nv) && xmax >= xmin && ymax >= ymin && bw * bh <= CAP;
    // not staged: origin far away and limits 0, so no cell ever passes the containment test
    box[v].x0 = __builtin_amdgcn_readfirstlane(ok ? xmin : -0x40000000);
    box[v].y0 = __builtin_amdgcn_readfirstlane(ok ? ymin : -0x40000000);
    box[v].bw = __builtin_amdgcn_readfirstlane(ok ? bw : 0);
    box[v].bh = __builtin_amdgcn_readfirstlane(ok ? bh : 0);
    box[v].staged = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    if (threadIdx.x == 0) {  // copy for the generic path (visible after the staging barrier)
      int* q = red + 4 * NV * 4 + v * 4;
      q[0] = box[v].x0; q[1] = box[v].y0; q[2] = box[v].bw; q[3] = box[v].bh;
    }
  }
}

// Cooperative copy of one box into its two LDS planes, zero outside the image.
// srcv points at the first of the 8 channels staged; cs = channels per texel of the source image (8, 16 or 32).
// All loads of a thread are issued back to back from clamped addresses (zeros selected afterwards) and land in LDS together: the
// row / lane loops with a guarded load this replaces compiled to ONE global_load_dwordx4 + s_waitcnt vmcnt(0) per iteration, ~10 serial
// memory round trips per wave, view and chunk (scripts/isa_scan.py; profiles/r05_k3_staging.md).
template <int CAP>
__device__ __forceinline__ void stage_box(const float* __restrict__ srcv, int cs, int h, int w, const Box& b,
                                          float4* __restrict__ dst) {
  if (!b.staged) return;
  constexpr int KMAX = (2 * CAP + 255) / 256;
  const int n4 = 2 * b.bw;  // float4 per row
  const int total = b.bh * n4;
  const float rn4 = 1.0f / (float)n4;
  float4 val[KMAX];
  int slot[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int idx = (int)threadIdx.x + 256 * k;
    const int row = (int)(((float)idx + 0.5f) * rn4);       // idx / n4 (exact: idx < 2^12, the quotient is >= 0.5 / n4 from an integer)
    const int col = idx - row * n4;
    const int gy = b.y0 + row, gx = b.x0 + (col >> 1);
    const bool in = idx < total;
    const bool ok = in && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
    const ptrdiff_t off = ok ? ((ptrdiff_t)gy * w + gx) * cs + (col & 1) * 4 : 0;
    const float4 v = *reinterpret_cast<const float4*>(srcv + off);
    val[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    slot[k] = in ? ((col & 1) ? CAP : 0) + row * b.bw + (col >> 1) : -1;
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (slot[k] >= 0) dst[slot[k]] = val[k];
}

// Two-wide float arithmetic (two planes / two channels per value).  Written for packed fp32 (v_pk_fma_f32 ...); since round 6 the
// library is compiled WITHOUT packed-fp32 instructions (Makefile: NOPK - they compute wrong lanes beside other waves' 16x16x32 MFMAs,
// profiles/r06_packed_fp32_hazard.md), so each v2f operation is two scalar instructions with the same per-element IEEE result.  The
// two-wide form stays: it fixes the operation order of the parity tests, and the plane-pair loop structure is what was tuned.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }

// a / c for a constant c with rc = RN(1/c): one correction gives the correctly rounded quotient.
__device__ __forceinline__ v2f div2_const(v2f a, float c, float rc) {
  v2f q = a * rc;
  v2f r = fma2(splat2(-c), q, a);
  return fma2(r, splat2(rc), q);
}

struct Geo {       // per launch constants
  int h, w;
  float half_w, half_h, rhw, rhh;
};

// Sample positions of two consecutive planes.  FAST = false (the DEFAULT of the product path, ops.WARP_EXACT): the reference's fp32
// operation order (= cds_taps: correctly rounded divisions by z + 1e-6 and by (w-1)/2, ATen's normalise / de-normalise round trip;
// sample positions bit-identical to F.grid_sample's).  FAST = true (per-call flag CDS_WARP_FAST_POSITIONS, opt-in): sample at
// (u, v) = p.xy * v_rcp(z) directly; positions then differ by up to ~1e-4 px at w = 640, which moves the volume of sharp feature
// maps by 8.4e-5 against the oracle - 8x the 1e-5 parity tolerance (test_fast_positions_leave_the_parity_tolerance_at_full_width;
// exact mode 1.2e-7) - while the depth mean-L1 is unaffected.
template <bool FAST>
__device__ __forceinline__ void positions2(const float r[3], const float* __restrict__ t, v2f d, const Geo& g, v2f& ix,
                                           v2f& iy) {
  const v2f px = r[0] * d + t[0];
  const v2f py = r[1] * d + t[1];
  const v2f pz = r[2] * d + t[2];
  const v2f z = pz + 1e-6f;
  v2f y0;
  y0.x = __builtin_amdgcn_rcpf(z.x);
  y0.y = __builtin_amdgcn_rcpf(z.y);
  if (FAST) {
    ix = px * y0;
    iy = py * y0;
    return;
  }
  const v2f e = fma2(-z, y0, splat2(1.0f));
  const v2f y = fma2(e, y0, y0);
  // u = px / z and v = py / z, written interleaved: two independent dependency chains.  y is the Newton-refined reciprocal
  // (correctly rounded for all but ~1e-6 of the operands), followed by ONE Markstein correction of the quotient.  That is only
  // GUARANTEED to be the correctly rounded quotient when y = RN(1/z); for the rare mis-rounded y it is an empirical statement: 0
  // mismatches against IEEE division over 12 M random operand pairs incl. +-1 ulp errors of v_rcp_f32 (profiles/r04_experiments.md;
  // the bit-equality tests against F.grid_sample's positions are unchanged).  A miss would be one ulp of a position (6e-8 relative):
  // it changes a sample only if it flips floor() exactly at a texel boundary, by a weight of that size.  Round 3 spent a
  // second correction: 4 more packed instructions per view and plane pair, K3 0.947 -> 0.929 ms, K1 0.790 -> 0.770 ms without it.
  const v2f qu = px * y, qv = py * y;
  const v2f ru = fma2(-z, qu, px), rv = fma2(-z, qv, py);
  const v2f u = fma2(ru, y, qu), v = fma2(rv, y, qv);
  // g = u / half - 1 (div2_const), interleaved as well
  v2f qx = u * g.rhw, qy = v * g.rhh;
  const v2f rx = fma2(splat2(-g.half_w), qx, u), ry = fma2(splat2(-g.half_h), qy, v);
  qx = fma2(rx, splat2(g.rhw), qx); qy = fma2(ry, splat2(g.rhh), qy);
  const v2f gx = qx - 1.0f, gy = qy - 1.0f;
  ix = (gx + 1.0f) * g.half_w;
  iy = (gy + 1.0f) * g.half_h;
}

// v_med3_i32 with an inline constant and one SGPR (constant-bus limit of gfx9 VOP3 = 1)
__device__ __forceinline__ int clamp_m2(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, -2, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}

struct Tex8 {
  cds_f4 lo, hi;
};

// The 2x2 texel cell of one (plane, view).  Fast path: the cell, clamped to [-2, n] (everything further out is
// zero padding anyway), lies inside the staged zero-bordered box -> four 32-byte texels from LDS, no per-tap
// validity logic.  Slow path (box not staged, or a cell outside the first/last-plane bounding box): global gathers
// with per-tap image tests.  Returns the weights to use (zeroed for out-of-image taps on the slow path).
template <int CAP>
__device__ __forceinline__ void fetch_cell(float x0f, float y0f, const Geo& g, const Box& b,
                                           const cds_f4* __restrict__ lds, const float* __restrict__ srcv, int cs,
                                           Tex8 t[4], float wgt[4]) {
  const int x0 = clamp_m2((int)x0f, g.w);  // v_cvt_i32_f32 saturates, NaN -> 0 (NaN weights then propagate)
  const int y0 = clamp_m2((int)y0f, g.h);
  const unsigned ux = (unsigned)(x0 - b.x0), uy = (unsigned)(y0 - b.y0);
  const bool inside = (ux + 2u <= (unsigned)b.bw) && (uy + 2u <= (unsigned)b.bh);  // bw = bh = 0 when not staged
  // LDS reads are unconditional (index 0 when the cell is not inside: the values are then replaced below); keeping
  // them out of an if/else stops the compiler from merging the LDS and global reads into slow generic flat loads.
  const unsigned i0 = inside ? __umul24(uy, (unsigned)b.bw) + ux : 0u;
  const cds_f4* r0 = lds + i0;
  const cds_f4* r1 = r0 + (inside ? b.bw : 0);
  t[0].lo = r0[0]; t[0].hi = r0[CAP];
  t[1].lo = r0[1]; t[1].hi = r0[CAP + 1];
  t[2].lo = r1[0]; t[2].hi = r1[CAP];
  t[3].lo = r1[1]; t[3].hi = r1[CAP + 1];
  if (__builtin_expect(!inside, 0)) {
    const bool x0ok = (unsigned)x0 < (unsigned)g.w, x1ok = (unsigned)(x0 + 1) < (unsigned)g.w;
    const bool y0ok = (unsigned)y0 < (unsigned)g.h, y1ok = (unsigned)(y0 + 1) < (unsigned)g.h;
    wgt[0] = (x0ok && y0ok) ? wgt[0] : 0.f;
    wgt[1] = (x1ok && y0ok) ? wgt[1] : 0.f;
    wgt[2] = (x0ok && y1ok) ? wgt[2] : 0.f;
    wgt[3] = (x1ok && y1ok) ? wgt[3] : 0.f;
    const int xa = min(max(x0, 0), g.w - 1), xb = min(max(x0 + 1, 0), g.w - 1);
    const int ya = min(max(y0, 0), g.h - 1), yb = min(max(y0 + 1, 0), g.h - 1);
    const cds_f4* p;
    p = reinterpret_cast<const cds_f4*>(srcv + ((size_t)ya * g.w + xa) * cs); t[0].lo = p[0]; t[0].hi = p[1];
    p = reinterpret_cast<const cds_f4*>(srcv + ((size_t)ya * g.w + xb) * cs); t[1].lo = p[0]; t[1].hi = p[1];
    p = reinterpret_cast<const cds_f4*>(srcv + ((size_t)yb * g.w + xa) * cs); t[2].lo = p[0]; t[2].hi = p[1];
    p = reinterpret_cast<const cds_f4*>(srcv + ((size_t)yb * g.w + xb) * cs); t[3].lo = p[0]; t[3].hi = p[1];
  }
}

// Branch-free variant for the common case: the cell is clamped into the staged box in the float domain (so the LDS
// address is always valid) and the caller is told whether the clamp changed anything.  If any lane of the wave
// reports a changed cell (or a box is not staged) the caller redoes the plane pair with fetch_cell.
// FIDX: linear texel index computed in the float domain (yc * bw + xc - org is an integer below 2^24: exact; one conversion
// instead of two conversions + an integer multiply-add).  Same-box A/B at M1: K1 0.79 vs 0.815 ms with it, K3 0.985-1.008 vs
// 0.956-0.962 ms (the longer dependent chain in front of the LDS reads costs K3 more than the two instructions it saves), so
// K1 uses it and K3 does not.
template <int CAP, bool FIDX>
__device__ __forceinline__ bool cell_addr_fast(float x0f, float y0f, float wf, float hf, const FastBox& b,
                                               const cds_f4* __restrict__ lds, const cds_f4*& r0, const cds_f4*& r1) {
  const float xa = __builtin_amdgcn_fmed3f(x0f, -2.0f, wf);
  const float ya = __builtin_amdgcn_fmed3f(y0f, -2.0f, hf);
  const float xc = __builtin_amdgcn_fmed3f(xa, b.fx0, b.fx1);
  const float yc = __builtin_amdgcn_fmed3f(ya, b.fy0, b.fy1);
#ifdef CDS_EXP_LDS_BCAST
  const int idx = 0 * ((int)yc + (int)xc);
#else
  const int idx = FIDX ? (int)fmaf(yc, b.bwf, xc - b.orgf) : __mul24((int)yc, b.bw) + (int)xc - b.org;
#endif
  r0 = lds + idx;
  r1 = r0 + b.bw;
  return (xc == xa) & (yc == ya);
}
template <int CAP>
__device__ __forceinline__ void load_cell(const cds_f4* r0, const cds_f4* r1, Tex8 t[4]) {
  t[0].lo = r0[0]; t[0].hi = r0[CAP];
  t[1].lo = r0[1]; t[1].hi = r0[CAP + 1];
  t[2].lo = r1[0]; t[2].hi = r1[CAP];
  t[3].lo = r1[1]; t[3].hi = r1[CAP + 1];
}

// bilinear interpolation of the 8 channels as four channel pairs (v_pk_mul / v_pk_fma), cds_interp's operation order
__device__ __forceinline__ void interp8(const Tex8 t[4], const float wgt[4], v2f o[4]) {
  const v2f w0 = splat2(wgt[0]), w1 = splat2(wgt[1]), w2 = splat2(wgt[2]), w3 = splat2(wgt[3]);
#define CDS_PAIR(j, F, A, B)                              \
  o[j] = (v2f){t[0].F.A, t[0].F.B} * w0;                  \
  o[j] = fma2((v2f){t[1].F.A, t[1].F.B}, w1, o[j]);       \
  o[j] = fma2((v2f){t[2].F.A, t[2].F.B}, w2, o[j]);       \
  o[j] = fma2((v2f){t[3].F.A, t[3].F.B}, w3, o[j]);
  CDS_PAIR(0, lo, x, y)
  CDS_PAIR(1, lo, z, w)
  CDS_PAIR(2, hi, x, y)
  CDS_PAIR(3, hi, z, w)
#undef CDS_PAIR
}

__device__ __forceinline__ void plane_weights(v2f ix, v2f iy, v2f& x0f, v2f& y0f, v2f w[4]) {
  x0f.x = floorf(ix.x); x0f.y = floorf(ix.y);
  y0f.x = floorf(iy.x); y0f.y = floorf(iy.y);
  const v2f wx = ix - x0f, ex = 1.0f - wx;
  const v2f ny = iy - y0f, sy = 1.0f - ny;
  w[0] = sy * ex;
  w[1] = sy * wx;
  w[2] = ny * ex;
  w[3] = ny * wx;
}

// ---------------------------------------------------------------------------------------------
// K3 with LDS-staged boxes, 8 channels per workgroup, V <= 4 (all views of a chunk resident: 4 x 19.75 KB).
// Two planes per iteration so the position / weight arithmetic issues as packed fp32 (v_pk_*).
// Accumulation: volume += (ref*vis) * warp as one fma per channel (re-association of the reference's
// (ref*warp)*vis, <= 2 ulp of a value below 1).  Normalisation a/(vis_sum+1e-6): reciprocal refined once per
// pixel and folded into the (ref*vis) factors.
// Addressing: per-channel slab base (uniform) + one 32-bit byte offset per plane (slab = D*h*w*4 < 4 GB).
// ---------------------------------------------------------------------------------------------
template <int VMAX, bool ACCUMULATE, bool NORMALIZE, bool FAST, int CAP>
__global__ __launch_bounds__(256, CDS_K3_MINW) void warp_aggregate_lds_kernel(
    const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ vis, const float* __restrict__ mats_d,
    const float* __restrict__ hyp, float* __restrict__ volume, const float* __restrict__ vis_sum, int C, int D, int h,
    int w, float rhw, float rhh, int flags, int tiles_x, int ntiles, int nseg, int seg_planes, int hs, int y_off) {
  // h x w: the reference-side grid of this call (features, hypotheses, weights, volume); it is the window of rows
  // [y_off, y_off + h) of the full image grid hs x w that the SOURCE feature maps cover (hs == h, y_off == 0: the whole grid)
  extern __shared__ __attribute__((aligned(16))) cds_f4 lds4[];  // VMAX * 2*CAP float4, then int red[4*VMAX*4], int boxes[VMAX*4]
  int* red = reinterpret_cast<int*>(lds4 + VMAX * 2 * CAP);
  const MatRegs<VMAX> mats(mats_d);

  // depth segment is the fastest-varying index: the nseg blocks of a tile run together and share its features in L2
  // then the group of 8 channels (C = 16 / 32: one workgroup per group; the groups of a tile run together, so the
  // 64 / 128-byte texels they share are fetched from HBM once)
  const int ngroups = C >> 3;
  int lin = cds_xcd_remap(blockIdx.x, ntiles * ngroups * nseg);
  const int seg = lin % nseg;
  lin /= nseg;
  const int c_off = (lin % ngroups) * C8;
  const int tile = lin / ngroups;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * TW + (threadIdx.x % TW);
  const int y = ty * TH + (threadIdx.x / TW);
  const bool active = x < w && y < h;
  const int xc = min(x, w - 1), yc = min(y, h - 1);  // inactive lanes shadow a valid pixel, never store
  Geo g;
  g.h = hs; g.w = w;
  g.half_w = (float)((w - 1) / 2.0); g.half_h = (float)((hs - 1) / 2.0);
  g.rhw = rhw; g.rhh = rhh;
  const unsigned hw = (unsigned)h * (unsigned)w;
  const size_t hws = (size_t)hs * w;                 // pixels of a source feature map
  const unsigned pix = (unsigned)yc * (unsigned)w + (unsigned)xc;
  const size_t slab = (size_t)D * hw;  // elements per channel of the volume
  ref += (size_t)c_off * hw;
  src += c_off;
  volume += (size_t)c_off * slab;

  v2f rv[VMAX][4];  // (ref * vis) per channel pair
  float r[VMAX][3];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    {
      const float vw = vis[(size_t)v * hw + pix];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rv[v][j].x = ref[((size_t)v * C + 2 * j) * hw + pix] * vw;
        rv[v][j].y = ref[((size_t)v * C + 2 * j + 1) * hw + pix] * vw;
      }
      cds_row_terms(mats.m[v], (float)xc, (float)(yc + y_off), r[v]);
    }
  }
  constexpr bool accumulate = ACCUMULATE, normalize = NORMALIZE;
  const bool cl = flags & CDS_AGG_CHANNELS_LAST;   // volume [D][h][w][C]: a voxel's 8 channels of this group are 32 contiguous bytes
  float* vol_cl = volume - (size_t)c_off * slab + c_off;   // (the planar base was advanced by the group's slabs above)
  // Normalisation volume_sum / (vis_sum + 1e-6) is folded into the per-pixel factors: (ref * vis) * RN(1/denom)
  // (a re-association of the reference's division, a few ulp of a value below 1; no per-plane work left).
  const float yden = normalize ? 1.0f / (vis_sum[pix] + 1e-6f) : 1.0f;
  if (normalize) {
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[v][j] = rv[v][j] * yden;
  }
  const char* hyp_b = reinterpret_cast<const char*>(hyp);
  const float wf = (float)w, hf = (float)hs;

  const int dseg0 = seg * seg_planes, dseg1 = min(D, dseg0 + seg_planes);
  for (int d0 = dseg0, d1 = 0; d0 < dseg1; d0 = d1) {
    d1 = min(dseg1, d0 + DC);
    Box box[VMAX];
    // Adaptive chunk length: if some view's footprint over DC planes does not fit its LDS budget (~10 % of the
    // tiles at M1), halve the chunk instead of sending the whole tile down the global-memory path.
    for (;;) {
      int cx0[VMAX], cy0[VMAX], cx1[VMAX], cy1[VMAX];
      float dfirst, dlast;
      chunk_depth_range<DC>(hyp, hw, pix, d0, d1, dfirst, dlast);
#pragma unroll
      for (int v = 0; v < VMAX; ++v) {
        cell_of(r[v], mats.m[v] + 9, dfirst, hs, w, g.half_w, g.half_h, cx0[v], cy0[v]);
        cell_of(r[v], mats.m[v] + 9, dlast, hs, w, g.half_w, g.half_h, cx1[v], cy1[v]);
      }
      __syncthreads();  // previous chunk's LDS reads are done (also protects `red`)
      reduce_boxes<VMAX, CAP>(cx0, cy0, cx1, cy1, active, VMAX, hs, w, red, box);
      bool fits = true;
#pragma unroll
      for (int v = 0; v < VMAX; ++v) fits = fits && box[v].staged;
      if (fits || d1 - d0 <= 8) break;
      d1 = d0 + ((((d1 - d0) >> 1) + 1) & ~1);  // even length: plane pairs stay whole
    }
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      stage_box<CAP>(src + v * hws * C, C, hs, w, box[v], reinterpret_cast<float4*>(lds4 + v * 2 * CAP));
    __syncthreads();
    bool all_staged = true;
    FastBox fb[VMAX];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      fb[v] = fast_box(box[v], hs, w);
      all_staged = all_staged && box[v].staged;
    }
    const int* boxmem = red + 4 * VMAX * 4;

    unsigned boff = ((unsigned)d0 * hw + pix) * 4u;  // byte offset of (plane d, pixel) inside a channel slab / hyp
    const unsigned bstep = hw * 4u;
    // hypotheses of the next plane pair are loaded one iteration ahead; plane indices are clamped to the chunk's last
    // plane so the loads are unconditional (no branch, no early wait)
    const unsigned blast = ((unsigned)(d1 - 1) * hw + pix) * 4u;
    v2f dnext;
    dnext.x = *reinterpret_cast<const float*>(hyp_b + boff);
    dnext.y = *reinterpret_cast<const float*>(hyp_b + min(boff + bstep, blast));
    asm volatile("" ::"v"(dnext.x), "v"(dnext.y));   // delivered before the loop: the loop head then joins two states without pending loads
    // ACCUMULATE (the second launch of a view list longer than four: BASELINE config 4): the partial sums of a plane pair are read
    // one iteration AHEAD, like the hypotheses.  Read at the top of their own iteration they were a full memory round trip in front
    // of the first view's accumulation at two waves per SIMD (1920x1056, N = 7: the accumulating launch 1.76x the first one).
    // Plane indices are clamped to the chunk's last plane: unconditional loads, the clamped repeats are never used.
#ifndef CDS_K3_ACC_SERIAL
    constexpr bool acc_ahead = ACCUMULATE;
#else
    constexpr bool acc_ahead = false;
#endif
    v2f accn[2][4];
    auto load_partial = [&](int dp, v2f out[2][4]) {     // planes dp, dp + 1 (clamped) of this pixel
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned dk = (unsigned)min(dp + k, d1 - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (cl) {
            const float* pv = vol_cl + ((size_t)dk * hw + pix) * C + 2 * j;
            out[k][j].x = pv[0];
            out[k][j].y = pv[1];
          } else {
            out[k][j].x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(volume + (size_t)(2 * j) * slab) + (dk * hw + pix) * 4u);
            out[k][j].y = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(volume + (size_t)(2 * j + 1) * slab) + (dk * hw + pix) * 4u);
          }
        }
      }
    };
    if (acc_ahead) load_partial(d0, accn);
    for (int d = d0; d < d1; d += 2, boff += 2u * bstep) {
      const bool two = d + 1 < d1;
#ifdef CDS_PROBE_K3_POS
      const v2f dv = dnext * 0.0f + 600.0f;      // probe: loop-invariant positions
#else
      const v2f dv = dnext;
#endif
      dnext.x = *reinterpret_cast<const float*>(hyp_b + min(boff + 2u * bstep, blast));
      dnext.y = *reinterpret_cast<const float*>(hyp_b + min(boff + 3u * bstep, blast));
      v2f acc[2][4];
      if (acc_ahead) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[k][j] = (k == 0 || two) ? accn[k][j] : splat2(0.f);
            if (normalize) acc[k][j] = acc[k][j] * yden;  // partial sums of an earlier launch
          }
        load_partial(d + 2, accn);
      }
      auto init_acc = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[k][j] = splat2(0.f);
            if (accumulate && (k == 0 || two)) {
              if (cl) {
                const float* pv = vol_cl + ((size_t)(d + k) * hw + pix) * C + 2 * j;
                acc[k][j].x = pv[0];
                acc[k][j].y = pv[1];
              } else {
                acc[k][j].x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(volume + (size_t)(2 * j) * slab) + boff + k * bstep);
                acc[k][j].y = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(volume + (size_t)(2 * j + 1) * slab) + boff + k * bstep);
              }
              if (normalize) acc[k][j] = acc[k][j] * yden;  // partial sums of an earlier launch
            }
          }
      };
      if (!acc_ahead) init_acc();
      bool ok = all_staged;
      if (all_staged) {  // block-uniform
        // Software pipeline over the views: the LDS reads of view v+1 are issued while view v is interpolated, and
        // the position / address arithmetic of view v+1 runs while the reads of view v are in flight.  The
        // sched_barriers pin that order (left alone, the scheduler hoists every read to the top and spills).
        v2f wt[2][4];
        const cds_f4 *p0[2], *p1[2];
        Tex8 tc[2][4];
        auto prep = [&](int v, v2f w4[4]) {
          v2f ix, iy, x0f, y0f;
          positions2<FAST>(r[v], mats.m[v] + 9, dv, g, ix, iy);
          plane_weights(ix, iy, x0f, y0f, w4);
          const cds_f4* lv = lds4 + v * 2 * CAP;
          ok &= cell_addr_fast<CAP, false>(x0f.x, y0f.x, wf, hf, fb[v], lv, p0[0], p1[0]);
          ok &= cell_addr_fast<CAP, false>(x0f.y, y0f.y, wf, hf, fb[v], lv, p0[1], p1[1]);
        };
        prep(0, wt[0]);
        // Probe builds (wrong results, right timing; scripts/ab/r06_k3_probe.sh, profiles/r06_k3_probe.md): CDS_PROBE_K3 = 1: the
        // second plane of a pair reuses the first plane's taps (half the ds_reads: what tap reuse along depth could save at best);
        // 2: no ds_read in the plane loop at all (the VALU-only floor of the loop); CDS_PROBE_K3_POS: (nearly) loop-invariant
        // sample positions (the position arithmetic may be hoisted: the LDS-read side on its own).
#if CDS_PROBE_K3 == 2
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 4; ++q) { tc[k][q].lo = (cds_f4){rv[0][0].x, rv[0][1].x, rv[0][2].x, rv[0][3].x}; tc[k][q].hi = tc[k][q].lo * dv.x; }
#else
        load_cell<CAP>(p0[0], p1[0], tc[0]);
        if (CDS_PROBE_K3 != 1) load_cell<CAP>(p0[1], p1[1], tc[1]);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < VMAX; ++v) {
          const int cur = v & 1, nxt = cur ^ 1;
          if (v + 1 < VMAX) prep(v + 1, wt[nxt]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float wgt[4] = {k ? wt[cur][0].y : wt[cur][0].x, k ? wt[cur][1].y : wt[cur][1].x,
                                  k ? wt[cur][2].y : wt[cur][2].x, k ? wt[cur][3].y : wt[cur][3].x};
            v2f o[4];
            interp8(tc[CDS_PROBE_K3 == 1 ? 0 : k], wgt, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = fma2(rv[v][j], o[j], acc[k][j]);
            if (v + 1 < VMAX && CDS_PROBE_K3 != 2 && !(CDS_PROBE_K3 == 1 && k == 1)) load_cell<CAP>(p0[k], p1[k], tc[k]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#ifndef CDS_K3_NOREDO
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {  // wave-uniform: redo the pair, any geometry
        init_acc();
        // opaque copy of the hypotheses: keeps CSE from pinning the fast path's per-view weights for this cold block
        float dgx = dv.x, dgy = dv.y;
        asm volatile("" : "+v"(dgx), "+v"(dgy));
        const v2f dvg = {dgx, dgy};
#pragma unroll
        for (int v = 0; v < VMAX; ++v) {
          const float* __restrict__ srcv = src + v * hws * C;
          const cds_f4* lv = lds4 + v * 2 * CAP;
          const Box bg = load_box(boxmem + v * 4);
          v2f ix, iy, x0f, y0f, wt[4];
          positions2<FAST>(r[v], mats.m[v] + 9, dvg, g, ix, iy);
          plane_weights(ix, iy, x0f, y0f, wt);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            float wgt[4] = {k ? wt[0].y : wt[0].x, k ? wt[1].y : wt[1].x, k ? wt[2].y : wt[2].x, k ? wt[3].y : wt[3].x};
            Tex8 t[4];
            fetch_cell<CAP>(k ? x0f.y : x0f.x, k ? y0f.y : y0f.x, g, bg, lv, srcv, C, t, wgt);
            v2f o[4];
            interp8(t, wgt, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = fma2(rv[v][j], o[j], acc[k][j]);
          }
        }
      }
#endif
      // Take delivery of the next pair's hypotheses HERE, before this pair's stores are issued.  gfx9 counts loads and stores in one
      // vmcnt and they complete out of order with respect to each other, so a wait for a load with younger stores in flight has to be
      // vmcnt(0): left at its first use (the top of the next iteration) it drained the stores just issued, every iteration.  At this
      // point only the previous iteration's stores are outstanding and they have had a whole iteration to complete.
      asm volatile("" ::"v"(dnext.x), "v"(dnext.y));
      if (active) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (cl) {
            if (k == 0 || two) {   // one voxel = 8 channels = two 16-byte stores on 32 contiguous bytes
              float4* dst = reinterpret_cast<float4*>(vol_cl + ((size_t)(d + k) * hw + pix) * C);
              dst[0] = make_float4(acc[k][0].x, acc[k][0].y, acc[k][1].x, acc[k][1].y);
              dst[1] = make_float4(acc[k][2].x, acc[k][2].y, acc[k][3].x, acc[k][3].y);
            }
          } else if (k == 0 || two) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v2f o = acc[k][j];
              float* p0 = reinterpret_cast<float*>(reinterpret_cast<char*>(volume + (size_t)(2 * j) * slab) + boff + k * bstep);
              float* p1 = reinterpret_cast<float*>(reinterpret_cast<char*>(volume + (size_t)(2 * j + 1) * slab) + boff + k * bstep);
#ifdef CDS_EXP_PLAIN_STORE
              *p0 = o.x;
              *p1 = o.y;
#else
              __builtin_nontemporal_store(o.x, p0);
              __builtin_nontemporal_store(o.y, p1);
#endif
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 with an LDS-staged box, C = 8, one (tile, view) per workgroup, two planes per iteration
// ---------------------------------------------------------------------------------------------
// Softmax-entropy statistics with a lazily updated reference m: Z = sum e^(s-m), T = sum (s-m) e^(s-m).
// m starts at the first plane's score and is moved when a score exceeds it by more than CDS_K1_LAZY.  The final
// log Z - T/Z cancels two terms of size ~(max s - m) + log D, so the threshold bounds the absolute error: with 40 (never
// triggers for |s| <= C) the C = 32 stages reached 2.1e-5 (300-case fuzz against the oracle), with 8: 1.8e-5, with 2: 5e-6 =
// the level of an exact running maximum (threshold 0), which costs 12 % of the kernel (0.90 vs 0.80 ms at M1) where 2 is free.
#ifndef CDS_K1_LAZY
#define CDS_K1_LAZY 2.0f
#endif
__device__ __forceinline__ void online_entropy_update(float s, float& mx, float& Z, float& T) {
  float dlt = s - mx;
  if (__builtin_expect(dlt > CDS_K1_LAZY, 0)) {  // also taken on the first plane (mx = -inf)
    const float sc = expf(mx - s);         // exp(-inf) = 0 on the first plane
    const float shift = (Z == 0.f) ? 0.f : (mx - s) * Z;
    T = sc * (T + shift);
    Z = Z * sc;
    mx = s;
    dlt = 0.f;
  }
  const float ev = __builtin_amdgcn_exp2f(dlt * 1.44269504088896340736f);  // v_exp_f32, ~1 ulp
  Z += ev;
  T = fmaf(dlt, ev, T);
}

#ifndef CDS_K1_MINW
#define CDS_K1_MINW 4   // 4 waves per SIMD (32.6 KB of LDS per workgroup allows it): 0.82 -> 0.79 ms at M1
#endif
// NG = groups of 8 channels (C = 8 NG); the box of a view holds all of them: 2 NG planes of CAP float4.
// C = 8: CAP 1016 texels, 64-plane chunks (32 KB);  C = 16 / 32: the K3 budget (CDS_K3_BOX texels, CDS_K3_DC planes: 40 /
// 79 KB, two workgroups per CU at C = 32).  A box that does not fit halves its chunk, as in K3.
template <int NG, int CAP, int DCK, bool FAST>
__global__ __launch_bounds__(256, (NG == 4 ? 2 : CDS_K1_MINW)) void warp_entropy_lds_kernel(
    const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ mats_d, const float* __restrict__ hyp,
    float* __restrict__ entropy, int V, int D, int h, int w, float rhw, float rhh, int tiles_x, int ntiles, int hs, int y_off) {
  // h x w = rows [y_off, y_off + h) of the full hs x w grid of the source maps (see warp_aggregate_lds_kernel)
  constexpr int C = NG * C8;
  extern __shared__ __attribute__((aligned(16))) cds_f4 lds4[];
  int* red = reinterpret_cast<int*>(lds4 + 2 * NG * CAP);
  const int lin = cds_xcd_remap(blockIdx.x, ntiles * V);
  const int v = lin % V;
  const int tile = lin / V;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * TW + (threadIdx.x % TW);
  const int y = ty * TH + (threadIdx.x / TW);
  const bool active = x < w && y < h;
  const int xc = min(x, w - 1), yc = min(y, h - 1);
  Geo g;
  g.h = hs; g.w = w;
  g.half_w = (float)((w - 1) / 2.0); g.half_h = (float)((hs - 1) / 2.0);
  g.rhw = rhw; g.rhh = rhh;
  const unsigned hw = (unsigned)h * (unsigned)w;
  const unsigned pix = (unsigned)yc * (unsigned)w + (unsigned)xc;
  const float* __restrict__ srcv = src + (size_t)v * hs * w * C;
  float m[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) m[i] = mats_d[v * 12 + i];
  v2f rf[NG][4];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rf[q][j].x = ref[((size_t)v * C + q * C8 + 2 * j) * hw + pix];
      rf[q][j].y = ref[((size_t)v * C + q * C8 + 2 * j + 1) * hw + pix];
    }
  float r[3];
  cds_row_terms(m, (float)xc, (float)(yc + y_off), r);
  const char* hyp_b = reinterpret_cast<const char*>(hyp);
  const float wf = (float)w, hf = (float)hs;
  float mx = -INFINITY, Z = 0.f, T = 0.f;
  for (int d0 = 0, d1 = 0; d0 < D; d0 = d1) {
    d1 = min(D, d0 + DCK);
    Box box[1];
    for (;;) {
      int cx0[1], cy0[1], cx1[1], cy1[1];
      float dlo, dhi;
      chunk_depth_range<DCK>(hyp, hw, pix, d0, d1, dlo, dhi);
      cell_of(r, m + 9, dlo, hs, w, g.half_w, g.half_h, cx0[0], cy0[0]);
      cell_of(r, m + 9, dhi, hs, w, g.half_w, g.half_h, cx1[0], cy1[0]);
      __syncthreads();
      reduce_boxes<1, CAP>(cx0, cy0, cx1, cy1, active, 1, hs, w, red, box);
      if (box[0].staged || d1 - d0 <= 8) break;
      d1 = d0 + ((((d1 - d0) >> 1) + 1) & ~1);  // even length: plane pairs stay whole
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) stage_box<CAP>(srcv + q * C8, C, hs, w, box[0], reinterpret_cast<float4*>(lds4 + q * 2 * CAP));
    __syncthreads();
    const FastBox fb = fast_box(box[0], hs, w);
    const bool staged = box[0].staged;
    unsigned boff = ((unsigned)d0 * hw + pix) * 4u;
    const unsigned bstep = hw * 4u;
    const unsigned blast = ((unsigned)(d1 - 1) * hw + pix) * 4u;
    v2f dnext;
    dnext.x = *reinterpret_cast<const float*>(hyp_b + boff);
    dnext.y = *reinterpret_cast<const float*>(hyp_b + min(boff + bstep, blast));
    for (int d = d0; d < d1; d += 2, boff += 2u * bstep) {
      const bool two = d + 1 < d1;
      const v2f dv = dnext;
      dnext.x = *reinterpret_cast<const float*>(hyp_b + min(boff + 2u * bstep, blast));
      dnext.y = *reinterpret_cast<const float*>(hyp_b + min(boff + 3u * bstep, blast));
      v2f ix, iy, x0f, y0f, wt[4];
      positions2<FAST>(r, m + 9, dv, g, ix, iy);
      plane_weights(ix, iy, x0f, y0f, wt);
      float sim[2];
      // sum_C ref*warp in ATen's outer-dim order: sequential inside 16-channel levels, level sums added in order
      // (C <= 16: plainly sequential).  `part` runs over the two groups of a level.
      auto correlate = [&](const Tex8 t[4], const float wgt[4], int q, float part) {
        v2f o[4];
        interp8(t, wgt, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const v2f p = rf[q][j] * o[j];
          part = part + p.x;
          part = part + p.y;
        }
        return part;
      };
      bool ok = staged;
      if (staged) {  // block-uniform
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float wgt[4] = {k ? wt[0].y : wt[0].x, k ? wt[1].y : wt[1].x, k ? wt[2].y : wt[2].x, k ? wt[3].y : wt[3].x};
          const cds_f4 *q0, *q1;
          ok &= cell_addr_fast<CAP, true>(k ? x0f.y : x0f.x, k ? y0f.y : y0f.x, wf, hf, fb, lds4, q0, q1);
          float s = 0.f, part = 0.f;
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            Tex8 t[4];
            load_cell<CAP>(q0 + q * 2 * CAP, q1 + q * 2 * CAP, t);
            part = correlate(t, wgt, q, part);
            if ((q & 1) || q == NG - 1) {
              s = s + part;
              part = 0.f;
            }
            if (NG > 1) __builtin_amdgcn_sched_barrier(0);  // one group's texels live at a time
          }
          sim[k] = s;
        }
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {  // wave-uniform: redo the pair, any geometry
        const Box bg = load_box(red + 4 * 4);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          float s = 0.f, part = 0.f;
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            float wgt[4] = {k ? wt[0].y : wt[0].x, k ? wt[1].y : wt[1].x, k ? wt[2].y : wt[2].x, k ? wt[3].y : wt[3].x};
            Tex8 t[4];
            fetch_cell<CAP>(k ? x0f.y : x0f.x, k ? y0f.y : y0f.x, g, bg, lds4 + q * 2 * CAP, srcv + q * C8, C, t, wgt);
            part = correlate(t, wgt, q, part);
            if ((q & 1) || q == NG - 1) {
              s = s + part;
              part = 0.f;
            }
            if (NG > 1) __builtin_amdgcn_sched_barrier(0);  // one group's texels live at a time
          }
          sim[k] = s;
        }
      }
      online_entropy_update(sim[0], mx, Z, T);
      if (two) online_entropy_update(sim[1], mx, Z, T);
    }
  }
  if (active) entropy[(size_t)v * hw + pix] = logf(Z) - T / Z;
}

}  // namespace

// Launchers used by the extern "C" entry points in warp.hip.  Return false if the shape is not covered.
bool cds_warp_aggregate_lds_launch(const float* ref, const float* src, const float* vis, const float* wm,
                                   const float* hyp, float* volume, const float* vis_sum, int V, int C, int D, int h,
                                   int w, int hyp_pp, int flags, hipStream_t st, int hs, int y_off) {
  if (hs <= 0) { hs = h; y_off = 0; }      // whole grid
  if ((C != 8 && C != 16 && C != 32) || V < 1 || V > CDS_MAX_VIEWS || !hyp_pp || w < 2 || hs < 2 || h < 1 || y_off < 0 ||
      y_off + h > hs || (size_t)D * h * w * 4 >= ((size_t)1 << 32))
    return false;
  // 5 / 6 source views (BASELINE config 4, N = 7) in ONE pass with smaller boxes was built and measured in round 3 at the config-4
  // stage shapes with the cascade's real hypothesis ranges: slower than two launches at every stage (1.78 vs 1.58, 2.16 vs 0.99,
  // 2.68 vs 1.44 ms: the smaller boxes overflow, the 6-view kernel spills; profiles/r03_costreg_experiments.md section 4) and removed.
  if (V > 4) {
    // more views than fit the LDS budget: two launches over halves of the view list; the second adds to the first's
    // partial sums (and normalises).  Costs one extra read of the volume, still far cheaper than L1 gathers.
    int v1 = (V + 1) / 2;
    if (const char* e = getenv("CDS_K3_SPLIT")) v1 = (atoi(e) >= 1 && atoi(e) <= 4 && atoi(e) < V && V - atoi(e) <= 4) ? atoi(e) : v1;   // A/B knob
    const size_t hw = (size_t)h * w, hws = (size_t)hs * w;
    const float* wm2 = wm + (size_t)v1 * 12;
    const int keep = flags & (CDS_AGG_CHANNELS_LAST | CDS_AGG_FAST_POSITIONS);
    return cds_warp_aggregate_lds_launch(ref, src, vis, wm, hyp, volume, vis_sum, v1, C, D, h, w, hyp_pp,
                                         (flags & CDS_AGG_ACCUMULATE) | keep, st, hs, y_off) &&
           cds_warp_aggregate_lds_launch(ref + (size_t)v1 * C * hw, src + (size_t)v1 * hws * C, vis + (size_t)v1 * hw, wm2,
                                         hyp, volume, vis_sum, V - v1, C, D, h, w, hyp_pp,
                                         CDS_AGG_ACCUMULATE | (flags & CDS_AGG_NORMALIZE) | keep, st, hs, y_off);
  }
  const int tiles_x = cds_ceil_div(w, TW), tiles_y = cds_ceil_div(h, TH);
  const int ntiles = tiles_x * tiles_y;
  const int ngroups = C / C8;
  const float rhw = (float)(1.0 / (double)(float)((w - 1) / 2.0)), rhh = (float)(1.0 / (double)(float)((hs - 1) / 2.0));
  // Depth segments: enough workgroups for ~10 waves per SIMD (2 are resident), each a whole number of DC chunks.
  const int chunks = cds_ceil_div(D, DC);
  int nseg = 1;
  while (nseg < chunks && (size_t)ntiles * ngroups * nseg * 4 < (size_t)10 * 1024) nseg *= 2;
  if (nseg > chunks) nseg = chunks;
  if (const char* e = getenv("CDS_K3_NSEG")) nseg = atoi(e) > 0 ? (atoi(e) < chunks ? atoi(e) : chunks) : nseg;  // tuning knob
  const int seg_planes = cds_ceil_div(chunks, nseg) * DC;
  nseg = cds_ceil_div(D, seg_planes);
  const bool acc_f = flags & CDS_AGG_ACCUMULATE, nrm_f = flags & CDS_AGG_NORMALIZE;
  const bool fast_f = flags & CDS_AGG_FAST_POSITIONS;
#define LAUNCH4(VM, A, N, F, CAPV)                                                                                      \
  hipLaunchKernelGGL((warp_aggregate_lds_kernel<VM, A, N, F, CAPV>), dim3(ntiles * ngroups * nseg), dim3(256),         \
                     (size_t)VM * 2 * CAPV * sizeof(float4) + 5 * VM * 4 * sizeof(int), st, ref, src, vis, wm, hyp,    \
                     volume, vis_sum, C, D, h, w, rhw, rhh, flags, tiles_x, ntiles, nseg, seg_planes, hs, y_off)
#define LAUNCH3(VM, A, N, CAPV)                    \
  do {                                             \
    if (fast_f) LAUNCH4(VM, A, N, true, CAPV);     \
    else LAUNCH4(VM, A, N, false, CAPV);           \
  } while (0)
#define LAUNCH(VM, CAPV)                                 \
  do {                                                   \
    if (acc_f && nrm_f) LAUNCH3(VM, true, true, CAPV);   \
    else if (acc_f) LAUNCH3(VM, true, false, CAPV);      \
    else if (nrm_f) LAUNCH3(VM, false, true, CAPV);      \
    else LAUNCH3(VM, false, false, CAPV);                \
  } while (0)
  switch (V) {   // the kernel is specialised on the exact view count
    case 1: LAUNCH(1, BOX_CAP); break;
    case 2: LAUNCH(2, BOX_CAP); break;
    case 3: LAUNCH(3, BOX_CAP); break;
    default: LAUNCH(4, BOX_CAP); break;
  }
#undef LAUNCH4
#undef LAUNCH3
#undef LAUNCH
  return true;
}

bool cds_warp_entropy_lds_launch(const float* ref, const float* src, const float* wm, const float* hyp,
                                 float* entropy, int V, int C, int D, int h, int w, int hyp_pp, bool fast, hipStream_t st, int hs,
                                 int y_off) {
  if (hs <= 0) { hs = h; y_off = 0; }
  if ((C != 8 && C != 16 && C != 32) || !hyp_pp || w < 2 || hs < 2 || h < 1 || y_off < 0 || y_off + h > hs ||
      (size_t)D * h * w * 4 >= ((size_t)1 << 32))
    return false;
  const int tiles_x = cds_ceil_div(w, TW), tiles_y = cds_ceil_div(h, TH);
  const int ntiles = tiles_x * tiles_y;
  const float rhw = (float)(1.0 / (double)(float)((w - 1) / 2.0)), rhh = (float)(1.0 / (double)(float)((hs - 1) / 2.0));
#define LAUNCH2(NG, CAP, DCK, F)                                                                                   \
  hipLaunchKernelGGL((warp_entropy_lds_kernel<NG, CAP, DCK, F>), dim3(ntiles * V), dim3(256),                      \
                     (size_t)2 * NG * CAP * sizeof(float4) + 5 * 4 * sizeof(int), st, ref, src, wm, hyp, entropy, V, D, \
                     h, w, rhw, rhh, tiles_x, ntiles, hs, y_off)
#define LAUNCH1(NG, CAP, DCK)                \
  do {                                       \
    if (fast) LAUNCH2(NG, CAP, DCK, true);   \
    else LAUNCH2(NG, CAP, DCK, false);       \
  } while (0)
  if (C == 8) LAUNCH1(1, BOX1, DC1);
  else if (C == 16) LAUNCH1(2, BOX_CAP, DC);
  else LAUNCH1(4, BOX_CAP, DC);
#undef LAUNCH2
#undef LAUNCH1
  return true;
}
