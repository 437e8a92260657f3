// Shared pieces of the split-bf16 (3xBF16 error-compensated) matrix-core kernels: vector types, the exact three-way split,
// the six-partial-product MFMA sequence.  See the header comment of conv3d_sbf.hip for the arithmetic.
#pragma once
#include "cds_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

union BV {
  uint4 u;
  bf16x8 v;
  f16x8 h;      // the same 16 bytes as eight fp16 values (split-f16 kernels)
};

// exact three-way split of two floats: packed (hi0,hi1), (mid0,mid1), (lo0,lo1)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
#ifdef CDS_SPLIT_SCALAR
  // A/B knob: plain v_sub_f32 instead of v_pk_add_f32 for the two residuals (packed fp32 VALU beside another wave's MFMAs is
  // priced higher than two plain instructions by the microarchitecture guide); same values either way.
  f32x2 hf = __builtin_convertvector(h, f32x2);
  float r0 = a - hf.x, r1 = b - hf.y;
  asm volatile("" : "+v"(r0), "+v"(r1));
  f32x2 r = {r0, r1};
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  f32x2 mf = __builtin_convertvector(m, f32x2);
  float q0 = r0 - mf.x, q1 = r1 - mf.y;
  asm volatile("" : "+v"(q0), "+v"(q1));
  f32x2 r2 = {q0, q1};
#else
  f32x2 r = v - __builtin_convertvector(h, f32x2);
  bf16x2 m = __builtin_convertvector(r, bf16x2);
  f32x2 r2 = r - __builtin_convertvector(m, f32x2);
#endif
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

// ---------------------------------------------------------------------------------------------------------------------------------
// "split-f16" (round 6): the same idea with TWO fp16 terms per operand and THREE products per K-step - half the matrix-pipe work.
// fp16 carries 11 significand bits: x s = hi + lo + d with hi = RN16(x s), lo = RN16(x s - hi), |d| <= 2^-22 |x s|, for a power-of-two
// tensor scale s that puts the tensor's largest magnitude B into [2^14, 2^15] (fp16's range is 6e-5 .. 65504: without the scale the
// low terms of ordinary activations would be subnormal).  s comes from an UPPER BOUND of max |x| that the producing kernel leaves in a
// device scalar (running maximum of what it stores, one atomic per wave) - a bound that is too large only costs low-order bits of
// the smallest elements.  Products: hi hi + hi lo + lo hi, each exact in the MFMA, accumulated in fp32; the scales are powers of two,
// so multiplying the accumulator by 1 / (s_x s_w) in the epilogue is exact.  What is dropped: the representation residues d (<= 2^-22
// relative per operand) and lo lo (<= 2^-22): measured against float64 on the CostRegNet layer shapes the representation error is
// 1.4e-7 rms / 7e-7 max of a unit-scale output where PyTorch's own fp32 convolution is at 3.4-4.6e-7 rms / 3-4e-6 max
// (scripts/ab/r06_split_f16_precision.py): below the rounding error of an fp32 convolution, i.e. still fp32-class - and held to the same
// bar as split-bf16 by tests/test_hip_parity.py (<= 1.5x the float64 error of an fp32 convolution per layer shape).
// Layout: the split-bf16 one ([term][8] x 16 B per position / per weight vector) with term 2 unused, so rings, pitches and packers keep
// their geometry.
__device__ __forceinline__ void split2_f16(float a, float b, float s, uint32_t& hi, uint32_t& lo) {
  f32x2 v = {a * s, b * s};
  f16x2 h = __builtin_convertvector(v, f16x2);
  f32x2 r = v - __builtin_convertvector(h, f32x2);
  f16x2 l = __builtin_convertvector(r, f16x2);
  hi = *reinterpret_cast<uint32_t*>(&h);
  lo = *reinterpret_cast<uint32_t*>(&l);
}
__device__ __forceinline__ void split_store8_f16(unsigned char* dst, const float4& a, const float4& b, float s) {
  uint32_t h[4], l[4];
  split2_f16(a.x, a.y, s, h[0], l[0]);
  split2_f16(a.z, a.w, s, h[1], l[1]);
  split2_f16(b.x, b.y, s, h[2], l[2]);
  split2_f16(b.z, b.w, s, h[3], l[3]);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  d4[0] = make_uint4(h[0], h[1], h[2], h[3]);
  d4[1] = make_uint4(l[0], l[1], l[2], l[3]);
}
// scale of a tensor whose magnitudes are bounded by B: the power of two s with B s in (2^14, 2^15]; B = 0 / not finite -> 1
__device__ __forceinline__ float sf16_scale(float B) {
  if (!(B > 0.f) || !(B < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(B, &e);                       // B = m 2^e, m in [0.5, 1)
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, 15 - e);
}
#define SF16_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((a).h, (b).h, acc, 0, 0, 0)
// the three partial products of one K-step, smallest first; W[term] = weights (hi, lo), X[q][term] = data of N-tile q
#define SF16_TERMS(ACC, T0, NQ, W, X)                                            \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SF16_MFMA(ACC[(T0) + q_], (W)[1], (X)[q_][0]); /* lo x hi */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SF16_MFMA(ACC[(T0) + q_], (W)[0], (X)[q_][1]); /* hi x lo */ \
  __builtin_amdgcn_sched_barrier(0);                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NQ); ++q_) SF16_MFMA(ACC[(T0) + q_], (W)[0], (X)[q_][0]); /* hi x hi */ \
  __builtin_amdgcn_sched_barrier(0);
// running maximum of |v| over a wave, left in `slot` with one atomic (non-negative floats order like their bit patterns)
__device__ __forceinline__ void sf16_publish_bound(float amax, float* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  // most waves do not raise the running maximum: a relaxed read first keeps the read-modify-writes on the ONE address down to the few
  // that can matter (one atomic per wave was 59 000 atomics = +0.24 ms on the 1600x1184 x 8-plane stage); a stale read only costs
  // an atomic that changes nothing
  if ((threadIdx.x & 63) == 0 && slot) {
    unsigned int* u = reinterpret_cast<unsigned int*>(slot);
    const unsigned int mine = __float_as_uint(amax);
    if (mine > __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(u, mine);
  }
}

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


}  // namespace
