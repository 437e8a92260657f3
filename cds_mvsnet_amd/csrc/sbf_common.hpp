// Shared pieces of the split-bf16 (3xBF16 error-compensated) matrix-core kernels: vector types, the exact three-way split,
// the six-partial-product MFMA sequence.  See the header comment of conv3d_sbf.hip for the arithmetic.
#pragma once
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
