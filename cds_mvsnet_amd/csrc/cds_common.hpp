// Shared helpers for the gfx950 kernels of libcdsmvs_hip.so.
// Built with -ffp-contract=off: every fused multiply-add in this library is an explicit fmaf(),
// every separately rounded mul/add is written as such, so the fp32 operation order documented in
// DESIGN.md is what the hardware executes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/cds_mvsnet_hip.h"

#define CDS_WAVE 64

// conv3d_zmg.hip: the z-marching split-bf16 kernels behind cds_conv3d_sbf_f32; CDS_ZMG_UNSUPPORTED = shape stays on the tiled kernels
#define CDS_ZMG_UNSUPPORTED 1
int cds_conv3d_zmg_dispatch(const float* x, const void* wsp, const float* bias, float* out, int Cin, int Cout, int D, int H, int W,
                            int stride, int pair, int act, hipStream_t st, const float* in_bound = nullptr, float w_inv = 1.f,
                            float* out_bound = nullptr);

// native 4-float vector (volatile-loadable, unlike HIP's float4 struct): pins a 16-byte LDS read
using cds_f4 = float __attribute__((ext_vector_type(4)));

extern "C" int cds_debug_poison_lds(unsigned pattern);   // lib.hip

static inline int cds_launch_status() {
  hipError_t e = hipGetLastError();
  if (const char* p = getenv("CDS_DEBUG_POISON_LDS")) {    // debug aid, see lib.hip: the next kernel starts on poisoned LDS
    if (e == hipSuccess) (void)cds_debug_poison_lds((unsigned)strtoul(p, nullptr, 16));
  }
  return e == hipSuccess ? 0 : -(int)e;
}

static inline int cds_ceil_div(int a, int b) { return (a + b - 1) / b; }

// A/B and test knobs are read PER LAUNCH (a getenv is ~100 ns next to a ~5 us launch): a test that sets one after the first
// call gets the new value.  cds_env_is(name, c): the variable is set and starts with c; cds_env_int: its integer value or dflt.
static inline bool cds_env_is(const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; }
static inline bool cds_env_set(const char* name) { return getenv(name) != nullptr; }
static inline int cds_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// Opt a kernel into more than 64 KB of dynamic LDS once per DEVICE (function attributes are per device: one process may
// drive several, e.g. nn.DataParallel replicas; `mask` is a per-kernel static).
static inline int cds_allow_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& mask) {
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (mask.load(std::memory_order_acquire) & bit) return 0;
  // idempotent: two host threads may both apply it; the bit is published only AFTER the attribute is in place,
  // so no thread can launch the > 64 KB kernel on this device before it
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return -(int)e; }
  mask.fetch_or(bit, std::memory_order_release);
  return 0;
}

// Bijective XCD-aware remap of a linear workgroup id (guide T1): the dispatcher places block b on
// XCD b % 8, so consecutive logical tiles are handed to the same XCD to share its private L2.
__device__ __forceinline__ int cds_xcd_remap(int bid, int nwg) {
  const int nx = 8;
  if (nwg < 2 * nx) return bid;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Tell the compiler a pointer is wave-uniform (it is derived from blockIdx / kernel arguments only): the loads
// through it become scalar-cache loads with SALU address arithmetic instead of per-load v_readfirstlane pairs.
__device__ __forceinline__ const float* cds_uniform_ptr(const float* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}

// lean epilogue activation for the conv kernels (none / ReLU / sigmoid)
__device__ __forceinline__ float cds_act_conv(float v, int act) {
  if (act == CDS_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == CDS_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

__device__ __forceinline__ float cds_apply_act(float v, int act) {
  switch (act) {
    case CDS_ACT_RELU: return fmaxf(v, 0.0f);
    case CDS_ACT_LEAKY01: return v > 0.0f ? v : v * 0.1f;
    case CDS_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case CDS_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
