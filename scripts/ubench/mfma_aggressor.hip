// Synthetic aggressor (round 6): a kernel that does NOTHING but matrix-core instructions on register-resident operands - no LDS, no
// loads inside the loop - next to K1 (scripts/ab/r06_synthetic_aggressor.py).  If K1 computes wrong packed-fp32 results beside THIS,
// the effect is not a property of an instruction sequence of ours but of dense MFMA activity itself; the knobs say which activity:
//   kind  0 f16 16x16x32   1 bf16 16x16x32   2 fp32 16x16x4   3 VALU fma only (control)
//   data  0 random-ish operands   1 all-zero operands (same instruction stream, almost no toggling in the multipliers)
//   burst MFMAs issued back to back before the pause;  pause = s_sleep argument (0: none; 1 = 64 cycles)
// Built by the script: hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_aggressor.hip -o mfma_aggressor.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void aggr_kernel(float* __restrict__ out, int iters, int burst8, int pause, int zero) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  union { f16x8 h; bf16x8 b; uint32_t u[4]; float f[4]; } A, B;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A.u[k] = zero ? 0u : (0x3c003800u ^ (uint32_t)(lane * 0x01010101u + k * 0x00110011u) & 0x03ff03ffu);   // f16 / bf16 values of order 1
    B.u[k] = zero ? 0u : (0x38003c00u ^ (uint32_t)(lane * 0x00030005u + k * 0x01000100u) & 0x03ff03ffu);
  }
  float va = zero ? 0.f : 1.0f + lane * 0.001f, vb = zero ? 0.f : 0.999f;
  for (int i = 0; i < iters; ++i) {
    for (int bb = 0; bb < burst8; ++bb) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (KIND == 0) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.h, B.h, acc[r], 0, 0, 0);
        else if (KIND == 1) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.b, B.b, acc[r], 0, 0, 0);
        else if (KIND == 2) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.f[0], B.f[0], acc[r], 0, 0, 0);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_fmaf(acc[r][j], vb, va);
        }
      }
    }
    if (pause) __builtin_amdgcn_s_sleep(1);
    if (pause > 1) __builtin_amdgcn_s_sleep(2);
    if (pause > 2) __builtin_amdgcn_s_sleep(8);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) s += acc[r].x + acc[r].y + acc[r].z + acc[r].w;
  if (s == 12345.678f) out[0] = s;
}

extern "C" int mfma_aggressor(int kind, float* out, int grid, int iters, int burst8, int pause, int zero, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(aggr_kernel<0>, dim3(grid), dim3(256), 0, st, out, iters, burst8, pause, zero); break;
    case 1: hipLaunchKernelGGL(aggr_kernel<1>, dim3(grid), dim3(256), 0, st, out, iters, burst8, pause, zero); break;
    case 2: hipLaunchKernelGGL(aggr_kernel<2>, dim3(grid), dim3(256), 0, st, out, iters, burst8, pause, zero); break;
    default: hipLaunchKernelGGL(aggr_kernel<3>, dim3(grid), dim3(256), 0, st, out, iters, burst8, pause, zero); break;
  }
  return (int)hipGetLastError();
}
