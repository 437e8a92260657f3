// Micro-benchmark: v_mfma_f32_16x16x32_bf16 issue rate as a function of the number of independent accumulators between
// two MFMAs on the SAME accumulator (the dependent distance of the split-bf16 term loop), at 1 and 2 waves per SIMD, with
// and without three ds_read_b128 per 6 MFMAs (the data-operand pattern of conv3d_sbf.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_bf16_dep mfma_bf16_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union BV { uint4 u; bf16x8 v; };

template <int NACC, int LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(seed + i, seed, i, seed ^ i);
  __syncthreads();
  BV a, b[NACC];
  a.u = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  for (int i = 0; i < NACC; ++i) b[i].u = make_uint4(0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u);
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) b[i].u = lds[(threadIdx.x * 3 + 64 * i + it) & 4095];
    }
#pragma unroll
    for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b[i].v, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int LDS>
void run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)iters * 6 * NACC * waves_per_simd;          // MFMAs per SIMD
  printf("independent accumulators %d, %s, waves/SIMD=%d: %.2f ns per MFMA per SIMD -> %.0f TFLOP/s (bf16)\n", NACC,
         LDS ? "one ds_read_b128 per 6 MFMAs" : "register operands", waves_per_simd, ms * 1e6 / nm,
         nm * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int w : {1, 2}) {
    run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<8, 0>(w);
    run<4, 1>(w); run<8, 1>(w);
  }
  return 0;
}
