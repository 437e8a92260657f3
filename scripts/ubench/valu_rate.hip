// Micro-benchmark: issue rate of v_pk_fma_f32 / v_fma_f32 / v_pk_mul_f32 on gfx950 for different operand patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, const float* __restrict__ wts) {
  v2f acc[16], a[4], b[4];
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){seed + i, seed - i};
  for (int i = 0; i < 4; ++i) { a[i] = (v2f){1.0f + seed * i, 1.0f - seed * i}; b[i] = (v2f){seed * 0.5f * i, seed * 0.25f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) acc[i] = __builtin_elementwise_fma(a[i & 3], b[(i >> 2) & 3], acc[i]);          // pk_fma: 3 VGPR pairs
      if (MODE == 1) { acc[i].x = __builtin_fmaf(a[i & 3].x, b[(i >> 2) & 3].x, acc[i].x); acc[i].y = __builtin_fmaf(a[i & 3].y, b[(i >> 2) & 3].y, acc[i].y); }  // 2 plain fma
      if (MODE == 2) acc[i] = __builtin_elementwise_fma(acc[i], a[i & 3], acc[i]);                     // pk_fma: acc reused as source
      if (MODE == 3) acc[i] = acc[i] * a[i & 3];                                                        // pk_mul
      if (MODE == 4) acc[i] = __builtin_elementwise_fma(a[i & 3], (v2f){b[i & 3].x, b[i & 3].x}, acc[i]);  // pk_fma with a broadcast (op_sel) operand
      if (MODE == 5) { acc[i].x = __builtin_fmaf(a[i & 3].x, wts[(it & 7) * 32 + i], acc[i].x); acc[i].y = __builtin_fmaf(a[i & 3].y, wts[(it & 7) * 32 + 16 + i], acc[i].y); }  // v_fmac with an SGPR (s_load) operand
    }
    if (MODE == 1 || MODE == 5) asm volatile("" ::: "memory");
  }
  v2f s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 CUs, 4 waves per block -> waves_per_simd per SIMD
  float* out; hipMalloc(&out, blocks * 256 * 4);
  float* wts; hipMalloc(&wts, 4096); hipMemset(wts, 0, 4096);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, wts);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, wts);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * 16 * ((MODE == 1 || MODE == 5) ? 2 : 1) * waves_per_simd;   // per SIMD
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("pk_fma 3 distinct pairs", w);
    run<1>("2x v_fma_f32", w);
    run<2>("pk_fma acc as src", w);
    run<3>("pk_mul", w);
    run<4>("pk_fma broadcast src", w);
    run<5>("2x v_fmac SGPR operand", w);
  }
  return 0;
}
