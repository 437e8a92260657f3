// Micro-benchmark: do v_mfma_f32_16x16x4_f32 and v_pk_fma_f32 streams from DIFFERENT waves of the same SIMD co-execute?
// Workgroup = 8 waves (2 per SIMD); waves with (wave_id & 1) == 0 run the MFMA loop, the others the packed-FMA loop
// (mode 2); modes 0 / 1 run the same loop in every wave for reference.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o coexec coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mfma_loop(int iters, float a, float b) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  return s;
}
__device__ __forceinline__ float fma_loop(int iters, float seed) {
  v2f acc[16], a[4], b[4];
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){seed + i, seed - i};
  for (int i = 0; i < 4; ++i) { a[i] = (v2f){1.0f + seed * i, 1.0f - seed * i}; b[i] = (v2f){seed * 0.5f * i, seed * 0.25f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(a[i & 3], b[(i >> 2) & 3], acc[i]);
  }
  v2f s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  return s.x + s.y;
}

// mode 0: all waves MFMA; 1: all waves FMA; 2: even waves MFMA, odd waves FMA
__global__ __launch_bounds__(512) void k(float* out, int mode, int it_mfma, int it_fma, float seed) {
  const int wave = threadIdx.x >> 6;
  float r;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave >> 2) == 0);   // waves 0-3 -> SIMD 0-3 first slot
  if (do_mfma) r = mfma_loop(it_mfma, seed + threadIdx.x, seed);
  else r = fma_loop(it_fma, seed);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

float run(int mode, int it_mfma, int it_fma) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, 10, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, it_mfma, it_fma, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}
int main() {
  const int IM = 4000, IF = 4000;   // 8 MFMA (8*14 ns = 113 ns) vs 16 pk_fma (16*2.3 = 37 ns) per iteration
  float t0 = run(0, IM, IF), t1 = run(1, IM, 3 * IF), t2 = run(2, IM, 3 * IF);
  printf("all 8 waves MFMA  (2/SIMD): %.3f ms  -> %.1f TF\n", t0, 256.0 * 8 * IM * 8 * 2048 / (t0 * 1e-3) / 1e12);
  printf("all 8 waves pkFMA (2/SIMD): %.3f ms  -> %.1f TF\n", t1, 256.0 * 8 * 3 * IF * 16 * 256 / (t1 * 1e-3) / 1e12);
  printf("4 waves MFMA + 4 waves pkFMA: %.3f ms -> MFMA %.1f TF + FMA %.1f TF if both finish together\n", t2,
         256.0 * 4 * IM * 8 * 2048 / (t2 * 1e-3) / 1e12, 256.0 * 4 * 3 * IF * 16 * 256 / (t2 * 1e-3) / 1e12);
  printf("(alone, 1 wave/SIMD each: MFMA half = %.3f ms, FMA half = %.3f ms expected if independent)\n", t0 / 2, t1 / 2);
  return 0;
}
