// Micro-benchmark: achievable rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with independent accumulators,
// with and without one ds_read_b32 per MFMA (the A operand pattern of conv3d_mfma.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float s = 0.f;
  if (MODE == 0 || MODE == 1) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float av = a;
        if (MODE == 1) av = lds[(threadIdx.x + 64 * i + it) & 4095];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)iters * (MODE < 2 ? 16 : 4) * waves_per_simd;          // MFMAs per SIMD
  double flop_per = (MODE < 2 ? 2048.0 : 4096.0);
  double tf = nm * 1024 * flop_per / (ms * 1e-3) / 1e12;
  printf("%-34s waves/SIMD=%d  %.3f ms  %.1f ns per MFMA per SIMD  -> %.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e6 / nm, tf);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("mfma_f32_16x16x4, register operands", w);
    run<1>("mfma_f32_16x16x4 + ds_read_b32 each", w);
    run<2>("mfma_f32_32x32x2, register operands", w);
  }
  return 0;
}
