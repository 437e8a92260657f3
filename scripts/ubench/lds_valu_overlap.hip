// Micro-benchmark: do ds_read_b128 and v_pk_fma_f32 overlap on gfx950?  Per loop iteration: 16 independent pk_fma and
// K conflict-free ds_read_b128 whose results are folded in with K cheap ops at the end of the iteration.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o lds_valu_overlap lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int K, int NF, int MASK = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  __shared__ v4f lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (v4f){seed, seed, seed, seed};
  __syncthreads();
  v2f acc[16], a[4], b[4];
  for (int i = 0; i < 16; ++i) acc[i] = (v2f){seed + i, seed - i};
  for (int i = 0; i < 4; ++i) { a[i] = (v2f){1.0f + seed * i, 1.0f - seed * i}; b[i] = (v2f){seed * 0.5f * i, seed * 0.25f}; }
  v4f sink = (v4f){0, 0, 0, 0};
  int base = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    v4f t[K > 0 ? K : 1];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (MASK == 0 || (threadIdx.x & MASK) == 0) t[j] = lds[(base + 64 * j + it) & 2047];   // MASK: only 1 of (MASK+1) lanes reads
      else t[j] = (v4f){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i & 15] = __builtin_elementwise_fma(a[i & 3], b[(i >> 2) & 3], acc[i & 15]);
#pragma unroll
    for (int j = 0; j < K; ++j) sink += t[j];
  }
  v2f s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + sink.x + sink.y + sink.z + sink.w;
}

template <int K, int NF, int MASK = 0>
void run(int waves_per_simd) {
  int blocks = 256 * waves_per_simd;
  float* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, NF, MASK>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, NF, MASK>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mask=%d pk_fma=%2d ds_read_b128=%2d waves/SIMD=%d: %.3f ms -> %.1f ns per iteration per wave-slot\n", MASK, NF, K, waves_per_simd, ms,
         ms * 1e6 / iters / waves_per_simd);
  hipFree(out);
}
int main() {
  for (int w : {2, 4}) {
    run<0, 16>(w); run<2, 16>(w); run<4, 16>(w); run<8, 16>(w);
    run<8, 0>(w); run<4, 0>(w);
    run<8, 32>(w);
    run<8, 16, 3>(w); run<8, 16, 7>(w); run<8, 0, 3>(w);
  }
  return 0;
}
