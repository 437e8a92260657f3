// Micro-benchmark of the split-bf16 consumer loop: per step 12 ds_read_b128 (3 terms x 4 N-tiles) + 24 MFMAs (6 partial
// products x 4 accumulators), double-buffered like conv3d_sbf.hip, optionally with the 3 weight loads per step from global
// memory (L1 / L2 resident).  2 waves per SIMD (512-thread workgroups, one per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_bf16_feed mfma_bf16_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union BV { uint4 u; bf16x8 v; };
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16((a).v, (b).v, acc, 0, 0, 0)

template <int LDSREADS, int WLOAD, int STRIDE>
__global__ __launch_bounds__(512) void k(float* out, const uint4* w, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f813f80u, 0x3f803f82u);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  BV a[2][3], b[2][4][3];
  for (int t = 0; t < 3; ++t) a[0][t].u = a[1][t].u = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  for (int q = 0; q < 4; ++q) for (int t = 0; t < 3; ++t) b[0][q][t].u = b[1][q][t].u = make_uint4(0x3f803f80u, 0x3f803f80u + q, 0x3f803f80u, 0x3f803f80u);
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  const uint4* wl = w + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int nb = s ^ 1;
      if (LDSREADS) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 3; ++t) b[nb][q][t].u = lds[((lane * STRIDE + q * 64 * STRIDE + it * 7) & 4095) / 1 % 4096 * 0 + ((lane * STRIDE + q * 193 + t + it * 3) & 4095)];
      }
      if (WLOAD) {
#pragma unroll
        for (int t = 0; t < 3; ++t) a[nb][t].u = wl[((it * 2 + s) % 7 * 3 + t) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][2], b[s][q][0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][1], b[s][q][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][0], b[s][q][2]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][1], b[s][q][0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][0], b[s][q][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) MF(acc[q], a[s][0], b[s][q][0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int LDSREADS, int WLOAD, int STRIDE>
void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  uint4* w; hipMalloc(&w, 21 * 64 * 16); hipMemset(w, 0x3f, 21 * 64 * 16);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<LDSREADS, WLOAD, STRIDE>), dim3(256), dim3(512), 65536, 0, out, w, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<LDSREADS, WLOAD, STRIDE>), dim3(256), dim3(512), 65536, 0, out, w, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)iters * 48 * 2;          // MFMAs per SIMD (2 waves)
  printf("%-58s %.2f ns per MFMA per SIMD -> %.0f TFLOP/s (bf16), %.0f%% of the 8.1 ns register-operand rate\n", name, ms * 1e6 / nm,
         nm * 1024 * 16384.0 / (ms * 1e-3) / 1e12, 100 * 8.1 / (ms * 1e6 / nm));
  hipFree(out); hipFree(w);
}
int main() {
  run<0, 0, 1>("24 MFMAs per step, register operands");
  run<1, 0, 1>("+ 12 ds_read_b128 per step (16 B lane stride)");
  run<1, 0, 3>("+ 12 ds_read_b128 per step (48 B lane stride)");
  run<0, 1, 1>("+ 3 global weight loads per step");
  run<1, 1, 3>("+ 12 ds_read_b128 (48 B stride) + 3 weight loads per step");
  return 0;
}
