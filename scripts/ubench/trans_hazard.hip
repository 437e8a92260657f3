// Micro-benchmark of the transcendental-forwarding hazard on gfx950 (round 5): v_rcp_f32 / v_exp_f32 followed by a dependent
// non-transcendental VALU op with different instruction "gaps" in between.  The destination register holds a marker before the
// transcendental op; a consumer that reads the register too early sees the marker (or a partial result).  Mismatches are counted
// per lane quarter.  hipcc --offload-arch=gfx950 -O2 trans_hazard.hip -o trans_hazard && ./trans_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GAP_NONE ""
#define GAP_NOP0 "s_nop 0\n"
#define GAP_NOP1 "s_nop 1\n"
#define GAP_SMOV1 "s_mov_b32 s20, 0x1234\n"
#define GAP_SMOV3 "s_mov_b32 s20, 0x1234\ns_mov_b32 s21, 0x1235\ns_mov_b32 s22, 0x1236\n"
#define GAP_SLOAD1 "s_load_dword s20, %[p], 0x0\n"
#define GAP_SLOAD3 "s_load_dword s20, %[p], 0x0\ns_load_dwordx2 s[22:23], %[p], 0x8\ns_load_dwordx4 s[24:27], %[p], 0x10\n"
#define GAP_VALU1 "v_mov_b32 %[t], %[x]\n"
#define GAP_WAITCNT "s_waitcnt lgkmcnt(0)\n"

#define KERNEL(NAME, TRANS, GAP)                                                                                         \
  __global__ void NAME(const float* __restrict__ in, const float* __restrict__ sp, unsigned* __restrict__ bad, int iters) { \
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;                                                                \
    const int lane = threadIdx.x & 63;                                                                                    \
    float x = in[tid];                                                                                                    \
    unsigned nbad = 0;                                                                                                    \
    for (int i = 0; i < iters; ++i) {                                                                                     \
      float r, y, t, ref;                                                                                                 \
      asm volatile(TRANS " %[ref], %[x]\ns_nop 7\ns_nop 7\n" : [ref] "=v"(ref) : [x] "v"(x));                            \
      asm volatile("v_mov_b32 %[r], 0x42f60000\ns_nop 7\n"                                                               \
                   TRANS " %[r], %[x]\n" GAP "v_mul_f32 %[y], 1.0, %[r]\ns_nop 7\ns_waitcnt lgkmcnt(0)\n"                \
                   : [r] "=&v"(r), [y] "=&v"(y), [t] "=&v"(t)                                                             \
                   : [x] "v"(x), [p] "s"(sp)                                                                              \
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                             \
      if (y != ref) ++nbad;                                                                                               \
      x = x * 1.0009765625f + 0.001f;                                                                                     \
      if ((i & 63) == 63) x = in[(tid + i) & 0xffff];      /* a memory stall now and then: waves fall out of lock-step */ \
    }                                                                                                                     \
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);                                                                           \
  }

KERNEL(rcp_none, "v_rcp_f32", GAP_NONE)
KERNEL(rcp_nop0, "v_rcp_f32", GAP_NOP0)
KERNEL(rcp_nop1, "v_rcp_f32", GAP_NOP1)
KERNEL(rcp_smov1, "v_rcp_f32", GAP_SMOV1)
KERNEL(rcp_smov3, "v_rcp_f32", GAP_SMOV3)
KERNEL(rcp_sload1, "v_rcp_f32", GAP_SLOAD1)
KERNEL(rcp_sload3, "v_rcp_f32", GAP_SLOAD3)
KERNEL(rcp_valu1, "v_rcp_f32", GAP_VALU1)
KERNEL(exp_none, "v_exp_f32", GAP_NONE)
KERNEL(exp_nop0, "v_exp_f32", GAP_NOP0)
KERNEL(exp_sload3, "v_exp_f32", GAP_SLOAD3)
KERNEL(sqrt_sload3, "v_sqrt_f32", GAP_SLOAD3)
KERNEL(sqrt_nop0, "v_sqrt_f32", GAP_NOP0)

typedef void (*kern_t)(const float*, const float*, unsigned*, int);

int main() {
  const int N = 1 << 16;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0.5f + (float)(i % 977) * 0.0131f;
  float *in, *sp;
  unsigned* bad;
  hipMalloc(&in, N * 4);
  hipMalloc(&sp, 256);
  hipMalloc(&bad, 16);
  hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(sp, 0, 256);
  struct { const char* name; kern_t k; } ks[] = {
      {"rcp | (none)", rcp_none}, {"rcp | s_nop 0", rcp_nop0}, {"rcp | s_nop 1", rcp_nop1}, {"rcp | s_mov x1", rcp_smov1},
      {"rcp | s_mov x3", rcp_smov3}, {"rcp | s_load x1", rcp_sload1}, {"rcp | s_load x3", rcp_sload3}, {"rcp | v_mov x1", rcp_valu1},
      {"exp | (none)", exp_none}, {"exp | s_nop 0", exp_nop0}, {"exp | s_load x3", exp_sload3}, {"sqrt | s_load x3", sqrt_sload3},
      {"sqrt | s_nop 0", sqrt_nop0}};
  const int iters = 4096;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int blocks = 256 * waves_per_simd;      // 256-thread blocks: one per CU and "wave per SIMD"
    printf("---- %d waves per SIMD (%d blocks of 256), %d iterations per lane\n", waves_per_simd, blocks, iters);
    for (auto& e : ks) {
      hipMemset(bad, 0, 16);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, in, sp, bad, iters);
      unsigned hb[4];
      hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
      const double tot = (double)blocks * 256 * iters / 4;
      printf("%-18s mismatches by lane quarter: %10u %10u %10u %10u   (rate of the last quarter %.2e)\n", e.name, hb[0], hb[1], hb[2], hb[3],
             hb[3] / tot);
    }
  }
  return 0;
}
