// Micro-benchmark (round 5): how long after v_mfma_f32_16x16x32_bf16 can a VALU instruction READ the result, or WRITE a result
// register (WAW), when the gap between the two is filled with (a) one s_nop N, (b) K scalar-ALU instructions, (c) K independent
// VALU instructions?  The accumulator holds a marker (100.0) before the MFMA; A = B = 1.0 (bf16) so the result is 132.0.
// A read that comes too early sees 100.0; a write (v_mov 7.0) that comes too early is overwritten by the late MFMA write (132.0).
// Counts per (lane quarter, register).  hipcc --offload-arch=gfx950 -O2 mfma_result_hazard.hip -o mfma_result_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define SALU "s_mov_b32 s20, 0x1234\n"
#define VALU "v_mov_b32 %[t], %[t]\n"

// (the asm operand syntax for register tuples is awkward from C++; the kernels below use fixed physical registers instead)
#define KBODY(GAP, TAIL)                                                                                                 \
  asm volatile(                                                                                                          \
      "v_mov_b32 v40, 0x3f803f80\nv_mov_b32 v41, 0x3f803f80\nv_mov_b32 v42, 0x3f803f80\nv_mov_b32 v43, 0x3f803f80\n"      \
      "v_mov_b32 v44, 0x42c80000\nv_mov_b32 v45, 0x42c80000\nv_mov_b32 v46, 0x42c80000\nv_mov_b32 v47, 0x42c80000\n"      \
      "s_nop 7\ns_nop 7\ns_nop 7\n"                                                                                      \
      "v_mfma_f32_16x16x32_bf16 v[44:47], v[40:43], v[40:43], v[44:47]\n" GAP TAIL                                       \
      "s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n"                                                                             \
      "v_mov_b32 %[o0], v48\nv_mov_b32 %[o1], v49\nv_mov_b32 %[o2], v50\nv_mov_b32 %[o3], v51\n"                          \
      : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)                                                       \
      :                                                                                                                  \
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "s20")
// read test: copy the result registers right after the gap
#define TAIL_READ "v_mov_b32 v48, v44\nv_mov_b32 v49, v45\nv_mov_b32 v50, v46\nv_mov_b32 v51, v47\n"
// WAW test: overwrite the result registers with 7.0 right after the gap, read them back much later
#define TAIL_WAW "v_mov_b32 v44, 0x40e00000\nv_mov_b32 v45, 0x40e00000\nv_mov_b32 v46, 0x40e00000\nv_mov_b32 v47, 0x40e00000\n" \
                 "s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n" TAIL_READ

#define KERNEL(NAME, GAP, TAIL, EXPECT)                                                          \
  __global__ void NAME(unsigned* __restrict__ bad, int iters) {                                  \
    const int lane = threadIdx.x & 63;                                                           \
    unsigned nb[4] = {0, 0, 0, 0};                                                               \
    for (int i = 0; i < iters; ++i) {                                                            \
      float o0, o1, o2, o3;                                                                      \
      KBODY(GAP, TAIL);                                                                          \
      nb[0] += o0 != EXPECT; nb[1] += o1 != EXPECT; nb[2] += o2 != EXPECT; nb[3] += o3 != EXPECT; \
    }                                                                                            \
    for (int r = 0; r < 4; ++r) if (nb[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], nb[r]);          \
  }

#define FAMILY(P, TAIL, EXPECT)                                   \
  KERNEL(P##_nop0, "s_nop 0\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop1, "s_nop 1\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop3, "s_nop 3\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop5, "s_nop 5\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop7, "s_nop 7\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop9, "s_nop 9\n", TAIL, EXPECT)                     \
  KERNEL(P##_nop11, "s_nop 11\n", TAIL, EXPECT)                   \
  KERNEL(P##_nop15, "s_nop 15\n", TAIL, EXPECT)                   \
  KERNEL(P##_nop15_7, "s_nop 15\ns_nop 7\n", TAIL, EXPECT)        \
  KERNEL(P##_salu2, REP2(SALU), TAIL, EXPECT)                     \
  KERNEL(P##_salu4, REP4(SALU), TAIL, EXPECT)                     \
  KERNEL(P##_salu8, REP8(SALU), TAIL, EXPECT)                     \
  KERNEL(P##_salu12, REP8(SALU) REP4(SALU), TAIL, EXPECT)         \
  KERNEL(P##_salu16, REP16(SALU), TAIL, EXPECT)                   \
  KERNEL(P##_salu24, REP16(SALU) REP8(SALU), TAIL, EXPECT)        \
  KERNEL(P##_salu32, REP16(SALU) REP16(SALU), TAIL, EXPECT)       \
  KERNEL(P##_valu2, REP2("v_mov_b32 v52, v40\n"), TAIL, EXPECT)   \
  KERNEL(P##_valu4, REP4("v_mov_b32 v52, v40\n"), TAIL, EXPECT)   \
  KERNEL(P##_valu8, REP8("v_mov_b32 v52, v40\n"), TAIL, EXPECT)   \
  KERNEL(P##_valu12, REP8("v_mov_b32 v52, v40\n") REP4("v_mov_b32 v52, v40\n"), TAIL, EXPECT) \
  KERNEL(P##_valu16, REP16("v_mov_b32 v52, v40\n"), TAIL, EXPECT)

FAMILY(rd, TAIL_READ, 132.0f)
FAMILY(ww, TAIL_WAW, 7.0f)

typedef void (*kern_t)(unsigned*, int);
#define ENTRY(P, S) {#P " | " #S, P##_##S}
#define ENTRIES(P) ENTRY(P, nop0), ENTRY(P, nop1), ENTRY(P, nop3), ENTRY(P, nop5), ENTRY(P, nop7), ENTRY(P, nop9), ENTRY(P, nop11), \
                   ENTRY(P, nop15), ENTRY(P, nop15_7), ENTRY(P, salu2), ENTRY(P, salu4), ENTRY(P, salu8), ENTRY(P, salu12), ENTRY(P, salu16), \
                   ENTRY(P, salu24), ENTRY(P, salu32), ENTRY(P, valu2), ENTRY(P, valu4), ENTRY(P, valu8), ENTRY(P, valu12), ENTRY(P, valu16)

int main() {
  unsigned* bad;
  if (hipMalloc(&bad, 64) != hipSuccess) return 1;
  struct { const char* name; kern_t k; } ks[] = {ENTRIES(rd), ENTRIES(ww)};
  const int iters = 2048;
  for (int wps : {1, 2}) {
    printf("---- %d wave(s) per SIMD; rd = VALU READ of the result after the gap (stale = 100), ww = VALU WRITE of the result registers after the gap "
           "(lost = overwritten by the late MFMA write); mismatches [lane quarter][register]\n", wps);
    for (auto& e : ks) {
      (void)hipMemset(bad, 0, 64);
      hipLaunchKernelGGL(e.k, dim3(256 * wps), dim3(256), 0, 0, bad, iters);
      unsigned hb[16];
      (void)hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost);
      unsigned tot = 0;
      for (int i = 0; i < 16; ++i) tot += hb[i];
      printf("%-14s total %10u  q0[%u %u %u %u] q1[%u %u %u %u] q2[%u %u %u %u] q3[%u %u %u %u]\n", e.name, tot, hb[0], hb[1], hb[2], hb[3], hb[4],
             hb[5], hb[6], hb[7], hb[8], hb[9], hb[10], hb[11], hb[12], hb[13], hb[14], hb[15]);
    }
  }
  return 0;
}
