"""Timing probe (never part of the product library): a copy of csrc/deconv_prob_zm.hip with s_memtime stamps around the phases
of one workgroup's half-step loop -> cds_mvsnet_amd/_variants/libcdsmvs_hip.dpz_timeline.so, read back with cds_dpz_probe_dump()."""
import os, subprocess, glob
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "cds_mvsnet_amd", "csrc")
out = os.path.join(root, "cds_mvsnet_amd", "_variants"); os.makedirs(out, exist_ok=True)
s = open(os.path.join(src, "deconv_prob_zm.hip")).read()
def rep(a, b, n=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, n)
rep('__device__ __forceinline__ float dpz_fma(',
    '__device__ long long g_dbg[3 * 8 * 256];\n'
    '#define STAMP(w, e) do { if (dbgwg && (t - qs) < 256 && lane == 0) g_dbg[((w) * 8 + (e)) * 256 + (t - qs)] = __builtin_readcyclecounter(); } while (0)\n'
    '__device__ __forceinline__ float dpz_fma(')
rep('  if (a0 >= a1) return;', '  if (a0 >= a1) return;\n  const bool dbgwg = blockIdx.x == 100;')
# prob / producer wave 8
rep('''      const int q = t - 1;
      if (q >= qs && q <= qe) process(q);''', '''      if (wave == C::CW) STAMP(2, 0);
      const int q = t - 1;
      if (q >= qs && q <= qe) process(q);
      if (wave == C::CW) STAMP(2, 1);''')
rep('''      const int o = t - 2;
      if (o >= 2 * a0''', '''      if (wave == C::CW) STAMP(2, 2);           // (no staging in the prob waves since round 5)
      const int o = t - 2;
      if (o >= 2 * a0''')
rep('''        A[2][r] = 0.f;
      }
      __syncthreads();''', '''        A[2][r] = 0.f;
      }
      if (wave == C::CW) STAMP(2, 3);
      __syncthreads();''')
# consumers: waves 0 and 7
rep('''  for (int t = qs; t <= te; ++t) {
    if (t <= qe) {
      const int a = t >> 1;''', '''  const int cwv = wave == 0 ? 0 : 1; const bool cst = wave == 0 || wave == 7;
  for (int t = qs; t <= te; ++t) {
    if (cst) STAMP(cwv, 0);
    if (t <= qe) {
      const int a = t >> 1;''')
rep('''        load_b(b1, cur + C::ROUNDB + b_h1);
        __builtin_amdgcn_sched_barrier(0);''', '''        load_b(b1, cur + C::ROUNDB + b_h1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F); if (cst) STAMP(cwv, 1);
        __builtin_amdgcn_sched_barrier(0);''')
rep('''        SBF_TERMS(acc[1], 0, C::NT, w1, b1);
        epilogue(sk0, 0);''', '''        SBF_TERMS(acc[1], 0, C::NT, w1, b1);
        if (cst) STAMP(cwv, 2);
        epilogue(sk0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F); if (cst) STAMP(cwv, 4);''')
rep('''            if (c_dst[h] >= 0) split_store8(base + c_dst[h], va[h], vb[h]);
        }''', '''            if (c_dst[h] >= 0) split_store8(base + c_dst[h], va[h], vb[h]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); if (cst) STAMP(cwv, 3);''')
rep('''          SBF_TERMS(acc[1], 0, C::NT, w0, b1);
        }
        epilogue(sk1, 1);''', '''          SBF_TERMS(acc[1], 0, C::NT, w0, b1);
        }
        if (cst) STAMP(cwv, 2);
        epilogue(sk1, 1);
        __builtin_amdgcn_s_waitcnt(0xC07F); if (cst) STAMP(cwv, 3);''')
import sys
tag = "dpz_timeline"
if "nomfma" in sys.argv:
    rep('#include "sbf_common.hpp"', '#include "sbf_common.hpp"\n#undef SBF_MFMA\n#define SBF_MFMA(acc, a, b) asm volatile("" : "+v"(acc) : "v"((a).v), "v"((b).v))')
    tag += "_nomfma"
if "pk" in sys.argv:           # A/B: packed FMAs over channel pairs (two partial sums per output) instead of the plain ones
    rep('    float A[3][C::PR];', '    f32x2 A[3][C::PR];')
    rep('      for (int r = 0; r < C::PR; ++r) A[s][r] = 0.f;', '      for (int r = 0; r < C::PR; ++r) A[s][r] = (f32x2){0.f, 0.f};')
    rep('''            float acc = A[2 - kz][r];
            acc = dpz_fma(dv.x, wv.x, acc);
            acc = dpz_fma(dv.y, wv.y, acc);
            acc = dpz_fma(dv.z, wv.z, acc);
            acc = dpz_fma(dv.w, wv.w, acc);
            A[2 - kz][r] = acc;''', '''            A[2 - kz][r] = __builtin_elementwise_fma((f32x2){dv.x, dv.y}, (f32x2){wv.x, wv.y}, A[2 - kz][r]);
            A[2 - kz][r] = __builtin_elementwise_fma((f32x2){dv.z, dv.w}, (f32x2){wv.z, wv.w}, A[2 - kz][r]);''')
    rep('          if (st_off[r] >= 0) po[st_off[r]] = A[0][r];', '          if (st_off[r] >= 0) po[st_off[r]] = A[0][r].x + A[0][r].y;')
    rep('        A[2][r] = 0.f;', '        A[2][r] = (f32x2){0.f, 0.f};')
    tag += "_pk"
if "noconsumer" in sys.argv:   # consumers only keep the barriers
    rep('    if (t <= qe) {\n      const int a = t >> 1;', '    if (t < 0) {\n      const int a = t >> 1;')
    tag += "_noconsumer"
s += '''
extern "C" int cds_dpz_probe_dump(long long* host, int n) {
  return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
'''
p = os.path.join(src, "_probe_dpz_timeline.hip")
open(p, "w").write(s)
obj = os.path.join(out, "_probe_dpz_timeline.o")
try:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-c", p, "-o", obj])
    others = [o for o in glob.glob(os.path.join(src, "*.o")) if not o.endswith("deconv_prob_zm.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *others, obj, "-o", os.path.join(out, f"libcdsmvs_hip.{tag}.so")])
finally:
    os.remove(p)
    if os.path.exists(obj): os.remove(obj)
print("built")
