"""Timing probe (never part of the product library): a copy of csrc/conv3d_zmg.hip with s_memtime stamps around the phases of one
workgroup's stage loop -> cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_timeline.so, read back with cds_zmg_probe_dump()."""
import os, subprocess, glob
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "cds_mvsnet_amd", "csrc")
out = os.path.join(root, "cds_mvsnet_amd", "_variants"); os.makedirs(out, exist_ok=True)
s = open(os.path.join(src, "conv3d_zmg.hip")).read()
def rep(a, b, n=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, n)
rep('__device__ __attribute__((aligned(32))) float g_zmg_zeros[8];',
    '__device__ __attribute__((aligned(32))) float g_zmg_zeros[8];\n__device__ long long g_dbg[3 * 8 * 256];\n'
    '#define STAMP(w, e) do { if (dbgwg && st < 256 && lane == 0) g_dbg[((w) * 8 + (e)) * 256 + st] = __builtin_readcyclecounter(); } while (0)')
rep('  const int zin0 = S * z0 - 1;', '  const int zin0 = S * z0 - 1;\n  const bool dbgwg = blockIdx.x == 300;')
# producer stamps (wave 8 = first producer of a CW = 8 kernel, wave 4 for CW = 4): after barrier / deposit done / issue done
rep('''      if (st + 1 < nstages) {
        deposit(st + 1, 1);
        if (st + 3 < nstages) issue(st + 3, 1);
      }
      __syncthreads();                                 // #(st + 1): stage st consumed, stage st + 1 staged''',
'''      if (wave == Cfg::CW) STAMP(2, 0);
      if (st + 1 < nstages) {
        deposit(st + 1, 1);
        if (wave == Cfg::CW) STAMP(2, 1);
        if (st + 3 < nstages) issue(st + 3, 1);
      }
      if (wave == Cfg::CW) STAMP(2, 2);
      __syncthreads();                                 // #(st + 1): stage st consumed, stage st + 1 staged''')
rep('''      if (st + 2 < nstages) {
        deposit(st + 2, 0);
        if (st + 4 < nstages) issue(st + 4, 0);
      }
      __syncthreads();                                 // #(st + 2)''',
'''      { const int st_ = st; { const int st = st_ + 1; if (wave == Cfg::CW) STAMP(2, 0);
      if (st + 1 < nstages) {
        deposit(st + 1, 0);
        if (wave == Cfg::CW) STAMP(2, 1);
        if (st + 3 < nstages) issue(st + 3, 0);
      }
      if (wave == Cfg::CW) STAMP(2, 2); } }
      __syncthreads();                                 // #(st + 2)''')
# consumer stamps: wave 0 (early) and wave 4 of CW = 8 (late)
rep('    if (late && st > 0) finish(st - 1);', '    const int cwv = wave == 0 ? 0 : 1; const bool cst = wave == 0 || (Cfg::CW == 8 && wave == 4);\n    if (cst) STAMP(cwv, 0);\n    if (late && st > 0) finish(st - 1);')
rep('    __builtin_amdgcn_s_setprio(0);\n#pragma unroll\n    for (int t = 0; t < NTW; ++t) acc[t]', '    __builtin_amdgcn_s_setprio(0);\n    if (cst) STAMP(cwv, 1);\n#pragma unroll\n    for (int t = 0; t < NTW; ++t) acc[t]')
rep('    __builtin_amdgcn_s_setprio(CDS_ZMG_CPRIO);\n    if (Cfg::KSPL == 2) {', '    __builtin_amdgcn_s_setprio(CDS_ZMG_CPRIO);\n    if (cst) STAMP(cwv, 2);\n    if (Cfg::KSPL == 2) {')
rep('    __syncthreads();                                   // #(st + 1)\n  }\n  if (Cfg::KSPL >= 2 || late)', '    if (cst) STAMP(cwv, 3);\n    __syncthreads();                                   // #(st + 1)\n  }\n  if (Cfg::KSPL >= 2 || late)')
s += '''
extern "C" int cds_zmg_probe_dump(long long* host, int n) {
  return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
'''
p = os.path.join(src, "_probe_timeline.hip")
open(p, "w").write(s)
obj = os.path.join(out, "_probe_timeline.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-c", p, "-o", obj])
others = [o for o in glob.glob(os.path.join(src, "*.o")) if not o.endswith("conv3d_zmg.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *others, obj, "-o", os.path.join(out, "libcdsmvs_hip.probe_timeline.so")])
os.remove(p); os.remove(obj)
print("built")
