"""Timing probe (never part of the product library): csrc/conv2d_sbf.hip with s_memtime stamps around the phases of the DynamicConv
kernel (one workgroup = one tile) -> cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_dyn.so, read back with cds_dyn_probe_dump()."""
import os, subprocess, glob
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "cds_mvsnet_amd", "csrc")
out = os.path.join(root, "cds_mvsnet_amd", "_variants"); os.makedirs(out, exist_ok=True)
s = open(os.path.join(src, "conv2d_sbf.hip")).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep('constexpr int POSB = 48;', 'constexpr int POSB = 48;\n__device__ long long g_dyn[8 * 4096];\n'
    '#define STAMP(e) do { if (tid == 0 && blockIdx.x < 4096) g_dyn[(e) * 4096 + blockIdx.x] = __builtin_readcyclecounter(); } while (0)')
rep('  f32x4 acc[NBR][NBLK][4];       // M-tiles', '  STAMP(0);\n  f32x4 acc[NBR][NBLK][4];       // M-tiles')
rep('''      unsigned char* dst = lds + (row * IXP + 4 * q) * POSB;''', '''      if (v[7].x == 1.2345e30f) STAMP(7);   // forces the loads to complete before the next stamp
      if (u == tid) STAMP(1);
      unsigned char* dst = lds + (row * IXP + 4 * q) * POSB;''')
rep('''    __syncthreads();

    const uint4* __restrict__ wr = wl''', '''    STAMP(2);
    __syncthreads();
    STAMP(3);

    const uint4* __restrict__ wr = wl''')
rep('''  if (MODE == 2) {
    // visibility CNN layer''', '''  STAMP(4);
  if (MODE == 2) {
    // visibility CNN layer''')
rep('''    __syncthreads();
    // one record per (tile, channel): the four waves in a fixed order''', '''    STAMP(5);
    __syncthreads();
    // one record per (tile, channel): the four waves in a fixed order''')
s += '''
extern "C" int cds_dyn_probe_dump(long long* host, int n) {
  return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dyn), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
'''
p = os.path.join(src, "_probe_dyn.hip")
open(p, "w").write(s)
obj = os.path.join(out, "_probe_dyn.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-c", p, "-o", obj])
others = [o for o in glob.glob(os.path.join(src, "*.o")) if not o.endswith("conv2d_sbf.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *others, obj, "-o", os.path.join(out, "libcdsmvs_hip.probe_dyn.so")])
os.remove(p); os.remove(obj)
print("built")
