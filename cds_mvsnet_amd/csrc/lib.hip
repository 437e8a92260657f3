// libcdsmvs_hip.so - version entry point and the LDS-poisoning debug aid.
#include "cds_common.hpp"
extern "C" int cds_version(void) { return 100; /* 0.1.0 */ }

// Debug aid (round 5): fill the LDS of every CU with a pattern.  A kernel that reads LDS it never wrote sees whatever the previous
// workgroup on that CU left there: reproducible as long as the same kernels run in the same order on one stream, different as soon as
// another stream's kernels share the CUs.  With CDS_DEBUG_POISON_LDS=<hex pattern> in the environment every entry point of the library
// synchronises and poisons the LDS after its launch (cds_launch_status), so the NEXT kernel starts on poisoned LDS whatever ran before;
// tests/test_hip_parity.py::test_results_do_not_depend_on_stale_lds runs the stages under two patterns and compares bit for bit.
namespace {
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned pattern, int words) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < words; i += 1024) lds_words[i] = pattern;
}
}  // namespace

extern "C" int cds_debug_poison_lds(unsigned pattern) {
  const int bytes = 160 * 1024;
  static std::atomic<unsigned long long> ok{0};
  if (int e = cds_allow_lds(reinterpret_cast<const void*>(poison_lds_kernel), bytes, ok)) return e;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(1024), bytes, 0, pattern, bytes / 4);   // one workgroup per CU at a time
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return 0;
}
