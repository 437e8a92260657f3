#!/bin/bash
# Timing probes of the z-marching kernels (WRONG RESULTS by construction, never part of the product library): patched copies of
# csrc/conv3d_zmg.hip linked against the product objects -> cds_mvsnet_amd/_variants/libcdsmvs_hip.probe_<name>.so
#   krep2   : the K-loop of every stage runs twice        -> (krep2 - base) = cost of one pass of the K-loops
#   nostore : the epilogue's global stores are skipped
#   noload  : the producers read the zero block only (no HBM input traffic)
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/cds_mvsnet_amd/csrc
out=$root/cds_mvsnet_amd/_variants; mkdir -p $out
others=$(ls $src/*.o | grep -v conv3d_zmg.o)
build() {  # name, sed expression
  sed -E "$2" $src/conv3d_zmg.hip > $src/_probe_$1.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c $src/_probe_$1.hip -o $out/_probe_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $out/_probe_$1.o -o $out/libcdsmvs_hip.probe_$1.so
  rm -f $src/_probe_$1.hip $out/_probe_$1.o
}
build krep2 's/^    for \(int ss = 0; ss < NS; \+\+ss\) \{/    for (int rep_ = 0; rep_ < 2; ++rep_) for (int ss = 0; ss < NS; ++ss) {/' &
build nostore 's/^    sbf_store4\(obase/    if (o.x == 1234.5f) sbf_store4(obase/' &
build noload 's/const float\* __restrict__ src = ok \? x \+ \(\(long long\)z \* plane_elems \+ s_off\[h\]\) : g_zmg_zeros;/const float* __restrict__ src = g_zmg_zeros;/' &
# halfstage: the producers load, split and store only every second item of a stage (upper bound for handing part of the staging to the
# consumer waves, which wait 500-2000 cycles per stage at the barrier); nostage: none inside the march
build halfstage 's/^      for \(int h = 0; h < PPT; \+\+h\) \{$/      for (int h = 0; h < PPT; h += 2) {/' &
build nostage 's/^      for \(int h = 0; h < PPT; \+\+h\) \{$/      for (int h = 0; h < 0; ++h) {/' &
wait
ls -la $out/*probe*
