#!/bin/bash
# Timing probes of the conv9 z-march kernel (csrc/deconv3d_zm.hip; WRONG RESULTS by construction, never part of the product library):
#   nostage : no loads / split / LDS stores inside the march;  halfstage : every second item only
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/cds_mvsnet_amd/csrc
out=$root/cds_mvsnet_amd/_variants; mkdir -p $out
others=$(ls $src/*.o | grep -v deconv3d_zm.o)
build() {
  sed -E "$2" $src/deconv3d_zm.hip > $src/_probe_dzm_$1.hip
  if cmp -s $src/_probe_dzm_$1.hip $src/deconv3d_zm.hip; then echo "probe $1: patch did not apply"; rm -f $src/_probe_dzm_$1.hip; exit 1; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c $src/_probe_dzm_$1.hip -o $out/_probe_dzm_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $out/_probe_dzm_$1.o -o $out/libcdsmvs_hip.dzm_$1.so
  rm -f $src/_probe_dzm_$1.hip $out/_probe_dzm_$1.o
}
build nostage 's/^      deposit\(a \+ 2\);.*$/      \/\* probe \*\//; s/^      issue\(a \+ 3\);$/      \/\* probe \*\//' &
build halfstage 's/^      for \(int h = 0; h < C::IPT; \+\+h\)( \{)?$/      for (int h = 0; h < C::IPT; h += 2)\1/' &
wait
ls $out | grep dzm
