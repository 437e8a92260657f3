#!/bin/bash
# Build a tagged copy of the library with extra -D flags for A/B timing on the GPU box:
#   scripts/build_variant.sh noredo -DCDS_K3_NOREDO   ->  cds_mvsnet_amd/_variants/libcdsmvs_hip.noredo.so
#   CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.noredo.so python scripts/time_warp.py
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/cds_mvsnet_amd/_variants
mkdir -p $out/obj_$tag
cd $root/cds_mvsnet_amd/csrc
objs=""
for f in lib warp warp_lds warp_bwd regress conv3d conv3d_mfma conv3d_sbf conv3d_zmg deconv_prob_zm deconv3d_zm conv2d conv2d_mfma conv2d_sbf fusion refine train3d train2d; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o $out/obj_$tag/$f.o &
  objs="$objs $out/obj_$tag/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libcdsmvs_hip.$tag.so
rm -rf $out/obj_$tag
echo $out/libcdsmvs_hip.$tag.so
