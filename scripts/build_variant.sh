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
# as in the Makefile: no packed-fp32 instructions; PK=1 scripts/build_variant.sh pk builds WITH them (the round-5 code generation)
if [ -z "$PK" ]; then NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"; else NOPK=""; fi
# ONLY="feat_cl conv2d" scripts/build_variant.sh tag -D...: recompile only those sources with the flags, link the tree's other objects
for f in lib warp warp_lds warp_bwd regress conv3d conv3d_mfma conv3d_sbf conv3d_zmg deconv_prob_zm deconv3d_zm conv2d conv2d_mfma conv2d_sbf feat_cl fusion refine train3d train2d loss; do
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $f "; then objs="$objs $root/cds_mvsnet_amd/csrc/$f.o"; continue; fi
  extra=""; if [ "$f" = feat_cl ] || [ "$f" = conv2d_sbf ]; then extra="-fno-slp-vectorize"; fi   # as in the Makefile
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $NOPK $extra "$@" -c $f.hip -o $out/obj_$tag/$f.o 2> >(grep -v "packed-fp32-ops' is not a recognized" >&2) &
  objs="$objs $out/obj_$tag/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libcdsmvs_hip.$tag.so
rm -rf $out/obj_$tag
echo $out/libcdsmvs_hip.$tag.so
