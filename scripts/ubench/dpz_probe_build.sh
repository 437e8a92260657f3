#!/bin/bash
# Probe builds of the fused conv11 + prob kernel (csrc/deconv_prob_zm.hip): a patched COPY of the source is compiled and linked
# with the product objects into cds_mvsnet_amd/_variants/libcdsmvs_hip.dpz_<tag>.so; the product source holds no probe code.
#   dpz_probe_build.sh noprob|nomfma|noskip|nostore|nostage|prio0|prio3|base [extra hipcc flags]
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/cds_mvsnet_amd/_variants; mkdir -p $out
src=$root/cds_mvsnet_amd/csrc
tmp=$src/_dpz_probe_$tag.hip
cp $src/deconv_prob_zm.hip $tmp
case $tag in
  base) ;;
  noprob) sed -i 's|if (q >= qs \&\& q <= qe) process(q);|/* probe: no prob arithmetic */|' $tmp ;;
  nomfma) sed -i 's|#include "sbf_common.hpp"|#include "sbf_common.hpp"\n#undef SBF_MFMA\n#define SBF_MFMA(acc, a, b) asm volatile("" : "+v"(acc) : "v"((a).v), "v"((b).v))|' $tmp ;;
  noskip) sed -i 's|    if (t > qe) return;|    if (t >= 0) { for (int py = 0; py < 2; ++py) for (int q = 0; q < C::NT; ++q) sk[py][q] = make_float4(0.f, 0.f, 0.f, 0.f); return; }|' $tmp ;;
  nostore) sed -i 's|if (o >= 2 \* a0 \&\& o < 2 \* a1 \&\& lane_ok) {|if (o >= 2 * a0 \&\& o < 2 * a1 \&\& lane_ok \&\& A[0][0].x == 123.456f) {|' $tmp ;;
  nostage) # no input staging inside the march (the consumers' loads, split and LDS stores compiled out): upper bound for removing it from
           # the kernel altogether.  (Round 5 ran this probe - and `nosplit`, an unsplit 48-byte copy - against the tree in which the PROB
           # waves still staged: 975 / 1 350 us against 1 180, profiles/r05_costreg_tail_closure.md; that tree is the parent of commit a1b02ac.)
    sed -i 's|^            if (c_dst\[h\] >= 0) split_store8(base + c_dst\[h\], va\[h\], vb\[h\]);|            ; /* probe: no staging */|' $tmp ;;
  prio0) sed -i 's|__builtin_amdgcn_s_setprio(2);|__builtin_amdgcn_s_setprio(0);|' $tmp ;;
  prio3) sed -i 's|__builtin_amdgcn_s_setprio(2);|__builtin_amdgcn_s_setprio(3);|' $tmp ;;
  *) echo "unknown probe $tag"; exit 1 ;;
esac
if [ "$tag" != base ] && cmp -s $tmp $src/deconv_prob_zm.hip; then echo "probe $tag: patch did not apply"; rm -f $tmp; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $tmp -o $out/dpz_$tag.o
rm -f $tmp
objs=$(ls $src/*.o | grep -v deconv_prob_zm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/dpz_$tag.o -o $out/libcdsmvs_hip.dpz_$tag.so
rm -f $out/dpz_$tag.o
echo $out/libcdsmvs_hip.dpz_$tag.so
