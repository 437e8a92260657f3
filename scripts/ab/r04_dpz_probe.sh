#!/bin/bash
# A/B of the fused conv11 + prob kernel: probe builds (scripts/ubench/dpz_probe_build.sh) and the z-segment count.
O=gpurun_out/r04; mkdir -p $O
{
python scripts/ab/dpz_time.py
for t in $(ls cds_mvsnet_amd/_variants/ | grep "dpz_" | sed 's/libcdsmvs_hip.\(.*\).so/\1/'); do
  CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$t.so python scripts/ab/dpz_time.py
done
for n in 1 2 3 4 7 8; do CDS_DPZ_NSEG=$n python scripts/ab/dpz_time.py; done
} > $O/dpz_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC"; do
  rm -rf $R/$O/pmc_dpz
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-include-regex deconv_prob -d $R/$O/pmc_dpz -o p --output-format csv -- python scripts/ab/dpz_time.py > /dev/null 2>&1)
  f=$(find $R/$O/pmc_dpz -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $R/$O/dpz_probe.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"    {k:34s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
rm -rf $R/$O/pmc_dpz
