#!/bin/bash
# Counters of the fused conv11 + prob kernel at the headline shape (counter-only passes, no trace domains).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
out=$O/dpz_counters.txt; : > $out
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $O/pmc_dpz
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-include-regex deconv_prob -d $O/pmc_dpz -o p --output-format csv -- python scripts/ab/dpz_time.py > /dev/null 2>&1)
  f=$(find $O/pmc_dpz -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"    {k:34s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
rm -rf $O/pmc_dpz
cd $R && python scripts/ab/dpz_time.py 2>&1 | grep -v amdgpu >> $out
