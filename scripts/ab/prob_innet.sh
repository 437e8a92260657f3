#!/bin/bash
# in-network kernel times of conv11 + prob with the VALU prob kernel (default) and the matrix-core one (CDS_PROB_MFMA=1)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/prof_pi$v
  CDS_PROB_MFMA=$v CDS_PROB_TY=8 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_pi$v -o t -- python $R/bench.py --no-pmc --no-extras --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_pi$v -name "*.db" | head -1)
  echo "== CDS_PROB_MFMA=$v"; (cd $R; TOPN=40 python scripts/kernel_breakdown.py $f 2 | grep -E "deconv3d_sbf_ws|conv3d_k3_pipe|prob_sbf|softargmin|total")
  find $R/gpurun_out/prof_pi$v -name "*.db" -delete
done
