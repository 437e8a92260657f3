#!/bin/bash
# round 4: the profile set committed under profiles/r04_* (kernel stats, MFMA utilisation, HBM traffic, cascade breakdown, positions A/B)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O/r04
cd $R
scripts/profile_round.sh r04 > /dev/null 2>&1
scripts/profile_m3.sh r04 > /dev/null 2>&1
cp $O/r04_kernel_stats.md $O/r04_pmc_mfma.txt $O/pmc_traffic.md $O/pmc_traffic.json $O/r04_m3_breakdown.txt $O/r04_m3_pmc_mfma.txt $O/r04/ 2>/dev/null
# positions-once lever: fast positions remove 80 of the 132 position instructions per view and plane pair for EVERY
# channel group; sharing exact positions between the C / 8 groups could at most remove 132 for all groups but one
for shp in "296 400 48 32" "592 800 32 16 600 720" "1184 1600 8 8 600 615" "512 640 192 8"; do
  for ex in 1 0; do TAG=positions EXACT=$ex CL=1 python scripts/time_warp.py $shp 2>&1 | grep -v amdgpu; done
done > $O/r04/positions_ab.txt
python bench.py > $O/r04/bench_b.json 2> $O/r04/bench_b.err
tail -3 $O/r04/positions_ab.txt
