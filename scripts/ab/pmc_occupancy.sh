# average resident waves per CU per kernel of the M1 step: SQ_WAVE_CYCLES (quad-cycles, summed over waves) x 4 / (GRBM_GUI_ACTIVE / 8 x 256 CUs)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_occ
rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_occ -o p -- python $R/bench.py --no-extras --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_occ.log 2>&1
cd $R
python scripts/pmc_summary.py $(find gpurun_out/pmc_occ -name "*.db" | head -1) > gpurun_out/pmc_occ_summary.txt 2>&1
find gpurun_out/pmc_occ -name "*.db" -delete
