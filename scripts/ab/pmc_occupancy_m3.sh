R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_occ
rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_occ -o p -- python $R/scripts/time_forward.py 1184 1600 5 > $R/gpurun_out/pmc_occ.log 2>&1
cd $R
python scripts/pmc_summary.py $(find gpurun_out/pmc_occ -name "*.db" | head -1) > gpurun_out/pmc_occ_m3_summary.txt 2>&1
find gpurun_out/pmc_occ -name "*.db" -delete
