set -x
python -m pytest tests/test_hip_parity.py -x -q -k "warp or fast_positions" -s 2>&1 | grep -v "^$" | tail -8
V=cds_mvsnet_amd/_variants/libcdsmvs_hip.intidx.so
for i in 1 2; do
for ex in 1 0; do
  TAG=floatidx EXACT=$ex CL=1 python scripts/time_warp.py
  TAG=intidx CDS_MVSNET_LIB=$V EXACT=$ex CL=1 python scripts/time_warp.py
done; done
