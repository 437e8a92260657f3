cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base ""; do
  if [ -n "$v" ]; then export CDS_MVSNET_LIB=cds_mvsnet_amd/_variants/libcdsmvs_hip.$v.so; else unset CDS_MVSNET_LIB; fi
  echo -n "${v:-new}: "
  python bench.py --no-extras --steps 8 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value',round(d['value'],2),'kernel_ms',d['kernel_ms'])"
done; done
unset CDS_MVSNET_LIB
python -m pytest tests/test_hip_parity.py -q -x -k "conv or deconv or costreg or cost_reg" 2>&1 | tail -2
