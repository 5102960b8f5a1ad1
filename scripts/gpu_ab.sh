cd $GRAFT_REPO_ROOT
for v in "" alt_d4 alt_d6 alt_d8 "" alt_d4; do
  if [ -n "$v" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/$v.so; else unset CRX_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('${v:-default}  value %.2f G/s  kernel_ms %.4f  frac %.4f'%(r['value']/1e9, r['roofline']['kernel_ms'], r['roofline']['frac']))"
done
