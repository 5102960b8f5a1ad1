cd $GRAFT_REPO_ROOT
for nv in 65536 131072 262144 1048576; do
 for v in default oldtrig default oldtrig; do
  if [ "$v" != "default" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/alt_$v.so; else unset CRX_LIB_PATH; fi
  T=1000; if [ $nv -ge 1048576 ]; then T=250; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --vehicles $nv --T $T 2>/dev/null | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print('$nv $v  value %.2f G/s  kernel_ms %.4f  frac %.4f'%(r['value']/1e9, r['roofline']['kernel_ms'], r['roofline']['frac']))
except Exception as e: print('$v FAILED', e)"
 done
done
