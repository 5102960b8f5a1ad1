# A/B of libcrx variants on the headline EKF workload.  usage: gpu_ab.sh [variant ...]  ("default" = libcrx.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" != "default" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/alt_$v.so; else unset CRX_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 2>gpurun_out/ab/$v.err | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print('$v  value %.2f G/s  kernel_ms %.4f  frac %.4f'%(r['value']/1e9, r['roofline']['kernel_ms'], r['roofline']['frac']))
except Exception as e: print('$v FAILED', e)"
done 2>&1 | tee -a gpurun_out/ab/results.txt
