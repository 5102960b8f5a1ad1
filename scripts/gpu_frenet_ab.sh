# A/B of libcrx variants on the Frenet side bench.  usage: gpu_frenet_ab.sh [variant ...]  ("default" = libcrx.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" != "default" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/alt_$v.so; else unset CRX_LIB_PATH; fi
  timeout 300 python scripts/side_bench.py --frenet-only 2>gpurun_out/ab/fr_$v.err | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print('$v  plans/s %.3f M  ms %.3f  parity %s'%(r['plans_per_s']/1e6, r['ms'], r['parity']))
except Exception as e: print('$v FAILED', e)"
done 2>&1 | tee -a gpurun_out/ab/fr_results.txt
