# A/B of libcrx variants on the fused EKF with the PEst history written out.  usage: gpu_phist_ab.sh [variant ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" != "default" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/alt_$v.so; else unset CRX_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --settle 60 2>gpurun_out/ab/ph_$v.err | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); e=r['extra']['ekf_with_P_history']; print('$v  value %.2f G/s | P-hist %.2f G updates/s  %.0f GB/s  frac %.3f'%(r['value']/1e9, e['updates_per_s']/1e9, e['GB_per_s'], e['frac_of_8TBps']))
except Exception as ex: print('$v FAILED', ex)"
done 2>&1 | tee -a gpurun_out/ab/ph_results.txt
