# A/B of libcrx variants on the single-step EKF kernel.  usage: gpu_step_ab.sh [variant ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" != "default" ]; then export CRX_LIB_PATH=$GRAFT_REPO_ROOT/cpprobotics_amd/alt_$v.so; else unset CRX_LIB_PATH; fi
  echo "== $v"; timeout 300 python scripts/gpu_step_time.py 2>gpurun_out/ab/step_$v.err
done 2>&1 | tee -a gpurun_out/ab/step_results.txt
