cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ekf_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_ekf_gpu.log
