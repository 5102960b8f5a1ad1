cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_frenet_gpu.py -x -q -m gpu 2>&1 | tail -40 | cut -c1-400 | tee gpurun_out/pytest_frenet_gpu.log
