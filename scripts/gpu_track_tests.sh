cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_track_gpu.py -x -q -m gpu 2>&1 | tail -25 | cut -c1-300 | tee gpurun_out/pytest_track_gpu.log
