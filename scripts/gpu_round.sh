# Full GPU check of a round: tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Usage on the GPU box: bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-400 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-300 > $OUT/smoke.log; cat $OUT/smoke.log
bash scripts/gpu_prof.sh $TAG 2>&1 | grep -v "^+" | cut -c1-600 | tail -30
