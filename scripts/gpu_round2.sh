# Round-2 measurement pass on one MI355X.  Usage (through gpurun): bash scripts/gpu_round2.sh [tag]
TAG=${1:-r02}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/tests.log; cat $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --force-dist --steps 20 --no-cpu-baseline --no-extras > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; tail -c 1500 $OUT/bench_forcedist.json; tail -3 $OUT/bench_forcedist.err
timeout 600 python scripts/swarm_bench.py --agents 131072 > $OUT/swarm_1gpu.json 2> $OUT/swarm.err; cat $OUT/swarm_1gpu.json; tail -3 $OUT/swarm.err
