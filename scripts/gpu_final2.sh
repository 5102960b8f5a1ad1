# Round-2 final pass: tests, bench line, side benches, swarm shard, profiles.  Usage (through gpurun): bash scripts/gpu_final2.sh [tag]
TAG=${1:-r02final}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $OUT/host.txt; cat $OUT/host.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/tests.log; cat $OUT/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 1500 $OUT/bench.json; echo; tail -2 $OUT/bench.err
timeout 600 python bench.py --force-dist --steps 20 --no-cpu-baseline --no-extras > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
timeout 900 python scripts/side_bench.py > $OUT/side_bench.jsonl 2> $OUT/side_bench.err; cut -c1-400 $OUT/side_bench.jsonl
timeout 600 python scripts/swarm_bench.py --agents 131072 > $OUT/swarm_1gpu.json 2> $OUT/swarm.err; cat $OUT/swarm_1gpu.json
bash scripts/gpu_prof2.sh $TAG/prof > $OUT/prof.log 2>&1; tail -30 $OUT/prof.log
