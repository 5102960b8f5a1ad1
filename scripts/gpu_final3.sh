# Round-3 final pass: tests, smoke, bench line, A/B tables, side benches, swarm shard, profiles.  Usage (through gpurun): bash scripts/gpu_final3.sh [tag]
TAG=${1:-r03final}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $OUT/host.txt; cat $OUT/host.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/tests.log; cat $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 1200 $OUT/bench.json; echo; tail -2 $OUT/bench.err
timeout 300 python bench.py --force-dist --steps 20 --no-cpu-baseline --no-extras > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
timeout 200 python scripts/gpu_dare_lanes_ab.py > $OUT/dare_lanes_ab.jsonl 2> $OUT/dare_ab.err; cut -c1-200 $OUT/dare_lanes_ab.jsonl
timeout 300 python scripts/gpu_mpc_lanes_ab.py > $OUT/mpc_lanes_ab.jsonl 2> $OUT/mpc_ab.err; cut -c1-200 $OUT/mpc_lanes_ab.jsonl
timeout 300 python scripts/gpu_loop_lanes_ab.py 2> $OUT/loop_ab.err | grep -v amdgpu > $OUT/loop_lanes_ab.jsonl; cut -c1-250 $OUT/loop_lanes_ab.jsonl
timeout 200 python scripts/gpu_mpc_loop_err.py > $OUT/mpc_loop_err.jsonl 2>&1; cat $OUT/mpc_loop_err.jsonl
timeout 900 python scripts/side_bench.py > $OUT/side_bench.jsonl 2> $OUT/side_bench.err; cut -c1-300 $OUT/side_bench.jsonl
timeout 300 python scripts/swarm_bench.py --agents 131072 > $OUT/swarm_1gpu.json 2> $OUT/swarm.err; cut -c1-600 $OUT/swarm_1gpu.json; tail -2 $OUT/swarm.err
timeout 2400 bash scripts/gpu_prof3.sh $TAG/prof > $OUT/prof.log 2>&1; tail -30 $OUT/prof.log
