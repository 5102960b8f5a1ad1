#!/bin/bash
# One measurement pass on one MI355X, in steps.  Usage (through gpurun):  bash scripts/gpu_pass.sh TAG [STEP ...]
#   steps (default: all, in this order):
#     host       nproc / cgroup quota / affinity of the box
#     tests      pytest -m gpu   (TESTS="tests/test_lqr_gpu.py -k dense" narrows it)
#     smoke      __graft_entry__.smoke()
#     bench      python bench.py                       -> bench.json
#     forcedist  bench.py --force-dist (N > 1 code path, one rank)
#     oversub    bench.py --gpus 2 --oversubscribe (self-launched ranks sharing the GPU: dry run of the N > 1 launch path)
#     ab         the lanes-per-agent A/B scripts (DARE structured / dense, MPC, closed loop), the MPC closed-loop drift, the lane-refilling
#                Riccati kernel against the masked one, the HBM calibration
#     side       scripts/side_bench.py (DARE, MPC, tracking, PF, DWA, Frenet)
#     swarm      scripts/swarm_bench.py, one GPU's shard of BASELINE configs[4]
#     fuzz       scripts/gpu_fuzz_bitexact.py (SEED0=first seed, default 300; SEEDS=how many, default 20)
#     prof       scripts/gpu_prof.sh TAG/prof (rocprofv3 kernel stats + PMC passes, markers) and scripts/gpu_mpc_traffic.sh (the MPC solve's HBM traffic at 262,144 agents)
# Everything lands in gpurun_out/TAG/; `python scripts/collect_profiles.py TAG rNN` copies the judged summaries into profiles/rNN/.
# (Rounds 1-3 kept one copy of this script per round — gpu_round.sh, gpu_round2.sh, gpu_final2.sh, gpu_final3.sh — and one-off
# variants of single steps; this is their union.)
TAG=${1:?usage: gpu_pass.sh TAG [STEP ...]}; shift
STEPS=${*:-host tests smoke bench forcedist oversub ab side swarm fuzz prof}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in $STEPS; do
  echo "== $s"
  case $s in
    host) nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $OUT/host.txt; rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1 >> $OUT/host.txt; cat $OUT/host.txt ;;
    tests) timeout 900 python -m pytest ${TESTS:-tests} -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/tests.log; cat $OUT/tests.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
    bench) rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature" | head -8 > $OUT/bench_smi_before.txt
           timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 1500 $OUT/bench.json; echo; tail -2 $OUT/bench.err
           rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature" | head -8 > $OUT/bench_smi_after.txt; cat $OUT/bench_smi_after.txt ;;
    forcedist) timeout 300 python bench.py --force-dist --steps 20 --no-cpu-baseline --no-extras > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; head -c 600 $OUT/bench_forcedist.json; echo ;;
    oversub) timeout 600 python bench.py --gpus 2 --oversubscribe --steps 10 --warmup 2 --settle 20 > $OUT/bench_oversub2.json 2> $OUT/bench_oversub2.err; tail -c 1200 $OUT/bench_oversub2.json; echo; tail -3 $OUT/bench_oversub2.err ;;
    ab) timeout 200 python scripts/gpu_dare_lanes_ab.py > $OUT/dare_lanes_ab.jsonl 2> $OUT/dare_ab.err; cut -c1-200 $OUT/dare_lanes_ab.jsonl
        timeout 300 python scripts/gpu_mpc_lanes_ab.py > $OUT/mpc_lanes_ab.jsonl 2> $OUT/mpc_ab.err; cut -c1-200 $OUT/mpc_lanes_ab.jsonl
        timeout 300 python scripts/gpu_loop_lanes_ab.py 2> $OUT/loop_ab.err | grep -v amdgpu > $OUT/loop_lanes_ab.jsonl; cut -c1-250 $OUT/loop_lanes_ab.jsonl
        timeout 200 python scripts/gpu_mpc_loop_err.py > $OUT/mpc_loop_err.jsonl 2>&1; cat $OUT/mpc_loop_err.jsonl
        timeout 300 python scripts/gpu_dare_dense_ab.py > $OUT/dare_dense_lanes_ab.jsonl 2> $OUT/dare_dense_ab.err; cut -c1-200 $OUT/dare_dense_lanes_ab.jsonl
        timeout 600 python scripts/gpu_dare_refill_ab.py > $OUT/dare_refill_ab.jsonl 2> $OUT/dare_refill_ab.err; cut -c1-200 $OUT/dare_refill_ab.jsonl
        timeout 300 python scripts/gpu_hbm_calib.py > $OUT/hbm_calibration.jsonl 2> $OUT/hbm_calib.err; tail -4 $OUT/hbm_calibration.jsonl | cut -c1-200 ;;
    side) timeout 900 python scripts/side_bench.py > $OUT/side_bench.jsonl 2> $OUT/side_bench.err; cut -c1-300 $OUT/side_bench.jsonl ;;
    swarm) timeout 300 python scripts/swarm_bench.py --agents 131072 > $OUT/swarm_1gpu.json 2> $OUT/swarm.err; cut -c1-600 $OUT/swarm_1gpu.json; tail -2 $OUT/swarm.err
           timeout 600 python scripts/gpu_swarm_pipeline_ab.py > $OUT/swarm_pipeline_ab.jsonl 2> $OUT/swarm_pipeline_ab.err; cut -c1-160 $OUT/swarm_pipeline_ab.jsonl
           timeout 400 python scripts/gpu_mpc_variants_ab.py > $OUT/mpc_variants_ab.jsonl 2> $OUT/mpc_variants_ab.err; cut -c1-260 $OUT/mpc_variants_ab.jsonl ;;
    fuzz) timeout 1500 python scripts/gpu_fuzz_bitexact.py ${SEED0:-300} ${SEEDS:-20} > $OUT/fuzz_bitexact.txt 2>&1; tail -12 $OUT/fuzz_bitexact.txt ;;
    prof) timeout 2400 bash scripts/gpu_prof.sh $TAG/prof > $OUT/prof.log 2>&1; tail -40 $OUT/prof.log
          timeout 600 bash scripts/gpu_mpc_traffic.sh $TAG/prof > $OUT/mpc_traffic.log 2>&1; tail -4 $OUT/mpc_traffic.log ;;
    *) echo "unknown step $s" ;;
  esac
done
du -sh $OUT
