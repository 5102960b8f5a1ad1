#!/bin/bash
# rocprofv3 passes on one MI355X: kernel stats and PMC counters — always in separate runs, never combined with tracing domains — for
#   ekf    the fused EKF launch of bench.py (kernel stats; FETCH_SIZE / WRITE_SIZE / SQ_* counter passes -> traffic.json)
#   side   the DARE (structured in both layouts, dense-signature, dense kernel) and MPC launches of scripts/prof_kernels.py
#          and the single-step EKF update at 4 M / 1 M vehicles (kernel stats; SQ_* / flop / scalar-memory / FETCH_SIZE / WRITE_SIZE
#          counter passes -> side_counters.json)
#   swarm  one shard of the mixed swarm round (kernel stats)
#   marks  a --marker-trace + --kernel-trace pass of a few host-pointer calls: the roctx ranges crx puts around every entry point
# Usage (through gpurun): bash scripts/gpu_prof.sh TAG [ekf side swarm marks]     (default: ekf side marks)
# (Rounds 1-3: gpu_prof.sh, gpu_prof2.sh, gpu_prof3.sh, gpu_pmc.sh, gpu_side.sh, gpu_swarm_prof.sh — one script now.)
TAG=${1:?usage: gpu_prof.sh TAG [ekf side swarm marks]}; shift
WHAT=${*:-ekf side marks}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
K="python $REPO/scripts/prof_kernels.py"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for w in $WHAT; do
  case $w in
    ekf)
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o ekf -- $B --steps 100 --warmup 10 > $OUT/prof_stats.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc $SQ -d $OUT/pmc_sq -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1 ;;
    side)
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/side_stats -o side -- $K 10 > $OUT/side_stats.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc $SQ -d $OUT/side_sq -o side -- $K 3 > $OUT/side_sq.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 -d $OUT/side_flop -o side -- $K 3 > $OUT/side_flop.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/side_fetch -o side -- $K 2 > $OUT/side_fetch.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/side_write -o side -- $K 2 > $OUT/side_write.log 2>&1
      timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $OUT/side_sq2 -o side -- $K 3 > $OUT/side_sq2.log 2>&1 ;;
    swarm)
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/swarm_stats -o swarm -- python $REPO/scripts/swarm_bench.py --agents 131072 --steps 10 > $OUT/swarm.json 2> $OUT/swarm.err ;;
    marks)
      timeout 300 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $OUT/marks -o marks -- python $REPO/scripts/prof_kernels.py 1 --host-calls > $OUT/marks.log 2>&1 ;;
  esac
done
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
