# Full GPU check of a round: tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Usage on the GPU box: bash scripts/gpu_round.sh [tag]
set -x
TAG=${1:-r01}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $OUT/smoke.log; cat $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
# kernel trace + stats of the same command (short run, no cpu baseline / extras)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o ekf -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof_stats.log 2>&1
tail -2 $OUT/prof_stats.log
# PMC passes (separate runs, counters only)
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ekf -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o ekf -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o ekf -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep only small artefacts
find $OUT -name "*.csv" -size +3M -delete
du -sh $OUT
