# bench + rocprofv3 only (no tests).  Usage: bash scripts/gpu_prof.sh [tag]

TAG=${1:-r01}
REPO=$GRAFT_REPO_ROOT
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o ekf -- $B --steps 100 --warmup 10 > $OUT/prof_stats.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_grbm.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
