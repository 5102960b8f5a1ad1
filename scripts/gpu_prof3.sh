# Round-3 profile pass on one MI355X: rocprofv3 kernel stats + PMC (separate passes, never combined with tracing) for the fused EKF
# launch of bench.py and for the DARE / MPC launches in both register layouts.  Usage (through gpurun): bash scripts/gpu_prof3.sh [tag]
TAG=${1:-r03prof}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
K="python $REPO/scripts/prof_kernels3.py"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o ekf -- $B --steps 100 --warmup 10 > $OUT/prof_stats.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc $SQ -d $OUT/pmc_sq -o ekf -- $B --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/side_stats -o side -- $K 10 > $OUT/side_stats.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc $SQ -d $OUT/side_sq -o side -- $K 3 > $OUT/side_sq.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 -d $OUT/side_flop -o side -- $K 3 > $OUT/side_flop.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $OUT/side_sq2 -o side -- $K 3 > $OUT/side_sq2.log 2>&1
cd $REPO
tail -2 $OUT/side_sq.log
python scripts/summarize_prof3.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
