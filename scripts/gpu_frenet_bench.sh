TAG=${1:-frenet}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python scripts/side_bench.py --frenet-only > $OUT/side_bench_frenet.jsonl 2> $OUT/side_bench.err; cat $OUT/side_bench_frenet.jsonl; tail -3 $OUT/side_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o frenet -- python $REPO/scripts/side_bench.py --frenet-only --quick > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o frenet -- python $REPO/scripts/side_bench.py --frenet-only --quick > $OUT/pmc.log 2>&1
cd $REPO
python - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "crx::" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
acc={}
for f in glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "crx::" in k: acc.setdefault((k[:60],r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "%.6g"%(sum(v)/len(v)), len(v))
PY
find $OUT -name "*.db" -delete
