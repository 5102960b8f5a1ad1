# Kernel times of one shard of the mixed swarm round (scripts/swarm_bench.py).  Usage (through gpurun): bash scripts/gpu_swarm_prof.sh [tag]
TAG=${1:-swarmprof}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o swarm -- python $REPO/scripts/swarm_bench.py --agents 131072 --steps 10 > $OUT/swarm.json 2> $OUT/swarm.err
cd $REPO
python - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cut -c1-400 $OUT/swarm.json
find $OUT -name "*.db" -delete
