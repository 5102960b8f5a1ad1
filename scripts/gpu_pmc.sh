# SQ counters for the fused EKF kernel.  Usage: bash scripts/gpu_pmc.sh tag
TAG=${1:-pmc}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o ekf -- $B > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F64 -d $OUT/pmc_sq2 -o ekf -- $B > $OUT/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o ekf -- $B > $OUT/pmc_grbm.log 2>&1
cd $REPO
python - <<PY
import csv,glob
for tag in ("pmc_sq","pmc_sq2","pmc_grbm"):
    acc={}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ekf_run_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(tag,k,"%.6g"%(sum(v)/len(v)))
PY
find $OUT -name "*.db" -delete
