#!/bin/bash
# HBM traffic and time of the MPC solve's variants in the throughput regime (scripts/prof_mpc_tp.py): rocprofv3 kernel stats, then
# FETCH_SIZE and WRITE_SIZE in separate PMC passes (never combined with tracing), condensed into mpc_traffic.json.
# Usage (through gpurun): bash scripts/gpu_mpc_traffic.sh TAG [agents] [variants: private,tile,refill]
TAG=${1:?usage: gpu_mpc_traffic.sh TAG [agents]}; N=${2:-262144}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
K="python $REPO/scripts/prof_mpc_tp.py $N 2 ${3:-private,tile,tile_refill}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mpc_tp_stats -o mpc -- $K > $OUT/mpc_tp_stats.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/mpc_tp_fetch -o mpc -- $K > $OUT/mpc_tp_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/mpc_tp_write -o mpc -- $K > $OUT/mpc_tp_write.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/mpc_tp_sq -o mpc -- $K > $OUT/mpc_tp_sq.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/mpc_tp_sq2 -o mpc -- $K > $OUT/mpc_tp_sq2.log 2>&1
cd $REPO
python - $OUT $N <<'PY'
import csv, glob, json, os, sys
out, n = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, os.getcwd())
from cpprobotics_amd._lib import kernel_code_hash
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)
res = {}
for sub, key in (("mpc_tp_fetch", "FETCH_SIZE"), ("mpc_tp_write", "WRITE_SIZE"), ("mpc_tp_sq", None), ("mpc_tp_sq2", None)):
    for r in rows(sub + "/**/*counter_collection.csv"):
        k = r.get("Kernel_Name", "")
        if "mpc_" not in k: continue
        name = k.split("(")[0].replace("void crx::", "")
        res.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for extra in ("Scratch_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count"):      # per-dispatch columns of the counter csv
            if r.get(extra) not in (None, ""):
                res[name][extra] = [float(r[extra])]
for r in rows("mpc_tp_stats/**/*kernel_stats.csv"):
    if "mpc_" in r["Name"]:
        name = r["Name"].split("(")[0].replace("void crx::", "")
        res.setdefault(name, {})["avg_ms"] = [float(r["AverageNs"]) / 1e6]
summ = {}
for name, c in res.items():
    a = {k: sum(v) / len(v) for k, v in c.items()}
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        a["fetch_bytes_x2"] = a["FETCH_SIZE"] * 1024 * 2; a["write_bytes"] = a["WRITE_SIZE"] * 1024
        a["hbm_bytes"] = a["fetch_bytes_x2"] + a["write_bytes"]
        if "avg_ms" in a: a["hbm_TB_per_s"] = a["hbm_bytes"] / (a["avg_ms"] * 1e-3) / 1e12
        a["hbm_bytes_per_agent"] = a["hbm_bytes"] / n
    if "SQ_WAVE_CYCLES" in a: a["valu_active_frac"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]
    summ[name] = a
json.dump({"agents": n, "T": 21, "kernel_code_hash": kernel_code_hash("mpc"), "kernels": summ, "note": "FETCH_SIZE in KB, doubled (gfx950 half-count, MI355X_MICROARCH.md); WRITE_SIZE KB"}, open(os.path.join(out, "mpc_traffic.json"), "w"), indent=1)
for k, v in summ.items(): print(k, {a: round(b, 3) if isinstance(b, float) else b for a, b in v.items() if a in ("avg_ms", "hbm_bytes", "hbm_TB_per_s", "hbm_bytes_per_agent", "valu_active_frac", "SQ_INSTS_VMEM", "SQ_INSTS_VALU", "Scratch_Size", "LDS_Block_Size", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS")})
PY
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
