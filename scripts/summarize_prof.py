"""Condense rocprofv3 output directories (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("prof_stats/**/*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    print(f"== {tag} (per-dispatch counters, ekf_run_kernel only) ==")
    acc = {}
    for f, r in rows(f"{tag}/**/*counter_collection.csv"):
        if "ekf_run_kernel" not in r.get("Kernel_Name", ""):
            continue
        key = r["Counter_Name"]
        acc.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k}: n={len(v)} mean={sum(v) / len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
