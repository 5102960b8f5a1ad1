"""Copy the judged summaries of a scripts/gpu_pass.sh pass from gpurun_out/<tag>/ into profiles/<round>/ (tracked); steps that were
not run are skipped.     python scripts/collect_profiles.py r04final r04"""
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KEEP = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
        "Counter_Name", "Counter_Value"]
SIDE = ["mpc_kernel", "mpc_quad_kernel", "mpc_portfolio_kernel", "dare_from_v_kernel", "dare_from_v_quad_kernel", "dare_from_v_masked_kernel", "dare_dense_kernel", "dare_dense_quad_kernel", "ekf_step_kernel", "lqr_closed_loop"]
done = []


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, "prof", sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counters(sub, out, match):
    path = find(sub, "counter_collection.csv")
    if not path:
        return
    rows = [r for r in csv.DictReader(open(path)) if any(m in r["Kernel_Name"] for m in match)]
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.writer(f); w.writerow(KEEP)
        for r in rows:
            w.writerow([r.get(k, "") for k in KEEP])
    done.append(out)


def stats(sub, out, match):
    path = find(sub, "kernel_stats.csv")
    if not path:
        return
    cols = ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")
    rows = [r for r in csv.DictReader(open(path)) if any(m in r["Name"] for m in match)]
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.writer(f); w.writerow(cols)
        for r in rows:
            w.writerow([r.get(k, "") for k in cols])
    done.append(out)


def first_json_line(name, out):
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return
    line = next((l for l in open(path) if l.lstrip().startswith("{")), None)
    if line is None:
        return
    json.loads(line)
    open(os.path.join(dst, out), "w").write(line if line.endswith("\n") else line + "\n")
    done.append(out)


def copy(name, out=None, base=src):
    if os.path.exists(os.path.join(base, name)):
        shutil.copy(os.path.join(base, name), os.path.join(dst, out or os.path.basename(name)))
        done.append(out or os.path.basename(name))


first_json_line("bench.json", "bench_n1.json")
first_json_line("bench_forcedist.json", "bench_forcedist_1rank.json")
first_json_line("bench_oversub2.json", "bench_oversubscribed_2ranks_dry_run.json")
copy("swarm_1gpu.json", "swarm_shard_1gpu.json")
for f in ("side_bench.jsonl", "dare_lanes_ab.jsonl", "mpc_lanes_ab.jsonl", "loop_lanes_ab.jsonl", "mpc_loop_err.jsonl", "tests.log", "smoke.log", "host.txt",
          "fuzz_bitexact.txt", "dare_dense_lanes_ab.jsonl", "dare_refill_ab.jsonl", "hbm_calibration.jsonl"):
    copy(f)
stats("prof_stats", "ekf_kernel_stats.csv", ["ekf_run_kernel", "ekf_simulate_inputs"])
stats("side_stats", "side_kernel_stats.csv", SIDE)
stats("swarm_stats", "swarm_kernel_stats.csv", ["crx::"])
for name in ("fetch", "write", "sq"):
    counters(f"pmc_{name}", f"pmc_{name}_ekf_run_kernel.csv", ["ekf_run_kernel"])
for name, o in (("side_sq", "side_pmc_sq.csv"), ("side_flop", "side_pmc_flop.csv"), ("side_sq2", "side_pmc_sq2.csv"),
                ("side_fetch", "side_pmc_fetch.csv"), ("side_write", "side_pmc_write.csv")):
    counters(name, o, SIDE)
p = os.path.join(src, "prof")
for f in ("summary.txt", "traffic.json", "side_counters.json", "mpc_traffic.json"):
    copy(f, base=p)
for f in ("swarm_pipeline_ab.jsonl", "mpc_variants_ab.jsonl"):
    copy(f)
for f in ("traffic.json", "side_counters.json", "mpc_traffic.json"):        # what bench.py cites as `counters_source`
    if os.path.exists(os.path.join(p, f)):
        shutil.copy(os.path.join(p, f), os.path.join(ROOT, "profiles", f))
print("collected", sorted(done))
