"""Copy the judged summaries of a scripts/gpu_final2.sh pass from gpurun_out/<tag>/ into profiles/<round>/ (tracked).
   python scripts/collect_profiles.py r02final3 r02"""
import csv, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
KEEP = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]


def counters(path, out, match):
    rows = [r for r in csv.DictReader(open(path)) if any(m in r["Kernel_Name"] for m in match)]
    with open(out, "w", newline="") as f:
        w = csv.writer(f); w.writerow(KEEP)
        for r in rows:
            w.writerow([r[k] for k in KEEP])
    return len(rows)


def first_json_line(path, out):
    line = next(l for l in open(path) if l.lstrip().startswith("{"))
    json.loads(line)
    open(out, "w").write(line if line.endswith("\n") else line + "\n")


def stats(path, out, match=None):
    rows = list(csv.DictReader(open(path)))
    if match:
        rows = [r for r in rows if any(m in r["Name"] for m in match)]
    with open(out, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")])


first_json_line(os.path.join(src, "bench.json"), os.path.join(dst, "bench_n1.json"))
first_json_line(os.path.join(src, "bench_forcedist.json"), os.path.join(dst, "bench_forcedist_1rank.json"))
shutil.copy(os.path.join(src, "side_bench.jsonl"), os.path.join(dst, "side_bench.jsonl"))
shutil.copy(os.path.join(src, "swarm_1gpu.json"), os.path.join(dst, "swarm_shard_1gpu.json"))
for f in ("dare_lanes_ab.jsonl", "mpc_lanes_ab.jsonl", "loop_lanes_ab.jsonl", "mpc_loop_err.jsonl", "tests.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
shutil.copy(os.path.join(src, "host.txt"), os.path.join(dst, "host.txt"))
p = os.path.join(src, "prof")
stats(os.path.join(p, "prof_stats", "ekf_kernel_stats.csv"), os.path.join(dst, "ekf_kernel_stats.csv"), ["ekf_run_kernel", "ekf_simulate_inputs"])
stats(os.path.join(p, "side_stats", "side_kernel_stats.csv"), os.path.join(dst, "side_kernel_stats.csv"), ["mpc_kernel", "mpc_quad_kernel", "dare_from_v_kernel", "dare_from_v_quad_kernel", "lqr_closed_loop"])
for name in ("fetch", "write", "sq"):
    counters(os.path.join(p, f"pmc_{name}", "ekf_counter_collection.csv"), os.path.join(dst, f"pmc_{name}_ekf_run_kernel.csv"), ["ekf_run_kernel"])
for name, o in (("side_sq", "side_pmc_sq.csv"), ("side_flop", "side_pmc_flop.csv"), ("side_sq2", "side_pmc_sq2.csv")):
    counters(os.path.join(p, name, "side_counter_collection.csv"), os.path.join(dst, o), ["mpc_kernel", "mpc_quad_kernel", "dare_from_v_kernel", "dare_from_v_quad_kernel", "lqr_closed_loop"])
for f in ("summary.txt", "traffic.json", "side_counters.json"):
    shutil.copy(os.path.join(p, f), os.path.join(dst, f))
shutil.copy(os.path.join(p, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
shutil.copy(os.path.join(p, "side_counters.json"), os.path.join(ROOT, "profiles", "side_counters.json"))
print("collected", sorted(os.listdir(dst)))
