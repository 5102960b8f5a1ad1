"""configs[4] (one GPU's shard: 131,072 vehicles x 100 EKF steps + 16,384 MPC solves per round, depth 6 on 16 hardware queues) with the
planner's working-set layout forced: private memory / tile / tile with refilled lanes.  Prints bench.py's swarm block per variant.
usage (gpurun): python scripts/gpu_swarm_store_ab.py > gpurun_out/<tag>/swarm_store_ab.jsonl"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
from cpprobotics_amd import experimental as X

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
variants = {
    "product": None,
    "private": lambda est, xref, Tm, out: X.mpc_solve_store(est, xref, Tm, 0, out=out),
    "tile": lambda est, xref, Tm, out: X.mpc_solve_store(est, xref, Tm, 1, out=out),
    "tile_lite": lambda est, xref, Tm, out: X.mpc_solve_store(est, xref, Tm, 3, out=out),
    "tile_refill_64_16": lambda est, xref, Tm, out: X.mpc_solve_tile_refill(est, xref, Tm, 64 * 4, 16, out=out),
    "tile_refill_128_16": lambda est, xref, Tm, out: X.mpc_solve_tile_refill(est, xref, Tm, 64 * 8, 16, out=out),
}
want = sys.argv[1:] or list(variants)
for name in want:
    r = bench.measure_swarm_configs4(dev, 0, 1, depth=int(os.environ.get('DEPTH', '6')), rounds=40, warmup=12, blocks=3, mpc_fn=variants[name], mpc_label=name)
    print(json.dumps({"variant": name, "round_ms": r["round_ms"], "blocks": r["round_ms_of_every_block"], "ekf_frac_sharing": r["roofline"]["frac"],
                      "ekf_frac_alone": r["roofline"]["launch_alone"]["frac"], "mpc_solves_per_s": r["mpc_solves_per_s"],
                      "host_issue_ms": r["host_issue_ms_per_round"], "converged": r["mpc_sweeps"]["converged_frac"]}), flush=True)
