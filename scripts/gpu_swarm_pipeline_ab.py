#!/usr/bin/env python
"""A/B of the configs[4] round on one GPU's shard (131,072 agents): how many planner launches in flight (depth), and which planner
kernel.  One JSON line per configuration -> profiles/rNN/swarm_pipeline_ab.jsonl.   python scripts/gpu_swarm_pipeline_ab.py [quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # a hardware queue per stream (cpprobotics_amd/swarm.py: want_hw_queues); up to depth 12 here
    import torch

    import bench
    from swarm_bench import mpc_launcher
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    configs = [("product", d) for d in ((1, 6) if quick else (1, 2, 3, 4, 5, 6, 7, 8, 10, 12))]
    if not quick:
        configs += [("refill:128", 4), ("refill:128", 6), ("refill:128", 8), ("refill:256", 8), ("refill:256", 12), ("portfolio", 6)]
    for spec, depth in configs:
        fn, label = mpc_launcher(spec)
        try:
            out = bench.measure_swarm_configs4(dev, 0, 1, rounds=100, warmup=20, depth=depth, mpc_fn=fn, mpc_label=label, blocks=5)
            line = {"mpc": spec, "depth": depth, "round_ms": out["round_ms"], "round_ms_of_every_block": out["round_ms_of_every_block"], "ekf_ms_per_round": out["roofline"]["kernel_ms_per_round"],
                    "ekf_frac_of_8TBps": out["roofline"]["frac"], "ekf_updates_per_s": out["ekf_updates_per_s"],
                    "mpc_solves_per_s": out["mpc_solves_per_s"], "mpc_sweeps": out["mpc_sweeps"], "label": label}
        except Exception as e:
            line = {"mpc": spec, "depth": depth, "error": repr(e)[:300]}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
