#!/usr/bin/env python
"""BASELINE.json configs[4]: a mixed EKF + MPC swarm sharded over the GPUs of one node — the command-line form of the block
bench.py prints as `extra.swarm_configs4` (N = 1) / `multi_gpu.swarm_configs4` (N > 1).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/swarm_bench.py
  python scripts/swarm_bench.py --agents 131072 --depth 4          # one GPU, one shard of the 1,048,576-agent swarm

The round itself is cpprobotics_amd/swarm.py: SwarmShard / MixedSwarmRound (its docstring describes the streams and the slot ring), the
measurement bench.py: measure_swarm_configs4.  --mpc selects the planner launch for A/Bs: `product` (crx_mpc_solve_batch_dev),
`portfolio`, `refill:<agents per wave>[:<hold>]` (the lane-refilling A/B kernel, include/crx_experimental.h).
Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mpc_launcher(spec):
    """-> (mpc_fn(est, xref, Tm, out), label) for a --mpc specification."""
    import cpprobotics_amd as crx
    kind, *rest = spec.split(":")
    if kind == "product":
        return None, "crx_mpc_solve_batch_dev"
    if kind == "portfolio":
        return (lambda est, xref, Tm, out: crx.mpc_solve(est, xref, Tm, portfolio=True, out=out)), "crx_mpc_solve_portfolio_batch_dev"
    from cpprobotics_amd import experimental as X
    if kind == "refill":
        apw, hold = int(rest[0]), int(rest[1]) if len(rest) > 1 else 16
        return (lambda est, xref, Tm, out: X.mpc_solve_refill(est, xref, Tm, apw, hold, out=out)), f"mpc_refill_kernel, {apw} agents per wave, hold {hold}"
    raise SystemExit(f"--mpc {spec}: product | portfolio | refill:<apw>[:<hold>]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=131072, help="agents per GPU (1,048,576 / 8)")
    ap.add_argument("--T", type=int, default=100, help="EKF steps per round")
    ap.add_argument("--gpus", type=int, default=None, help="must equal WORLD_SIZE when given (bench.py's contract)")
    ap.add_argument("--steps", "--rounds", dest="steps", type=int, default=60, help="timed rounds per block (three blocks, the median is reported)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gather", choices=["traj", "final", "none"], default="traj")
    ap.add_argument("--chunks", type=int, default=4, help="EKF launches per round where the trajectory gather overlaps them (N > 1)")
    ap.add_argument("--depth", type=int, default=6, help="planner launches in flight")
    ap.add_argument("--mpc", default="product")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(16, args.depth + 10)))     # a hardware queue per stream: cpprobotics_amd/swarm.py: want_hw_queues
    import torch
    import torch.distributed as dist

    import bench
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus is None or args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    fn, label = mpc_launcher(args.mpc)
    out = bench.measure_swarm_configs4(dev, rank, world, agents=args.agents, T=args.T, rounds=args.steps, warmup=args.warmup, depth=args.depth,
                                       gather=args.gather, chunks=args.chunks, mpc_fn=fn, mpc_label=label)
    if rank == 0:
        line = {"metric": "EKF updates/s of a mixed EKF + MPC swarm (BASELINE.json configs[4]); MPC solves/s beside it",
                "value": out["ekf_updates_per_s"], "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": out["round_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (EKF), f64 (MPC)", "data": "synthetic", "config": {"workload": out["workload"]},
                "secondary": {"metric": "MPC horizon solves/s (T = 21) of the same rounds", "value": out["mpc_solves_per_s"], "unit": "solves/s"},
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None}
        line.update(out)
        print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
