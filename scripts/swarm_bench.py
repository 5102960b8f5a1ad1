#!/usr/bin/env python
"""BASELINE.json configs[4]: a mixed EKF + MPC swarm sharded over the GPUs of one node.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/swarm_bench.py
  python scripts/swarm_bench.py --agents 131072          # one GPU, one shard of the 1,048,576-agent swarm

Every rank owns a contiguous shard of the agents (no data-path collective: agents are independent).  Per round:
  1. all vehicles of the shard run T fused EKF steps (crx_ekf_run_batch_dev), estimated trajectory [T][n][4] out;
  2. one agent in eight also plans: calc_ref_trajectory on the shared course from its estimated state, then
     mpc_solve over N = 20 control intervals (T = 21 knots);
  3. the estimated trajectories are concatenated over the ranks — the exchange step north_star names — chunked: the T-step
     launch is cut into --chunks launches and the all-gather of chunk k (RCCL's stream, xGMI) overlaps the compute of chunk
     k+1 (cpprobotics_amd/swarm.py: ChunkedTrajectoryGather).  `--gather final` gathers the final estimates only.
The EKF launches and the planning kernels run on two HIP streams: the planners of round r (which need only the final
estimates) overlap the EKF launches of round r+1.  Every input is keyed by the GLOBAL agent id (Philox draws, numpy
generators over the whole swarm), so a shard computes what the whole swarm would have computed for its agents.
Prints one JSON line on rank 0: EKF updates/s, MPC solves/s, end-to-end rounds/s, bytes gathered."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=131072, help="agents per GPU (1,048,576 / 8)")
    ap.add_argument("--T", type=int, default=100, help="EKF steps per round")
    ap.add_argument("--gpus", type=int, default=None, help="must equal WORLD_SIZE when given (bench.py's contract)")
    ap.add_argument("--steps", "--rounds", dest="steps", type=int, default=5, help="timed rounds")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gather", choices=["traj", "final", "none"], default="traj")
    ap.add_argument("--chunks", type=int, default=4, help="launches per round of the chunked trajectory gather")
    ap.add_argument("--mpc-portfolio", action="store_true",
                    help="the planners through crx_mpc_solve_portfolio_batch_dev (four solver variants per agent, the first to converge wins)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import cpprobotics_amd as crx
    from cpprobotics_amd import swarm
    from common import ekf_QR, mpc_course_f32

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus is None or args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    n, T = args.agents, args.T
    n_total, n_mpc, Tm = n * world, (n + 7) // 8, 21
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dc = crx.Course.from_numpy(course, device=dev)
    # agent parameters keyed by the global agent id: drawn for the whole swarm, this rank keeps its slice
    ci = torch.from_numpy(np.random.default_rng(99).integers(0, len(course[0]) - 30, n_total)[rank * n:(rank + 1) * n]).to(dev)
    cx, cy, cyaw = (torch.from_numpy(a).to(dev) for a in course[:3])
    x0 = torch.stack([cx[ci], cy[ci], cyaw[ci], torch.full((n,), 2.5, device=dev)], dim=1).contiguous()   # 2.5 m/s = v_cmd
    u_true = torch.stack([torch.full((n,), 0.0, device=dev), torch.zeros(n, device=dev)], dim=1).contiguous()  # (accel, yaw rate)
    w = crx.normal_draws(n, T, agent0=rank * n, seed=99, device=dev)
    z, ud = crx.ekf_simulate_inputs(u_true, x0.clone(), x0.clone(), w)
    del w
    P0 = torch.eye(4, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
    x, P = x0.clone(), P0.clone()
    tind = torch.zeros(n_mpc, dtype=torch.int32, device=dev)
    v_cmd = 2.5
    ekf_evs = []

    def ekf_launch(c, t0_, t1_, hist):
        if c == 0:
            x.copy_(x0); P.copy_(P0)
        ev = ekf_evs[-1] if ekf_evs and len(ekf_evs[-1]) < 2 * args.chunks else None
        if ev is not None:
            e0 = torch.cuda.Event(enable_timing=True); e0.record(); ev.append(e0)
        crx.ekf_run(x, P, z[t0_:t1_], ud[t0_:t1_], Q, R, x_hist=hist)
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev.append(e1)

    def plan_launch(est):
        est[:, 3] = v_cmd            # the planners take the estimated pose at the commanded speed: the filter's 4th state integrates the
                                     # noisy velocity input every step (F(3,3) = 1 and B(3,0) = 1, src/extended_kalman_filter.cpp:27,34)
        crx.calc_nearest_index(est, dc, tind)
        xref = crx.calc_ref_trajectory(est, dc, tind, Tm)
        return crx.mpc_solve(est, xref, Tm, portfolio=args.mpc_portfolio)

    rnd = swarm.MixedSwarmRound(n, T, 4, args.chunks, 8, dev, ekf_launch, lambda: x, plan_launch, gather=args.gather, n_total=n_total)

    def sync():
        rnd.wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rnd.run()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ekf_evs.append([])
        rnd.run()
    sync()
    dt = (time.perf_counter() - t0) / args.steps
    ekf_ms = sum(sum(a.elapsed_time(b) for a, b in zip(ev[0::2], ev[1::2])) for ev in ekf_evs) / max(1, len(ekf_evs))
    rates = torch.tensor([n * T / dt], dtype=torch.float64, device=dev)
    per_rank = [float(rates.item())]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
        allr = [torch.zeros_like(rates) for _ in range(world)]
        dist.all_gather(allr, rates)
        per_rank = [float(r.item()) for r in allr]
    if rank == 0:
        algo_bytes = 32.0 * n * T + 160.0 * n                       # the EKF launches of one round on one GPU
        achieved = algo_bytes / (ekf_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "EKF updates/s of a mixed EKF + MPC swarm (BASELINE.json configs[4]); MPC solves/s beside it",
            "value": n_total * T / dt, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (EKF), f64 (MPC)",
            "data": "synthetic",
            "config": {"workload": f"mixed swarm: {n_total} agents over {world} GPU(s), {T} EKF steps for every agent + one MPC solve (T = 21) "
                                   f"for every eighth agent per round; gather = {args.gather if world > 1 else 'n/a'}",
                       "agents_per_gpu": n, "ekf_steps_per_round": T, "mpc_agents_per_gpu": n_mpc, "chunks": args.chunks,
                       "mpc_solver": "four-variant portfolio" if args.mpc_portfolio else "single"},
            "secondary": {"metric": "MPC horizon solves/s (T = 21) of the same rounds", "value": n_mpc * world / dt, "unit": "solves/s"},
            "roofline": {"bound": "valu", "kernel": "crx::ekf_run_kernel (the round's EKF launches)", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "kernel_ms_per_round": ekf_ms, "algorithmic_bytes_per_round": algo_bytes,
                         "traffic": None},
            "multi_gpu": {"ranks": world, "per_rank_ekf_updates_per_s": per_rank,
                          "gather": args.gather if world > 1 else "n/a", "gathered_bytes_per_rank_per_round": rnd.gathered_bytes_per_rank(),
                          "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 else None,
                          "layout": "[chunk][rank][t][agent][4] (cpprobotics_amd/swarm.py: ChunkedTrajectoryGather)"},
            "round_ms": dt * 1e3, "ekf_updates_per_s": n_total * T / dt, "mpc_solves_per_s": n_mpc * world / dt}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
