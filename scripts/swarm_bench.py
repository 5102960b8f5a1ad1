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
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--gather", choices=["traj", "final", "none"], default="traj")
    ap.add_argument("--chunks", type=int, default=4, help="launches per round of the chunked trajectory gather")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import cpprobotics_amd as crx
    from cpprobotics_amd import swarm
    from common import ekf_QR, mpc_course_f32

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    n, T = args.agents, args.T
    n_total, n_mpc, Tm = n * world, n // 8, 21
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
    x_hist = torch.empty((T, n, 4), device=dev)
    tind = torch.zeros(n_mpc, dtype=torch.int32, device=dev)

    plan_stream = torch.cuda.Stream(device=dev)
    ekf_done = torch.cuda.Event()
    cg = swarm.ChunkedTrajectoryGather(T, n, 4, args.chunks, dev) if (world > 1 and args.gather == "traj") else None
    st = torch.empty((n_mpc, 4), device=dev)
    v_cmd = 2.5

    def one_round():
        main = torch.cuda.current_stream()
        x.copy_(x0); P.copy_(P0)
        if cg is not None:
            cg.run(lambda c, t0_, t1_, hist: crx.ekf_run(x, P, z[t0_:t1_], ud[t0_:t1_], Q, R, x_hist=hist))
        else:
            crx.ekf_run(x, P, z, ud, Q, R, x_hist=x_hist)
        main.wait_stream(plan_stream)                                 # the planners of the previous round still read st
        st[:, :3].copy_(x[::8, :3])                                   # every eighth agent plans from its estimated pose ...
        st[:, 3] = v_cmd                                              # ... at the commanded speed: the filter's 4th state integrates
                                                                      # the noisy velocity input every step (F(3,3) = 1 and B(3,0) = 1,
                                                                      # src/extended_kalman_filter.cpp:27,34) — a random walk, not a speed
        ekf_done.record(main)
        with torch.cuda.stream(plan_stream):                          # planning overlaps the next round's EKF launches
            plan_stream.wait_event(ekf_done)
            crx.calc_nearest_index(st, dc, tind)
            xref = crx.calc_ref_trajectory(st, dc, tind, Tm)
            sol = crx.mpc_solve(st, xref, Tm)
        out = None
        if world > 1 and args.gather == "final":
            out = swarm.gather_agents(x, n_total)
        return sol, out

    def sync():
        if cg is not None:
            cg.wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(3):
        one_round()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.rounds):
        one_round()
    sync()
    dt = (time.perf_counter() - t0) / args.rounds
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    if rank == 0:
        gb = {"traj": 16.0 * T * n_total, "final": 16.0 * n_total, "none": 0.0}[args.gather] if world > 1 else 0.0
        print(json.dumps({"workload": f"mixed swarm: {n_total} agents over {world} GPU(s), {T} EKF steps + 1/8 of the agents one MPC solve (T=21) per round",
                          "round_ms": dt * 1e3, "ekf_updates_per_s": n_total * T / dt, "mpc_solves_per_s": n_mpc * world / dt,
                          "gather": args.gather if world > 1 else "n/a", "gathered_bytes_per_rank_per_round": gb}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
