"""World-size-2 (and 3, ragged) gloo tests of the swarm sharding / gather path on CPU.

The per-shard compute here is the CPU oracle (test infrastructure) standing in for the HIP launch:
what is under test is the partitioning and the collective — sharded result == unsharded result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, T, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from common import ekf_QR, ekf_agents, ekf_noise
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    u, x0, P0 = ekf_agents(n, 7)                 # global problem, keyed by global agent id
    lo, hi = swarm.shard_range(n, rank, world)
    # the noise each shard draws for ITS agents: counter-based, keyed by (seed, global agent id, step) — crx_philox.h
    w = oracle.normal_draws(hi - lo, T, agent0=lo, seed=8)
    z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u[lo:hi], x0[lo:hi], x0[lo:hi], w)
    x, P, xh, _ = oracle.ekf_run(x0[lo:hi], P0[lo:hi], z, ud, Q, R)
    xg = swarm.gather_agents(torch.from_numpy(x), n)
    hg = swarm.gather_time_major(torch.from_numpy(xh), n)
    if n % world == 0:
        # the chunked, overlapped trajectory gather: 4 launches of T/4 steps, state carried over, one async all-gather each
        cg = swarm.ChunkedTrajectoryGather(T, hi - lo, 4, 4, "cpu")
        xc, Pc = x0[lo:hi].copy(), P0[lo:hi].copy()

        def launch(c, t0, t1, hist):
            nonlocal xc, Pc
            xc, Pc, h, _ = oracle.ekf_run(xc, Pc, z[t0:t1], ud[t0:t1], Q, R)
            hist.copy_(torch.from_numpy(h))
        hc = cg.run(launch).time_major()
        assert torch.equal(hc, hg) and cg.gathered.shape == (4, world, T // 4, hi - lo, 4)
        assert torch.equal(cg.gathered[1, rank, 2], torch.from_numpy(xh[T // 4 + 2]))     # [chunk][rank][t][agent] index map
        # the per-step final-estimate gather of bench.py: ring of 4 buffers, the caller waits once per lap
        rgat = swarm.RingGather(n, (4,), 4, "cpu")
        nl = hi - lo
        state = [torch.empty((nl, 4)) for _ in range(4)]                 # the caller's ring of state buffers
        expect = lambda step: torch.arange(n * 4, dtype=torch.float32).reshape(n, 4) * 0.5 + step
        for step in range(11):
            rgat.begin_step(step)
            if step and step % 4 == 0:                                    # after the lap's wait: the whole previous lap is there
                for j in range(4):
                    assert torch.equal(rgat.out[j], expect(step - 4 + j))
            state[rgat.slot(step)].copy_(expect(step)[lo:hi])             # "launch": overwrite the slot's state
            got = rgat.gather(step, state[rgat.slot(step)])
            assert got is rgat.out[step % 4]
        rgat.wait()
        for step in (8, 9, 10):
            assert torch.equal(rgat.out[step % 4], expect(step))
    # tracking swarm: each rank runs the closed LQR loop on its shard of the agents (shared course), results gathered
    from common import lqr_course, tracking_agents
    course, goal = lqr_course()
    st = tracking_agents(n, tuple(c[:100] for c in course), 11, spread=0.3)
    s1, ticks, *_ = oracle.lqr_closed_loop(st[lo:hi], course, goal, dim=5, max_ticks=300)
    sg = swarm.gather_agents(torch.from_numpy(s1), n)
    tg = swarm.gather_agents(torch.from_numpy(ticks), n)
    # planner swarm: each rank plans for its shard of the agents (shared course and obstacles)
    fr = oracle.frenet_run(_frenet_states(n)[lo:hi], oracle.frenet_spline_build(), [70.465, 0.007], 3)
    fg = swarm.gather_agents(torch.from_numpy(fr["state"]), n)
    if rank == 0:
        q.put((xg.numpy(), hg.numpy(), sg.numpy(), tg.numpy(), fg.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _frenet_states(n):
    rng = np.random.default_rng(13)
    return np.stack([rng.uniform(0.0, 60.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-2.0, 2.0, n), rng.uniform(-0.5, 0.5, n),
                     rng.uniform(-0.3, 0.3, n)], axis=1).astype(np.float32)


def _collect(procs, q, budget=600.0):
    """The rank-0 result of a group of spawned ranks.  Polls so that a rank that died is reported at once (with its exit code)
    instead of after the whole budget; a slow first `import torch` in a fresh container alone can take minutes."""
    import queue, time
    t0 = time.time()
    while True:
        try:
            out = q.get(timeout=2.0)
            break
        except queue.Empty:
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > budget:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"ranks failed or timed out after {time.time() - t0:.0f} s: dead = {dead}")
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    return out



def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n", [(2, 64), (2, 33), (3, 50)])
def test_sharded_equals_unsharded(world, n, oracle_mod):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import ekf_QR, ekf_agents, ekf_noise
    T = 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    xg, hg, sg, tg, fg = _collect(procs, q)
    Q, R = ekf_QR()
    u, x0, P0 = ekf_agents(n, 7)
    w = oracle_mod.normal_draws(n, T, agent0=0, seed=8)        # one "GPU" draws for everybody: the shards must have seen the same bytes
    z, ud, _, _, _, _ = oracle_mod.ekf_simulate_inputs(u, x0, x0, w)
    x, P, xh, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    assert np.array_equal(xg, x) and np.array_equal(hg, xh)
    from common import lqr_course, tracking_agents
    course, goal = lqr_course()
    st = tracking_agents(n, tuple(c[:100] for c in course), 11, spread=0.3)
    s1, ticks, *_ = oracle_mod.lqr_closed_loop(st, course, goal, dim=5, max_ticks=300)
    assert np.array_equal(sg, s1) and np.array_equal(tg, ticks)
    fr = oracle_mod.frenet_run(_frenet_states(n), oracle_mod.frenet_spline_build(), [70.465, 0.007], 3)
    assert np.array_equal(fg, fr["state"])


def _swarm_worker(rank, world, port, n, T, q):
    """One round of the mixed EKF + MPC swarm (cpprobotics_amd/swarm.py: MixedSwarmRound, what scripts/swarm_bench.py runs on the
    GPUs) with the CPU oracle standing in for the launches: chunked trajectory gather + planners on every eighth vehicle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from common import ekf_QR, ekf_agents, mpc_course_f32
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    u, x0, P0 = ekf_agents(n, 17)
    x0[:, :3] = np.stack([course[0][:n], course[1][:n], course[2][:n]], axis=1)        # vehicles start along the course
    lo, hi = swarm.shard_range(n, rank, world)
    w = oracle.normal_draws(hi - lo, T, agent0=lo, seed=18)
    z, ud, *_ = oracle.ekf_simulate_inputs(u[lo:hi], x0[lo:hi], x0[lo:hi], w)
    st = dict(x=x0[lo:hi].copy(), P=P0[lo:hi].copy())

    def ekf_launch(c, t0, t1, hist):
        if c == 0:
            st["x"], st["P"] = x0[lo:hi].copy(), P0[lo:hi].copy()
        st["x"], st["P"], h, _ = oracle.ekf_run(st["x"], st["P"], z[t0:t1], ud[t0:t1], Q, R)
        hist.copy_(torch.from_numpy(h))

    def plan_launch(est, slot):
        e = est.numpy().copy(); e[:, 3] = 2.5
        tind = oracle.calc_nearest_index(e, course)[0].astype(np.int32)
        xref, _ = oracle.calc_ref_trajectory(e, course, tind, 6)
        return oracle.mpc_solve(e, xref, 6)[0]

    outs = []
    for kind in ("traj", "final"):
        rnd = swarm.MixedSwarmRound(hi - lo, T, 4, 4, 8, "cpu", ekf_launch, lambda: torch.from_numpy(st["x"]), plan_launch, gather=kind, n_total=n,
                                    depth=2 if kind == "final" else 1)
        for _ in range(2):                       # two rounds: the second one reuses every buffer of the first
            rnd.run()
        rnd.wait()
        full = rnd.trajectory_time_major() if kind == "traj" else rnd.final
        plans = swarm.gather_agents(torch.from_numpy(np.ascontiguousarray(rnd.plans)), n // 8)
        outs.append((full.numpy(), plans.numpy(), rnd.gathered_bytes_per_rank()))
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_swarm_round_sharded_equals_unsharded(oracle_mod):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import ekf_QR, ekf_agents, mpc_course_f32
    world, n, T = 2, 64, 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_swarm_worker, args=(r, world, port, n, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    (traj, plans_t, bytes_t), (final, plans_f, bytes_f) = _collect(procs, q)
    o = oracle_mod
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    u, x0, P0 = ekf_agents(n, 17)
    x0[:, :3] = np.stack([course[0][:n], course[1][:n], course[2][:n]], axis=1)
    w = o.normal_draws(n, T, agent0=0, seed=18)
    z, ud, *_ = o.ekf_simulate_inputs(u, x0, x0, w)
    x, P, xh, _ = o.ekf_run(x0, P0, z, ud, Q, R)
    assert np.array_equal(traj, xh) and np.array_equal(final, x)
    e = x[::8].copy(); e[:, 3] = 2.5
    tind = o.calc_nearest_index(e, course)[0].astype(np.int32)
    xref, _ = o.calc_ref_trajectory(e, course, tind, 6)
    sol = o.mpc_solve(e, xref, 6)[0]
    assert np.array_equal(plans_t, sol) and np.array_equal(plans_f, sol)
    assert bytes_t == 16 * T * n and bytes_f == 16 * n


def test_shard_range_partitions():
    from cpprobotics_amd import swarm
    for n in (0, 1, 7, 64, 65536, 1048576):
        for world in (1, 2, 3, 8):
            r = [swarm.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == swarm.shard_sizes(n, world)
