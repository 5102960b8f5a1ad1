"""The configs[4] parity checks of tests/test_swarm_gpu.py, runnable as a script: `python tests/swarm_checks.py <check> [args]`.

They run in a process of their own because the number of hardware queues is fixed when the HIP runtime first touches the device
(GPU_MAX_HW_QUEUES, cpprobotics_amd/swarm.py: want_hw_queues): inside the pytest process, which other GPU tests have initialised long
before, every SwarmShard would silently run on the default 4 queues — not the configuration bench.py measures (depth 6 on 16 queues,
VERDICT r5 "what's weak" 1b).  Every check asserts `shard.hw_queues >= depth + 2` first.

Reference shape: one pass of /root/reference/src/extended_kalman_filter.cpp:171-188 for every vehicle, one pass of
/root/reference/src/model_predictive_control.cpp:371-385 for every eighth."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from common import ekf_QR, floored_rel_err, mpc_course_f32, mpc_solve_threads  # noqa: E402

TM = 21


def oracle_round(o, shard, s, agents):
    """The oracle's round on the given (sorted) agent subset with input set s: -> x_hist [T, m, 4], x, P."""
    Q, R = ekf_QR()
    idx = np.asarray(agents)
    z, ud = shard.z[s][:, idx].cpu().numpy(), shard.ud[s][:, idx].cpu().numpy()
    x0, P0 = shard.x0[idx].cpu().numpy(), shard.P0[idx].cpu().numpy()
    x, P, xh, _ = o.ekf_run(x0, P0, np.ascontiguousarray(z), np.ascontiguousarray(ud), Q, R)
    return xh, x, P


def oracle_plans(o, course, est, v_cmd):
    e = est.copy()
    e[:, 3] = np.float32(v_cmd)
    tind = o.calc_nearest_index(e, course)[0].astype(np.int32)
    xref, _ = o.calc_ref_trajectory(e, course, tind, TM)
    so, sto, co = mpc_solve_threads(o, e, xref, TM)
    return e, xref, so, sto, co


def plan_errors(b, rows, e, xref, so, sto, co):
    """The planner comparison as numbers (bench.py prints them; check_plans asserts on them)."""
    sd, std, cd = b["sol"].cpu().numpy()[rows], b["status"].cpu().numpy()[rows], b["cost"].cpu().numpy()[rows]
    conv = (sto & 1) == 1
    crel = np.abs(cd - co) / np.maximum(np.abs(co), 1.0)
    return {"xref_bit_identical": bool(np.array_equal(b["xref"].cpu().numpy()[rows], xref)),
            "status_equal": bool(np.array_equal(std & 3, sto & 3)),
            "status_mismatches": np.flatnonzero((std & 3) != (sto & 3))[:8].tolist(),
            "max_sweep_count_diff": int(np.abs((std >> 8) - (sto >> 8)).max(initial=0)),
            "converged_frac_twin": float(conv.mean()) if conv.size else 1.0,
            "sol_max_rel_err_floored": floored_rel_err(sd[conv], so[conv], 1.0),
            "cost_max_rel_err_converged": float(crel[conv].max(initial=0.0)),
            "cost_max_rel_err": float(crel.max(initial=0.0)), "plans_checked": int(len(sto))}


def check_plans(b, rows, e, xref, so, sto, co):
    r = plan_errors(b, rows, e, xref, so, sto, co)
    assert r["xref_bit_identical"], "calc_ref_trajectory differs from the oracle"
    assert r["status_equal"], f"status differs for planners {r['status_mismatches']}"
    assert r["max_sweep_count_diff"] <= 1
    assert r["converged_frac_twin"] >= 0.95
    assert r["sol_max_rel_err_floored"] <= 1e-6
    assert r["cost_max_rel_err_converged"] <= 1e-9 and r["cost_max_rel_err"] <= 1e-6
    return r


def _queues_ok(shard, depth):
    assert shard.hw_queues is not None and shard.hw_queues >= depth + 2, \
        f"SwarmShard runs on {shard.hw_queues} hardware queues: depth {depth} needs {depth + 2}"


def mixed_rounds(crx, o, depth, n=8192, T=100, rounds=5):
    """`n` vehicles x T EKF steps, n/8 planners, `rounds` consecutive rounds on three different measurement sets with `depth` planner
    launches in flight: every round's history / final state / covariance bit for bit, every plan against the twin."""
    import torch
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    shard = swarm.SwarmShard(n, T, course, Q, R, dev, depth=depth, input_sets=3, seed=7)
    _queues_ok(shard, depth)
    hists, finals, plans = [], [], []
    for r in range(rounds):
        shard.run()
        # snapshot this round's EKF outputs on the launch stream (the next round resets the state and reuses the history buffer);
        # the planners of up to `depth` rounds keep running on their own streams meanwhile
        hists.append(shard.rnd.trajectory_time_major())
        finals.append((shard.x.clone(), shard.P.clone()))
        # a slot's plans stay valid until round r + depth overwrites them: copy them on the slot's own stream, in stream order
        slot = r % depth
        with torch.cuda.stream(shard.rnd.plan_streams[slot]):
            plans.append({k: v.clone() for k, v in shard.slots[slot].items()})
    shard.wait()
    torch.cuda.synchronize()
    all_agents = np.arange(n)
    for r in range(rounds):
        xh, x, P = oracle_round(o, shard, r % 3, all_agents)
        assert np.array_equal(hists[r].cpu().numpy(), xh), f"round {r}: EKF history differs from the oracle"
        assert np.array_equal(finals[r][0].cpu().numpy(), x) and np.array_equal(finals[r][1].cpu().numpy(), P)
        e, xref, so, sto, co = oracle_plans(o, course, x[::8], shard.v_cmd)
        check_plans(plans[r], slice(None), e, xref, so, sto, co)
    # the three measurement sets really differ (otherwise the race check above would be vacuous)
    assert not np.array_equal(finals[0][0].cpu().numpy(), finals[1][0].cpu().numpy())


def full_shard(crx, o, depth, rounds):
    """One GPU's shard of the 1,048,576-agent swarm — 131,072 vehicles, 16,384 planners — `rounds` rounds with `depth` planner launches
    in flight (bench.py's measured configuration is depth 6): the last round's history and final state of every 64th vehicle bit for
    bit, every 16th planner against the twin."""
    import torch
    from cpprobotics_amd import swarm
    n, T = 131072, 100
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    shard = swarm.SwarmShard(n, T, course, Q, R, dev, depth=depth, input_sets=2, seed=99)
    _queues_ok(shard, depth)
    for r in range(rounds):
        shard.run()
    hist = shard.rnd.trajectory_time_major()
    xf = shard.x.clone()
    shard.wait()
    torch.cuda.synchronize()
    last = rounds - 1
    agents = np.arange(0, n, 64)
    xh, x, P = oracle_round(o, shard, last % 2, agents)
    assert np.array_equal(hist[:, agents].cpu().numpy(), xh)
    assert np.array_equal(xf[agents].cpu().numpy(), x)
    # planner j plans for vehicle 8 j: every 16th planner = every 128th vehicle = every second sampled agent
    e, xref, so, sto, co = oracle_plans(o, course, x[::2], shard.v_cmd)
    rows = np.arange(0, shard.n_plan, 16)
    check_plans(shard.rnd.plans_of(last), rows, e, xref, so, sto, co)
    st = shard.rnd.plans_of(last)["status"].cpu().numpy()
    assert ((st & 1) == 1).mean() > 0.99 and not np.any(st & 2)


def one_ekf_launch(crx, o):
    """A single process has nothing to overlap the chunks with: the round must issue ONE fused EKF launch (round 4 issued four)."""
    import torch
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    shard = swarm.SwarmShard(2048, 40, course, Q, R, torch.device("cuda", 0), depth=2, chunks=4, record_ekf_events=True)
    shard.run(); shard.wait()
    torch.cuda.synchronize()
    assert shard.rnd.chunks == 1 and shard.rnd.requested_chunks == 4 and len(shard.ekf_events) == 1


def shared_planner_streams(crx, o):
    """Round objects built one after another must not each bring fresh planner streams (every hardware queue a solver launch with
    private memory has run on keeps its reservation; and crx_mpc_solve_batch_dev refuses a 13th distinct stream)."""
    import torch
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    seen = set()
    for depth in (3, 6, 2, 6):
        shard = swarm.SwarmShard(1024, 20, course, Q, R, dev, depth=depth)
        shard.run(); shard.wait()
        seen |= {s.cuda_stream for s in shard.rnd.plan_streams}
        assert [s.cuda_stream for s in shard.rnd.plan_streams] == [s.cuda_stream for s in swarm.planner_streams(dev, depth)]
    torch.cuda.synchronize()
    assert len(seen) == 6


def c_round(crx, o, depth, rounds, n=8192, T=100):
    """crx_swarm_round_dev (csrc/api_swarm.inl), the round issued by one C call, against the Python round (swarm.SwarmShard, itself held
    to the oracle by mixed_rounds): same start states, same measurement sets round by round, `depth` planner launches in flight — every
    byte of the EKF history, the final state and of every plan buffer (est, xref, sol, status, cost) must be equal."""
    import torch
    from cpprobotics_amd import cswarm, swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    shard = swarm.SwarmShard(n, T, course, Q, R, dev, depth=depth, input_sets=3, seed=7)
    _queues_ok(shard, depth)
    assert cswarm.hw_queues() == int(os.environ["GPU_MAX_HW_QUEUES"])
    ref = []
    for r in range(rounds):
        shard.run()
        h = shard.rnd.trajectory_time_major(); xf = shard.x.clone()
        slot = r % depth
        with torch.cuda.stream(shard.rnd.plan_streams[slot]):
            ref.append((h, xf, {k: v.clone() for k, v in shard.slots[slot].items()}))
    shard.wait()
    torch.cuda.synchronize()
    # depth 1: the object's own slot stream; deeper: the caller's streams (crx_swarm_config.planner_streams), here the Python round's
    cs = cswarm.CSwarm(shard.x0, shard.P0, shard.dc, Q, R, T, Tm=TM, plan_every=8, depth=depth, v_cmd=shard.v_cmd,
                       streams=swarm.planner_streams(dev, depth) if depth > 1 else None)
    got = []
    hist = [torch.empty((T, n, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    for r in range(rounds):
        rr = cs.round(shard.z[r % 3], shard.ud[r % 3], hist[r % 2])
        assert rr == r
        got.append([hist[r % 2].clone(), cs.state(), None])
        if r >= depth - 1:                       # round r - depth + 1 is the oldest still in its slot: fetch it before the next round reuses the slot
            k = r - depth + 1
            cs.wait()
            got[k][2] = cs.plans(k)
    cs.wait()
    for k in range(max(0, rounds - depth + 1), rounds):
        got[k][2] = cs.plans(k)
    torch.cuda.synchronize()
    for r in range(rounds):
        h, xf, b = ref[r]
        gh, gx, gb = got[r]
        assert torch.equal(gh.view(torch.int32), h.view(torch.int32)), f"round {r}: history"
        assert torch.equal(gx.view(torch.int32), xf.view(torch.int32)), f"round {r}: final state"
        for key in ("sol", "xref"):
            assert torch.equal(gb[key].view(torch.int32), b[key].view(torch.int32)), f"round {r}: {key}"
        assert torch.equal(gb["status"], b["status"]) and torch.equal(gb["cost"].view(torch.int64), b["cost"].view(torch.int64)), f"round {r}: status / cost"
    assert not torch.equal(got[0][1], got[1][1])
    cs.close()


def c_round_refuses_too_few_queues(crx, o):
    """crx_swarm_create says no (CRX_ERR_INVALID + an explanation) when depth + 1 streams exceed the hardware queues, unless told to accept it."""
    import torch
    from cpprobotics_amd import Course, cswarm
    from cpprobotics_amd._lib import CrxError
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    dc = Course.from_numpy(course, device=dev)
    x0 = torch.zeros((256, 4), device=dev); P0 = torch.eye(4, device=dev).reshape(1, 16).repeat(256, 1).contiguous()
    q = cswarm.hw_queues()
    try:
        cswarm.CSwarm(x0, P0, dc, Q, R, 10, depth=q)
        raise AssertionError("crx_swarm_create accepted depth + 1 > hardware queues")
    except CrxError as e:
        assert "GPU_MAX_HW_QUEUES" in str(e)
    cs = cswarm.CSwarm(x0, P0, dc, Q, R, 10, depth=q, allow_shared_queues=True)
    cs.close()


def comm_one_rank(crx, o):
    """crx_comm_* / crx_allgather_dev on a one-rank communicator (the GPU box has one GPU): RCCL is found, the communicator comes up, the
    gather returns the send buffer — the trajectory concat of a one-process swarm."""
    import torch
    from cpprobotics_amd import cswarm
    uid = cswarm.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = cswarm.Comm(uid, 0, 1)
    assert (c.rank, c.world) == (0, 1)
    for shape, dt in (((100, 8192, 4), torch.float32), ((131072, 4), torch.float32), ((3,), torch.int32), ((7, 5), torch.float64)):
        t = (torch.rand(shape, device="cuda") * 100).to(dt)
        out = c.allgather(t)
        torch.cuda.synchronize()
        assert out.shape == (1,) + tuple(shape) and torch.equal(out[0], t)
    c.close()


def main(argv):
    warnings.filterwarnings("error", message=".*hardware queues.*")      # SwarmShard's warning must not fire here
    import oracle
    oracle.build()
    import cpprobotics_amd as crx
    crx.lib()
    check = argv[0]
    if check == "mixed_rounds":
        mixed_rounds(crx, oracle, int(argv[1]), rounds=int(argv[2]) if len(argv) > 2 else 5)
    elif check == "full_shard":
        full_shard(crx, oracle, int(argv[1]), int(argv[2]))
    elif check == "one_ekf_launch":
        one_ekf_launch(crx, oracle)
    elif check == "shared_planner_streams":
        shared_planner_streams(crx, oracle)
    elif check == "c_round":
        c_round(crx, oracle, int(argv[1]), int(argv[2]))
    elif check == "c_round_refuses_too_few_queues":
        c_round_refuses_too_few_queues(crx, oracle)
    elif check == "comm_one_rank":
        comm_one_rank(crx, oracle)
    else:
        raise SystemExit(f"unknown check {check}")
    import torch
    print(f"swarm check ok: {check} {' '.join(argv[1:])} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} "
          f"initialized={torch.cuda.is_initialized()}")


if __name__ == "__main__":
    main(sys.argv[1:])
