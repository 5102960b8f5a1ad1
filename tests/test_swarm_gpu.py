"""GPU parity of BASELINE.json configs[4]: the mixed EKF + MPC swarm round (cpprobotics_amd/swarm.py: SwarmShard / MixedSwarmRound) on
the engine's kernels, with the planners of several rounds in flight next to the EKF launches of the following rounds.

Reference shape: one pass of /root/reference/src/extended_kalman_filter.cpp:171-188 for every vehicle, one pass of
/root/reference/src/model_predictive_control.cpp:371-385 for every eighth.  The EKF part must reproduce the oracle's bits (whole
history, final state, final covariance); every plan must match the solver's CPU twin like tests/test_mpc_gpu.py demands (status bits,
sweep counts, 1e-6 floored on the solution, 1e-9 on the cost); consecutive rounds get DIFFERENT measurements, so a planner launch that
read another round's estimates (a race on the slot ring) would fail the comparison."""
import numpy as np
import pytest

from common import ekf_QR, floored_rel_err, mpc_course_f32, mpc_solve_threads

pytestmark = pytest.mark.gpu
TM = 21


def _oracle_round(o, shard, s, agents):
    """The oracle's round on the given (sorted) agent subset with input set s: -> x_hist [T, m, 4], x, P."""
    Q, R = ekf_QR()
    idx = np.asarray(agents)
    z, ud = shard.z[s][:, idx].cpu().numpy(), shard.ud[s][:, idx].cpu().numpy()
    x0, P0 = shard.x0[idx].cpu().numpy(), shard.P0[idx].cpu().numpy()
    x, P, xh, _ = o.ekf_run(x0, P0, np.ascontiguousarray(z), np.ascontiguousarray(ud), Q, R)
    return xh, x, P


def _oracle_plans(o, course, est, v_cmd):
    e = est.copy()
    e[:, 3] = np.float32(v_cmd)
    tind = o.calc_nearest_index(e, course)[0].astype(np.int32)
    xref, _ = o.calc_ref_trajectory(e, course, tind, TM)
    so, sto, co = mpc_solve_threads(o, e, xref, TM)
    return e, xref, so, sto, co


def _check_plans(b, rows, e, xref, so, sto, co):
    sd, std, cd = b["sol"].cpu().numpy()[rows], b["status"].cpu().numpy()[rows], b["cost"].cpu().numpy()[rows]
    assert np.array_equal(b["xref"].cpu().numpy()[rows], xref), "calc_ref_trajectory differs from the oracle"
    assert np.array_equal(std & 3, sto & 3), f"status differs for planners {np.flatnonzero((std & 3) != (sto & 3))[:8]}"
    assert (np.abs((std >> 8) - (sto >> 8)) <= 1).all()
    conv = (sto & 1) == 1
    assert conv.mean() >= 0.95
    assert floored_rel_err(sd[conv], so[conv], 1.0) <= 1e-6
    crel = np.abs(cd - co) / np.maximum(np.abs(co), 1.0)
    assert crel[conv].max(initial=0.0) <= 1e-9 and crel.max(initial=0.0) <= 1e-6


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_mixed_swarm_rounds_match_the_oracle(crx, oracle_mod, depth):
    """8,192 vehicles x 100 EKF steps, 1,024 planners, `rounds` consecutive rounds on three different measurement sets with `depth`
    planner launches in flight: every round's history / final state / covariance bit for bit, every plan against the twin."""
    import torch
    from cpprobotics_amd import swarm
    o = oracle_mod
    n, T, rounds = 8192, 100, 5
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    shard = swarm.SwarmShard(n, T, course, Q, R, dev, depth=depth, input_sets=3, seed=7)
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
        xh, x, P = _oracle_round(o, shard, r % 3, all_agents)
        assert np.array_equal(hists[r].cpu().numpy(), xh), f"round {r}: EKF history differs from the oracle"
        assert np.array_equal(finals[r][0].cpu().numpy(), x) and np.array_equal(finals[r][1].cpu().numpy(), P)
        e, xref, so, sto, co = _oracle_plans(o, course, x[::8], shard.v_cmd)
        _check_plans(plans[r], slice(None), e, xref, so, sto, co)
    # the three measurement sets really differ (otherwise the race check above would be vacuous)
    assert not np.array_equal(finals[0][0].cpu().numpy(), finals[1][0].cpu().numpy())


def test_full_shard_on_a_strided_sample(crx, oracle_mod):
    """One GPU's shard of the 1,048,576-agent swarm — 131,072 vehicles, 16,384 planners (the lane-refilling kernel is not selected
    below 65,536 planners; both kernels are bit-identical per agent anyway, tests/test_mpc_gpu.py) — three rounds at depth 4:
    every 64th vehicle's history and final state bit for bit, every 16th planner against the twin."""
    import torch
    from cpprobotics_amd import swarm
    o = oracle_mod
    n, T = 131072, 100
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    dev = torch.device("cuda", 0)
    shard = swarm.SwarmShard(n, T, course, Q, R, dev, depth=4, input_sets=2, seed=99)
    for r in range(3):
        shard.run()
    hist = shard.rnd.trajectory_time_major()
    xf = shard.x.clone()
    shard.wait()
    torch.cuda.synchronize()
    agents = np.arange(0, n, 64)
    xh, x, P = _oracle_round(o, shard, 2 % 2, agents)
    assert np.array_equal(hist[:, agents].cpu().numpy(), xh)
    assert np.array_equal(xf[agents].cpu().numpy(), x)
    # planner j plans for vehicle 8 j: every 16th planner = every 128th vehicle = every second sampled agent
    e, xref, so, sto, co = _oracle_plans(o, course, x[::2], shard.v_cmd)
    rows = np.arange(0, shard.n_plan, 16)
    _check_plans(shard.rnd.plans_of(2), rows, e, xref, so, sto, co)
    st = shard.rnd.plans_of(2)["status"].cpu().numpy()
    assert ((st & 1) == 1).mean() > 0.99 and not np.any(st & 2)


def test_round_is_one_ekf_launch_without_a_gather(crx):
    """A single process has nothing to overlap the chunks with: the round must issue ONE fused EKF launch (round 4 issued four)."""
    import torch
    from cpprobotics_amd import swarm
    Q, R = ekf_QR()
    course, goal = mpc_course_f32()
    shard = swarm.SwarmShard(2048, 40, course, Q, R, torch.device("cuda", 0), depth=2, chunks=4, record_ekf_events=True)
    shard.run(); shard.wait()
    torch.cuda.synchronize()
    assert shard.rnd.chunks == 1 and len(shard.ekf_events) == 1


def test_round_objects_of_a_process_share_their_planner_streams(crx):
    """Every hardware queue the MPC solve has run on keeps a scratch reservation, and 16 of them abort the process
    (profiles/r05/scratch_queues_probe.jsonl): round objects built one after another must not each bring fresh streams."""
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
