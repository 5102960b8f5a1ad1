"""GPU parity of BASELINE.json configs[4]: the mixed EKF + MPC swarm round (cpprobotics_amd/swarm.py: SwarmShard / MixedSwarmRound) on
the engine's kernels, with the planners of several rounds in flight next to the EKF launches of the following rounds.

Reference shape: one pass of /root/reference/src/extended_kalman_filter.cpp:171-188 for every vehicle, one pass of
/root/reference/src/model_predictive_control.cpp:371-385 for every eighth.  The EKF part must reproduce the oracle's bits (whole
history, final state, final covariance); every plan must match the solver's CPU twin like tests/test_mpc_gpu.py demands (status bits,
sweep counts, 1e-6 floored on the solution, 1e-9 on the cost); consecutive rounds get DIFFERENT measurements, so a planner launch that
read another round's estimates (a race on the slot ring) would fail the comparison.

Every check runs in a process of its own (tests/swarm_checks.py) that exports GPU_MAX_HW_QUEUES=16 before HIP initialises — the
configuration bench.py measures (depth 7 on 16 hardware queues since round 6; 6 before) — and asserts that the shard really got `depth + 2` queues; in
round 5 these tests ran inside the pytest process on the runtime's default 4 queues and never at depth 6."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HW_QUEUES = 16          # bench.py: max(16, swarm_depth + 10)


def _run(*args, timeout=900, queues=HW_QUEUES):
    env = dict(os.environ)
    env["GPU_MAX_HW_QUEUES"] = str(queues)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "swarm_checks.py")] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "swarm check ok" in r.stdout and f"GPU_MAX_HW_QUEUES={queues}" in r.stdout
    assert "hardware queues" not in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("depth", [1, 2, 4, 6, 7])
def test_mixed_swarm_rounds_match_the_oracle(depth):
    """8,192 vehicles x 100 EKF steps, 1,024 planners, consecutive rounds on three different measurement sets with `depth` planner
    launches in flight (depth + 3 rounds at least: every slot is reused): every round's history / final state / covariance bit for
    bit, every plan against the twin."""
    _run("mixed_rounds", depth, max(5, depth + 3))


@pytest.mark.parametrize("depth,rounds", [(4, 3), (6, 9), (7, 10)])
def test_full_shard_on_a_strided_sample(depth, rounds):
    """One GPU's shard of the 1,048,576-agent swarm, 131,072 vehicles and 16,384 planners; (6, 9) is the configuration of bench.py's
    `swarm_configs4` line up to round 5 (depth 6 on 16 hardware queues, slots reused), (7, 10) the one since round 6."""
    _run("full_shard", depth, rounds)


def test_round_is_one_ekf_launch_without_a_gather():
    _run("one_ekf_launch")


def test_round_objects_of_a_process_share_their_planner_streams():
    _run("shared_planner_streams")


@pytest.mark.parametrize("depth,rounds", [(1, 3), (6, 9), (7, 10)])
def test_c_round_equals_the_python_round(depth, rounds):
    """crx_swarm_round_dev — the whole round issued by one C call (include/crx.h) — gives the bytes of the Python
    round (which the tests above hold to the oracle): history, final state, every plan buffer, slots reused."""
    _run("c_round", depth, rounds)


def test_c_round_refuses_more_streams_than_hardware_queues():
    _run("c_round_refuses_too_few_queues", queues=4)


def test_c_allgather_on_a_one_rank_communicator():
    """crx_comm_unique_id / crx_comm_init_rank / crx_allgather_dev / crx_comm_destroy: RCCL behind the C ABI."""
    _run("comm_one_rank")
