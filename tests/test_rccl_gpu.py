"""RCCL on a real device, every round: bench.py's N > 1 code path with a one-rank process group (`--force-dist`).  The multi-GPU run
itself is the driver's (8 GPUs are not ours to launch); this puts process-group init on the nccl backend (= RCCL), the asynchronous
all-gathers on RCCL's stream, the ring / lap waits of cpprobotics_amd/swarm.py (RingGather, ChunkedTrajectoryGather) and the configs[4]
rounds with their gathers through the GPU box, so that the first real 8-GPU run is not also the first RCCL run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = "29541"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--settle", "4",
           "--vehicles", "16384", "--T", "200", "--no-extras", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("gather", ["final", "traj"])
def test_bench_force_dist_runs_on_rccl(gather):
    res = _bench("--gather", gather)
    mg = res["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["ranks"] == 1 and mg["rccl_version"]
    assert res["n_gpus"] == 1 and res["value"] > 0
    # the collectives delivered this rank's own bytes (bench.py asserts it too; here it is part of the record)
    assert mg["gathered_equals_local"] and all(mg["gathered_equals_local"].values())
    if gather == "final":
        assert mg["gather_traj_chunked"]["updates_per_s"] > 0 and mg["gathered_equals_local"].keys() >= {"final_estimates", "trajectory_chunks"}
    # BASELINE configs[4] with its gathers, on the same process group
    sw = mg["swarm_configs4"]
    for kind in ("gather_final", "gather_traj"):
        assert "error" not in sw[kind], sw[kind]
        assert sw[kind]["round_ms"] > 0 and sw[kind]["mpc_sweeps"]["converged_frac"] > 0.99
