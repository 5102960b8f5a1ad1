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


_SWARM_SCRIPT = r'''
import os, sys
import torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cpprobotics_amd import swarm
from common import ekf_QR, mpc_course_f32
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
Q, R = ekf_QR(); course, goal = mpc_course_f32()
for kind in ("final", "traj"):
    shard = swarm.SwarmShard(8192, 40, course, Q, R, dev, depth=3, gather=kind, input_sets=2, seed=5)
    snaps = []
    for r in range(5):                              # slots are reused from round 3 on: the asynchronous gathers' buffers too
        shard.run()
        snaps.append((shard.x.clone(), shard.rnd.local_hist.clone() if shard.rnd.local_hist is not None else shard.rnd.cg.local.clone()))
    shard.wait(); torch.cuda.synchronize()
    if kind == "final":
        assert torch.equal(shard.rnd.final, snaps[-1][0]), "gathered final estimates differ from the local state"
    else:
        assert torch.equal(shard.rnd.trajectory_time_major(), snaps[-1][1].reshape(40, 8192, 4)), "gathered trajectory differs from the local history"
    assert not torch.equal(snaps[-1][0], snaps[-2][0])        # consecutive rounds really had different inputs
dist.barrier(); dist.destroy_process_group()
print("swarm gathers ok", torch.cuda.nccl.version())
'''


def test_swarm_round_gathers_on_rccl():
    """MixedSwarmRound's two gather forms on a one-rank RCCL group: the asynchronous final-estimate gather (slot ring) and the chunked
    trajectory gather return, after five rounds with slot reuse, exactly the last round's local results."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", "29543"
    r = subprocess.run([sys.executable, "-c", _SWARM_SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "swarm gathers ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
