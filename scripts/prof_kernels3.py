#!/usr/bin/env python
"""The DARE and MPC launches of BASELINE configs[2] / configs[3] in BOTH register layouts (one agent per lane / a DPP quad per
agent), a few times each, for rocprofv3 (--kernel-trace --stats or --pmc).  Prints the iteration statistics the counter
post-processing needs (scripts/gpu_prof3.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import closed_loop_prediction_lanes, dlqr_from_v_lanes, mpc_solve_lanes  # noqa: E402
from common import lqr_course, lqr_speeds, mpc_problem, tracking_agents  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
v = torch.from_numpy(lqr_speeds(16384, 3)).cuda()
for lanes in (1, 4):
    for dim in (5, 4):
        for _ in range(reps):
            K, X, it = dlqr_from_v_lanes(v, dim, lanes)
it5 = dlqr_from_v_lanes(v, 5, 4)[2].cpu().numpy().astype(np.int64)
x0, xref = mpc_problem(8192, 21, 4)
x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
for lanes in (1, 4):
    for _ in range(reps):
        sol, st, cost = mpc_solve_lanes(x0, xref, 21, lanes)
# the persistent LQR closed loop (side_bench's workload) in both layouts
course, goal = lqr_course()
dc = crx.Course.from_numpy(course)
stl = torch.from_numpy(tracking_agents(16384, tuple(c[:200] for c in course), 5, spread=0.4)).cuda()
for lanes in (1, 4):
    for dim in (5, 4):
        for _ in range(max(1, reps // 3)):
            closed_loop_prediction_lanes(stl.clone(), dc, goal, lanes, dim=dim, max_ticks=400)
torch.cuda.synchronize()
mit = (st.cpu().numpy() >> 8).astype(np.int64)
wsum = lambda a, k: int(a.reshape(-1, k).max(axis=1).sum())
print(json.dumps({"dare5": {"agents": 16384, "iters_sum": int(it5.sum()), "wave_max_iters_sum": wsum(it5, 64)},
                  "dare5_quad": {"agents": 16384, "iters_sum": int(it5.sum()), "wave_max_iters_sum": wsum(it5, 16)},
                  "mpc_T21": {"agents": 8192, "iters_sum": int(mit.sum()), "iters_max": int(mit.max()), "wave_max_iters_sum": wsum(mit, 64)},
                  "mpc_T21_quad": {"agents": 8192, "iters_sum": int(mit.sum()), "iters_max": int(mit.max()), "wave_max_iters_sum": wsum(mit, 16)}}))
