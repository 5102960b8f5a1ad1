#!/usr/bin/env python
"""How far the persistent MPC closed loop (HIP solver) drifts from the oracle's loop (CPU twin of the solver): floored relative
error of the trajectories per tick count, T = 6 and T = 21, incl. the reference's own start driven to the goal."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpprobotics_amd as crx, oracle
from common import floored_rel_err, mpc_course_f32, tracking_agents
course, goal = mpc_course_f32()
dc = crx.Course.from_numpy(course)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for T, n, max_ticks in ((6, 130, 40), (21, 70, 40), (6, 64, 700), (21, 32, 700)):
    st = tracking_agents(n, tuple(c[:150] for c in course), 9, spread=0.5)
    st[:, 3] = np.random.default_rng(10).uniform(0.5, 4.0, n).astype(np.float32)
    st[0] = (course[0][0], course[1][0], course[2][0], course[4][0])
    tind0 = oracle.calc_nearest_index(st, course)[0].astype(np.int32)
    so, tio, histo, tindo = oracle.mpc_closed_loop(st, course, goal, T=T, max_ticks=max_ticks, target_ind=tind0, want_hist=True)
    sd, td = t(st), t(tind0)
    ticks, hist = crx.mpc_simulation(sd, dc, goal, T, max_ticks, target_ind=td, want_hist=True)
    ticks, hist = ticks.cpu().numpy(), hist.cpu().numpy()
    errs = [floored_rel_err(hist[: tio[a], a], histo[: tio[a], a], 1.0) if tio[a] else 0.0 for a in range(n)]
    print(json.dumps({"T": T, "agents": n, "max_ticks": max_ticks, "ticks_equal": bool(np.array_equal(ticks, tio)), "reached_goal": int((tio < max_ticks).sum()),
                      "ticks_agent0": int(tio[0]), "max_err": float(max(errs)), "err_agent0": float(errs[0]),
                      "agents_bit_identical": int(sum(np.array_equal(hist[: tio[a], a], histo[: tio[a], a]) for a in range(n)))}), flush=True)
