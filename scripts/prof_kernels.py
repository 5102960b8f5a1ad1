#!/usr/bin/env python
"""The launches rocprofv3 looks at besides the fused EKF launch of bench.py (scripts/gpu_prof.sh: --kernel-trace --stats, or --pmc):
  * structured Riccati solve of BASELINE configs[2] in BOTH register layouts (one agent per lane / a DPP quad per agent), 5x5 and 4x4;
  * the solve_DARE(A, B, Q, R) signature (crx_dare_batch_dev) on the same agents' matrices — recognised and served by the structured
    kernels — and the DENSE kernel forced on them, and on general dense matrices;
  * the MPC horizon solve of configs[3] (the four-lane A/B variant too when libcrx_x.so is there);
  * the persistent LQR closed loop in both layouts.
`--host-calls`: instead, a few host-pointer calls (EKF run through the pipeline, one-vehicle zero-copy calls) for a marker trace.
Prints the iteration statistics the counter post-processing needs (scripts/summarize_prof.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import closed_loop_prediction_lanes, dare_dense, dlqr_from_v_lanes, mpc_solve_lanes  # noqa: E402
from common import ekf_QR, ekf_agents, lqr_course, lqr_speeds, mpc_problem, tracking_agents  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(args[0]) if args else 3

if "--host-calls" in sys.argv:
    Q, R = ekf_QR()
    n, T = 65536, 64
    u, x0, P0 = ekf_agents(n, 1)
    z = np.random.default_rng(0).standard_normal((T, n, 2)).astype(np.float32)
    ud = np.tile(u[None], (T, 1, 1)).astype(np.float32)
    for _ in range(2):
        crx.host.ekf_run(x0.copy(), P0.copy(), z, ud, Q, R, want_hist=True)          # the three-stream pipeline
    for _ in range(5):
        crx.host.ekf_run(x0[:1].copy(), P0[:1].copy(), z[:1, :1].copy(), ud[:1, :1].copy(), Q, R)   # ekf_estimation, one vehicle
    crx.host.dare_from_v(lqr_speeds(16384, 3), 5)
    print(json.dumps({"host_calls": "ekf_run 65536 x 64 (x2), ekf_estimation n = 1 (x5), dare_from_v 16384"}))
    sys.exit(0)

vh = lqr_speeds(16384, 3)
v = torch.from_numpy(vh).cuda()
for lanes in (1, 4):
    for dim in (5, 4):
        for _ in range(reps):
            K, X, it = dlqr_from_v_lanes(v, dim, lanes)
it5 = dlqr_from_v_lanes(v, 5, 4)[2].cpu().numpy().astype(np.int64)
# the dense signature on the reference's matrices, and the dense kernel itself
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
A, B, Qm, Rm = (torch.from_numpy(a).cuda() for a in bench.lqr_pattern_mats(vh))
rng = np.random.default_rng(5)
Ag = torch.from_numpy((np.eye(5)[None] * 0.9 + 0.15 * rng.standard_normal((16384, 5, 5))).astype(np.float32).reshape(16384, 25)).cuda()
Bg = torch.from_numpy(rng.standard_normal((16384, 10)).astype(np.float32)).cuda()
for _ in range(reps):
    crx.solve_DARE(A, B, Qm, Rm)
for _ in range(max(1, reps // 3)):
    Xd, Kd, itd = dare_dense(A, B, Qm, Rm)                      # 16,384 agents: the quad layout (dare_dense_quad_kernel)
    dare_dense(A, B, Qm, Rm, lanes_per_agent=1)                # and the one-lane layout beside it
    Xg, itg = crx.solve_DARE(Ag, Bg, Qm, Rm, eps=1e-3, maxiter=60)
itd, itg = itd.cpu().numpy().astype(np.int64), itg.cpu().numpy().astype(np.int64)
x0, xref = mpc_problem(8192, 21, 4)
x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
layouts = [1]
try:
    from cpprobotics_amd.experimental import ablib
    ablib()
    layouts.append(4)
except Exception:
    pass
for lanes in layouts:
    for _ in range(reps):
        sol, st, cost = mpc_solve_lanes(x0, xref, 21, lanes)
for _ in range(reps):                                   # the four-variant portfolio on the same batch (round 4)
    solp, stp, costp = crx.mpc_solve(x0, xref, 21, return_status=True, portfolio=True)
course, goal = lqr_course()
dc = crx.Course.from_numpy(course)
stl = torch.from_numpy(tracking_agents(16384, tuple(c[:200] for c in course), 5, spread=0.4)).cuda()
for lanes in (1, 4):
    for dim in (5, 4):
        for _ in range(max(1, reps // 3)):
            closed_loop_prediction_lanes(stl.clone(), dc, goal, lanes, dim=dim, max_ticks=400)
# the single-step EKF update in its HBM-bound regime: 4 M vehicles (ekf_step_kernel<true>, covariance rows past the caches) and 1 M
# (ekf_step_kernel<false>), back-to-back launches; the FETCH_SIZE / WRITE_SIZE passes of gpu_prof.sh give their HBM traffic per launch
Qe, Re = ekf_QR()
for nstep in (1 << 22, 1 << 20):
    xs_ = torch.zeros((nstep, 4), dtype=torch.float32, device="cuda")
    Ps_ = torch.eye(4, dtype=torch.float32, device="cuda").reshape(1, 16).repeat(nstep, 1).contiguous()
    zs_ = torch.randn((nstep, 2), dtype=torch.float32, device="cuda")
    us_ = torch.tensor([1.0, 0.1], dtype=torch.float32, device="cuda").repeat(nstep, 1).contiguous()
    for _ in range(reps + 2):
        crx.ekf_estimation(xs_, Ps_, zs_, us_, Qe, Re)
    del xs_, Ps_, zs_, us_
torch.cuda.synchronize()
mit = (st.cpu().numpy() >> 8).astype(np.int64)
pit = (stp.cpu().numpy() >> 8).astype(np.int64)
wsum = lambda a, k: int(a.reshape(-1, k).max(axis=1).sum())
print(json.dumps({"dare5": {"agents": 16384, "iters_sum": int(it5.sum()), "wave_max_iters_sum": wsum(it5, 64)},
                  "dare5_quad": {"agents": 16384, "iters_sum": int(it5.sum()), "wave_max_iters_sum": wsum(it5, 16)},
                  "dare5_dense_reference_matrices": {"agents": 16384, "iters_sum": int(itd.sum()), "wave_max_iters_sum": wsum(itd, 16)},
                  "dare5_dense_reference_matrices_one_lane": {"agents": 16384, "iters_sum": int(itd.sum()), "wave_max_iters_sum": wsum(itd, 64)},
                  "dare5_dense_general_matrices": {"agents": 16384, "iters_sum": int(itg.sum()), "wave_max_iters_sum": wsum(itg, 16)},
                  "mpc_T21": {"agents": 8192, "iters_sum": int(mit.sum()), "iters_max": int(mit.max()), "wave_max_iters_sum": wsum(mit, 64)},
                  "mpc_T21_quad": {"agents": 8192, "iters_sum": int(mit.sum()), "iters_max": int(mit.max()), "wave_max_iters_sum": wsum(mit, 16)},
                  "mpc_T21_portfolio": {"agents": 8192, "iters_sum": int(pit.sum()), "iters_max": int(pit.max()), "wave_max_iters_sum": wsum(pit, 16)}}))
