"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference cannot run in this image (no Eigen / OpenCV / CppAD / IPOPT)
and holds no test vectors of its own, so these fixtures are produced by the CPU oracle — the
Eigen-order restatement validated in tests/test_oracle_*.py — with its deterministic trig mode
(trig=1: glibc's FMA-variant sinf/cosf restated; identical to libm on FMA hosts).  They pin the
oracle against silent drift and give the GPU tests a fixed target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from common import (ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, mpc_course_f32, mpc_problem,  # noqa: E402
                    tracking_agents)

Q, R = ekf_QR()
n, T = 32, 250
u, x0, P0 = ekf_agents(n, 42)
u[0] = (1.0, 0.1); x0[0] = 0      # vehicle 0 = the reference's own scenario
w = ekf_noise(T, n, 43)
z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u, x0, x0, w, trig=1)
x, P, xh, ph = oracle.ekf_run(x0, P0, z, ud, Q, R, trig=1, want_phist=True)
np.savez_compressed(os.path.join(HERE, "ekf_golden.npz"), u_true=u, x0=x0, P0=P0, w=w, z=z, ud=ud, Q=Q, R=R,
                    x_hist=xh, P_final=ph[-1])

v = lqr_speeds(256, 44)
v[:4] = (0.0, 0.05, 2.78, -1.0)
out = dict(v=v)
for dim in (5, 4):
    A, B, Qm, Rm = oracle.lqr_build(v, dim)
    X, K, it = oracle.dare(A, B, Qm, Rm)
    out[f"X{dim}"], out[f"K{dim}"], out[f"it{dim}"] = X, K, it
np.savez_compressed(os.path.join(HERE, "lqr_golden.npz"), **out)

Tm = 21
mx0, mxref = mpc_problem(64, Tm, 45)
sol, st, cost = oracle.mpc_solve(mx0, mxref, Tm)
np.savez_compressed(os.path.join(HERE, "mpc_golden.npz"), x0=mx0, xref=mxref, T=Tm, sol=sol, status=st, cost=cost)

# course tracking (oracle/track_ref.cpp): one control evaluation, update, whole closed loops, MPC front-end
oracle.oracle_lib.lib().oracle_track_set_trig_mode(1)
import oracle.oracle_lib as _ol  # noqa: E402
_ol.trig_mode = lambda: 1            # deterministic trig flavour for the fixtures (see the module docstring)
course, goal = lqr_course()
st = tracking_agents(48, tuple(c[:200] for c in course), 46, spread=0.4)
st[0] = 0.0
out = dict(course=np.stack(course), goal=np.float32(goal), state=st)
rng = np.random.default_rng(47)
pe, pth = rng.normal(0, 0.3, 48).astype(np.float32), rng.normal(0, 0.2, 48).astype(np.float32)
out["pe"], out["pth"] = pe, pth
for dim in (5, 4):
    ctl, ind, pe1, pth1 = oracle.lqr_steering_control(st, course, pe, pth, dim=dim)
    out[f"ctl{dim}"], out[f"ind{dim}"], out[f"pe{dim}"], out[f"pth{dim}"] = ctl, ind, pe1, pth1
    s1, ticks, _, _, _, _ = oracle.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=600)
    out[f"loop_state{dim}"], out[f"loop_ticks{dim}"] = s1, ticks
a, d = rng.uniform(-1.5, 1.5, 48).astype(np.float32), rng.uniform(-1.0, 1.0, 48).astype(np.float32)
out["a"], out["delta"] = a, d
out["update_lqr"] = oracle.update(st, a, d)
out["update_mpc"] = oracle.update(st, a, d, dt=0.2, wheelbase=2.5, clamp_speed=True)
mcourse, mgoal = mpc_course_f32()
mst = tracking_agents(48, mcourse, 48, spread=1.0)
tind0 = rng.integers(0, len(mcourse[0]), 48).astype(np.int32)
xr, tind = oracle.calc_ref_trajectory(mst, mcourse, tind0, 21)
out.update(mcourse=np.stack(mcourse), mstate=mst, tind0=tind0, xref21=xr, tind=tind)
np.savez_compressed(os.path.join(HERE, "track_golden.npz"), **out)
print("wrote", os.listdir(HERE))
