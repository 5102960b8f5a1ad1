"""Generates tests/golden/planner_golden.npz: the dynamic-window and Frenet planners' oracle outputs on fixed inputs.

Same role as make_golden.py (which see): the planners' functions are cut out of /root/reference and compiled unmodified
(oracle/ref_build.sh -> oracle/_ref/libref.so); the generator asserts that the CPU oracle reproduces that library bit for bit
(DWA: whole episodes; Frenet: spline table, polynomial coefficients, every candidate's cost and verdict, whole episodes)
and stores the arrays.  The Frenet oracle calls the host libm (pow, cos/sin in double, atan2f) — consumers on other hosts
compare it at the 1e-5 tolerance of its parity contract; DWA uses the deterministic trig mode (== the host libm here).
Run from the repo root:  python tests/golden/make_golden_planners.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
import oracle.oracle_lib as O  # noqa: E402
from oracle import ref_lib as ref  # noqa: E402

assert ref.build(), "needs /root/reference (oracle/ref_build.sh)"
assert oracle.libm_is_fma_flavour()


def same(a, b, what):
    assert np.array_equal(a, b), "oracle and reference lines disagree: " + what
    return b


O.trig_mode = lambda: 1
out = {}
# ---- dynamic window: agent 0 = the reference's start (src/dynamic_window_approach.cpp:167-181) --------------------------
rng = np.random.default_rng(51)
n = 12
st = np.stack([rng.uniform(-1, 9, n), rng.uniform(-1, 9, n), rng.uniform(-3.2, 3.2, n), rng.uniform(-0.5, 1.0, n),
               rng.uniform(-0.69, 0.69, n)], axis=1).astype(np.float32)
st[0] = (0.0, 0.0, 3.141592653 / 8.0, 0.0, 0.0)
u = st[:, 3:5].copy()
goal = np.stack([rng.uniform(8, 12, n), rng.uniform(8, 12, n)], axis=1).astype(np.float32)
goal[0] = (10.0, 10.0)
u1, ns, bi = oracle.dwa_control(st, u, goal)
s60, u60, t60, _ = oracle.dwa_run(st, u, goal, 60)
rs60, ru60, rt60, _ = ref.dwa_run(st, u, goal, O.DWA_OBSTACLES if hasattr(O, "DWA_OBSTACLES") else oracle.dwa_run.__defaults__[0], 60)
same(s60, rs60, "dwa state"); same(u60, ru60, "dwa u"); same(t60, rt60, "dwa ticks")
rs1, ru1, _, _ = ref.dwa_run(st, u, goal, oracle.dwa_run.__defaults__[0], 1)
same(u1, ru1, "dwa_control")
out.update(dwa_state=st, dwa_u=u, dwa_goal=goal, dwa_u1=u1, dwa_ns=ns, dwa_best=bi, dwa_state60=s60, dwa_u60=u60, dwa_ticks60=t60)
# ---- Frenet: agent 0 = the reference's start (src/frenet_optimal_trajectory.cpp:186-219) ---------------------------------
coef = same(oracle.frenet_spline_build(), ref.frenet_spline_build(O.FRENET_WX, O.FRENET_WY), "Spline2D table")
rx, ry = oracle.frenet_course_samples(coef)
n = 24
fs = np.stack([rng.uniform(0.0, 65.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(-0.8, 0.8, n),
               rng.uniform(-0.5, 0.5, n)], axis=1).astype(np.float32)
fs[0] = O.FRENET_STATE0
p = oracle.frenet_plan(fs, coef)
rp = ref.frenet_run(fs, O.FRENET_WX, O.FRENET_WY, [rx[-1], ry[-1]], O.FRENET_OBSTACLES, 1, want_paths=True, cap=p["path_cf"].shape[1])
same(p["n_paths"], rp["n_paths"], "candidate count"); same(p["path_ok"], rp["path_ok"], "check_paths verdicts")
assert np.array_equal(p["path_cf"], rp["path_cf"], equal_nan=True), "candidate costs"
same(p["status"] & 1, rp["status"] & 1, "no-survivor flag")
out.update(fr_wx=O.FRENET_WX, fr_wy=O.FRENET_WY, fr_ob=O.FRENET_OBSTACLES, fr_coef=coef, fr_goal=np.array([rx[-1], ry[-1]], np.float32),
           fr_nsamples=np.int32(len(rx)), fr_state=fs, fr_out=p["out"], fr_best=p["best"], fr_nvalid=p["n_valid"], fr_status=p["status"],
           fr_path_cf=p["path_cf"], fr_path_ok=p["path_ok"])
r = oracle.frenet_run(fs[:6], coef, out["fr_goal"], 120, want_hist=True)
rr = ref.frenet_run(fs[:6], O.FRENET_WX, O.FRENET_WY, out["fr_goal"], O.FRENET_OBSTACLES, 120)
same(r["ticks"], rr["ticks"], "episode length"); same(r["status"] & 1, rr["status"] & 1, "episode status")
for a in range(6):
    same(r["hist"][: r["ticks"][a], a], rr["hist"][: r["ticks"][a], a], "episode trajectory")
out["fr_run_state"], out["fr_run_ticks"], out["fr_run_status"] = r["state"], r["ticks"], r["status"]
out["fr_run_hist0"] = r["hist"][: r["ticks"][0], 0]
np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **out)
print("wrote planner_golden.npz", {k: v.shape for k, v in out.items()})
