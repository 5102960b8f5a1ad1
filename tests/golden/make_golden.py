"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference holds no test vectors of its own and its executables cannot be built in this image (no Eigen / OpenCV /
CppAD / IPOPT).  Its hot-path FUNCTIONS can: oracle/ref_build.sh cuts them out of /root/reference and
compiles them unmodified against the Eigen stand-in (or the host's Eigen) into oracle/_ref/libref.so.  Every EKF / DARE /
tracking array stored here is the output of that library — the reference's own lines — and the generator asserts that the
CPU oracle reproduces it bit for bit (host libm = glibc's FMA flavour, which is also what the oracle's deterministic trig
mode 1 and the HIP kernels restate).  The MPC solution is the oracle solver's (the reference's IPOPT result is not
reproducible); `ref_eigen` records which Eigen libref.so was built against (0 = stand-in, 1 = host Eigen).
Needs /root/reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from oracle import ref_lib as ref  # noqa: E402
from common import (ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, mpc_course_f32, mpc_problem,  # noqa: E402
                    tracking_agents)

assert ref.build(), "needs /root/reference (oracle/ref_build.sh)"
assert oracle.libm_is_fma_flavour(), "the fixtures are defined on glibc's FMA-flavour libm"
KIND = np.int32(ref.lib().ref_eigen_kind())


def same(a, b):
    assert np.array_equal(a, b), "oracle and reference lines disagree"
    return b


Q, R = ekf_QR()
n, T = 32, 250
u, x0, P0 = ekf_agents(n, 42)
u[0] = (1.0, 0.1); x0[0] = 0      # vehicle 0 = the reference's own scenario
w = ekf_noise(T, n, 43)
z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u, x0, x0, w, trig=1)
x, P, xh, ph = oracle.ekf_run(x0, P0, z, ud, Q, R, trig=1, want_phist=True)
xr, Pr, xhr, phr = ref.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
m = ref.ekf_main(w[:, 0, :].astype(np.float64))               # vehicle 0 through the reference's own main loop
same(z[:, 0], m["hz"]); same(ud[:, 0], m["hud"]); same(xhr[:, 0], m["hxEst"])
np.savez_compressed(os.path.join(HERE, "ekf_golden.npz"), u_true=u, x0=x0, P0=P0, w=w, z=z, ud=ud, Q=Q, R=R,
                    x_hist=same(xh, xhr), P_final=same(ph[-1], phr[-1]), ref_eigen=KIND)

v = lqr_speeds(256, 44)
v[:4] = (0.0, 0.05, 2.78, -1.0)
out = dict(v=v)
for dim in (5, 4):
    A, B, Qm, Rm = oracle.lqr_build(v, dim)
    X, K, it = oracle.dare(A, B, Qm, Rm)
    Xr, Kr = ref.dare(A, B, Qm, Rm)
    out[f"X{dim}"], out[f"K{dim}"], out[f"it{dim}"] = same(X, Xr), same(K, Kr), it   # iteration counts: the oracle's (the reference returns none)
np.savez_compressed(os.path.join(HERE, "lqr_golden.npz"), ref_eigen=KIND, **out)

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
    ctlr, indr, pe1r, pth1r = ref.lqr_steering_control(st, course, pe, pth, dim=dim)
    if dim == 4:
        same(ind, indr)
    out[f"ctl{dim}"], out[f"ind{dim}"], out[f"pe{dim}"], out[f"pth{dim}"] = same(ctl, ctlr), ind, same(pe1, pe1r), same(pth1, pth1r)
    s1, ticks, _, _, _, _ = oracle.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=600)
    s1r, ticksr, _ = ref.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=600)
    out[f"loop_state{dim}"], out[f"loop_ticks{dim}"] = same(s1, s1r), same(ticks, ticksr)
a, d = rng.uniform(-1.5, 1.5, 48).astype(np.float32), rng.uniform(-1.0, 1.0, 48).astype(np.float32)
out["a"], out["delta"] = a, d
out["update_lqr"] = same(oracle.update(st, a, d), ref.lqr_update(st, a, d))
out["update_mpc"] = same(oracle.update(st, a, d, dt=0.2, wheelbase=2.5, clamp_speed=True), ref.mpc_update(st, a, d))
mcourse, mgoal = mpc_course_f32()
mst = tracking_agents(48, mcourse, 48, spread=1.0)
tind0 = rng.integers(0, len(mcourse[0]) - 10, 48).astype(np.int32)   # the reference reads cx[pind .. pind+9] unchecked (:110)
xr, tind = oracle.calc_ref_trajectory(mst, mcourse, tind0, 21)
xrr, tindr = ref.calc_ref_trajectory(mst, mcourse, tind0, 21)
same(xr, xrr); same(tind, tindr)
out.update(ref_eigen=KIND, mcourse=np.stack(mcourse), mstate=mst, tind0=tind0, xref21=xr, tind=tind)
np.savez_compressed(os.path.join(HERE, "track_golden.npz"), **out)
print("wrote", os.listdir(HERE))
