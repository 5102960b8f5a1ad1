"""Generates tests/golden/pf_golden.npz: outputs of the reference's own particle-filter lines (src/particle_filter.cpp:25-148,
compiled unmodified by oracle/ref_build.sh with the random draws injected — oracle/ref_shim/ref_pf.cpp) on fixed inputs, for hosts
that have neither /root/reference nor libref.so (tests/test_ref_golden.py::test_pf_fixture).  The generator asserts that the CPU
oracle reproduces the library first — bit for bit for everything but the three Eigen reductions pw.sum(), px*pw, pw'pw, see
oracle/pf_ref.cpp.  Host-libm defined (cosf / sinf / expf of glibc >= 2.28).  Run from the repo root:
  python tests/golden/make_golden_pf.py
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
O.trig_mode = lambda: 0
NP = ref.pf_np()
rng = np.random.default_rng(81)
RFID = np.array([[10.0, 0.0], [10.0, 10.0], [0.0, 15.0], [-5.0, 20.0]], np.float32)
rsim4 = (1.0, 0.0, 0.0, float(np.float32(O.PF_RSIM[1])))
rsim2 = (rsim4[0], rsim4[3])
u = np.array([1.0, 0.1], np.float32)
out = dict(rsim2=np.array(rsim2, np.float32), two_u=u)


def same(a, b, what):
    assert np.array_equal(a, b, equal_nan=True), "oracle and reference lines disagree: " + what
    return b


def case(spread):
    px = np.stack([rng.normal(3, spread, NP), rng.normal(4, spread, NP), rng.uniform(-3.2, 3.2, NP), rng.normal(1, 0.3, NP)], axis=1).astype(np.float32)
    nz = int(rng.integers(0, 5))
    z = np.zeros((4, 3), np.float32)
    z[:nz] = np.concatenate([(np.hypot(3 - RFID[:nz, :1], 4 - RFID[:nz, 1:]) + rng.normal(0, 0.2, (nz, 1))), RFID[:nz]], axis=1)
    return px, nz, z, rng.standard_normal((NP, 2)).astype(np.float32), rng.uniform(1.0, 2.0, NP).astype(np.float32)


# small functions
n = 2000
x = np.stack([rng.normal(0, 50, n), rng.normal(0, 50, n), rng.uniform(-200, 200, n), rng.normal(0, 5, n)], axis=1).astype(np.float32)
uu = np.stack([rng.normal(1, 2, n), rng.normal(0, 1, n)], axis=1).astype(np.float32)
out.update(mm_x=x, mm_u=uu, mm_out=same(oracle.motion_model(x, uu, trig=0), ref.pf_motion_model(x, uu), "motion_model"))
dz = np.concatenate([rng.normal(0, 0.3, n), rng.normal(0, 30, n)]).astype(np.float32)
sg = np.concatenate([np.full(n, np.sqrt(np.float32(0.01)), np.float32), rng.uniform(0.01, 3.0, n).astype(np.float32)])
out.update(gl_x=dz, gl_sigma=sg, gl_out=same(oracle.pf_gauss_likelihood(dz, sg), ref.pf_gauss_likelihood(dz, sg), "gauss_likelihood"))

two = {k: [] for k in ("px", "pw", "z", "nz", "nrm", "uni", "px_out", "pw_out", "xEst", "PEst")}
for k in range(24):
    px, nz, z, nrm, uni = case(1.0)
    live = rng.choice(NP, 2 if k % 4 else 1, replace=False)
    pw = np.zeros(NP, np.float32); pw[live] = rng.uniform(0.1, 1.0, len(live)).astype(np.float32)
    px[live, :2] = (np.array([3.0, 4.0]) + rng.normal(0, 0.05, (len(live), 2))).astype(np.float32)
    nrm[live] *= 0.05
    pr, wr, xr, Pr = ref.pf_localization(px, pw, z[:nz], u, nrm, rsim=rsim4)
    p2, w2 = ref.pf_resampling(pr, wr, uni)
    obs = z[None]
    po, wo, xo, Po, _, _ = oracle.pf_step_parts(px[None], pw[None], obs, np.array([nz], np.int32), u[None], nrm[None], uni[None], 3, rsim=rsim2)
    same(po[0], p2, "two-live px"); same(wo[0], w2, "two-live pw"); same(xo[0], xr, "two-live xEst"); same(Po[0], Pr, "two-live PEst")
    for key, v in zip(two, (px, pw, z, nz, nrm, uni, p2, w2, xr, Pr)):
        two[key].append(v)
out.update({"two_" + k: np.array(v) for k, v in two.items()})

gen = {k: [] for k in ("px", "pw", "z", "nz", "nrm", "uni", "px_out", "pw_out", "xEst", "PEst")}
res = {k: [] for k in ("pw", "px_out", "pw_out")}
for k in range(24):
    px, nz, z, nrm, uni = case(rng.uniform(0.2, 3.0))
    pw = rng.uniform(0.2, 1.0, NP).astype(np.float32); pw = (pw / pw.sum()).astype(np.float32)
    pr, wr, xr, Pr = ref.pf_localization(px, pw, z[:nz], u, nrm, rsim=rsim4)
    po, *_ = oracle.pf_step_parts(px[None], pw[None], z[None], np.array([nz], np.int32), u[None], nrm[None], uni[None], 1, rsim=rsim2)
    same(po[0], pr, "general px")
    for key, v in zip(gen, (px, pw, z, nz, nrm, uni, pr, wr, xr, Pr)):
        gen[key].append(v)
    w = (rng.uniform(0.9, 1.0, NP) if k % 3 == 0 else rng.exponential(1.0, NP) ** rng.uniform(2.0, 6.0))
    w = (w / w.sum()).astype(np.float32)
    assert abs(1.0 / float(np.sum(w.astype(np.float64) ** 2)) - 50.0) > 5.0
    p2, w2 = ref.pf_resampling(pr, w, uni)
    p2o, w2o, *_ = oracle.pf_step_parts(pr[None], w[None], z[None], np.array([nz], np.int32), u[None], nrm[None], uni[None], 2)
    same(p2o[0], p2, "resampling px"); same(w2o[0], w2, "resampling pw")
    for key, v in zip(res, (w, p2, w2)):
        res[key].append(v)
out.update({"gen_" + k: np.array(v) for k, v in gen.items()})
out.update({"res_" + k: np.array(v) for k, v in res.items()})
np.savez_compressed(os.path.join(HERE, "pf_golden.npz"), **out)
print("wrote pf_golden.npz", {k: np.asarray(v).shape for k, v in out.items()})
