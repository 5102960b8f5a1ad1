"""Bit-exact kernels against the CPU oracle on seeds and sizes beyond the fixed ones of tests/: fused EKF, DARE / dlqr (structured and
dense), lqr_steering_control, the LQR closed loops, the dynamic-window episode.  Prints the number of mismatching agents per family."""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
import oracle
from common import ekf_QR, ekf_agents, ekf_noise, lqr_course, lqr_speeds, tracking_agents
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = {}
Q, R = ekf_QR()
for seed in range(200, 206):
    rng = np.random.default_rng(seed)
    n, T = int(rng.integers(1, 3000)), int(rng.integers(1, 400))
    u, x0, P0 = ekf_agents(n, seed)
    z, ud, *_ = oracle.ekf_simulate_inputs(u, x0, x0, ekf_noise(T, n, seed + 1))
    xo, Po, xho, pho = oracle.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
    xd, Pd = t(x0), t(P0); xh = torch.empty((T, n, 4), device='cuda'); ph = torch.empty((T, n, 16), device='cuda')
    crx.ekf_run(xd, Pd, t(z), t(ud), Q, R, x_hist=xh, P_hist=ph)
    bad['ekf'] = bad.get('ekf', 0) + int((np.any(xh.cpu().numpy() != xho, axis=(0, 2)) | np.any(ph.cpu().numpy() != pho, axis=(0, 2))).sum())
    for dim in (5, 4):
        v = lqr_speeds(4096, seed + dim)
        v[:64] = rng.uniform(-0.2, 0.2, 64).astype(np.float32)
        A, B, Qm, Rm = oracle.lqr_build(v, dim)
        Xo, Ko, ito = oracle.dare(A, B, Qm, Rm)
        K, X, it = crx.dlqr_from_v(t(v), dim=dim)
        Xd, itd = crx.solve_DARE(t(A), t(B), t(Qm), t(Rm))
        m = np.any(X.cpu().numpy() != Xo, axis=1) | np.any(K.cpu().numpy() != Ko, axis=1) | (it.cpu().numpy() != ito) | np.any(Xd.cpu().numpy() != Xo, axis=1) | (itd.cpu().numpy() != ito)
        bad[f'dare{dim}'] = bad.get(f'dare{dim}', 0) + int(m.sum())
    course, goal = lqr_course()
    dc = crx.Course.from_numpy(course)
    for dim in (5, 4):
        st = tracking_agents(512, tuple(c[:120] for c in course), seed * 3 + dim, spread=0.5)
        so, to, ho, *_ = oracle.lqr_closed_loop(st, course, goal, dim=dim, max_ticks=600, want_hist=True)
        sd = t(st)
        ticks, hist = crx.closed_loop_prediction(sd, dc, goal, dim=dim, max_ticks=600, want_hist=True)
        ticks, hist = ticks.cpu().numpy(), hist.cpu().numpy()
        m = (ticks != to) | np.any(sd.cpu().numpy() != so, axis=1)
        for a in np.flatnonzero(~m):
            if not np.array_equal(hist[: to[a], a], ho[: to[a], a]): m[a] = True
        bad[f'loop{dim}'] = bad.get(f'loop{dim}', 0) + int(m.sum())
    O = oracle.oracle_lib
    n = 64
    st = np.stack([rng.uniform(-1, 9, n), rng.uniform(-1, 9, n), rng.uniform(-3.2, 3.2, n), rng.uniform(-0.5, 1.0, n), rng.uniform(-0.69, 0.69, n)], axis=1).astype(np.float32)
    uu = st[:, 3:5].copy(); g = np.stack([rng.uniform(8, 12, n), rng.uniform(8, 12, n)], axis=1).astype(np.float32)
    so, uo, to, ho = oracle.dwa_run(st, uu, g, 60, want_hist=True)
    sd, udv = t(st), t(uu)
    ticks, hist, *_ = crx.dwa_run(sd, udv, t(g), t(O.DWA_OBSTACLES), 60, want_hist=True)
    bad['dwa'] = bad.get('dwa', 0) + int(((ticks.cpu().numpy() != to) | np.any(sd.cpu().numpy() != so, axis=1)).sum())
print("mismatching agents per family over 6 seeds:", bad)
