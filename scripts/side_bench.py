#!/usr/bin/env python
"""Side benchmark for the two compute-bound rows of SURVEY.md 8(d): batched DARE+dlqr (configs[2]) and batched MPC
(configs[3]).  Prints one JSON line per workload with throughput, the CPU-oracle rate on this host, and parity.
Used by scripts/gpu_side.sh under rocprofv3 to produce profiles/<round>/lqr_mpc_*.csv."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
import oracle  # noqa: E402
from common import lqr_course, lqr_speeds, mpc_course_f32, mpc_problem, tracking_agents  # noqa: E402


def gpu_time(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def cpu_parallel(fn, n, cores):
    """fn(a0, a1) over a static partition of [0,n) on `cores` threads; returns wall seconds."""
    bounds = [(i * n // cores, (i + 1) * n // cores) for i in range(cores)]
    th = [threading.Thread(target=fn, args=b) for b in bounds]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0


def bench_frenet(quick):
    # ---- Frenet optimal-trajectory planner: one agent per wavefront, 168 candidate paths per planning call ----------
    import ctypes as C
    O = oracle.oracle_lib
    n, max_ticks = 8192, 10
    course = crx.FrenetCourse(O.FRENET_WX, O.FRENET_WY)
    rng = np.random.default_rng(21)
    fst = np.stack([rng.uniform(0.0, 55.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n), rng.uniform(-0.8, 0.8, n),
                    rng.uniform(-0.5, 0.5, n)], axis=1).astype(np.float32)
    fsd, fob = torch.from_numpy(fst).cuda(), torch.from_numpy(O.FRENET_OBSTACLES).cuda()
    course.to(fsd.device)
    r = crx.frenet_run(fsd.clone(), course, fob, max_ticks)
    t_fr = gpu_time(lambda: crx.frenet_run(fsd.clone(), course, fob, max_ticks), 1 if quick else 3)
    tk = r["ticks"].cpu().numpy().astype(np.int64)
    P = crx.frenet_num_paths()
    ns = 512
    gomp = C.CDLL("libgomp.so.1")
    gomp.omp_get_max_threads.restype = C.c_int
    nthreads = gomp.omp_get_max_threads()
    t1 = time.perf_counter()
    ro = oracle.frenet_run(fst[:ns], course.coef, course.goal, max_ticks)
    t_cpu = time.perf_counter() - t1
    gomp.omp_set_num_threads(1)
    t1 = time.perf_counter()
    r1 = oracle.frenet_run(fst[:16], course.coef, course.goal, max_ticks)
    single = float(r1["ticks"].sum()) / (time.perf_counter() - t1)
    gomp.omp_set_num_threads(nthreads)
    # parity on one planning call of the first ns agents: costs, verdicts, winners
    po = oracle.frenet_plan(fst[:ns], course.coef)
    pg = crx.frenet_optimal_planning(fsd[:ns].clone(), course, fob, want_paths=True)
    cfg_, cfo = pg["path_cf"].cpu().numpy(), po["path_cf"]
    print(json.dumps({
        "workload": f"Frenet optimal-trajectory planner, {n} agents, {max_ticks} planning ticks, {P} candidate paths x ~25 time steps x {len(O.FRENET_OBSTACLES)} obstacles per call, one agent per wavefront",
        "plans_per_s": float(tk.sum()) / t_fr, "candidate_paths_per_s": float(tk.sum()) * P / t_fr, "ms": t_fr * 1e3,
        "mean_ticks": float(tk.mean()), "no_survivor_frac": float((r["status"].cpu().numpy() & 1).mean()),
        "cpu_baseline": {"value": float(ro["ticks"].sum()) / t_cpu, "unit": "plans/s", "cores": nthreads, "kind": "port",
                         "sample": f"first {ns} agents (OpenMP over agents)", "single_thread_value": single},
        "parity": {"path_cost_bit_identical_frac": float((cfg_.view(np.uint32) == cfo.view(np.uint32)).mean()),
                   "path_cost_max_rel_err": float(np.nanmax(np.abs(cfg_ - cfo) / np.maximum(np.abs(cfo), 1e-30))),
                   "verdicts_identical_frac": float((pg["path_ok"].cpu().numpy() == po["path_ok"]).mean()),
                   "winner_identical_frac": float((pg["best_idx"].cpu().numpy() == po["best"]).mean())}}))


def main():
    cores = os.cpu_count() or 1
    quick = "--quick" in sys.argv
    if "--frenet-only" in sys.argv:
        return bench_frenet(quick)
    # ---- DARE + dlqr, 16,384 agents (configs[2]) ------------------------------------------------------
    n = 16384
    v = lqr_speeds(n, 3)
    vd = torch.from_numpy(v).cuda()
    for dim in (5, 4):
        A, B, Q, R = oracle.lqr_build(v, dim)
        Ad, Bd, Qd, Rd = (torch.from_numpy(a).cuda() for a in (A, B, Q, R))
        K, X, it = crx.dlqr_from_v(vd, dim=dim)
        t_struct = gpu_time(lambda: crx.dlqr_from_v(vd, dim=dim), 5 if quick else 20)
        t_dense = gpu_time(lambda: crx.dlqr(Ad, Bd, Qd, Rd), 3 if quick else 10)
        iters = it.cpu().numpy()
        Xo = np.zeros_like(A); Ko = np.zeros((n, (2 if dim == 5 else 1) * dim), np.float32); ito = np.zeros(n, np.int32)
        import ctypes as C
        lib = oracle.oracle_lib.lib(); vp = lambda a: a.ctypes.data_as(C.c_void_p)

        def work(a0, a1):
            lib.oracle_dare(C.c_int(n), C.c_int(dim), vp(A), vp(B), vp(Q), vp(R), C.c_float(0.01), C.c_int(150), vp(Xo), vp(Ko),
                            vp(ito), C.c_int(0), C.c_int(a0), C.c_int(a1))
        t1 = time.perf_counter(); work(0, 64); single = 64 / (time.perf_counter() - t1)
        t_cpu = min(cpu_parallel(work, n, cores) for _ in range(2))
        flop_iter = {5: 300.0, 4: 150.0}[dim]           # structured kernel, as executed
        flops = float(iters.sum()) * flop_iter
        print(json.dumps({
            "workload": f"DARE+dlqr {dim}x{dim}, {n} agents, v~U(-3,6) with 5% |v|<0.1 (BASELINE configs[2])",
            "solves_per_s_structured": n / t_struct, "solves_per_s_dense": n / t_dense, "ms_structured": t_struct * 1e3,
            "mean_iters": float(iters.mean()), "max_iters": int(iters.max()),
            "gflops_structured": flops / t_struct / 1e9,
            "cpu_baseline": {"value": n / t_cpu, "unit": "solves/s", "cores": cores, "kind": "port", "single_thread_value": single},
            "parity": {"X_bit_identical": bool(np.array_equal(X.cpu().numpy(), Xo)), "K_bit_identical": bool(np.array_equal(K.cpu().numpy(), Ko)),
                       "iters_identical": bool(np.array_equal(iters, ito))}}))
    # ---- MPC, 8,192 agents, T = 21 (configs[3]) ------------------------------------------------------------
    for n, T in ((8192, 21), (8192, 6)):
        x0, xref = mpc_problem(n, T, 4)
        x0d, xrd = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
        sol, st, cost = crx.mpc_solve(x0d, xrd, T, return_status=True)
        t_gpu = gpu_time(lambda: crx.mpc_solve(x0d, xrd, T), 2 if quick else 5)
        st = st.cpu().numpy(); sol = sol.cpu().numpy()
        ns = 2048
        t1 = time.perf_counter(); so, sto, co = oracle.mpc_solve(x0[:16], xref[:16], T); single = 16 / (time.perf_counter() - t1)
        so = np.zeros((ns, sol.shape[1]), np.float32); sto = np.zeros(ns, np.int32); co = np.zeros(ns)
        pp = oracle.oracle_lib._mpc_params(None)
        import ctypes as C
        lib = oracle.oracle_lib.lib(); vp = lambda a: a.ctypes.data_as(C.c_void_p)
        xs, xr = np.ascontiguousarray(x0[:ns]), np.ascontiguousarray(xref[:ns])

        def work(a0, a1):
            lib.oracle_mpc_solve(C.c_int(ns), C.c_int(T), vp(xs), vp(xr), vp(pp), C.c_int(50), vp(so), vp(sto), vp(co), C.c_int(a0), C.c_int(a1))
        t_cpu = min(cpu_parallel(work, ns, min(cores, ns // 8)) for _ in range(2))
        both = ((sto & 1) == 1) & ((st[:ns] & 1) == 1)
        err = float(np.max(np.abs(sol[:ns][both] - so[both]) / np.maximum(np.abs(so[both]), 1.0)))
        print(json.dumps({
            "workload": f"MPC speed+steer, {n} agents, T={T} knots ({T - 1} control intervals) (BASELINE configs[3])" if T == 21 else
                        f"MPC speed+steer, {n} agents, T={T} (the reference's own horizon)",
            "solves_per_s": n / t_gpu, "ms": t_gpu * 1e3, "converged_frac": float((st & 1).mean()),
            "mean_iters": float((st >> 8).mean()), "max_iters": int((st >> 8).max()),
            "cpu_baseline": {"value": ns / t_cpu, "unit": "solves/s", "cores": min(cores, ns // 8), "kind": "port",
                             "sample": f"first {ns} agents", "single_thread_value": single},
            "parity": {"max_rel_err_floored_vs_cpu_twin": err, "both_converged_frac": float(both.mean())}}))

    # ---- course tracking: one control evaluation, and the closed LQR loop as one kernel ----
    course, goal = lqr_course()
    dc = crx.Course.from_numpy(course)
    n = 16384
    st = tracking_agents(n, tuple(c[:200] for c in course), 5, spread=0.4)
    for dim in (5, 4):
        std = torch.from_numpy(st).cuda()
        pe, pth = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        t_ctl = gpu_time(lambda: crx.lqr_steering_control(std, dc, pe, pth, dim=dim), 3 if quick else 10)
        max_ticks = 400
        ticks, _ = crx.closed_loop_prediction(std.clone(), dc, goal, dim=dim, max_ticks=max_ticks)
        tk = ticks.cpu().numpy().astype(np.int64)
        t_loop = gpu_time(lambda: crx.closed_loop_prediction(std.clone(), dc, goal, dim=dim, max_ticks=max_ticks), 1 if quick else 3)
        ns = 512
        t1 = time.perf_counter()
        so, tio, *_ = oracle.lqr_closed_loop(st[:ns], course, goal, dim=dim, max_ticks=max_ticks, agents=(0, 8))
        single = float(tio[:8].sum()) / (time.perf_counter() - t1)
        res = {}

        def work(a0, a1):
            res[a0] = oracle.lqr_closed_loop(st[:ns], course, goal, dim=dim, max_ticks=max_ticks, agents=(a0, a1))
        t_cpu = cpu_parallel(work, ns, min(cores, ns // 4))
        # parity of the sampled agents: final state + tick count
        sd = std.clone(); tk2, _ = crx.closed_loop_prediction(sd, dc, goal, dim=dim, max_ticks=max_ticks)
        same = True
        for a0, r in res.items():
            a1 = min(ns, a0 + (ns // min(cores, ns // 4)) + 1)
        full = oracle.lqr_closed_loop(st[:64], course, goal, dim=dim, max_ticks=max_ticks)
        same = bool(np.array_equal(sd.cpu().numpy()[:64], full[0]) and np.array_equal(tk2.cpu().numpy()[:64], full[1]))
        print(json.dumps({
            "workload": f"LQR tracking {dim}-state, {n} agents on the reference course ({len(course[0])} points): "
                        f"lqr_steering_control + update + goal test per tick, whole episode in one kernel (max {max_ticks} ticks)",
            "agent_ticks_per_s": float(tk.sum()) / t_loop, "episodes_per_s": n / t_loop, "ms_episode_batch": t_loop * 1e3,
            "mean_ticks": float(tk.mean()), "reached_goal_frac": float((tk < max_ticks).mean()),
            "control_evals_per_s_single_launch": n / t_ctl,
            "cpu_baseline": {"value": float(tk[:ns].sum()) / t_cpu, "unit": "agent-ticks/s", "cores": min(cores, ns // 4), "kind": "port",
                             "sample": f"first {ns} agents", "single_thread_value": single},
            "parity": {"first_64_agents_bit_identical": same}}))
    # ---- MPC closed loop (mpc_simulation): one persistent kernel for the episode ---------------------------------------------------------
    mcourse, mgoal = mpc_course_f32()
    mdc = crx.Course.from_numpy(mcourse)
    n, T, max_ticks = 8192, 6, 50
    mst = tracking_agents(n, tuple(c[:150] for c in mcourse), 9, spread=0.5)
    mst[:, 3] = np.random.default_rng(10).uniform(0.5, 4.0, n).astype(np.float32)
    tind0 = torch.from_numpy(oracle.calc_nearest_index(mst, mcourse)[0].astype(np.int32)).cuda()
    mstd = torch.from_numpy(mst).cuda()
    t_mpc = gpu_time(lambda: crx.mpc_simulation(mstd.clone(), mdc, mgoal, T, max_ticks, target_ind=tind0.clone()), 1 if quick else 2)
    print(json.dumps({"workload": f"MPC closed loop (mpc_simulation), {n} agents, T={T}, {max_ticks} ticks: calc_ref_trajectory + mpc_solve + update + goal test per tick, one persistent kernel",
                      "agent_ticks_per_s": n * max_ticks / t_mpc, "ms_per_tick": t_mpc / max_ticks * 1e3}))

    # ---- particle filter: one vehicle per wavefront, T fused ticks ---------------------------------
    n, T, NP = 16384, 100, 100
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    u1 = np.tile(np.array([[1.0, 0.1]], np.float32), (n, 1))
    rng = np.random.default_rng(8)
    w_u = rng.standard_normal((T, n, 2)).astype(np.float32); w_z = rng.standard_normal((T, n, 4)).astype(np.float32)
    ud, obs, nobs, xth, xdh = oracle.pf_simulate_inputs(u1, np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), w_u, w_z)
    ut = np.tile(u1[None], (T, 1, 1))
    nrm = torch.randn((T, n, NP, 2), generator=g, device="cuda")
    uni = torch.rand((T, n, NP), generator=g, device="cuda") + 1.0
    obs_d, nobs_d, ut_d = torch.from_numpy(obs).cuda(), torch.from_numpy(nobs).cuda(), torch.from_numpy(ut).cuda()
    px0 = torch.zeros((n, NP, 4), device="cuda"); pw0 = torch.full((n, NP), 1.0 / NP, device="cuda")
    xe, Pe, hist, nres = crx.pf_run(px0.clone(), pw0.clone(), obs_d, nobs_d, ut_d, nrm, uni)
    t_pf = gpu_time(lambda: crx.pf_run(px0.clone(), pw0.clone(), obs_d, nobs_d, ut_d, nrm, uni), 2 if quick else 5)
    h = hist.cpu().numpy()
    err = float(np.hypot(h[..., 0] - xth[..., 0], h[..., 1] - xth[..., 1]).mean())
    ns = 256
    nrm_h, uni_h = nrm[:, :ns].cpu().numpy(), uni[:, :ns].cpu().numpy()
    t1 = time.perf_counter()
    ro = oracle.pf_run(np.zeros((ns, NP, 4), np.float32), np.full((ns, NP), 1.0 / NP, np.float32), obs[:, :ns], nobs[:, :ns], ut[:, :ns],
                       nrm_h, uni_h, agents=(0, 8))
    single = 8 * T / (time.perf_counter() - t1)
    res = {}

    def work(a0, a1):
        res[a0] = oracle.pf_run(np.zeros((ns, NP, 4), np.float32), np.full((ns, NP), 1.0 / NP, np.float32), obs[:, :ns], nobs[:, :ns],
                                ut[:, :ns], nrm_h, uni_h, agents=(a0, a1))
    t_cpu = cpu_parallel(work, ns, min(cores, ns // 2))
    in_bytes = (NP * 2 * 4 + NP * 4 + 4 * 3 * 4 + 4 + 8) + 16
    print(json.dumps({
        "workload": f"particle filter, {n} vehicles x {T} ticks x {NP} particles, one vehicle per wavefront (pf_localization + resampling)",
        "vehicle_ticks_per_s": n * T / t_pf, "particle_updates_per_s": n * T * NP / t_pf, "ms": t_pf * 1e3,
        "GB_per_s_inputs": n * T * in_bytes / t_pf / 1e9, "mean_position_error_m": err,
        "resampling_fraction": float(nres.float().mean().item()) / T,
        "cpu_baseline": {"value": ns * T / t_cpu, "unit": "vehicle-ticks/s", "cores": min(cores, ns // 2), "kind": "port",
                         "sample": f"first {ns} vehicles", "single_thread_value": single}}))

    # ---- dynamic-window planner: one agent per wavefront, whole episode fused ------------------------
    n, max_ticks = 8192, 200
    rng = np.random.default_rng(12)
    dst = np.stack([rng.uniform(-1, 3, n), rng.uniform(-1, 3, n), rng.uniform(0, 1.2, n), np.zeros(n), np.zeros(n)], axis=1).astype(np.float32)
    dgoal = np.tile(np.array([[10.0, 10.0]], np.float32), (n, 1))
    du = np.zeros((n, 2), np.float32)
    ob = oracle.oracle_lib.DWA_OBSTACLES
    dsd, dud, dgd, obd = (torch.from_numpy(a).cuda() for a in (dst, du, dgoal, ob))
    tk, _, stt, _, nsamp = crx.dwa_run(dsd.clone(), dud.clone(), dgd, obd, max_ticks)
    t_dwa = gpu_time(lambda: crx.dwa_run(dsd.clone(), dud.clone(), dgd, obd, max_ticks), 1 if quick else 3)
    tk = tk.cpu().numpy().astype(np.int64)
    ns = 128
    t1 = time.perf_counter()
    ro = oracle.dwa_run(dst[:ns], du[:ns], dgoal[:ns], max_ticks, agents=(0, 4))
    single = float(ro[2][:4].sum()) / (time.perf_counter() - t1)
    res = {}

    def work(a0, a1):
        res[a0] = oracle.dwa_run(dst[:ns], du[:ns], dgoal[:ns], max_ticks, agents=(a0, a1))
    t_cpu = cpu_parallel(work, ns, min(cores, ns))
    full = oracle.dwa_run(dst[:32], du[:32], dgoal[:32], max_ticks)
    s2, u2 = dsd.clone(), dud.clone()
    tk2, *_ = crx.dwa_run(s2, u2, dgd, obd, max_ticks)
    same = bool(np.array_equal(s2.cpu().numpy()[:32], full[0]) and np.array_equal(tk2.cpu().numpy()[:32], full[2]))
    print(json.dumps({
        "workload": f"dynamic-window planner, {n} agents, up to {max_ticks} control steps, ~405 sampled trajectories x 31 steps x {len(ob)} obstacles per step, one agent per wavefront",
        "agent_steps_per_s": float(tk.sum()) / t_dwa, "trajectories_per_s": float(tk.sum()) * 405 / t_dwa, "ms": t_dwa * 1e3,
        "mean_ticks": float(tk.mean()), "reached_goal_frac": float((tk < max_ticks).mean()),
        "cpu_baseline": {"value": float(tk[:ns].sum()) / t_cpu, "unit": "agent-steps/s", "cores": min(cores, ns), "kind": "port",
                         "sample": f"first {ns} agents", "single_thread_value": single},
        "parity": {"first_32_agents_bit_identical": same}}))

    bench_frenet(quick)


if __name__ == "__main__":
    main()
