#!/usr/bin/env python
"""A/B of the fused EKF launch: one vehicle per lane (production, ekf_run_kernel) against two lanes per vehicle with DPP
cross-lane moves (ekf_run_pair_kernel, the per-wavefront layout north_star sketches), at the BASELINE batch and at 1 M vehicles.
Writes the table of profiles/r02/ekf_wave_ab.txt to stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cpprobotics_amd as crx  # noqa: E402
from cpprobotics_amd.experimental import ekf_run_pair  # noqa: E402
from common import ekf_QR, ekf_agents  # noqa: E402

Q, R = ekf_QR()


def run(n, T, reps):
    u, x0, P0 = ekf_agents(n, 2024)
    w = crx.normal_draws(n, T, seed=0xC0FFEE)
    xT, xDR = torch.from_numpy(x0).cuda(), torch.from_numpy(x0).cuda()
    z, ud = crx.ekf_simulate_inputs(torch.from_numpy(u).cuda(), xT, xDR, w)
    del w
    xi, Pi = torch.from_numpy(x0).cuda(), torch.from_numpy(P0).cuda()
    out = {}
    res = {}
    for name, fn in (("lane_per_vehicle", lambda x, P, xh: crx.ekf_run(x, P, z, ud, Q, R, x_hist=xh)),
                     ("two_lanes_per_vehicle", lambda x, P, xh: ekf_run_pair(x, P, z, ud, Q, R, x_hist=xh))):
        x, P = xi.clone(), Pi.clone()
        xh = torch.empty((T, n, 4), device="cuda")
        for _ in range(max(3, reps // 4)):                       # settle
            x.copy_(xi); P.copy_(Pi); fn(x, P, xh)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            x.copy_(xi); P.copy_(Pi)
            a.record(); flag = fn(x, P, xh); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        res[name] = (x.cpu().numpy(), P.cpu().numpy(), xh[-1].cpu().numpy(), xh[T // 2].cpu().numpy())
        out[name] = {"ms_median": ms[len(ms) // 2], "ms_min": ms[0], "updates_per_s": n * T / (ms[len(ms) // 2] * 1e-3),
                     "hbm_frac": (32.0 * n * T + 160.0 * n) / (ms[len(ms) // 2] * 1e-3) / 8e12,
                     "waves": (n + 63) // 64 if name == "lane_per_vehicle" else (2 * n + 63) // 64}
        if name != "lane_per_vehicle":
            out[name]["left_fast_domain"] = int(not flag)
        del xh
    a, b = res["lane_per_vehicle"], res["two_lanes_per_vehicle"]
    out["results_equal_as_ieee_values"] = bool(all(np.array_equal(p, q) for p, q in zip(a, b)))
    out["speed_of_two_lane_variant"] = out["lane_per_vehicle"]["ms_median"] / out["two_lanes_per_vehicle"]["ms_median"]
    print(json.dumps({"vehicles": n, "steps_per_launch": T, **out}))


run(65536, 1000, 60)
run(131072, 500, 30)
run(1048576, 100, 20)
