"""A/B of the MPC solve's working-set layouts on one MI355X: private memory (crx::mpc_kernel) against the tile layout
(crx::mpc_tile_kernel: LDS + accumulator registers), same problems, bit-identical answers (checked), by batch size.
usage (gpurun): python scripts/gpu_mpc_store_ab.py [sizes ...] > gpurun_out/<tag>/mpc_store_ab.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import cpprobotics_amd as crx
from common import mpc_problem
from cpprobotics_amd.experimental import mpc_solve_store, mpc_solve_tile_refill
from cpprobotics_amd.mpc import mpc_n_vars

sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384, 65536, 262144, 1048576]
dev = torch.device("cuda", 0)
T = 21
pw = None
for n in sizes:
    pw = None
    # the configs[3] distribution at this size, NOT a tiled copy of a smaller draw: the largest draws hold the rare agents that need
    # the full 50 sweeps, and those set the tail of a launch whose lanes are refilled
    hx0, hxref = mpc_problem(n, T, 4)
    x0, xref = torch.from_numpy(hx0).to(dev), torch.from_numpy(hxref).to(dev)
    out = (torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
           torch.empty(n, dtype=torch.float64, device=dev))
    res = {"agents": n, "T": T}
    st0 = mpc_solve_store(x0, xref, T, 0)[1].cpu().numpy()
    sw = (st0 >> 8).astype("int64")
    res["sweeps"] = {"mean": float(sw.mean()), "max": int(sw.max()), "mean_of_wave_max": float(sw[: n // 64 * 64].reshape(-1, 64).max(axis=1).mean()),
                     "at_cap_50": int((sw >= 50).sum()), "converged_frac": float((st0 & 1).mean())}
    ref = None
    variants = [("private", lambda: mpc_solve_store(x0, xref, T, 0, out=out)), ("tile", lambda: mpc_solve_store(x0, xref, T, 1, out=out)),
                ("tile2", lambda: mpc_solve_store(x0, xref, T, 2, out=out)), ("tile_lite", lambda: mpc_solve_store(x0, xref, T, 3, out=out))]
    from cpprobotics_amd.experimental import mpc_solve_two_phase
    wk = torch.empty(n + 64, dtype=torch.int32, device=dev)
    for K in (8, 9, 10, 12):
        variants.append((f"two_phase_{K}", (lambda k: (lambda: mpc_solve_two_phase(x0, xref, T, k, out=out, work=wk)))(K)))
    from cpprobotics_amd.experimental import mpc_solve_phased
    pw = None
    for st in (0, 1):
        for caps in ((6, 8, 10, 13), (7, 10), (6, 9, 12), (6, 7, 8, 9, 10, 12, 16)) + (((5, 6, 7, 8, 9, 10, 11, 12, 14, 17, 22, 30), (6, 8, 10, 13, 20, 30), (8,), (8, 12)) if st else ()):
            def f(c=caps, q=st):
                global pw
                pw = mpc_solve_phased(x0, xref, T, c, out=out, work=pw, store=q)[3]
            variants.append((("phased_" if st == 0 else "phased_tile_") + "_".join(str(c) for c in caps), f))
    if n >= 65536:
        for apw, hold in ((n // 1024, 16), (n // 2048, 16), (max(64, n // 4096), 32)):
            if apw >= 128:
                for st in (1, 2):
                    variants.append((f"tile{st}_refill_{apw}_{hold}", (lambda a, h, q: (lambda: mpc_solve_tile_refill(x0, xref, T, a, h, out=out, store=q)))(apw, hold, st)))
    for name, fn in variants:
        fn()
        torch.cuda.synchronize()
        k = 5 if n <= 262144 else 3
        ts = []
        for _ in range(k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        res[name] = {"ms": ms, "solves_per_s": n / (ms * 1e-3), "all_ms": ts}
        snap = tuple(t.clone() for t in out)
        if ref is None:
            ref = snap
        else:
            res[name]["bit_identical"] = bool(torch.equal(ref[1], snap[1]) and torch.equal(ref[0].view(torch.int32), snap[0].view(torch.int32))
                                              and torch.equal(ref[2].view(torch.int64), snap[2].view(torch.int64)))
            res[name]["over_private"] = res["private"]["ms"] / res[name]["ms"]
        del res[name]["all_ms"]
    print(json.dumps(res), flush=True)
