#!/usr/bin/env python
"""A/B of the one-lane-per-agent MPC solve's kernels, standalone launches at 8,192 .. 1 M agents (T = 21): mpc_kernel (lockstep) and
mpc_refill_kernel at several range lengths.  Every one must reproduce mpc_kernel's bits.  (Round 5 also measured a private copy of the
reference trajectory, a 256-register build and a trig-storing build beside the recomputing one: profiles/r05/mpc_variants_ab_run*.jsonl.)  One JSON line per batch size -> profiles/rNN/mpc_variants_ab.jsonl.
  python scripts/gpu_mpc_variants_ab.py [sizes ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch

    import cpprobotics_amd as crx
    from cpprobotics_amd import experimental as X
    from common import mpc_problem
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384, 65536, 262144, 1048576]
    T = 21

    def timeit(fn, reps):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    for n in sizes:
        x0, xref = mpc_problem(n, T, 4)
        x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
        base = X.mpc_solve_lanes(x0, xref, T, lanes_per_agent=1)                 # mpc_kernel forced (the product refills from 65,536 on)
        sw = (base[1].cpu().numpy() >> 8).astype(np.int64)
        reps = 5 if n <= 65536 else 2
        line = {"agents": n, "T": T, "sweeps_mean": float(sw.mean()), "sweeps_max": int(sw.max()),
                "ms_mpc_kernel": timeit(lambda: X.mpc_solve_lanes(x0, xref, T, lanes_per_agent=1), reps),
                "ms_product_entry": timeit(lambda: crx.mpc_solve(x0, xref, T), reps)}
        same = True
        variants = []
        if n >= 16384:
            variants += [("refill_%d" % c, c) for c in (128, 256, 512, 1024) if n // c >= 64]
        for name, apw in variants:
            out = X.mpc_solve_refill(x0, xref, T, apw, 16)
            same = same and all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(out, base))
            line["ms_" + name] = timeit(lambda: X.mpc_solve_refill(x0, xref, T, apw, 16, poison=False), reps)
        line["bit_identical"] = bool(same)
        best = min((k for k in line if k.startswith("ms_")), key=lambda k: line[k])
        line["best"] = best
        line["solves_per_s_best"] = n / (line[best] * 1e-3)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
