#!/usr/bin/env python
"""The MPC solve with the backward sweep's trig stored by the rollout against recomputed in the sweep (crx_x_mpc_solve_trig_dev; the product
picks by batch size): same bits?  and which is faster where?  One JSON line per seeded problem set -> profiles/rNN/mpc_two_builds.jsonl.
(Until round 5 spelled the solver's fused multiply-adds out, the two disagreed in the last bits of ~63 % of the costs: the compiler's
contraction of a sum of products depended on the code around it.)
  python scripts/gpu_mpc_two_builds.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SETS = [(8192, 21, s) for s in (4, 5, 6, 7)] + [(65536, 21, 11), (131072, 21, 12), (196608, 21, 13), (262144, 21, 14), (1048576, 21, 15), (8192, 6, 3), (4096, 40, 9)]


def main():
    import numpy as np
    import torch
    from cpprobotics_amd import experimental as X
    from common import mpc_problem

    def timeit(fn, reps):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    for n, T, seed in SETS:
        m = min(n, 65536)
        x0, xref = mpc_problem(m, T, seed)
        x0, xref = torch.from_numpy(x0).cuda().repeat(n // m, 1), torch.from_numpy(xref).cuda().repeat(n // m, 1)
        a = X.mpc_solve_trig(x0, xref, T, 0)
        b = X.mpc_solve_trig(x0, xref, T, 1)
        reps = 5 if n <= 65536 else 2
        ca, cb = a[2].cpu().numpy(), b[2].cpu().numpy()
        line = {"agents": n, "T": T, "seed": seed,
                "ms_trig_stored": timeit(lambda: X.mpc_solve_trig(x0, xref, T, 0), reps), "ms_trig_recomputed": timeit(lambda: X.mpc_solve_trig(x0, xref, T, 1), reps),
                "sol_floats_differing": int((a[0].view(torch.int32) != b[0].view(torch.int32)).sum()),
                "status_differing": int((a[1] != b[1]).sum()),
                "cost_doubles_differing": int((ca.view(np.uint64) != cb.view(np.uint64)).sum())}
        print(json.dumps(line), flush=True)
        del x0, xref, a, b


if __name__ == "__main__":
    main()
