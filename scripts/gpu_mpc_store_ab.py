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
from cpprobotics_amd.experimental import mpc_solve_store
from cpprobotics_amd.mpc import mpc_n_vars

sizes = [int(a) for a in sys.argv[1:]] or [8192, 16384, 65536, 262144, 1048576]
dev = torch.device("cuda", 0)
T = 21
base_x0, base_xref = mpc_problem(65536, T, 4)
for n in sizes:
    reps = (n + 65535) // 65536
    x0 = torch.from_numpy(base_x0).repeat(reps, 1)[:n].contiguous().to(dev)
    xref = torch.from_numpy(base_xref).repeat(reps, 1)[:n].contiguous().to(dev)
    out = (torch.empty((n, mpc_n_vars(T)), dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
           torch.empty(n, dtype=torch.float64, device=dev))
    res = {"agents": n, "T": T}
    ref = None
    for store, name in ((0, "private"), (1, "tile")):
        mpc_solve_store(x0, xref, T, store, out=out)
        torch.cuda.synchronize()
        k = 5 if n <= 262144 else 3
        ts = []
        for _ in range(k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); mpc_solve_store(x0, xref, T, store, out=out); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        res[name] = {"ms": ms, "solves_per_s": n / (ms * 1e-3), "all_ms": ts}
        snap = tuple(t.clone() for t in out)
        if ref is None:
            ref = snap
        else:
            res["bit_identical"] = bool(torch.equal(ref[1], snap[1]) and torch.equal(ref[0].view(torch.int32), snap[0].view(torch.int32))
                                        and torch.equal(ref[2].view(torch.int64), snap[2].view(torch.int64)))
    res["tile_over_private"] = res["private"]["ms"] / res["tile"]["ms"]
    print(json.dumps(res), flush=True)
