#!/usr/bin/env python
"""Do two builds of libcrx.so give the SAME BITS on the MPC solve?  (Round 5: the build whose backward sweep recomputes the rollout's
trig — scripts/build_variants.sh lean="-DCRX_MPC_LEAN=1 -Wl,-Bsymbolic" — against the product.)  Each build runs in its own process
(CRX_LIB_PATH) on the same seeded problems and dumps sol / status / cost; the parent compares the bytes and prints one JSON line per set.
  python scripts/gpu_mpc_two_builds.py [variant]          (default variant: lean -> cpprobotics_amd/alt_lean.so)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [(8192, 21, s) for s in (4, 5, 6, 7)] + [(65536, 21, 11), (8192, 6, 3), (4096, 40, 9)]


def worker(out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import cpprobotics_amd as crx
    from common import mpc_problem
    res = {}
    for n, T, seed in SETS:
        x0, xref = mpc_problem(n, T, seed)
        x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
        sol, status, cost = crx.mpc_solve(x0, xref, T, return_status=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            crx.mpc_solve(x0, xref, T)
        b.record(); torch.cuda.synchronize()
        k = f"{n}_{T}_{seed}"
        res[k + "_sol"], res[k + "_status"], res[k + "_cost"] = sol.cpu().numpy(), status.cpu().numpy(), cost.cpu().numpy()
        res[k + "_ms"] = np.array(a.elapsed_time(b) / 5)
    np.savez(out, **res)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(sys.argv[2])
    import numpy as np
    variant = sys.argv[1] if len(sys.argv) > 1 else "lean"
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        for name, lib in (("product", None), (variant, os.path.join(ROOT, "cpprobotics_amd", f"alt_{variant}.so"))):
            env = dict(os.environ)
            env.pop("CRX_LIB_PATH", None)
            if lib:
                env["CRX_LIB_PATH"] = lib
            f = os.path.join(d, name + ".npz")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", f], env=env)
            outs[name] = dict(np.load(f))
    a, b = outs["product"], outs[variant]
    for n, T, seed in SETS:
        k = f"{n}_{T}_{seed}"
        cost_a, cost_b = a[k + "_cost"], b[k + "_cost"]
        line = {"agents": n, "T": T, "seed": seed, "ms_product": float(a[k + "_ms"]), f"ms_{variant}": float(b[k + "_ms"]),
                "sol_floats_differing": int((a[k + "_sol"].view(np.uint32) != b[k + "_sol"].view(np.uint32)).sum()),
                "status_differing": int((a[k + "_status"] != b[k + "_status"]).sum()),
                "cost_doubles_differing": int((cost_a.view(np.uint64) != cost_b.view(np.uint64)).sum()),
                "cost_max_rel_diff": float(np.max(np.abs(cost_a - cost_b) / np.maximum(np.abs(cost_a), 1e-300)))}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
