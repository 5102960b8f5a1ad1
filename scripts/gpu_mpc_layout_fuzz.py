"""Every layout / scheduler of the MPC solve against crx::mpc_kernel on further draws of the configs[3] distribution, bit for bit
(status with its sweep count, every solution float, the double cost): the tile layout (the product's kernel from 131,072 agents on),
the checkpointed tile layout, lanes refilled, the phased solve on both layouts, two phases.
usage (gpurun): python scripts/gpu_mpc_layout_fuzz.py [seed0=5000] [seeds=12] [agents=262144] > gpurun_out/<tag>/mpc_layout_fuzz.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import cpprobotics_amd as crx  # noqa: F401
from common import mpc_problem
from cpprobotics_amd import experimental as X

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
T = 21
dev = torch.device("cuda", 0)
tail = torch.cuda.Stream()
bad = {}
total = 0
for seed in range(seed0, seed0 + seeds):
    hx0, hxref = mpc_problem(n, T, seed)
    x0, xref = torch.from_numpy(hx0).to(dev), torch.from_numpy(hxref).to(dev)
    ref = X.mpc_solve_store(x0, xref, T, 0)
    prod = crx.mpc_solve(x0, xref, T, return_status=True)

    def two_phase():
        r = X.mpc_solve_two_phase(x0, xref, T, 9, tail_stream=tail)
        torch.cuda.current_stream().wait_stream(tail)
        return r
    variants = {"product entry point": lambda: prod, "tile": lambda: X.mpc_solve_store(x0, xref, T, 1), "tile2": lambda: X.mpc_solve_store(x0, xref, T, 2),
                "tile refilled": lambda: X.mpc_solve_tile_refill(x0, xref, T, max(128, n // 1024), 16),
                "tile2 refilled": lambda: X.mpc_solve_tile_refill(x0, xref, T, max(128, n // 1024), 16, store=2),
                "phased private": lambda: X.mpc_solve_phased(x0, xref, T, (6, 8, 10, 13)), "phased tile": lambda: X.mpc_solve_phased(x0, xref, T, (6, 8, 10, 13), store=1),
                "two phases": two_phase}
    line = {}
    for name, fn in variants.items():
        r = fn()
        torch.cuda.synchronize()
        diff = (r[1] != ref[1]) | (r[0].view(torch.int32) != ref[0].view(torch.int32)).any(dim=1) | (r[2].view(torch.int64) != ref[2].view(torch.int64))
        line[name] = int(diff.sum().item())
        bad[name] = bad.get(name, 0) + line[name]
    total += n
    sw = (ref[1] >> 8)
    print(f"seed {seed}: sweeps mean {sw.float().mean().item():.3f} max {int(sw.max().item())} converged {((ref[1] & 1) == 1).float().mean().item():.6f}  differing agents {line}", flush=True)
print(f"differing agents per variant over seeds {seed0}..{seed0 + seeds - 1}, {total} problems each: {bad}")
