"""Kernel vs CPU twin on many seeds (beyond the fixed seeds of tests/test_mpc_gpu.py): sweeps, status, cost, solution."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
from oracle import oracle_lib as ol
from common import mpc_problem, mpc_solve_threads, floored_rel_err
tot = 0; bad_it = 0; bad_sol = 0; bad_cost = 0; bad_status = 0
for T in (21, 6):
    for seed in range(100, 112):
        n = 4096
        x0, xref = mpc_problem(n, T, seed)
        so, sto, co = mpc_solve_threads(ol, x0, xref, T)
        sd, std, cd = crx.mpc_solve(torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda(), T, return_status=True)
        sd, std, cd = sd.cpu().numpy(), std.cpu().numpy(), cd.cpu().numpy()
        dit = np.abs((std >> 8) - (sto >> 8))
        conv = (sto & 1) == 1
        err = np.abs(sd - so).max(1) / 1.0
        crel = np.abs(cd - co) / np.maximum(np.abs(co), 1.0)
        b1 = (dit > 1); b2 = conv & ((np.abs(sd - so) / np.maximum(np.abs(so), 1.0)).max(1) > 1e-6); b3 = conv & (crel > 1e-9); b4 = (std & 3) != (sto & 3)
        tot += n; bad_it += b1.sum(); bad_sol += b2.sum(); bad_cost += b3.sum(); bad_status += b4.sum()
        if b1.any() or b2.any() or b3.any() or b4.any():
            i = int(np.flatnonzero(b1 | b2 | b3 | b4)[0])
            print(f"T={T} seed={seed}: sweeps>1 {b1.sum()} sol {b2.sum()} cost {b3.sum()} status {b4.sum()}; e.g. agent {i}: sweeps {std[i] >> 8} vs {sto[i] >> 8}, cost {cd[i]:.12f} vs {co[i]:.12f}")
print(f"{tot} problems: sweep count differs by > 1: {bad_it}; solution > 1e-6: {bad_sol}; cost > 1e-9: {bad_cost}; status bits: {bad_status}")
