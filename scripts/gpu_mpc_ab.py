"""MPC kernel time for the library in CRX_LIB_PATH (A/B of build variants) and for forced launch geometries
(`python scripts/gpu_mpc_ab.py [agents_per_wave [waves_per_workgroup]]`, through crx_x_mpc_solve_geometry_dev): BASELINE configs[3] and T = 6."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
from cpprobotics_amd.experimental import mpc_solve_geometry
from common import mpc_problem
live = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for n, T in ((8192, 21), (8192, 6), (16384, 21), (65536, 21)):
    x0, xref = mpc_problem(n, T, 4)
    x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    for _ in range(3):
        sol, st, cost = mpc_solve_geometry(x0, xref, T, live, wg)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(os.environ.get("CRX_REPS", "15")))]
    for a, b in evs:
        a.record(); mpc_solve_geometry(x0, xref, T, live, wg); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    it = (st.cpu().numpy() >> 8)
    print(f"agents/wave={live} waves/wg={wg}", f"n={n} T={T}: median {ms[len(ms) // 2]:.4f} ms min {ms[0]:.4f}; sweeps mean {it.mean():.2f} max {it.max()}; cost sum {cost.sum().item():.9f}")
