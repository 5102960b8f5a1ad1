"""In-kernel shader ticks of the MPC solve per wave (a -DCRX_MPC_TICKS=1 build of the library, CRX_LIB_PATH pointing at it: lane 0
of every wave reports the wave's ticks in `cost`), regressed on the wave's number of backward sweeps and candidate rollouts
(counted on the CPU twin: profiles/r02/mpc_wave_counts_seed4.npz) -> ticks of one backward sweep and of one rollout."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
from common import mpc_problem
n, T = 8192, 21
x0, xref = mpc_problem(n, T, 4)
x0, xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
for _ in range(3):
    sol, st, cost = crx.mpc_solve(x0, xref, T, return_status=True)
torch.cuda.synchronize()
tots = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sol, st, cost = crx.mpc_solve(x0, xref, T, return_status=True); b.record(); torch.cuda.synchronize()
    tots.append(cost.cpu().numpy().reshape(-1, 64)[:, 0]); ms = a.elapsed_time(b)
tot = np.median(np.stack(tots), 0)
c = np.load('profiles/r02/mpc_wave_counts_seed4.npz')
nb, nf = c['sweeps'].astype(float) + 1.0, c['trials'].astype(float)
A = np.stack([np.ones_like(nb), nb, nf], 1)
coef, res, *_ = np.linalg.lstsq(A, tot, rcond=None)
pred = A @ coef
print(f"launch {ms:.4f} ms; slowest wave {tot.max():.0f} ticks ({tot.max() / ms / 1e3:.0f} MHz); fit ticks = {coef[0]:.0f} + {coef[1]:.0f} x backward sweeps + {coef[2]:.0f} x rollouts;"
      f" residual rms {np.sqrt(np.mean((pred - tot) ** 2)):.0f} of mean {tot.mean():.0f}")
print(f"   per stage: backward {coef[1] / (T - 1):.0f} ticks, rollout {coef[2] / (T - 1):.0f} ticks")
w = int(np.argmax(tot)); print(f"   slowest wave {w}: {nb[w]:.0f} backward sweeps, {nf[w]:.0f} rollouts")
