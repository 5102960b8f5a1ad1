import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx, oracle
from common import *
oracle.build()
Q, R = ekf_QR()
n, T = 100, 8
u, x0, P0 = ekf_agents(n, n + T)
w = ekf_noise(T, n, n + T + 1000)
z, ud, *_ = oracle.ekf_simulate_inputs(u, x0, x0, w)
xo, Po, xho, pho = oracle.ekf_run(x0, P0, z, ud, Q, R, want_phist=True)
t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
xd, Pd = t_(x0), t_(P0)
xh = torch.zeros((T, n, 4), device='cuda'); ph = torch.full((T, n, 16), -7.0, device='cuda')
crx.ekf_run(xd, Pd, t_(z), t_(ud), Q, R, x_hist=xh, P_hist=ph)
ph = ph.cpu().numpy()
bad = np.argwhere(ph != pho)
print("mismatches", len(bad))
print("by t:", np.bincount(bad[:, 0], minlength=T))
print("by col:", np.bincount(bad[:, 2] // 4, minlength=4))
print("vehicles:", np.unique(bad[:, 1])[:20], len(np.unique(bad[:, 1])))
t, a, e = bad[0]
print("first", t, a, e, ph[t, a], pho[t, a])
