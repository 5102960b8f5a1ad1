"""Per-launch duration of the headline kernel over a long run (clock ramp / throttling picture)."""
import sys, time, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
from common import *
Q, R = ekf_QR()
n, T = 65536, 1000
u, x0, P0 = ekf_agents(n, 2024)
g = torch.Generator(device='cuda'); g.manual_seed(1234)
w = torch.randn((T, n, 4), generator=g, device='cuda')
xT, xDR = torch.from_numpy(x0).cuda(), torch.from_numpy(x0).cuda()
z, ud = crx.ekf_simulate_inputs(torch.from_numpy(u).cuda(), xT, xDR, w); del w
xi, Pi = torch.from_numpy(x0).cuda(), torch.from_numpy(P0).cuda()
x, P = xi.clone(), Pi.clone()
xh = torch.empty((T, n, 4), device='cuda')
N = 300
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
torch.cuda.synchronize()
for a, b in evs:
    x.copy_(xi); P.copy_(Pi)
    a.record(); crx.ekf_run(x, P, z, ud, Q, R, x_hist=xh); b.record()
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in evs])
print("first 12:", np.round(ms[:12], 3))
for lo in range(0, N, 50):
    print(f"launches {lo:3d}-{lo+49:3d}: mean {ms[lo:lo+50].mean():.4f} min {ms[lo:lo+50].min():.4f} max {ms[lo:lo+50].max():.4f}")
