"""Where a round of scripts/swarm_bench.py goes on one GPU: each launch timed alone (HIP events), and the MPC sweeps of the shard."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx
from common import ekf_QR, mpc_course_f32
dev = torch.device("cuda", 0)
n, T, Tm = 131072, 100, 21
n_mpc = n // 8
Q, R = ekf_QR()
course, goal = mpc_course_f32()
dc = crx.Course.from_numpy(course, device=dev)
ci = torch.from_numpy(np.random.default_rng(99).integers(0, len(course[0]) - 30, n)).to(dev)
cx, cy, cyaw = (torch.from_numpy(a).to(dev) for a in course[:3])
x0 = torch.stack([cx[ci], cy[ci], cyaw[ci], torch.full((n,), 2.5, device=dev)], dim=1).contiguous()
u_true = torch.zeros((n, 2), device=dev)
w = crx.normal_draws(n, T, agent0=0, seed=99, device=dev)
z, ud = crx.ekf_simulate_inputs(u_true, x0.clone(), x0.clone(), w)
P0 = torch.eye(4, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
x, P = x0.clone(), P0.clone()
x_hist = torch.empty((T, n, 4), device=dev)
tind = torch.zeros(n_mpc, dtype=torch.int32, device=dev)
st = torch.empty((n_mpc, 4), device=dev)

def timed(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    return sorted(ms)[len(ms) // 2]

def ekf():
    x.copy_(x0); P.copy_(P0); crx.ekf_run(x, P, z, ud, Q, R, x_hist=x_hist)
print("ekf launch (131072 x 100)      %.3f ms" % timed(ekf))
st.copy_(x[::8]); st[:, 3] = 2.5        # pose from the filter, commanded speed (see swarm_bench.py)
print("calc_nearest_index (whole course) %.3f ms" % timed(lambda: crx.calc_nearest_index(st, dc, tind)))
print("calc_ref_trajectory              %.3f ms" % timed(lambda: crx.calc_ref_trajectory(st, dc, tind, Tm)))
xref = crx.calc_ref_trajectory(st, dc, tind, Tm)
print("mpc_solve (16384, T=21)          %.3f ms" % timed(lambda: crx.mpc_solve(st, xref, Tm)))
sol, status, cost = crx.mpc_solve(st, xref, Tm, return_status=True)
it = (status.cpu().numpy() >> 8); conv = status.cpu().numpy() & 1
print("   sweeps mean %.2f max %d; converged %.4f; histogram >= 15: %s" % (it.mean(), it.max(), conv.mean(), np.bincount(it)[15:]))
print("   per-wave max sweeps: mean %.2f" % it.reshape(-1, 64).max(1).mean())
d = torch.hypot(st[:, 0] - cx[ci[::8]], st[:, 1] - cy[ci[::8]])
print("   estimate's distance from its course point: mean %.2f max %.2f m; speed mean %.2f" % (d.mean().item(), d.max().item(), st[:, 3].mean().item()))
if os.environ.get("CRX_DUMP"):
    os.makedirs(os.path.join(ROOT, "gpurun_out", "swarm_dump"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "swarm_dump", "mpc_inputs.npz"), st=st.cpu().numpy(), xref=xref.cpu().numpy(),
                        status=status.cpu().numpy(), cost=cost.cpu().numpy(), tind=tind.cpu().numpy())
