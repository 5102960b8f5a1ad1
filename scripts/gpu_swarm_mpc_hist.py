"""Sweep-count histogram of the MPC solves of one mixed-swarm round (the planner inputs scripts/swarm_bench.py forms), and the
inputs of the slowest agents saved for CPU analysis (gpurun_out/swarm_mpc_worst.npz)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx
from common import ekf_QR, mpc_course_f32
n, T, Tm = 131072, 100, 21
dev = torch.device("cuda", 0)
Q, R = ekf_QR()
course, goal = mpc_course_f32()
dc = crx.Course.from_numpy(course, device=dev)
ci = torch.from_numpy(np.random.default_rng(99).integers(0, len(course[0]) - 30, n)).to(dev)
cx, cy, cyaw = (torch.from_numpy(a).to(dev) for a in course[:3])
x0 = torch.stack([cx[ci], cy[ci], cyaw[ci], torch.full((n,), 2.5, device=dev)], dim=1).contiguous()
u_true = torch.zeros((n, 2), device=dev)
w = crx.normal_draws(n, T, agent0=0, seed=99, device=dev)
z, ud = crx.ekf_simulate_inputs(u_true, x0.clone(), x0.clone(), w)
P = torch.eye(4, device=dev).reshape(1, 16).repeat(n, 1).contiguous()
x = x0.clone()
crx.ekf_run(x, P, z, ud, Q, R)
est = x[::8].contiguous()
est[:, 3] = 2.5
tind = torch.zeros(est.shape[0], dtype=torch.int32, device=dev)
crx.calc_nearest_index(est, dc, tind)
xref = crx.calc_ref_trajectory(est, dc, tind, Tm)
sol, st, cost = crx.mpc_solve(est, xref, Tm, return_status=True)
torch.cuda.synchronize()
st = st.cpu().numpy(); sw = st >> 8
print("agents", len(sw), "sweeps mean %.2f max %d" % (sw.mean(), sw.max()), "not converged", int((st & 1).sum()), "flags", np.bincount(st & 0xff)[:8])
print("hist", np.bincount(sw))
worst = np.argsort(-sw)[:64]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "swarm_mpc_worst.npz"), x0=est.cpu().numpy()[worst], xref=xref.cpu().numpy()[worst], sweeps=sw[worst],
         status=st[worst], cost=cost.cpu().numpy()[worst], sol=sol.cpu().numpy()[worst], est_true_err=(est.cpu().numpy() - x0[::8].cpu().numpy())[worst])
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "swarm_mpc_all.npz"), x0=est.cpu().numpy(), xref=xref.cpu().numpy(), status=st, cost=cost.cpu().numpy())
e = (est - x0[::8]).cpu().numpy()
print("estimate minus start: |dx| mean %.3f max %.3f, |dyaw| mean %.3f max %.3f" % (np.abs(e[:, :2]).mean(), np.abs(e[:, :2]).max(), np.abs(e[:, 2]).mean(), np.abs(e[:, 2]).max()))
