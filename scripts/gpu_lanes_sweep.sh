# Sweep of active lanes per wave for the latency-bound kernels (CRX_LANES override in crx_api.hip).
cd $GRAFT_REPO_ROOT
for L in 64 32 16 8 4 0; do
  echo "== CRX_LANES=$L (0 = automatic)"
  CRX_LANES=$L timeout 600 python - <<'PY' 2>/dev/null
import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cpprobotics_amd as crx
from common import *
def gt(fn, reps):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
v=torch.from_numpy(lqr_speeds(16384,3)).cuda()
print("  dare5 %.4f ms  dare4 %.4f ms"%(gt(lambda: crx.dlqr_from_v(v,dim=5),20), gt(lambda: crx.dlqr_from_v(v,dim=4),20)), end="")
x0,xr=mpc_problem(8192,21,4); x0,xr=torch.from_numpy(x0).cuda(),torch.from_numpy(xr).cuda()
print("  mpc21 %.3f ms"%gt(lambda: crx.mpc_solve(x0,xr,21),3), end="")
x0,xr=mpc_problem(8192,6,4); x0,xr=torch.from_numpy(x0).cuda(),torch.from_numpy(xr).cuda()
print("  mpc6 %.3f ms"%gt(lambda: crx.mpc_solve(x0,xr,6),5), end="")
course,goal=lqr_course(); dc=crx.Course.from_numpy(course)
st=torch.from_numpy(tracking_agents(16384,tuple(c[:200] for c in course),5,spread=0.4)).cuda()
print("  loop5 %.3f ms  loop4 %.3f ms"%(gt(lambda: crx.closed_loop_prediction(st.clone(),dc,goal,dim=5,max_ticks=400),2), gt(lambda: crx.closed_loop_prediction(st.clone(),dc,goal,dim=4,max_ticks=400),2)))
PY
done
