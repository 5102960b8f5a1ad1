cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mpc_gpu.py tests/test_golden_gpu.py tests/test_track_gpu.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
timeout 600 python - <<'PY' 2>/dev/null
import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cpprobotics_amd as crx, oracle
from common import *
def gt(fn, reps):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
for T in (21,6):
    x0,xr=mpc_problem(8192,T,4); x0d,xrd=torch.from_numpy(x0).cuda(),torch.from_numpy(xr).cuda()
    ms=gt(lambda: crx.mpc_solve(x0d,xrd,T),5)
    sol,st,cost=crx.mpc_solve(x0d,xrd,T,return_status=True)
    so,sto,co=oracle.mpc_solve(x0[:1024],xr[:1024],T)
    st=st.cpu().numpy()[:1024]; both=((st&1)==1)&((sto&1)==1)
    err=np.max(np.abs(sol.cpu().numpy()[:1024][both]-so[both])/np.maximum(np.abs(so[both]),1.0))
    print(f"T={T}: {ms:.3f} ms  {8192/ms*1e3/1e6:.3f} M solves/s  err vs twin {err:.2e}  iters equal {np.mean((st>>8)==(sto>>8)):.4f} both_conv {both.mean():.4f}")
PY
