cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, sys, numpy as np
sys.path.insert(0,'tests')
import cpprobotics_amd as crx
from common import *
Q,R=ekf_QR()
def run(n,T,hist,reps=10):
    u,x0,P0=ekf_agents(n,1)
    z=torch.randn((T,n,2),device='cuda')*0.3; ud=torch.randn((T,n,2),device='cuda')*0.1+1
    x=torch.from_numpy(x0).cuda(); P=torch.from_numpy(P0).cuda()
    xh=torch.empty((T,n,4),device='cuda') if hist else None
    xs=[x.clone() for _ in range(reps+1)]; Ps=[P.clone() for _ in range(reps+1)]
    crx.ekf_run(xs[-1],Ps[-1],z,ud,Q,R,x_hist=xh); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): crx.ekf_run(xs[i],Ps[i],z,ud,Q,R,x_hist=xh)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/reps
    B=32 if hist else 16
    print(f"n={n:8d} T={T:5d} hist={int(hist)}: {ms:.3f} ms {n*T/ms/1e6:7.2f} G upd/s {n*T*B/ms/1e6:7.1f} GB/s  waves/SIMD={n/65536:.1f}")
for n,T in [(65536,1000),(131072,500),(196608,400),(262144,250),(524288,125),(1048576,100),(4194304,32)]:
    run(n,T,True); 
run(65536,1000,False); run(1048576,100,False)
PY
