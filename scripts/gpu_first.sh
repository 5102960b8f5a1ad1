set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python - <<'PY' 2>&1 | tee gpurun_out/quick_bench.log
import torch, time, numpy as np, sys
sys.path.insert(0,'tests')
import cpprobotics_amd as crx
from common import *
print(torch.cuda.get_device_name(0))
Q,R=ekf_QR()
for n,T in [(65536,1000),(1048576,100),(4194304,32)]:
    u,x0,P0=ekf_agents(n,1)
    z=torch.randn((T,n,2),device='cuda'); ud=torch.randn((T,n,2),device='cuda')*0.1+1
    x=torch.from_numpy(x0).cuda(); P=torch.from_numpy(P0).cuda()
    xh=torch.empty((T,n,4),device='cuda')
    for hist in (xh,None):
        crx.ekf_run(x.clone(),P.clone(),z,ud,Q,R,x_hist=hist)
        torch.cuda.synchronize()
        xs=[x.clone() for _ in range(5)]; Ps=[P.clone() for _ in range(5)]
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(5): crx.ekf_run(xs[i],Ps[i],z,ud,Q,R,x_hist=hist)
        e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e)/5
        print(f"ekf_run n={n} T={T} hist={'x' if hist is not None else '-'}: {ms:.3f} ms  {n*T/ms/1e6:.2f} G upd/s  {n*T*(32 if hist is not None else 16)/ms/1e6:.1f} GB/s")
    del z,ud,xh
# single step
for n in (65536, 4194304):
    u,x0,P0=ekf_agents(n,1)
    x=torch.from_numpy(x0).cuda(); P=torch.from_numpy(P0).cuda(); z=torch.randn((n,2),device='cuda'); ud=torch.randn((n,2),device='cuda')
    crx.ekf_estimation(x,P,z,ud,Q,R); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(20): crx.ekf_estimation(x,P,z,ud,Q,R)
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/20
    print(f"ekf_step n={n}: {ms*1e3:.1f} us {n/ms/1e6:.2f} G upd/s {n*176/ms/1e6:.1f} GB/s")
for dim in (5,4):
    n=16384
    v=torch.from_numpy(lqr_speeds(n,3)).cuda()
    crx.dlqr_from_v(v,dim=dim); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(10): crx.dlqr_from_v(v,dim=dim)
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/10
    print(f"dare_from_v dim={dim} n={n}: {ms:.3f} ms {n/ms/1e3:.2f} M solves/s")
PY
