set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mpc_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_mpc.log
cat gpurun_out/pytest_mpc.log
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/mpc_bench.log
import torch, sys, numpy as np
sys.path.insert(0,'tests')
import cpprobotics_amd as crx
from common import *
for n,T in [(8192,21),(65536,21),(8192,6)]:
    x0,xref=mpc_problem(n,T,4)
    x0=torch.from_numpy(x0).cuda(); xref=torch.from_numpy(xref).cuda()
    sol,st,c=crx.mpc_solve(x0,xref,T,return_status=True); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(3): crx.mpc_solve(x0,xref,T)
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/3
    it=(st>>8).cpu().numpy()
    print(f"mpc n={n} T={T}: {ms:.3f} ms  {n/ms/1e3:.3f} M solves/s  conv {(st&1).float().mean().item():.4f} mean_it {it.mean():.2f} max_it {it.max()}")
PY
