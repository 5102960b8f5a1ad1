import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import cpprobotics_amd as crx
from common import mpc_problem
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out={"lib": os.path.basename(os.environ.get("CRX_LIB_PATH","libcrx.so"))}
for n,T in ((8192,21),(8192,6),(65536,21)):
    x0,xref=mpc_problem(n,T,4); x0,xref=torch.from_numpy(x0).cuda(),torch.from_numpy(xref).cuda()
    out[f"{n}_T{T}"]=round(timeit(lambda: crx.mpc_solve(x0,xref,T,return_status=True)),4)
print(json.dumps(out))
