cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lqr_gpu.py tests/test_golden_gpu.py tests/test_track_gpu.py tests/test_dropin_cpp.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
timeout 600 python - <<'PY' 2>/dev/null
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
print("dare5 %.4f ms  dare4 %.4f ms"%(gt(lambda: crx.dlqr_from_v(v,dim=5),30), gt(lambda: crx.dlqr_from_v(v,dim=4),30)))
course,goal=lqr_course(); dc=crx.Course.from_numpy(course)
st=torch.from_numpy(tracking_agents(16384,tuple(c[:200] for c in course),5,spread=0.4)).cuda()
print("loop5 %.3f ms  loop4 %.3f ms"%(gt(lambda: crx.closed_loop_prediction(st.clone(),dc,goal,dim=5,max_ticks=400),3), gt(lambda: crx.closed_loop_prediction(st.clone(),dc,goal,dim=4,max_ticks=400),3)))
PY
