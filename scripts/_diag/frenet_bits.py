import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cpprobotics_amd as crx, oracle
from test_frenet_gpu import _states, _course, _cfg
t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
course, ob = _course(crx, oracle)
tot=0; bad=0; badv=0; badbest=0
for seed in range(5):
    st=_states(2000, 100+seed)
    o=oracle.frenet_plan(st, course.coef, ob)
    r=crx.frenet_optimal_planning(t(st), course, t(ob), _cfg(crx), want_paths=True)
    cf, ok = r["path_cf"].cpu().numpy(), r["path_ok"].cpu().numpy()
    m = ~((cf==o["path_cf"]) | (np.isnan(cf)&np.isnan(o["path_cf"])))
    tot+=cf.size; bad+=int(m.sum()); badv+=int((ok!=o["path_ok"]).sum()); badbest+=int((r["best_idx"].cpu().numpy()!=o["best"]).sum())
    if m.any():
        i=np.argwhere(m)[0]; print('e.g.', cf[tuple(i)], o["path_cf"][tuple(i)])
print(f"candidate costs: {bad} of {tot} differ in bits; verdicts {badv}; winners {badbest} of {5*2000}")
