import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cpprobotics_amd as crx, oracle
from common import *
from test_oracle_pf import _scenario
t_=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n,T,NP=96,300,100
ut,obs,nobs,nrm,uni,xth,xdh=_scenario(oracle,n,T,NP,5)
px,pw=np.zeros((n,NP,4),np.float32),np.full((n,NP),1.0/NP,np.float32)
_,_,xeo,Peo,xho,nreso=oracle.pf_run(px,pw,obs,nobs,ut,nrm,uni)
xe,Pe,hist,nres=crx.pf_run(t_(px),t_(pw),t_(obs),t_(nobs),t_(ut),t_(nrm),t_(uni))
h=hist.cpu().numpy()
same=np.abs(h-xho).max(axis=(0,2))<1e-3
print("PF episode: same frac", same.mean(), "nres diff max", np.abs(nres.cpu().numpy().astype(int)-nreso).max(), "err", np.hypot(h[...,0]-xth[...,0],h[...,1]-xth[...,1]).mean())
course,goal=mpc_course_f32(); dc=crx.Course.from_numpy(course)
n,Tm,mt=48,6,40
st=tracking_agents(n,tuple(c[:150] for c in course),9,spread=0.5); st[:,3]=np.random.default_rng(10).uniform(0.5,4.0,n).astype(np.float32)
st[0]=(course[0][0],course[1][0],course[2][0],course[4][0])
tind0=oracle.calc_nearest_index(st,course)[0].astype(np.int32)
so,tio,histo,tindo=oracle.mpc_closed_loop(st,course,goal,T=Tm,max_ticks=mt,target_ind=tind0,want_hist=True)
sd,td=t_(st),t_(tind0)
ticks,hist=crx.mpc_simulation(sd,dc,goal,Tm,mt,target_ind=td,want_hist=True)
print("MPC loop err", floored_rel_err(hist.cpu().numpy(),histo,1.0), floored_rel_err(sd.cpu().numpy(),so,1.0))
