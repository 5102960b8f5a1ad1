#!/usr/bin/env python3
"""Round-4 experiment on the CPU twin (no GPU): predicted launch cost of the quad-lane MPC portfolio for every 4-subset of twelve solver
variants that contains the engine's own.  Model: a wave-sweep costs one backward sweep (76 k ticks) + 26 k ticks per candidate rollout,
the rollouts of a sweep being the maximum over the wave's live lanes (scripts/gpu_mpc_ticks.py); a single-variant wave holds 64 agents, a
portfolio wave 16 agents x 4 variants and an agent's lanes stop when its first variant converges; the launch is the dearest wave.
Printed: the twelve best sets as (predicted cost relative to the single-variant launch, slowest agent) on three problem draws, and the
set the kernel uses.  The model leaves out the 4x waves' shared memory paths (~9 %): the kernel's set is predicted at 0.867 on the BASELINE
draw and measures 0.947 (profiles/r04/mpc_portfolio_ab.jsonl); the best set found would be 0.82 -> ~0.89, i.e. the portfolio's ceiling
is ~1.12x on that draw."""
import sys, ctypes as C, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle
from common import mpc_problem
from oracle import oracle_lib
from concurrent.futures import ThreadPoolExecutor
oracle.build(); lib=oracle_lib.lib()
pp=oracle_lib._mpc_params(None)
p=lambda a:a.ctypes.data_as(C.c_void_p)
variants=[("base",2,1.0),("gn3",3,1.0),("gn1",1,1.0),("t2",2,2.0),("gn1 t2",1,2.0),("gn3 t2",3,2.0),("t1.5",2,1.5),("gn1 t1.5",1,1.5),("gn3 t1.5",3,1.5),("t3",2,3.0),("gn1 t3",1,3.0),("gn3 t3",3,3.0)]
B,R=76.0,26.0   # k-ticks per backward sweep / per candidate rollout of a wave (scripts/gpu_mpc_ticks.py)
def profile(x0,xref,T,n_gn,ts):
    n=len(x0); its=np.zeros(n,np.int64); conv=np.zeros(n,bool); LS=np.zeros((n,50),np.int8)
    def work(k0,k1):
        ls=(C.c_int*50)()
        for a in range(k0,k1):
            xa=np.ascontiguousarray(x0[a]); ra=np.ascontiguousarray(xref[a])
            st=lib.oracle_mpc_ls_profile(C.c_int(T),p(xa),p(ra),p(pp),C.c_int(50),C.c_int(n_gn),C.c_double(ts),ls)
            its[a]=st>>8; conv[a]=st&1; LS[a]=np.frombuffer(ls,dtype=np.int32)
    TH=len(os.sched_getaffinity(0)); cuts=[n*k//TH for k in range(TH+1)]
    with ThreadPoolExecutor(TH) as ex: list(ex.map(lambda k: work(cuts[k],cuts[k+1]), range(TH)))
    return its,conv,LS
def single_cost(its,conv,LS):
    # waves of 64 agents, one variant: a wave runs sweeps until its slowest lane; per sweep backward + max rollouts over live lanes
    n=len(its); worst=0
    for w in range(0,n,64):
        it=its[w:w+64]; ls=LS[w:w+64].astype(int)
        S=int(it.max())+1          # sweeps 0..it (the last one only checks)
        c=0.0
        for s in range(min(S,50)):
            live=it>=s
            c+=B+R*int(ls[live,s].max()) if live.any() else 0
        worst=max(worst,c)
    return worst
def portfolio_cost(prof, comb):
    its=np.array([prof[v][0] for v in comb]); conv=np.array([prof[v][1] for v in comb]); LS=np.array([prof[v][2] for v in comb]).astype(int)
    it_eff=np.where(conv,its,99)
    stop=it_eff.min(axis=0)                      # sweep index at which the agent's first variant converges
    n=its.shape[1]; worst=0; 
    for w in range(0,n,16):
        st=np.minimum(stop[w:w+16],50)
        S=int(st.max())+1
        c=0.0
        for s in range(min(S,50)):
            live=st>=s                           # agents still running at sweep s (all four lanes run)
            if not live.any(): break
            m=0
            for v in range(len(comb)):
                # a lane that itself converged at sweep s does no rollout in s (ls recorded 0 there)
                m=max(m,int(LS[v,w:w+16][live,s].max()))
            c+=B+R*m
        worst=max(worst,c)
    return worst, int(np.minimum(stop,50).max())
res={}
for seed in (4,5,7):
    x0,xref=mpc_problem(8192,21,seed)
    prof=[profile(x0,xref,21,g,t) for _,g,t in variants]
    sc=single_cost(*prof[0])
    print("seed",seed,"single-variant predicted k-ticks",sc,"max sweeps",int(prof[0][0].max()))
    for comb in itertools.combinations(range(len(variants)),4):
        if 0 not in comb: continue
        c,mx=portfolio_cost(prof,comb)
        res.setdefault(comb,[]).append((c/sc,mx))
rank=sorted(res.items(), key=lambda kv: max(r for r,_ in kv[1]))
for comb,v in rank[:12]: print([variants[i][0] for i in comb], [(round(r,3),m) for r,m in v])
cur=(0,1,3,4)
print("current set",[variants[i][0] for i in cur], [(round(r,3),m) for r,m in res[cur]])
