#!/usr/bin/env python3
"""Round-4 experiment on the CPU twin of the MPC solver (no GPU): a PORTFOLIO of solver variants.  The BASELINE batch (8,192 agents =
128 waves) leaves 7/8 of the SIMDs idle and the launch lasts as long as its slowest agent's chain of sweeps; different globalisation
settings have different stragglers, so k variants run side by side on idle SIMDs — each agent taking the variant that converges in
the fewest sweeps (ties: lowest variant index: deterministic, twin-reproducible) — would shorten the tail at k times the work.
Variants: number of leading Gauss-Newton sweeps, trust box of a Newton step (rad / m/s^2), warm start.  Printed per seed: every variant's
slowest agent and mean sweeps, then the best 2-, 3- and 4-variant portfolios containing the base variant as
(slowest agent of the portfolio, mean sweeps, mean over waves of the wave maximum, variant indices).
Result (profiles/r04/mpc_experiments.txt): the best 4-variant portfolio brings the slowest agent from 16 / 18 / 19 / 28 (four seeds) to
13 on every seed — at most 1.23x on the BASELINE batch for 4x the work plus cross-wave coordination.  Not built."""
import sys, ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle
from common import mpc_problem
from oracle import oracle_lib
from concurrent.futures import ThreadPoolExecutor
oracle.build(); lib=oracle_lib.lib()
TH=len(os.sched_getaffinity(0))
def run(x0,xref,T,n_gn,ts,ta,warm=0):
    lib.oracle_mpc_warm(warm); lib.oracle_mpc_tune(n_gn, C.c_double(10.0), C.c_double(0.1)); lib.oracle_mpc_trust(C.c_double(ts), C.c_double(ta))
    n=len(x0); cuts=[n*k//TH for k in range(TH+1)]; res=[None]*TH
    def f(k): res[k]=oracle.mpc_solve(x0[cuts[k]:cuts[k+1]], xref[cuts[k]:cuts[k+1]], T)
    with ThreadPoolExecutor(TH) as ex: list(ex.map(f, range(TH)))
    lib.oracle_mpc_warm(0); lib.oracle_mpc_tune(2, C.c_double(10.0), C.c_double(0.1)); lib.oracle_mpc_trust(C.c_double(0.4), C.c_double(0.5))
    return tuple(np.concatenate([r[j] for r in res]) for j in range(3))
variants=[("base gn2 .4/.5",2,0.4,0.5,0),("gn1 .4/.5",1,0.4,0.5,0),("gn3 .4/.5",3,0.4,0.5,0),("gn2 .2/.25",2,0.2,0.25,0),("gn2 .8/1",2,0.8,1.0,0),("gn2 .3/.7",2,0.3,0.7,0),("gn2 .6/.35",2,0.6,0.35,0),("gn2 warm2",2,0.4,0.5,2),("gn0 .4/.5",0,0.4,0.5,0),("gn2 1e9",2,1e9,1e9,0)]
for seed in (4,5,6,7):
    x0,xref=mpc_problem(8192,21,seed)
    its=[]; costs=[]
    for name,n_gn,ts,ta,w in variants:
        sol,st,c=run(x0,xref,21,n_gn,ts,ta,w)
        it=(st>>8).astype(np.int64); it[(st&1)==0]=99
        its.append(it); costs.append(c)
    its=np.array(its)
    print("seed",seed," per-variant max:",[int(i.max()) for i in its]," mean:",[round(float(np.minimum(i,50).mean()),2) for i in its])
    import itertools
    best=[]
    for k in (2,3,4):
        for comb in itertools.combinations(range(len(variants)),k):
            if 0 not in comb: continue
            m=its[list(comb)].min(axis=0)
            best.append((int(m.max()), round(float(m.mean()),2), round(float(m.reshape(-1,64).max(axis=1).mean()),2), comb))
    best.sort()
    for b in best[:6]: print("   ",b)
    # cost differences of winners vs base


# ---- second experiment: when the trust box of a Newton step applies -------------------------------------------------------------
# mode 0 = on every Newton sweep (the engine); 1 = only after the first refused Newton step; 2 = doubled (up to 8x) after every accepted
# full step, back to the base after a refused one.  Result: modes 1 and 2 lower the MEAN (6.8 -> 6.4 / 6.6 sweeps) and lengthen the TAIL
# (slowest agent 16-28 -> 17-50, a few agents no longer converge) — the launch time is the tail.
print("\nseed trust-mode | mean sweeps, slowest agent, mean of wave max, converged, agents ending lower / higher in cost than mode 0")
def run_trust_mode(x0,xref,T,mode):
    lib.oracle_mpc_trust_mode(mode)
    n=len(x0); cuts=[n*k//TH for k in range(TH+1)]; res=[None]*TH
    def f(k): res[k]=oracle.mpc_solve(x0[cuts[k]:cuts[k+1]], xref[cuts[k]:cuts[k+1]], T)
    with ThreadPoolExecutor(TH) as ex: list(ex.map(f, range(TH)))
    lib.oracle_mpc_trust_mode(0)
    return tuple(np.concatenate([r[j] for r in res]) for j in range(3))
for seed in (4,5,6,7,8,9):
    x0,xref=mpc_problem(8192,21,seed)
    base=None
    for mode in (0,1,2):
        sol,st,c=run_trust_mode(x0,xref,21,mode)
        it=(st>>8); conv=(st&1)==1
        if mode==0: base=c
        d=c-base; s_=np.maximum(1,abs(base))
        print(seed,mode,"mean %.2f max %d wavemax %.2f conv %.5f lower %d higher %d"%(it.mean(),it.max(),it.reshape(-1,64).max(axis=1).mean(),conv.mean(),(d<-1e-6*s_).sum(),(d>1e-6*s_).sum()))
