import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cpprobotics_amd as crx, oracle
from test_oracle_pf import _scenario
t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for NP in (100, 64, 128, 37):
    n,T=200,150
    ut,obs,nobs,nrm,uni,xth,xdh=_scenario(oracle,n,T,NP,5+NP)
    px,pw=np.zeros((n,NP,4),np.float32),np.full((n,NP),1.0/NP,np.float32)
    pxo,pwo,xeo,Peo,xho,nro=oracle.pf_run(px,pw,obs,nobs,ut,nrm,uni,wave_order=True)
    pxd,pwd=t(px),t(pw)
    xe,Pe,hist,nres=crx.pf_run(pxd,pwd,t(obs),t(nobs),t(ut),t(nrm),t(uni))
    h=hist.cpu().numpy()
    same_v=(h==xho).all(axis=(0,2))
    print('NP',NP,'vehicles bit-equal over the whole episode:',same_v.sum(),'of',n,'| nres equal',(nres.cpu().numpy()==nro).all(),'| px',np.array_equal(pxd.cpu().numpy(),pxo),'pw',np.array_equal(pwd.cpu().numpy(),pwo),'P',np.array_equal(Pe.cpu().numpy(),Peo))
    if not same_v.all():
        a=int(np.flatnonzero(~same_v)[0]); tt=int(np.flatnonzero((h[:,a]!=xho[:,a]).any(1))[0]); print('  first diff vehicle',a,'tick',tt,h[tt,a],xho[tt,a])
