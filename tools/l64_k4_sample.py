#!/usr/bin/env python
"""CPU (numpy fp64): step doubling with ETD3RK vs Krogstad ETDRK4 steps behind the tick on the 48-env random-action sample episode (tests/bridge_cases.py
episode_random_sample): histogram of the accepted levels, cost in right-hand sides per env step (per env / worst env of the batch: what a launch waits for),
error against the bridge golden.    python tools/l64_k4_sample.py [steps]"""
import sys,os; sys.path[:0]=['/root/repo','/root/repo/gym-solarpvder-environment_amd','/root/repo/tests']
import numpy as np
from gym_PVDER_amd import der_model, etd_linear as el
import bridge_cases
par=der_model.param_vector("50"); c=el.folded(par); L,_=el.jacobian(par,"50")
name="episode_random_sample"; g=bridge_cases.load(name); vg,si,acts,cc=bridge_cases.case_inputs(name)
N=vg.shape[1]; n=15; steps=int(sys.argv[1]) if len(sys.argv)>1 else 60
y0f=der_model.steady_state(par,"50"); y0=np.concatenate([y0f[:6],y0f[18:]]).astype(np.float64)
dq,dv=25.0*n/50e3,0.02*n/500.0
rowof=lambda gk: 0 if gk<60 else min((gk-60)//60+1,vg.shape[0]-1)
cache={}
def m3(h):
    k=('3',round(h*1e13))
    if k not in cache: cache[k]=el.etd3rk_f_matrices(L,h)
    return cache[k]
def m4(h):
    k=('4',round(h*1e13))
    if k not in cache: cache[k]=el.k4_f_matrices(L,h)
    return cache[k]
def observe(y,vgv,siv,q,v):
    i=y[0]+1j*y[1]; va=complex(c["GR"],c["GI"])*vgv+complex(c["kviR"],c["kviI"])*i
    S=0.5*c["PH"]*va*np.conj(i)
    ppv=np.maximum(0.0,y[6]*(c["pvA"]*siv+c["pvB"]-c["pvB"]*np.exp(c["pvK"]*y[6])))
    return np.stack([i.real,i.imag,va.real,va.imag,S.real,S.imag,y[6],ppv,v,q],axis=-1)
rtol,atol=1e-6,1e-8
def run(scheme,accept,maxlvl=8):
    stepf,mats=(el.etd3rk_step,m3) if scheme=="etd3rk" else (el.k4_step,m4)
    y=np.tile(y0[:,None],(1,N)); q=np.zeros(N); v=np.full(N,par[der_model.IDX["VDC_REF0"]])
    ksub=0; out=[]; hist=np.zeros(maxlvl+1,int); worst=[]
    h0=el.tick_length(32)/(60.0*32); Tr=n/60.0-h0
    for k in range(steps):
        a=acts[k]; q=q+dq*((a==1).astype(float)-(a==2)); v=v+dv*((a==3).astype(float)-(a==4))
        row=rowof(ksub); inp=(vg[row],si[row],q,v)
        yt,_=el.etd3rk_step(c,m3(h0),y,inp)
        prev=None; acc=np.zeros(N,bool); res=np.zeros_like(y); lv=np.full(N,maxlvl)
        for j in range(maxlvl+1):
            K=2**(j+1); M=mats(Tr/K); u=yt
            for _ in range(K): u,_=stepf(c,M,u,inp)
            if prev is not None:
                sc=accept*(atol+rtol*np.maximum(np.abs(u),np.abs(prev)))
                ok=np.sqrt(np.mean(((u-prev)/sc)**2,axis=0))<=1.0
                new=ok&~acc; res[:,new]=u[:,new]; lv[new]=j; acc|=ok
            prev=u
            if acc.all(): break
        res[:,~acc]=prev[:,~acc]
        for l in lv: hist[l]+=1
        worst.append(lv.max())
        y=res; ksub+=n
        row2=rowof(ksub); out.append(observe(y,vg[row2],si[row2],q,v))
    got=np.array(out); want=g["obs"][:steps,:,:10]
    err=np.linalg.norm(got-want,axis=2)/np.linalg.norm(want,axis=2)
    traj=np.linalg.norm((got-want).transpose(1,0,2).reshape(N,-1),axis=1)/np.linalg.norm(want.transpose(1,0,2).reshape(N,-1),axis=1)
    return hist,np.array(worst),err.max(),traj.max()
for scheme,accept in (("etd3rk",1.0),("etd3rk",2.0),("k4",1.0),("k4",2.0),("k4",4.0)):
    hist,worst,e,tr=run(scheme,accept)
    cost=lambda l: sum(2**(j+1) for j in range(l+1))*(3 if scheme=="etd3rk" else 4)
    print("%-7s accept %g: levels histogram %s ; batch-worst level per step: mean cost %.0f RHS (max-over-envs), per-env mean cost %.0f RHS ; worst env step %.2e, worst trajectory %.2e"%(
        scheme,accept,hist.tolist(),np.mean([cost(l) for l in worst]), sum(hist[l]*cost(l) for l in range(len(hist)))/hist.sum(), e,tr),flush=True)
