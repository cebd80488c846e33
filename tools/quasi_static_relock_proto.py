#!/usr/bin/env python
"""CPU (fp64): a QUASI-STATIC hybrid step for the duty-cycle limit on the oracle's right-hand side, against the converged golden (env 1 of ramp_a4) --
clamped phases advance STUCK (both u frozen, x frozen outward), a free phase that crosses |m| = 1 inside a step is put on the circle at the located
instant (+ first-order current correction), and at the end of each step an axis whose demand has turned inward RE-LOCKS algebraically to the duty cycle
that holds d i_c/dt = d iref_c/dt, the other axis following the circle.  Result: entry 8e-5, stuck stretch 2e-4, median 2-4e-3 --
but the first saturated block stays at 6-7e-2: the true re-lock OVERSHOOTS the quasi-static duty cycle by ~20 % (the released axis is the underdamped
90 kHz current loop, zeta = 0.17, and the anti-windup freezes it at the first extremum of its swing back), which no slow-time-scale rule knows.
    python tools/quasi_static_relock_proto.py 1 2 4 8"""
import sys; sys.path[:0]=['/root/repo','/root/repo/tools','/root/repo/tests']
import numpy as np
from switched_proto import SwRK, run_ramp
from saturated_step_prototypes import RM
from oracle import pvder_model as pm

class QS(RM):
    """STUCK steps + algebraic re-lock at step ends + located entry (quasi-static hybrid)."""
    def __init__(self):
        super().__init__()
        self.clamp=[False]*3
        self.ev={"entry":0,"relock":0,"release":0}
    def mneed(self, y, t, inp, p, f):
        """duty cycle of phase p that holds di_p/dt at d iref_p/dt (complex), from the free field f at y"""
        m=self.m; K=m.Kp_GCC
        o=pm.algebraic(m,y,t,*inp)
        i=o.i[p]; v=o.v[p]
        wr=o.we/pm.WBASE
        # d iref_a/dt
        vg,si,qref,vdcref=inp
        dVdc=f[18]; dxDC=f[19]; dxQ=f[20]
        # dQ/dt = Im d(S_PCC)/dt ; S_PCC = 0.5 sum v_p conj(i_p), v_p = kvg vg_p + kvi i_p
        dS=0
        for q in range(3):
            di=complex(f[6*q],f[6*q+1]); dv=m.kv_i*di
            dS+=0.5*(dv*o.i[q].conjugate()+o.v[q]*di.conjugate())
        dQ=dS.imag
        dia=complex(dxDC-m.Kp_DC*dVdc, dxQ+m.Kp_Q*dQ)
        rot=[1.0,pm.A240,pm.A120][p]
        diref=dia*rot
        mn=(2.0/o.Vdc)*(v+m.Rf*i+m.Lf*(diref+1j*wr*i))
        return mn
    def step_sw(self,y,t,h,inp):
        K=self.m.Kp_GCC
        modes=tuple(3 if c else 0 for c in self.clamp)
        y1=self.etd(y,t,h,inp,modes,tuple(self.clamp))
        # x of clamped phases: inward motion only
        for p in range(3):
            if self.clamp[p]:
                for c in range(2):
                    u=y[6*p+4+c]; x=y[6*p+2+c]
                    if u*x<0: y1[6*p+2+c]=x+h*self.m.Ki_GCC*u
        # (a) entry of free phases
        for p in range(3):
            if self.clamp[p]: continue
            m0=complex(y[6*p+2]+K*y[6*p+4],y[6*p+3]+K*y[6*p+5]); m1=complex(y1[6*p+2]+K*y1[6*p+4],y1[6*p+3]+K*y1[6*p+5])
            if abs(m1)<=1.0: continue
            s0=abs(m0)**2-1; s1=abs(m1)**2-1
            th=min(1.0,max(0.0,s0/(s0-s1))) if s0<0 else 0.0
            xu0=y[6*p+2:6*p+6]; xu1=y1[6*p+2:6*p+6]
            xu=xu0+th*(xu1-xu0)
            mth=complex(xu[0]+K*xu[2],xu[1]+K*xu[3])
            # radial polish through u
            sc=1.0/abs(mth); mt=mth*sc
            xu[2]+= (mt.real-mth.real)/K; xu[3]+=(mt.imag-mth.imag)/K
            y1[6*p+2:6*p+6]=xu
            # current correction: the overshoot (m_free - m_clamped) acted over (1-th)h, growing linearly
            o=pm.algebraic(self.m,y1,t+h,*inp)
            dm=(m1-mt)
            y1[6*p]  -= (o.Vdc/(2*self.m.Lf))*dm.real*(1-th)*h*0.5
            y1[6*p+1]-= (o.Vdc/(2*self.m.Lf))*dm.imag*(1-th)*h*0.5
            self.clamp[p]=True; self.ev["entry"]+=1
        # (b) re-lock of clamped phases
        f=self.f_free(y1,t+h,inp)
        for p in range(3):
            if not self.clamp[p]: continue
            u=[y1[6*p+4],y1[6*p+5]]; x=[y1[6*p+2],y1[6*p+3]]
            du=[f[6*p+4],f[6*p+5]]
            inward=[(du[c]*u[c]<0) for c in range(2)]
            if not any(inward): continue
            mn=self.mneed(y1,t+h,inp,p,f)
            mnc=[mn.real,mn.imag]
            newu=list(u)
            for c in range(2):
                if inward[c]:
                    cand=(mnc[c]-x[c])/K
                    # an inward move only (towards zero, not through it)
                    if abs(cand)<abs(u[c]) and cand*u[c]>=0: newu[c]=cand
                    elif cand*u[c]<0: newu[c]=0.0
            mm=complex(x[0]+K*newu[0],x[1]+K*newu[1])
            if all(inward):
                if abs(mm)<1.0:
                    self.clamp[p]=False; self.ev["release"]+=1
                    y1[6*p+4],y1[6*p+5]=newu
                    continue
            # one axis regulated: the other follows the circle (outward move of its u only)
            for c in range(2):
                o_=1-c
                if inward[c] and not inward[o_]:
                    mc=x[c]+K*newu[c]
                    if abs(mc)<1.0:
                        mo=np.sqrt(1.0-mc*mc)*np.sign(x[o_]+K*u[o_])
                        cand=(mo-x[o_])/K
                        if abs(cand)>=abs(u[o_]) and cand*u[o_]>=0: newu[o_]=cand
            if newu!=list(u): self.ev["relock"]+=1
            y1[6*p+4],y1[6*p+5]=newu
        return y1
if __name__=="__main__":
    for quiet in [int(a) for a in sys.argv[1:]]:
        R=QS()
        err=run_ramp(R,env=1,steps=135,quiet=quiet)
        print("qs quiet",quiet,"blocks from 111:"," ".join("%.1e"%err[k:k+5].max() for k in range(111,135,5)), "median %.1e"%np.median(err[116:135]),R.ev,flush=True)
