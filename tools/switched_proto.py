#!/usr/bin/env python
"""fp64 prototype (CPU) of the switched-operator ETDRK4 step of the 23-state tail (csrc etd_switched_step) on the oracle's own
right-hand side, along the Vdc_ref ramp into the duty-cycle limit (tests/golden/bridge_ramp_a4.npz): per-step error against the
converged bridge before, inside and behind the saturated stretches.     python tools/switched_proto.py"""
import sys, dataclasses, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from scipy.linalg import expm
from oracle import pvder_model as pm
from oracle.exp_rk_proto import ExpRK, DT, NS
import bridge_cases

class SwRK(ExpRK):
    def __init__(self):
        super().__init__("50")
        self.m_free = dataclasses.replace(self.m, m_limit=1e9)
        self.Lblk = self.L[np.ix_([0, 1, 4, 5], [0, 1, 4, 5])].copy()      # (iR, iI, uR, uI) of one phase
        self._c2 = {}
        self.proj = True
        self.stats = dict(sw=0, proj=0)

    def f_free(self, y, t, inp):
        return np.array(pm.rhs(self.m_free, y, t, *inp))

    def modes_of(self, y, t, inp):
        f = self.f_free(y, t, inp)
        modes = []
        for p in range(3):
            x = complex(y[6*p+2], y[6*p+3]); u = complex(y[6*p+4], y[6*p+5])
            mm = self.m.Kp_GCC * u + x
            md = 0
            if abs(mm) > self.m.m_limit:
                if f[6*p+4] * u.real > 0: md |= 1
                if f[6*p+5] * u.imag > 0: md |= 2
            modes.append(md)
        return tuple(modes)

    def L_of(self, modes):
        L = np.zeros((NS, NS))
        for p, md in enumerate(modes):
            B = self.Lblk.copy()
            if md & 1: B[2] = 0
            if md & 2: B[3] = 0
            idx = [6*p, 6*p+1, 6*p+4, 6*p+5]
            L[np.ix_(idx, idx)] = B
        return L

    def phis_m(self, h, modes):
        key = (h, modes)
        if key not in self._c2:
            L = self.L_of(modes)
            out = [expm(h * L)]
            for k in (1, 2, 3):
                M = np.zeros((NS * (k + 1), NS * (k + 1)))
                M[:NS, :NS] = h * L
                for j in range(k):
                    M[j*NS:(j+1)*NS, (j+1)*NS:(j+2)*NS] = np.eye(NS)
                out.append(expm(M)[:NS, k*NS:(k+1)*NS])
            self._c2[key] = (L, out)
        return self._c2[key]

    def f_masked(self, y, t, inp, modes):
        f = self.f_free(y, t, inp)
        for p, md in enumerate(modes):
            x = complex(y[6*p+2], y[6*p+3]); u = complex(y[6*p+4], y[6*p+5])
            if abs(self.m.Kp_GCC * u + x) > self.m.m_limit:
                if f[6*p+2] * x.real > 0: f[6*p+2] = 0
                if f[6*p+3] * x.imag > 0: f[6*p+3] = 0
            if md & 1: f[6*p+4] = 0
            if md & 2: f[6*p+5] = 0
        return f

    def step_sw(self, y, t, h, inp):
        modes = self.modes_of(y, t, inp)
        if any(modes): self.stats["sw"] += 1
        L, (f0, f1, f2, f3) = self.phis_m(h, modes)
        _, (g0, g1, g2, g3) = self.phis_m(h / 2, modes)
        N = lambda yy, tt: self.f_masked(yy, tt, inp, modes) - L @ yy
        N1 = N(y, t)
        N2 = N(g0 @ y + h * (0.5 * g1) @ N1, t + h / 2)
        N3 = N(g0 @ y + h * ((0.5 * g1 - g2) @ N1 + g2 @ N2), t + h / 2)
        N4 = N(f0 @ y + h * ((f1 - 2 * f2) @ N1 + f2 @ (N2 + N3)), t + h)
        a52 = 0.5 * g2 - f3 + 0.25 * f2 - 0.5 * g3
        a54 = 0.25 * g2 - a52
        a51 = 0.5 * g1 - 2 * a52 - a54
        N5 = N(g0 @ y + h * (a51 @ N1 + a52 @ (N2 + N3) + a54 @ N4), t + h / 2)
        yn = f0 @ y + h * ((f1 - 3 * f2 + 4 * f3) @ N1 + (-f2 + 4 * f3) @ N4 + (4 * f2 - 8 * f3) @ N5)
        if self.proj:
            lim = self.m.m_limit ** 2 * 1.0004
            for p in range(3):
                x1 = complex(yn[6*p+2], yn[6*p+3]); u0 = complex(y[6*p+4], y[6*p+5]); u1 = complex(yn[6*p+4], yn[6*p+5])
                m1 = self.m.Kp_GCC * u1 + x1
                if abs(m1) ** 2 > lim:
                    du = u1 - u0
                    dout = complex(du.real if du.real * u1.real > 0 else 0.0, du.imag if du.imag * u1.imag > 0 else 0.0)
                    d = self.m.Kp_GCC * dout; mb = m1 - d
                    qa = abs(d) ** 2; qb = 2 * (mb.real * d.real + mb.imag * d.imag); qc = abs(mb) ** 2 - lim
                    beta = 0.0
                    disc = qb * qb - 4 * qa * qc
                    if qc < 0 and qa > 0 and disc >= 0:
                        beta = min(1.0, max(0.0, (-qb + np.sqrt(disc)) / (2 * qa)))
                    u1 = (u1 - dout) + beta * dout
                    yn[6*p+4], yn[6*p+5] = u1.real, u1.imag
                    self.stats["proj"] += 1
        return yn

def run_ramp(R, name="ramp_a4", env=0, steps=160, quiet=2, start=0, log=None):
    g = bridge_cases.load(name)
    vgt, sit = g["vgrid"][:, env], g["sinsol"][:, env]
    acts = g["actions"][:, env]
    n = 15; K = len(vgt) - 1; s0 = ds = 60
    y, q, v = R.y0.copy(), 0.0, R.m.Vdc_ref0
    dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
    ksub, prev_row = 0, 0
    out = []
    for k in range(steps):
        a = int(acts[k])
        q += dq * (int(a == 1) - int(a == 2)); v += dv * (int(a == 3) - int(a == 4))
        for s in range(n):
            gk = ksub + s
            row = 0 if gk < s0 else min((gk - s0) // ds + 1, K)
            fracs = [1 / 32, 31 / 32] if (s == 0 or row != prev_row) else [1.0 / quiet] * quiet
            prev_row = row
            t = gk * DT
            for fr in fracs:
                y = R.step_sw(y, t, DT * fr, (vgt[row], sit[row], q, v)); t += DT * fr
        ksub += n
        row = 0 if ksub < s0 else min((ksub - s0) // ds + 1, K)
        o = R.observe(y, ksub * DT, vgt[row], sit[row], q, v, ksub)[:10]
        w = g["obs"][k, env, :10]
        out.append(np.linalg.norm(o - w) / np.linalg.norm(w))
        if log and k >= log:
            mm = [abs(R.m.Kp_GCC * complex(y[6*p+4], y[6*p+5]) + complex(y[6*p+2], y[6*p+3])) for p in range(3)]
            print("step %3d err %.1e |m| %s  bridge |m_a| %.4f  obs P,Q,Vdc %s want %s" % (k, out[-1], " ".join("%.4f" % x for x in mm), g["m_abs"][k, env], np.round(o[[4, 5, 6]], 4), np.round(w[[4, 5, 6]], 4)), flush=True)
    return np.array(out)

if __name__ == "__main__":
    g = bridge_cases.load("ramp_a4")
    first = [int(np.argmax(g["m_abs"][:, e] > 0.9995)) for e in range(4)]
    print("first saturated env step per env:", first)
    e = int(np.argmin(first))
    for proj in (False, True):
        R = SwRK(); R.proj = proj
        err = run_ramp(R, env=e, steps=160)
        print("projection %-5s env %d, switched steps %d: worst env step by 5-step block from %d: " % (proj, e, R.stats["sw"], first[e] - 5)
              + " ".join("%.1e" % err[k:k + 5].max() for k in range(first[e] - 5, 160, 5)), flush=True)
