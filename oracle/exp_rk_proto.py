"""CPU ORACLE (test infrastructure only) -- fp64 prototype of the exponential Runge-Kutta schemes the
HIP kernels implement, on the oracle's own right-hand side (oracle/pvder_model.py::rhs).

Used by tests/ to pin the ALGORITHMIC claims of DESIGN.md section 5 on the CPU: (1) Krogstad's 4-stage
ETDRK4 loses order on the slaved fast block (its error is set by the coarsest step), Hochbruck &
Ostermann's 5-stage scheme of stiff order 4 (SIAM J. Numer. Anal. 43, 2005, eq. 5.19) does not;
(2) with that scheme the error of a sim time step that starts at an input jump is set by the size of
the FIRST step behind the jump.  The linear part is the 4x4 real block (i_p, u_p) of the Jacobian at the
steady state, per phase (the device uses the equivalent constant complex 2x2 block Lz).

The product path never imports this file.  PARITY UNPINNED (see oracle/pvder_model.py).
"""
from __future__ import annotations

import os

import numpy as np
from scipy.linalg import expm

from . import pvder_model as pm

DT = 1.0 / 60.0
NS = 23


class ExpRK:
    def __init__(self, der_id: str = "50"):
        self.m = pm.build_model(der_id)
        self.y0 = np.array(pm.steady_state(self.m), dtype=np.float64)
        J = np.zeros((NS, NS))
        for k in range(NS):
            d = np.zeros(NS); d[k] = 1e-7
            J[:, k] = (self.f(self.y0 + d, 0.0, (1.0, 100.0, 0.0, self.m.Vdc_ref0))
                       - self.f(self.y0 - d, 0.0, (1.0, 100.0, 0.0, self.m.Vdc_ref0))) / 2e-7
        self.L = np.zeros((NS, NS))
        for p in range(3):
            idx = [6 * p, 6 * p + 1, 6 * p + 4, 6 * p + 5]
            self.L[np.ix_(idx, idx)] = J[np.ix_(idx, idx)]
        self._cache = {}

    def f(self, y, t, inp):
        return np.array(pm.rhs(self.m, y, t, *inp))

    def phis(self, h):
        """phi_0..phi_3 of h*L (augmented-matrix exponentials)"""
        if h not in self._cache:
            out = [expm(h * self.L)]
            for k in (1, 2, 3):
                M = np.zeros((NS * (k + 1), NS * (k + 1)))
                M[:NS, :NS] = h * self.L
                for j in range(k):
                    M[j * NS:(j + 1) * NS, (j + 1) * NS:(j + 2) * NS] = np.eye(NS)
                out.append(expm(M)[:NS, k * NS:(k + 1) * NS])
            self._cache[h] = out
        return self._cache[h]

    def step_krogstad(self, y, t, h, inp):
        f0, f1, f2, f3 = self.phis(h)
        g0, g1, g2, _ = self.phis(h / 2)
        N = lambda yy, tt: self.f(yy, tt, inp) - self.L @ yy
        N1 = N(y, t)
        U2 = g0 @ y + (h / 2) * g1 @ N1
        N2 = N(U2, t + h / 2)
        U3 = g0 @ y + (h / 2) * g1 @ N1 + h * g2 @ (N2 - N1)
        N3 = N(U3, t + h / 2)
        U4 = f0 @ y + h * f1 @ N1 + 2 * h * f2 @ (N3 - N1)
        N4 = N(U4, t + h)
        return f0 @ y + h * (f1 @ N1 + f2 @ (-3 * N1 + 2 * N2 + 2 * N3 - N4) + 4 * f3 @ (N1 - N2 - N3 + N4))

    def step_ho4(self, y, t, h, inp):
        f0, f1, f2, f3 = self.phis(h)
        g0, g1, g2, g3 = self.phis(h / 2)
        N = lambda yy, tt: self.f(yy, tt, inp) - self.L @ yy
        N1 = N(y, t)
        N2 = N(g0 @ y + h * (0.5 * g1) @ N1, t + h / 2)
        N3 = N(g0 @ y + h * ((0.5 * g1 - g2) @ N1 + g2 @ N2), t + h / 2)
        N4 = N(f0 @ y + h * ((f1 - 2 * f2) @ N1 + f2 @ (N2 + N3)), t + h)
        a52 = 0.5 * g2 - f3 + 0.25 * f2 - 0.5 * g3
        a54 = 0.25 * g2 - a52
        a51 = 0.5 * g1 - 2 * a52 - a54
        N5 = N(g0 @ y + h * (a51 @ N1 + a52 @ (N2 + N3) + a54 @ N4), t + h / 2)
        return f0 @ y + h * ((f1 - 3 * f2 + 4 * f3) @ N1 + (-f2 + 4 * f3) @ N4 + (4 * f2 - 8 * f3) @ N5)

    def observe(self, y, t, vg, si, q, v, ksub, max_ksub=2400):
        o = pm.algebraic(self.m, y, t, vg, si, q, v)
        return np.array([o.i[0].real, o.i[0].imag, o.v[0].real, o.v[0].imag, o.S_PCC.real, o.S_PCC.imag, o.Vdc,
                         o.Ppv, v, q, ksub / max_ksub])

    def run_golden(self, golden_npz, stepper, jump_fracs, quiet_steps, n_env_steps=None):
        """Integrate a golden case: a sim time step that starts at an input jump takes the steps
        `jump_fracs` (fractions of 1/60 s), every other one `quiet_steps` uniform steps.  Returns the
        rel-L2 error of the observation trajectory (first 10 components) against the golden."""
        g = np.load(golden_npz)
        n = int(g["n"])
        acts = g["actions"] if n_env_steps is None else g["actions"][:n_env_steps]
        vgt, sit = g["vgrid"], g["sinsol"]
        K = len(vgt) - 1
        s0, ds = int(round(float(g["ev_t0"]) * 60)), int(round(float(g["ev_dt"]) * 60))
        y, q, v = self.y0.copy(), 0.0, self.m.Vdc_ref0
        dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
        ksub, prev_row = 0, 0
        out = [self.observe(y, 0.0, vgt[0], sit[0], q, v, 0)]
        step = getattr(self, "step_" + stepper)
        for a in acts:
            a = int(a)
            q += dq * (int(a == 1) - int(a == 2))
            v += dv * (int(a == 3) - int(a == 4))
            for s in range(n):
                gk = ksub + s
                row = 0 if gk < s0 else min((gk - s0) // ds + 1, K)
                fracs = jump_fracs if (s == 0 or row != prev_row) else [1.0 / quiet_steps] * quiet_steps
                prev_row = row
                t = gk * DT
                for fr in fracs:
                    y = step(y, t, DT * fr, (vgt[row], sit[row], q, v))
                    t += DT * fr
            ksub += n
            row = 0 if ksub < s0 else min((ksub - s0) // ds + 1, K)
            out.append(self.observe(y, ksub * DT, vgt[row], sit[row], q, v, ksub))
        got, want = np.array(out)[:, :10], g["obs"][:len(out), :10]
        return float(np.linalg.norm(got - want) / np.linalg.norm(want))


def golden_path(name: str) -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "traj_%s.npz" % name)
