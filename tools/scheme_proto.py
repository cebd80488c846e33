#!/usr/bin/env python
"""CPU (fp64 / fp32 numpy): accuracy of candidate step SCHEDULES of the etdlin scheme on the committed converged-bridge
episodes -- what an env step costs in integrator stages against what it does to the parity figures.

    python tools/scheme_proto.py [case ...] [--f32]

Schedules (per segment between two input jumps; a PVDER-v0 env step is one segment):
  t2   tick + 2 equal ETD3RK steps                       (the shipped schedule)
  t1   tick + 1 ETD3RK step
  e2q1 tick + 2 when the segment starts at an EVENT row change, tick + 1 behind a mere action
  e2q1n the same, and no tick behind a mere action (one ETD3RK step over the whole env step)
  k4   tick + 1 step of Krogstad's ETDRK4 (4 stages) in f-form
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gym-solarpvder-environment_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bridge_cases  # noqa: E402
from gym_PVDER_amd import der_model, etd_linear as el  # noqa: E402

MICRO = 32


def krogstad_tableau(L, h):
    """Krogstad's ETDRK4 (J. Comput. Phys. 203, 2005): c = 0, 1/2, 1/2, 1."""
    f = el.phis(L, h)
    g = el.phis(L, 0.5 * h)
    Z = np.zeros_like(L)
    Es = [None, g[0], g[0], f[0]]
    A = [[Z, Z, Z, Z],
         [0.5 * h * g[1], Z, Z, Z],
         [0.5 * h * g[1] - h * g[2], h * g[2], Z, Z],
         [h * f[1] - 2.0 * h * f[2], Z, 2.0 * h * f[2], Z]]
    B = [h * (f[1] - 3.0 * f[2] + 4.0 * f[3]), h * (2.0 * f[2] - 4.0 * f[3]), h * (2.0 * f[2] - 4.0 * f[3]), h * (-f[2] + 4.0 * f[3])]
    return Es, A, f[0], B


class Scheme:
    def __init__(self, par, dtype):
        self.par, self.dtype = par, dtype
        self.c = el.folded(par)
        self.L, _ = el.jacobian(par, "50")
        self.cache = {}
        self.evals = 0

    def cc(self):
        return {k: (self.dtype(v) if isinstance(v, float) else v) for k, v in self.c.items()}

    def f(self, y, inp):
        self.evals += 1
        return el.f_bal(self.cc(), y, *inp).astype(self.dtype)

    def tick(self, y, inp, ticks=2):
        h = ticks / (60.0 * MICRO)
        key = ("tick", ticks)
        if key not in self.cache:
            self.cache[key] = (h * el.phis(self.L, h, 1)[1]).astype(self.dtype)
        return y + self.cache[key] @ self.f(y, inp)

    def etd3(self, y, inp, h):
        key = ("e3", round(h * 60 * MICRO * 1024))
        if key not in self.cache:
            self.cache[key] = [m.astype(self.dtype) for m in el.etd3rk_f_matrices(self.L, h)]
        M21, M31, M32, M41, M42, M43 = self.cache[key]
        F1 = self.f(y, inp)
        a = y + M21 @ F1
        F2 = self.f(a, inp)
        b = y + M31 @ F1 + M32 @ F2
        F3 = self.f(b, inp)
        return y + M41 @ F1 + M42 @ F2 + M43 @ F3

    def k4(self, y, inp, h):
        key = ("k4", round(h * 60 * MICRO * 1024))
        if key not in self.cache:
            r = el.f_form(self.L, *krogstad_tableau(self.L, h))
            self.cache[key] = [[m.astype(self.dtype) for m in row] for row in r]
        r = self.cache[key]
        F = [self.f(y, inp)]
        for row in r[:-1]:
            u = y + sum(M @ Fj for M, Fj in zip(row, F))
            F.append(self.f(u, inp))
        return y + sum(M @ Fj for M, Fj in zip(r[-1], F))

    def segment(self, kind, y, inp, ticks_total, at_event):
        if kind == "t2" or (kind in ("e2q1", "e2q1n") and at_event):
            y = self.tick(y, inp)
            h = (ticks_total - 2) / 2.0 / (60.0 * MICRO)
            return self.etd3(self.etd3(y, inp, h), inp, h)
        if kind in ("t1", "e2q1"):
            y = self.tick(y, inp)
            return self.etd3(y, inp, (ticks_total - 2) / (60.0 * MICRO))
        if kind == "e2q1n":
            return self.etd3(y, inp, ticks_total / (60.0 * MICRO))
        if kind == "k4" or (kind == "e2k4" and not at_event):
            y = self.tick(y, inp)
            return self.k4(y, inp, (ticks_total - 2) / (60.0 * MICRO))
        if kind == "k4n" or (kind == "e2k4n" and not at_event):
            return self.k4(y, inp, ticks_total / (60.0 * MICRO))
        if kind in ("e2k4", "e2k4n"):
            y = self.tick(y, inp)
            h = (ticks_total - 2) / 2.0 / (60.0 * MICRO)
            return self.etd3(self.etd3(y, inp, h), inp, h)
        raise ValueError(kind)


def observe(c, y, vg, si, q, v, tfrac):
    y = y.astype(np.float64)
    i = y[0] + 1j * y[1]
    va = complex(c["GR"], c["GI"]) * vg + complex(c["kviR"], c["kviI"]) * i
    S = 0.5 * c["PH"] * va * np.conj(i)
    ppv = np.maximum(0.0, y[6] * (c["pvA"] * si + c["pvB"] - c["pvB"] * np.exp(c["pvK"] * y[6])))
    return np.stack([i.real, i.imag, va.real, va.imag, S.real, S.imag, y[6], ppv, v + 0 * vg, q + 0 * vg, tfrac + 0 * vg], axis=-1)


def run(case, kind, dtype):
    g = bridge_cases.load(case)
    vgt, sit, acts = g["vgrid"], g["sinsol"], g["actions"]
    n, s0, ds = int(g["n"]), int(g["ev_s0"]), int(g["ev_ds"])
    K = vgt.shape[0] - 1
    N = acts.shape[1]
    par = der_model.param_vector("50")
    S = Scheme(par, dtype)
    y0 = der_model.steady_state(par, "50")
    y = np.tile(np.concatenate([y0[:6], y0[18:]])[:, None], (1, N)).astype(dtype)
    q = np.zeros(N, dtype=dtype)
    v = np.full(N, par[der_model.IDX["VDC_REF0"]], dtype=dtype)
    dq, dv = dtype(25.0 * n / 50e3), dtype(0.02 * n / 500.0)
    e = np.arange(N)
    rowof = lambda gk: 0 if gk < s0 else min((gk - s0) // ds + 1, K)
    ksub, out = 0, []
    for a in acts:
        q = (q + dq * ((a == 1).astype(dtype) - (a == 2).astype(dtype))).astype(dtype)
        v = (v + dv * ((a == 3).astype(dtype) - (a == 4).astype(dtype))).astype(dtype)
        s = 0
        while s < n:
            row = rowof(ksub + s)
            end = s + 1
            while end < n and rowof(ksub + end) == row:
                end += 1
            at_event = row > 0 and (ksub + s) == s0 + (row - 1) * ds
            inp = (vgt[row, e].astype(dtype), sit[row, e].astype(dtype), q, v)
            y = S.segment(kind, y, inp, (end - s) * MICRO, at_event)
            s = end
        ksub += n
        r = rowof(ksub)
        out.append(observe(S.c, y, vgt[r, e], sit[r, e], q, v, ksub / 60.0 / 40.0))
    got = np.array(out)                       # [steps, N, 11]
    want = g["obs"][:len(out)]
    err = np.linalg.norm(got - want, axis=2) / np.linalg.norm(want, axis=2)
    traj = np.linalg.norm((got - want).reshape(len(out), N, -1).transpose(1, 0, 2).reshape(N, -1), axis=1) / \
        np.linalg.norm(want.transpose(1, 0, 2).reshape(N, -1), axis=1)
    valid = np.ones_like(err, dtype=bool)
    if "m_abs" in g:                           # (beyond the duty-cycle / current limits the balanced prototype is not the model)
        valid = (np.maximum.accumulate(g["m_abs"][:len(out)], axis=0) < 0.99) & (np.maximum.accumulate(g["iref_abs"][:len(out)], axis=0) < 1.6)
    return err, traj, valid, S.evals / float(len(acts))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dtype = np.float32 if "--f32" in sys.argv else np.float64
    cases = args or ["episode_random", "episode_random_sample", "ramp_a3"]
    kinds = ["t2", "e2q1n", "k4", "k4n", "e2k4", "e2k4n"]
    print("# %s; per case: worst env step (rel-L2 of the 11 observations, steps inside the limits) / worst whole trajectory / f evaluations per env step" % dtype.__name__)
    for case in cases:
        for kind in kinds:
            err, traj, valid, ev = run(case, kind, dtype)
            w = np.where(valid, err, 0.0)
            print("%-24s %-6s worst step %.2e  median-env worst step %.2e  trajectory %.2e  f evals %.1f" % (
                case, kind, w.max(), np.median(w.max(axis=0)), traj.max(), ev))


if __name__ == "__main__":
    main()
