#!/usr/bin/env python
"""CPU (fp64, the oracle's own right-hand side): what kinds of LARGE-step schemes reach at the duty-cycle limit, against the converged
(event-locating) golden of the action-4 episode (tests/golden/bridge_ramp_a4.npz, env 1: first saturated env step 116) --
  held    the device's 23-state step (csrc etd_switched_step; tools/switched_proto.py SwRK): per phase the operator set of the anti-windup
          mode found at the START of an integrator step, held over the step -- at 1, 2, 4, ... steps per sim time step (15 ... 480 per
          env step): does it converge under refinement?
  rm      a predictor / corrector ("return mapping"): free exponential step; a phase whose |m| ends beyond the limit gets the outward
          increments of (x_p, u_p) scaled back onto the circle and the step is redone with that phase frozen; a phase AT the limit whose
          four masks all hold (every component wants outward) stays frozen without a predictor ("stuck" rule)
rel-L2 of the observations per env step, worst of each 5-step block from step 111, and the median over steps 116-134.
    python tools/saturated_step_prototypes.py > profiles/r05_saturated_step_prototypes.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
from switched_proto import SwRK, run_ramp


class RM(SwRK):
    def etd(self, y, t, h, inp, modes, xfrozen):
        L, (f0, f1, f2, f3) = self.phis_m(h, modes)
        _, (g0, g1, g2, g3) = self.phis_m(h / 2, modes)

        def F(yy, tt):
            f = self.f_free(yy, tt, inp)
            for p, md in enumerate(modes):
                if md & 1: f[6 * p + 4] = 0
                if md & 2: f[6 * p + 5] = 0
                if xfrozen[p]: f[6 * p + 2] = 0; f[6 * p + 3] = 0
            return f
        N = lambda yy, tt: F(yy, tt) - L @ yy
        N1 = N(y, t); N2 = N(g0 @ y + h * (0.5 * g1) @ N1, t + h / 2); N3 = N(g0 @ y + h * ((0.5 * g1 - g2) @ N1 + g2 @ N2), t + h / 2)
        N4 = N(f0 @ y + h * ((f1 - 2 * f2) @ N1 + f2 @ (N2 + N3)), t + h)
        a52 = 0.5 * g2 - f3 + 0.25 * f2 - 0.5 * g3; a54 = 0.25 * g2 - a52; a51 = 0.5 * g1 - 2 * a52 - a54
        N5 = N(g0 @ y + h * (a51 @ N1 + a52 @ (N2 + N3) + a54 @ N4), t + h / 2)
        return f0 @ y + h * ((f1 - 3 * f2 + 4 * f3) @ N1 + (-f2 + 4 * f3) @ N4 + (4 * f2 - 8 * f3) @ N5)

    def step_sw(self, y, t, h, inp):
        K = self.m.Kp_GCC
        f = self.f_free(y, t, inp)
        stuck = [False] * 3
        for p in range(3):
            c0 = np.array([y[6 * p + 2], y[6 * p + 3], y[6 * p + 4], y[6 * p + 5]])
            if abs(complex(c0[0] + K * c0[2], c0[1] + K * c0[3])) >= 1.0 - 1e-9:
                stuck[p] = all((c0[k] == 0.0) or (f[6 * p + 2 + k] * c0[k] > 0) for k in range(4))
        ys = self.etd(y, t, h, inp, tuple(3 if s else 0 for s in stuck), tuple(stuck))      # predictor
        clamp, newxu = list(stuck), [None] * 3
        for p in range(3):
            if stuck[p]:
                continue
            c0 = np.array([y[6 * p + 2], y[6 * p + 3], y[6 * p + 4], y[6 * p + 5]]); c1 = np.array([ys[6 * p + 2], ys[6 * p + 3], ys[6 * p + 4], ys[6 * p + 5]])
            w = np.array([1, 1, K, K])
            if abs(complex(c1[0] + K * c1[2], c1[1] + K * c1[3])) <= 1.0:
                continue
            clamp[p] = True
            d = c1 - c0
            out = np.array([d[k] * c1[k] > 0 for k in range(4)])
            base = c1.copy(); base[out] = c0[out]
            wd = w * d * out
            dm, mb = complex(wd[0] + wd[2], wd[1] + wd[3]), complex(base[0] + K * base[2], base[1] + K * base[3])
            qa, qb, qc = abs(dm) ** 2, 2 * (mb.real * dm.real + mb.imag * dm.imag), abs(mb) ** 2 - 1.0
            beta = 0.0
            if qc < 0 and qa > 0:
                beta = min(1.0, max(0.0, (-qb + np.sqrt(qb * qb - 4 * qa * qc)) / (2 * qa)))
            newxu[p] = base + beta * d * out
        if clamp == stuck:
            return ys
        y2 = y.copy()
        for p in range(3):
            if clamp[p] and not stuck[p]:
                y2[6 * p + 2:6 * p + 6] = newxu[p]
        return self.etd(y2, t, h, inp, tuple(3 if c else 0 for c in clamp), tuple(clamp))                 # corrector


if __name__ == "__main__":
    print(__doc__.split("    python")[0])
    for name, cls, quiets in (("held", SwRK, (1, 2, 4, 8, 16, 32)), ("rm", RM, (1, 2, 4, 8))):
        for quiet in quiets:
            R = cls()
            R.proj = False
            err = run_ramp(R, env=1, steps=135, quiet=quiet)
            print("%-5s %3d steps per sim time step: blocks from step 111: %s   median (116..134) %.1e" % (
                name, quiet, " ".join("%.1e" % err[k:k + 5].max() for k in range(111, 135, 5)), np.median(err[116:135])), flush=True)
