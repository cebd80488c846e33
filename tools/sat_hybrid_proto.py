#!/usr/bin/env python
"""CPU (fp64): the duty-cycle limit on the SLOW time scale -- a hybrid step for the 23-state path that follows the converged
(event-locating, Filippov) solution of the oracle with LARGE steps (reference path PVDER_env.py:166, :218-221, :229,
:480-487).  Statement of what the device's tail has to do; runs on the oracle's own right-hand side.

What the converged solution does at the limit (oracle/pvder_bridge_filippov.h, zoomed with .scratch tools):
per phase p and axis c (real / imaginary part) the current controller's low-pass state u_c can only move INWARD (towards zero) while
|m_p| >= m_limit: with D_c = e_c - u_c (e_c = iref_c - i_c; du_c/dt = wp D_c) the axis is FROZEN while h_c = D_c sgn(u_c) > 0.

  flight    every axis of the phase frozen: m_p constant, the phase's current loop OPEN -- di_c/dt = g (m_c - m_needed), g = Vdc / 2 Lf
            -- nothing fast, rates of 1-10 /s
  impact    h_c reaches zero from above at rate r = -dh_c/dt: the axis is released, the closed loop (omega_0^2 = wp Kp g, zeta =
            wp / 2 omega_0 = 0.17) swings u_c inward and the anti-windup freezes it again at the FIRST extremum, half a damped period
            (5.4 us) later: u_c jumps inward by (1 + rho) r / (g Kp), rho = exp(-pi zeta / sqrt(1 - zeta^2)) = 0.58, and h_c leaves at
            + rho r -- a bouncing ball with restitution rho; i_c does not move (1e-6).  If the jump takes |m_p| below the limit the OTHER
            axis, which wants outward, follows the circle (Filippov sliding for those 5 us)
  contact   the impacts' Zeno limit: the axis regulates (closed loop, u_c creeping inward), the other one follows the circle
  clamp     a component that reaches zero sits there while the phase is saturated
  entry / exit   |m_p| reaches the limit from inside with outward-moving components / no frozen component wants outward any more

    python tools/sat_hybrid_proto.py            (needs .scratch checkpoints for the window tests; the ramp test needs only the golden)
"""
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import numpy as np
from scipy.linalg import expm

from oracle import pvder_model as pm
from oracle.exp_rk_proto import DT, NS
from switched_proto import SwRK

ROT = [1.0, pm.A240, pm.A120]


class Hybrid(SwRK):
    """modes[p]: bit 0 = u_R frozen, bit 1 = u_I frozen; sat[p]: the phase is at the limit (its x components that would grow are held,
    frozen components that want outward follow the circle when |m| would fall below the limit)."""

    def __init__(self, contact_dm=5e-5, verbose=False):
        super().__init__()
        self.modes = [0, 0, 0]
        self.sat = [False, False, False]
        self.verbose = verbose
        self.contact_dm = contact_dm
        self.ev = {"entry": 0, "impact": 0, "contact": 0, "freeze": 0, "exit": 0, "clamp": 0, "follow": 0}
        self.K = self.m.Kp_GCC
        self.last_impact = None
        self.jump_margin = 4e-3          # an input jump counts as an entry when |m|^2 > (1 - margin) limit^2

    # ---- pieces of the state
    def xu(self, y, p):
        return np.array([y[6 * p + 2], y[6 * p + 3]]), np.array([y[6 * p + 4], y[6 * p + 5]])

    def mvec(self, y, p):
        x, u = self.xu(y, p)
        return x + self.K * u

    def s_of(self, y, p):
        mm = self.mvec(y, p)
        return mm[0] * mm[0] + mm[1] * mm[1] - self.m.m_limit ** 2

    def demand(self, y, t, inp):
        """f (free field) and D[p] = (e - u) of phase p as a 2-vector"""
        f = self.f_free(y, t, inp)
        return f, [np.array([f[6 * p + 4], f[6 * p + 5]]) / self.m.wp for p in range(3)]

    def demand_rate(self, y, t, inp, f, p):
        """d/dt (e_p) with u_p held: d iref_p/dt - d i_p/dt (complex -> 2-vector), from the free field at y"""
        m = self.m
        o = pm.algebraic(m, y, t, *inp)
        dS = 0
        for q in range(3):
            di = complex(f[6 * q], f[6 * q + 1])
            dS += 0.5 * ((m.kv_i * di) * o.i[q].conjugate() + o.v[q] * di.conjugate())
        dia = complex(f[19] - m.Kp_DC * f[18], f[20] + m.Kp_Q * dS.imag)
        de = dia * ROT[p] - complex(f[6 * p], f[6 * p + 1])
        return np.array([de.real, de.imag]), o

    def loop(self, Vdc):
        """(g, rho) of the released axis: di_c/dt = g dm_c; restitution of the swing"""
        m = self.m
        g = Vdc / (2.0 * m.Lf)
        w0 = np.sqrt(m.wp * self.K * g)
        z = m.wp / (2.0 * w0)
        return g, np.exp(-np.pi * z / np.sqrt(1.0 - z * z))

    def loop_of(self, y, t, inp, p, c, de):
        """the same from the field itself: g_c = -d(de_c/dt)/dm_c includes what the current does to its own reference"""
        y2 = y.copy(); y2[6 * p + 4 + c] += 1e-9
        # if the inward jump takes |m| below the limit and the other axis (frozen, wanting outward) follows the circle, the two move
        # together: dm_o = -(m_c / m_o) dm_c
        mm = self.mvec(y, p); o_ = 1 - c
        x, u = self.xu(y, p)
        Dp = np.array([self.f_free(y, t, inp)[6 * p + 4 + k] for k in range(2)]) / self.m.wp
        follows = (self.modes[p] >> o_) & 1 and Dp[o_] * u[o_] > 0
        if follows and mm[c] * u[c] > 0 and abs(mm[o_]) > 1e-3:
            y2[6 * p + 4 + o_] += -(mm[c] / mm[o_]) * 1e-9
        f2 = self.f_free(y2, t, inp)
        de2, _ = self.demand_rate(y2, t, inp, f2, p)
        g = -(de2[c] - de[c]) / (1e-9 * self.K)
        w0 = np.sqrt(self.m.wp * self.K * g)
        z = self.m.wp / (2.0 * w0)
        return g, np.exp(-np.pi * z / np.sqrt(1.0 - z * z))

    # ---- the exponential step with per-component masks
    def etd(self, y, t, h, inp, modes, xmask):
        L, (f0, f1, f2, f3) = self.phis_m(h, tuple(modes))
        _, (g0, g1, g2, g3) = self.phis_m(h / 2, tuple(modes))

        def F(yy, tt):
            f = self.f_free(yy, tt, inp)
            for p, md in enumerate(modes):
                if md & 1: f[6 * p + 4] = 0
                if md & 2: f[6 * p + 5] = 0
                if xmask[p][0]: f[6 * p + 2] = 0
                if xmask[p][1]: f[6 * p + 3] = 0
            return f
        N = lambda yy, tt: F(yy, tt) - L @ yy
        N1 = N(y, t); N2 = N(g0 @ y + h * (0.5 * g1) @ N1, t + h / 2); N3 = N(g0 @ y + h * ((0.5 * g1 - g2) @ N1 + g2 @ N2), t + h / 2)
        N4 = N(f0 @ y + h * ((f1 - 2 * f2) @ N1 + f2 @ (N2 + N3)), t + h)
        a52 = 0.5 * g2 - f3 + 0.25 * f2 - 0.5 * g3; a54 = 0.25 * g2 - a52; a51 = 0.5 * g1 - 2 * a52 - a54
        N5 = N(g0 @ y + h * (a51 @ N1 + a52 @ (N2 + N3) + a54 @ N4), t + h / 2)
        yn = f0 @ y + h * ((f1 - 3 * f2 + 4 * f3) @ N1 + (-f2 + 4 * f3) @ N4 + (4 * f2 - 8 * f3) @ N5)
        for p, md in enumerate(modes):            # held components are held exactly
            if md & 1: yn[6 * p + 4] = y[6 * p + 4]
            if md & 2: yn[6 * p + 5] = y[6 * p + 5]
            if xmask[p][0]: yn[6 * p + 2] = y[6 * p + 2]
            if xmask[p][1]: yn[6 * p + 3] = y[6 * p + 3]
        return yn

    def xmask_of(self, y):
        """x_c of a saturated phase is held while it would grow: dx_c = Ki u_c has the sign of u_c"""
        out = []
        for p in range(3):
            x, u = self.xu(y, p)
            out.append([self.sat[p] and (u[c] * x[c] > 0 or x[c] == 0.0) for c in range(2)])
        return out

    # ---- the circle
    def follow_circle(self, y, p, D, mover=-1, m_from=None, exclude=-1):
        """The circle constraint of a saturated phase after component `mover` (or, mover = -1, the slow motion of a step) has moved:
        wherever |m_p| would fall below the limit the frozen components that want outward (h_c > 0; a component AT zero moves with its
        demand, through m_c = 0 if need be) move out -- instantly on this time scale, K wp D_c ~ 1e8 /s -- and, being frozen, do NOT come
        back when the mover's path lets |m_p| grow again (a ratchet): what counts is the SMALLEST |m_mover| along its path (zero if it
        changes sign).  Returns False when the phase is inside the limit and nothing may move (it then leaves the limit)."""
        x, u = self.xu(y, p)
        lim2 = self.m.m_limit ** 2
        mm = x + self.K * u
        cand = []
        for c in range(2):
            if c == mover or c == exclude or not (self.modes[p] >> c) & 1:
                continue
            if D[c] * u[c] > 1e-12 * abs(u[c]):          # (a component AT zero is no pusher: once nothing holds the phase at the limit it
                cand.append(c)                            #  runs free with the rest -- exit, and `input_jump` finds where that ends)
        if mover >= 0:
            o = 1 - mover
            m0, m1 = m_from, mm[mover]
            mmin2 = 0.0 if m0 * m1 < 0 else min(m0 * m0, m1 * m1)
            if o in cand and lim2 - mmin2 > mm[o] * mm[o]:
                dirn = np.sign(D[o])
                tgt = dirn * np.sqrt(lim2 - mmin2)
                if (tgt - mm[o]) * dirn > 0:
                    # ... as far as its demand lasts: D_o = e_o - u_o falls one for one with the move
                    du_o = min((tgt - mm[o]) * dirn / self.K, abs(D[o]))
                    y[6 * p + 4 + o] = u[o] + dirn * du_o
                    self.ev["follow"] += 1
            return self.s_of(y, p) >= -1e-13
        if mm @ mm >= lim2 - 1e-13:
            return True
        if not cand:
            return False
        d = np.zeros(2)
        for c in cand:
            d[c] = D[c]
        dm = self.K * d
        qa, qb, qc = dm @ dm, 2 * (mm @ dm), mm @ mm - lim2
        disc = qb * qb - 4 * qa * qc
        if disc < 0:
            return False
        beta = (-qb + np.sqrt(disc)) / (2 * qa)
        if beta < 0:
            return False
        y[6 * p + 4], y[6 * p + 5] = u + min(beta, 1.0) * d          # (beta = 1: the demands are used up)
        self.ev["follow"] += 1
        return beta <= 1.0

    # ---- one macro step with the slow-time-scale events located inside it
    def advance(self, y, t, h, inp, depth=0):
        """from (t, y) over h; returns y(t + h).  Events split the step."""
        K, m = self.K, self.m
        t_end = t + h
        self.input_jump(y, t, inp)
        guard = 0
        while t < t_end - 1e-15:
            guard += 1
            if guard > 200:
                raise RuntimeError("event loop does not advance at t = %.9f (modes %s)" % (t, self.modes))
            hh = t_end - t
            f0, D0 = self.demand(y, t, inp)
            modes0 = list(self.modes)
            y1 = self.etd(y, t, hh, inp, modes0, self.xmask_of(y))
            f1, D1 = self.demand(y1, t + hh, inp)
            # -- event functions at both ends: (kind, p, c, value0, value1); an event = sign change - to + of `value`
            cands = []
            for p in range(3):
                x0, u0 = self.xu(y, p); x1, u1 = self.xu(y1, p)
                if not self.sat[p]:
                    s0, s1 = self.s_of(y, p), self.s_of(y1, p)
                    if s1 > 0:
                        cands.append(("entry", p, -1, s0, s1))
                    continue
                for c in range(2):
                    if (modes0[p] >> c) & 1:                       # frozen: impact when h_c = D_c sgn(u_c) falls through zero
                        if u0[c] == 0.0:
                            continue                                # clamped at zero: stays while saturated
                        sg = np.sign(u0[c])
                        a, b = -D0[p][c] * sg, -D1[p][c] * sg
                        if a >= 0 and (self.last_impact == (p, c, t) or getattr(self, "last_settle", {}).get(p) == t):
                            # it has just been frozen ON h_c = 0 / the phase has just been settled: no second release at this
                            # instant -- but it may well come back later in the step
                            if b <= 0:
                                continue
                            a = -1e-30
                        if b > 0 or a > 1e-9:
                            if 0 <= a <= 1e-9:   # h_c = 0 at the start (just impacted): an impact only if falling
                                de0, _ = self.demand_rate(y, t, inp, f0, p)
                                if -de0[c] * sg <= 0:
                                    continue
                            cands.append(("impact", p, c, a, b))
                    else:                                           # regulating axis of a saturated phase: freezes when it turns outward
                        if u1[c] * u0[c] < 0:
                            cands.append(("clamp", p, c, -abs(u0[c]), abs(u1[c])))
                        elif abs(u1[c]) > abs(u0[c]) * (1 + 1e-9) + 1e-15:
                            cands.append(("freeze", p, c, -1.0, 1.0))
            if not cands:
                y, t = y1, t + hh
                self.after_step(y, t, inp)
                continue
            # -- the earliest one, located by the secant on the step length (two refinements)
            best = None
            for kind, p, c, a, b in cands:
                th = 0.0 if a >= 0 else a / (a - b)
                if kind == "freeze":
                    th = 0.0
                if best is None or th < best[0]:
                    best = (th, kind, p, c, a, b)
            th, kind, p, c, a, b = best
            if kind == "impact" and th > 0 and self.max_locate:
                # the device's way: the cubic through the event function's values and slopes at the two ends (both come with right-hand
                # sides the scheme has), the step re-taken to that instant, a few Newton corrections with the slope found there; the
                # landing is pushed 0.05 us beyond the estimate (the event must have HAPPENED where it is handled)
                sa = self.event_slope(kind, p, c, y, y, t, inp) * hh
                sb = self.event_slope(kind, p, c, y, y1, t + hh, inp) * hh
                th = self.cubic_root(a, b, sa, sb, th)
                for it in range(self.max_locate):
                    yt = self.etd(y, t, hh * th, inp, modes0, self.xmask_of(y))
                    val = self.event_value(kind, p, c, y, yt, t + hh * th, inp)
                    sl = self.event_slope(kind, p, c, y, yt, t + hh * th, inp) * hh
                    if sl <= 0: break
                    dth = -val / sl
                    if abs(dth) * hh < 5e-8: break
                    th = min(max(th + dth, 1e-9), 1.0)
                th = min(th + 5e-8 / hh, 1.0)
                yt = self.etd(y, t, hh * th, inp, modes0, self.xmask_of(y))
            elif kind not in ("freeze",) and th > 0:
                lo, flo, hi, fhi = 0.0, a, 1.0, b
                # (an ENTRY comes from a phase whose loop is closed: its low-pass state carries K 1e-5 of noise behind a large step, the
                #  slope of |m|^2 - 1 is useless -- values only; entries are rare: 8 re-taken steps when the device's way is asked for)
                for it in range(8 if self.max_locate else 60):        # secant and bisection in turns: the bracket ends below 0.1 us
                    yt = self.etd(y, t, hh * th, inp, modes0, self.xmask_of(y))
                    val = self.event_value(kind, p, c, y, yt, t + hh * th, inp)
                    if val > 0: hi, fhi = th, val
                    else: lo, flo = th, val
                    if (hi - lo) * hh < 1e-7: break
                    th = lo + (hi - lo) * (-flo) / (fhi - flo) if it % 2 else 0.5 * (lo + hi)
                    th = min(max(th, lo + 0.02 * (hi - lo)), hi - 0.02 * (hi - lo))
                th = hi
                yt = self.etd(y, t, hh * th, inp, modes0, self.xmask_of(y))
            elif th > 0:
                yt = self.etd(y, t, hh * th, inp, modes0, self.xmask_of(y))
            else:
                yt = y.copy()
            y, t = yt, t + hh * th
            self.apply_event(kind, p, c, y, t, inp, at_start=(th == 0.0))
            self.after_step(y, t, inp, exclude=(p, c) if kind == "impact" else (-1, -1))
        return y

    def input_jump(self, y, t, inp):
        """An action moves iref by 0.0012: the low-pass state of a FREE phase would follow within 5 us -- a move of 7 in m: every
        action takes every phase to the limit within nanoseconds.  Clear of the limit the phase leaves it again
        within 0.05-1 ms and ignoring the episode costs 1e-5; NEAR the limit it may stay -- the entry then happens HERE, along the
        demand, and the usual rules (impact of the components released inward, clamp, exit) take it from there."""
        f, D = self.demand(y, t, inp)
        for p in range(3):
            if self.sat[p]:
                continue
            x, u = self.xu(y, p)
            if self.K * np.hypot(*D[p]) < 1e-3:
                continue
            mv = x + self.K * (u + D[p])
            lim2 = self.m.m_limit ** 2
            mm = x + self.K * u
            if mv @ mv <= lim2 or mm @ mm < lim2 * (1.0 - self.jump_margin):
                continue                     # (clear of the limit the phase is back inside within |d iref| / (g (1 - |m|)) < 1 ms: 1e-5)
            dm = self.K * D[p]
            qa, qb, qc = dm @ dm, 2 * (mm @ dm), mm @ mm - lim2
            beta = (-qb + np.sqrt(qb * qb - 4 * qa * qc)) / (2 * qa)
            y[6 * p + 4], y[6 * p + 5] = u + beta * D[p]
            self.sat[p], self.modes[p] = True, 3
            self.ev["entry"] += 1
            if self.verbose: print("  t %.6f entry   phase %d behind an input jump (beta %.3e)" % (t, p, beta))

    max_locate = 0

    @staticmethod
    def cubic_root(a, b, sa, sb, th0):
        """first root in (0, 1) of the Hermite cubic with values a < 0 < b and slopes sa, sb (per unit of theta); Newton from the
        secant's estimate, bisection-safe"""
        c0, c1, c2, c3 = a, sa, 3 * (b - a) - 2 * sa - sb, 2 * (a - b) + sa + sb
        lo, hi, th = 0.0, 1.0, th0
        for _ in range(20):
            v = ((c3 * th + c2) * th + c1) * th + c0
            if v > 0: hi = th
            else: lo = th
            d = (3 * c3 * th + 2 * c2) * th + c1
            tn = th - v / d if d > 0 else 0.5 * (lo + hi)
            if not (lo < tn < hi): tn = 0.5 * (lo + hi)
            if abs(tn - th) < 1e-12: break
            th = tn
        return th

    def event_slope(self, kind, p, c, y0, y, t, inp):
        """d/dt of event_value along the held-mode flow (positive where the event function rises through zero)"""
        f = self.f_free(y, t, inp)
        K = self.K
        if kind == "entry":
            x, u = self.xu(y, p)
            mm = x + K * u
            return 2.0 * (mm[0] * (f[6 * p + 2] + K * f[6 * p + 4]) + mm[1] * (f[6 * p + 3] + K * f[6 * p + 5]))
        if kind == "impact":
            de, _ = self.demand_rate(y, t, inp, f, p)
            return -de[c] * np.sign(y0[6 * p + 4 + c])
        if kind == "clamp":
            return -f[6 * p + 4 + c] * np.sign(y0[6 * p + 4 + c])
        return 1.0

    def event_value(self, kind, p, c, y0, y, t, inp):
        if kind == "entry":
            return self.s_of(y, p)
        if kind == "impact":
            f, D = self.demand(y, t, inp)
            return -D[p][c] * np.sign(y0[6 * p + 4 + c])
        if kind == "clamp":
            return -y[6 * p + 4 + c] * np.sign(y0[6 * p + 4 + c])
        return 1.0

    def apply_event(self, kind, p, c, y, t, inp, at_start=False):
        K, m = self.K, self.m
        self.ev[kind] += 1
        f, D = self.demand(y, t, inp)
        x, u = self.xu(y, p)
        if kind == "entry":
            # onto the circle (the located point is within the secant's tolerance of it); components whose demand points outward freeze
            mm = x + K * u
            sc = m.m_limit / np.hypot(mm[0], mm[1])
            y[6 * p + 4], y[6 * p + 5] = (mm * sc - x) / K
            x, u = self.xu(y, p)
            md = 0
            for cc in range(2):
                if D[p][cc] * u[cc] > 0 or u[cc] == 0.0:
                    md |= 1 << cc
            self.modes[p] = md
            self.sat[p] = md != 0
            if self.verbose: print("  t %.6f entry   phase %d modes %d" % (t, p, md))
            return
        if kind == "clamp":
            y[6 * p + 4 + c] = 0.0
            self.modes[p] |= 1 << c
            if self.verbose: print("  t %.6f clamp   phase %d axis %d" % (t, p, c))
            return
        if kind == "freeze":
            self.modes[p] |= 1 << c
            if self.verbose: print("  t %.6f freeze  phase %d axis %d" % (t, p, c))
            return
        # impact: the released axis' closed loop, zeta = inward displacement of u_c, v = -h_c (> 0: released):
        #   zeta' = wp v,  v' = r - g K zeta - zeta'   ->   zeta'' + wp zeta' + wp g K zeta = wp r,   zeta(0) = 0, zeta'(0) = wp v0
        # frozen again at the first t* > 0 with zeta' = 0 (v = 0); v0 = 0 (a located crossing): zeta(t*) = (1 + rho) r / (g K)
        de, o = self.demand_rate(y, t, inp, f, p)
        sg = np.sign(u[c])
        r = -de[c] * sg                                   # rate at which h_c falls
        v0 = max(-D[p][c] * sg, 0.0) if at_start else 0.0
        if v0 <= 1e-9: v0 = 0.0      # (a LOCATED crossing is on the surface: what is left is the secant's tolerance)
        g, rho = self.loop_of(y, t, inp, p, c, de)
        wp = m.wp
        w0 = np.sqrt(wp * K * g); sig = 0.5 * wp; wd = np.sqrt(w0 * w0 - sig * sig)
        zeq = r / (g * K)
        A = -zeq; B = (wp * v0 + sig * A) / wd
        # zeta'(t) = e^{-sig t} [ (wd B - sig A) cos - (wd A + sig B) sin ]
        num, den = wd * B - sig * A, wd * A + sig * B
        if v0 == 0.0:
            zeta = (1.0 + rho) * zeq                      # (half a damped period: the closed form of the docstring)
        else:
            ts = np.arctan2(num, den) / wd
            if ts <= 1e-12: ts += np.pi / wd
            zeta = zeq + np.exp(-sig * ts) * (A * np.cos(wd * ts) + B * np.sin(wd * ts))
        dm = K * zeta
        if v0 == 0.0 and dm < self.contact_dm:
            if r <= 0:
                return
            # the impacts' Zeno limit: the axis regulates from here on
            self.modes[p] &= ~(1 << c)
            self.ev["contact"] += 1
            if self.verbose: print("  t %.6f contact phase %d axis %d (r %.3e, g %.1f rho %.3f)" % (t, p, c, r, g, rho))
            return
        du = max(zeta, 0.0)
        self.last_impact = (p, c, t)
        m_from = x[c] + K * u[c]
        if du >= abs(u[c]):
            y[6 * p + 4 + c] = 0.0
            self.ev["clamp"] += 1
        else:
            y[6 * p + 4 + c] = u[c] - sg * du
            # frozen again ON the surface h_c = 0: what the current moved during the swing (<= 1e-6)
            y[6 * p + c] -= -(D[p][c] + sg * du)
        if not self.follow_circle(y, p, D[p], mover=c, m_from=m_from):
            self.sat[p] = False; self.modes[p] = 0; self.ev["exit"] += 1
            if self.verbose: print("  t %.6f exit    phase %d (behind the impact: nothing holds |m| at the limit)" % (t, p))
            self.input_jump(y, t, inp)
        if self.verbose: print("  t %.6f impact  phase %d axis %d  r %.4f  K du %.3e -> K u %.6f" % (t, p, c, r, dm, K * y[6 * p + 4 + c]))

    def after_step(self, y, t, inp, exclude=(-1, -1)):
        """the circle constraint of saturated phases behind the slow motion of a step, and their exit"""
        for p in range(3):
            if not self.sat[p]:
                continue
            if self.s_of(y, p) < -1e-13:
                f, D = self.demand(y, t, inp)
                if not self.follow_circle(y, p, D[p], exclude=exclude[1] if exclude[0] == p else -1):
                    self.sat[p] = False
                    self.modes[p] = 0
                    self.ev["exit"] += 1
                    if self.verbose: print("  t %.6f exit    phase %d" % (t, p))
                    self.input_jump(y, t, inp)


class HybridMicro(Hybrid):
    """The same macro scheme -- large exponential steps, the slow-time-scale events located inside them -- with every discrete happening
    (an axis released, an input jump, the circle constraint violated, entry) resolved by ONE routine: `settle`, a micro-integration of the
    phase's FAST pair (D_p, u_p) -- 4 real states, the slow ones frozen, linearised where it starts -- under the oracle's own rules
    (oracle/pvder_bridge_filippov.h: regions IN / OUT / SLIDE, G masks, Filippov's alpha, components put on zero), until nothing fast is
    left.  No closed-form impact map, no follower rules: the swing, the restitution, the ratchet, ping-pong between the axes, the clamp
    and the exit all come out of the same 4-state integration (0.25 us steps, 5.4 us per swing)."""

    hmic = 0.25e-6
    _rep = {}
    tau_about = 20e-6            # a held component that will be released within this time keeps the micro-integration going
    max_micro = 20000

    def settle(self, y, t, inp, p, why="", trust_u=False):
        m, K = self.m, self.K
        lim2 = m.m_limit ** 2
        f, D = self.demand(y, t, inp)
        if trust_u:
            # an axis that arrives from a LARGE step with its loop closed (a free phase; the regulating axis of a saturated one) carries
            # its current to 1e-5 -- nothing for the current, a demand of K 1e-5 = 0.06 in m for the low-pass state.  On the slow
            # manifold D = (du/dt) / wp ~ 1e-9: put the current there.
            for c in range(2):
                if not (self.modes[p] >> c) & 1:
                    y[6 * p + c] += D[p][c]
            f, D = self.demand(y, t, inp)
        de0, o = self.demand_rate(y, t, inp, f, p)
        x, u0 = self.xu(y, p)
        # linearisation of d(e)/dt in u (2 x 2: -g K on the diagonal, the reference's own feedback and the cross terms beside it)
        Ge = np.zeros((2, 2))
        for c in range(2):
            y2 = y.copy(); y2[6 * p + 4 + c] += 1e-9
            f2 = self.f_free(y2, t, inp)
            de2, _ = self.demand_rate(y2, t, inp, f2, p)
            Ge[:, c] = (de2 - de0) / 1e-9
        wp = m.wp
        u, Dm = u0.copy(), D[p].copy()
        h = self.hmic

        def decide(u, Dm):
            """the region from the state alone (fil_decide) and the G masks"""
            fu = wp * Dm
            mm = x + K * u
            sv = mm @ mm - lim2
            G = [(u[c] == 0.0) or (fu[c] * u[c] > 0) for c in range(2)]
            if sv > 1e-11: return 1, G
            if sv < -1e-11: return 0, G
            N = 2 * K * (mm @ fu)
            Dn = 2 * K * sum(mm[c] * fu[c] for c in range(2) if G[c])
            if N <= 0: return 0, G
            if N - Dn >= 0: return 1, G
            return 2, G

        def field(u, Dm, reg):
            """(u', D') with the region held over the step (as the oracle holds it: fil_rhs)"""
            fu = wp * Dm
            up = fu.copy()
            if reg != 0:
                mm = x + K * u
                G = [(u[c] == 0.0) or (fu[c] * u[c] > 0) for c in range(2)]
                if reg == 1:
                    for c in range(2):
                        if G[c]: up[c] = 0.0
                else:
                    N = 2 * K * (mm @ fu)
                    Dn = 2 * K * sum(mm[c] * fu[c] for c in range(2) if G[c])
                    al = min(max(N / Dn, 0.0), 1.0) if Dn > 0 else 0.0
                    for c in range(2):
                        if G[c]: up[c] *= (1.0 - al)
            Dp = de0 + Ge @ (u - u0) - up
            return up, Dp

        n, quiet, elapsed = 0, 0, 0.0
        while n < self.max_micro:
            reg, G = decide(u, Dm)
            up1, Dp1 = field(u, Dm, reg)
            # quiet: nothing moves fast, and no held component is about to be released (h_c = 0 and falling)
            fast = K * np.abs(up1).max() * 5e-6 >= 1e-7
            about = False
            for c in range(2):
                if reg != 0 and G[c] and u[c] != 0.0:
                    sg = np.sign(u[c])
                    if Dp1[c] * sg < 0 and Dm[c] * sg < max(1e-10, -Dp1[c] * sg * self.tau_about): about = True
            if not fast and not about:
                quiet += 1
                if quiet >= 2: break
            else:
                quiet = 0
            # a component that moves fast (a violent release: r ~ 50 /s) takes the step in pieces: K |du| <= 1e-3 per micro step
            h = self.hmic / max(1.0, np.ceil(K * np.abs(up1).max() * self.hmic / 1e-3))
            up2, Dp2 = field(u + 0.5 * h * up1, Dm + 0.5 * h * Dp1, reg)
            up3, Dp3 = field(u + 0.5 * h * up2, Dm + 0.5 * h * Dp2, reg)
            up4, Dp4 = field(u + h * up3, Dm + h * Dp3, reg)
            du = (h / 6) * (up1 + 2 * up2 + 2 * up3 + up4)
            dD = (h / 6) * (Dp1 + 2 * Dp2 + 2 * Dp3 + Dp4)
            th = 1.0
            mm0 = x + K * u
            s0 = mm0 @ mm0 - lim2
            dmm = K * du
            qa, qb = dmm @ dmm, 2 * (mm0 @ dmm)
            s1 = s0 + qb + qa
            if qa > 0 and ((reg == 0 and s1 > 0) or (reg == 1 and s1 < 0)):     # onto the circle (from inside / from outside): stop ON it
                disc = qb * qb - 4 * qa * s0
                if disc >= 0:
                    r1, r2 = (-qb - np.sqrt(disc)) / (2 * qa), (-qb + np.sqrt(disc)) / (2 * qa)
                    roots = [r for r in (r1, r2) if 1e-14 < r <= 1.0]
                    if roots: th = min(roots)
            zc = -1
            if reg != 0:
                for c in range(2):                              # a component through zero while the phase is not inside: onto zero
                    if u[c] != 0.0 and (u[c] + du[c]) * u[c] < 0:
                        tz = u[c] / (-du[c])
                        if tz < th: th, zc = tz, c
            un, Dn_ = u + th * du, Dm + th * dD
            if zc >= 0: un[zc] = 0.0
            if reg == 2:                                        # sliding: back onto the circle along the components that moved
                mm1 = x + K * un
                sv = mm1 @ mm1 - lim2
                gr = np.array([2 * K * mm1[c] if du[c] != 0.0 else 0.0 for c in range(2)])
                if gr @ gr > 0: un = un - sv * gr / (gr @ gr)
            u, Dm = un, Dn_
            n += 1
            elapsed += th * h
        # the current moved by what the demand did not account for: e = D + u
        de_total = (Dm + u) - (D[p] + u0)
        y[6 * p + 4], y[6 * p + 5] = u
        y[6 * p] += (de0[0] + f[6 * p]) * elapsed - de_total[0]              # d i = d iref - d e
        y[6 * p + 1] += (de0[1] + f[6 * p + 1]) * elapsed - de_total[1]
        # classification for the macro step
        reg, G = decide(u, Dm)
        md = 0
        if reg != 0:
            up1, Dp1 = field(u, Dm, reg)
            for c in range(2):
                if not G[c]:
                    continue
                # held -- unless it sits ON its release surface with the demand still falling: it regulates (the impacts' Zeno limit;
                # the macro step carries it with its loop closed and freezes it when it turns outward)
                sg = np.sign(u[c])
                if u[c] != 0.0 and K * abs(Dm[c]) < 1e-4 and Dp1[c] * sg < 0:
                    self.ev["contact"] += 1
                    continue
                md |= 1 << c
        self.modes[p] = md
        self.sat[p] = reg != 0
        if not hasattr(self, "last_settle"): self.last_settle = {}
        self.last_settle[p] = t
        self.ev["settle"] = self.ev.get("settle", 0) + 1
        self.ev["micro"] = self.ev.get("micro", 0) + n
        if self.verbose:
            print("  t %.6f settle  phase %d (%s): %d micro steps -> modes %d sat %d  K u (%+.6f, %+.6f)  K D (%+.2e, %+.2e)" % (
                t, p, why, n, self.modes[p], self.sat[p], K * u[0], K * u[1], K * Dm[0], K * Dm[1]))

    rederive = False
    round32 = False

    def derive_modes(self, y, t, inp):
        """modes and sat from the STATE alone (what a kernel launch has: no discrete state is carried between env steps) -- the oracle
        does the same at the start of every sim time step (fil_settle)"""
        f, D = self.demand(y, t, inp)
        lim2 = self.m.m_limit ** 2
        for p in range(3):
            x, u = self.xu(y, p)
            mm = x + self.K * u
            sv = mm @ mm - lim2
            md = 0
            if sv > -self.derive_tol:
                for c in range(2):
                    if u[c] == 0.0 or D[p][c] * u[c] > 0: md |= 1 << c
            self.modes[p], self.sat[p] = md, md != 0

    derive_tol = 1e-6

    def input_jump(self, y, t, inp):
        if self.rederive and abs(t / (15 * DT) - round(t / (15 * DT))) < 1e-9:
            if self.round32:            # the state rows between two launches are fp32 (the PLL angle in its relative form: delta = wte - wbase t)
                wte = y[22]
                y[:] = y.astype(np.float32).astype(np.float64)
                y[22] = float(np.float32(wte - pm.WBASE * t)) + pm.WBASE * t
            self.derive_modes(y, t, inp)
        f, D = self.demand(y, t, inp)
        lim2 = self.m.m_limit ** 2
        for p in range(3):
            x, u = self.xu(y, p)
            mm = x + self.K * u
            if self.sat[p]:
                # a saturated phase whose frozen components are released by the jump (h_c < 0 by a finite amount)
                rel = any(((self.modes[p] >> c) & 1) and u[c] != 0.0 and -D[p][c] * np.sign(u[c]) > 1e-9 for c in range(2))
                if rel: self.settle(y, t, inp, p, "released by an input jump")
                continue
            # where the phase's closed loop is heading: the rest point of the fast pair, D = 0 and dD/dt = 0 -- an action moves D by
            # 0.0012 (K D = 7), a grid-voltage event moves de/dt by g dm_needed.  Beyond the limit: the phase is about to enter and
            # stay -- the fast pair takes it there (an entry within microseconds, which a large free step cannot locate)
            if self.K * np.hypot(*D[p]) < 1e-3 * 0 + 0.0 and False:
                continue
            de0, o = self.demand_rate(y, t, inp, f, p)
            jump = self.K * np.hypot(*D[p]) > 0.5 or np.hypot(*de0) > 0.5
            if not jump:
                continue                      # (a slow approach: the entry is located inside the step as usual)
            # where the phase ENDS once its current has caught up (0.05 - 1 ms): the duty cycle that holds the current on its new
            # reference.  Inside the limit: the phase visits the limit and leaves again, which costs 1e-5 to ignore; at or beyond it:
            # the phase enters now and stays
            g = o.Vdc / (2.0 * self.m.Lf)
            y2 = y.copy()
            y2[6 * p] += D[p][0]; y2[6 * p + 1] += D[p][1]               # i := iref - u
            f2 = self.f_free(y2, t, inp)
            mq = x + self.K * u - np.array([f2[6 * p], f2[6 * p + 1]]) / g
            if mq @ mq <= lim2 * (1.0 - self.jump_margin):
                continue
            self.settle(y, t, inp, p, "input jump that ends beyond the limit")

    def apply_event(self, kind, p, c, y, t, inp, at_start=False):
        self.ev[kind] += 1
        if kind == "freeze":
            self.modes[p] |= 1 << c
            if self.verbose: print("  t %.6f freeze  phase %d axis %d" % (t, p, c))
            return
        if kind == "clamp":
            y[6 * p + 4 + c] = 0.0
            self.modes[p] |= 1 << c
            return
        # a release that repeats within a microsecond of the last one of the same axis, again and again: the two time scales have
        # met (the impacts' Zeno limit inside one macro step) -- the axis regulates for the rest of the step
        key = (p, c)
        if self._rep.get(key, (None, 0))[0] is not None and t - self._rep[key][0] < 2e-6:
            self._rep[key] = (t, self._rep[key][1] + 1)
        else:
            self._rep[key] = (t, 1)
        if self._rep[key][1] > 4 and kind == "impact":
            self.modes[p] &= ~(1 << c)
            self.ev["contact"] += 1
            if self.verbose: print("  t %.6f contact phase %d axis %d (repeated releases)" % (t, p, c))
            return
        self.settle(y, t, inp, p, kind + " axis %d" % c, trust_u=True)
        self.last_impact = (p, c, t)

    def after_step(self, y, t, inp, exclude=(-1, -1)):
        for p in range(3):
            if self.sat[p] and self.s_of(y, p) < -1e-12:
                self.settle(y, t, inp, p, "inside the limit behind a step", trust_u=True)


class DevHybrid(HybridMicro):
    """HybridMicro in the SHAPE the device takes (csrc: the 23-state tail), so that what the kernel does has a host statement:
      * no operator tables: the only stiff blocks are the closed current loops, per AXIS a real 2 x 2 pair (i_c, u_c) with
        L2_c = [[a_c, g K], [-wp_c, -wp]] taken once at the operating point (complex-conjugate eigenvalues -1e5 +- 5.9e5 j); every
        phi_k(h L2) = c1 I + c2 L2 from ONE complex exponential -- exact for ANY h, which is what locating an event inside a step
        needs; everything else (the cross coupling of the axes at 7 /s, frozen axes, the outer loops) sits in the non-stiff remainder;
      * an event is located on the cubic through the event function's values and slopes at the two ends of the step (both come with
        the right-hand sides the scheme evaluates anyway), the step is re-taken to that instant, ONE secant correction;
      * modes and `sat` are re-derived from the state at the start of every env step (no discrete state between launches), the state
        rows between launches are fp32 (`round32`)."""

    rederive = True
    round32 = True
    max_locate = 2

    def __init__(self, **kw):
        super().__init__(**kw)
        self._rep = {}
        B = self.Lblk
        self.L2 = [np.array([[B[0, 0], B[0, 2]], [B[2, 0], B[2, 2]]]), np.array([[B[1, 1], B[1, 3]], [B[3, 1], B[3, 3]]])]
        self._phi2 = {}

    def phi_pair(self, h, c):
        """[(c1, c2)] for k = 0..3: phi_k(h L2_c) = c1 I + c2 L2_c"""
        key = (h, c)
        if key not in self._phi2:
            L = self.L2[c]
            tr, det = L[0, 0] + L[1, 1], L[0, 0] * L[1, 1] - L[0, 1] * L[1, 0]
            al, be = 0.5 * tr, np.sqrt(det - 0.25 * tr * tr)
            z = h * complex(al, be)
            ph = [np.exp(z)]
            fact = 1.0
            for k in range(3):
                ph.append((ph[-1] - 1.0 / fact) / z)
                fact *= (k + 1)
            out = []
            for v in ph:
                c2 = v.imag / be
                out.append((v.real - c2 * al, c2))
            self._phi2[key] = out
        return self._phi2[key]

    def etd(self, y, t, h, inp, modes, xmask):
        pairs = [(p, c) for p in range(3) for c in range(2) if not (modes[p] >> c) & 1]
        Ly = lambda v: self._Lv(v, pairs)

        def phi(k, hh, v):
            """phi_k(hh L) v"""
            out = v * (1.0, 1.0, 0.5, 1.0 / 6.0)[k]
            for (p, c) in pairs:
                c1, c2 = self.phi_pair(hh, c)[k]
                i0, i1 = 6 * p + c, 6 * p + 4 + c
                L = self.L2[c]
                out[i0] = c1 * v[i0] + c2 * (L[0, 0] * v[i0] + L[0, 1] * v[i1])
                out[i1] = c1 * v[i1] + c2 * (L[1, 0] * v[i0] + L[1, 1] * v[i1])
            return out

        def F(yy, tt):
            f = self.f_free(yy, tt, inp)
            for p, md in enumerate(modes):
                if md & 1: f[6 * p + 4] = 0
                if md & 2: f[6 * p + 5] = 0
                if xmask[p][0]: f[6 * p + 2] = 0
                if xmask[p][1]: f[6 * p + 3] = 0
            return f
        N = lambda yy, tt: F(yy, tt) - Ly(yy)
        h2 = h / 2
        N1 = N(y, t)
        e2y = phi(0, h2, y)
        N2 = N(e2y + h * 0.5 * phi(1, h2, N1), t + h2)
        N3 = N(e2y + h * (0.5 * phi(1, h2, N1) - phi(2, h2, N1) + phi(2, h2, N2)), t + h2)
        N4 = N(phi(0, h, y) + h * (phi(1, h, N1) - 2 * phi(2, h, N1) + phi(2, h, N2 + N3)), t + h)
        # a52 = 0.5 g2 - f3 + 0.25 f2 - 0.5 g3; a54 = 0.25 g2 - a52; a51 = 0.5 g1 - 2 a52 - a54
        a52 = lambda v: 0.5 * phi(2, h2, v) - phi(3, h, v) + 0.25 * phi(2, h, v) - 0.5 * phi(3, h2, v)
        a54 = lambda v: 0.25 * phi(2, h2, v) - a52(v)
        a51 = lambda v: 0.5 * phi(1, h2, v) - 2 * a52(v) - a54(v)
        N5 = N(e2y + h * (a51(N1) + a52(N2 + N3) + a54(N4)), t + h2)
        yn = phi(0, h, y) + h * (phi(1, h, N1) - 3 * phi(2, h, N1) + 4 * phi(3, h, N1) - phi(2, h, N4) + 4 * phi(3, h, N4)
                                  + 4 * phi(2, h, N5) - 8 * phi(3, h, N5))
        for p, md in enumerate(modes):
            if md & 1: yn[6 * p + 4] = y[6 * p + 4]
            if md & 2: yn[6 * p + 5] = y[6 * p + 5]
            if xmask[p][0]: yn[6 * p + 2] = y[6 * p + 2]
            if xmask[p][1]: yn[6 * p + 3] = y[6 * p + 3]
        return yn

    def _Lv(self, v, pairs):
        out = np.zeros_like(v)
        for (p, c) in pairs:
            i0, i1 = 6 * p + c, 6 * p + 4 + c
            L = self.L2[c]
            out[i0] = L[0, 0] * v[i0] + L[0, 1] * v[i1]
            out[i1] = L[1, 0] * v[i0] + L[1, 1] * v[i1]
        return out


def ramp_test(env=1, steps=160, quiet=1, verbose=False, cls=None):
    from switched_proto import run_ramp
    R = (cls or DevHybrid)(verbose=verbose)
    R.step_sw = lambda y, t, h, inp: R.advance(y, t, h, inp)
    err = run_ramp(R, env=env, steps=steps, quiet=quiet)
    return R, err


def load_ck(e=1):
    import pickle
    return pickle.load(open(os.path.join(ROOT, ".scratch", "sub_e%d.pkl" % e), "rb"))


def window_test(k0, s0, k1, s1, nsteps_per_sub=1, verbose=True, e=int(os.environ.get("ENV", "1"))):
    """start from the golden state at (env step k0, sim step s0) with modes from the oracle's regions, integrate to (k1, s1), compare"""
    from oracle import bridge
    ck = load_ck(e)
    R = HybridMicro(verbose=verbose)
    y_d, reg, inp = ck[(k0, s0)]
    t = (15 * k0 + s0) * DT
    y = bridge.to_wte(y_d, t)
    f, D = R.demand(y, t, inp)
    for p in range(3):
        if reg[p] != 0:
            x, u = R.xu(y, p)
            md = 0
            for c in range(2):
                if D[p][c] * u[c] > 0 or u[c] == 0.0: md |= 1 << c
            R.modes[p], R.sat[p] = md, True
    print("start modes", R.modes, "sat", R.sat, "regions", reg)
    k, s = k0, s0
    while (k, s) != (k1, s1):
        y_d, reg, inp = ck[(k, s)]
        t = (15 * k + s) * DT
        for j in range(nsteps_per_sub):
            y = R.advance(y, t + j * DT / nsteps_per_sub, DT / nsteps_per_sub, inp)
        s += 1
        if s == 15: k, s = k + 1, 0
        if (k, s) in ck:
            want = bridge.to_wte(ck[(k, s)][0], (15 * k + s) * DT)
            d = y - want
            wi = np.abs(d[[0, 1, 6, 7, 12, 13]]).max()
            if verbose:
                print("      per phase |di| %s  K du %s" % (" ".join("(%.1e,%.1e)" % (abs(d[6 * p]), abs(d[6 * p + 1])) for p in range(3)),
                                                           " ".join("(%+.1e,%+.1e)" % (R.K * d[6 * p + 4], R.K * d[6 * p + 5]) for p in range(3))))
            print("(%d,%2d) modes %s sat %s oracle %s  max|di| %.2e  dVdc %.2e  max|K du| %.2e  |dx| %.2e" % (
                k, s, R.modes, [int(b) for b in R.sat], ck[(k, s)][1], wi, abs(d[18]), R.K * np.abs(d[[4, 5, 10, 11, 16, 17]]).max(),
                np.abs(d[[2, 3, 8, 9, 14, 15]]).max()), flush=True)
    print(R.ev)
    return R


if __name__ == "__main__" and sys.argv[1:2] == ["ramp"]:
    e = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    R, err = ramp_test(env=e, quiet=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    print("env %d: worst env step by 5-step block from 106: %s" % (e, " ".join("%.1e" % err[k:k + 5].max() for k in range(106, 160, 5))))
    print("max %.2e  median beyond 116 %.2e   events %s" % (err.max(), np.median(err[116:]), R.ev))
elif __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [116, 4, 117, 0]
    window_test(*a, nsteps_per_sub=int(sys.argv[5]) if len(sys.argv) > 5 else 1)
