#!/usr/bin/env python
"""fp64 prototype (CPU, numpy): cheaper forms of the careful pass (envs at the inverter's CURRENT limit: the reference's
fixed-action Q_ref ramps, calc_returns, PVDER_env.py:480-487) against the committed converged-bridge samples
(tests/golden/bridge_ramp_a{1,2}_sample.npz, 48 envs each).

The device's careful pass (csrc lin_integrate_careful, etd_linear.hybrid_step) takes a tick and CAREFUL_STEPS = 8 equal
ETD3RK steps per env step, each up to THREE evaluations (free or frozen; the crossing located; sliding): 1.9 us per
evaluation and wave (one wave per SIMD, latency bound), and a wave pays for the worst of its 64 lanes -- 45 us per env step
once the whole batch sits at the limit (profiles/r05_careful_counters.txt).  Schemes compared here, per step count:

    hybrid   the shipped step (etd_linear.hybrid_step)
    hybrid2  the shipped step without its third evaluation: a step that should slide ends on s = 0 of the affine combination of
             the two evaluations it has
    affine   TWO evaluations, alpha = 0 and alpha = 1, for a step that crosses or slides; the state at any alpha in between
             is their affine combination -- crossing fraction and the slide onto |iref| = iref_limit are then scalar root
             searches on iref2, no further evaluation
    sens     ONE evaluation; y+(alpha) from the LINEAR response h phi1(hL) g to the integrators' forcing
    eqctl    ONE evaluation: alpha per STAGE from Filippov's equivalent control with a Baumgarte term,
             alpha = clamp(-(A + gamma s) / B, 0, 1),  ds/dt = A + alpha B,  gamma = 1 / h
    warm1/2  the shipped step, but a step behind one that SLID starts from the alpha that one ended with: one evaluation and
             the last step's dy+/dalpha (warm1), or two evaluations one secant step apart (warm2)
    slide    warm1 / warm2 as the device has them (etd_linear.hybrid_step with a carry; gamma*h column = the threshold on the
             alpha the one-evaluation form may move by)
    warmeq   ... or the equivalent control stage by stage, then onto s = 0 along the last step's dy+/dalpha
Result (profiles/r05_careful_schemes_proto.txt): only warm2 keeps the shipped step's accuracy (7e-5 worst env step); every
one-evaluation form and both two-evaluation forms of the FIRST sliding step lose a factor 3-30.

    python tools/careful_schemes_proto.py [n_envs]      -> profiles/r05_careful_schemes_proto.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
from gym_PVDER_amd import der_model, etd_linear as el  # noqa: E402
import bridge_cases  # noqa: E402

par = der_model.param_vector("50")
c = el.folded(par)
L, _ = el.jacobian(par, "50")
ILIM2 = par[der_model.IDX["IREF_LIMIT"]] ** 2
VDC0 = par[der_model.IDX["VDC_REF0"]]
_cache = {}
EVALS = [0]


def mats(h, k4=False):
    k = (round(h * 1e13), k4)
    if k not in _cache:
        _cache[k] = el.k4_f_matrices(L, h) if k4 else el.etd3rk_f_matrices(L, h)
    return _cache[k]


def observe(y, vg, si, q, v):
    i = complex(y[0], y[1]); va = complex(c["GR"], c["GI"]) * vg + complex(c["kviR"], c["kviI"]) * i
    S = 0.5 * c["PH"] * va * i.conjugate()
    ppv = max(0.0, y[6] * (c["pvA"] * si + c["pvB"] - c["pvB"] * np.exp(c["pvK"] * y[6])))
    return np.array([i.real, i.imag, va.real, va.imag, S.real, S.imag, y[6], ppv, v, q])


def s_of(y, inp):
    return el.iref2(c, y, inp[0], inp[2], inp[3]) - ILIM2


def etd3(y, M, inp, alpha):
    """one ETD3RK step with the outward-moving integrators at alpha times their rate; frozen ones held (HOLD_FROZEN)"""
    EVALS[0] += 1
    F1 = el.f_bal_limited(c, y, inp, alpha, ILIM2); a = y + M[0] @ F1
    F2 = el.f_bal_limited(c, a, inp, alpha, ILIM2); b = y + M[1] @ F1 + M[2] @ F2
    F3 = el.f_bal_limited(c, b, inp, alpha, ILIM2)
    yn = y + M[3] @ F1 + M[4] @ F2 + M[5] @ F3
    if alpha == 0.0:
        for k in (7, 8):
            if F1[k] == 0.0 and F2[k] == 0.0 and F3[k] == 0.0:
                yn[k] = y[k]
    return yn, s_of(a, inp)


def affine_step(y, M, inp):
    s0 = s_of(y, inp)
    free = s0 <= 0
    a1 = 1.0 if free else 0.0
    y1, sm = etd3(y, M, inp, a1)
    s1 = s_of(y1, inp)
    if (s1 > 0) != free or s1 == 0:
        return y1
    yo, _ = etd3(y, M, inp, 1.0 - a1)
    y_at = (lambda al: yo + al * (y1 - yo)) if free else (lambda al: y1 + al * (yo - y1))     # y+(alpha), alpha = share of the step spent free
    tau = el.cross_fraction(s0, sm, s1)
    a2 = tau if free else 1.0 - tau
    s2 = s_of(y_at(a2), inp)
    if (s2 < 0) != free or s2 == 0:
        return y_at(a2)
    # sliding: the alpha with s(y+(alpha)) = 0 between a2 and a1 -- scalar secant steps on iref2, no evaluation
    lo, slo, hi, shi = a2, s2, a1, s1
    al = a2
    for _ in range(6):
        al = lo - slo * (hi - lo) / (shi - slo)
        sa = s_of(y_at(al), inp)
        if abs(sa) < 1e-14:
            break
        if (sa > 0) == (shi > 0):
            hi, shi = al, sa
        else:
            lo, slo = al, sa
    return y_at(al)


def hybrid2_step(y, M, inp):
    """the shipped step without its third evaluation: free / frozen; the crossing located (second evaluation at the share tau of
    the step spent free); a step that should SLIDE ends on s = 0 of the affine combination of the two evaluations it has"""
    s0 = s_of(y, inp)
    free = s0 <= 0
    a1 = 1.0 if free else 0.0
    y1, sm = etd3(y, M, inp, a1)
    s1 = s_of(y1, inp)
    if (s1 > 0) != free or s1 == 0:
        return y1
    tau = el.cross_fraction(s0, sm, s1)
    a2 = tau if free else 1.0 - tau
    y2, _ = etd3(y, M, inp, a2)
    s2 = s_of(y2, inp)
    if (s2 < 0) != free or s2 == 0 or a2 == a1:
        return y2
    y_at = lambda al: y2 + (al - a2) / (a1 - a2) * (y1 - y2)
    lo, slo, hi, shi = a2, s2, a1, s1
    al = a2
    for _ in range(4):
        al = lo - slo * (hi - lo) / (shi - slo)
        sa = s_of(y_at(al), inp)
        if abs(sa) < 1e-14:
            break
        if (sa > 0) == (shi > 0):
            hi, shi = al, sa
        else:
            lo, slo = al, sa
    return y_at(al)


def warm_step(y, M, inp, W, evals_warm=1):
    """the shipped step, and behind a step that SLID the next one starts from where that one ended: alpha of its first evaluation =
    the alpha the last one came to, y+(alpha) around it from the last step's dy+/dalpha (evals_warm = 1) or from a second
    evaluation one secant step further (evals_warm = 2); W: what an env carries from one step of an env step to the next"""
    s0 = s_of(y, inp)
    if W.get("alpha") is not None and evals_warm == 0:
        # a step that starts ON the limit, sliding: the equivalent control stage by stage (gamma: Baumgarte term), ONE evaluation,
        # then onto s = 0 along the last step's dy+/dalpha
        ye = eqctl_step(y, M, inp, W["h"], W.get("gs", 0.0))
        D = W["D"]
        al = 0.0
        for _ in range(4):
            f0 = s_of(ye + al * D, inp)
            f1 = s_of(ye + (al + 1e-4) * D, inp)
            d = (f1 - f0) / 1e-4
            if d == 0:
                break
            al -= f0 / d
        if abs(al) < 0.25:
            return ye + al * D
        W["alpha"] = None
    if W.get("alpha") is not None:
        a = W["alpha"]
        ya, _ = etd3(y, M, inp, a)
        sa = s_of(ya, inp)
        D = W["D"]
        if evals_warm == 3:
            # adaptive: the last step's dy+/dalpha if the polish moves alpha by less than thr, the second evaluation otherwise
            eps = 1e-3
            ds = (s_of(ya + eps * D, inp) - sa) / eps
            evals_here = 1 if (ds != 0 and abs(sa / ds) < W.get("thr", 0.03)) else 2
            W["n1"] = W.get("n1", 0) + (evals_here == 1); W["n2"] = W.get("n2", 0) + (evals_here == 2)
        else:
            evals_here = evals_warm
        if evals_here == 2:
            ds = None
            # slope ds/dalpha from the last step's derivative of the STATE: one scalar evaluation of iref2
            eps = 1e-3
            ds = (s_of(ya + eps * D, inp) - sa) / eps
            b = a - sa / ds if ds != 0 else a
            b = min(1.0, max(0.0, b))
            if b != a:
                yb, _ = etd3(y, M, inp, b)
                D = (yb - ya) / (b - a)
                ya, a, sa = yb, b, s_of(yb, inp)
        # the alpha with s(ya + (alpha - a) D) = 0: Newton steps on the scalar function
        al = a
        for _ in range(4):
            f0 = s_of(ya + (al - a) * D, inp)
            f1 = s_of(ya + (al - a + 1e-4) * D, inp)
            d = (f1 - f0) / 1e-4
            if d == 0:
                break
            al -= f0 / d
        if 0.0 < al < 1.0 and abs(al - a) < 0.25:
            W["alpha"] = al
            W["D"] = D
            return ya + (al - a) * D
        W["alpha"] = None                      # it left the limit (or the guess was poor): the step again, as a first one
    free = s0 <= 0
    a1 = 1.0 if free else 0.0
    y1, sm = etd3(y, M, inp, a1)
    s1 = s_of(y1, inp)
    if (s1 > 0) != free or s1 == 0:
        return y1
    tau = el.cross_fraction(s0, sm, s1)
    a2 = tau if free else 1.0 - tau
    y2, _ = etd3(y, M, inp, a2)
    s2 = s_of(y2, inp)
    if (s2 < 0) != free or s2 == 0:
        return y2
    a3 = a2 - s2 * (1.0 - a2) / (s1 - s2) if free else a2 * s1 / (s1 - s2)
    y3, _ = etd3(y, M, inp, a3)
    s3 = s_of(y3, inp)
    if s3 != s2 and a3 != a2:
        w = -s3 / (s3 - s2)
        if abs(w) <= 1.0:
            W["alpha"] = a3 + w * (a3 - a2)
            W["D"] = (y3 - y2) / (a3 - a2)
            y3 = y3 + w * (y3 - y2)
    return y3


def slide_step(y, M, inp, W, thr=0.03, dtype=np.float64):
    """The step as the device takes it from round 5 on (etd_linear.hybrid_step with its `carry`): a FIRST sliding step as shipped
    (three evaluations); a step behind one that slid starts at the alpha that one ended with -- ONE evaluation, then along the
    last step's dy+/dalpha onto s = 0 (s is quadratic in the state: its directional derivative from s(y + D), s(y - D)) when that
    moves alpha by less than `thr`, otherwise a second evaluation one Newton step further and the secant through the two"""
    return el.hybrid_step(c, M, y, inp, ILIM2, carry=W, slide_thr=thr)


def sens_step(y, M, inp, PH1, weights=(1.0 / 6, 4.0 / 6, 1.0 / 6)):
    """ONE evaluation; what another alpha would have given from the LINEAR response to the integrators' forcing:
    y+(alpha) = y+(a1) + (alpha - a1) h phi1(hL) g,  g = the outward-moving integrators' free rate averaged over the stages"""
    s0 = s_of(y, inp)
    free = s0 <= 0
    a1 = 1.0 if free else 0.0
    EVALS[0] += 1
    Fs, Gs = [], []
    def F(u):
        f = el.f_bal(c, u, *inp)
        g = np.zeros(11)
        for k in (7, 8):
            if f[k] * u[k] > 0:
                g[k] = f[k]
                f[k] = a1 * f[k]
        Gs.append(g)
        return f
    F1 = F(y); a = y + M[0] @ F1
    F2 = F(a); b = y + M[1] @ F1 + M[2] @ F2
    F3 = F(b)
    y1 = y + M[3] @ F1 + M[4] @ F2 + M[5] @ F3
    if a1 == 0.0:
        for k in (7, 8):
            if F1[k] == 0.0 and F2[k] == 0.0 and F3[k] == 0.0:
                y1[k] = y[k]
    sm, s1 = s_of(a, inp), s_of(y1, inp)
    if (s1 > 0) != free or s1 == 0:
        return y1
    D = PH1 @ (weights[0] * Gs[0] + weights[1] * Gs[1] + weights[2] * Gs[2])
    y_at = lambda al: y1 + (al - a1) * D
    tau = el.cross_fraction(s0, sm, s1)
    a2 = tau if free else 1.0 - tau
    s2 = s_of(y_at(a2), inp)
    if (s2 < 0) != free or s2 == 0:
        return y_at(a2)
    lo, slo, hi, shi = a2, s2, a1, s1
    al = a2
    for _ in range(6):
        al = lo - slo * (hi - lo) / (shi - slo)
        sa = s_of(y_at(al), inp)
        if abs(sa) < 1e-14:
            break
        if (sa > 0) == (shi > 0):
            hi, shi = al, sa
        else:
            lo, slo = al, sa
    return y_at(al)


def f_eqctl(y, inp, gamma):
    """f with the outward-moving integrators at the equivalent control's rate, alpha = clamp(-(A + gamma s) / B, 0, 1)"""
    vg, si, q, v = inp
    f = el.f_bal(c, y, *inp)
    iR, iI = y[0], y[1]
    aR, aI = c["GR"] * vg, c["GI"] * vg
    Q = 0.5 * c["PH"] * (aI * iR - aR * iI + c["kviI"] * (iR * iR + iI * iI))
    rR = y[7] + c["KpDC"] * (v - y[6])
    rI = y[8] - c["KpQ"] * (q - Q)
    s = rR * rR + rI * rI - ILIM2
    Qd = 0.5 * c["PH"] * ((aI + 2 * c["kviI"] * iR) * f[0] + (-aR + 2 * c["kviI"] * iI) * f[1])
    out7, out8 = f[7] * y[7] > 0, f[8] * y[8] > 0
    A = 2 * (rR * (-c["KpDC"] * f[6] + (0.0 if out7 else f[7])) + rI * (c["KpQ"] * Qd + (0.0 if out8 else f[8])))
    B = 2 * (rR * (f[7] if out7 else 0.0) + rI * (f[8] if out8 else 0.0))
    if B > 0:
        alpha = min(1.0, max(0.0, -(A + gamma * s) / B))
    else:
        alpha = 0.0 if s > 0 else 1.0
    if out7:
        f[7] *= alpha
    if out8:
        f[8] *= alpha
    return f


def eqctl_step(y, M, inp, h, gscale, k4=False):
    EVALS[0] += 1
    g = gscale / h
    F1 = f_eqctl(y, inp, g); a = y + M[0] @ F1
    F2 = f_eqctl(a, inp, g); b = y + M[1] @ F1 + M[2] @ F2
    F3 = f_eqctl(b, inp, g)
    if not k4:
        return y + M[3] @ F1 + M[4] @ F2 + M[5] @ F3
    u4 = y + M[3] @ F1 + M[4] @ F2 + M[5] @ F3
    F4 = f_eqctl(u4, inp, g)
    return y + M[6] @ F1 + M[7] @ F2 + M[8] @ F3 + M[9] @ F4


def ramp(action, vgt, sit, steps, nst, mode, n=15, gscale=1.0):
    y0f = der_model.steady_state(par, "50")
    y = np.concatenate([y0f[:6], y0f[18:]])
    q, v = 0.0, VDC0
    dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
    K = len(vgt) - 1
    rowof = lambda gk: 0 if gk < 60 else min((gk - 60) // 60 + 1, K)
    tick = 2 / 1920.0
    P1 = tick * el.phis(L, tick, 1)[1]
    ksub, out = 0, []
    h = (n / 60.0 - tick) / nst
    k4 = mode.endswith("k4")
    M = mats(h, k4)
    PH1 = h * el.phis(L, h, 1)[1]
    for t in range(steps):
        q += dq * ((action == 1) - (action == 2)); v += dv * ((action == 3) - (action == 4))
        row = rowof(ksub)
        inp = (vgt[row], sit[row], q, v)
        y = el.careful_tick(c, P1, y, inp, ILIM2)
        W = {"h": h, "gs": gscale}
        for _ in range(nst):
            if mode == "hybrid":
                y, ne = el.hybrid_step(c, M, y, inp, ILIM2); EVALS[0] += ne
            elif mode == "affine":
                y = affine_step(y, M, inp)
            elif mode == "warmeq":
                y = warm_step(y, M, inp, W, 0)
            elif mode == "slide":
                y, ne = slide_step(y, M, inp, W, gscale); EVALS[0] += ne
            elif mode == "warm12":
                W["thr"] = gscale
                y = warm_step(y, M, inp, W, 3)
            elif mode == "warm1":
                y = warm_step(y, M, inp, W, 1)
            elif mode == "warm2":
                y = warm_step(y, M, inp, W, 2)
            elif mode == "hybrid2":
                y = hybrid2_step(y, M, inp)
            elif mode == "sens":
                y = sens_step(y, M, inp, PH1)
            else:
                y = eqctl_step(y, M, inp, h, gscale, k4)
        ksub += n
        r = rowof(ksub)
        out.append(observe(y, vgt[r], sit[r], q, v))
    return np.array(out)


def main():
    n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    lines = []
    for name, a in (("ramp_a1_sample", 1), ("ramp_a2_sample", 2)):
        g = bridge_cases.load(name)
        envs = list(range(0, g["obs"].shape[1], max(1, g["obs"].shape[1] // n_envs)))[:n_envs]
        for mode, counts, gs in (("hybrid", (8, 6, 4), 1.0), ("slide", (8,), 0.03), ("slide", (8,), 0.1), ("hybrid2", (8,), 1.0), ("affine", (8,), 1.0), ("sens", (8,), 1.0), ("eqctl", (8,), 1.0),
                                 ("warm1", (8,), 1.0), ("warm2", (8, 6), 1.0), ("warmeq", (8,), 0.0)):
            for nst in counts:
                worst, traj = [], 0.0
                EVALS[0] = 0
                for e in envs:
                    got = ramp(a, g["vgrid"][:, e], g["sinsol"][:, e], 160, nst, mode, gscale=gs)
                    w = g["obs"][:, e, :10]
                    per = np.linalg.norm(got - w, axis=1) / np.linalg.norm(w, axis=1)
                    worst.append(float(per[100:].max()))
                    traj = max(traj, float(np.linalg.norm(got - w) / np.linalg.norm(w)))
                worst = np.array(worst)
                ln = ("%s %-9s gamma*h %.1f  %d steps: worst env step (steps 100..159) over %d envs: max %.2e median %.2e  #>1e-4: %d   "
                      "trajectory max %.2e   evaluations per env step %.1f"
                      % (name, mode, gs, nst, len(envs), worst.max(), np.median(worst), int((worst > 1e-4).sum()), traj,
                         EVALS[0] / (160.0 * len(envs))))
                print(ln, flush=True)
                lines.append(ln)
    with open(os.path.join(ROOT, "profiles", "r05_careful_schemes_proto.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
