#!/usr/bin/env python
"""CPU fp64 prototype: per-env parameter sets (domain randomisation, tests/bridge_cases.py lane_params: C x 0.7-1.4, outer PI gains
x 0.8-1.25, 10 / 11 strings, insolation scale 0.9-1.0) on the ETDLIN path -- the exponential operator of the GROUP's nominal
parameters (L shared, one coefficient block), the env's own constants only in the right-hand side f (f-form: the step evaluates f, the
splitting f = L y + N(y) holds for any L) -- against the converged bridge with a model built from each env's own config block
(tests/golden/bridge_lane_params.npz).  The device today sends such a batch through the 23-state ETDRK4 lane kernels: 48-78 us per
batched step against 7.8 for etdlin.
    python tools/lane_etdlin_proto.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
from gym_PVDER_amd import der_model, etd_linear as el  # noqa: E402
import bridge_cases  # noqa: E402

name = "lane_params"
g = bridge_cases.load(name)
sets = bridge_cases.lane_param_sets(name)
N = g["obs"].shape[1]
par0 = der_model.param_vector("50")
L0, _ = el.jacobian(par0, "50")
n = 15
tick = 2 / 1920.0
P1 = tick * el.phis(L0, tick, 1)[1]


def observe(c, y, vg, si, q, v):
    i = complex(y[0], y[1]); va = complex(c["GR"], c["GI"]) * vg + complex(c["kviR"], c["kviI"]) * i
    S = 0.5 * c["PH"] * va * i.conjugate()
    ppv = max(0.0, y[6] * (c["pvA"] * si + c["pvB"] - c["pvB"] * np.exp(c["pvK"] * y[6])))
    return np.array([i.real, i.imag, va.real, va.imag, S.real, S.imag, y[6], ppv, v, q])


MK = {}


def run(e, own_L, scheme):
    ov = {k: float(sets[k][e]) for k in ("C_actual", "Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q", "Np")}
    par = der_model.param_vector("50", overrides=ov)
    scale = float(sets["Sinsol_scale"][e])
    c = el.folded(par)
    par_ss = par.copy(); par_ss[der_model.IDX["SINSOL0"]] *= scale
    y0f = der_model.steady_state(par_ss, "50")
    y = np.concatenate([y0f[:6], y0f[18:]])
    L = el.jacobian(par, "50")[0] if own_L else L0
    Pt = tick * el.phis(L, tick, 1)[1]
    Mk = el.k4_f_matrices(L, el.quiet_step_length(n) if hasattr(el, 'quiet_step_length') else n / 60.0)
    M2 = el.etd3rk_f_matrices(L, (n / 60.0 - tick) / 2)
    vgt, sit, acts = g["vgrid"][:, e], g["sinsol"][:, e] * scale, g["actions"][:, e]
    K = len(vgt) - 1
    rowof = lambda gk: 0 if gk < 60 else min((gk - 60) // 60 + 1, K)
    q, v = 0.0, par[der_model.IDX["VDC_REF0"]]
    dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
    ksub, out = 0, []
    for t in range(acts.shape[0]):
        a = int(acts[t])
        q += dq * ((a == 1) - (a == 2)); v += dv * ((a == 3) - (a == 4))
        row = rowof(ksub)
        inp = (vgt[row], sit[row], q, v)
        at_event = ksub >= 60 and (ksub - 60) % 60 == 0 and (ksub - 60) // 60 < K
        if scheme.startswith("k4x"):
            nk = int(scheme[3:])
            Mh = MK.setdefault((own_L if own_L else 0, e if own_L else -1, nk), el.k4_f_matrices(L, (n / 60.0 - (tick if at_event else 0.0)) / nk)) if not at_event else el.k4_f_matrices(L, (n / 60.0 - tick) / nk)
            if at_event:
                y = el.euler_tick(c, Pt, y, inp)
            for _ in range(nk):
                y, _ = el.k4_step(c, Mh, y, inp)
        elif scheme == "device" and not at_event:
            y, _ = el.k4_step(c, Mk, y, inp)
        else:
            y = el.euler_tick(c, Pt, y, inp)
            for _ in range(2):
                y = el.etd3rk_step(c, M2, y, inp)[0]
        ksub += n
        r = rowof(ksub)
        out.append(observe(c, y, vgt[r], sit[r], q, v))
    return np.array(out)


for own_L, scheme in ((False, "device"), (True, "device"), (False, "tick+2"), (False, "k4x2"), (False, "k4x3"), (False, "k4x4")):
    worst, traj = [], []
    for e in range(N):
        got = run(e, own_L, scheme)
        w = g["obs"][:, e, :10]
        worst.append(float((np.linalg.norm(got - w, axis=1) / np.linalg.norm(w, axis=1)).max()))
        traj.append(float(np.linalg.norm(got - w) / np.linalg.norm(w)))
    worst = np.array(worst)
    print("operator of %s parameters, schedule %-7s: worst env step over %d sets: max %.2e median %.2e  #>1e-4: %d   trajectory max %.2e"
          % ("the env's OWN" if own_L else "the GROUP's  ", scheme, N, worst.max(), np.median(worst), int((worst > 1e-4).sum()), max(traj)), flush=True)
