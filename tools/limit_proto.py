#!/usr/bin/env python
"""fp64 prototypes (CPU, numpy) behind round 3's limit handling -- what was measured before the kernels were written.

    python tools/limit_proto.py current      the reference's fixed-action Q_ref ramps (calc_returns holds ONE action,
                                             PVDER_env.py:480-487) into the inverter's current limit: chattering explicit
                                             treatment of the conditional integration vs the hybrid steps of the careful
                                             pass (etd_linear.hybrid_step), per step count, against the committed
                                             converged-bridge goldens (tests/golden/bridge_ramp_a{1,2}.npz)
    python tools/limit_proto.py floor        the Vdc_ref ramp into the duty-cycle limit: how far fp64 integrators agree
                                             with each other once the duty cycle saturates (RK4 at two step sizes, RK45
                                             at two tolerances vs the golden's RK45 at 1e-8) -- minutes of CPU

Output of both: profiles/r03_limit_proto_current.txt, profiles/r03_saturation_floor.txt."""
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


def mats(h):
    k = round(h * 1e13)
    if k not in _cache:
        _cache[k] = el.etd3rk_f_matrices(L, h)
    return _cache[k]


def observe(y, vg, si, q, v):
    i = complex(y[0], y[1]); va = complex(c["GR"], c["GI"]) * vg + complex(c["kviR"], c["kviI"]) * i
    S = 0.5 * c["PH"] * va * i.conjugate()
    ppv = max(0.0, y[6] * (c["pvA"] * si + c["pvB"] - c["pvB"] * np.exp(c["pvK"] * y[6])))
    return np.array([i.real, i.imag, va.real, va.imag, S.real, S.imag, y[6], ppv, v, q])


def ramp(action, vgt, sit, steps, nst, mode, n=15):
    """balanced 11-state etdlin along a fixed-action ramp: tick + nst equal steps per env step;
    mode 'chatter': the model's sign tests evaluated stage by stage; 'hybrid': el.hybrid_step"""
    y0f = der_model.steady_state(par, "50")
    y = np.concatenate([y0f[:6], y0f[18:]])
    q, v = 0.0, VDC0
    dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
    K = len(vgt) - 1
    rowof = lambda gk: 0 if gk < 60 else min((gk - 60) // 60 + 1, K)
    tick = 2 / 1920.0
    P1 = tick * el.phis(L, tick, 1)[1]
    ksub, out = 0, []
    for t in range(steps):
        q += dq * ((action == 1) - (action == 2)); v += dv * ((action == 3) - (action == 4))
        row = rowof(ksub)
        inp = (vgt[row], sit[row], q, v)
        y = el.careful_tick(c, P1, y, inp, ILIM2)
        M = mats((n / 60.0 - tick) / nst)
        for _ in range(nst):
            if mode == "hybrid":
                y, _ = el.hybrid_step(c, M, y, inp, ILIM2)
            else:
                F1 = el.f_bal_limited(c, y, inp, -1.0, ILIM2); a = y + M[0] @ F1
                F2 = el.f_bal_limited(c, a, inp, -1.0, ILIM2); b = y + M[1] @ F1 + M[2] @ F2
                F3 = el.f_bal_limited(c, b, inp, -1.0, ILIM2)
                y = y + M[3] @ F1 + M[4] @ F2 + M[5] @ F3
        ksub += n
        r = rowof(ksub)
        out.append(observe(y, vgt[r], sit[r], q, v))
    return np.array(out)


def current():
    for name, a in (("ramp_a1", 1), ("ramp_a2", 2)):
        g = bridge_cases.load(name)
        for mode, counts in (("chatter", (2, 15, 60)), ("hybrid", (2, 4, 8))):
            for nst in counts:
                worst, traj = 0.0, 0.0
                for e in range(g["obs"].shape[1]):
                    got = ramp(a, g["vgrid"][:, e], g["sinsol"][:, e], 160, nst, mode)
                    w = g["obs"][:, e, :10]
                    worst = max(worst, float((np.linalg.norm(got - w, axis=1) / np.linalg.norm(w, axis=1)).max()))
                    traj = max(traj, float(np.linalg.norm(got - w) / np.linalg.norm(w)))
                print("%s (4 envs, 160 steps) %-8s %2d steps per env step: worst env step %.2e, whole trajectory %.2e"
                      % (name, mode, nst, worst, traj), flush=True)


def floor():
    from oracle.bridge_env import BridgeVecEnv
    g = bridge_cases.load("ramp_a4")
    e = int(np.argmin([int(np.argmax(g["m_abs"][:, k] > 0.9995)) for k in range(4)]))
    first = int(np.argmax(g["m_abs"][:, e] > 0.9995))
    print("ramp_a4 env %d: the duty cycle first saturates in env step %d" % (e, first))
    want = g["obs"][:, e, :10]
    for label, kw in (("RK4 4096 micro-steps", dict(method="rk4", micro=4096)), ("RK4 8192 micro-steps", dict(method="rk4", micro=8192)),
                      ("RK45 rtol 1e-6", dict(method="rk45", rtol=1e-6, atol=1e-9)), ("RK45 rtol 1e-10", dict(method="rk45", rtol=1e-10, atol=1e-13))):
        ref = BridgeVecEnv(1, n=15, **kw)
        ref.reset(g["vgrid"][:, e:e + 1], g["sinsol"][:, e:e + 1], ev_s0=60, ev_ds=60)
        got = np.array([ref.step(np.full(1, 4))[0][0, :10].copy() for _ in range(160)])
        per = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
        print("%-22s vs RK45 rtol 1e-8, worst env step by 10-step block from %d: " % (label, first - 10)
              + " ".join("%.1e" % per[k:k + 10].max() for k in range(first - 10, 160, 10)), flush=True)


if __name__ == "__main__":
    {"current": current, "floor": floor}[sys.argv[1] if len(sys.argv) > 1 else "current"]()
