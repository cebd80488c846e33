#!/usr/bin/env python
"""CPU fp64: how far per-env parameter sets may sit from the group's before the etdlin lane path (the GROUP's exponential operator,
the env's own constants in the right-hand side, N Krogstad steps per env step) leaves 1e-4 of the converged per-set bridge --
spreads wider than the committed golden's (tools/lane_etdlin_proto.py), bridge runs made here (not committed).
    python tools/lane_etdlin_envelope.py [n_envs] [steps]      -> profiles/r05_lane_etdlin_envelope.txt"""
import multiprocessing as mp, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
from gym_PVDER_amd import der_model, etd_linear as el  # noqa: E402
import bridge_cases  # noqa: E402

NENV = int(sys.argv[1]) if len(sys.argv) > 1 else 28
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
SPREADS = {"golden's (C 0.7-1.4, gains 0.8-1.25)": (0.7, 1.4, 0.8, 1.25), "C 0.5-2.0, gains 0.67-1.5": (0.5, 2.0, 0.67, 1.5),
           "C 0.33-3.0, gains 0.5-2.0": (0.33, 3.0, 0.5, 2.0)}
n = 15
vg_all, si_all = bridge_cases.counter_tables(NENV, 21, episode=1, spec=bridge_cases.both_events_spec())
acts_all = np.random.default_rng(99).integers(0, 5, size=(STEPS, NENV))


def sets_for(spread, seed=5):
    from oracle.pvder_model import DER_CONFIG
    clo, chi, glo, ghi = SPREADS[spread]
    rng = np.random.default_rng(seed)
    base = DER_CONFIG["50"]
    # log-uniform: as many envs below the group's value as above
    lu = lambda lo, hi: np.exp(rng.uniform(np.log(lo), np.log(hi), NENV))
    s = {"C_actual": base["C_actual"] * lu(clo, chi)}
    for k in ("Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q"):
        s[k] = base[k] * lu(glo, ghi)
    s["Np"] = rng.integers(10, 12, NENV).astype(np.float64)
    s["Sinsol_scale"] = rng.uniform(0.9, 1.0, NENV)
    return s


def bridge_run(job):
    spread, e = job
    from oracle import pvder_model as pm, bridge
    from oracle.bridge_env import BridgeVecEnv
    sets = sets_for(spread)
    ref = BridgeVecEnv(1, n=n, method="rk45", rtol=1e-8, atol=1e-11, overrides={k: float(v[e]) for k, v in sets.items() if k != "Sinsol_scale"})
    ref.reset(vg_all[:, e:e + 1], si_all[:, e:e + 1] * sets["Sinsol_scale"][e], ev_s0=60, ev_ds=60)
    ref.Y[0] = bridge.to_delta(pm.steady_state(ref.model, Sinsol=ref.model.Sinsol0 * sets["Sinsol_scale"][e]), 0.0)
    out = []
    for a in acts_all[:, e]:
        o, r, d = ref.step(np.array([a]))
        out.append(o[0, :10].copy())
    return spread, e, np.array(out)


par0 = der_model.param_vector("50")
L0, _ = el.jacobian(par0, "50")
tick = 2 / 1920.0
P1 = tick * el.phis(L0, tick, 1)[1]
MK = {}


def observe(c, y, vg, si, q, v):
    i = complex(y[0], y[1]); va = complex(c["GR"], c["GI"]) * vg + complex(c["kviR"], c["kviI"]) * i
    S = 0.5 * c["PH"] * va * i.conjugate()
    ppv = max(0.0, y[6] * (c["pvA"] * si + c["pvB"] - c["pvB"] * np.exp(c["pvK"] * y[6])))
    return np.array([i.real, i.imag, va.real, va.imag, S.real, S.imag, y[6], ppv, v, q])


def lane_run(sets, e, nk):
    ov = {k: float(sets[k][e]) for k in ("C_actual", "Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q", "Np")}
    par = der_model.param_vector("50", overrides=ov)
    scale = float(sets["Sinsol_scale"][e])
    c = el.folded(par)
    par_ss = par.copy(); par_ss[der_model.IDX["SINSOL0"]] *= scale
    y0f = der_model.steady_state(par_ss, "50")
    y = np.concatenate([y0f[:6], y0f[18:]])
    vgt, sit = vg_all[:, e], si_all[:, e] * scale
    K = len(vgt) - 1
    rowof = lambda gk: 0 if gk < 60 else min((gk - 60) // 60 + 1, K)
    q, v = 0.0, par[der_model.IDX["VDC_REF0"]]
    dq, dv = 25.0 * n / 50e3, 0.02 * n / 500.0
    ksub, out = 0, []
    for t in range(STEPS):
        a = int(acts_all[t, e])
        q += dq * ((a == 1) - (a == 2)); v += dv * ((a == 3) - (a == 4))
        inp = (vgt[rowof(ksub)], sit[rowof(ksub)], q, v)
        at_event = ksub >= 60 and (ksub - 60) % 60 == 0 and (ksub - 60) // 60 < K
        h = (n / 60.0 - (tick if at_event else 0.0)) / nk
        M = MK.setdefault((round(h * 1e12),), el.k4_f_matrices(L0, h))
        if at_event:
            y = el.euler_tick(c, P1, y, inp)
        for _ in range(nk):
            y, _ = el.k4_step(c, M, y, inp)
        ksub += n
        r = rowof(ksub)
        out.append(observe(c, y, vgt[r], sit[r], q, v))
    return np.array(out)


if __name__ == "__main__":
    jobs = [(s, e) for s in SPREADS for e in range(NENV)]
    with mp.Pool(7) as pool:
        res = pool.map(bridge_run, jobs, chunksize=1)
    lines = ["etdlin lane path (group's operator) against the converged per-set bridge: %d envs x %d env steps per spread, random actions, "
             "voltage + insolation events; worst env step (rel-L2 of the ten dynamic / set-point observations) over the envs" % (NENV, STEPS)]
    for spread in SPREADS:
        want = {e: o for s, e, o in res if s == spread}
        sets = sets_for(spread)
        row = []
        for nk in (1, 2, 3, 4, 6, 8):
            worst = []
            for e in range(NENV):
                got = lane_run(sets, e, nk)
                worst.append(float((np.linalg.norm(got - want[e], axis=1) / np.linalg.norm(want[e], axis=1)).max()))
            row.append("%d step%s: max %.1e median %.1e" % (nk, "s" if nk > 1 else " ", max(worst), float(np.median(worst))))
        lines.append("%-40s %s" % (spread, "   ".join(row)))
        print(lines[-1], flush=True)
    with open(os.path.join(ROOT, "profiles", "r05_lane_etdlin_envelope.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
