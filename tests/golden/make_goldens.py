#!/usr/bin/env python
"""Generate the BUILD-ORACLE golden vectors (tests/golden/*.npz).

These are NOT reference goldens: the reference holds none and its ODE lives in the absent,
unpinned third-party ``pvder`` package ("parity unpinned").  They are produced by
the build's own fp64 oracle so that the -m gpu tests can check the HIP kernels on the GPU box
(where nothing is regenerated) and so that the CPU tests can pin the oracle's pieces against each
other.  Every trajectory is integrated by TWO independent integrators of the same restated model,

  (A) SciPy ``odeint`` (LSODA/BDF, the reference's solver, at rtol = atol = 1e-6 -- 100x tighter than
      the reference's own 1e-4; one call per env step with the piecewise-constant inputs frozen: in
      every golden case the events fall on env-step boundaries), Python RHS
      oracle/pvder_model.py::rhs (absolute PLL angle), and
  (B) Dormand-Prince RK45 at rtol 1e-10 in C, oracle/pvder_bridge.c (relative PLL angle) -- the
      converged solution,

(B) is stored as the golden (obs, reward, done, states); (A) is stored beside it (obs_lsoda) and
the two must agree to < 2e-5 rel-L2 on the observation trajectory (achieved figure stored).

Run:  python tests/golden/make_goldens.py      (~3 min on 8 cores)
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scipy.integrate import odeint  # noqa: E402

from oracle import pvder_model as pm  # noqa: E402
from oracle.bridge_env import BridgeVecEnv  # noqa: E402
from oracle.pvder_oracle_env import DEFAULT_EVENTS_SPEC, OraclePVDER, SIM_TIME_STEP  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class TightOracle(OraclePVDER):
    """OraclePVDER with the converged-LSODA integrate() described in the module docstring."""

    def integrate(self):
        m = self.model
        t0 = self.k_sub * SIM_TIME_STEP
        ins = {self.events.at((self.k_sub + j) * SIM_TIME_STEP + 1e-7, True) for j in range(self.n)}
        assert len(ins) == 1, "golden cases keep events on env-step boundaries"
        vg, si = ins.pop()
        f = lambda yy, tau: pm.rhs(m, yy, t0 + tau, vg, si, self.Q_ref, self.Vdc_ref)
        jac = lambda yy, tau: pm.jac_fd(m, yy, t0 + tau, vg, si, self.Q_ref, self.Vdc_ref)
        # rtol = atol = 1e-6 (100x tighter than the reference's 1e-4).  At 1e-8 and below LSODA's
        # stiffness switch never fires on this system (accuracy- and stability-limited Adams steps
        # coincide) and it grinds through >5e5 Adams steps per env step.
        sol, info = odeint(f, self.y, [0.0, self.n * SIM_TIME_STEP], Dfun=jac, rtol=1e-6, atol=1e-6,
                           mxstep=500000, full_output=1)
        if info["message"] != "Integration successful.":
            raise RuntimeError(info["message"])
        self.y = sol[-1]
        self.k_sub += self.n


def action_script(name, steps):
    if name == "noop":
        return np.zeros(steps, dtype=np.int64)
    if name == "q_up":
        return np.ones(steps, dtype=np.int64)
    if name == "q_down":
        return np.full(steps, 2, dtype=np.int64)
    if name == "vdc_alt":
        return np.where(np.arange(steps) % 2 == 0, 3, 4).astype(np.int64)
    if name == "random":
        return np.random.default_rng(1234).integers(0, 5, size=steps)
    raise KeyError(name)


def events_spec(kind):
    s = {k: dict(v) for k, v in DEFAULT_EVENTS_SPEC.items()}
    s["voltage"]["ENABLE"] = kind in ("voltage", "both")
    s["insolation"]["ENABLE"] = kind == "both"
    return s


def run_case(case):
    name, script, ev_kind, n, steps, goal, discrete = case
    spec = events_spec(ev_kind)
    table = pm.make_event_table(spec, 100.0, counter_key=(777, sum(map(ord, name)) % 1000, 1))
    acts = action_script(script, steps)
    t0 = time.time()
    env = TightOracle(n_sim_time_steps_per_env_step=n, DISCRETE_REWARD=discrete, goals_list=[goal])
    obs = [env.reset(event_table=table)]
    rew, done, states = [], [], [env.y.copy()]
    for a in acts:
        o, r, d, _ = env.step(int(a))
        obs.append(o); rew.append(r); done.append(d); states.append(env.y.copy())
    # cross-check with the independent C RK45 integrator
    b = BridgeVecEnv(1, n=n, discrete=discrete, goal=goal, method="rk45", rtol=1e-10, atol=1e-12)
    ob = [b.reset(table.vgrid[:, None], table.sinsol[:, None])[0]]
    from oracle import bridge as _br
    brew, bdone, bstates = [], [], [_br.to_wte(b.Y[0], 0.0)]
    for a in acts:
        o, r, d = b.step(np.array([a]))
        ob.append(o[0]); brew.append(r[0]); bdone.append(d[0])
        bstates.append(_br.to_wte(b.Y[0], b.ksub[0] * SIM_TIME_STEP))
    A, B = np.array(obs), np.array(ob)
    agree = float(np.linalg.norm(A - B) / np.linalg.norm(B))
    out = dict(name=name, actions=acts, vgrid=table.vgrid, sinsol=table.sinsol, obs=B, obs_lsoda=A,
               reward=np.array(brew, dtype=np.float64), reward_lsoda=np.array(rew, dtype=np.float64),
               done=np.array(bdone), states=np.array(bstates), n=n, goal=goal, discrete=discrete,
               ab_rel_l2=agree, ev_t0=table.t0, ev_dt=table.dt, secs=time.time() - t0)
    return out


CASES = [
    # name, action script, events, n, env steps, goal, discrete
    ("noop_noev_n15", "noop", "none", 15, 24, "voltage_regulation", True),
    ("qup_volt_n15", "q_up", "voltage", 15, 24, "voltage_regulation", True),
    ("qdown_volt_n15", "q_down", "voltage", 15, 24, "voltage_regulation", False),
    ("vdcalt_both_n15", "vdc_alt", "both", 15, 24, "power_regulation", True),
    ("random_both_n15", "random", "both", 15, 40, "voltage_regulation", True),
    ("random_both_n4", "random", "both", 4, 75, "Q_regulation", True),
    ("random_volt_n10", "random", "voltage", 10, 30, "voltage_regulation", False),
]


def main():
    with mp.Pool(min(len(CASES), os.cpu_count() or 1)) as pool:
        res = pool.map(run_case, CASES)
    meta = {}
    for r in res:
        assert r["ab_rel_l2"] < 2e-5, (r["name"], r["ab_rel_l2"])
        np.savez_compressed(os.path.join(HERE, "traj_%s.npz" % r["name"]), **{k: v for k, v in r.items() if k != "name"})
        meta[r["name"]] = {"ab_rel_l2": r["ab_rel_l2"], "secs": r["secs"], "steps": int(len(r["actions"]))}
        print(r["name"], meta[r["name"]])
    # G1: steady states
    g1 = {}
    for der in ("50", "10"):
        m = pm.build_model(der)
        y0 = pm.steady_state(m)
        o = pm.algebraic(m, y0, 0.0, 1.0, m.Sinsol0, 0.0, m.Vdc_ref0)
        g1["y0_" + der] = y0
        g1["obs0_" + der] = np.array([o.i[0].real, o.i[0].imag, o.v[0].real, o.v[0].imag, o.S_PCC.real,
                                      o.S_PCC.imag, o.Vdc, o.Ppv, m.Vdc_ref0, 0.0, 0.0])
        g1["rhs0_" + der] = np.array(pm.rhs(m, y0, 0.0, 1.0, m.Sinsol0, 0.0, m.Vdc_ref0))
    np.savez_compressed(os.path.join(HERE, "steady_state.npz"), **g1)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump({"kind": "build-oracle goldens (parity unpinned: not reference outputs)", "cases": meta}, f, indent=1)


if __name__ == "__main__":
    main()
