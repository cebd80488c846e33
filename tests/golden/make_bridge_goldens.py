#!/usr/bin/env python
"""Integrate the named bridge episodes of tests/bridge_cases.py on the CPU and store them as
tests/golden/bridge_<name>.npz (BUILD-ORACLE goldens: parity unpinned, see make_goldens.py).

    python tests/golden/make_bridge_goldens.py [case ...]        (default: every case without a file)
    python tests/golden/make_bridge_goldens.py calc_returns      (the returns of calc_returns' policies)
    python tests/golden/make_bridge_goldens.py filippov_window   (three env steps at the duty-cycle limit: CPU test fixture)

One process per env (the C bridge is single-threaded); ~25 s per env and 160-step episode."""
import multiprocessing as mp
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):     # one process per env: BLAS threads on top only fight
    os.environ.setdefault(_v, "1")                                            # (48 envs: 109 s with, 14 min without)
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import bridge_cases  # noqa: E402
from oracle import pvder_model as pm  # noqa: E402
from oracle.bridge_env import BridgeVecEnv  # noqa: E402


def run_env(job):
    name, e, coarser = job
    vg, si, acts, c = bridge_cases.case_inputs(name)
    rtol = c["rtol_coarser"] if coarser else c.get("rtol", 1e-8)
    method, atol = c.get("method", "rk45"), rtol * c.get("atol_ratio", 1e-3)
    if "lane_seed" in c:
        # an env with its own parameter set: a model from its own config block, the insolation it sees scaled, started from
        # ITS OWN steady state (the reference builds each DER with steadyStateInitialization=True, PVDER_env.py:374-378)
        sets = bridge_cases.lane_param_sets(name)
        ref = BridgeVecEnv(1, n=c["n"], method=method, rtol=rtol, atol=atol,
                           overrides={k: float(v[e]) for k, v in sets.items() if k != "Sinsol_scale"})
        ref.reset(vg[:, e:e + 1], si[:, e:e + 1] * sets["Sinsol_scale"][e], ev_s0=60, ev_ds=60)
        from oracle import bridge
        ref.Y[0] = bridge.to_delta(pm.steady_state(ref.model, Sinsol=ref.model.Sinsol0 * sets["Sinsol_scale"][e]), 0.0)
    else:
        ref = BridgeVecEnv(1, n=c["n"], method=method, rtol=rtol, atol=atol)
        ref.reset(vg[:, e:e + 1], si[:, e:e + 1], ev_s0=60, ev_ds=60)
    m = ref.model
    obs0 = ref.observe()[0].copy()
    obs, rew, done, mabs, iabs = [], [], [], [], []
    for a in acts[:, e]:
        o, r, d = ref.step(np.array([a]))
        obs.append(o[0].copy()); rew.append(r[0]); done.append(d[0])
        Y = ref.Y[0]
        mabs.append(abs(complex(Y[2] + m.Kp_GCC * Y[4], Y[3] + m.Kp_GCC * Y[5])))
        iabs.append(float(np.hypot(Y[19] + m.Kp_DC * (ref.vdcref[0] - Y[18]), Y[20] - m.Kp_Q * (ref.qref[0] - o[0, 5]))))
    if "lane_seed" in c:
        obs = [obs0] + obs                    # (row 0: the reset observation at the env's own operating point)
    return name, e, np.array(obs), np.array(rew), np.array(done), np.array(mabs), np.array(iabs), coarser


def make_calc_returns():
    """bridge_calc_returns.npz: mean return and final refs of calc_returns' four policies (PVDER_env.py:458-497) for
    the cells of bridge_cases.CALC_RETURNS, from the converged bridge with the reference's reward rules."""
    ACTION_SPECS = {'random': None, 'inc': 0, 'dec': 1, 'no_change': 2}      # PVDER_env.py:462, :480-487
    c = bridge_cases.CALC_RETURNS
    K, rand, vg, si = bridge_cases.calc_returns_inputs()
    out = dict(rand=rand, vgrid=vg, sinsol=si)
    for goal, discrete in c["cells"]:
        for name, fixed in ACTION_SPECS.items():
            ref = BridgeVecEnv(c["episodes"], n=c["n"], max_sim_time=c["T"], discrete=discrete, goal=goal, method="rk45",
                               rtol=1e-8, atol=1e-11)
            ref.reset(vg, si)
            total = np.zeros(c["episodes"])
            for k in range(K):
                a = np.full(c["episodes"], fixed) if fixed is not None else rand[k]
                _, r, d = ref.step(a)
                total += r
            assert bool(d.all())
            out["%s/%s" % (goal, name)] = np.array([total.mean(), ref.vdcref.mean(), ref.qref.mean()])
    np.savez_compressed(os.path.join(HERE, "bridge_calc_returns.npz"), **out)
    print("calc_returns: %d cells" % (len(out) - 3))


def make_filippov_window():
    """filippov_window.npz: env 1 of ramp_a4 by the event-locating integrator at rtol 1e-12 -- its state at the end of env step
    116 (the last one that ends clear of the duty-cycle limit) and the next three env steps (at the limit, leaving it): what
    tests/test_oracle.py holds coarser settings and the chattering RK45 against without integrating 116 steps first."""
    vg, si, acts, c = bridge_cases.case_inputs("ramp_a4")
    e, k0 = 1, 116
    ref = BridgeVecEnv(1, n=c["n"], method="filippov", rtol=1e-12, atol=1e-19)
    ref.reset(vg[:, e:e + 1], si[:, e:e + 1], ev_s0=60, ev_ds=60)
    obs, Y = [], []
    for k in range(k0 + 3):
        o, r, d = ref.step(np.array([acts[k, e]]))
        obs.append(o[0].copy()); Y.append(ref.Y[0].copy())
    np.savez_compressed(os.path.join(HERE, "filippov_window.npz"), Y0=Y[k0 - 1], k0=k0, env=e, obs=np.array(obs[k0:]), Y=np.array(Y[k0:]),
                        vgrid=vg[:, e], sinsol=si[:, e], rtol=1e-12, atol=1e-19)
    print("filippov_window: env %d, env steps %d..%d" % (e, k0, k0 + 2))


def main():
    if sys.argv[1:] == ["calc_returns"]:
        return make_calc_returns()
    if sys.argv[1:] == ["filippov_window"]:
        return make_filippov_window()
    names = sys.argv[1:] or [n for n in bridge_cases.CASES if not os.path.exists(bridge_cases.path(n))]
    jobs = [(n, e, co) for n in names for e in range(bridge_cases.CASES[n]["N"])
            for co in ((False, True) if "rtol_coarser" in bridge_cases.CASES[n] else (False,))]
    t0 = time.time()
    with mp.Pool(min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as pool:
        res = pool.map(run_env, jobs, chunksize=1)
    for n in names:
        vg, si, acts, c = bridge_cases.case_inputs(n)
        mine = sorted([r for r in res if r[0] == n and not r[7]], key=lambda r: r[1])
        extra = {}
        if "rtol_coarser" in c:          # the same case at a coarser tolerance: the golden's OWN spread travels with it
            co = sorted([r for r in res if r[0] == n and r[7]], key=lambda r: r[1])
            extra = dict(obs_coarser=np.stack([r[2] for r in co], axis=1), rtol_coarser=c["rtol_coarser"])
        obs_all = np.stack([r[2] for r in mine], axis=1)
        if "lane_seed" in c:
            extra["obs_reset"], obs_all = obs_all[0], obs_all[1:]
        np.savez_compressed(bridge_cases.path(n), vgrid=vg, sinsol=si, actions=acts,
                            obs=obs_all, reward=np.stack([r[3] for r in mine], axis=1),
                            done=np.stack([r[4] for r in mine], axis=1), m_abs=np.stack([r[5] for r in mine], axis=1),
                            iref_abs=np.stack([r[6] for r in mine], axis=1), ev_s0=60, ev_ds=60, n=c["n"],
                            rtol=c.get("rtol", 1e-8), atol=c.get("rtol", 1e-8) * c.get("atol_ratio", 1e-3),
                            method=c.get("method", "rk45"), **extra)
        print("%s: %d envs x %d steps" % (n, c["N"], c["steps"]))
    print("%.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
