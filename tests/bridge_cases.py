"""Named episodes of the converged fp64 bridge (oracle/pvder_bridge.c, RK45 at rtol 1e-8; the event-locating integrator of
oracle/pvder_bridge_filippov.h where the duty cycle saturates) that the -m gpu tests
replay on the device: the case definitions live here, ``tests/golden/make_bridge_goldens.py`` integrates them on
the CPU (minutes each) and commits ``tests/golden/bridge_<name>.npz``, the tests load them -- the GPU suite no longer
spends its time in the checker.  Event tables come from the oracle's counter generator, which the device generator
reproduces bit for bit: a test asserts that equality before it trusts a golden.  BUILD-ORACLE goldens (parity
unpinned: SURVEY.md 8(c)), not reference outputs."""
import os

import numpy as np

from helpers import both_events_spec, counter_tables
from oracle.pvder_oracle_env import DEFAULT_EVENTS_SPEC

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ramp(action):
    # the reference's evaluation driver holds ONE action for a whole episode (calc_returns, PVDER_env.py:480-487);
    # four envs on their own event tables (voltage + insolation), 160 env steps = max_sim_time
    return dict(N=4, seed=11, spec="both", n=15, steps=160, actions=("const", action))


_SAMPLE_IDS = tuple(int(i) for i in np.sort(np.random.default_rng(77).choice(65536, size=48, replace=False)))

CASES = {
    "ramp_a1": _ramp(1), "ramp_a2": _ramp(2), "ramp_a3": _ramp(3),
    # action 4 SATURATES the duty cycle from env step ~116: there the right-hand side is discontinuous across |m_p| = m_limit and
    # the solution SLIDES along that surface.  RK45 on the raw right-hand side only chatters around it (error ~ tol^0.17: 6e-3 /
    # 3.5e-3 / 2.3e-3 per env step at rtol 1e-8 / 1e-10 / 1e-12, profiles/r04_saturation_convergence.txt); round 5: the golden is
    # the EVENT-LOCATING integrator's run (oracle/pvder_bridge_filippov.h: regions IN / OUT / SLIDE per phase, switching instants
    # located, Filippov's equivalent dynamics while sliding) at rtol 1e-12, atol 1e-19 (u_p ~ 1e-6 carries the duty cycle through
    # Kp_GCC = 6000: the absolute tolerance has to sit below rtol |u|), and carries the rtol 1e-9 run beside it (obs_coarser): its
    # OWN spread, 5.1e-6 per env step at worst over all 160 steps (profiles/r05_filippov_convergence.txt)
    "ramp_a4": dict(_ramp(4), rtol=1e-12, atol_ratio=1e-7, method="filippov", rtol_coarser=1e-9),
    # four envs of the 65 536-env action-1 ramp (seed 3) in which the +2 % voltage event of env step 121 takes the duty
    # cycle -- |m| = 0.97 by then -- across its limit for a moment: one env step on the 23-state path (switched operator
    # sets), a 1e-3 imbalance in the current controller's integrators behind it that takes the rest of the episode to die
    # (carried on the matrix-core path by symmetrical components), the current limit from step ~125
    "ramp_a1_touch": dict(N=4, seed=3, spec="both", n=15, steps=160, actions=("const", 1), env_ids=(17360, 20888, 45096, 63022)),
    # 48 envs drawn from the 65 536-env batch of the 'dec' / 'no_change' episodes (seed 3): the distribution of the error along
    # the Q_ref ramps over envs nobody picked -- current limit, momentary duty-cycle saturations and their aftermath included
    "ramp_a1_sample": dict(N=48, seed=3, spec="both", n=15, steps=160, actions=("const", 1), env_ids=_SAMPLE_IDS),
    "ramp_a2_sample": dict(N=48, seed=3, spec="both", n=15, steps=160, actions=("const", 2), env_ids=_SAMPLE_IDS),
    # ... and the same 48 envs through a complete episode of RANDOM actions (the benchmark's workload)
    "episode_random_sample": dict(N=48, seed=3, spec="both", n=15, steps=160, actions=("random", 4242), env_ids=_SAMPLE_IDS),
    # a complete PVDER-v0 episode, random actions (tests of drift: etdlin and ETDRK4 replay it)
    "episode_random": dict(N=8, seed=1234, spec="both", n=15, steps=160, actions=("random", 2718)),
    # 64 envs, each with its OWN DC-link capacitance, outer-loop gains, PV array size and insolation scale (lane_param_sets:
    # the per-lane override of the ETDRK4 integrator), random actions; every env starts from ITS OWN steady
    # state (round 5; the reference builds each DER with steadyStateInitialization=True, PVDER_env.py:374-378).  The bridge
    # integrates each env with a model built from its own config block; obs_reset = its reset observation.
    "lane_params": dict(N=64, seed=21, spec="both", n=15, steps=40, actions=("random", 99), lane_seed=5),
    # ... twice that spread (round 5: the etdlin lane path picks its number of Krogstad steps per env step by the spread)
    "lane_params_wide": dict(N=24, seed=21, spec="both", n=15, steps=24, actions=("random", 99), lane_seed=6,
                             lane_spread=(0.5, 2.0, 0.67, 1.5)),
    # ... four times (round 6): BEYOND what the etdlin lane path is validated for -- the default constructor sends such a batch to the
    # ETDRK4 lane kernels by itself (vec_env.PVDERVec._lane_envelope)
    "lane_params_x4": dict(N=16, seed=21, spec="both", n=15, steps=16, actions=("random", 99), lane_seed=7,
                           lane_spread=(0.25, 4.0, 0.4, 2.5)),
}


def lane_param_sets(name):
    """{key: [N] values} of a case with per-env parameters: config_der.json keys (:2-21) + 'Sinsol_scale'."""
    from oracle.pvder_model import DER_CONFIG
    c = CASES[name]
    rng = np.random.default_rng(c["lane_seed"])
    N, base = c["N"], DER_CONFIG["50"]
    if "lane_spread" in c:           # log-uniform: as many envs below the group's value as above
        clo, chi, glo, ghi = c["lane_spread"]
        lu = lambda lo, hi: np.exp(rng.uniform(np.log(lo), np.log(hi), N))
        sets = {"C_actual": base["C_actual"] * lu(clo, chi)}
        for k in ("Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q"):
            sets[k] = base[k] * lu(glo, ghi)
    else:
        sets = {k: base[k] * rng.uniform(lo, hi, N) for k, (lo, hi) in
                (("C_actual", (0.7, 1.4)), ("Kp_DC", (0.8, 1.25)), ("Ki_DC", (0.8, 1.25)), ("Kp_Q", (0.8, 1.25)), ("Ki_Q", (0.8, 1.25)))}
    # (10 or 11 strings at 90-100 % insolation: the operating points stay clear of the duty-cycle limit -- |m| = 1.00004 with 9
    #  strings at 80 % --, where the device's switched steps are 1e-2 off: tests/test_gpu_etdlin.py, action 4)
    sets["Np"] = rng.integers(10, 12, N).astype(np.float64)           # 10 or 11 parallel strings (11: the reference's)
    sets["Sinsol_scale"] = rng.uniform(0.9, 1.0, N)
    return sets


def case_inputs(name):
    """(event tables [K+1, N] x 2, actions [steps, N], case dict) of a case, from its definition alone."""
    c = CASES[name]
    spec = both_events_spec() if c["spec"] == "both" else {k: dict(v) for k, v in DEFAULT_EVENTS_SPEC.items()}
    if "env_ids" in c:              # named envs of a larger batch: the tables the device generator gives THOSE env ids
        cols = [counter_tables(1, c["seed"], episode=1, env_id_base=i, spec=spec) for i in c["env_ids"]]
        vg, si = np.concatenate([v for v, _ in cols], axis=1), np.concatenate([s_ for _, s_ in cols], axis=1)
    else:
        vg, si = counter_tables(c["N"], c["seed"], episode=1, spec=spec)
    kind, arg = c["actions"]
    if kind == "const":
        acts = np.full((c["steps"], c["N"]), arg, dtype=np.int64)
    else:
        acts = np.random.default_rng(arg).integers(0, 5, size=(c["steps"], c["N"]))
    return vg, si, acts, dict(c, spec_dict=spec)


def path(name):
    return os.path.join(GOLD, "bridge_%s.npz" % name)


def load(name):
    """The golden of a case: dict with vgrid, sinsol, actions, obs [steps, N, 11], reward, done, m_abs [steps, N]
    (|m_a| at the end of each step: where the duty cycle saturates the solution stops being reproducible),
    iref_abs [steps, N]."""
    p = path(name)
    if not os.path.exists(p):
        raise FileNotFoundError("%s missing: run python tests/golden/make_bridge_goldens.py %s" % (p, name))
    g = dict(np.load(p))
    vg, si, acts, _ = case_inputs(name)
    assert np.array_equal(g["vgrid"], vg) and np.array_equal(g["sinsol"], si) and np.array_equal(g["actions"], acts), \
        "golden %s does not belong to its case definition any more: regenerate it" % name
    return g


# ---- the returns of calc_returns' four policies (tests/test_gpu_golden.py::test_calc_returns_vs_oracle) -------------
CALC_RETURNS = dict(seed=11, episodes=2, T=2.0, n=15, rand_seed=3,
                    cells=(("voltage_regulation", False), ("Q_regulation", False), ("power_regulation", True)))


def calc_returns_inputs():
    c = CALC_RETURNS
    K = int(c["T"] * 60 / c["n"])
    rand = np.random.default_rng(c["rand_seed"]).integers(0, 5, size=(K + 1, c["episodes"]))
    vg, si = counter_tables(c["episodes"], c["seed"], episode=1, env_id_base=0)
    return K, rand, vg, si


def load_calc_returns():
    p = os.path.join(GOLD, "bridge_calc_returns.npz")
    if not os.path.exists(p):
        raise FileNotFoundError("%s missing: run python tests/golden/make_bridge_goldens.py calc_returns" % p)
    g = dict(np.load(p))
    K, rand, vg, si = calc_returns_inputs()
    assert np.array_equal(g["rand"], rand) and np.array_equal(g["vgrid"], vg) and np.array_equal(g["sinsol"], si)
    return g
