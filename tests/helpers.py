"""Shared helpers of the parity tests (the oracle is imported here and in tests only)."""
import numpy as np

from oracle import pvder_model as pm
from oracle.pvder_oracle_env import DEFAULT_EVENTS_SPEC


def rel_l2(a, b, axis=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b, axis=axis) / np.maximum(np.linalg.norm(b, axis=axis), 1e-300)


def counter_tables(N, seed, episode=1, env_id_base=0, spec=None, sinsol0=100.0):
    """[K+1, N] event tables from the oracle's counter generator (device-identical)."""
    spec = spec or DEFAULT_EVENTS_SPEC
    cols = [pm.make_event_table(spec, sinsol0, counter_key=(seed, env_id_base + e, episode)) for e in range(N)]
    vg = np.stack([c.vgrid for c in cols], axis=1)
    si = np.stack([c.sinsol for c in cols], axis=1)
    return vg, si


def both_events_spec():
    s = {k: dict(v) for k, v in DEFAULT_EVENTS_SPEC.items()}
    s["insolation"]["ENABLE"] = True
    return s


_BRIDGE_EPISODES = {}


def bridge_episode(vg, si, ev_s0, ev_ds, actions, n=15, rtol=1e-8, atol=1e-11):
    """Observation / reward / done trajectory of the converged fp64 bridge (RK45) for the given event tables
    and actions ([steps, N]); cached per process -- the whole-episode tests of several integrators replay the
    same episode, and 160 env steps at rtol 1e-8 are ~100 s of CPU."""
    import hashlib
    from oracle.bridge_env import BridgeVecEnv
    vg, si = np.ascontiguousarray(vg, dtype=np.float64), np.ascontiguousarray(si, dtype=np.float64)
    actions = np.ascontiguousarray(actions, dtype=np.int64)
    key = hashlib.sha1(vg.tobytes() + si.tobytes() + actions.tobytes()
                       + repr((ev_s0, ev_ds, n, rtol, atol)).encode()).hexdigest()
    if key not in _BRIDGE_EPISODES:
        ref = BridgeVecEnv(actions.shape[1], n=n, method="rk45", rtol=rtol, atol=atol)
        ref.reset(vg, si, ev_s0=ev_s0, ev_ds=ev_ds)
        obs, rew, done = [], [], []
        for a in actions:
            o, r, d = ref.step(a)
            obs.append(o.copy()); rew.append(np.array(r, copy=True)); done.append(np.array(d, copy=True))
        _BRIDGE_EPISODES[key] = (np.stack(obs), np.stack(rew), np.stack(done))
    return _BRIDGE_EPISODES[key]


def assert_discrete_rewards(got_r, want_r, want_obs, goal="voltage_regulation", der_id="50", tol=2e-4, what=""):
    """Discrete rewards must equal the oracle's -- except where the oracle's own observation puts the goal's error
    within ``tol`` of a ladder threshold (PVDER_env.py:263-299: 1 % / 5 %, power 1 % / 3 %), where the 1e-4-level
    difference between the two integrators may land on the other side.  Every mismatch is ASSERTED to sit there
    (an allowance that does not check its excuse is not a test).  ``want_obs``: [..., 11] of the
    oracle (balanced envs: Vrms = |v_a| / sqrt 2).  Returns the number of mismatches."""
    from oracle.pvder_oracle_env import P_REF, Q_REF_TARGET
    got_r, want_r, want_obs = np.asarray(got_r), np.asarray(want_r), np.asarray(want_obs, dtype=np.float64)
    bad = got_r != want_r
    if not bad.any():
        return 0
    m = pm.build_model(der_id)
    o = want_obs[bad]
    if goal == "voltage_regulation":
        err, th = np.abs(np.hypot(o[..., 2], o[..., 3]) / np.sqrt(2.0) - m.Vrms_ref) / m.Vrms_ref, (0.01, 0.05)
    elif goal == "power_regulation":
        err, th = np.abs(o[..., 4] - P_REF / pm.SBASE) / (P_REF / pm.SBASE), (0.01, 0.03)
    else:
        err, th = np.abs(o[..., 5] - Q_REF_TARGET / pm.SBASE) / (Q_REF_TARGET / pm.SBASE), (0.01, 0.05)
    dist = np.minimum(np.abs(err - th[0]), np.abs(err - th[1]))
    assert dist.max() < tol, "%s: %d discrete reward(s) differ away from a threshold (distance %.2e, errors %s)" % (
        what, int(bad.sum()), dist.max(), err[:5])
    return int(bad.sum())


OBS_NAMES = ("ia.real", "ia.imag", "va.real", "va.imag", "P_PCC", "Q_PCC", "Vdc", "Ppv", "Vdc_ref", "Q_ref", "t/max")   # PVDER_env.py:44-49


def per_observation_error(got, want, floor=0.05):
    """|got - want| / max(|want|, floor) of the EIGHT dynamic observations (PVDER_env.py:531-542: entries 0-7; 8-10 are the
    set points and the clock, bookkeeping that is never wrong and carries half of the 11-vector's norm): [..., 8].  The
    floor keeps the entries that sit near zero (ia.imag, va.imag, Q_PCC ~ 0.01 pu) from dividing by nothing: for them the
    bound is absolute, rel * floor."""
    got, want = np.asarray(got, dtype=np.float64)[..., :8], np.asarray(want, dtype=np.float64)[..., :8]
    return np.abs(got - want) / np.maximum(np.abs(want), floor)


def assert_per_observation(got, want, rel=1e-4, floor=0.05, what=""):
    """a bound on EVERY dynamic observation by itself -- the rel-L2 over the 11-vector hides Q_PCC (what
    Q_regulation rewards, PVDER_env.py:263-275) and ia.imag behind entries of norm 1.  Names the worst entry when it fails."""
    e = per_observation_error(got, want, floor)
    flat = e.reshape(-1, 8)
    k = int(flat.max(axis=0).argmax())
    where = np.unravel_index(int(e[..., k].argmax()), e[..., k].shape)
    assert flat.max() <= rel, "%s: observation %d (%s) off by %.3e of max(|want|, %g) at %s (got %.8g, want %.8g); worst per entry %s" % (
        what, k, OBS_NAMES[k], flat.max(), floor, where, np.asarray(got)[..., k][where], np.asarray(want)[..., k][where],
        " ".join("%.1e" % v for v in flat.max(axis=0)))
    return flat.max(axis=0)
