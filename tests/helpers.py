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
