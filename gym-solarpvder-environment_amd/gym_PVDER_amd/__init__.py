"""gym_PVDER_amd -- MI355X-native batched PVDER-v0 (drop-in for ``gym_PVDER``'s step path).

``make('PVDER-v0', **kwargs)`` mirrors ``gym.make`` with the registration of the reference
(/root/reference/gym_PVDER/__init__.py:3-10): same id, same default kwargs, TimeLimit of 500
steps.  ``gym`` / ``gymnasium`` are optional: when one is importable the id is also registered
there; otherwise the built-in registry below serves ``make``/``spec``.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

# Kernel arguments in device memory instead of host memory (every wave of a launch reads them): worth
# 2-3 % on the step kernel.  Only effective if set before the process makes its first HIP call.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ENV_ID = 'PVDER-v0'
DEFAULT_KWARGS = {'n_sim_time_steps_per_env_step': 15, 'max_sim_time': 40.0, 'DISCRETE_REWARD': True,
                  'goals_list': ['voltage_regulation']}            # gym_PVDER/__init__.py:6-9
MAX_EPISODE_STEPS = 500                                             # gym_PVDER/__init__.py:10

_registry = {ENV_ID: SimpleNamespace(id=ENV_ID, entry_point='gym_PVDER_amd.envs:PVDER',
                                     kwargs=dict(DEFAULT_KWARGS), max_episode_steps=MAX_EPISODE_STEPS)}


def spec(env_id: str):
    if env_id not in _registry:
        raise KeyError("No registered env with id: %s" % env_id)
    return _registry[env_id]


def make(env_id: str = ENV_ID, **kwargs):
    """Create the single-env ``PVDER`` (old-gym API).  For N envs use :class:`PVDERVec`."""
    from .envs import PVDER
    s = spec(env_id)
    kw = dict(s.kwargs)
    kw.update(kwargs)
    env = PVDER(**kw)
    env.spec = s
    return env


def make_vec(num_envs: int, env_id: str = ENV_ID, **kwargs):
    """Create a batched :class:`PVDERVec` with the registered default kwargs."""
    from .vec_env import PVDERVec
    kw = dict(spec(env_id).kwargs)
    kw.update(kwargs)
    return PVDERVec(num_envs, **kw)


def _register_with_gym():
    """Register ``PVDER-v0`` with every importable gym package, as the reference does at import time
    (gym_PVDER/__init__.py:1-10).  A package that is not installed is not an error; anything else that goes wrong -- the id
    already taken because the reference's own ``gym_PVDER`` was imported first, a registry API that moved -- is LOGGED, never
    swallowed: ``gym.make('PVDER-v0')`` would otherwise fail (or build the other package's env) without a word."""
    import importlib
    import logging
    done = []
    for mod in ("gym", "gymnasium"):
        try:
            reg = importlib.import_module(mod + ".envs.registration")
        except ImportError:
            continue
        try:
            reg.register(id=ENV_ID, entry_point='gym_PVDER_amd.envs:PVDER', kwargs=dict(DEFAULT_KWARGS),
                         max_episode_steps=MAX_EPISODE_STEPS)
            done.append(mod)
        except Exception as e:                    # noqa: BLE001  (whatever the registry raises: report it)
            logging.getLogger("gym_PVDER_amd").warning("could not register %s with %s: %s: %s -- %s.make(%r) will not build this "
                                                       "package's env (gym_PVDER_amd.make does)", ENV_ID, mod, type(e).__name__, e,
                                                       mod, ENV_ID)
    return done


REGISTERED_WITH = _register_with_gym()

__all__ = ["make", "make_vec", "spec", "ENV_ID", "DEFAULT_KWARGS"]
