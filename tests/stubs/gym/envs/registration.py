import importlib


class EnvSpec:
    def __init__(self, id, entry_point=None, kwargs=None, max_episode_steps=None, reward_threshold=None):
        self.id, self.entry_point, self._kwargs = id, entry_point, dict(kwargs or {})
        self.max_episode_steps, self.reward_threshold = max_episode_steps, reward_threshold

    def make(self, **kwargs):
        kw = dict(self._kwargs)
        kw.update(kwargs)
        if callable(self.entry_point):
            env = self.entry_point(**kw)
        else:
            mod, attr = self.entry_point.split(':')
            env = getattr(importlib.import_module(mod), attr)(**kw)
        env.unwrapped.spec = self
        return env


registry = {}


def register(id, **kwargs):
    if id in registry:
        raise ValueError('Cannot re-register id: {}'.format(id))
    registry[id] = EnvSpec(id, **kwargs)


def spec(id):
    if id not in registry:
        raise KeyError('No registered env with id: {}'.format(id))
    return registry[id]


def make(id, **kwargs):
    s = spec(id)
    env = s.make(**kwargs)
    if s.max_episode_steps is not None:
        from ..wrappers import TimeLimit
        env = TimeLimit(env, max_episode_steps=s.max_episode_steps)
    return env
