from ..core import Wrapper


class TimeLimit(Wrapper):
    """Ends an episode after ``max_episode_steps`` steps (``info['TimeLimit.truncated']``), as old gym's does."""

    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._max_episode_steps is not None and self._elapsed_steps >= self._max_episode_steps:
            info = dict(info)
            info['TimeLimit.truncated'] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)
