"""A STAND-IN for the old ``gym`` package (pre-0.26 API), for tests only: `gym` is not installed in this image and the
reference's entry into the path is ``gym.make('PVDER-v0')`` (/root/reference/gym_PVDER/__init__.py:1-10;
examples/gym_PVDER_environment_import_test.py:14-38).  It implements what that entry touches -- ``Env``, ``Wrapper``,
``envs.registration.register / make / spec`` with string entry points and ``max_episode_steps`` -> a ``TimeLimit`` wrapper --
written from gym's documented behaviour, not from its source.  Put on sys.path by tests/test_gym_entry.py only."""
from .core import Env, Wrapper                      # noqa: F401
from .envs.registration import make, register, spec  # noqa: F401
from . import wrappers                              # noqa: F401

__version__ = "0.0.stub"
