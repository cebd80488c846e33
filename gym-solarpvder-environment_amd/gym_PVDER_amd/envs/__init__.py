from .pvder_env import PVDER  # noqa: F401  (mirrors gym_PVDER/envs/__init__.py:1)
