from . import registration  # noqa: F401
