from .qagent import BatchedQAgent  # noqa: F401
