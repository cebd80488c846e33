from .qagent import BatchedQAgent  # noqa: F401
from .dqn import BatchedDQNAgent, QNetwork  # noqa: F401  (the TF-Agents DQN demo's loop, env and acting policy on the device)
