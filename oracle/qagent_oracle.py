"""CPU ORACLE (test infrastructure only) -- the reference's tabular Q-learning update, restated for ONE
env with a sparse table (the reference's dense 6^11 x 5 float64 table is 14.5 GB).

Follows /root/reference/gym_PVDER/agents/qagent.py: table initialised to -2 (:37), discretisation
``(state - low) / win`` truncated to int (:43-45; the reference's ``np.int`` is gone from current NumPy --
``int`` is what it was), update ``Q <- (1-lr) Q + lr (r + gamma max_a' Q')`` while the episode runs (:83-95) and
``Q <- 0`` for the transition that ends it (:100-103), epsilon decay once per episode
(agent_utilities.py:79-84).  The product path never imports this file.  PARITY UNPINNED (it is driven by the
build's env)."""
from collections import defaultdict

import numpy as np

LEARNING_RATE, DISCOUNT = 0.1, 0.95               # qagent.py:14-15
EPSILON_DECAY, MIN_EPSILON = 0.98, 0.001          # qagent.py:24-25


class OracleQTable:
    def __init__(self, n_obs=11, n_act=5, bins=6, low=-10.0, high=10.0):
        self.n_act, self.bins, self.low = n_act, bins, low
        self.win = (high - low) / bins                                     # :38
        self.q = defaultdict(lambda: np.full(n_act, -2.0))                 # :37
        self.epsilon = 1.0

    def discrete(self, state):                                             # :43-45
        d = ((np.asarray(state, dtype=np.float64) - self.low) / self.win).astype(int)
        return tuple(np.clip(d, 0, self.bins - 1))                         # (clip: the Box bound itself)

    def update(self, s, a, r, s2, done):
        if not done:                                                       # :83-95
            max_future_q = np.max(self.q[s2])
            current_q = self.q[s][a]
            self.q[s][a] = (1 - LEARNING_RATE) * current_q + LEARNING_RATE * (r + DISCOUNT * max_future_q)
        else:                                                              # :100-103
            self.q[s][a] = 0.0

    def update_epsilon(self):                                              # agent_utilities.py:79-84
        if self.epsilon > MIN_EPSILON:
            self.epsilon = max(MIN_EPSILON, self.epsilon * EPSILON_DECAY)
