"""CPU ORACLE (test infrastructure only) -- one worker of bench.py's all-cores CPU baseline:
steps one OraclePVDER env (reference LSODA settings) for `seconds` of CPU and prints env-steps/s.
Started as a plain child process BEFORE bench.py touches the GPU; imports no torch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed: int, seconds: float) -> float:
    import numpy as np
    from oracle.pvder_oracle_env import OraclePVDER
    env = OraclePVDER(solver="reference", seed=seed)      # PVDER-v0 defaults, voltage events on
    env.reset()
    env.step(0)               # untimed: the first step builds the analytic Jacobian (sympy -> lambdify, one-off;
    env.reset()               # tens of seconds on a box whose page cache is cold)
    rng = np.random.default_rng(seed)
    n, used = 0, 0.0
    while used < seconds:
        a = int(rng.integers(0, 5))
        t0 = time.time()
        _, _, done, _ = env.step(a)
        used += time.time() - t0
        n += 1
        if done:
            env.reset()
    return n / used, n, used


if __name__ == "__main__":
    v, n, used = run(int(sys.argv[1]), float(sys.argv[2]))
    print("%.6f %d %.3f" % (v, n, used))
