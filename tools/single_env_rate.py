#!/usr/bin/env python
"""Steps/s of the single-env old-gym API (gym.make('PVDER-v0'), random agent, 1000 steps: BASELINE
configs[0] shape) on the HIP path -- launch + host read-back per step (GPU box only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))
import gym_PVDER_amd as gp

for rec in (True, False):
    env = gp.make('PVDER-v0', record_trajectory=rec)
    env.reset()
    for _ in range(50):
        env.step(env.action_space.sample())
    env.reset()
    t0 = time.perf_counter()
    n = 0
    for _ in range(1000):
        _, _, done, _ = env.step(env.action_space.sample())
        n += 1
        if done:
            env.reset()
    dt = time.perf_counter() - t0
    print("single env, record_trajectory=%s: %d steps in %.3f s = %.0f env-steps/s (%.0f us per step)" % (rec, n, dt, n / dt, dt / n * 1e6))
