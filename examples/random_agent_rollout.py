#!/usr/bin/env python
"""Random-agent rollout -- the shape of the reference's examples/gym_PVDER_environment_import_test.py
(:23-31), first with the single env (old-gym API), then with 65 536 envs in one fused launch per step.

    python examples/random_agent_rollout.py            (needs an MI355X and the built library)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))

import torch  # noqa: E402

import gym_PVDER_amd as gym_PVDER  # noqa: E402

env = gym_PVDER.make('PVDER-v0')
observation, done, ret = env.reset(), False, 0
while not done:
    action = env.action_space.sample()
    observation, reward, done, _ = env.step(action)
    ret += reward
env.render()
print("single env: %d steps, return %d, Q_ref %.1f VAR" % (env.steps, ret, env.env.sim.PV_model.Q_ref * env.sim.Sbase))

vec = gym_PVDER.make_vec(65536)
obs = vec.reset()
t0 = time.time()
for _ in range(320):                                      # two episodes of every env
    actions = torch.randint(0, 5, (vec.num_envs,), device=vec.device, dtype=torch.uint8)
    obs, reward, done, _ = vec.step(actions)
torch.cuda.synchronize()
dt = time.time() - t0
print("batched: %d envs x 320 steps in %.3f s = %.3g env-steps/s; last-step stats %s"
      % (vec.num_envs, dt, vec.num_envs * 320 / dt, vec.step_stats()))
