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

# the same rollout as ONE open-loop call: the actions of a whole episode up front, the env steps fused into a
# handful of launches (state in registers from step to step), per-step output blocks [K, 13, N] on the device
K = 160
plan_actions = torch.randint(0, 5, (K, vec.num_envs), device=vec.device, dtype=torch.uint8)
blocks = torch.empty((K, 13, vec.num_envs), device=vec.device)
vec.rollout(plan_actions, out=blocks)                     # warm-up (fused: etdlin, three-phase, K >= 4)
torch.cuda.synchronize()
t0 = time.time()
vec.rollout(plan_actions, out=blocks)
torch.cuda.synchronize()
dt = time.time() - t0
print("fused rollout: %d envs x %d steps in %.4f s = %.3g env-steps/s; mean return %.2f"
      % (vec.num_envs, K, dt, vec.num_envs * K / dt, float(blocks[:, 11].sum(0).mean())))

# domain randomisation over parameter sets: three config blocks x 4096 envs, one batched API
from gym_PVDER_amd.param_groups import PVDERVecGroups  # noqa: E402

groups = PVDERVecGroups([4096] * 3, [{}, {"C_actual": 360e-6, "Kp_DC": -2.5}, {"Np": 10, "Rf_actual": 0.003}],
                        device=vec.device)
obs = groups.reset()
obs, reward, done, _ = groups.step(torch.randint(0, 5, (groups.num_envs,), device=vec.device, dtype=torch.uint8))
print("parameter sets: obs %s, Vdc of the first env of each set: %s"
      % (tuple(obs.shape), [round(float(obs[k * 4096, 6]), 5) for k in range(3)]))
