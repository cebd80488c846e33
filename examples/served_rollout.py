#!/usr/bin/env python
"""A policy that runs a few env steps BEHIND the batch it drives, on the served rollout (PVDERVec.rollout_served, round 6): the env
state stays in registers inside ONE resident launch sequence while the actions arrive row by row -- the shape of a collect loop
whose policy works on observations that are D steps old (the reference's DQN demo collects with a policy updated every so often:
examples/gym_PVDER_environment_tf_agents_DQN_demo.ipynb cells 23-28; its env steps are PVDER.step, PVDER_env.py:138-196).

    python examples/served_rollout.py            (needs an MI355X and the built library)

Also shown: a desynchronised batch (every env in its own episode phase) re-ordered by clock with sort_lanes() -- the step launch
then runs at 11 us instead of 22 (homogeneous waves; event tables written as the episode advances)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))

import torch  # noqa: E402

import gym_PVDER_amd as gym_PVDER  # noqa: E402

N, K, D = 65536, 640, 8
vec = gym_PVDER.make_vec(N)
vec.reset()
dev = vec.device
actions = torch.zeros((K, N), dtype=torch.uint8, device=dev)           # row k is WRITTEN by the policy just before it is posted
blocks = torch.empty((K, 13, N), dtype=torch.float32, device=dev)      # per-step output blocks: 11 observations, reward, done
side = torch.cuda.Stream(device=dev)                                   # the policy's stream (the current one holds the served launches)


policy_seconds = [0.0]


def policy(k):
    """Row k from the observation of step k - D - 1 (all of it complete: rows are posted D steps behind the slowest workgroup):
    a voltage regulator with a dead band -- Q_ref one notch up where the PCC voltage magnitude is more than 2 % low, one notch down
    where it is more than 2 % high, nothing in between, |Q_ref| <= 0.3 pu -- an env AT its current limit leaves the resident loop for the
    per-step body, and a paced rollout then pays a launch pair and a host round trip per step (the served loop writes its output rows through the L2 before it reports
    them: a kernel that starts on this side stream reads them whole)."""
    t_in = time.time()
    with torch.cuda.stream(side):
        if k <= D:
            actions[k].fill_(0)
        else:
            o = blocks[k - D - 1]
            vmag = torch.hypot(o[2], o[3])
            up, down = (vmag < 0.98) & (o[9] < 0.3), (vmag > 1.02) & (o[9] > -0.3)     # (|Q_ref| kept below 0.3 pu: clear of the current limit)
            actions[k].copy_(torch.where(up, 1, torch.where(down, 2, 0)).to(torch.uint8))
    side.synchronize()
    policy_seconds[0] += time.time() - t_in


torch.cuda.synchronize()
t0 = time.time()
vec.rollout_served(actions, out=blocks, depth=D, fill=policy)
dt = time.time() - t0
print("served rollout: %d envs x %d steps, policy %d steps behind: %.3f s = %.3g env-steps/s (%.1f us per env step, %.1f of them this "
      "policy's own torch launches + sync per row; the env side alone: tools/served_experiment/ab_served.py, 5.5 us); mean reward %.3f"
      % (N, K, D, dt, N * K / dt, 1e6 * dt / K, 1e6 * policy_seconds[0] / K, float(blocks[:, 11].mean())))

# ---- a desynchronised batch: staggered resets, then the lanes re-ordered by clock
vec2 = gym_PVDER.make_vec(N)
vec2.reset()
a = torch.randint(0, 5, (320, N), device=dev, dtype=torch.uint8)
phase = (torch.arange(N, device=dev) * 7919) % 160
for k in range(160):                                                   # env e restarts its episode at step phase[e]
    vec2.step(a[k], return_tensors=False)
    vec2.reset(phase == k)
for name, prepare in (("as the resets left them", lambda: None), ("after sort_lanes()", lambda: vec2.sort_lanes(lane_io=True))):
    prepare()
    vec2.rollout(a[:160], fused=False)
    torch.cuda.synchronize(); t0 = time.time()
    vec2.rollout(a, fused=False)
    torch.cuda.synchronize()
    print("desynchronised batch %-26s %.1f us per batched step" % (name + ":", 1e6 * (time.time() - t0) / 320))
