#!/usr/bin/env python3
"""Q-learning loop rates (GPU box): the torch-op agent (train), the one-launch agent step (train_fused: two launches per
env step) and the closed loop inside the fused rollout (train_fused(closed_loop=True): a handful of launches per episode).
    python tools/agent_rate.py [N] [bins]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd")]
import torch
from gym_PVDER_amd.agents import BatchedQAgent
from gym_PVDER_amd.vec_env import PVDERVec

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
for name, kw in (("train", {}), ("train_fused", {}), ("closed_loop", dict(closed_loop=True))):
    env = PVDERVec(N, device=dev, auto_reset=False, seed=3)
    agent = BatchedQAgent(env, bins=bins, seed=1)
    fn = agent.train if name == "train" else agent.train_fused
    fn(episodes=1, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(episodes=3, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 3 * 160
    print("%-12s %d envs, table %d^11 x 5 (%.1f GB): %.1f us per env step, %.3e env-steps/s"
          % (name, N, bins, agent.q_table.numel() * 4 / 1e9, dt / steps * 1e6, N * steps / dt), flush=True)
    del agent, env
    torch.cuda.empty_cache()

# ---- the Q-network policy (PVDER_POLICY_QNET: the QNetwork of the reference's DQN demo) as the action source: a greedy evaluation
#      episode with env launch + policy launch per step, and inside the fused closed-loop rollout; one DQN iteration (collect 1 + train 1)
from gym_PVDER_amd.agents import BatchedDQNAgent
tenv = PVDERVec(N, device=dev, auto_reset=True, seed=3)
ag = BatchedDQNAgent(tenv, seed=1, replay_buffer_capacity=N * 8)
for fused in (False, True):
    ev = PVDERVec(N, device=dev, auto_reset=False, seed=4)
    ag.evaluate(ev, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = ag.evaluate(ev, fused=fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-12s %d envs, Q-network 11 -> 100 -> 5 as the policy (%s): %.1f us per env step, %.3e env-steps/s"
          % ("qnet_eval", N, "inside the fused rollout" if fused else "env launch + policy launch per step", dt / (3 * r["steps"]) * 1e6,
             N * 3 * r["steps"] / dt), flush=True)
ag.train(num_iterations=20, initial_collect_steps=4)
torch.cuda.synchronize()
t0 = time.perf_counter()
ag.train(num_iterations=200, initial_collect_steps=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%-12s %d envs: one DQN iteration (collect one batched step under the device policy + one Adam step on 64 replayed rows): %.1f us"
      % ("dqn_iter", N, dt / 200 * 1e6), flush=True)
