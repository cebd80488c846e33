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
