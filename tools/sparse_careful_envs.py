#!/usr/bin/env python3
"""GPU box: every STRIDE-th env of 65 536 held beyond the current limit (Q_ref = 1.1 pu: careful pass, frozen integrators), the
others at Q_ref = 0 on no-op actions: microseconds per batched step against how the careful lanes are spread over the waves.
    python tools/sparse_careful_envs.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
N = 65536
dev = torch.device("cuda:0")
acts = torch.zeros((40, N), device=dev, dtype=torch.uint8)
for stride in (0, 65536, 256, 64, 32, 8, 2, 1):
    env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=False)
    env.reset()
    if stride:
        env.fstate[23, ::stride] = 1.1
    env.rollout(acts, fused=False)
    times = []
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout(acts, fused=False)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 40 * 1e6)
    print("every %5d-th env at Q_ref 1.1 pu (%5d envs): %s us per batched step" % (stride, N // stride if stride else 0, " ".join("%.1f" % t for t in times)), flush=True)
