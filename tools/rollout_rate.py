#!/usr/bin/env python3
"""Open-loop rollouts: K per-step launches from one host call (pvder_plan_rollout) vs the K steps fused
into one launch (pvder_plan_rollout_fused), same envs, same actions; env-steps/s of each (GPU box).
    python tools/rollout_rate.py [N] [K ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Ks = [int(a) for a in sys.argv[2:]] or [8, 16, 40, 159, 160]
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
POOL = 2048          # (distinct action rows: a short cycle of actions walks Q_ref / Vdc_ref off to the limits)
pool = torch.randint(0, 5, (POOL, N), generator=gen, device=dev, dtype=torch.uint8)
env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=True)
env.reset()
for K in Ks:
    blocks = torch.empty((K, 13, N), dtype=torch.float32, device=dev)
    for fused in (True, False, True, False):
        reps = max(4, 3200 // K)
        pos = 0
        for _ in range(2):
            env.rollout(pool[:K], out=blocks, fused=fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if pos + K > POOL:
                pos = 0
            env.rollout(pool[pos:pos + K], out=blocks, fused=fused)
            pos += K
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("fused=%d K=%4d  %.2f us/step  %.3e env-steps/s  (host %.2f us/step)"
              % (fused, K, dt / (reps * K) * 1e6, N * reps * K / dt, (t1 - t0) / (reps * K) * 1e6), flush=True)
