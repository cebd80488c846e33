#!/usr/bin/env python3
"""The reference's fixed-action episodes (calc_returns, PVDER_env.py:480-487) as ROLLOUTS: 160 steps of one action from
one host call, per-step launches (pvder_plan_rollout) vs the fused kernel pair (pvder_plan_rollout_fused, what
evaluate.calc_returns uses); microseconds per batched env step, and the two results compared.
    python tools/ramp_rollout_rate.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
K = 160
for act in (1, 2, 3, 4):
    acts = torch.full((K, N), act, dtype=torch.uint8, device=dev)
    res = {}
    for fused in (False, True):
        env = PVDERVec(N, device=dev, seed=3, events_spec=both_events_spec(), auto_reset=False)
        blocks = torch.empty((K, 13, N), dtype=torch.float32, device=dev)
        best = 1e9
        for rep in range(3):
            env.reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            env.rollout(acts, out=blocks, fused=fused)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[fused] = (best, blocks.clone())
        print("action %d fused=%d: %.1f us per batched step (160-step episode, best of 3)" % (act, fused, best / K * 1e6), flush=True)
    same = torch.equal(res[False][1], res[True][1])
    print("   fused == per-step launches, bit for bit: %s" % same, flush=True)
