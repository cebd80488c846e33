#!/usr/bin/env python3
"""model_1 (single phase, derId 10): microseconds per batched step at 65 536 envs -- random actions, and the reference's
fixed-action episodes (calc_returns), by 20-step block.    python tools/model1_rate.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.randint(0, 5, (100, N), generator=gen, device=dev, dtype=torch.uint8)
env = PVDERVec(N, device=dev, model_type="model_1", seed=2, events_spec=both_events_spec(), auto_reset=True)
env.reset(); env.rollout(acts, fused=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): env.rollout(acts, fused=False)
torch.cuda.synchronize(); print("random actions: %.1f us per batched step" % ((time.perf_counter() - t0) / 500 * 1e6), flush=True)
for act in (1, 2, 3, 4):
    env = PVDERVec(N, device=dev, model_type="model_1", seed=3, events_spec=both_events_spec(), auto_reset=False)
    env.reset()
    a = torch.full((20, N), act, dtype=torch.uint8, device=dev)
    blocks = []
    for k in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout(a, fused=False)
        torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / 20 * 1e6)
    f = env.fstate
    m = (torch.complex(f[2], f[3]) + 6000.0 * torch.complex(f[4], f[5])).abs()
    print("action %d: us per step by 20-step block: %s   |m| max at the end %.4f  nonfinite %d" % (act, " ".join("%.0f" % b for b in blocks), float(m.max()), env.step_stats()["n_nonfinite"]), flush=True)
