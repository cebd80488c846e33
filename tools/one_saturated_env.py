#!/usr/bin/env python3
"""ONE env of 65 536 at the duty-cycle limit (Vdc_ref -7 / -9 / -10 %), the others on random actions: microseconds per batched step
(off_range="integrate": the whole launch waits for that env's workgroup on the 23-state path).  Round 2: 55 000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
N = 65536
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.randint(0, 5, (40, N), generator=gen, device=dev, dtype=torch.uint8)
acts[:, 0] = 0
for sat in (0.0, 0.07, 0.09, 0.10):
    env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=False)
    env.reset()
    if sat:
        env.fstate[24, 0] = 1.1 * (1.0 - sat)
    env.rollout(acts, fused=False)
    times = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout(acts, fused=False)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 40 * 1e6)
    f = env.fstate
    m0 = float((torch.complex(f[2, 0], f[3, 0]) + 6000.0 * torch.complex(f[4, 0], f[5, 0])).abs())
    print("%s: %s us per batched step (three 40-step blocks); |m| of env 0: %.4f" % ("env 0 at Vdc_ref -%d %%" % round(100 * sat) if sat else "all envs in range", " ".join("%.1f" % t for t in times), m0), flush=True)
