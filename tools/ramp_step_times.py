#!/usr/bin/env python3
"""GPU box: microseconds of EVERY batched step of a fixed-action episode (65 536 envs, both events), launch by launch
(HIP events around each launch; no host synchronisation inside the episode).
    python tools/ramp_step_times.py [action] [first step] [last step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
act = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 160
dev = torch.device("cuda:0")
N = 65536
env = PVDERVec(N, device=dev, seed=2024, events_spec=both_events_spec(), auto_reset=False)
env.reset()
a = torch.full((N,), act, dtype=torch.uint8, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(161)]
torch.cuda.synchronize()
ev[0].record()
for t in range(160):
    env.step(a, return_tensors=False)
    ev[t + 1].record()
torch.cuda.synchronize()
ts = [ev[t].elapsed_time(ev[t + 1]) * 1e3 for t in range(160)]
print("action %d, us per batched step (incl. ~2 us of event packets), steps %d..%d:" % (act, lo, hi - 1))
for t0 in range(lo, hi, 20):
    print("  %3d: " % t0 + " ".join("%5.0f" % x for x in ts[t0:min(t0 + 20, hi)]))
print("mean over the episode %.1f" % (sum(ts) / 160))
