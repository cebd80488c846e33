#!/usr/bin/env python
"""Per-step wall times (one sync per step) of a batch held at a Q_ref: which steps are slow, and how slow."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from gym_PVDER_amd import _native
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda:0")
dbg = hasattr(_native.lib(), "pvder_debug_counters")
buf = (ctypes.c_ulonglong * 16)()
N = 65536
q = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
env = PVDERVec(N, device=dev, seed=3, auto_reset=False, events_spec=both_events_spec())
env.reset(); env.fstate[23].fill_(q)
a = torch.zeros(N, dtype=torch.uint8, device=dev)
for _ in range(20): env.step(a, return_tensors=False)
rows = []
for t in range(60):
    torch.cuda.synchronize()
    if dbg: _native.lib().pvder_debug_counters(buf, 1)
    t0 = time.perf_counter(); env.step(a, return_tensors=False); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    if dbg: _native.lib().pvder_debug_counters(buf, 1)
    rows.append((dt, buf[1] + buf[3], buf[7], buf[8]))
for t, r in enumerate(rows):
    print("step %2d %7.1f us  deferred lanes %d  2nd evals %d  3rd evals %d" % ((t,) + r))
