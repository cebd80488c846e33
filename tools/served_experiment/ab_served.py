#!/usr/bin/env python
"""GPU A/B (round 6, VERDICT r5 #5 option B -- "a resident step server"): the SERVED rollout (pvder_plan_rollout_served /
PVDERVec.rollout_served: the fused loop's register-resident state, every env step gated by the caller's posts) against one launch
per env step (the headline's path) and the open-loop fused rollout, on the headline's workload (65 536 envs, events on, random
actions, auto-reset), 4 000 steps each, with bit-identity of the final state / counters / outputs / statistics.
    python tools/served_experiment/ab_served.py > profiles/r06_ab_served_rollout.txt
  ahead    every row posted before the launch             (what the gate costs the loop itself)
  depth D  row k posted when every workgroup has reported step k - D done   (a producer D steps behind the env, host in the loop)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd"]
import torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec

dev = torch.device("cuda:0")
N, K = 65536, 4000


def make():
    s = default_events_spec()
    for k in s:
        s[k]["ENABLE"] = True
    env = PVDERVec(N, device=dev, seed=2024, events_spec=s, auto_reset=True)
    env.reset()
    env._served_alloc()                  # (the protocol's pinned words: not part of what is timed)
    return env


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K


gen = torch.Generator(device=dev); gen.manual_seed(77)
acts = torch.randint(0, 5, (K, N), generator=gen, device=dev, dtype=torch.uint8)
warm = make(); warm.rollout_served(acts[:8]); warm.rollout(acts[:8], fused=True); del warm       # (code objects loaded: not what is timed)
ref = make()
print("one launch per env step (the headline's path)   %7.3f us per step" % timed(lambda: ref.rollout(acts, fused=False)))
fz = make()
print("fused rollout, open loop (actions known ahead)   %7.3f us per step" % timed(lambda: fz.rollout(acts, fused=True)))
for depth in (None, 16, 8, 4, 2, 1):
    env = make()
    us = timed(lambda: env.rollout_served(acts, depth=depth))
    same = all(torch.equal(getattr(env, n), getattr(ref, n)) for n in ("fstate", "istate", "out", "stats"))
    print("served rollout, %-9s                        %7.3f us per step   bit-identical to single launches: %s" % (
        "ahead" if depth is None else "depth %d" % depth, us, same), flush=True)
