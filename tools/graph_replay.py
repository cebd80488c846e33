#!/usr/bin/env python
"""Eager launches vs HIP-graph replay of K captured steps (debug tool, GPU box only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec

dev = torch.device("cuda", 0)
spec = default_events_spec(); spec["insolation"]["ENABLE"] = True
N, K, R = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 250
env = PVDERVec(N, device=dev, events_spec=spec, seed=2024, auto_reset=True)
env.reset()
gen = torch.Generator(device=dev); gen.manual_seed(1234)
pool = torch.randint(0, 5, (R, K, N), generator=gen, device=dev, dtype=torch.uint8)
for r in range(20):
    for k in range(K):
        env.step(pool[r, k], return_tensors=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(R):
    for k in range(K):
        env.step(pool[r, k], return_tensors=False)
torch.cuda.synchronize()
e = time.perf_counter() - t0
print("eager : %.2f us per step, %.3e env-steps/s" % (e / (R * K) * 1e6, N * R * K / e))
static = pool[0].clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(K):
        env.step(static[k], return_tensors=False)
for r in range(20):
    static.copy_(pool[r]); g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(R):
    static.copy_(pool[r]); g.replay()
torch.cuda.synchronize()
e = time.perf_counter() - t0
print("graph : %.2f us per step, %.3e env-steps/s (K=%d steps per graph)" % (e / (R * K) * 1e6, N * R * K / e, K))
