#!/usr/bin/env python
"""GPU: how close the duty cycle of ordinary PVDER-v0 rollouts (65 536 envs, random actions, voltage + insolation events) comes to its
limit at env-step boundaries -- what a threshold on |m| may be without ever firing in normal operation.
    python tools/m_distribution.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd")); sys.path.insert(0, ROOT)
import torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec
dev = torch.device("cuda:0")
spec = default_events_spec(); spec["insolation"]["ENABLE"] = True
env = PVDERVec(65536, device=dev, seed=2024, events_spec=spec, auto_reset=True, off_range="held_modes")
env.reset()
gen = torch.Generator(device=dev); gen.manual_seed(1)
mx = torch.zeros(65536, device=dev)
q = []
for t in range(480):
    a = torch.randint(0, 5, (65536,), generator=gen, device=dev, dtype=torch.uint8)
    env.step(a, return_tensors=False)
    f = env.fstate
    m = (torch.complex(f[2], f[3]) + 6000.0 * torch.complex(f[4], f[5])).abs()
    mx = torch.maximum(mx, m)
    q.append(float(m.max()))
s = torch.sort(mx).values
print("max over 480 steps (3 episodes) per env: median %.4f  99 %% %.4f  99.9 %% %.4f  99.99 %% %.4f  max %.4f" % (
    float(s[32768]), float(s[int(0.99 * 65536)]), float(s[int(0.999 * 65536)]), float(s[int(0.9999 * 65536)]), float(s[-1])))
for thr in (0.95, 0.96, 0.97, 0.975, 0.98, 0.985, 0.99):
    print("  envs that ever exceed %.3f: %d ; steps (of 480) in which some env does: %d" % (thr, int((mx > thr).sum()), sum(1 for v in q if v > thr)))
