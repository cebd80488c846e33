#!/usr/bin/env python
"""How far off balance are the envs the etdlin step hands to its 23-state tail?  Holds one action (default 1: Q_ref up)
for 160 steps at 65 536 envs with events and, from step `first` on, reports per step: wall time, lanes the device's
balance test rejects, the largest relative imbalance, and |m| / m_limit.  A histogram over all those steps at the end."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda:0")
N = 65536
act = int(sys.argv[1]) if len(sys.argv) > 1 else 1
first = int(sys.argv[2]) if len(sys.argv) > 2 else 95
env = PVDERVec(N, device=dev, seed=3, auto_reset=False, events_spec=both_events_spec())
env.reset()
a = torch.full((N,), act, dtype=torch.uint8, device=dev)
c, s = -0.5, math.sqrt(3.0) / 2


def rot(z, sign):       # z * e^{sign j 120deg}
    return torch.stack((c * z[0] - sign * s * z[1], sign * s * z[0] + c * z[1]))


def imbalance(fs):
    out = []
    for k in (0, 2, 4):
        za, zb, zc = fs[k:k + 2], fs[6 + k:8 + k], fs[12 + k:14 + k]
        e = ((zb - rot(za, -1.0)) ** 2).sum(0) + ((zc - rot(za, 1.0)) ** 2).sum(0)
        out.append(e)
    i2, x2 = (fs[0:2] ** 2).sum(0), (fs[2:4] ** 2).sum(0)
    return torch.sqrt(out[0] / i2), torch.sqrt(out[1] / (x2 + 1e-30)), torch.sqrt(out[2])


edges = np.array([0, 2e-5, 5e-5, 1e-4, 2e-4, 5e-4, 1e-3, 1e-2, 1e9])
hist_i = np.zeros(len(edges) - 1, dtype=np.int64)
hist_u = np.zeros(len(edges) - 1, dtype=np.int64)
uedges = np.array([0, 2e-9, 5e-9, 1e-8, 2e-8, 5e-8, 1e-7, 1e-6, 1e9])
for t in range(160):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(a, return_tensors=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    if t < first:
        continue
    fs = env.fstate[:18].double()
    ri, rx, au = imbalance(fs)
    bad = (ri > 2e-5) | (rx > 2e-5) | (au > 2e-9)
    nb = int(bad.sum())
    hist_i += np.histogram(torch.maximum(ri, rx)[bad].cpu().numpy(), edges)[0]
    hist_u += np.histogram(au[bad].cpu().numpy(), uedges)[0]
    if nb and nb <= 40 and t % 4 == 1:
        print("   envs:", torch.nonzero(bad).flatten().tolist(), flush=True)
    print("step %3d %7.1f us  off balance %4d  max rel(i,x) %.2e  max abs u %.2e  Q_ref %.3f" %
          (t, dt, nb, float(torch.maximum(ri, rx).max()), float(au.max()), float(env.fstate[23].mean())), flush=True)
print("rel imbalance (i, x) of rejected lanes, bins", edges.tolist(), hist_i.tolist())
print("abs imbalance u of rejected lanes, bins", uedges.tolist(), hist_u.tolist())
