#!/usr/bin/env python3
"""EVERY env of the batch at the duty-cycle limit (Vdc_ref held 9 % down, own grid-voltage / insolation events): milliseconds per
batched step of the step launch + the duty-cycle-limit kernel behind it (off_range="integrate"), and a checksum of the state for
A/B builds (PVDER_LIB_PATH) -- the per-env arithmetic must not depend on how the kernel spreads envs over waves.
    python3 tools/hybrid_batch_rate.py [n_env] [steps]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=False)
env.reset()
env.fstate[24, :] = 1.1 * (1.0 - 0.09)
acts = torch.zeros((1, N), device=dev, dtype=torch.uint8)
ts = []
for k in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout(acts, fused=False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
f = env.fstate
m = (torch.complex(f[2], f[3]) + 6000.0 * torch.complex(f[4], f[5])).abs()
h = hashlib.sha1(f.cpu().numpy().tobytes()).hexdigest()[:12]
print("N %d: ms per batched step %s | envs with |m_a| > 0.99: %d | nonfinite %d | state sha %s" % (
    N, " ".join("%.2f" % t for t in ts), int((m > 0.99).sum()), int((~torch.isfinite(f)).sum()), h), flush=True)
