#!/usr/bin/env python
"""GPU: the etdlin64 kernel's acceptance rule (two consecutive step-doubling levels agree to ACCEPT x the tolerance) against
cost and accuracy: ms per batched step over 40 steps of the bench's workload, and the error against the converged bridge on the
128 sampled envs of tests/test_gpu_fullsize.py (8 steps).    python tools/l64_accept_sweep.py > profiles/r05_l64_accept_sweep.txt"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd", ROOT + "/tests"]
import numpy as np
import torch

from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec, per_observation_error, rel_l2
from oracle.bridge_env import BridgeVecEnv
from test_gpu_fullsize import sample_ids

dev = torch.device("cuda:0")
N = 65536


def timing(steps=40):
    env = PVDERVec(N, device=dev, integrator="etdlin64", seed=2024, events_spec=both_events_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(78)
    acts = torch.randint(0, 5, (steps + 4, N), generator=gen, device=dev, dtype=torch.uint8)
    for i in range(4):
        env.step(acts[i], return_tensors=False)
    per = []
    for i in range(4, steps + 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step(acts[i], return_tensors=False)
        torch.cuda.synchronize(); per.append(1e3 * (time.perf_counter() - t0))
    return per


_ref = {}


def accuracy(steps=8, seed=32):
    rng = np.random.default_rng(seed)
    ids = sample_ids(rng)
    spec = both_events_spec()
    spec["voltage"]["t_events_start"] = spec["insolation"]["t_events_start"] = 0.5
    spec["voltage"]["t_events_step"] = 0.5
    env = PVDERVec(N, device=dev, integrator="etdlin64", seed=77, events_spec=spec, auto_reset=False, rtol=1e-6, atol=1e-8)
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    acts = torch.randint(0, 5, (steps, N), generator=gen, device=dev, dtype=torch.uint8)
    sel = torch.as_tensor(ids, device=dev)
    got = []
    for t in range(steps):
        obs, _, _, _ = env.step(acts[t])
        got.append(obs[sel].cpu().numpy().astype(np.float64))
    if "want" not in _ref:
        ref = BridgeVecEnv(len(ids), n=15, method="rk45", rtol=1e-8, atol=1e-11)
        ref.reset(env.ev_vg[:, sel].cpu().numpy().astype(np.float64), env.ev_si[:, sel].cpu().numpy().astype(np.float64), ev_s0=env.ev_s0, ev_ds=env.ev_ds)
        _ref["want"] = np.stack([ref.step(acts[t][sel].cpu().numpy())[0] for t in range(steps)])
    got, want = np.stack(got), _ref["want"]
    flat = lambda a: a.transpose(1, 0, 2).reshape(len(ids), -1)
    return rel_l2(flat(got), flat(want), axis=1).max(), per_observation_error(got, want).max(), int((env.istate[3] != 0).sum())


print("# ACCEPT MINLEVEL | ms per batched step: mean, max, by step | worst rel-L2 of a sampled env's 8-step trajectory vs the bridge, worst observation entry")
for acc, minl in ((1.0, 1), (2.0, 1), (4.0, 1), (7.0, 1), (1e30, 1), (1e30, 2)):
    os.environ["PVDER_L64_ACCEPT"] = repr(acc); os.environ["PVDER_L64_MINLEVEL"] = str(minl)
    per = timing()
    err, pobs, nbad = accuracy()
    print("%-7g %d | %.3f %.3f | %s | %.2e %.2e" % (acc, minl, sum(per) / len(per), max(per), " ".join("%.2f" % p for p in per), err, pobs), flush=True)
