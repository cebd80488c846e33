#!/usr/bin/env python3
"""GPU box: the per-lane parameter override by batch size -- the phase-per-lane kernel against the lane-per-env kernel
(PVDER_LANE3=1 / PVDER_LANE1=1 force one): us per batched step.
    PVDER_LANE3=1 python tools/lane_params_sizes.py;  PVDER_LANE1=1 python tools/lane_params_sizes.py"""
import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+"/gym-solarpvder-environment_amd", ROOT+"/tests"]
import numpy as np, torch
from gym_PVDER_amd.vec_env import PVDERVec
from gym_PVDER_amd import der_model
from helpers import both_events_spec
dev=torch.device("cuda:0")
for N in (16, 1024, 16384, 24576, 32768, 49152, 65536):
    sets={"Np": np.full(N, 11.0)}
    env=PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=True, integrator="etdrk4", lane_params=sets)
    env.reset()
    a=torch.randint(0,5,(200,N),device=dev,dtype=torch.uint8)
    env.rollout(a, fused=False)
    ts=[]
    for rep in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter(); env.rollout(a, fused=False); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/200*1e6)
    print("N %6d: %s us per batched step" % (N, " ".join("%.1f"%t for t in ts)), flush=True)
