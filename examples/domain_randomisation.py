#!/usr/bin/env python
"""Domain randomisation: every env of the batch with its OWN DER parameters (config_der.json:2-21 -- DC-link capacitance, the outer
PI gains, the number of PV strings, an insolation scale), each started at its own steady state, all stepped by one launch per env
step on the matrix-core path (the reference would build one SolarPV_DER per parameter set, PVDER_env.py:366-398).

    python examples/domain_randomisation.py            (needs an MI355X and the built library)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))

import torch  # noqa: E402

import gym_PVDER_amd as gym_PVDER  # noqa: E402
from gym_PVDER_amd import der_model  # noqa: E402

N = 65536
rng = np.random.default_rng(0)
base = der_model.DER_CONFIG["50"]
sets = {"C_actual": base["C_actual"] * rng.uniform(0.7, 1.4, N),          # 210 ... 420 uF
        "Kp_DC": base["Kp_DC"] * rng.uniform(0.8, 1.25, N), "Ki_DC": base["Ki_DC"] * rng.uniform(0.8, 1.25, N),
        "Kp_Q": base["Kp_Q"] * rng.uniform(0.8, 1.25, N), "Ki_Q": base["Ki_Q"] * rng.uniform(0.8, 1.25, N),
        "Np": rng.integers(10, 12, N).astype(np.float64),                 # 10 or 11 parallel strings
        "Sinsol_scale": rng.uniform(0.9, 1.0, N)}                         # what the env sees of the insolation events
vec = gym_PVDER.make_vec(N, lane_params=sets)
obs = vec.reset()
print("reset: Ppv of the batch %.3f ... %.3f pu (every env at its own operating point)" % (float(obs[:, 7].min()), float(obs[:, 7].max())))
for k in range(20):
    vec.step(torch.randint(0, 5, (N,), device=vec.device, dtype=torch.uint8))
torch.cuda.synchronize()
t0 = time.time()
for k in range(320):
    obs, reward, done, _ = vec.step(torch.randint(0, 5, (N,), device=vec.device, dtype=torch.uint8))
torch.cuda.synchronize()
dt = time.time() - t0
print("%d envs with %d distinct parameter sets x 320 steps in %.3f s = %.3g env-steps/s; last-step stats %s"
      % (N, N, dt, N * 320 / dt, vec.step_stats()))
