#!/usr/bin/env python
"""How reproducible is a momentary duty-cycle saturation (tests/bridge_cases.py ramp_a1_touch, env step 121)?  The bridge's
other integrators against the golden (RK45 rtol 1e-8): profiles/r03_touch_floor.txt.  CPU, ~2 minutes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, bridge_cases
from oracle.bridge_env import BridgeVecEnv
g = bridge_cases.load("ramp_a1_touch")
vg, si, acts, c = bridge_cases.case_inputs("ramp_a1_touch")
for label, kw in (("RK45 1e-6", dict(method="rk45", rtol=1e-6, atol=1e-9)), ("RK45 1e-10", dict(method="rk45", rtol=1e-10, atol=1e-13)), ("RK4 8192", dict(method="rk4", micro=8192))):
    for e in (1, 2):
        ref = BridgeVecEnv(1, n=15, **kw)
        ref.reset(vg[:, e:e + 1], si[:, e:e + 1], ev_s0=60, ev_ds=60)
        got = np.array([ref.step(np.full(1, 1))[0][0, :10].copy() for _ in range(130)])
        w = g["obs"][:130, e, :10]
        per = np.linalg.norm(got - w, axis=1) / np.linalg.norm(w, axis=1)
        print(label, "env", e, "steps 118..129:", " ".join("%.1e" % v for v in per[118:130]), flush=True)
