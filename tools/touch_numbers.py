#!/usr/bin/env python
"""GPU: the momentary duty-cycle saturation of the Q_ref-up ramp (tests/bridge_cases.py ramp_a1_touch: the +2 % voltage event of env
step 121 rings |m| = 0.97 across its limit for microseconds) per `off_range` mode, against the converged-bridge golden.
    python tools/touch_numbers.py > profiles/r06_touch_numbers.txt"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd", ROOT + "/tests"]
import numpy as np, torch
import bridge_cases
from helpers import rel_l2
from gym_PVDER_amd.vec_env import PVDERVec
dev = torch.device("cuda:0")
name = "ramp_a1_touch"
g = bridge_cases.load(name); c = bridge_cases.CASES[name]; vg, si, acts, cc = bridge_cases.case_inputs(name)
for mode in ("integrate", "integrate_always", "held_modes"):
    env = PVDERVec(c["N"], device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], off_range=mode)
    env.reset()
    env.set_event_tables(torch.from_numpy(vg.astype(np.float32)).to(dev), torch.from_numpy(si.astype(np.float32)).to(dev))
    env.reset()
    a = torch.full((c["N"],), 1, dtype=torch.uint8, device=dev)
    got = []
    for t in range(c["steps"]):
        obs, _, _, _ = env.step(a)
        got.append(obs.cpu().numpy().astype(np.float64))
    per = rel_l2(np.stack(got), g["obs"], axis=2)
    ev = 121
    print("%-17s before the event %.2e; the event's step per env %s; steps %d-%d max %.2e; beyond max %.2e" % (
        mode, per[:ev].max(), " ".join("%.1e" % v for v in per[ev]), ev + 1, ev + 9, per[ev + 1:ev + 10].max(), per[ev + 10:].max()), flush=True)
