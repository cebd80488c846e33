#!/usr/bin/env python
"""GPU: the action-4 episode (Vdc_ref lowered every env step: PVDER_env.py:218-221, :229, :480-487) on the device against the converged
event-locating golden, per `off_range` mode -- "integrate" (the hybrid kernel when an env comes near the duty-cycle limit),
"integrate_always" (the hybrid kernel behind every step launch), "held_modes" (rounds 3-5's held operator sets) -- with the time per step.
    python tools/a4_hybrid_numbers.py [modes ...]"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd", ROOT + "/tests"]
import numpy as np, torch
import bridge_cases
from helpers import rel_l2
from gym_PVDER_amd.vec_env import PVDERVec
dev = torch.device("cuda:0")
g = bridge_cases.load("ramp_a4"); c = bridge_cases.CASES["ramp_a4"]; vg, si, acts, cc = bridge_cases.case_inputs("ramp_a4")
want = g["obs"]
sat = g["m_abs"].max(axis=1) > 0.999; last = int(np.argmax(sat))
for mode in (sys.argv[1:] or ["integrate_always", "integrate", "held_modes"]):
    env = PVDERVec(c["N"], device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], off_range=mode); env.reset()
    a = torch.full((c["N"],), 4, dtype=torch.uint8, device=dev); got = []; ms = []
    for t in range(160):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        obs, _, _, _ = env.step(a)
        torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t0))
        got.append(obs.cpu().numpy().astype(np.float64))
    got = np.stack(got)
    e_all = rel_l2(got, want, axis=2)
    e = e_all[last:]
    print("%-17s first saturated step %d; before it max %.2e; beyond it per env max %s median %.2e; ms per step (N = %d, synchronous): before %.3f beyond %.3f max %.3f; nonfinite %d" % (
        mode, last, e_all[:last].max(), " ".join("%.1e" % v for v in e.max(axis=0)), np.median(e), c["N"], np.median(ms[5:last]), np.median(ms[last:]), max(ms[last:]),
        env.step_stats()["n_nonfinite"]), flush=True)
    print("   by 10-step block beyond: " + " ".join("%.1e" % e[i:i + 10].max() for i in range(0, len(e), 10)), flush=True)
