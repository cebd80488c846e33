#!/usr/bin/env python3
"""GPU box: do the two kernel forms of the per-lane parameter override (a phase per lane / a lane per env) agree BIT FOR BIT?
Runs the 64-set golden case (40 steps) in two child processes (PVDER_LANE3=1, PVDER_LANE1=1) and compares every output.
    python tools/lane_forms_bitwise.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
    import numpy as np, torch
    import bridge_cases
    from gym_PVDER_amd.vec_env import PVDERVec
    dev = torch.device("cuda:0")
    name = "lane_params"
    c = bridge_cases.CASES[name]
    vg, si, acts, cc = bridge_cases.case_inputs(name)
    env = PVDERVec(c["N"], device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator="etdrk4",
                   lane_params=bridge_cases.lane_param_sets(name))
    env.reset()
    out = []
    for t in range(c["steps"]):
        env.step(torch.as_tensor(acts[t], dtype=torch.uint8, device=dev), return_tensors=False)
        out.append(torch.cat([env.out.flatten(), env.fstate.flatten()]).cpu().numpy().copy())
    np.save(sys.argv[2], np.stack(out))
    sys.exit(0)
import numpy as np
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
files = []
for form in ("PVDER_LANE3", "PVDER_LANE1"):
    f = os.path.join(ROOT, "gpurun_out", "lane_form_%s.npy" % form)
    env = dict(os.environ); env.pop("PVDER_LANE1", None); env.pop("PVDER_LANE3", None); env[form] = "1"
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", f], env=env)
    files.append(f)
a, b = np.load(files[0]), np.load(files[1])
same = np.array_equal(a.view(np.int32), b.view(np.int32))
d = np.abs(a - b)
print("phase per lane vs lane per env, %d steps x %d values: %s (max abs difference %.3e, first differing step %s)" % (
    a.shape[0], a.shape[1], "BIT-IDENTICAL" if same else "differ", d.max(), "-" if same else int(np.argmax(d.max(axis=1) > 0))))
sys.exit(0 if same else 1)
