#!/usr/bin/env python3
"""GPU box: where the etdlin64 step's time goes -- the cost of ONE fp64 ETD3RK step of a wave (PVDER_L64_ACCEPT=1e30: every env
accepts at level 1, i.e. tick + 2 + 4 steps) against the default acceptance rule, by env step (quiet / behind an event).
    python tools/l64_profile.py [N]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
    import torch
    from gym_PVDER_amd.vec_env import PVDERVec
    from helpers import both_events_spec
    N = int(sys.argv[1])
    dev = torch.device("cuda:0")
    env = PVDERVec(N, device=dev, integrator="etdlin64", seed=2024, events_spec=both_events_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    a = torch.randint(0, 5, (24, N), generator=gen, device=dev, dtype=torch.uint8)
    ts = []
    for t in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step(a[t], return_tensors=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    print("N %6d ACCEPT %-6s MINLEVEL %s: us per env step, steps 0..23 (events start behind step 3, 7, 11, ...): %s" % (
        N, os.environ.get("PVDER_L64_ACCEPT", "1"), os.environ.get("PVDER_L64_MINLEVEL", "1"), " ".join("%.0f" % x for x in ts)), flush=True)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "65536"
    for envs in ({"PVDER_L64_ACCEPT": "1e30"}, {"PVDER_L64_ACCEPT": "1e30", "PVDER_L64_MINLEVEL": "3"}, {}):
        subprocess.run([sys.executable, os.path.abspath(__file__), N, "child"], env=dict(os.environ, **envs))
