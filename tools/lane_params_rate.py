#!/usr/bin/env python3
"""GPU box: the per-lane parameter override of the ETDRK4 integrator (pvder_plan_lane_params) -- accuracy against the
bridge's 64 parameter sets (tests/golden/bridge_lane_params.npz) and microseconds per batched step at 65 536 envs, every
env with its own set, beside the plain ETDRK4 and etdlin steps.
    python tools/lane_params_rate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import bridge_cases
from helpers import rel_l2, both_events_spec
from gym_PVDER_amd.vec_env import PVDERVec
from gym_PVDER_amd import der_model
dev = torch.device("cuda:0")
name = "lane_params"
g, c = bridge_cases.load(name), bridge_cases.CASES[name]
vg, si, acts, cc = bridge_cases.case_inputs(name)
for integ in ("etdrk4", "etdlin"):
    env = PVDERVec(c["N"], device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], integrator=integ,
                   lane_params=bridge_cases.lane_param_sets(name))
    env.reset()
    got = []
    for t in range(c["steps"]):
        obs, rew, done, _ = env.step(torch.as_tensor(acts[t], dtype=torch.uint8, device=dev))
        got.append(obs.cpu().numpy().astype(np.float64))
    got, want = np.stack(got)[:, :, :10], g["obs"][:, :, :10]
    per = rel_l2(got, want, axis=2)
    traj = rel_l2(got.transpose(1, 0, 2).reshape(c["N"], -1), want.transpose(1, 0, 2).reshape(c["N"], -1), axis=1)
    print("%s: 64 parameter sets in one launch vs the bridge (40 steps, random actions): worst env step %.2e (median over envs of the worst %.2e), trajectory max %.2e"
          % (integ, per.max(), np.median(per.max(axis=0)), traj.max()))
N = 65536
rng = np.random.default_rng(1)
base = der_model.DER_CONFIG["50"]
sets = {"C_actual": base["C_actual"] * rng.uniform(0.7, 1.4, N), "Kp_DC": base["Kp_DC"] * rng.uniform(0.8, 1.25, N),
        "Ki_DC": base["Ki_DC"] * rng.uniform(0.8, 1.25, N), "Kp_Q": base["Kp_Q"] * rng.uniform(0.8, 1.25, N),
        "Ki_Q": base["Ki_Q"] * rng.uniform(0.8, 1.25, N), "Np": rng.integers(10, 12, N).astype(np.float64),
        "Sinsol_scale": rng.uniform(0.9, 1.0, N)}
a = torch.randint(0, 5, (40, N), device=dev, dtype=torch.uint8)
for label, kw in (("etdlin + lane parameters (65 536 distinct sets)", dict(integrator="etdlin", lane_params=sets)),
                  ("... 6 Krogstad steps per env step (twice the spread)", dict(integrator="etdlin", lane_params=sets, _lane_steps=6)),
                  ("... 8 Krogstad steps per env step (three times)", dict(integrator="etdlin", lane_params=sets, _lane_steps=8)),
                  ("etdrk4 + lane parameters (65 536 distinct sets)", dict(integrator="etdrk4", lane_params=sets)),
                  ("etdrk4, one set (balanced fast path)", dict(integrator="etdrk4")), ("etdlin, one set (default)", dict())):
    steps_override = kw.pop("_lane_steps", None)
    env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=True, **kw)
    if steps_override:
        env.lane_steps = steps_override
    env.reset()
    env.rollout(a, fused=False)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout(a, fused=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e6)
    print("%-50s %s us per batched step (65 536 envs)" % (label, " ".join("%.1f" % t for t in ts)), flush=True)
