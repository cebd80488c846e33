#!/usr/bin/env python
"""GPU: where the desynchronised batch's time goes (bench.py `desynchronised`): the same 65 536 envs, events on,
random actions, one launch per env step -- (a) in lockstep, (b) episode phases spread over 160 clocks with NO env ending its
episode inside the timed window (long episodes, events to the end), (c) spread with 160-step episodes (one env in 160 ends per step).
    python tools/desync_breakdown.py > profiles/r05_desync_breakdown.txt"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd"]
import torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec

dev = torch.device("cuda:0")
N = 65536


def spec(stop):
    s = default_events_spec()
    for k in s:
        s[k]["ENABLE"] = True
        s[k]["t_events_stop"] = stop
    return s


def run(label, max_sim_time, stop, spread, steps=300, sort=False, envvars=(), off_range="held_modes", lazy=None):
    for k in envvars:
        os.environ[k] = "1"
    env = PVDERVec(N, device=dev, seed=2024, events_spec=spec(stop), auto_reset=True, max_sim_time=max_sim_time,
                   off_range=off_range, lazy_events=lazy)     # (held modes: 500-step random walks reach the duty-cycle limit -- not what is timed here)
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(77)
    acts = torch.randint(0, 5, (max(steps, 160), N), generator=gen, device=dev, dtype=torch.uint8)
    if spread:
        phase = (torch.arange(N, device=dev, dtype=torch.int64) * 7919) % 160
        for k in range(160):
            env.step(acts[k], return_tensors=False)
            env.reset(phase == k)
    if sort:
        env.sort_lanes(lane_io=True)
    env.rollout(acts[:60], fused=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); env.rollout(acts[:steps], fused=False); e1.record()
    torch.cuda.synchronize()
    clocks = int(torch.unique(env.istate[0]).numel())
    print("%-78s %6.2f us per step   (%d distinct clocks at the end, %d episodes started)" % (
        label, e0.elapsed_time(e1) * 1e3 / steps, clocks, int(env.istate[2].sum().item())), flush=True)
    for k in envvars:
        del os.environ[k]


if os.environ.get("PVDER_NO_K4") == "1":      # (read once per process by the library: a run of its own)
    run("(g) = (d) with one schedule for every lane (PVDER_NO_K4=1)", 125.0, 124.0, True, sort=True)
    run("(h) = (a) with one schedule for every lane (PVDER_NO_K4=1)", 40.0, 39.0, False)
    sys.exit(0)
run("(a) lockstep, 160-step episodes", 40.0, 39.0, False)
run("(b) phases spread over 160 clocks, 500-step episodes (nobody ends in the window)", 125.0, 124.0, True)
run("(c) phases spread over 160 clocks, 160-step episodes (1 env in 160 ends per step)", 40.0, 39.0, True)
# round 6: the same batches behind PVDERVec.sort_lanes (homogeneous waves), and what the clock hint / the double schedule are worth
run("(d) = (b) behind sort_lanes()", 125.0, 124.0, True, sort=True)
run("(e) = (c) behind sort_lanes()", 40.0, 39.0, True, sort=True)
run("(f) = (a) without the clock hint (PVDER_NO_HINT)", 40.0, 39.0, False, envvars=("PVDER_NO_HINT",))
run("(j) = (e) with whole tables at the auto-reset (lazy_events=False: rounds 1-5)", 40.0, 39.0, True, sort=True, lazy=False)
run("(k) = (a) with the rows written as the episode advances (lazy_events=True)", 40.0, 39.0, False, lazy=True)
run("(a) again", 40.0, 39.0, False)
run("(k) again", 40.0, 39.0, False, lazy=True)
