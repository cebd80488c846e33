#!/usr/bin/env python
"""The reference's fixed-action ramps (calc_returns holds ONE action for a whole episode, PVDER_env.py:480-487) on the
device against the committed converged-bridge goldens (tests/golden/bridge_ramp_a*.npz): per-env-step rel-L2 of the
observation, and the time per batched step at 65 536 envs along the ramp (20-step blocks).

    python tools/ramp_check.py [--envs 65536] [--actions 1 2 3 4]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
from gym_PVDER_amd.vec_env import PVDERVec  # noqa: E402
import bridge_cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--actions", type=int, nargs="*", default=[1, 2, 3, 4])
    ap.add_argument("--cases", nargs="*", default=None, help="case names of tests/bridge_cases.py instead of --actions (e.g. ramp_a1_touch)")
    ap.add_argument("--off-range", default="integrate")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in (a.cases or ["ramp_a%d" % k for k in a.actions]):
        g = bridge_cases.load(name)
        c = bridge_cases.CASES[name]
        act = c["actions"][1]
        vg, si, acts, cc = bridge_cases.case_inputs(name)
        env = PVDERVec(c["N"], device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], off_range=a.off_range)
        env.reset()
        if "env_ids" in c:           # named envs of a larger batch: their tables, set explicitly
            env.set_event_tables(torch.from_numpy(vg.astype(np.float32)).to(dev), torch.from_numpy(si.astype(np.float32)).to(dev))
            env.reset()
        assert np.array_equal(env.ev_vg.cpu().numpy(), vg.astype(np.float32)) and np.array_equal(env.ev_si.cpu().numpy(), si.astype(np.float32))
        at = torch.full((c["N"],), act, dtype=torch.uint8, device=dev)
        got = []
        for t in range(c["steps"]):
            obs, _, _, _ = env.step(at)
            got.append(obs.cpu().numpy().astype(np.float64))
        got = np.stack(got)
        want = g["obs"]
        per = np.linalg.norm(got - want, axis=2) / np.linalg.norm(want, axis=2)            # [steps, N]
        ok = g["m_abs"].max(axis=1) < 0.999
        last = int(np.argmin(ok)) if not ok.all() else c["steps"]
        traj = np.linalg.norm((got - want)[:last]) / np.linalg.norm(want[:last])
        print("%s, action %d: steps compared %d (|m| < 0.999 in the bridge), trajectory rel-L2 %.2e, worst env step %.2e at step %d"
              % (name, act, last, traj, per[:last].max(), int(per[:last].max(axis=1).argmax())))
        if "env_ids" in c:
            if c["N"] <= 8:
                print("   env steps 118..128, per env: " + " | ".join(" ".join("%.1e" % v for v in per[t]) for t in range(118, 129)))
            else:
                w = per.max(axis=0)
                tr = np.linalg.norm((got - want).transpose(1, 0, 2).reshape(c["N"], -1), axis=1) / np.linalg.norm(want.transpose(1, 0, 2).reshape(c["N"], -1), axis=1)
                print("   worst env step per env, sorted: " + " ".join("%.1e" % v for v in np.sort(w)))
                print("   trajectory per env, sorted:     " + " ".join("%.1e" % v for v in np.sort(tr)))
                print("   max |m_a| per env (bridge), in the order of the worst env step: " + " ".join("%.3f" % v for v in g["m_abs"].max(axis=0)[np.argsort(w)]))
                print("   worst env step by 20-step block (all envs): " + " ".join("%.1e" % per[k:k + 20].max() for k in range(0, c["steps"], 20)))
                print("   90 %% quantile over envs of the per-step error, by 20-step block: " + " ".join("%.1e" % np.quantile(per[k:k + 20].max(axis=0), 0.9) for k in range(0, c["steps"], 20)))
            continue
        print("   worst env step by 20-step block: " + " ".join("%.1e" % per[k:k + 20].max() for k in range(0, c["steps"], 20)))
        print("   |iref| / limit at block ends:    " + " ".join("%.3f" % (g["iref_abs"][min(k + 19, c["steps"] - 1)].max() / 1.7316) for k in range(0, c["steps"], 20)))
        print("   |m_a| at block ends:             " + " ".join("%.3f" % g["m_abs"][min(k + 19, c["steps"] - 1)].max() for k in range(0, c["steps"], 20)))
        # timing at batch size
        N = a.envs
        env = PVDERVec(N, device=dev, seed=c["seed"], auto_reset=False, events_spec=cc["spec_dict"], off_range=a.off_range)
        env.reset()
        at = torch.full((N,), act, dtype=torch.uint8, device=dev)
        times = []
        for k in range(0, c["steps"], 20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                env.step(at, return_tensors=False)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 20 * 1e6)
            if times[-1] > 5000 and k + 20 < c["steps"]:
                print("   (stopping the timing run: %.0f us per step)" % times[-1]); break
        print("   us per batched step (%d envs) by 20-step block: " % N + " ".join("%.1f" % t for t in times) + "   mean %.1f" % np.mean(times), flush=True)


if __name__ == "__main__":
    main()
