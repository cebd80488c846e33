#!/usr/bin/env python
"""A/B of two builds of the library on the deferral scenarios (debug tool, GPU box only).

  python tools/ab_deferral.py run <lib.so> <out.npz>     one process per library (a process loads one)
  python tools/ab_deferral.py cmp <a.npz> <b.npz>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))


def rel(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-30)


def run(lib, out):
    import torch
    from gym_PVDER_amd import _native
    _native.LIB_PATH = lib
    from gym_PVDER_amd.vec_env import PVDERVec
    dev = torch.device("cuda", 0)
    res = {}
    # saturation scenario (tests/test_gpu_parity.py::test_duty_cycle_saturation_anti_windup)
    N = 64
    vdc = np.linspace(0.975, 0.99, N)
    for name, integ, micro, knob in (("sat_rk4", "rk4", 4096, 0), ("sat_etd", "etdrk4", 8, 0), ("sat_etd_full", "etdrk4", 8, 1)):
        env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=4, integrator=integ, micro_steps=micro, seed=5,
                       auto_reset=False)
        env._debug_knob = knob
        env.reset()
        env.fstate[18].copy_(torch.as_tensor(vdc, dtype=env.fstate.dtype, device=dev))
        env.fstate[24].copy_(torch.as_tensor(vdc, dtype=env.fstate.dtype, device=dev))
        outs = []
        a = torch.zeros(N, dtype=torch.uint8, device=dev)
        for t in range(3):
            obs, _, _, _ = env.step(a)
            outs.append(obs.cpu().numpy().astype(np.float64))
        res[name] = np.stack(outs)
    # unbalanced-only scenario (no saturation): smooth, so builds must agree to rounding
    N = 256
    for name, knob in (("unb_fast", 0), ("unb_full", 1)):
        env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=4, integrator="etdrk4", seed=41, auto_reset=False)
        env._debug_knob = knob
        env.reset()
        env.fstate[6, 1::2] *= 1.001
        gen = torch.Generator(device=dev); gen.manual_seed(3)
        outs = []
        for t in range(4):
            a = torch.randint(0, 5, (N,), generator=gen, device=dev, dtype=torch.uint8)
            obs, _, _, _ = env.step(a)
            outs.append(obs.cpu().numpy().astype(np.float64))
        res[name] = np.stack(outs)
    np.savez(out, **res)
    for k in ("sat_etd", "sat_etd_full"):
        print(k, "vs sat_rk4 per step:", [float(rel(res[k][t], res["sat_rk4"][t]).max()) for t in range(3)])
    print("sat_etd vs sat_etd_full:", [float(rel(res["sat_etd"][t], res["sat_etd_full"][t]).max()) for t in range(3)])
    print("unb_fast vs unb_full:", [float(rel(res["unb_fast"][t], res["unb_full"][t]).max()) for t in range(4)])


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        print(k, "A vs B per step:", [float(rel(A[k][t], B[k][t]).max()) for t in range(A[k].shape[0])])


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        cmp(sys.argv[2], sys.argv[3])
