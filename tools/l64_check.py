import os, sys, time
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"gym-solarpvder-environment_amd"), os.path.join(ROOT,"tests")]
import numpy as np, torch
from gym_PVDER_amd.vec_env import PVDERVec
from oracle.bridge_env import BridgeVecEnv
from helpers import both_events_spec, rel_l2
dev = torch.device("cuda:0")
N, steps = 64, 14
rng = np.random.default_rng(7)
acts = rng.integers(0, 5, size=(steps, N))
outs = {}
for integ in ("etdlin64",) + (("rk45",) if os.environ.get("L64_RK45") else ()):
    env = PVDERVec(N, device=dev, integrator=integ, seed=5, events_spec=both_events_spec(), auto_reset=False, rtol=1e-6, atol=1e-8)
    env.reset()
    got = []
    t0 = time.perf_counter()
    for t in range(steps):
        obs, r, d, _ = env.step(torch.as_tensor(acts[t], device=dev)); got.append(obs.cpu().numpy().astype(np.float64))
    torch.cuda.synchronize(); print(integ, "time per step %.1f ms" % ((time.perf_counter() - t0) / steps * 1e3), env.step_stats(), flush=True)
    outs[integ] = np.stack(got)
    vg, si = env.ev_vg.cpu().numpy().astype(np.float64), env.ev_si.cpu().numpy().astype(np.float64)
ref = BridgeVecEnv(N, n=15, method="rk45", rtol=1e-8, atol=1e-11)
ref.reset(vg, si, ev_s0=60, ev_ds=60)
want = np.stack([ref.step(acts[t])[0] for t in range(steps)])
for k, v in outs.items():
    per = rel_l2(v, want, axis=2)
    print(k, "vs bridge: trajectory max over envs %.2e, worst step %.2e; by step" % (rel_l2(v.transpose(1,0,2).reshape(N,-1), want.transpose(1,0,2).reshape(N,-1), axis=1).max(), per.max()), " ".join("%.1e" % x for x in per.max(axis=1)))
# timing at full size
for integ in ("etdlin64",):
    env = PVDERVec(65536, device=dev, integrator=integ, seed=2024, events_spec=both_events_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    a = torch.randint(0, 5, (40, 65536), generator=gen, device=dev, dtype=torch.uint8)
    for t in range(4): env.step(a[t], return_tensors=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(4, 40): env.step(a[t], return_tensors=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 36
    print(integ, "65536 envs: %.1f us per step = %.3e env-steps/s" % (dt * 1e6, 65536 / dt), env.step_stats(), flush=True)
    # a longer run across an auto-reset (160 steps)
    if dt < 0.01:
        t0 = time.perf_counter()
        for t in range(200): env.step(a[t % 40], return_tensors=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(integ, "200 steps incl. an episode end: %.1f us per step" % (dt * 1e6), env.step_stats(), int(env.istate[2].min()), int(env.istate[2].max()))
