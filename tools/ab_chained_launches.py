#!/usr/bin/env python
"""GPU A/B: does ordering consecutive env steps per WORKGROUP instead of per launch hide the launch floor?
65 536 envs, events on, random actions, auto-reset, one launch per env step, 4000 steps:
  w1       step_etdlin_kernel_w1 on one stream (the shipped path)
  w2       step_etdlin_kernel (two waves per SIMD) on one stream (PVDER_NO_W1=1)
  chained  step_etdlin_kernel_chain on two alternating streams, workgroup j of step t+1 waiting for workgroup j of step t
           (PVDER_CHAIN=1; csrc step_etdlin_entry<.., CHAIN>)
  chain1s  the same kernel -- waits, fences and all -- on ONE stream (PVDER_CHAIN_ONE_STREAM=1): what the ordering machinery costs by itself
Each in its own process (the switches are read once); results compared bit for bit.  The experiment is NOT in the shipped library:
    hipcc ... -DPVDER_CHAIN_EXPERIMENT [-DPVDER_CHAIN_FENCE_ALL] -o gpurun_out/libpvder_chain.so csrc/*.hip     (tools/ab_variant.sh's flags)
    PVDER_LIB_PATH=$PWD/gpurun_out/libpvder_chain.so python tools/ab_chained_launches.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, ctypes, time
ROOT = %r
sys.path[:0] = [ROOT, ROOT + "/gym-solarpvder-environment_amd"]
import numpy as np, torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec
dev = torch.device("cuda:0")
N, K = 65536, 4000
spec = default_events_spec()
for k in spec: spec[k]["ENABLE"] = True
env = PVDERVec(N, device=dev, seed=2024, events_spec=spec, auto_reset=True)
env.reset()
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.randint(0, 5, (500, N), generator=gen, device=dev, dtype=torch.uint8)
for r in range(4): env.rollout(acts, fused=False)                  # warm: 2000 steps
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(K // 500): env.rollout(acts, fused=False)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / K)
to = 0
if os.environ.get("PVDER_CHAIN") == "1":
    env._lib.pvder_debug_chain_timeout.restype = ctypes.c_int
    env._lib.pvder_debug_chain_timeout.argtypes = [ctypes.c_void_p]
    to = env._lib.pvder_debug_chain_timeout(env._plan)
np.savez(sys.argv[1], fstate=env.fstate.cpu().numpy(), istate=env.istate.cpu().numpy(), out=env.out.cpu().numpy(), stats=env.stats.cpu().numpy())
print("RESULT %%.3f %%d %%d" %% (best, to, env.step_stats()["n_nonfinite"]))
''' % ROOT

res = {}
tmp = tempfile.mkdtemp()
for name, extra in (("w1", {}), ("w2", {"PVDER_NO_W1": "1"}), ("chained", {"PVDER_CHAIN": "1"}),
                    ("chain1s", {"PVDER_CHAIN": "1", "PVDER_CHAIN_ONE_STREAM": "1"})):
    env = dict(os.environ)
    for k in ("PVDER_NO_W1", "PVDER_CHAIN", "PVDER_CHAIN_ONE_STREAM"):
        env.pop(k, None)
    env.update(extra)
    f = os.path.join(tmp, name + ".npz")
    r = subprocess.run([sys.executable, "-c", CHILD, f], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if r.returncode != 0 or not line:
        print("%s FAILED rc %d\n%s" % (name, r.returncode, r.stderr[-1500:]))
        continue
    us, to, nf = line[0].split()[1:]
    res[name] = (float(us), int(to), int(nf), f)
import numpy as np
print(__doc__.split("Each in its own")[0])
base = np.load(res["w1"][3]) if "w1" in res else None
for name, (us, to, nf, f) in res.items():
    same = "-"
    if base is not None and name != "w1":
        d = np.load(f)
        same = "bit-identical to w1" if all(np.array_equal(d[k].view(np.int32), base[k].view(np.int32)) for k in ("fstate", "istate", "out", "stats")) else "DIFFERS from w1"
    print("%-8s %7.3f us per step (best of 3 x 4000 steps)   timeouts %d   non-finite %d   %s" % (name, us, to, nf, same))
