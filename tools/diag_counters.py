#!/usr/bin/env python
"""Why envs leave the standard path: lane counts per reason from a -DPVDER_DEBUG_COUNTERS build of the library
(hipcc ... -DPVDER_DEBUG_COUNTERS -o tools/_ab/libdbg.so; PVDER_LIB_PATH=tools/_ab/libdbg.so python tools/diag_counters.py [q_ref])."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd import _native
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda:0")
L = _native.lib()
buf = (ctypes.c_ulonglong * 16)()
names = ["lanes run", "deferred by the standard pass", "careful lanes", "careful -> deferred", "careful tick trip", "careful step trip",
         "hybrid steps (lane)", "... 2nd evaluation", "... 3rd evaluation", "deferred AND careful by the standard pass",
         "lanes into the 23-state tail", "workgroups running the tail",
         "off-balance lanes carried by their positive-sequence component"]
N = 65536
if len(sys.argv) > 2 and sys.argv[1] == "ramp":        # one action held from the reset on: counters per step from step 115
    act = int(sys.argv[2])
    env = PVDERVec(N, device=dev, seed=3, auto_reset=False, events_spec=both_events_spec())
    env.reset()
    a = torch.full((N,), act, dtype=torch.uint8, device=dev)
    show = [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12]
    for t in range(160):
        L.pvder_debug_counters(buf, 1)
        env.step(a, return_tensors=False)
        torch.cuda.synchronize()
        L.pvder_debug_counters(buf, 1)
        if t >= (60 if act == 4 else int(os.environ.get("FROM", "115"))):
            print("step %3d" % t, "  ".join("%s %d" % (names[k][:28], buf[k]) for k in show), flush=True)
    sys.exit(0)
for q in [float(x) for x in sys.argv[1:]] or [1.0, -1.0]:
    env = PVDERVec(N, device=dev, seed=3, auto_reset=False, events_spec=both_events_spec())
    env.reset()
    env.fstate[23].fill_(q)
    a = torch.zeros((20, N), dtype=torch.uint8, device=dev)
    env.rollout(a, fused=False)
    torch.cuda.synchronize()
    L.pvder_debug_counters(buf, 1)
    for rep in range(3):
        t0 = time.perf_counter()
        env.rollout(a, fused=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        L.pvder_debug_counters(buf, 1)
        print("Q_ref %+.2f, steps %d..%d: %.1f us per step (debug build); per step:" % (q, 20 + 20 * rep, 39 + 20 * rep, dt * 1e6),
              {n: buf[k] / 20.0 for k, n in enumerate(names)}, flush=True)
