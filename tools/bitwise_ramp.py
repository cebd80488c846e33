#!/usr/bin/env python3
"""GPU box: are two builds of the library bit-identical along the fixed-action episodes?  Runs each action's 160-step episode
(8192 envs, events on) with the library named by PVDER_LIB_PATH and prints one checksum per step over the bit patterns of
the observation block and the state; tools/bitwise_ramp.py compare A.txt B.txt diffs two such logs.
    PVDER_LIB_PATH=tools/_ab/libold.so python tools/bitwise_ramp.py > gpurun_out/bits_old.txt
    python tools/bitwise_ramp.py > gpurun_out/bits_new.txt;  python tools/bitwise_ramp.py compare gpurun_out/bits_old.txt gpurun_out/bits_new.txt"""
import os, sys, time
if len(sys.argv) > 3 and sys.argv[1] == "compare":
    a, b = [open(f).read().split("\n") for f in sys.argv[2:4]]
    a, b = [[l for l in x if l.startswith("action")] for x in (a, b)]
    bad = [(x, y) for x, y in zip(a, b) if x.split("us")[0].rsplit(" ", 2)[0] != y.split("us")[0].rsplit(" ", 2)[0]]
    print("%d lines compared, %d differ" % (min(len(a), len(b)), len(bad)))
    for x, y in bad[:10]:
        print("  ", x, "\n  ", y)
    for act in (1, 2, 4):
        ta = [float(l.split()[-2]) for l in a if l.startswith("action %d" % act)]
        tb = [float(l.split()[-2]) for l in b if l.startswith("action %d" % act)]
        if ta and tb:
            blk = lambda t: " ".join("%.1f" % (sum(t[k:k + 20]) / 20) for k in range(0, 160, 20))
            print("action %d us per step by 20-step block:  A %s  mean %.1f\n%s B %s  mean %.1f" % (act, blk(ta), sum(ta) / len(ta), " " * 40, blk(tb), sum(tb) / len(tb)))
    sys.exit(1 if bad or len(a) != len(b) else 0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "65536"))
for act in (1, 2, 4):
    env = PVDERVec(N, device=dev, seed=2024, auto_reset=False, events_spec=both_events_spec())
    env.reset()
    a = torch.full((N,), act, dtype=torch.uint8, device=dev)
    for t in range(160):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = env.step(a, return_tensors=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
        ob = env.out.view(torch.int32).to(torch.int64).sum().item()
        st = env.fstate.view(torch.int32).to(torch.int64).sum().item()
        print("action %d step %3d obs %d state %d  %.1f us" % (act, t, ob, st, dt), flush=True)
