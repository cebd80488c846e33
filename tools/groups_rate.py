#!/usr/bin/env python3
"""GPU box: the benchmark batch (65 536 envs, one launch per env step) as G sub-batches of N/G envs on G streams --
the waves of ONE launch load, compute and store in phase (HBM idle while the ALUs work and the reverse:
tools/skeleton_probe.hip prices the two bursts at 2.4 us of a 7.8 us step); sub-batches that run free of each other
overlap one's bursts with another's arithmetic.  Same envs (event generator keyed by the global env id), same results.

    python tools/groups_rate.py [N] [K]     -> us per env step of the WHOLE batch for G = 1, 2, 4, 8; threads / HIP graph
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
spec = default_events_spec()
spec["insolation"]["ENABLE"] = True
gen = torch.Generator(device=dev)
gen.manual_seed(1)
acts_all = torch.randint(0, 5, (K, N), generator=gen, device=dev, dtype=torch.uint8)


def build(G):
    n = N // G
    envs = [PVDERVec(n, device=dev, seed=2024, events_spec=spec, auto_reset=True, env_id_base=g * n) for g in range(G)]
    acts = [acts_all[:, g * n:(g + 1) * n].contiguous() for g in range(G)]
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    for e in envs:
        e.reset()
    torch.cuda.synchronize()
    return envs, acts, streams


def run_threads(envs, acts, streams):
    def work(g):
        with torch.cuda.stream(streams[g]):
            envs[g].rollout(acts[g], fused=False)
    th = [threading.Thread(target=work, args=(g,)) for g in range(len(envs))]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return time.perf_counter() - t0, host


def run_graph(envs, acts, streams, reps=3):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                envs[i].rollout(acts[i], fused=False)
        for s in streams:
            cur.wait_stream(s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


ref = None
for G in (1, 2, 4, 8):
    envs, acts, streams = build(G)
    run_threads(envs, acts, streams)                  # warm
    el, host = run_threads(envs, acts, streams)
    out = torch.cat([e.out for e in envs], dim=1).clone()
    line = "G %d (%6d envs per launch): threads %.2f us per env step of the batch (host %.2f)" % (G, N // G, el / K * 1e6, host / K * 1e6)
    try:
        eg = run_graph(envs, acts, streams)
        line += "; one HIP graph of %d x %d launches: %.2f us per env step" % (G, K, eg / K * 1e6)
    except Exception as exc:
        line += "; graph failed: %r" % (exc,)
    # same envs, same actions, same number of steps: the groups' output blocks side by side == the single batch's
    if ref is None:
        ref = out
    else:
        line += "; == G 1 bit for bit: %s" % bool(torch.equal(ref, out))
    print(line, flush=True)
