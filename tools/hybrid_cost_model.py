#!/usr/bin/env python3
"""What an env step at the duty-cycle limit costs ONE LANE, counted on the host build of csrc/pvder_sat_hybrid.hpp (-DPVDER_HYB_COUNT):
calls of etd_step (~9.3e3 device instructions: five field evaluations inside), of field outside it (~0.9e3), micro steps of settle
(~0.3e3), located events by kind -- per env step of the four action-4 episodes (tests/golden/bridge_ramp_a4.npz), and the errors
against the converged golden beside them (the same numbers tests/test_sat_hybrid_host.py asserts).
    python3 tools/hybrid_cost_model.py [extra g++ flags, e.g. -DPVDER_HYB_EVENT_CAP=24]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bridge_cases
import test_sat_hybrid_host as T

flags = sys.argv[1:]
so = os.path.join(ROOT, "oracle", "_build", "libsat_hybrid_count.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = os.path.join(ROOT, "oracle", "_build", "sat_hybrid_count.cpp")
open(src, "w").write('#define PVDER_HYB_COUNT 1\n#include "%s"\nnamespace pvder { namespace hyb { long g_count[8]; } }\n'
                     'extern "C" long *sat_hybrid_counts() { return pvder::hyb::g_count; }\n' % os.path.join(ROOT, "tests", "csrc", "sat_hybrid_host.cpp"))
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + T.CSRC] + flags + ["-o", so, src])
L = C.CDLL(so)
L.sat_hybrid_env_step_host.restype = C.c_int
L.sat_hybrid_counts.restype = C.POINTER(C.c_long)
cnt = L.sat_hybrid_counts()

g = bridge_cases.load("ramp_a4")
rows = []
orig = L.sat_hybrid_env_step_host
per_step = []
class Wrap:
    def __getattr__(self, k): return getattr(L, k)
    def sat_hybrid_env_step_host(self, *a):
        for i in range(8): cnt[i] = 0
        r = orig(*a)
        st = a[-1]
        per_step.append([cnt[i] for i in range(8)] + [st[0], st[1], st[2], st[3]])
        return r
W = Wrap()
allb, allc = [], []
for env in range(4):
    first = int(np.argmax(g["m_abs"][:, env] > 0.9995))
    per_step.clear()
    err, tot = T.run_episode(W, "ramp_a4", env, 160)
    ps = np.array(per_step, dtype=np.float64)
    cost = 9.3e3 * ps[:, 0] + 0.9e3 * (ps[:, 1] - 5 * ps[:, 0]) + 0.3e3 * ps[:, 9]
    b = slice(first, 160)
    allb.append(err[b]); allc.append(cost[b])
    print("env %d first %d | err before %.2e  beyond max %.2e at %d median %.2e | per env step beyond: etd_step mean %.0f max %.0f, events mean %.1f max %.0f "
          "(entry %.1f impact %.1f clamp %.1f freeze %.1f), micro mean %.0f max %.0f, capped %d | modelled instructions mean %.2e max %.2e (before: %.2e)" % (
              env, first, err[:first].max(), err[b].max(), first + int(err[b].argmax()), np.median(err[b]), ps[b, 0].mean(), ps[b, 0].max(),
              ps[b, 10].mean(), ps[b, 10].max(), ps[b, 4].mean(), ps[b, 5].mean(), ps[b, 6].mean(), ps[b, 7].mean(), ps[b, 9].mean(), ps[b, 9].max(),
              int(ps[:, 11].sum()), cost[b].mean(), cost[b].max(), cost[:first].mean()))
c = np.concatenate(allc)
print("all four: median beyond %.2e | modelled instructions per env step beyond: mean %.2e  p50 %.2e  p90 %.2e  p99 %.2e  max %.2e" % (
    np.median(np.concatenate(allb)), c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max()))
