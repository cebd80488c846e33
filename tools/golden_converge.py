#!/usr/bin/env python
"""CPU: integrate a bridge case of tests/bridge_cases.py at a given RK45 tolerance, one process per env, and keep the
observation trajectory (+ seconds per env step) -- the convergence evidence behind the goldens of the regimes where the
right-hand side switches (duty-cycle limit, current limit): VERDICT r3 #3.

    python tools/golden_converge.py ramp_a4 1e-10 [1e-12 ...]      -> .scratch/conv/<case>_rtol<r>.npz
    python tools/golden_converge.py --compare ramp_a4 1e-8 1e-10 1e-12

    CONV_METHOD=filippov python tools/golden_converge.py ramp_a4 1e-8 1e-10 1e-12     (round 5: the event-locating
        integrator of oracle/pvder_bridge_filippov.h; files <case>_filippov_rtol<r>.npz)
    CONV_METHOD=filippov python tools/golden_converge.py --compare ramp_a4 1e-8 1e-10 1e-12 [--against-rk45]

atol = rtol * 1e-3 (the goldens' ratio: 1e-8 / 1e-11)."""
import multiprocessing as mp
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bridge_cases  # noqa: E402
from oracle import pvder_model as pm  # noqa: E402
from oracle.bridge_env import BridgeVecEnv  # noqa: E402

OUT = os.path.join(ROOT, ".scratch", "conv")
METHOD = os.environ.get("CONV_METHOD", "rk45")
ATOL_RATIO = float(os.environ.get("CONV_ATOL_RATIO", "1e-3"))     # (filippov goldens: 1e-7 -- u_p ~ 1e-6 carries the duty cycle through Kp_GCC = 6000)


def fname(case, rtol, method=None):
    method = method or METHOD
    tag = "" if ATOL_RATIO == 1e-3 or method == "rk45" else "_ar%g" % ATOL_RATIO
    return os.path.join(OUT, "%s_%srtol%g%s.npz" % (case, "" if method == "rk45" else method + "_", rtol, tag))


def run_env(job):
    case, rtol, e = job
    vg, si, acts, c = bridge_cases.case_inputs(case)
    m = pm.build_model("50")
    ref = BridgeVecEnv(1, n=c["n"], method=METHOD, rtol=rtol, atol=rtol * ATOL_RATIO)
    ref.reset(vg[:, e:e + 1], si[:, e:e + 1], ev_s0=60, ev_ds=60)
    obs, secs, mabs, Ys = [], [], [], []
    for k, a in enumerate(acts[:, e]):
        t0 = time.time()
        o, r, d = ref.step(np.array([a]))
        secs.append(time.time() - t0)
        obs.append(o[0].copy())
        Y = ref.Y[0]
        Ys.append(Y.copy())
        mabs.append(abs(complex(Y[2] + m.Kp_GCC * Y[4], Y[3] + m.Kp_GCC * Y[5])))
        if k % 10 == 9:
            print("  %s rtol %g env %d step %d: %.1f s for the last one" % (case, rtol, e, k + 1, secs[-1]), flush=True)
    return e, np.array(obs), np.array(secs), np.array(mabs), np.array(Ys)


def compare(case, rtols):
    g = {r: dict(np.load(fname(case, r))) for r in rtols if os.path.exists(fname(case, r))}
    if 1e-8 in rtols and 1e-8 not in g and METHOD == "rk45":
        gg = bridge_cases.load(case)
        g[1e-8] = {"obs": gg["obs"], "m_abs": gg["m_abs"]}
    rs = sorted(g, reverse=True)
    ref = g[rs[-1]]["obs"]
    print("# %s: rel-L2 of the 11 observations per env step against rtol %g, worst env; by 10-step block" % (case, rs[-1]))
    for r in rs[:-1]:
        o = g[r]["obs"]
        err = np.linalg.norm(o - ref, axis=2) / np.linalg.norm(ref, axis=2)        # [steps, N]
        w = err.max(axis=1)
        print("rtol %-6g whole-episode worst step %.2e; blocks: %s" % (r, w.max(), " ".join("%.1e" % w[i:i + 10].max() for i in range(0, len(w), 10))))
    m = g[rs[-1]].get("m_abs")
    if m is not None:
        print("|m| max over envs by block: " + " ".join("%.4f" % m[i:i + 10].max() for i in range(0, m.shape[0], 10)))


def against_rk45(case, rtols):
    """The raw right-hand side's RK45 runs against the tightest event-locating run: do they APPROACH it as their tolerance falls?"""
    fil = dict(np.load(fname(case, min(rtols), "filippov")))["obs"]
    print("# %s: RK45 on the raw right-hand side against the event-locating run at rtol %g; worst env, by 10-step block" % (case, min(rtols)))
    for r in (1e-8, 1e-10, 1e-12):
        if r == 1e-8:
            gg = bridge_cases.load(case)
            o = gg["obs_coarser"] if False else None
        p = fname(case, r, "rk45")
        if not os.path.exists(p):
            continue
        o = dict(np.load(p))["obs"]
        err = np.linalg.norm(o - fil, axis=2) / np.linalg.norm(fil, axis=2)
        w = err.max(axis=1)
        print("rk45 rtol %-6g worst step %.2e; blocks: %s" % (r, w.max(), " ".join("%.1e" % w[i:i + 10].max() for i in range(0, len(w), 10))))


def main():
    if sys.argv[1] == "--compare":
        if sys.argv[-1] == "--against-rk45":
            return against_rk45(sys.argv[2], [float(x) for x in sys.argv[3:-1]])
        return compare(sys.argv[2], [float(x) for x in sys.argv[3:]])
    case = sys.argv[1]
    rtols = [float(x) for x in sys.argv[2:]]
    os.makedirs(OUT, exist_ok=True)
    N = bridge_cases.CASES[case]["N"]
    jobs = [(case, r, e) for r in rtols for e in range(N)]
    t0 = time.time()
    with mp.Pool(min(len(jobs), int(os.environ.get("CONV_PROCS", "4")))) as pool:
        res = pool.map(run_env, jobs, chunksize=1)
    for i, r in enumerate(rtols):
        mine = sorted(res[i * N:(i + 1) * N], key=lambda x: x[0])
        np.savez_compressed(fname(case, r), obs=np.stack([x[1] for x in mine], axis=1), secs=np.stack([x[2] for x in mine], axis=1),
                            m_abs=np.stack([x[3] for x in mine], axis=1), Y=np.stack([x[4] for x in mine], axis=1), rtol=r)
        print("%s rtol %g: %.0f s of CPU" % (case, r, sum(x[2].sum() for x in mine)))
    print("%.0f s wall" % (time.time() - t0))


if __name__ == "__main__":
    main()
