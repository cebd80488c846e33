#!/usr/bin/env python
"""CPU: how far the observations of the seven golden trajectories move under the OTHER resolutions of the model questions
that /root/reference cannot answer (oracle/pvder_model.py::ModelOptions) -- the uncertainty band of an
unpinned oracle.  Same cases, event tables and action scripts as tests/golden/make_goldens.py, integrated by
SciPy LSODA at rtol = atol = 1e-6 on the oracle's Python right-hand side with the option flipped; distance = rel-L2 of the
observation trajectory to the committed golden (the shipped resolution; its own LSODA run agrees with it to <= 2.7e-6).

    python tools/oracle_uncertainty.py [U1 U5 U7 ...]     -> profiles/r04_oracle_uncertainty.txt
"""
import multiprocessing as mp
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_goldens as mg  # noqa: E402
from oracle import pvder_model as pm  # noqa: E402

VARIANTS = {
    "shipped": {},
    "U1": dict(winv_is_pll=False),           # cross-coupling frequency = grid frequency instead of the PLL's
    "U5": dict(ppv_uses_vdc=False),          # Ppv at the MPP voltage instead of the instantaneous Vdc
    "U7": dict(ratio_is_vbase=True),         # transformer ratio Vgridrated/Vbase
    "U8": dict(wte0=6.28),                   # PLL angle after reset as config_der.json:18 prints it (a rounded 2 pi)
}


def run(job):
    vname, case = job
    name, script, ev_kind, n, steps, goal, discrete = case
    opts = pm.ModelOptions(**VARIANTS[vname])
    g = np.load(os.path.join(mg.HERE, "traj_%s.npz" % name))
    table = pm.make_event_table(mg.events_spec(ev_kind), 100.0, counter_key=(777, sum(map(ord, name)) % 1000, 1))
    assert np.array_equal(table.vgrid, g["vgrid"]) and np.array_equal(table.sinsol, g["sinsol"])
    try:
        env = mg.TightOracle(n_sim_time_steps_per_env_step=n, DISCRETE_REWARD=discrete, goals_list=[goal], opts=opts)
        obs = [env.reset(event_table=table)]
        rew = []
        mmax = 0.0
        for a in g["actions"]:
            o, r, d, _ = env.step(int(a))
            obs.append(o); rew.append(r)
            al = pm.algebraic(env.model, env.y, env.k_sub / 60.0, *env.events.at(env.k_sub / 60.0, True), env.Q_ref, env.Vdc_ref)
            mmax = max(mmax, abs(al.m[0]))
    except Exception as exc:            # (no steady state / LSODA gives up: part of the answer)
        return vname, name, None, repr(exc)[:120]
    A, B = np.array(obs), g["obs"]
    per_obs = np.linalg.norm(A - B, axis=0) / np.maximum(np.linalg.norm(B, axis=0), 1e-300)
    return vname, name, dict(traj=float(np.linalg.norm(A - B) / np.linalg.norm(B)),
                             step=float((np.linalg.norm(A - B, axis=1) / np.linalg.norm(B, axis=1)).max()),
                             reset=float(np.linalg.norm(A[0] - B[0]) / np.linalg.norm(B[0])),
                             rew=int((np.array(rew) != g["reward"]).sum()) if discrete else None,
                             worst_obs=int(np.argmax(per_obs[:10])), worst_obs_rel=float(per_obs[:10].max()), mmax=mmax), ""


def main():
    which = sys.argv[1:] or list(VARIANTS)
    jobs = [(v, c) for v in which for c in mg.CASES]
    with mp.Pool(int(os.environ.get("UNC_PROCS", "6"))) as pool:
        res = pool.map(run, jobs, chunksize=1)
    names = ["ia.real", "ia.imag", "va.real", "va.imag", "P_PCC", "Q_PCC", "Vdc", "Ppv", "Vdc_ref", "Q_ref"]
    lines = ["# rel-L2 distance of the observation trajectory to the committed golden (shipped resolution), by option flipped",
             "# variant case: whole trajectory | worst env step | reset observation | discrete rewards that differ | most moved observation | max |m_a|"]
    for v, n, r, err in res:
        if r is None:
            lines.append("%-8s %-18s FAILED: %s" % (v, n, err))
        else:
            lines.append("%-8s %-18s traj %.2e  step %.2e  reset %.2e  rewards %s  %s %.2e  |m| %.3f" % (
                v, n, r["traj"], r["step"], r["reset"], r["rew"], names[r["worst_obs"]], r["worst_obs_rel"], r["mmax"]))
    out = "\n".join(lines)
    print(out)
    with open(os.path.join(ROOT, "profiles", "r04_oracle_uncertainty.txt"), "w") as f:
        f.write(out + "\n")


if __name__ == "__main__":
    main()
