"""CPU ORACLE (test infrastructure only) -- batched env semantics around the C bridge integrators.

Same transition logic as :class:`oracle.pvder_oracle_env.OraclePVDER` (which cites
PVDER_env.py line by line) for N envs at once, with the ODE advanced by oracle/pvder_bridge.c:
fp64/fp32 classic RK4 micro-stepping or fp64 Dormand-Prince RK45.  Used to check the HIP kernels
at sizes where per-env LSODA would be too slow; itself pinned against the LSODA oracle in
tests/test_oracle.py.  PARITY UNPINNED (see oracle/pvder_model.py).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import bridge
from . import pvder_model as pm
from .pvder_oracle_env import (DEL_QREF, DEL_VDCREF, MAX_EPISODE_STEPS, MIN_SIM_TIME, P_REF, Q_REF_TARGET,
                               SIM_TIME_STEP, discrete3)


class BridgeVecEnv:
    def __init__(self, num_envs: int, n: int = 15, max_sim_time: float = 40.0, discrete: bool = True,
                 goal: str = "voltage_regulation", der_id: str = "50", method: str = "rk4",
                 micro: int = 4096, dtype=np.float64, kahan: bool = False, rtol: float = 1e-8,
                 atol: float = 1e-10, overrides=None, opts=None):
        self.N, self.n = num_envs, int(n)
        self.model = pm.build_model(der_id, opts, overrides=overrides)
        self.dq = DEL_QREF * self.n / pm.SBASE
        self.dv = DEL_VDCREF * self.n / pm.VDCBASE
        cap = MAX_EPISODE_STEPS * self.n * SIM_TIME_STEP
        self.max_sim_time = min(max(float(max_sim_time), MIN_SIM_TIME), cap)
        self.max_ksub = int(np.ceil(self.max_sim_time * 60.0 - 1e-6))
        self.discrete, self.goal = discrete, goal
        self.method, self.micro, self.dtype, self.kahan = method, micro, dtype, kahan
        self.rtol, self.atol = rtol, atol

    def reset(self, vg_table: np.ndarray, si_table: np.ndarray, ev_s0: int = 60, ev_ds: int = 60):
        """tables: [K+1, N]; event j>=1 starts at sim time step ev_s0 + (j-1)*ev_ds."""
        m = self.model
        y0 = bridge.to_delta(pm.steady_state(m), 0.0)
        self.Y = np.tile(y0, (self.N, 1)).astype(self.dtype)
        self.comp = np.zeros_like(self.Y) if self.kahan else None
        self.qref = np.zeros(self.N)
        self.vdcref = np.full(self.N, m.Vdc_ref0)
        self.ksub = np.zeros(self.N, dtype=np.int64)
        self.steps = np.zeros(self.N, dtype=np.int64)
        self.done = np.zeros(self.N, dtype=bool)
        self.hstep = np.full(self.N, 1e-6)
        self.region = np.zeros((self.N, 3), dtype=np.int32)      # method "filippov": IN / OUT / SLIDE per phase
        self.n_events = np.zeros(self.N, dtype=np.int64)
        self.vg_t, self.si_t = np.asarray(vg_table, dtype=np.float64), np.asarray(si_table, dtype=np.float64)
        self.ev_s0, self.ev_ds = ev_s0, ev_ds
        self.reward = np.zeros(self.N)
        return self.observe()

    def _row(self, g: np.ndarray) -> np.ndarray:
        K = self.vg_t.shape[0] - 1
        j = np.where(g < self.ev_s0, 0, (g - self.ev_s0) // self.ev_ds + 1)
        return np.clip(j, 0, K)

    def _inputs(self, g: np.ndarray):
        r = self._row(g)
        e = np.arange(self.N)
        return self.vg_t[r, e], self.si_t[r, e]

    def step(self, actions: np.ndarray):
        a = np.asarray(actions)
        run = ~self.done
        self.qref = self.qref + np.where(run, np.where(a == 1, self.dq, 0.0) - np.where(a == 2, self.dq, 0.0), 0.0)
        self.vdcref = self.vdcref + np.where(run, np.where(a == 3, self.dv, 0.0) - np.where(a == 4, self.dv, 0.0), 0.0)
        idx = np.nonzero(run)[0]
        if len(idx):
            if self.comp is not None:
                self.comp[:] = 0                       # the kernel keeps the compensation per launch
            for s in range(self.n):
                g = self.ksub + s
                vg, si = self._inputs(g)
                t0 = g[idx] * SIM_TIME_STEP
                if self.method == "rk4":
                    comp = None
                    if self.comp is not None:
                        comp = np.ascontiguousarray(self.comp[idx])
                    Y = bridge.rk4_batch(self.model, self.Y[idx], t0, SIM_TIME_STEP / self.micro, self.micro,
                                         vg[idx], si[idx], self.qref[idx], self.vdcref[idx],
                                         dtype=self.dtype, comp=comp)
                    if comp is not None:
                        self.comp[idx] = comp
                elif self.method == "filippov":
                    reg = np.ascontiguousarray(self.region[idx])
                    Y, hs, _, _, nev = bridge.filippov_batch(self.model, self.Y[idx], t0, SIM_TIME_STEP, self.rtol, self.atol,
                                                             self.hstep[idx], vg[idx], si[idx], self.qref[idx],
                                                             self.vdcref[idx], reg)
                    self.region[idx] = reg
                    self.n_events[idx] += nev
                    self.hstep[idx] = hs
                else:
                    Y, hs, _, _ = bridge.rk45_batch(self.model, self.Y[idx], t0, SIM_TIME_STEP, self.rtol,
                                                    self.atol, self.hstep[idx], vg[idx], si[idx],
                                                    self.qref[idx], self.vdcref[idx])
                    self.hstep[idx] = hs
                self.Y[idx] = Y
            self.ksub[idx] += self.n
            self.steps[idx] += 1
        obs, vrms = self.observe(return_vrms=True)
        m = self.model
        rew = np.zeros(self.N)
        for e in range(self.N):
            if self.goal == "voltage_regulation":
                d = vrms[e] - m.Vrms_ref
                rew[e] = discrete3(abs(d) / abs(m.Vrms_ref), 0.01, 0.05) if self.discrete else -(d * d)
            elif self.goal == "power_regulation":
                pt = P_REF / pm.SBASE
                d = obs[e, 4] - pt
                rew[e] = discrete3(abs(d) / pt, 0.01, 0.03) if self.discrete else -(d * d)
            else:
                qt = Q_REF_TARGET / pm.SBASE
                d = obs[e, 5] - qt
                rew[e] = discrete3(abs(d) / abs(qt), 0.01, 0.05) if self.discrete else -(d * d)
        self.reward = np.where(run, rew, self.reward)
        self.done = self.done | (run & ((self.ksub >= self.max_ksub) | (self.steps >= MAX_EPISODE_STEPS)))
        return obs, self.reward.copy(), self.done.copy()

    def observe(self, return_vrms: bool = False):
        vg, si = self._inputs(self.ksub)
        obs = np.zeros((self.N, 11))
        vrms = np.zeros(self.N)
        for e in range(self.N):
            _, o = bridge.evaluate(self.model, self.Y[e].astype(np.float64), self.ksub[e] * SIM_TIME_STEP,
                                   vg[e], si[e], self.qref[e], self.vdcref[e])
            obs[e, :8] = o[:8]
            obs[e, 8], obs[e, 9] = self.vdcref[e], self.qref[e]
            obs[e, 10] = self.ksub[e] * SIM_TIME_STEP / self.max_sim_time
            vrms[e] = o[8]
        return (obs, vrms) if return_vrms else obs
