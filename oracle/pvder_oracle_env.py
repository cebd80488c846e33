"""CPU ORACLE (test infrastructure only) -- the reference's ``PVDER.step`` path on SciPy odeint.

Restates, for ONE environment, the control flow of
``/root/reference/gym_PVDER/envs/PVDER_env.py``:

* ``step``        :138-196   (time bookkeeping, action, integrate, reward, done, obs)
* ``action_calc`` :198-229
* ``reward_calc`` :231-301
* ``state``       :531-542
* ``reset`` / ``setup_PVDER_simulation`` :316-334, :366-398
* ``generate_simulation_events`` :400-411, setters :561-620

with ``self.sim.run_simulation()`` (:166, third-party ``pvder`` -> ``scipy.integrate.odeint``)
replaced by :func:`oracle.pvder_model.rhs` integrated by the *same* SciPy LSODA driver
(``solverType='odeint'``, ``jacFlag=True``, PVDER_env.py:384-386).

PARITY UNPINNED (see ``oracle/pvder_model.py`` header): this is the build's restatement, used
as (1) the parity checker of the HIP kernels and (2) the ``cpu_baseline`` ("port") of bench.py.
Nothing in the product path imports it.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional

import numpy as np
from scipy.integrate import odeint

from . import pvder_model as pm

SIM_TIME_STEP = 1.0 / 60.0          # env_sim_spec['sim_time_step'], PVDER_env.py:63
MIN_SIM_TIME = 1.0                  # PVDER_env.py:65
MAX_EPISODE_STEPS = 500             # gym_PVDER/__init__.py:10
DEL_QREF = 25.0                     # PVDER_env.py:75 (VAR per sim time step)
DEL_VDCREF = 0.02                   # PVDER_env.py:75 (V per sim time step)
P_REF = 45.4e3                      # PVDER_env.py:69
Q_REF_TARGET = 5.5e3                # PVDER_env.py:69

DEFAULT_EVENTS_SPEC = {             # PVDER_env.py:60-61
    "insolation": {"t_events_start": 1.0, "t_events_stop": 39.0, "t_events_step": 1.0,
                   "min": 85.0, "max": 100.0, "ENABLE": False},
    "voltage": {"t_events_start": 1.0, "t_events_stop": 39.0, "t_events_step": 1.0,
                "min": 0.98, "max": 1.02, "ENABLE": True},
}

GOAL_REWARDS = {                    # 'required' lists, PVDER_env.py:78-93 via :448-452
    "voltage_regulation": ["voltage_error"],
    "power_regulation": ["power_error"],
    "Q_regulation": ["Q_error"],
}


def reward_terms(goal: str, discrete: bool, Vrms: float, Vrms_ref: float, S_PCC: complex, Vdc: float, Vdc_ref: float,
                 my_spec: Optional[List[str]] = None, Q_ref: float = 0.0):
    """``reward_calc`` (PVDER_env.py:231-301) on explicit model quantities: one entry per reward type
    of ``env_goal_spec[goal]['reward']['my_spec']`` (default: the goal's 'required' list, :448-452),
    summed (:301).  The targets are the ones ``reward_calc`` defines per goal (:237-245): under
    voltage_regulation ``Qtarget = PV_model.Q_ref`` (:238, argument ``Q_ref``), under Q_regulation the spec's
    ``Q_ref / Sbase`` (:241), under power_regulation ``Vdctarget = PV_model.Vdc_ref`` and ``Ptarget`` (:243-245);
    a reward type whose target the goal does not define is a NameError in the reference -- ValueError here."""
    rewards: List = []
    for rt in (my_spec if my_spec is not None else GOAL_REWARDS[goal]):   # :249
        if rt == "Vdc_error":                                           # :251-262
            if goal != "power_regulation":
                raise ValueError("Vdc_error needs Vdctarget, which reward_calc defines for power_regulation only")
            if discrete:
                rewards.append(discrete3(abs(Vdc - Vdc_ref) / Vdc_ref, 0.02, 0.05))
            else:
                rewards.append(-(Vdc_ref - Vdc) ** 2)
        elif rt == "voltage_error":                                     # :276-287
            if discrete:
                rewards.append(discrete3(abs(Vrms - Vrms_ref) / abs(Vrms_ref), 0.01, 0.05))
            else:
                rewards.append(-(Vrms - Vrms_ref) ** 2)
        elif rt == "Q_error":                                           # :263-275
            if goal == "power_regulation":
                raise ValueError("Q_error needs Qtarget, which reward_calc does not define for power_regulation")
            Qt = Q_ref if goal == "voltage_regulation" else Q_REF_TARGET / pm.SBASE      # :238 / :241
            if discrete:
                if Qt == 0.0:
                    Qt = 1e-6
                rewards.append(discrete3(abs(S_PCC.imag - Qt) / abs(Qt), 0.01, 0.05))
            else:
                rewards.append(-(S_PCC.imag - Qt) ** 2)
        elif rt == "power_error":                                       # :290-299
            if goal != "power_regulation":
                raise ValueError("power_error needs Ptarget, which reward_calc defines for power_regulation only")
            Pt = P_REF / pm.SBASE                                       # :245
            if discrete:
                rewards.append(discrete3(abs(S_PCC.real - Pt) / Pt, 0.01, 0.03))
            else:
                rewards.append(-(S_PCC.real - Pt) ** 2)
    return sum(rewards)                                                 # :301


def discrete3(e: float, lo: float, hi: float) -> int:
    """+1 / -1 / -5 ladder of PVDER_env.py:255-260, :268-273, :280-285, :292-297."""
    if e <= lo:
        return 1
    if e >= hi:
        return -5
    return -1


class OraclePVDER:
    """Single-env fp64 SciPy-odeint restatement of the reference env (old-gym 4-tuple API)."""

    def __init__(self, n_sim_time_steps_per_env_step: int = 15, max_sim_time: float = 40.0,
                 DISCRETE_REWARD: bool = True, goals_list=("voltage_regulation",),
                 der_id: str = "50", events_spec: Optional[dict] = None,
                 opts: Optional[pm.ModelOptions] = None, solver: str = "reference",
                 seed: Optional[int] = None):
        self.n = max(1, int(n_sim_time_steps_per_env_step))           # :603-614
        self._sim_time_per_env_step = SIM_TIME_STEP * self.n           # :616
        self._delQref = DEL_QREF * self.n                              # :617
        self._delVdcref = DEL_VDCREF * self.n                          # :618
        cap = MAX_EPISODE_STEPS * self._sim_time_per_env_step          # :569-571
        self.max_sim_time = min(max(float(max_sim_time), MIN_SIM_TIME), cap)
        self.DISCRETE_REWARD = bool(DISCRETE_REWARD)
        self.goals_list = list(goals_list)
        if not set(self.goals_list).issubset(GOAL_REWARDS):            # :582-586
            raise ValueError("invalid goal")
        self.model = pm.build_model(der_id, opts)
        self.events_spec = {k: dict(v) for k, v in (events_spec or DEFAULT_EVENTS_SPEC).items()}
        self.solver = solver        # "reference": pvder's loose LSODA settings; "tight": converged
        self.rng = random.Random(seed)
        self.nfev = 0
        self.y = None
        self._jac = None

    # -- reset (PVDER_env.py:316-334, 366-398) ------------------------------------------------
    def reset(self, event_table: Optional[pm.EventTable] = None, counter_key=None):
        m = self.model
        self.y = pm.steady_state(m)
        self.Q_ref = 0.0
        self.Vdc_ref = m.Vdc_ref0
        self.tStart = 0.0
        self.k_sub = 0                  # substep counter: t = k_sub/60 exactly
        self._steps = 0
        self._reward = 0
        self.done = False
        if event_table is not None:
            self.events = event_table
        else:
            self.events = pm.make_event_table(self.events_spec, m.Sinsol0, rng=self.rng,
                                              counter_key=counter_key)
        return np.array(self.state)

    # -- one env transition (PVDER_env.py:138-196) ---------------------------------------------
    def step(self, action: int):
        assert action in (0, 1, 2, 3, 4)                               # :201
        if self.done:                                                   # :145-152 stale tuple
            return np.array(self.state), self._reward, self.done, {}
        m = self.model
        self._steps += 1                                                # :156
        self.action_calc(action)                                        # :162
        self.integrate()                                                # :166
        self._reward = self.reward_calc()                               # :179
        self.tStart = self.k_sub * SIM_TIME_STEP                        # :180
        if self.tStart >= self.max_sim_time - 1e-9:                     # :183 (6-dp time grid)
            self.done = True
        return np.array(self.state), self._reward, self.done, {}

    def action_calc(self, action: int):                                 # :198-229
        if action == 1:
            self.Q_ref += self._delQref / pm.SBASE
        elif action == 2:
            self.Q_ref -= self._delQref / pm.SBASE
        elif action == 3:
            self.Vdc_ref += self._delVdcref / pm.VDCBASE
        elif action == 4:
            self.Vdc_ref -= self._delVdcref / pm.VDCBASE

    def f(self, y, t):
        self.nfev += 1
        vg, si = self.events.at(t, self.model.opts.event_inclusive)
        return pm.rhs(self.model, y, t, vg, si, self.Q_ref, self.Vdc_ref)

    def jac(self, y, t):
        """Analytic Jacobian (jacFlag=True, PVDER_env.py:386); finite differences inside an
        anti-windup branch, where the analytic (un-saturated) expression does not apply."""
        m = self.model
        vg, si = self.events.at(t, m.opts.event_inclusive)
        o = pm.algebraic(m, y, t, vg, si, self.Q_ref, self.Vdc_ref)
        if max(abs(mm) for mm in o.m) > m.m_limit or abs(o.iref[0]) > m.iref_limit:
            return pm.jac_fd(m, y, t, vg, si, self.Q_ref, self.Vdc_ref)
        if self._jac is None:
            self._jac = pm.analytic_jacobian(m)
        return self._jac(y, t, vg, si, self.Q_ref, self.Vdc_ref)

    def integrate(self):
        """n output intervals of 1/60 s (np.linspace of :158).  "reference" = one odeint call per
        env step with pvder's settings (cold LSODA start each step, overshoot allowed);
        "tight" = converged solution: rtol=atol=1e-10 and a tcrit at every 1/60 s grid point so
        the piecewise-constant inputs are never sampled across a discontinuity."""
        o = self.model.opts
        t = np.array([(self.k_sub + j) * SIM_TIME_STEP for j in range(self.n + 1)])
        if self.solver == "reference":
            sol, info = odeint(self.f, self.y, t, Dfun=self.jac, full_output=1, hmax=o.hmax,
                               mxstep=o.mxstep, atol=o.atol, rtol=o.rtol)
            if info["message"] != "Integration successful.":
                raise ValueError("LSODA: " + info["message"])          # -> :169
            self.y = sol[-1]
        else:
            y = self.y
            for j in range(self.n):
                # stay strictly inside [t_j, t_{j+1}) for event lookup: integrate on a local clock
                tj = t[j]
                vg, si = self.events.at(tj + 1e-7, True)
                fj = lambda yy, tau, tj=tj, vg=vg, si=si: self._f_frozen(yy, tj + tau, vg, si)
                sol = odeint(fj, y, [0.0, SIM_TIME_STEP], rtol=1e-11, atol=1e-12, mxstep=200000,
                             hmax=o.hmax)
                y = sol[-1]
            self.y = y
        self.k_sub += self.n

    def _f_frozen(self, y, t, vg, si):
        self.nfev += 1
        return pm.rhs(self.model, y, t, vg, si, self.Q_ref, self.Vdc_ref)

    # -- reward (PVDER_env.py:231-301) -----------------------------------------------------------
    def outputs(self) -> pm.Algebraic:
        t = self.k_sub * SIM_TIME_STEP
        # model evaluated at (y(tStop), tStop): an event at exactly tStop is already visible
        # (grid_events(t) returns the latest event with T <= t)
        vg, si = self.events.at(t + 1e-7, True)
        return pm.algebraic(self.model, self.y, t, vg, si, self.Q_ref, self.Vdc_ref)

    def reward_calc(self):
        o = self.outputs()
        return reward_terms(self.goals_list[0], self.DISCRETE_REWARD, o.Vrms, self.model.Vrms_ref, o.S_PCC, o.Vdc,
                            self.Vdc_ref, getattr(self, "reward_spec", None), Q_ref=self.Q_ref)

    # -- observation (PVDER_env.py:531-542) ------------------------------------------------------
    @property
    def state(self):
        o = self.outputs()
        return (o.i[0].real, o.i[0].imag, o.v[0].real, o.v[0].imag, o.S_PCC.real, o.S_PCC.imag,
                o.Vdc, o.Ppv, self.Vdc_ref, self.Q_ref, self.tStart / self.max_sim_time)

    @property
    def steps(self):                                                    # env_utilities.py:48-52
        return self._steps
