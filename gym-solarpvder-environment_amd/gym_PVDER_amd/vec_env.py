"""PVDERVec -- N independent PVDER-v0 environments stepped by ONE fused HIP launch per step.

Batched host-side mirror of the reference env class
(/root/reference/gym_PVDER/envs/PVDER_env.py:39 ``class PVDER``): same constructor arguments and
validation (:104-125, setters :561-620), same class-level specs (:56-93), same semantics of
``step`` (:138-196), ``action_calc`` (:198-229), ``reward_calc`` (:231-301), ``state`` (:531-542),
``reset`` (:316-334) and ``update_env_events`` (:413-438) -- but ``actions`` is a ``[N]`` device
tensor and observations/rewards/dones come back as device tensors.  All arithmetic of the step
runs in csrc/pvder_kernels.hip through the C ABI of include/pvder_hip.h; this file only owns
buffers, validates arguments and launches.  There is no CPU fallback.
"""
from __future__ import annotations

import contextlib
import copy
import ctypes as C
import logging
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native, der_model, etd_linear

logger = logging.getLogger("gym_PVDER_amd")

SIM_TIME_STEP = 1.0 / 60.0               # env_sim_spec['sim_time_step'], PVDER_env.py:63
MAX_EPISODE_STEPS = 500                  # gym_PVDER/__init__.py:10
# integrator steps per 1/60 s sim time step (DESIGN.md "Kernels"): classic RK4 is stability-bound
# by the current loop (h <= ~4.5 us); ETDRK4 propagates that loop exactly and is accuracy-bound
DEFAULT_MICRO = {"etdlin": 32, "etdrk4": 32, "rk4": 4096, "rk45": 1, "etdlin64": 32}
# ETDRK4 step schedule: a sim time step that starts at an input jump (new action / event) takes a first
# step of 1/(60*micro) s and one more over the remainder (graded), every other one DEFAULT_MICRO_COARSE
DEFAULT_MICRO_COARSE = 1
# pvder_step_cfg.off_range (include/pvder_hip.h): what happens to an env at the duty-cycle limit
OFF_RANGE = {"integrate": 0, "terminate": 1, "held_modes": 2, "integrate_always": 3}
# etdlin lane path: validated range of (Np / group Np) x Sinsol_scale (profiles/r05_lane_etdlin_envelope.txt)
LANE_OP_RANGE = (0.8, 1.0)

OBSERVED_QUANTITIES = ['iaR', 'iaI', 'vaR', 'vaI', 'P_PCC', 'Q_PCC', 'Vdc', 'Ppv', 'Vdc_ref', 'Q_ref',
                       'tStart']         # PVDER_env.py:44-49

GOAL_IDS = {'voltage_regulation': 0, 'power_regulation': 1, 'Q_regulation': 2}


def default_events_spec() -> Dict[str, dict]:
    """env_events_spec, PVDER_env.py:60-61 (dict order matters: insolation first)."""
    return {'insolation': {'t_events_start': 1.0, 't_events_stop': 39.0, 't_events_step': 1.0,
                           'min': 85.0, 'max': 100.0, 'ENABLE': False},
            'voltage': {'t_events_start': 1.0, 't_events_stop': 39.0, 't_events_step': 1.0,
                        'min': 0.98, 'max': 1.02, 'ENABLE': True}}


class EnvSpecMixin:
    """Argument validation + class-level specs shared by PVDERVec and the single-env PVDER."""

    env_model_spec = {'model_1': {'DERModelType': 'SinglePhase', 'derId': '10'},
                      'model_2': {'DERModelType': 'ThreePhaseUnbalanced', 'derId': '50'}}   # :56-58
    env_sim_spec = {'sim_time_step': SIM_TIME_STEP,
                    'n_sim_time_steps_per_env_step': {'default': 60, 'min': 1},
                    'min_sim_time': 1.0}                                                  # :63-65
    env_reward_spec = {'reward_list': {'default': ['voltage_error'],
                                       'valid': ['voltage_error', 'power_error', 'Q_error', 'Vdc_error']},
                       'reference_values': {'P_ref': 45.4e3, 'Q_ref': 5.5e3},
                       'DISCRETE_REWARD': True}                                           # :67-71
    env_action_spec = {'action_list': {'default': ['Q_control'], 'valid': ['Q_control', 'Vdc_control']},
                       'delQref': 25, 'delVdcref': 0.02}                                  # :73-75
    env_goal_spec = {'voltage_regulation': {'reward': {'required': ['voltage_error']},
                                            'action': {'required': ['Q_control']}},
                     'power_regulation': {'reward': {'required': ['power_error']},
                                          'action': {'required': ['Vdc_control']}},
                     'Q_regulation': {'reward': {'required': ['Q_error']},
                                      'action': {'required': ['Q_control']}}}             # :78-93
    default_goals = ['voltage_regulation']                                                # :95
    max_episode_steps = MAX_EPISODE_STEPS

    # -- validating setters (PVDER_env.py:561-620); same error behaviour ----------------------
    def _set_n(self, n):
        if n is None:
            n = self.env_sim_spec['n_sim_time_steps_per_env_step']['default']
        elif isinstance(n, bool) or not isinstance(n, int):
            raise ValueError("n_sim_time_steps_per_env_step must be an integer!")          # :614
        elif n < self.env_sim_spec['n_sim_time_steps_per_env_step']['min']:
            logger.warning('n_sim_time_steps_per_env_step should be greater than %s!',
                           self.env_sim_spec['n_sim_time_steps_per_env_step']['min'])     # :612
            n = self.env_sim_spec['n_sim_time_steps_per_env_step']['min']
        self._n = int(n)
        self._sim_time_per_env_step = self.env_sim_spec['sim_time_step'] * self._n        # :616
        self._delQref = self.env_action_spec['delQref'] * self._n                         # :617
        self._delVdcref = self.env_action_spec['delVdcref'] * self._n                     # :618

    def _set_max_sim_time(self, max_sim_time):
        cap = self.max_episode_steps * self._sim_time_per_env_step
        if max_sim_time is None:
            t = cap                                                                       # :565
        elif isinstance(max_sim_time, (int, float)) and not isinstance(max_sim_time, bool):
            if max_sim_time < self.env_sim_spec['min_sim_time']:
                t = self.env_sim_spec['min_sim_time']                                     # :568
            elif max_sim_time > cap:
                logger.warning('User specifed maximum simulation time is %s s, but allowable '
                               'simulation time is only %s s!', max_sim_time, cap)        # :570
                t = cap
            else:
                t = max_sim_time
        else:
            raise ValueError("max_sim_time must be a float!")                             # :575
        self._max_sim_time = float(t)

    def _set_goals(self, goals_list):
        if goals_list is None:
            goals_list = list(self.default_goals)       # the reference raises AttributeError here
        elif not set(goals_list).issubset(self.env_goal_spec.keys()):
            raise ValueError('Goal list:{} contains invalid elements, available elements are:{}'.format(
                goals_list, self.env_goal_spec.keys()))                                   # :586
        if len(goals_list) == 0:
            raise ValueError('goals_list must name at least one goal')
        self._goals_list = list(goals_list)

    def _set_discrete(self, flag):
        if flag is None:
            flag = self.env_reward_spec['DISCRETE_REWARD']
        elif not isinstance(flag, bool):
            raise ValueError("DISCRETE_REWARD must be a boolean!")                        # :598
        self._discrete = flag


class PVDERVec(EnvSpecMixin):
    """``num_envs`` PVDER-v0 environments on one MI355X.

    Args (first four as the reference's ``PVDER.__init__``, PVDER_env.py:104-105):
        goals_list, n_sim_time_steps_per_env_step, max_sim_time, DISCRETE_REWARD
        num_envs:   N
        device:     torch device (a ROCm GPU)
        integrator: 'etdlin' (default: fp32 ETD3RK on the model linearised at its operating point --
                    the whole linear response sits in the matrix exponential; two steps per env step),
                    'etdrk4' (fp32 exponential RK of stiff order 4 with only the stiff current loop in
                    the exponential; 16 steps per env step), 'rk4' (fp32 fixed-step classic RK4
                    micro-stepping), 'rk45' (fp64 adaptive Dormand-Prince: the literal tolerance-match path,
                    stability-bound at 0.5 s per batched step) or 'etdlin64' (fp64 etdlin with step doubling on a dyadic
                    menu as its error control -- rtol / atol in the RK45 kernel's norm --, the RK45 kernel behind it
                    for the envs the balanced reduction does not cover: the tolerance-match path at ~2e2 x the rate,
                    bench.py legs `rk45` / `etdlin64`)
        micro_steps: classic RK4: steps per 1/60 s sim time step (default 4096); ETDRK4 / etdlin: the
                    first step behind an input jump is 1/(60*micro_steps) s (default 32)
        micro_coarse: ETDRK4 only -- steps per sim time step that does not start at an input jump
                    (new action or event); default 1, 0 = uniform micro_steps
        graded: ETDRK4 only -- True (default): the sim time step that starts at an input jump takes
                    two steps, 1/(60*micro_steps) s and the remainder (the size of the first step
                    behind a jump sets the error); False: micro_steps uniform steps
        auto_reset: reset finished envs inside ``step`` (their returned obs is the reset obs)
        model_type: 'model_2' (three phase, derId 50, the PVDER-v0 default) or 'model_1'
        seed / env_id_base: key of the on-device event generator (env_id_base = global index of
                    env 0: results do not depend on how envs are sharded over GPUs)
    """

    observed_quantities = OBSERVED_QUANTITIES

    def __init__(self, num_envs: int, goals_list=('voltage_regulation',),
                 n_sim_time_steps_per_env_step: int = 15, max_sim_time: float = 40.0,
                 DISCRETE_REWARD: bool = True, device="cuda", integrator: Optional[str] = None,
                 micro_steps: Optional[int] = None, micro_coarse: Optional[int] = None, auto_reset: bool = True,
                 graded: bool = True,
                 model_type: str = 'model_2',
                 seed: int = 0, env_id_base: int = 0, rtol: float = 1e-6, atol: float = 1e-8,
                 events_spec: Optional[dict] = None, der_config: Optional[dict] = None, off_range: str = "integrate",
                 model_options: Optional[dict] = None, lane_params: Optional[dict] = None, lazy_events: Optional[bool] = None):
        """``der_config``: overrides of the model's config block (keys of config_der.json's derId entry, e.g.
        ``{"C_actual": 330e-6, "Kp_DC": -2.2}``) -- ONE parameter set per PVDERVec (the steady state, the
        Jacobian of the etdlin path and the coefficient tables are derived from it); for a batch spread over
        several parameter sets see gym_PVDER_amd.param_groups.PVDERVecGroups.
        ``lane_params``: per-ENV values (domain randomisation) of the parameters outside the exponential operator --
        ``{"C_actual": [N], "Kp_DC": [N], "Ki_DC": [N], "Kp_Q": [N], "Ki_Q": [N], "Np": [N], "Sinsol_scale": [N]}``, any subset,
        NaN = the set's value (three-phase model).  With the default ``integrator="etdlin"`` the batch stays on the matrix-core
        path: the exponential operator is the set's, each env's own constants enter the right-hand side, 2 / 6 / 8 Krogstad steps per
        env step by the spread of the sets (``lane_steps`` overrides; include/pvder_hip.h pvder_plan_lane_params, ABI 16; 11 / 22 /
        27 us per batched step); with ``integrator="etdrk4"`` every env
        steps on the 23-state ETDRK4 path with its own constants (45-75 us per batched step).  reset() and
        the same-step auto-reset start every env from ITS OWN steady state (a Newton root per env, folded on the host: the
        reference builds each DER with steadyStateInitialization=True, PVDER_env.py:374-378).
        ``off_range``: what happens to an env whose duty cycle reaches its limit (Vdc_ref walked down 6 %: the reference's
        action 4 held for 115 env steps) -- ``"integrate"`` (default): integrated through, as the reference does, by the hybrid
        scheme of csrc/pvder_sat_hybrid.hpp in a kernel of its own that follows the step launches while an env is (or lately
        was) beyond 98.75 % of the limit: <= 6e-3 per env step (median 1.5e-5) of the converged event-locating solution; nothing
        on any launch until then, milliseconds per env step for the envs at the limit; ``"integrate_always"``: the same with
        that kernel behind EVERY launch (+4 us: deterministic whatever the host's timing; tests); ``"held_modes"``: rounds 3-5's
        step on the operator sets of the anti-windup modes (55 us per env step, 1e-2 ... 1e-1 off beyond saturation; A/B);
        ``"terminate"``: the episode ends (done, reward -100, the reference's failure branch PVDER_env.py:169-172).  Envs at the
        CURRENT limit are not off range: they stay on the matrix-core path (careful pass) and are never terminated."""
        if off_range not in OFF_RANGE:
            raise ValueError("off_range must be one of %s" % (sorted(OFF_RANGE),))
        self.off_range = off_range
        # lazy_events: the generator's event tables are written as the episode advances (pvder_autoreset.lazy_events: the step that
        # ends an episode does not stop to hash 38 rows per env -- 4.6 us that a desynchronised batch pays in EVERY launch, a batch
        # in lockstep once per episode); ``ev_vg`` / ``ev_si`` read them whole.  None (default): on while the envs' clocks differ
        # (a masked reset or sort_lanes() since the last full reset), off in lockstep (there the rows' share of a launch is 0.03 us
        # whole and 0.1 us one at a time); True / False: always / never.  Same results bit for bit either way.
        self.lazy_events = None if lazy_events is None else bool(lazy_events)
        self._desync = False                 # the envs' clocks differ (a masked reset / sort_lanes since the last full reset)
        self._lazy_active = bool(lazy_events) if lazy_events is not None else False
        if not isinstance(num_envs, int) or num_envs < 1:
            raise ValueError("num_envs must be a positive integer")
        if model_type not in self.env_model_spec:
            raise ValueError("model_type must be one of %s" % list(self.env_model_spec))
        if integrator is None:
            integrator = "etdlin"
        if integrator not in _native.INTEGRATOR_IDS:
            raise ValueError("integrator must be one of %s" % sorted(_native.INTEGRATOR_IDS))
        if micro_steps is None:
            micro_steps = DEFAULT_MICRO[integrator]
            if integrator == "rk4" and model_type == 'model_1':
                micro_steps = 8192      # derId 10: current-loop poles at 1.02e6 rad/s (h <= 2.8 us)
        if isinstance(micro_steps, bool) or not isinstance(micro_steps, int) or micro_steps < 1:
            raise ValueError("micro_steps must be a positive integer")
        if micro_coarse is None:
            # (etdlin: the ETDRK4 schedule drives its 23-state tail -- envs off balance or at the duty-cycle limit;
            # envs far from the nominal operating point stay on the matrix-core path since round 3: careful pass)
            micro_coarse = min(DEFAULT_MICRO_COARSE, micro_steps) if integrator in ("etdrk4", "etdlin") else 0
        if isinstance(micro_coarse, bool) or not isinstance(micro_coarse, int) or micro_coarse < 0:
            raise ValueError("micro_coarse must be a non-negative integer")
        self.micro_coarse = int(micro_coarse)
        if not isinstance(graded, bool):
            raise ValueError("graded must be a bool")
        self.graded = bool(graded) and integrator in ("etdrk4", "etdlin") and self.micro_coarse > 0
        self.num_envs = num_envs
        self._set_n(n_sim_time_steps_per_env_step)
        self.max_sim_time_user = max_sim_time
        self._set_max_sim_time(max_sim_time)
        self.env_goal_spec = copy.deepcopy(type(self).env_goal_spec)      # per instance: the reference's class-level
        #                                  dict is shared by (and mutated across) every instance, SURVEY.md section 5
        self._set_goals(list(goals_list) if goals_list is not None else None)
        self._set_discrete(DISCRETE_REWARD)
        self.integrator = integrator
        self.micro_steps = int(micro_steps)
        self.auto_reset = bool(auto_reset)
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        self.rtol, self.atol = float(rtol), float(atol)
        self.env_events_spec = copy.deepcopy(events_spec) if events_spec else default_events_spec()
        self.der_id = self.env_model_spec[model_type]['derId']
        self.der_config = dict(der_config) if der_config else None
        # ``model_options``: der_model.MODEL_OPTIONS -- the resolutions of questions about the (absent) pvder model that the
        # reference cannot decide and that move the observations: {"ppv_at_mpp": True} evaluates the PV array's power at the
        # MPP voltage instead of the instantaneous DC-link voltage
        self.model_options = dict(model_options) if model_options else None
        self.par = der_model.param_vector(self.der_id, self.der_config, self.model_options)
        self.phases = int(self.par[0])
        self.ns = 6 * self.phases + 5
        self.y0 = der_model.steady_state(self.par, self.der_id)

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PVDERVec runs on a ROCm GPU only (device=%s); there is no CPU path" % device)
        self._lib = _native.lib()
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.fdtype = torch.float64 if integrator in ("rk45", "etdlin64") else torch.float32
        N = num_envs
        self._alloc_event_tables()
        # state rows and output block are the two halves of ONE allocation: a trajectory snapshot is then one device copy
        self._state_and_out = torch.zeros((self.ns + 2 + 13, N), dtype=self.fdtype, device=self.device)
        self.fstate = self._state_and_out[:self.ns + 2]
        self.istate = torch.zeros((4, N), dtype=torch.int32, device=self.device)
        self.out = self._state_and_out[self.ns + 2:]
        # per-workgroup partial statistics of the last step ([n_wg, 4]; summed on demand by step_stats)
        n_wg = (num_envs + self._lib.pvder_block_size() - 1) // self._lib.pvder_block_size()
        self.stats = torch.zeros((n_wg * (self._lib.pvder_block_size() // 64), 4), dtype=torch.float32,
                                 device=self.device)           # one slot per wave (include/pvder_hip.h)
        self._progress = torch.zeros(n_wg, dtype=torch.int32, device=self.device)     # scratch of the fused rollouts
        self.hstep = torch.zeros(N, dtype=torch.float64, device=self.device) if integrator in ("rk45", "etdlin64") else None
        self._par_c = (C.c_double * len(self.par))(*self.par)
        self._y0_c = (C.c_double * len(self.y0))(*self.y0)
        self._lane_folded = None
        if lane_params:
            self.set_lane_params(lane_params)
        self._needs_reset = True
        self._plan = None
        self._env_of_lane = None     # sort_lanes(): int32 [N], the env id column l of every buffer carries (None = identity)
        self._lane_of_env = None
        self.lane_io = False         # True: step / rollout / reset take and return LANE order (no gathers); see sort_lanes
        self._ksub_hint = -1         # pvder_plan_hint: the clock the batch's envs are at (survives plan rebuilds; -1 = unknown)

    def __del__(self):
        try:
            self._drop_plan()
            w = self.__dict__.pop("_served_words", None)
            if w is not None:
                self._lib.pvder_served_free(w[0], w[2])
        except Exception:          # interpreter shutdown: the library may be gone already
            pass

    # -- everything a plan folds in: assigning any of these drops the plan (rebuilt at the next step)
    _PLAN_KEYS = frozenset(("seed", "env_id_base", "rtol", "atol", "micro_steps", "micro_coarse", "graded", "env_reward_spec", "lane_steps",
                            "est_guard", "auto_reset", "_debug_knob", "_keep_events", "_n", "_max_sim_time",
                            "_goals_list", "_discrete", "max_episode_steps", "env_events_spec", "env_goal_spec", "integrator",
                            "off_range", "_lazy_active"))

    def _lane_envelope(self, rows: np.ndarray):
        """(Krogstad steps per env step, None) for per-env sets inside the validated envelope of the etdlin lane path, else
        (None, what left it) -- see set_lane_params."""
        base = der_model.lane_param_rows({k: np.full(self.num_envs, float(self._der_cfg_value(k))) for k in
                                          ("C_actual", "Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q", "Np")}, self.num_envs)
        with np.errstate(invalid="ignore", divide="ignore"):
            ratio = rows[:5] / base[:5]
            op = (rows[5] / base[5]) * rows[6]                     # Np / the group's Np x insolation scale (NaN = 1)
        if np.any(np.isfinite(ratio) & (ratio <= 0)) or np.any(np.isfinite(op) & (op <= 0)):
            return None, "a parameter of the opposite sign"
        ratio = np.where(np.isfinite(ratio), ratio, 1.0)                              # (NaN / a missing key: the group's value)
        npr = np.where(np.isfinite(rows[5]), rows[5] / base[5], 1.0)
        op = npr * np.where(np.isfinite(rows[6]), rows[6], 1.0)
        spread = np.maximum(ratio, 1.0 / ratio).max(axis=1)                           # per key: how far the farthest env sits
        s_c, s_g = float(spread[0]), float(spread[1:].max())
        op_lo, op_hi = float(op.min()), float(op.max())
        if op_lo < LANE_OP_RANGE[0] - 1e-9 or op_hi > LANE_OP_RANGE[1] + 1e-9:
            return None, "PV array x insolation scale %.2f-%.2f of the group's; validated %.2f-%.2f" % ((op_lo, op_hi) + LANE_OP_RANGE)
        if s_c > 3.0 + 1e-9 or s_g > 2.0 + 1e-9:
            return None, "C x %.2f, outer PI gains x %.2f of the group's value; validated to x 3 / x 2" % (s_c, s_g)
        if s_c <= 1.0 / 0.7 + 1e-9 and s_g <= 1.25 + 1e-9:
            return 2, None
        if s_c <= 2.0 + 1e-9 and s_g <= 1.5 + 1e-9:
            return 6, None
        return 8, None

    def set_lane_params(self, lane_params: Optional[dict]):
        """Install (or, with None, remove) the per-env parameter override -- see the constructor's ``lane_params``."""
        self._drop_plan()
        user = self.__dict__.get("_integrator_user")
        if user is not None and self.integrator != user:          # (an earlier set had sent this env to the ETDRK4 lane kernels)
            self.integrator = user
        object.__setattr__(self, "_integrator_user", self.integrator)
        if not lane_params:
            object.__setattr__(self, "_lane_folded", None)
            return
        if self.integrator not in ("etdrk4", "etdlin") or self.ns != 23:
            raise ValueError("lane_params need integrator='etdlin' or 'etdrk4' and the three-phase model (for whole parameter "
                             "GROUPS with their own operators see param_groups.PVDERVecGroups)")
        rows = np.ascontiguousarray(der_model.lane_param_rows(lane_params, self.num_envs))
        if self.integrator == "etdlin":
            # The matrix-core path keeps the GROUP's exponential operator and takes K Krogstad steps per env step: how many it needs
            # for 1e-4 of the per-set converged solution follows the spread of the sets around the group's values
            # (profiles/r05_lane_etdlin_envelope.txt, fp64 against bridge runs): C x 0.7-1.4 and outer PI gains x 0.8-1.25: K = 2
            # (5.6e-5); C x 0.5-2 / gains x 0.67-1.5: K = 6 (5.2e-5); C x 0.33-3 / gains x 0.5-2: K = 8 (8.1e-5).  The PV array's
            # size and insolation scale move each env's OPERATING POINT away from the one the shared Jacobian was taken at: validated
            # for Np x Sinsol_scale within 0.8-1.0 of the group's array (Np 10-11 of 11, scale 0.9-1.0; same file).  BEYOND either
            # envelope the scheme is 5e-4 ... 3e-3 off, so the env FAILS SAFE: it steps with the ETDRK4 lane kernels (every env on the
            # 23-state path with its own constants: 3e-5 at any spread, 45-76 us per batched step instead of 11-24), one INFO line.
            # `lane_steps` (attribute, before reset()) overrides the step count inside the envelope.
            steps, why = self._lane_envelope(rows)
            if steps is None:
                logger.info("lane_params reach beyond the spread the etdlin lane path is validated for (%s): this env steps with "
                            "integrator='etdrk4' (per-env constants on the 23-state path, 3e-5 of the per-set converged solution "
                            "at any spread)", why)
                self.integrator = "etdrk4"
            else:
                object.__setattr__(self, "_lane_steps_auto", steps)
        folded = np.empty((_native.LPF_ROWS, self.num_envs), dtype=np.float32)
        _native.check(self._lib.pvder_fold_lane_params(self._par_c, rows.ctypes.data, self.num_envs, folded.ctypes.data),
                      "pvder_fold_lane_params")
        object.__setattr__(self, "_lane_folded", torch.from_numpy(folded).to(self.device))

    def set_reward_spec(self, reward_types):
        """``env_goal_spec[goal]['reward']['my_spec'] = reward_types`` for the active goal (the reference's
        ``update_env_goal`` is "under construction", PVDER_env.py:454-456; ``reward_calc`` iterates this
        list, :249): the goal's required term, optionally plus the terms whose targets ``reward_calc`` defines
        under that goal -- 'Vdc_error' / 'voltage_error' (power_regulation), 'Q_error' with Qtarget = Q_ref
        (voltage_regulation), 'voltage_error' (Q_regulation)."""
        self.env_goal_spec[self._goals_list[0]]['reward']['my_spec'] = list(reward_types)
        self._drop_plan()
        self._cfg()                       # validate now

    def __setattr__(self, name, value):
        if name == "_n" and self.__dict__.get("_ev_vg") is not None and self.__dict__.get("_n") not in (None, value):
            self._complete_tables()       # (lazy_events: the rows a step leaves behind are its successor's AT THIS n -- whole tables first)
        object.__setattr__(self, name, value)
        if name in self._PLAN_KEYS and self.__dict__.get("_plan") is not None:
            self._drop_plan()

    def _drop_plan(self):
        p = self.__dict__.get("_plan")
        object.__setattr__(self, "_plan", None)
        if p is not None:
            object.__setattr__(self, "_ksub_hint", int(self._lib.pvder_plan_hint(p, -2)))     # (the plan advanced it step by step)
            self._lib.pvder_plan_destroy(p)

    _debug_knob = 0          # pvder_step_cfg.reserved (tests only)

    # -- properties mirroring the reference (PVDER_env.py:544-558, env_utilities.py:48-52) -------
    @property
    def n_sim_time_steps_per_env_step(self):
        return self._n

    @property
    def max_sim_time(self):
        return self._max_sim_time

    @property
    def goals_list(self):
        return self._goals_list

    @property
    def DISCRETE_REWARD(self):
        return self._discrete

    @property
    def steps(self) -> torch.Tensor:
        return self.istate[1]

    @property
    def tStart(self) -> torch.Tensor:
        return self.istate[0].to(torch.float64) * SIM_TIME_STEP

    @property
    def obs(self) -> torch.Tensor:
        """[N, 11] view of the SoA output block (valid until the next step/reset); a copy in env order behind sort_lanes() unless
        ``lane_io``."""
        return self.to_env_order(self.out[:11]).t() if self._mapped() else self.out[:11].t()

    @property
    def Q_ref(self) -> torch.Tensor:
        return self.fstate[self.ns]

    @property
    def Vdc_ref(self) -> torch.Tensor:
        return self.fstate[self.ns + 1]

    def use_out(self, buf: torch.Tensor):
        """Point the env at another [13, N] output block (same dtype/device): the next launches write
        their observation / reward / done rows there.  For callers that hand the block of step t to a
        consumer (e.g. an all-gather in flight) while step t+1 computes.  Rows of envs that do not run
        (finished, auto_reset off: the stale tuple, PVDER_env.py:145-152) are not rewritten by a launch, so
        without auto_reset the current block is copied into the new one first."""
        if buf.shape != self.out.shape or buf.dtype != self.out.dtype or buf.device != self.out.device \
                or not buf.is_contiguous():
            raise ValueError("use_out: need a contiguous %s tensor of shape %s on %s"
                             % (self.out.dtype, tuple(self.out.shape), self.out.device))
        if not self.auto_reset and buf.data_ptr() != self.out.data_ptr():
            buf.copy_(self.out)
        self.out = buf

    # -- events ---------------------------------------------------------------------------------
    def _event_grid(self):
        v = self.env_events_spec['voltage']       # grid always from the 'voltage' entry (:408-410)
        times = np.arange(v['t_events_start'], v['t_events_stop'], v['t_events_step'])
        s0 = v['t_events_start'] * 60.0
        ds = v['t_events_step'] * 60.0
        if len(times) and (abs(s0 - round(s0)) > 1e-6 or abs(ds - round(ds)) > 1e-6 or round(ds) < 1):
            logger.warning('event times are not multiples of 1/60 s: they are applied at the next '
                           'sim time step boundary')
        return len(times), int(math.ceil(s0 - 1e-6)), max(1, int(round(ds)))

    def _alloc_event_tables(self):
        K, s0, ds = self._event_grid()
        self.ev_count, self.ev_s0, self.ev_ds = K, s0, ds
        N = self.num_envs
        self._ev_vg = torch.ones((K + 1, N), dtype=torch.float32, device=self.device)
        self._ev_si = torch.full((K + 1, N), float(self.par[der_model.IDX["SINSOL0"]]), dtype=torch.float32,
                                device=self.device)

    def update_env_events(self, event_spec_list: List[dict]):
        """PVDER_env.py:413-438: mutate the event spec; as in the reference, an env that already has a
        simulation attached gets a fresh random event draw at once (:435-438), states untouched."""
        assert isinstance(event_spec_list, list), 'event_spec_list should be a list!'
        for event_spec in event_spec_list:
            assert isinstance(event_spec, dict), 'Event spec should be a dictionary!'
            assert len(event_spec.keys()) == 1, 'Only one event type should be specified at at time!'
            event_type = list(event_spec)[0]
            if event_type not in self.env_events_spec:
                raise ValueError('{} is not a valid event!'.format(event_type))
            for key, val in event_spec[event_type].items():
                if key not in self.env_events_spec[event_type]:
                    raise ValueError('{} is not a valid paramter for {} event!'.format(key, event_type))
                self.env_events_spec[event_type][key] = val
        self._drop_plan()
        K, s0, ds = self._event_grid()
        if (K, s0, ds) != (self.ev_count, self.ev_s0, self.ev_ds):
            self._alloc_event_tables()
            self._needs_reset = True
        elif not self._needs_reset and not self._keep_events:
            # regenerate the event tables of the running episodes (new episode key -> new draw)
            saved = (self.fstate.clone(), self.istate.clone(), self.out.clone())
            self._reset_launch(None)
            episode = self.istate[2].clone()
            self.fstate.copy_(saved[0]); self.istate.copy_(saved[1]); self.out.copy_(saved[2])
            self.istate[2].copy_(episode)

    def set_event_tables(self, vgrid: torch.Tensor, sinsol: torch.Tensor):
        """Explicit per-env event tables [K+1, N] (row 0 = nominal) -- parity runs share them
        with the oracle.  Kept across resets until :meth:`clear_event_tables`."""
        assert vgrid.shape == self._ev_vg.shape and sinsol.shape == self._ev_si.shape
        self._ev_vg.copy_(vgrid)
        self._ev_si.copy_(sinsol)
        self._keep_events = True
        self._drop_plan()

    def clear_event_tables(self):
        self._keep_events = False
        self._drop_plan()

    _keep_events = False

    # -- launch plumbing --------------------------------------------------------------------------
    def _cfg(self) -> _native.StepCfg:
        c = _native.StepCfg()
        c.n_sub = self._n
        c.micro = self.micro_steps
        c.micro_coarse = self.micro_coarse
        c.graded = 1 if self.graded else 0
        c.max_ksub = int(math.ceil(self._max_sim_time * 60.0 - 1e-6))
        c.max_steps = self.max_episode_steps
        c.goal = GOAL_IDS[self._goals_list[0]]                 # only the first goal counts (:234)
        c.discrete = 1 if self._discrete else 0
        c.ev_s0, c.ev_ds, c.ev_count = self.ev_s0, self.ev_ds, self.ev_count
        c.del_qref = self._delQref / der_model.SBASE           # :225
        c.del_vdcref = self._delVdcref / der_model.VDCBASE     # :229
        c.inv_max_sim_time = 1.0 / self._max_sim_time
        c.p_target = self.env_reward_spec['reference_values']['P_ref'] / der_model.SBASE   # :245
        c.q_target = self.env_reward_spec['reference_values']['Q_ref'] / der_model.SBASE   # :241
        c.rtol, c.atol = self.rtol, self.atol
        c.reserved = self._debug_knob
        c.off_range = OFF_RANGE[self.off_range]
        # reward types beyond the goal's required one: env_goal_spec[goal]['reward']['my_spec'] (:249; the
        # reference only ever fills it with the 'required' list, :448-452 -- its update path is "under
        # construction", :454-456 -- but reward_calc honours whatever it holds)
        goal = self._goals_list[0]
        spec = self.env_goal_spec[goal]['reward']
        mine = list(spec.get('my_spec', spec['required']))
        # which optional terms reward_calc can evaluate under a goal: the targets it defines there (:237-245) --
        # Qtarget under voltage_regulation (= PV_model.Q_ref) and Q_regulation, Vdctarget / Ptarget under
        # power_regulation; voltage_error needs none
        optional = {'voltage_regulation': {'Q_error'}, 'power_regulation': {'Vdc_error', 'voltage_error'},
                    'Q_regulation': {'voltage_error'}}[goal]
        unknown = set(mine) - set(self.env_reward_spec['reward_list']['valid'])
        if unknown or spec['required'][0] not in mine or len(set(mine)) != len(mine) \
                or set(mine) - set(spec['required']) - optional:
            raise ValueError("reward my_spec %s of goal %s: must hold the goal's required term %s, optionally plus %s "
                             "(the terms whose targets reward_calc defines for this goal, PVDER_env.py:237-245)"
                             % (mine, goal, spec['required'], sorted(optional)))
        c.extra_rewards = (1 if 'Vdc_error' in mine else 0) | (2 if ('Q_error' in mine and goal == 'voltage_regulation') else 0) \
            | (4 if ('voltage_error' in mine and goal != 'voltage_regulation') else 0)
        return c

    def _spec(self) -> _native.EventSpec:
        s = _native.EventSpec()
        s.ev_count = self.ev_count
        s.enable_voltage = 1 if self.env_events_spec['voltage']['ENABLE'] else 0
        s.enable_insol = 1 if self.env_events_spec['insolation']['ENABLE'] else 0
        s.v_min, s.v_max = self.env_events_spec['voltage']['min'], self.env_events_spec['voltage']['max']
        s.s_min, s.s_max = self.env_events_spec['insolation']['min'], self.env_events_spec['insolation']['max']
        return s

    def _etd_coef(self, cfg) -> torch.Tensor:
        """Device copy of the ETDRK4 coefficient block (host-folded in fp64 by the library), refreshed
        only when the step sizes change."""
        key = (cfg.micro, cfg.micro_coarse, cfg.graded)
        if getattr(self, "_coef_key", None) != key:
            n = self._lib.pvder_etdrk4_coef_count()
            host = (C.c_float * n)()
            _native.check(self._lib.pvder_etdrk4_coefficients(self._par_c, C.byref(cfg), host), "pvder_etdrk4_coefficients")
            self._coef = torch.tensor(list(host), dtype=torch.float32, device=self.device)
            self._coef_key = key
        return self._coef

    _lin_cache: Dict[tuple, np.ndarray] = {}      # (derId, micro, lengths) -> host block (1 s of expm per entry)

    def _lin_coef(self, cfg) -> _native.LinCoef:
        """``struct pvder_lin_coef`` of the etdlin path: the step-length menu of this env's schedule and
        the device copy of the f-form matrices (gym_PVDER_amd/etd_linear.py), rebuilt only when the
        schedule changes."""
        lens = etd_linear.choose_sets(cfg.n_sub, cfg.micro, cfg.ev_s0, cfg.ev_ds, cfg.ev_count, cfg.max_ksub,
                                      _native.LIN_MAX_SETS)
        # the quiet env step (no event row change at its start) as ONE Krogstad step: its block sits behind the sets
        k4_h = etd_linear.quiet_step_length(cfg.n_sub)
        # ... and, with per-env parameter sets installed, the two blocks of the lane path behind it (pvder_lin_coef.k4 = 3)
        lane_k4 = self.__dict__.get("_lane_folded") is not None and k4_h > 0.0 and self.phases == 3
        if lane_k4:
            lane_k4 = int(self.__dict__.get("lane_steps") or self.__dict__.get("_lane_steps_auto", 2))
            if not 1 <= lane_k4 <= 16:
                raise ValueError("lane_steps must be 1 .. 16")
        if self.__dict__.get("_lane_folded") is not None and not lane_k4:
            raise ValueError("lane_params on the etdlin path take the env step as two Krogstad steps: n_sim_time_steps_per_env_step "
                             "<= %d (three-phase model); use integrator='etdrk4' beyond" % int(etd_linear.K4_MAX_H * 60))
        key = (self.der_id if not (self.der_config or self.model_options) else (self.der_id, self.par.tobytes()), cfg.micro, cfg.n_sub, tuple(lens),
               lane_k4)
        if getattr(self, "_lin_key", None) != key:
            host = PVDERVec._lin_cache.get(key)
            if host is None:
                host = etd_linear.build_coefficients(self.par, self.der_id, cfg.micro, lens, k4_h, lane_k4)
                PVDERVec._lin_cache[key] = host
            assert host.size == etd_linear.MAT_FLOATS + (len(lens) - 1) * self._lib.pvder_lin_set_floats() \
                + (etd_linear.N_K4_MATS * etd_linear.MAT_FLOATS * (3 if lane_k4 else 1) if k4_h > 0.0 else 0)
            self._lin_dev = torch.from_numpy(host).to(self.device)
            sched = etd_linear.build_schedule(cfg.n_sub, cfg.micro, lens)
            self._lin_sched = torch.from_numpy(sched).to(self.device)
            self._lin_key = key
        lc = _native.LinCoef()
        lc.sets = self._lin_dev.data_ptr()
        lc.sched = self._lin_sched.data_ptr()
        lc.n_sched_rows = int(self._lin_sched.shape[0])
        lc.n_sets = len(lens)
        for k, t in enumerate(lens):
            lc.len_ticks[k] = int(math.ceil(t))      # (informational: the careful pass's set may end on a fraction of a tick)
        lc.est_guard = self.est_guard
        lc.k4 = ((3 | (int(lane_k4) << 8)) if lane_k4 else 1) if k4_h > 0.0 else 0
        if self.phases == 3:
            if getattr(self, "_seq_key", None) != key[:3]:       # (a steady state + an eig: cached with the coefficient block)
                self._seq_E, self._seq_key = etd_linear.seq_propagator(self.par, self.der_id, cfg.n_sub), key[:3]
            E = self._seq_E
            for r in range(3):
                for k in range(3):
                    lc.seq_prop[6 * r + 2 * k], lc.seq_prop[6 * r + 2 * k + 1] = float(E[r, k].real), float(E[r, k].imag)
        return lc

    est_guard = etd_linear.EST_GUARD

    def _der_cfg_value(self, key: str) -> float:
        """The group's value of a config_der.json key (with this env's der_config overrides)."""
        return float(der_model.der_config(self.der_id, self.der_config or None)[key])

    def _lin64_coef(self, cfg) -> _native.LinCoef:
        """``struct pvder_lin_coef`` of the etdlin64 path: the fp64 f-form matrices of the tick and of the dyadic menu
        (etd_linear.pack_block64) on the device."""
        key = (self.der_id if not (self.der_config or self.model_options) else (self.der_id, self.par.tobytes()), cfg.micro, cfg.n_sub, "f64")
        if getattr(self, "_lin64_key", None) != key:
            host = PVDERVec._lin_cache.get(key)
            if host is None:
                L, _ = etd_linear.jacobian(self.par, self.der_id)
                host = etd_linear.pack_block64(L, cfg.n_sub, cfg.micro)
                PVDERVec._lin_cache[key] = host
            self._lin64_dev = torch.from_numpy(host).to(self.device)
            self._lin64_key = key
        lc = _native.LinCoef()
        lc.sets = self._lin64_dev.data_ptr()
        lc.n_sets = 1 + etd_linear.LIN64_LEVELS
        lc.sched, lc.n_sched_rows, lc.est_guard = None, 0, 1.0
        return lc

    def _plan_fingerprint(self) -> bytes:
        """What a plan folds in, as the FOLDED structs themselves (a hand-picked list of attributes missed _delQref,
        the event grid, _keep_events ...): the bytes of ``pvder_step_cfg`` and ``pvder_autoreset`` as the next launch would
        pass them, the parameter vector and the etdlin guard.  reset() / load_state_dict() rebuild the plan only when this
        changes -- specs edited IN PLACE behind the env's back (attribute assignments drop the plan at once) -- so the reset
        launch and the step plan can never disagree about the configuration."""
        return bytes(self._cfg()) + bytes(self._autoreset(self.auto_reset)) + self.par.tobytes() + repr(self.est_guard).encode()

    def _make_plan(self):
        """Fold cfg / reset block / coefficients into a C-side plan (include/pvder_hip.h "plans")."""
        object.__setattr__(self, "_plan_fp", self._plan_fingerprint())
        cfg, ar = self._cfg(), self._autoreset(self.auto_reset)
        integrator = self.integrator
        coef = self._etd_coef(cfg).data_ptr() if integrator in ("etdrk4", "etdlin") else None
        lin = None
        if integrator == "etdlin":
            try:
                lin = C.byref(self._lin_coef(cfg))
            except ValueError as e:
                # (env steps of more than ~1000 sim time steps: the schedule table cannot list their segments)
                logger.warning("etdlin cannot tile this step schedule (%s): this env steps with integrator='etdrk4'", e)
                integrator = "etdrk4"
        if integrator == "etdlin64":
            try:
                if self.ns != 23:
                    raise ValueError("three-phase model only")
                lin = C.byref(self._lin64_coef(cfg))
                h = C.c_void_p()
                _native.check(self._lib.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS["etdlin64"], self._par_c,
                                                          C.byref(cfg), C.byref(ar), None, lin, self.num_envs), "pvder_plan_create")
                self._plan_integrator, self._plan = "etdlin64", h
                self._lib.pvder_plan_hint(h, self.__dict__.get("_ksub_hint", -1))
                return h
            except ValueError as e:
                # (single phase, or an event grid that cuts env steps: one segment per env step is what the kernel takes)
                logger.warning("etdlin64 does not cover this configuration (%s): this env steps with integrator='rk45'", e)
                integrator, lin = "rk45", None
        self._plan_integrator = integrator
        h = C.c_void_p()
        _native.check(self._lib.pvder_plan_create(C.byref(h), _native.INTEGRATOR_IDS[integrator], self._par_c,
                                                  C.byref(cfg), C.byref(ar), coef, lin, self.num_envs),
                      "pvder_plan_create")
        self._plan = h
        if self.__dict__.get("_lane_folded") is not None:
            if integrator not in ("etdrk4", "etdlin"):
                raise RuntimeError("lane_params are installed but this configuration steps with integrator=%r" % integrator)
            _native.check(self._lib.pvder_plan_lane_params(h, self._lane_folded.data_ptr()), "pvder_plan_lane_params")
        # where the batch's envs are (a prefetch hint for the step launch, include/pvder_hip.h): kept across plan rebuilds
        # (PVDER_NO_HINT=1: A/B runs without it)
        self._lib.pvder_plan_hint(h, -1 if os.environ.get("PVDER_NO_HINT") else self.__dict__.get("_ksub_hint", -1))
        if self.__dict__.get("_graph_hints"):
            self._lib.pvder_plan_graph_hints(h, 1)
        return h

    def kernel_name(self) -> str:
        """The step kernel a launch of this env runs, as rocprofv3 prints it (pvder_plan_kernel_name, ABI 17)."""
        plan = self._plan or self._make_plan()
        return self._lib.pvder_plan_kernel_name(plan).decode()

    def graph_hints(self, on: bool = True):
        """Let a HIP-graph CAPTURE of ``rollout(fused=False)`` keep the clock hints it was captured with (pvder_plan_graph_hints):
        only for a graph replayed exactly where it was captured (bench.py's window).  The default clears the hint in every captured
        launch -- a graph replayed in a loop would otherwise carry stale ones (slower than none; results never depend on it)."""
        object.__setattr__(self, "_graph_hints", bool(on))
        if self._plan is not None:
            self._lib.pvder_plan_graph_hints(self._plan, 1 if on else 0)

    # -- the order of the envs in the buffers (PVDER_env.py:316-334, :400-411: every env owns its clock and events) --
    def sort_lanes(self, key: Optional[torch.Tensor] = None, lane_io: Optional[bool] = None):
        """Re-order the envs IN the buffers -- by default by clock (``istate[0]``: sim time steps into the episode), stably -- so
        that the 64 lanes of a wave hold envs at the same point of their episodes: one step schedule per wave (a quiet step OR a step
        behind an event row, not both), ONE event row read per wave (4 cache lines instead of 64), i.e. a batch whose envs were reset
        one by one steps at the lockstep rate again (bench.py `desynchronised`: 21.8 -> ~9 us).  The envs keep their identity:
        ``env_of_lane`` ([N] int32) says which env column l carries, the event generator stays keyed by the env id (auto-resets
        draw what they would have drawn), results are bit-identical PER ENV ID to the unsorted batch
        (tests/test_gpu_etdlin.py::test_sorted_lanes_are_bit_identical_per_env_id).  I/O: with ``lane_io=False`` (default) step /
        rollout / reset still take and return ENV order -- two gathers per call; with ``lane_io=True`` they take and return LANE
        order (row l <-> env ``env_of_lane[l]``: what a vectorised policy, to which envs are exchangeable, wants).  Call again when
        masked resets have changed the phases; envs whose episodes have one length keep their order for ever."""
        if self._traj is not None:
            raise RuntimeError("sort_lanes: not while a trajectory is recorded (record_trajectory names lanes)")
        N = self.num_envs
        with self._device_ctx():
            if key is None:
                # by clock -- and, first, by the clock's position in the event grid: whether an env step starts at an event row (tick +
                # two ETD3RK steps) or not (one Krogstad step) depends on (ksub - ev_s0) mod ev_ds alone, so the envs that take the
                # SAME schedule in any given step are contiguous (an event every 4th env step: four runs of N / 4 envs; sorted by
                # clock alone every 4th boundary between two clocks would sit inside a wave that then runs both schedules)
                k = self.istate[0].to(torch.int64)
                if self.ev_count > 0 and self.ev_ds > 0:
                    k = ((k - self.ev_s0) % self.ev_ds) * (1 << 32) + k
            else:
                k = key.to(self.device).to(torch.int64)
            if k.shape != (N,):
                raise ValueError("key must have shape (%d,)" % N)
            order = torch.argsort(k, stable=True)                              # lane l of the new order takes what lane order[l] holds
            tensors = [self._state_and_out, self.istate, self._ev_vg, self._ev_si]
            if self.hstep is not None:
                tensors.append(self.hstep)
            if self.__dict__.get("_lane_folded") is not None:
                tensors.append(self._lane_folded)
            if self.out.data_ptr() != self._state_and_out[self.ns + 2:].data_ptr():
                tensors.append(self.out)                                       # (use_out: an output block of the caller's)
            for t in tensors:
                t.copy_(t.index_select(t.dim() - 1, order))
            if self._env_of_lane is None:
                object.__setattr__(self, "_env_of_lane", torch.arange(N, dtype=torch.int32, device=self.device))
                self._drop_plan()              # (the reset block now names the env ids: pvder_autoreset.env_ids)
            self._env_of_lane.copy_(self._env_of_lane.index_select(0, order))  # in place: plans keep the pointer
            inv = torch.empty(N, dtype=torch.int64, device=self.device)
            inv[self._env_of_lane.long()] = torch.arange(N, dtype=torch.int64, device=self.device)
            object.__setattr__(self, "_lane_of_env", inv)
            object.__setattr__(self, "_ksub_hint", -1)                         # no common clock any more
            if self._plan is not None:
                self._lib.pvder_plan_hint(self._plan, -1)
            self._set_desync(True)
        if lane_io is not None:
            self.lane_io = bool(lane_io)

    @property
    def env_of_lane(self) -> Optional[torch.Tensor]:
        return self._env_of_lane

    @property
    def lane_of_env(self) -> Optional[torch.Tensor]:
        return self._lane_of_env

    def to_lane_order(self, t: torch.Tensor) -> torch.Tensor:
        """[..., N] in env order -> lane order (identity while the batch is unsorted)."""
        return t if self._env_of_lane is None else t.index_select(t.dim() - 1, self._env_of_lane.long())

    def to_env_order(self, t: torch.Tensor) -> torch.Tensor:
        """[..., N] in lane order (``self.out``, ``self.fstate`` ...) -> env order."""
        return t if self._lane_of_env is None else t.index_select(t.dim() - 1, self._lane_of_env)

    def _mapped(self):
        return self._env_of_lane is not None and not self.lane_io

    def _device_ctx(self):
        """``torch.cuda.device(self.device)``, or nothing when it is the current device already."""
        if torch.cuda.current_device() == self.device.index:
            return contextlib.nullcontext()
        return torch.cuda.device(self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _autoreset(self, enabled: bool) -> _native.AutoReset:
        a = _native.AutoReset()
        a.enabled = 1 if enabled else 0
        a.keep_events = 1 if self._keep_events else 0
        a.seed = self.seed & (2 ** 64 - 1)
        a.env_id_base = self.env_id_base
        a.spec = self._spec()
        a.y0 = C.cast(self._y0_c, C.POINTER(C.c_double))
        eol = self.__dict__.get("_env_of_lane")
        a.env_ids = eol.data_ptr() if eol is not None else None
        a.lazy_events = 1 if (self._lazy_active and not self._keep_events) else 0
        return a

    # -- the event tables [K+1, N] (row 0 = nominal; create_random_events, PVDER_env.py:400-411) ----------
    def _complete_tables(self):
        """With ``lazy_events`` the step kernels write an episode's table rows as the envs come to them (pvder_autoreset.lazy_events):
        a READER gets them whole -- every row of the episode each env is in (pvder_complete_event_tables; idempotent)."""
        if not (self._lazy_active and self.auto_reset) or self._keep_events or self.__dict__.get("_ev_vg") is None:
            return
        cfg, ar = self._cfg(), self._autoreset(True)
        with self._device_ctx():
            rc = self._lib.pvder_complete_event_tables(self._par_c, C.byref(cfg), C.byref(ar), self.istate.data_ptr(),
                                                       self._ev_vg.data_ptr(), self._ev_si.data_ptr(), self.num_envs, self._stream())
        _native.check(rc, "pvder_complete_event_tables")

    def _set_desync(self, flag: bool):
        """The envs' clocks differ / are one: with ``lazy_events=None`` the table rows go lazy / whole (a new plan; whole tables
        first when the laziness ends)."""
        object.__setattr__(self, "_desync", bool(flag))
        want = bool(self.lazy_events) if self.lazy_events is not None else (bool(flag) and bool(self.auto_reset))
        if want != self._lazy_active:
            if not want:
                self._complete_tables()
            self._lazy_active = want          # (a plan key: the plan is folded anew)

    @property
    def ev_vg(self) -> torch.Tensor:
        """Grid-voltage event table (fp32 [K+1, N], lane order): row j holds the magnitude in force from event j on."""
        self._complete_tables()
        return self._ev_vg

    @property
    def ev_si(self) -> torch.Tensor:
        """Insolation event table (fp32 [K+1, N], lane order)."""
        self._complete_tables()
        return self._ev_si

    def _reset_launch(self, mask: Optional[torch.Tensor]):
        fn = self._lib.pvder_reset_masked_f32 if self.fdtype == torch.float32 else self._lib.pvder_reset_masked_f64
        cfg, ar = self._cfg(), self._autoreset(True)
        lanes = self.__dict__.get("_lane_folded")
        if lanes is not None:
            # per-lane parameters: every env restarts from ITS OWN steady state (PVDER_env.py:374-378) and its reset observation
            # carries its own Ppv -- the same observation the same-step auto-reset of the lane kernels returns
            rc = self._lib.pvder_reset_masked_lanes_f32(self._par_c, C.byref(cfg), C.byref(ar), self.fstate.data_ptr(),
                                                        self.istate.data_ptr(), self._ev_vg.data_ptr(), self._ev_si.data_ptr(),
                                                        mask.data_ptr() if mask is not None else None, self.out.data_ptr(),
                                                        self.num_envs, lanes.data_ptr(), self._stream())
            _native.check(rc, "pvder_reset_masked_lanes_f32")
            return
        rc = fn(self._par_c, C.byref(cfg), C.byref(ar), self.fstate.data_ptr(), self.istate.data_ptr(),
                self._ev_vg.data_ptr(), self._ev_si.data_ptr(), mask.data_ptr() if mask is not None else None,
                self.out.data_ptr(), self.num_envs, self._stream())
        _native.check(rc, "pvder_reset_masked")

    # -- Gym-like API (batched) ---------------------------------------------------------------------
    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reset all envs (or those with ``mask[e] != 0``); returns the [N, 11] observation view."""
        with torch.cuda.device(self.device):
            if mask is not None:
                mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
                assert mask.shape == (self.num_envs,)
                if self._mapped():
                    mask = self.to_lane_order(mask).contiguous()
            self._set_desync(mask is not None or self._env_of_lane is not None)
            self._reset_launch(mask)
            # the plan is checked at every reset (never inside a graph capture): specs edited in place
            # (env_events_spec[...]['ENABLE'], env_reward_spec['reference_values'], my_spec lists) take effect here at
            # the latest -- attribute assignments and update_env_events / set_reward_spec drop the plan at once -- and it
            # is folded anew only if one of them changed (_plan_fingerprint).
            # Coefficient uploads happen now too: step() must stay capturable into a HIP graph
            if self._plan is None or self.__dict__.get("_plan_fp") != self._plan_fingerprint():
                self._drop_plan()
                if mask is None:
                    self._ksub_hint = 0      # every env starts its episode here (a masked reset leaves the others' clock the hint)
                self._make_plan()
            elif mask is None:
                self._ksub_hint = 0
                self._lib.pvder_plan_hint(self._plan, -1 if os.environ.get("PVDER_NO_HINT") else 0)
            if self.hstep is not None:
                if mask is None:
                    self.hstep.zero_()
                else:
                    self.hstep.masked_fill_(mask.bool(), 0.0)
        self._needs_reset = False
        if self._traj is not None:
            if mask is None:
                self._traj_n = 0           # a full reset starts a new recording (the sim object is rebuilt, :323-331)
            self._traj_append()
        return self.obs

    def step(self, actions: torch.Tensor, return_tensors: bool = True):
        """One transition of every env.  ``actions``: [N] integer tensor with values in 0..4
        (Discrete(5), PVDER_env.py:50).  Returns (obs[N,11], reward[N], done[N] bool, info); obs and
        reward are views of the output block ``self.out`` ([13, N]), valid until the next step.
        ``return_tensors=False`` only enqueues the launch (results are in ``self.out``)."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        if actions.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        if actions.dtype != torch.uint8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        if self._mapped():
            actions = self.to_lane_order(actions).contiguous()
        plan = self._plan or self._make_plan()      # cfg / reset block / coefficients folded once (C side)
        with self._device_ctx():
            timed = False
            if self._timing:
                self._timing_k += 1
                timed = self._timing_k % self._timing_every == 0
            if timed:
                ev0, ev1 = self._event_pair()
                ev0.record()          # same stream the kernel is launched on (torch's current stream)
            # ONE launch: action -> integrate -> reward/done/obs -> same-step auto-reset of finished
            # envs (their obs rows hold the reset observation, reward/done rows the ended transition);
            # statistics: every workgroup overwrites its slot (no atomics, nothing to clear)
            rc = self._lib.pvder_plan_step(plan, self.fstate.data_ptr(), self.istate.data_ptr(), self._ev_vg.data_ptr(),
                                           self._ev_si.data_ptr(), actions.data_ptr(), self.out.data_ptr(),
                                           self.hstep.data_ptr() if self.hstep is not None else None,
                                           self.stats.data_ptr(), self._stream())
            if timed:
                ev1.record()
                self._events.append((ev0, ev1))
            _native.check(rc, "pvder_step")
            if self._traj is not None:
                self._traj_append()
            if not return_tensors:
                return None
            if self._mapped():
                o = self.to_env_order(self.out)
                return o[:11].t(), o[11], o[12] != 0, {}
            return self.obs, self.out[11], self.out[12] != 0, {}

    def rollout(self, actions: torch.Tensor, out: Optional[torch.Tensor] = None, fused: Optional[bool] = None):
        """``K`` transitions from ONE host call: ``actions`` is a [K, N] uint8 device tensor (row k =
        the actions of step k -- an open-loop action sequence: recorded actions, fixed policies of
        ``calc_returns``, the synthetic random-action rollouts of bench.py).  The same fused launch as
        ``step``, enqueued K times by the C side (pvder_plan_rollout): per-step results land in
        ``out[k]`` ([K, 13, N]) when given, otherwise every step overwrites ``self.out``.
        ``fused=True`` (etdlin, three-phase model): the K env steps run inside ONE launch with the state
        in registers from step to step (pvder_plan_rollout_fused) -- the same results bit for bit;
        ``None`` picks it when it applies and K >= 4."""
        if self._needs_reset:
            raise RuntimeError("call reset() before rollout()")
        if actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise ValueError("actions must have shape (K, %d)" % self.num_envs)
        if actions.dtype != torch.uint8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        if self._mapped():
            actions = self.to_lane_order(actions).contiguous()
        K = int(actions.shape[0])
        stride = 0
        dst = self.out
        if out is not None:
            if tuple(out.shape) != (K, 13, self.num_envs) or out.dtype != self.out.dtype or not out.is_contiguous() \
                    or out.device != self.device:
                raise ValueError("out must be a contiguous %s tensor of shape (%d, 13, %d)" % (self.out.dtype, K, self.num_envs))
            dst, stride = out, 13 * self.num_envs
        if self._traj is not None:
            for k in range(K):            # trajectory capture copies after every step
                self.step(actions[k], return_tensors=False)
                if out is not None:
                    out[k].copy_(self.out)
            return
        plan = self._plan or self._make_plan()
        can_fuse = self._plan_integrator == "etdlin" and self.ns == 23 and not self._timing and self.__dict__.get("_lane_folded") is None
        if fused and not can_fuse:
            raise ValueError("fused rollouts: integrator='etdlin', three-phase model, no per-launch timing, no per-env parameter sets")
        if fused is None:
            # (while an env is at / near the duty-cycle limit the hybrid kernel follows every step launch -- the fused kernels cannot
            #  host it: they would step such an env on the held operator sets, 1e-2 off: include/pvder_hip.h pvder_plan_hybrid_active)
            fused = can_fuse and K >= 4 and not self._lib.pvder_plan_hybrid_active(plan)
        with self._device_ctx():
            if out is not None and K > 0 and not self.auto_reset:
                out[0].copy_(self.out)               # stale tuples of finished envs: block k takes them from block k-1
            if fused:
                rc = self._lib.pvder_plan_rollout_fused(plan, self.fstate.data_ptr(), self.istate.data_ptr(),
                                                        self._ev_vg.data_ptr(), self._ev_si.data_ptr(), actions.data_ptr(), K,
                                                        dst.data_ptr(), stride, self.stats.data_ptr(),
                                                        self._progress.data_ptr(), self._stream())
                _native.check(rc, "pvder_plan_rollout_fused")
            else:
                rc = self._lib.pvder_plan_rollout(plan, self.fstate.data_ptr(), self.istate.data_ptr(), self._ev_vg.data_ptr(),
                                                  self._ev_si.data_ptr(), actions.data_ptr(), K, dst.data_ptr(), stride,
                                                  self.hstep.data_ptr() if self.hstep is not None else None,
                                                  self.stats.data_ptr(), self._stream())
                _native.check(rc, "pvder_plan_rollout")
            if out is not None and K > 0:
                self.out.copy_(out[K - 1])
                if self._mapped():
                    out.copy_(out.index_select(2, self._lane_of_env))      # the caller's blocks in ENV order (self.out stays in lane order)

    def _served_alloc(self):
        """The protocol words of rollout_served (pvder_served_alloc: pinned host words + device words), once per env."""
        w = self.__dict__.get("_served_words")
        if w is None:
            host, dev, scr = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
            with self._device_ctx():
                _native.check(self._lib.pvder_served_alloc(self.num_envs, C.byref(host), C.byref(dev), C.byref(scr)), "pvder_served_alloc")
            w = (host, dev, scr)
            object.__setattr__(self, "_served_words", w)
        return w

    def rollout_served(self, actions: torch.Tensor, out: Optional[torch.Tensor] = None, depth: Optional[int] = None,
                       fill=None, timeout_s: float = 60.0):
        """``K`` transitions with the env state resident in registers (the fused rollout's loop) for a producer that does NOT have
        its actions ahead: env step k is taken once row k of ``actions`` ([K, N] uint8, this env's device, LANE order if the lanes
        are sorted) has been POSTED (pvder_plan_rollout_served, include/pvder_hip.h).  Same results as ``rollout`` bit for bit.
        ``depth=None``: every row is there already -- posted before the launch (4.9 us per env step at 65 536 envs; one launch per
        step: 8.5).  ``depth=D >= 1``: row k is posted once step k - D is reported done by every workgroup, after ``fill(k)`` --
        the producer's hook: it writes ``actions[k]`` (stream-ordered work on ANOTHER stream than the current one, finished
        before it returns: the current stream holds the served launches) -- has returned; the output rows of the steps reported
        done are in memory for a kernel that starts on that other stream (``out[k - D - 1]`` and earlier, inside ``fill(k)``).  D = 8
        costs 5.5 us per step, D = 1 the PCIe round trip (27 us): what a host in the loop costs whatever it launches; an env that
        leaves the standard path (current limit, off balance) or ends its episode ends the resident launch for ONE step of the
        per-step body -- a launch pair and a host round trip, as often as that happens.  Blocks until the rollout is done (host
        sync); not capturable.  If ``fill`` raises, the launch is let go on whatever ``actions`` holds (codes that are not actions
        are no-ops the statistics count) and the exception propagates: ``reset()`` the env before using it again.  etdlin, three-phase model, one parameter set, no env at the duty-cycle limit
        (``pvder_plan_hybrid_active``: ValueError)."""
        if self._needs_reset:
            raise RuntimeError("call reset() before rollout_served()")
        if actions.dim() != 2 or actions.shape[1] != self.num_envs or actions.dtype != torch.uint8 or actions.device != self.device \
                or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous uint8 tensor of shape (K, %d) on %s" % (self.num_envs, self.device))
        if depth is not None and (not isinstance(depth, int) or depth < 1):
            raise ValueError("depth must be None (every row posted ahead) or an integer >= 1")
        K = int(actions.shape[0])
        stride, dst = 0, self.out
        if out is not None:
            if tuple(out.shape) != (K, 13, self.num_envs) or out.dtype != self.out.dtype or not out.is_contiguous() \
                    or out.device != self.device:
                raise ValueError("out must be a contiguous %s tensor of shape (%d, 13, %d)" % (self.out.dtype, K, self.num_envs))
            dst, stride = out, 13 * self.num_envs
        plan = self._plan or self._make_plan()
        if not (self._plan_integrator == "etdlin" and self.ns == 23 and not self._timing and self.__dict__.get("_lane_folded") is None
                and self._traj is None):
            raise ValueError("served rollouts: integrator='etdlin', three-phase model, no per-launch timing, no per-env parameter sets")
        if self._lib.pvder_plan_hybrid_active(plan):
            raise ValueError("an env is at / near the duty-cycle limit: its steps belong to the follow-up kernel -- use rollout()")
        if self._mapped():
            raise ValueError("sorted lanes with env-order I/O: rollout_served takes and gives LANE order (sort_lanes(lane_io=True))")
        if K == 0:
            return
        host, dev, scr = self._served_alloc()
        for i in range(32):
            host[i] = 0
        if depth is None:
            host[0] = K
        import time as _time
        with self._device_ctx():
            if out is not None and not self.auto_reset:
                out[0].copy_(self.out)
            pair_done = torch.cuda.Event()
            posted, resume, deadline = (K if depth is None else 0), 0, _time.perf_counter() + timeout_s
            try:
                while True:
                    # ONE pair of launches (the served loop until a workgroup meets a step it cannot carry, the finish kernel for that
                    # step), the rows posted while it runs; then `progress` says whether there is another pair to launch
                    rc = self._lib.pvder_plan_rollout_served(plan, self.fstate.data_ptr(), self.istate.data_ptr(), self._ev_vg.data_ptr(),
                                                             self._ev_si.data_ptr(), actions.data_ptr(), K, dst.data_ptr(), stride,
                                                             self.stats.data_ptr(), self._progress.data_ptr(), C.cast(dev, C.c_void_p),
                                                             C.cast(scr, C.c_void_p), 0 if depth is None else 1, resume, self._stream())
                    _native.check(rc, "pvder_plan_rollout_served")
                    pair_done.record()
                    while not pair_done.query():
                        if posted < K and posted <= host[16] + depth:        # (host[16]: the slowest workgroup's "steps < d are out")
                            if fill is not None:
                                fill(posted)
                            posted += 1
                            host[0] = posted
                        elif _time.perf_counter() > deadline:
                            raise TimeoutError("rollout_served: the launch stopped reporting (%d of %d rows posted)" % (posted, K))
                    behind = int(self._progress.min().item())
                    if behind >= K or depth is None:                         # (every row posted ahead: ONE call enqueued the whole rollout)
                        if behind < K:
                            raise RuntimeError("rollout_served: %d of %d steps taken" % (behind, K))
                        break
                    if _time.perf_counter() > deadline:
                        raise TimeoutError("rollout_served: %d of %d steps taken, %d rows posted" % (behind, K, posted))
                    resume = 1
            finally:
                host[0] = K                                                  # (whatever happened: nobody is left waiting)
                torch.cuda.synchronize(self.device)
            if out is not None:
                self.out.copy_(out[K - 1])

    def step_stats(self) -> Dict[str, float]:
        """Wave-reduced statistics of the last step (host sync): reward sum, #done, #non-finite,
        #invalid actions -- the batched counterpart of update_reward_stats (env_utilities.py:32-38)."""
        s = self.stats.sum(dim=0, dtype=torch.float64).tolist()
        return {"reward_sum": s[0], "n_done": int(s[1]), "n_nonfinite": int(s[2]), "n_bad_action": int(s[3])}

    # -- checkpoint / resume (the reference has none: reset() rebuilds the sim) ----
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Everything a bit-exact resume needs: ODE states + refs, counters, event tables, RK45 step
        sizes and the per-workgroup statistics (clones, safe to torch.save)."""
        d = {"fstate": self.fstate.clone(), "istate": self.istate.clone(), "ev_vg": self.ev_vg.clone(),
             "ev_si": self.ev_si.clone(), "out": self.out.clone(), "stats": self.stats.clone()}
        if self.hstep is not None:
            d["hstep"] = self.hstep.clone()
        if self._env_of_lane is not None:
            d["env_of_lane"] = self._env_of_lane.clone()
        return d

    def load_state_dict(self, d: Dict[str, torch.Tensor]):
        for k in ("fstate", "istate", "ev_vg", "ev_si", "out", "stats"):
            t = getattr(self, {"ev_vg": "_ev_vg", "ev_si": "_ev_si"}.get(k, k))       # (the buffers themselves: nothing to complete)
            if d[k].shape != t.shape or d[k].dtype != t.dtype:
                raise ValueError("state_dict entry %r does not match this env (%s %s vs %s %s)"
                                 % (k, tuple(d[k].shape), d[k].dtype, tuple(t.shape), t.dtype))
            t.copy_(d[k])
        if self.hstep is not None:
            self.hstep.copy_(d["hstep"])
        if "env_of_lane" in d:
            if self._env_of_lane is None:
                object.__setattr__(self, "_env_of_lane", torch.arange(self.num_envs, dtype=torch.int32, device=self.device))
                self._drop_plan()
            self._env_of_lane.copy_(d["env_of_lane"])
            inv = torch.empty(self.num_envs, dtype=torch.int64, device=self.device)
            inv[self._env_of_lane.long()] = torch.arange(self.num_envs, dtype=torch.int64, device=self.device)
            object.__setattr__(self, "_lane_of_env", inv)
        elif self._env_of_lane is not None:
            raise ValueError("this env's lanes are sorted (sort_lanes) and the state_dict holds no env_of_lane")
        self._needs_reset = False
        with self._device_ctx():
            hint = int(self.istate[0, 0].item())                 # (PVDER_I_KSUB of env 0: where a batch in lockstep is)
            self._set_desync(self._env_of_lane is not None or bool((self.istate[0] != hint).any().item()))
            if self._plan is None or self.__dict__.get("_plan_fp") != self._plan_fingerprint():
                self._drop_plan()
                self._ksub_hint = hint
                self._make_plan()
            else:
                self._ksub_hint = hint
                self._lib.pvder_plan_hint(self._plan, -1 if os.environ.get("PVDER_NO_HINT") else hint)

    # -- per-launch timing of the step kernel (bench.py roofline) ------------------------------------
    _timing = False

    _timing_every = 1
    _timing_k = 0

    def kernel_timing(self, on: bool = True, every: int = 1):
        """Bracket every ``every``-th step-kernel launch with HIP events on the launch stream (each
        event is a packet of its own on the queue: ~1 us of stream time apiece)."""
        self._timing = bool(on)
        self._timing_every = max(1, int(every))
        self._timing_k = 0
        self._events = []

    _event_pool: List[tuple] = []

    def reserve_timing_events(self, n: int):
        """Create (and warm) ``n`` event pairs now: the first use of a torch event costs tens of
        microseconds of host time, which a short timed window should not pay inside the window."""
        with self._device_ctx():
            self._event_pool = []
            for _ in range(int(n)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); b.record()
                self._event_pool.append((a, b))
            torch.cuda.synchronize(self.device)

    def _event_pair(self):
        if self._event_pool:
            return self._event_pool.pop()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def kernel_times_ms(self) -> List[float]:
        """Durations of the step-kernel launches recorded since kernel_timing(True) (host sync)."""
        torch.cuda.synchronize(self.device)
        return [a.elapsed_time(b) for a, b in getattr(self, "_events", [])]

    # -- trajectory capture (the reference keeps every solution row in `self.results`, PVDER_env.py:389,
    #    and plots it in render('human'), :358-364; here: an optional device ring buffer at env-step
    #    resolution, filled by device-to-device copies behind each launch, read back only on demand) ----
    _traj = None
    _traj_n = 0

    def record_trajectory(self, capacity: int, env_ids=None):
        """Keep the last ``capacity`` snapshots (reset + env steps) of the listed envs (default: all)
        on the device: states, refs, output block and time index.  ``capacity = 0`` turns recording
        off.  Not on by default: it adds three small device-to-device copies per step."""
        if capacity <= 0:
            self._traj = None
            return
        self._traj_ids = None if env_ids is None else torch.as_tensor(env_ids, dtype=torch.long, device=self.device)
        m = self.num_envs if env_ids is None else int(self._traj_ids.numel())
        self._traj = torch.zeros((int(capacity), self.ns + 2 + 13, m), dtype=self.fdtype, device=self.device)
        self._traj_n = 0

    def _traj_append(self):
        ids, ns = self._traj_ids, self.ns
        slot = self._traj[self._traj_n % self._traj.shape[0]]
        if ids is None:
            if self.out.data_ptr() == self._state_and_out[ns + 2:].data_ptr():
                slot[: ns + 15].copy_(self._state_and_out)           # (the env's own output block: ONE copy for both)
            else:
                slot[: ns + 2].copy_(self.fstate)
                slot[ns + 2: ns + 15].copy_(self.out)
        else:
            slot[: ns + 2].copy_(self.fstate.index_select(1, ids))
            slot[ns + 2: ns + 15].copy_(self.out.index_select(1, ids))
        self._traj_n += 1

    def trajectory(self, env: int = 0) -> Dict[str, np.ndarray]:
        """Recorded time series of recorded env number ``env`` in PHYSICAL units (host sync): ``t`` [s],
        the fields of ``der_model.physical_outputs`` (m*, i*, v*, vt*, S_inverter, S_PCC, Vdc, Q_ref,
        Vdc_ref), ``Ppv`` [W], ``vgrid`` [pu], ``reward``, ``done``.  Oldest snapshot first."""
        if self._traj is None:
            raise RuntimeError("record_trajectory(capacity) was not called")
        cap = self._traj.shape[0]
        n = min(self._traj_n, cap)
        order = [(self._traj_n - n + k) % cap for k in range(n)]
        a = self._traj[order, :, env].to(torch.float64).cpu().numpy().T          # [rows, n]
        ns = self.ns
        # grid-voltage magnitude in force, from the recorded PCC voltage: v_a = kvg vg + kvi i_a
        P = der_model.IDX
        kvg = complex(self.par[P["KVG_RE"]], self.par[P["KVG_IM"]]) * self.par[P["VG_SCALE"]]
        kvi = complex(self.par[P["KVI_RE"]], self.par[P["KVI_IM"]])
        va = a[ns + 2 + 2] + 1j * a[ns + 2 + 3]
        vgrid = np.real((va - kvi * (a[0] + 1j * a[1])) / kvg)
        d = der_model.physical_outputs(self.par, a[:ns], vgrid, a[ns], a[ns + 1])
        # the clock from the observation's own time entry (row 10 = tStart / max_sim_time, PVDER_env.py:540), snapped to the
        # sim-time-step grid: no separate copy of the counter row per snapshot
        d["t"] = np.round(a[ns + 2 + 10] * self._max_sim_time / SIM_TIME_STEP) * SIM_TIME_STEP
        d["vgrid"] = vgrid                          # grid-voltage event magnitude in force [pu]
        d["Ppv"] = a[ns + 2 + 7] * der_model.SBASE
        d["reward"], d["done"] = a[ns + 2 + 11], a[ns + 2 + 12] != 0
        return d

    def close(self):
        pass
