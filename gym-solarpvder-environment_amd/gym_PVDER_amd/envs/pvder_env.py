"""Single-environment ``PVDER`` with the reference's old-gym API, running on the HIP kernels.

Drop-in for /root/reference/gym_PVDER/envs/PVDER_env.py ``class PVDER`` (:39): same constructor
arguments (:104-105), ``reset() -> np.ndarray(11,)`` (:316-334), ``step(action) -> (obs, reward,
done, {})`` (:138-196) with a Python ``int`` reward when DISCRETE_REWARD else ``float``
(:301; gym_PVDER/tests/test_gym_PVDER.py:66,81) and a Python ``bool`` done (test :35),
``render(mode)`` for ``metadata['render.modes']`` (:42, :336-364), ``close()``,
``update_env_events`` (:413-438), ``steps`` (env_utilities.py:48-52) and the ``env.sim`` /
``env.sim.PV_model`` attribute surface the tests and examples touch (test :108, :124;
examples/gym_PVDER_environment_import_test.py:38).  It is a PVDERVec with num_envs = 1 -- the same
fused launch, no separate code path.
"""
from __future__ import annotations

import logging
import os
import pprint
import time
from types import SimpleNamespace

import numpy as np
import torch

from .. import _native, der_model
from ..spaces import make_spaces
from ..vec_env import EnvSpecMixin, PVDERVec, SIM_TIME_STEP, OBSERVED_QUANTITIES

logger = logging.getLogger("gym_PVDER_amd")


def _gym_env_base():
    """``gym.Env`` (or ``gymnasium.Env``) when such a package is importable (else None) -- the reference is ``class PVDER(gym.Env, Logging,
    Utilities)`` (PVDER_env.py:39) and wrappers / checkers test ``isinstance(env, gym.Env)`` --."""
    import importlib
    for mod in ("gym", "gymnasium"):
        try:
            return importlib.import_module(mod).Env
        except ImportError:
            continue
    return None


_GymEnv = _gym_env_base()
_BASES = (EnvSpecMixin,) if _GymEnv is None else (_GymEnv, EnvSpecMixin)


class _EventsView:
    """``env.sim.simulation_events`` -- exposes ``_events_spec`` (test_gym_PVDER.py:124)."""

    def __init__(self, env):
        self._env = env

    @property
    def _events_spec(self):
        return self._env.env_events_spec

    def update_events_spec(self, events_spec):
        self._env.update_env_events([{k: dict(v)} for k, v in events_spec.items()])


class _PVModelView:
    """``env.sim.PV_model`` -- the attributes PVDER_env.py reads back (:225-299, :339-348, :535-539)."""

    def __init__(self, env):
        self._env = env
        self.Sbase, self.Vdcbase = der_model.SBASE, der_model.VDCBASE
        self.Vrms_ref = float(env._vec.par[der_model.IDX["VRMS_REF"]])

    def _o(self, k):
        return self._env._host_out[k]

    @property
    def Q_ref(self):
        return float(self._env._vec.Q_ref[0])

    @Q_ref.setter
    def Q_ref(self, v):
        self._env._vec.Q_ref[0] = v

    @property
    def Vdc_ref(self):
        return float(self._env._vec.Vdc_ref[0])

    @Vdc_ref.setter
    def Vdc_ref(self, v):
        self._env._vec.Vdc_ref[0] = v

    ia = property(lambda s: complex(s._o(0), s._o(1)))
    va = property(lambda s: complex(s._o(2), s._o(3)))
    S_PCC = property(lambda s: complex(s._o(4), s._o(5)))
    Vdc = property(lambda s: s._o(6))
    Ppv = property(lambda s: s._o(7))


class _SimView:
    """``env.sim`` -- tStart/tStop/tInc/Sbase/Vbase/Ibase/name (PVDER_env.py:143-196, :340-349)."""

    def __init__(self, env):
        self._env = env
        self.tInc = SIM_TIME_STEP                                     # :396
        self.Sbase, self.Vbase, self.Ibase = der_model.SBASE, der_model.VBASE, der_model.IBASE
        self.name = 'sim_' + env.name
        self.simulation_events = _EventsView(env)
        self.PV_model = _PVModelView(env)
        self.SOLVER_CONVERGENCE = True
        self.jacFlag = True         # :386 (here: the exponential integrators are built on the model's analytic Jacobian)

    @property
    def tStart(self):
        return self._env._host_ist[0] * SIM_TIME_STEP

    @property
    def tStop(self):
        return self.tStart


class PVDER(*_BASES):
    """Old-gym style single env; see module docstring."""

    count = 0
    metadata = {'render.modes': ['vector', 'human']}                 # :42
    observed_quantities = OBSERVED_QUANTITIES
    action_space, observation_space = make_spaces()                  # :50-51
    reward_range = (-float('inf'), float('inf'))
    spec = None

    def __init__(self, goals_list=('voltage_regulation',), n_sim_time_steps_per_env_step=15,
                 max_sim_time=40.0, DISCRETE_REWARD=True, verbosity='INFO', device="cuda",
                 integrator=None, micro_steps=None, seed=0, model_type='model_2', record_trajectory=True):
        PVDER.count += 1                                             # :117
        self.name = 'GymDER_' + str(PVDER.count)
        self.verbosity = verbosity
        kw = {} if micro_steps is None else {"micro_steps": micro_steps}
        self._vec = PVDERVec(1, goals_list=goals_list,
                             n_sim_time_steps_per_env_step=n_sim_time_steps_per_env_step,
                             max_sim_time=max_sim_time, DISCRETE_REWARD=DISCRETE_REWARD, device=device,
                             integrator=integrator, auto_reset=False, seed=seed, model_type=model_type,
                             env_id_base=PVDER.count - 1, **kw)
        self.max_sim_time_user = max_sim_time
        self._actions = [torch.full((1,), a, dtype=torch.uint8, device=self._vec.device) for a in range(5)]
        # host mirrors of the output block and the counters: pinned, filled by two asynchronous copies behind the launch
        self._pin_out = torch.zeros(13, dtype=self._vec.fdtype).pin_memory()
        self._pin_ist = torch.zeros(4, dtype=torch.int32).pin_memory()
        # ... and of a step: 13 output rows + 4 counters + a sequence word the device writes itself (pvder_plan_step_mirror)
        self._pin_mirror = torch.zeros(_native.MIRROR_WORDS, dtype=torch.float64).pin_memory()
        self._mirror_seq = 0
        self._use_mirror = os.environ.get("PVDER_SINGLE_ENV_COPIES", "0") != "1"            # (A/B: the copy-packet read-back of ABI 15)
        self._host_out, self._host_ist = [0.0] * 13, [0] * 4
        if record_trajectory:      # the reference always keeps the solution rows (self.results, :389)
            self._vec.record_trajectory(self._vec.max_episode_steps + 1)
        self.env_events_spec = self._vec.env_events_spec
        self.env_goal_spec = self._vec.env_goal_spec          # the instance's own copy (the class keeps the defaults)
        self.env_reward_spec = self._vec.env_reward_spec
        self._reward = 0
        self.done = False
        self._step_time = 0.0
        self._time_stats = {'step': [], 'step_time': []}
        self._reward_stats = {'step': [], 'reward': []}
        self._action_stats = {i: 0 for i in range(5)}
        self.CONVERGENCE_FAILURE = False
        self.RUNTIME_ERROR = False
        logger.info('%s:Environment created!', self.name)            # :132

    # -- mirrors of the reference's validated properties (:544-620) --------------------------------
    n_sim_time_steps_per_env_step = property(lambda s: s._vec.n_sim_time_steps_per_env_step)
    max_sim_time = property(lambda s: s._vec.max_sim_time)
    goals_list = property(lambda s: s._vec.goals_list)
    DISCRETE_REWARD = property(lambda s: s._vec.DISCRETE_REWARD)
    unwrapped = property(lambda s: s)
    env = property(lambda s: s)            # examples use env.env.sim (gym TimeLimit wrapper idiom)

    def _read_back(self):
        """The env's output block and counters (17 words) into the host mirrors every accessor below reads: two asynchronous
        copies into PINNED memory behind the step launch and ONE stream wait, enqueued from C (pvder_fetch_env) -- no pack kernels,
        no pageable staging copy, no synchronising read per attribute (round 4: two device copies + a blocking .tolist(): 58 us
        per step)."""
        v = self._vec
        with v._device_ctx():       # the C side launches on the CURRENT device (null stream): make that the env's
            _native.check(v._lib.pvder_fetch_env(v.out.data_ptr(), v.istate.data_ptr(), 1, 0, 1 if v.fdtype == torch.float64 else 0,
                                                 self._pin_out.data_ptr(), self._pin_ist.data_ptr(), v._stream()), "pvder_fetch_env")
        self._host_out, self._host_ist = self._pin_out.tolist(), self._pin_ist.tolist()

    def _step_and_fetch(self, action: int):
        """``PVDERVec.step`` + the trajectory snapshot + ``_read_back`` from ONE native call (pvder_plan_step_fetch): what were
        three Python-level calls of ~10 us each.  Falls back to those when the vec env is in a state the fused call does not
        cover (per-launch timing, an output block that is not the env's own)."""
        v = self._vec
        own_out = v.out.data_ptr() == v._state_and_out[v.ns + 2:].data_ptr()
        if v._timing or not own_out or v._needs_reset:
            v.step(self._actions[action], return_tensors=False)
            self._read_back()
            return
        with v._device_ctx():       # plan creation, launch and the host-mirror pointers all resolve against the CURRENT device
            self._launch_and_fetch(v, action)

    def _launch_and_fetch(self, v, action: int):
        plan = v._plan or v._make_plan()
        snap, nbytes = None, 0
        if v._traj is not None:
            slot = v._traj[v._traj_n % v._traj.shape[0]]
            snap, nbytes = slot.data_ptr(), slot.numel() * slot.element_size()
        if self._use_mirror:
            # no copy packet, no stream wait: one small kernel behind the step writes rows, counters, the trajectory record and a
            # sequence word into pinned host memory; the call returns when the word is there
            rc = v._lib.pvder_plan_step_mirror(plan, v.fstate.data_ptr(), v.istate.data_ptr(), v._ev_vg.data_ptr(), v._ev_si.data_ptr(),
                                               self._actions[action].data_ptr(), v.out.data_ptr(),
                                               v.hstep.data_ptr() if v.hstep is not None else None, v.stats.data_ptr(), snap,
                                               v._state_and_out.data_ptr(), nbytes, self._pin_mirror.data_ptr(), self._mirror_seq + 1,
                                               v._stream())
            _native.check(rc, "pvder_plan_step_mirror")
            self._mirror_seq += 1
            if snap is not None:
                v._traj_n += 1          # (only behind a step that ran: a failed call leaves no hole in the trajectory ring)
            m = self._pin_mirror.tolist()
            self._host_out, self._host_ist = m[:13], [int(x) for x in m[13:17]]
            return
        rc = v._lib.pvder_plan_step_fetch(plan, v.fstate.data_ptr(), v.istate.data_ptr(), v._ev_vg.data_ptr(), v._ev_si.data_ptr(),
                                          self._actions[action].data_ptr(), v.out.data_ptr(),
                                          v.hstep.data_ptr() if v.hstep is not None else None, v.stats.data_ptr(), snap,
                                          v._state_and_out.data_ptr(), nbytes, self._pin_out.data_ptr(), self._pin_ist.data_ptr(),
                                          v._stream())
        _native.check(rc, "pvder_plan_step_fetch")
        if snap is not None:
            v._traj_n += 1
        self._host_out, self._host_ist = self._pin_out.tolist(), self._pin_ist.tolist()

    @property
    def steps(self):                                                 # env_utilities.py:48-52
        return self._host_ist[1] if hasattr(self, 'sim') else 0

    @property
    def state(self):                                                 # :531-542
        return tuple(self._host_out[:11])

    def reset(self):                                                 # :316-334
        self._vec.reset()
        self._read_back()
        self.sim = _SimView(self)
        self._reward, self.done = 0, False
        self.CONVERGENCE_FAILURE = self.RUNTIME_ERROR = False
        self._time_stats = {'step': [], 'step_time': []}
        self._reward_stats = {'step': [], 'reward': []}
        self._action_stats = {i: 0 for i in range(5)}
        return np.array(self.state)

    def step(self, action):                                          # :138-196
        t0 = time.time()
        if not hasattr(self, 'sim'):
            raise RuntimeError("call reset() before step()")
        assert action in self.action_space, \
            'The action "{}" is not available in the environment action space!'.format(action)   # :201
        if self.done:                                                # :145-152: stale tuple
            logger.warning('%s:Simulation completed - Reset environment to start new simulation!', self.name)
            return np.array(self.state), self._reward, self.done, {}
        self._action_stats[int(action)] += 1                         # :209
        self._step_and_fetch(int(action))                            # launch + trajectory snapshot + read-back: ONE native call
        o, ist = self._host_out, self._host_ist
        r, status = o[11], ist[3]
        self.done = o[12] != 0.0
        self.CONVERGENCE_FAILURE = status == 2
        self._reward = int(round(r)) if (self.DISCRETE_REWARD and status != 2) else r   # :301 / :172
        self._reward_stats['step'].append(ist[1])
        self._reward_stats['reward'].append(self._reward)
        self._step_time = time.time() - t0                           # :193
        self._time_stats['step'].append(ist[1])
        self._time_stats['step_time'].append(self._step_time)
        return np.array(o[:11]), self._reward, self.done, {}

    def _physical(self):
        """Current algebraic outputs in physical units (host, fp64) for render()."""
        v = self._vec
        if v._traj is not None and v._traj_n > 0:
            t = v.trajectory(0)
            return {k: val[-1] for k, val in t.items()}
        y = v.fstate[:, 0].to(torch.float64).cpu().numpy()
        ksub = int(v.istate[0, 0])
        row = 0 if (v.ev_count == 0 or ksub < v.ev_s0) else min((ksub - v.ev_s0) // max(v.ev_ds, 1) + 1, v.ev_count)
        d = der_model.physical_outputs(v.par, y[:v.ns], np.float64(float(v.ev_vg[row, 0])), y[v.ns], y[v.ns + 1])
        d["Ppv"] = float(v.out[7, 0]) * der_model.SBASE
        return d

    def render(self, mode='vector', show=False):                     # :336-364
        """``'vector'``: print the reference's observation line in physical units (:339-353);
        ``'human'``: additionally draw the six figures of the reference's SimulationResults calls
        (:358-364: active power, reactive power, DC-link voltage, duty cycle, LV voltages, currents)
        from the recorded trajectory (env-step resolution) and return them as matplotlib figures."""
        if mode not in self.metadata['render.modes']:
            raise ValueError("render mode %r not in %r" % (mode, self.metadata['render.modes']))
        sim = self.sim
        p = self._physical()
        obs = {'ma': p['ma'], 'ia': p['ia'], 'Vdc': p['Vdc'], 'vta': p['vta'], 'va': p['va'], 'Ppv': p['Ppv'],
               'S_inverter': p['S_inverter'], 'S_PCC': p['S_PCC'], 'Q_ref': p['Q_ref'], 'Vdc_ref': p['Vdc_ref'],
               'tStart': sim.tStart}
        line = " ".join('{}:{:.2f},'.format(k, complex(v) if np.iscomplexobj(v) else float(v)) for k, v in obs.items())
        print(line)
        print('Reward:{:.5f}'.format(self._reward))
        print('Time for step:{:.3f}'.format(self._step_time))        # env_utilities.py:54-57
        if mode != 'human':
            return line
        if self._vec._traj is None:
            raise RuntimeError("render('human') needs the recorded trajectory: PVDER(record_trajectory=True)")
        import matplotlib
        if not show:
            matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        t = self._vec.trajectory(0)
        ph = "abc"[:self._vec.phases]
        plots = [
            ('active_power_Ppv_Pac_PCC', 'Active power [W]', [('Ppv', t['Ppv']), ('Pac_PCC', t['S_PCC'].real)]),
            ('reactive_power_Q_PCC', 'Reactive power [VAR]', [('Q_PCC', t['S_PCC'].imag), ('Q_ref', t['Q_ref'])]),
            ('voltage_Vdc', 'DC link voltage [V]', [('Vdc', t['Vdc']), ('Vdc_ref', t['Vdc_ref'])]),
            ('duty_cycle', 'Duty cycle |m|', [('m' + q, np.abs(t['m' + q])) for q in ph]),
            ('voltage_LV', 'LV voltage |v| [V]', [('v' + q, np.abs(t['v' + q])) for q in ph]),
            ('current', 'Current |i| [A]', [('i' + q, np.abs(t['i' + q])) for q in ph]),
        ]
        figs = []
        for name, ylabel, series in plots:
            fig, ax = plt.subplots(figsize=(6, 3))
            for label, y in series:
                ax.plot(t['t'], y, label=label)
            ax.set_xlabel('Time [s]'); ax.set_ylabel(ylabel); ax.set_title(name); ax.legend(); ax.grid(True)
            figs.append(fig)
        if show:
            plt.show()
        return figs

    def update_env_events(self, event_spec_list):                    # :413-438
        self._vec.update_env_events(event_spec_list)
        self.env_events_spec = self._vec.env_events_spec

    def seed(self, seed=None):
        self._vec.seed = int(seed or 0)
        return [self._vec.seed]

    def close(self):
        self._vec.close()

    def show_step_time(self):
        print('Time for step:{:.3f}'.format(self._step_time))

    # -- the rest of the reference's public helper surface: statistics of the running episode
    #    (env_utilities.py:12-101) and the configuration / evaluation printers (PVDER_env.py:445-529) -----------
    pp = property(lambda self: pprint.PrettyPrinter(indent=4))       # env_utilities.py:10 (bound to the stdout of the call)

    def initialize_stats(self):                                      # env_utilities.py:12-23
        self._action_stats = {i: 0 for i in range(5)}
        self._reward_stats = {'step': [], 'reward': []}
        self._time_stats = {'step': [], 'step_time': []}

    def update_action_stats(self, action):                           # :25-30
        self._action_stats[action] = self._action_stats[action] + 1

    def update_reward_stats(self):                                   # :32-38
        self._reward_stats['step'].append(self.steps)
        self._reward_stats['reward'].append(self._reward)

    def update_time_stats(self):                                     # :40-46
        self._time_stats['step'].append(self.steps)
        self._time_stats['step_time'].append(self._step_time)

    def _plot(self, x, y, color, xlabel, ylabel, title, bar=False, ticks=None):
        import matplotlib.pyplot as plt
        if bar:
            plt.bar(x, y, color=color)
            plt.xticks(x, ticks)
        else:
            plt.plot(x, y, color=color)
        plt.xlabel(xlabel); plt.ylabel(ylabel); plt.title(title)
        plt.show()

    def show_action_stats(self, SHOW_PLOT=False):                    # :59-75
        self.pp.pprint(self._action_stats)
        if SHOW_PLOT:
            keys = list(self._action_stats.keys())
            self._plot(list(range(len(keys))), list(self._action_stats.values()), 'green', "Actions", "Count",
                       "Action totals after every episode.", bar=True, ticks=keys)

    def show_reward_stats(self, SHOW_PLOT=False):                    # :77-88
        if SHOW_PLOT:
            self._plot(self._reward_stats['step'], self._reward_stats['reward'], 'red', "Steps", "Reward", "Reward per step")
        else:
            self.pp.pprint(self._reward_stats)

    def show_time_stats(self, SHOW_PLOT=False):                      # :90-101
        if SHOW_PLOT:
            self._plot(self._time_stats['step'], self._time_stats['step_time'], 'blue', "Steps", "Time (s)", "Time per step")
        else:
            self.pp.pprint(self._time_stats)

    def update_env_goal(self, goal_type=None, goal_spec=None):       # PVDER_env.py:445-456
        if goal_type is None and goal_spec is None:
            for g in self._vec.env_goal_spec:
                self._vec.env_goal_spec[g]['reward']['my_spec'] = self._vec.env_goal_spec[g]['reward']['required']
                self._vec.env_goal_spec[g]['action']['my_spec'] = self._vec.env_goal_spec[g]['action']['required']
            self._vec._drop_plan()
        else:
            assert goal_type in self._vec.env_goal_spec, '{} is not a valid goal!'.format(goal_type)
            print('Update function is under construction!')             # :456

    def show_env_config(self):                                       # :499-504
        print('Environment name:{}'.format(self.name))
        self.show_goal_config()
        self.show_sim_config()

    def show_sim_config(self):                                       # :506-510
        v = self._vec
        print('time step:{:.4f},del Qref:{:.2f},del Vdcref:{:.2f}'.format(v._sim_time_per_env_step, v._delQref, v._delVdcref))
        print('Voltage events:{},Insolation events:{}'.format(self.env_events_spec['voltage']['ENABLE'],
                                                              self.env_events_spec['insolation']['ENABLE']))

    def show_goal_config(self):                                      # :512-524
        goal_type = self.goals_list[0]
        spec = self._vec.env_goal_spec[goal_type]
        print('Goal type:', goal_type)
        print('Reward type:{},Discrete:{}'.format(spec['reward'].get('my_spec', spec['reward']['required']), self.DISCRETE_REWARD))
        if goal_type == 'Q_regulation':
            print('Q reference:{}'.format(self._vec.env_reward_spec['reference_values']['Q_ref']))
        elif goal_type == 'power_regulation':
            print('P reference:{}'.format(self._vec.env_reward_spec['reference_values']['P_ref']))
        print('Action type:', spec['action'].get('my_spec', spec['action']['required']))

    def calc_returns(self, episodes=2):                              # :458-497
        """Average return per goal and fixed action policy (``env_average_return``): the reference loops
        3 goals x 4 policies x 2 episodes of its one env; here each (goal, policy) cell is one fused
        open-loop rollout of ``episodes`` envs on the device (gym_PVDER_amd/evaluate.py), with this env's
        step size, episode length, reward type, integrator and event spec."""
        from ..evaluate import calc_returns as batched
        v = self._vec
        model_type = [k for k, spec in v.env_model_spec.items() if spec['derId'] == v.der_id][0]
        kw = dict(n_sim_time_steps_per_env_step=v.n_sim_time_steps_per_env_step, max_sim_time=self.max_sim_time_user,
                  DISCRETE_REWARD=v.DISCRETE_REWARD, integrator=v.integrator, events_spec=v.env_events_spec,
                  model_type=model_type, der_config=v.der_config, micro_steps=v.micro_steps, micro_coarse=v.micro_coarse,
                  off_range=v.off_range)                 # the env itself is what the reference evaluates (:458-497)
        for goal in v.env_goal_spec:
            print('Calculating average return for "{}" goal with "random", "inc", "dec", "no_change" actions.'.format(goal))
        self.env_average_return = batched(episodes=episodes, device=v.device, seed=v.seed, **kw)
        self.show_env_returns()
        return self.env_average_return

    def show_env_returns(self):                                      # :526-529
        self.pp.pprint(self.env_average_return)
