"""Device-resident tabular Q-learning over a batched PVDER-v0.

Batched counterpart of the reference's ``QAgent`` (/root/reference/gym_PVDER/agents/qagent.py:10-135):
same discretisation (6 bins per observation dimension over the Box(-10, 10) range, :35-45), same
constant -2 initial table (:37), epsilon-greedy action choice (:63-68), the same update
``Q <- (1-lr) Q + lr (r + gamma max_a' Q')`` for running envs (:88-95) and ``Q <- 0`` for the
transition that ends an episode (:100-103), epsilon decayed once per episode
(agent_utilities.py:79-84) -- but N environments learn into ONE table in lockstep, and nothing
leaves the GPU: observations, actions, rewards and the table are device tensors, the env step is the
fused HIP launch of PVDERVec.  (6^11 x 5 fp32 entries = 7.3 GB: fits a 288 GB MI355X many times.)
When several envs hit the same (state, action) in one step, one of their updates wins.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from .. import _native
from ..vec_env import PVDERVec


class BatchedQAgent:
    train_spec = {'LEARNING_RATE': 0.1, 'DISCOUNT': 0.95, 'EPISODES': 200, 'STATS_EVERY': 5}     # qagent.py:14-18
    explore_spec = {'EPSILON_DECAY': 0.98, 'MIN_EPSILON': 0.001}                               # qagent.py:24-25

    def __init__(self, env: PVDERVec, bins: int = 6, seed: int = 0):
        if not isinstance(env, PVDERVec):
            raise TypeError("env must be a PVDERVec")
        self.env = env
        self.bins = int(bins)
        self.n_obs, self.n_act = 11, 5
        dev = env.device
        self.low, self.high = -10.0, 10.0                          # observation_space (PVDER_env.py:51)
        self.win = (self.high - self.low) / self.bins              # discrete_os_win_size (qagent.py:38)
        n_states = self.bins ** self.n_obs
        self.q_table = torch.full((n_states, self.n_act), -2.0, dtype=torch.float32, device=dev)   # :37
        self._strides = torch.tensor([self.bins ** (self.n_obs - 1 - k) for k in range(self.n_obs)],
                                     dtype=torch.int64, device=dev)
        self.epsilon = 1.0                                         # qagent.py:27
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.seed = int(seed)
        self.aggregate_episode_rewards: Dict[str, List[float]] = {'episode': [], 'average': [], 'max': [], 'min': []}

    def get_discrete_state(self, obs: torch.Tensor) -> torch.Tensor:
        """[N, 11] observations -> flat table row index [N] (qagent.py:43-45)."""
        # in float64 like the reference (its observations and window sizes are float64 arrays): Q_PCC = 0 sits
        # exactly on a bin edge of the 6-bin grid, where fp32 and fp64 arithmetic round to different bins
        d = ((obs.to(torch.float64) - self.low) / self.win).to(torch.int64).clamp_(0, self.bins - 1)
        return (d * self._strides).sum(dim=1)

    def agent_action(self, obs: torch.Tensor) -> torch.Tensor:
        """Greedy actions [N] uint8 (qagent.py:130-135)."""
        return self.q_table[self.get_discrete_state(obs)].argmax(dim=1).to(torch.uint8)

    def update_epsilon(self):                                       # agent_utilities.py:79-84
        if self.epsilon > self.explore_spec['MIN_EPSILON']:
            self.epsilon = max(self.explore_spec['MIN_EPSILON'], self.epsilon * self.explore_spec['EPSILON_DECAY'])

    def update(self, s: torch.Tensor, a: torch.Tensor, r: torch.Tensor, s2: torch.Tensor, done: torch.Tensor,
               alive: torch.Tensor):
        """One Q-learning update per running env (qagent.py:83-103): table rows ``s`` / ``s2`` (flat
        indices), actions ``a``, rewards ``r`` (fp32), ``done`` of this transition, ``alive`` = the env
        was still running before it (finished envs return stale tuples and must not learn from them)."""
        lr, gamma = self.train_spec['LEARNING_RATE'], self.train_spec['DISCOUNT']
        cur = self.q_table[s, a]
        new_q = (1.0 - lr) * cur + lr * (r + gamma * self.q_table[s2].max(dim=1).values)          # :88-95
        new_q = torch.where(done, torch.zeros_like(new_q), new_q)                                   # :100-103
        self.q_table.index_put_((s[alive], a[alive]), new_q[alive])

    # -- the reference's agent utilities (agents/agent_utilities.py:15-84), batched ---------------------------
    def initialize_aggregate_rewards(self):                          # :15-27 (keys as in train()'s statistics)
        self.aggregate_episode_rewards = {'episode': [], 'average': [], 'max': [], 'min': []}

    def collect_aggregate_rewards(self, episode, **reward_stats):    # :29-40
        print('Storing rewards @ Episode:{},Steps:{}'.format(episode, int(self.env.istate[1].max())))
        self.aggregate_episode_rewards['episode'].append(episode)
        for stat_type, value in reward_stats.items():
            if stat_type in self.aggregate_episode_rewards:
                self.aggregate_episode_rewards[stat_type].append(value)
                print(stat_type, ':', value)

    def show_plots(self, show: bool = True):                         # :50-63
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots()
        for stat_type in ('average', 'max', 'min'):
            ax.plot(self.aggregate_episode_rewards['episode'], self.aggregate_episode_rewards[stat_type], label=stat_type)
        ax.legend(loc=4)
        if show:
            plt.show()
        return fig

    def evaluate(self, n_episodes: int = 1) -> List[float]:          # :65-77
        """Greedy episodes of all N envs (``agent_action`` on every step); returns and prints the mean episode
        return of each (the reference prints every step of its one env and renders it)."""
        env = self.env
        assert not env.auto_reset, "construct the PVDERVec with auto_reset=False: episodes are reset here"
        means = []
        for i in range(n_episodes):
            obs = env.reset()
            ret = torch.zeros(env.num_envs, dtype=torch.float64, device=env.device)
            alive = torch.ones(env.num_envs, dtype=torch.bool, device=env.device)
            for _ in range(env.max_episode_steps + 1):
                obs, r, done, _ = env.step(self.agent_action(obs))
                ret += torch.where(alive, r.to(torch.float64), torch.zeros_like(ret))
                alive = alive & ~done
                if not bool(alive.any()):
                    break
            means.append(float(ret.mean()))
            print('Episode:{},Steps:{},mean return:{:.3f},min:{:.1f},max:{:.1f}'.format(
                i, int(env.istate[1].max()), means[-1], float(ret.min()), float(ret.max())))
        return means

    # -- the same loop with the agent side as ONE HIP launch per env step (include/pvder_agent.h) ------------
    def _agent_cfg(self, first: bool, choose: bool, counter: int) -> "_native.QAgentCfg":
        c = _native.QAgentCfg()
        c.bins, c.n_act, c.first, c.choose = self.bins, self.n_act, int(first), int(choose)
        c.lr, c.gamma = self.train_spec['LEARNING_RATE'], self.train_spec['DISCOUNT']
        c.epsilon, c.q_done = self.epsilon, 0.0
        c.low, c.win = self.low, self.win
        c.seed, c.counter, c.env_id_base = self.seed & (2 ** 64 - 1), counter, self.env.env_id_base
        return c

    def fused_step(self, state: torch.Tensor, action: torch.Tensor, alive: torch.Tensor, ep_reward: torch.Tensor,
                   first: bool, choose: bool = True, counter: int = 0):
        """Discretise ``env.out``, learn from the transition that led to it, choose the next action -- one
        launch (pvder_qagent_step_f32) on torch's current stream; all arguments are device tensors updated in
        place (``state`` int64 [N], ``action`` uint8 [N], ``alive`` uint8 [N], ``ep_reward`` float64 [N])."""
        env = self.env
        cfg = self._agent_cfg(first, choose, counter)
        rc = env._lib.pvder_qagent_step_f32(C.byref(cfg), self.q_table.data_ptr(), env.out.data_ptr(), state.data_ptr(),
                                            action.data_ptr(), alive.data_ptr(), ep_reward.data_ptr(), env.num_envs,
                                            env._stream())
        _native.check(rc, "pvder_qagent_step_f32")

    def closed_loop_episode(self, state, action, alive, ep_reward, steps: int, counter: int):
        """``steps`` x (env step, agent step) in a handful of launches (pvder_plan_rollout_fused_agent, include/
        pvder_agent.h): the env state and the agent's per-env registers stay on the chip from step to step.  Same
        arguments as ``fused_step``; ``action`` holds the action chosen behind the reset."""
        env = self.env
        if env._plan is None:
            env._make_plan()
        if env._plan_integrator != "etdlin" or env.ns != 23 or env.auto_reset:
            raise ValueError("closed-loop fused episodes: integrator='etdlin', three-phase model, auto_reset=False")
        cfg = self._agent_cfg(False, True, counter)
        with env._device_ctx():
            rc = env._lib.pvder_plan_rollout_fused_agent(env._plan, env.fstate.data_ptr(), env.istate.data_ptr(),
                                                         env._ev_vg.data_ptr(), env._ev_si.data_ptr(), int(steps),
                                                         env.out.data_ptr(), env.stats.data_ptr(), env._progress.data_ptr(),
                                                         C.byref(cfg), self.q_table.data_ptr(), state.data_ptr(),
                                                         action.data_ptr(), alive.data_ptr(), ep_reward.data_ptr(), env._stream())
        _native.check(rc, "pvder_plan_rollout_fused_agent")

    def train_fused(self, episodes: Optional[int] = None, closed_loop: bool = False) -> Dict[str, List[float]]:
        """``train`` with TWO launches per env step (the env step + the agent step) and no host
        synchronisation inside an episode: the episode length is known (max_sim_time / TimeLimit), envs that
        end early simply stop learning (``alive``).  Exploration draws come from the counter generator of the
        event tables, keyed (seed, global env id, step count): runs are reproducible.
        ``closed_loop=True``: the whole episode -- every env step AND the agent step between two of them -- in a
        handful of launches (``closed_loop_episode``).  That form is ASYNCHRONOUS Q-learning across workgroups
        (Hogwild-style: workgroups run through their env steps independently inside a launch, table updates of other
        workgroups are seen late -- within the launch at the latest -- or lose a read-modify-write race; see
        include/pvder_agent.h): bit-identical to the two-launch loop for batches of at most one workgroup's envs in
        lockstep, statistically equivalent beyond (returns, visited entries)."""
        env = self.env
        assert not env.auto_reset, "construct the PVDERVec with auto_reset=False: episodes are reset here"
        assert env.fdtype == torch.float32, "the agent step reads the fp32 output block"
        N, dev = env.num_envs, env.device
        episodes = self.train_spec['EPISODES'] if episodes is None else episodes
        state = torch.zeros(N, dtype=torch.int64, device=dev)
        action = torch.zeros(N, dtype=torch.uint8, device=dev)
        alive = torch.zeros(N, dtype=torch.uint8, device=dev)
        ep_reward = torch.zeros(N, dtype=torch.float64, device=dev)
        c = env._cfg()                                             # episode length: max_sim_time, TimeLimit
        steps = -(-c.max_ksub // c.n_sub)
        if c.max_steps > 0:
            steps = min(steps, c.max_steps)
        recent: List[torch.Tensor] = []
        for episode in range(episodes):
            env.reset()
            self.fused_step(state, action, alive, ep_reward, first=True, counter=self._counter)
            self._counter += 1
            if closed_loop:
                self.closed_loop_episode(state, action, alive, ep_reward, steps, self._counter)
                self._counter += steps
            else:
                for _ in range(steps):
                    env.step(action, return_tensors=False)
                    self.fused_step(state, action, alive, ep_reward, first=False, counter=self._counter)
                    self._counter += 1
            self.update_epsilon()
            recent.append(ep_reward.clone())
            if episode % self.train_spec['STATS_EVERY'] == 0:                                           # :113-120
                block = torch.stack(recent[-self.train_spec['STATS_EVERY']:])
                self.aggregate_episode_rewards['episode'].append(episode)
                self.aggregate_episode_rewards['average'].append(float(block.mean()))
                self.aggregate_episode_rewards['max'].append(float(block.max()))
                self.aggregate_episode_rewards['min'].append(float(block.min()))
        return self.aggregate_episode_rewards

    _counter = 0

    def train(self, episodes: Optional[int] = None) -> Dict[str, List[float]]:
        """``episodes`` lockstep episodes of all N envs (the env must not auto-reset)."""
        env = self.env
        assert not env.auto_reset, "construct the PVDERVec with auto_reset=False: episodes are reset here"
        lr, gamma = self.train_spec['LEARNING_RATE'], self.train_spec['DISCOUNT']
        N = env.num_envs
        episodes = self.train_spec['EPISODES'] if episodes is None else episodes
        recent: List[torch.Tensor] = []
        for episode in range(episodes):
            s = self.get_discrete_state(env.reset())
            ep_reward = torch.zeros(N, dtype=torch.float64, device=env.device)
            alive = torch.ones(N, dtype=torch.bool, device=env.device)
            for _ in range(env.max_episode_steps + 1):
                greedy = self.q_table[s].argmax(dim=1)
                rand_a = torch.randint(0, self.n_act, (N,), generator=self.gen, device=env.device)
                explore = torch.rand(N, generator=self.gen, device=env.device) <= self.epsilon       # :63
                a = torch.where(explore, rand_a, greedy)
                obs, r, done, _ = env.step(a.to(torch.uint8))
                s2 = self.get_discrete_state(obs)
                r32 = r.to(torch.float32)
                ep_reward += torch.where(alive, r.to(torch.float64), torch.zeros_like(ep_reward))
                self.update(s, a, r32, s2, done, alive)
                alive = alive & ~done
                s = s2
                if not bool(alive.any()):
                    break
            self.update_epsilon()
            recent.append(ep_reward)
            if episode % self.train_spec['STATS_EVERY'] == 0:                                           # :113-120
                block = torch.stack(recent[-self.train_spec['STATS_EVERY']:])
                self.aggregate_episode_rewards['episode'].append(episode)
                self.aggregate_episode_rewards['average'].append(float(block.mean()))
                self.aggregate_episode_rewards['max'].append(float(block.max()))
                self.aggregate_episode_rewards['min'].append(float(block.min()))
        return self.aggregate_episode_rewards
