"""Batched DQN over a PVDERVec (second half): the loop of the reference's TF-Agents demo,
/root/reference/examples/gym_PVDER_environment_tf_agents_DQN_demo.ipynb cells 12-28, with the env AND the acting policy on
the device.

The demo's pieces and their counterparts here:

* ``q_network.QNetwork(obs(11), actions(5), fc_layer_params=(100,))`` (cell 17) -> :class:`QNetwork`: two ``torch.nn.Linear``
  layers whose weights live in ONE flat device buffer in the layout the HIP kernels read (include/pvder_agent.h,
  PVDER_POLICY_QNET);
* the collect policy (epsilon-greedy on the network, cell 19) and ``DynamicStepDriver`` (cell 25) -> the Q-network
  evaluated INSIDE the agent launch (``pvder_qagent_step_f32``; one launch per env step next to the step launch) or inside
  the fused closed-loop rollout (``pvder_plan_rollout_fused_agent``: a whole evaluation episode in a handful of launches) --
  observations never leave the GPU, the host never synchronises inside an episode;
* ``TFUniformReplayBuffer`` (cell 23) -> a device ring buffer of (obs, action, reward, next obs, done) rows filled by
  device-to-device copies;
* ``DqnAgent.train`` with ``element_wise_squared_loss`` and Adam(1e-3) (cells 17, 28) -> a torch training step on sampled
  rows (torch owns the optimiser: the consumer side of the env, as TensorFlow is in the notebook);
* ``compute_average_return`` (cell 21) -> :meth:`BatchedDQNAgent.evaluate`.

N envs collect in lockstep: ``collect_steps_per_iteration = 1`` of the notebook is one batched step = N transitions.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from .. import _native
from ..vec_env import PVDERVec

N_OBS, N_ACT = 11, 5


class QNetwork(torch.nn.Module):
    """observation(11) -> Linear(hidden) -> relu -> Linear(5): tf_agents' QNetwork with fc_layer_params = (hidden,) (cell 17).
    ``flat`` is the [hidden*11 + hidden + 5*hidden + 5] fp32 buffer the device policy reads; the two Linear layers are VIEWS
    of it, so an optimiser step on the module is seen by the next launch without a copy."""

    def __init__(self, hidden: int = 100, device="cuda", seed: int = 0):
        super().__init__()
        if not 1 <= hidden <= 128:
            raise ValueError("hidden must be in 1..128 (the device policy keeps the network in LDS)")
        self.hidden = int(hidden)
        H = self.hidden
        self.flat = torch.nn.Parameter(torch.zeros(H * N_OBS + H + N_ACT * H + N_ACT, dtype=torch.float32, device=device))
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        with torch.no_grad():
            # (tf_agents initialises the hidden layer with variance scaling and the output layer uniformly in +-0.03)
            w1 = torch.randn(H, N_OBS, generator=g) * (2.0 / N_OBS) ** 0.5 / 1.5
            w2 = (torch.rand(N_ACT, H, generator=g) * 2 - 1) * 0.03
            self.flat[:H * N_OBS].copy_(w1.reshape(-1))
            self.flat[H * N_OBS + H: H * N_OBS + H + N_ACT * H].copy_(w2.reshape(-1))
            self.flat[-N_ACT:].fill_(-0.2)

    def _views(self):
        H, f = self.hidden, self.flat
        o = H * N_OBS
        return f[:o].view(H, N_OBS), f[o:o + H], f[o + H:o + H + N_ACT * H].view(N_ACT, H), f[o + H + N_ACT * H:]

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        w1, b1, w2, b2 = self._views()
        return torch.relu(obs @ w1.t() + b1) @ w2.t() + b2


class BatchedDQNAgent:
    train_spec = {'learning_rate': 1e-3, 'batch_size': 64, 'replay_buffer_capacity': 100000, 'gamma': 1.0,
                  'epsilon_greedy': 0.1, 'target_update_period': 1, 'fc_layer_params': (100,)}   # notebook cell 12; DqnAgent defaults

    def __init__(self, env: PVDERVec, seed: int = 0, **spec):
        if not isinstance(env, PVDERVec):
            raise TypeError("env must be a PVDERVec")
        if env.fdtype != torch.float32:
            raise ValueError("the agent launch reads the fp32 output block")
        self.env = env
        self.spec = dict(self.train_spec, **spec)
        dev = env.device
        self.q_net = QNetwork(self.spec['fc_layer_params'][0], device=dev, seed=seed)
        self.target = QNetwork(self.spec['fc_layer_params'][0], device=dev, seed=seed)
        self.target.load_state_dict(self.q_net.state_dict())
        self.optimizer = torch.optim.Adam(self.q_net.parameters(), lr=self.spec['learning_rate'])
        cap = int(self.spec['replay_buffer_capacity'])
        self.capacity = max(cap // env.num_envs, 1) * env.num_envs          # whole batched steps
        self.buf_obs = torch.zeros((self.capacity, N_OBS), dtype=torch.float32, device=dev)
        self.buf_next = torch.zeros((self.capacity, N_OBS), dtype=torch.float32, device=dev)
        self.buf_act = torch.zeros(self.capacity, dtype=torch.int64, device=dev)
        self.buf_rew = torch.zeros(self.capacity, dtype=torch.float32, device=dev)
        self.buf_done = torch.zeros(self.capacity, dtype=torch.float32, device=dev)
        self.buf_n, self.buf_pos = 0, 0
        self.seed, self._counter, self.train_step_counter = int(seed), 0, 0
        N = env.num_envs
        self._state = torch.zeros(N, dtype=torch.int64, device=dev)          # (unused by the Q-network policy)
        self.action = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.alive = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.ep_reward = torch.zeros(N, dtype=torch.float64, device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed + 1)

    # -- the policy on the device ------------------------------------------------------------------------------------
    def _cfg(self, first: bool, epsilon: float, counter: int) -> "_native.QAgentCfg":
        c = _native.QAgentCfg()
        c.policy, c.hidden = 1, self.q_net.hidden                       # PVDER_POLICY_QNET
        c.bins, c.n_act, c.first, c.choose = 1, N_ACT, int(first), 1
        c.lr, c.gamma, c.q_done, c.low, c.win = 0.0, 0.0, 0.0, 0.0, 1.0
        c.epsilon = float(epsilon)
        c.seed, c.counter, c.env_id_base = self.seed & (2 ** 64 - 1), counter, self.env.env_id_base
        return c

    def policy_step(self, first: bool, epsilon: float):
        """The Q-network on ``env.out`` -> ``self.action`` (epsilon-greedy), one launch (pvder_qagent_step_f32)."""
        env = self.env
        cfg = self._cfg(first, epsilon, self._counter)
        self._counter += 1
        rc = env._lib.pvder_qagent_step_f32(C.byref(cfg), self.q_net.flat.data_ptr(), env.out.data_ptr(), self._state.data_ptr(),
                                            self.action.data_ptr(), self.alive.data_ptr(), self.ep_reward.data_ptr(),
                                            env.num_envs, env._stream())
        _native.check(rc, "pvder_qagent_step_f32")

    def greedy_actions(self, obs: torch.Tensor) -> torch.Tensor:
        """torch reference of the device policy (tests): argmax of the network, first maximum."""
        with torch.no_grad():
            return self.q_net(obs.to(torch.float32)).argmax(dim=1).to(torch.uint8)

    # -- collect (DynamicStepDriver, cell 25) --------------------------------------------------------------------------
    def collect(self, num_steps: int, epsilon: Optional[float] = None):
        """``num_steps`` batched env steps under the collect policy into the replay buffer (an env that finishes restarts
        in the same launch: the PVDERVec must auto-reset, as TFPyEnvironment does)."""
        env = self.env
        assert env.auto_reset, "construct the PVDERVec with auto_reset=True: collection runs across episode ends"
        eps = self.spec['epsilon_greedy'] if epsilon is None else epsilon
        N = env.num_envs
        if env._needs_reset:
            env.reset()
            self.policy_step(True, eps)
        for _ in range(num_steps):
            p = self.buf_pos
            self.buf_obs[p:p + N].copy_(env.out[:N_OBS].t())
            self.buf_act[p:p + N].copy_(self.action)
            env.step(self.action, return_tensors=False)
            self.buf_rew[p:p + N].copy_(env.out[11])
            self.buf_done[p:p + N].copy_(env.out[12])
            # (an env that ended restarts in the same launch: its observation rows hold the RESET observation; the bootstrap
            #  target of that transition is masked by `done`, so the next-observation row is never used for it)
            self.buf_next[p:p + N].copy_(env.out[:N_OBS].t())
            self.policy_step(False, eps)
            self.buf_pos = (p + N) % self.capacity
            self.buf_n = min(self.buf_n + N, self.capacity)

    # -- train (DqnAgent.train, cell 28) -------------------------------------------------------------------------------
    def train_step(self) -> float:
        B = int(self.spec['batch_size'])
        idx = torch.randint(0, self.buf_n, (B,), generator=self.gen, device=self.env.device)
        obs, act, rew = self.buf_obs[idx], self.buf_act[idx], self.buf_rew[idx]
        nxt, done = self.buf_next[idx], self.buf_done[idx]
        with torch.no_grad():
            target = rew + self.spec['gamma'] * (1.0 - done) * self.target(nxt).max(dim=1).values
        q = self.q_net(obs).gather(1, act[:, None])[:, 0]
        loss = ((q - target) ** 2).mean()                            # element_wise_squared_loss
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.train_step_counter += 1
        if self.train_step_counter % int(self.spec['target_update_period']) == 0:
            with torch.no_grad():
                self.target.flat.copy_(self.q_net.flat)
        return float(loss.detach())

    def train(self, num_iterations: int, initial_collect_steps: int = 8, collect_steps_per_iteration: int = 1,
              log_interval: int = 0) -> List[float]:
        """The loop of cell 28: collect, sample, train."""
        self.collect(initial_collect_steps, epsilon=1.0)             # (the notebook fills the buffer with its collect policy too)
        losses = []
        for it in range(num_iterations):
            self.collect(collect_steps_per_iteration)
            losses.append(self.train_step())
            if log_interval and (it + 1) % log_interval == 0:
                print('step = {0}: loss = {1}'.format(it + 1, losses[-1]))
        return losses

    # -- evaluation (compute_average_return, cell 21) ------------------------------------------------------------------
    def evaluate(self, eval_env: PVDERVec, fused: bool = True) -> Dict[str, float]:
        """One greedy episode of every env of ``eval_env`` (auto_reset=False); ``fused``: the whole episode inside the fused
        closed-loop rollout (pvder_plan_rollout_fused_agent with the Q-network as the action source)."""
        assert not eval_env.auto_reset, "the evaluation env must not auto-reset"
        N, dev = eval_env.num_envs, eval_env.device
        state = torch.zeros(N, dtype=torch.int64, device=dev)
        action = torch.zeros(N, dtype=torch.uint8, device=dev)
        alive = torch.zeros(N, dtype=torch.uint8, device=dev)
        ret = torch.zeros(N, dtype=torch.float64, device=dev)
        lib = eval_env._lib
        c = eval_env._cfg()
        steps = -(-c.max_ksub // c.n_sub)
        if c.max_steps > 0:
            steps = min(steps, c.max_steps)

        def launch(first, counter):
            cfg = self._cfg(first, 0.0, counter)
            cfg.env_id_base = eval_env.env_id_base
            _native.check(lib.pvder_qagent_step_f32(C.byref(cfg), self.q_net.flat.data_ptr(), eval_env.out.data_ptr(), state.data_ptr(),
                                                    action.data_ptr(), alive.data_ptr(), ret.data_ptr(), N, eval_env._stream()),
                          "pvder_qagent_step_f32")
        eval_env.reset()
        launch(True, 0)
        if fused and eval_env.integrator == "etdlin" and eval_env.ns == 23:
            if eval_env._plan is None:
                eval_env._make_plan()
            cfg = self._cfg(False, 0.0, 1)
            cfg.env_id_base = eval_env.env_id_base
            with eval_env._device_ctx():
                rc = lib.pvder_plan_rollout_fused_agent(eval_env._plan, eval_env.fstate.data_ptr(), eval_env.istate.data_ptr(),
                                                        eval_env._ev_vg.data_ptr(), eval_env._ev_si.data_ptr(), int(steps),
                                                        eval_env.out.data_ptr(), eval_env.stats.data_ptr(), eval_env._progress.data_ptr(),
                                                        C.byref(cfg), self.q_net.flat.data_ptr(), state.data_ptr(), action.data_ptr(),
                                                        alive.data_ptr(), ret.data_ptr(), eval_env._stream())
            _native.check(rc, "pvder_plan_rollout_fused_agent")
        else:
            for k in range(steps):
                eval_env.step(action, return_tensors=False)
                launch(False, 1 + k)
        return {"average_return": float(ret.mean()), "min": float(ret.min()), "max": float(ret.max()), "steps": steps,
                "returns": ret}
