"""Batched counterpart of ``PVDER.calc_returns`` (/root/reference/gym_PVDER/envs/PVDER_env.py:458-497),
the reference's only evaluation driver: average undiscounted return per goal and per fixed action
policy.  The reference runs 3 goals x 4 policies x 2 episodes one env step at a time; here every
(goal, policy) cell is ONE batch of ``episodes`` envs stepped in lockstep on the device.

Policy names are the reference's, including its labels: 'inc' sends action 0 (a no-op in
``action_calc`` :211), 'dec' sends action 1 (+Q_ref) and 'no_change' sends action 2 (-Q_ref)
(:482-487 vs :211-221) -- kept as is so the returned table is comparable.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from .vec_env import PVDERVec

ACTION_SPECS = {'random': None, 'inc': 0, 'dec': 1, 'no_change': 2}       # PVDER_env.py:462, :480-487


def calc_returns(episodes: int = 2, device="cuda", goals: Optional[Sequence[str]] = None, seed: int = 0,
                 random_actions: Optional[torch.Tensor] = None,
                 **env_kwargs) -> Dict[str, Dict[str, Dict[str, object]]]:
    """{goal: {action_spec: {'return': mean episode return, 'ref': [Vdc_ref (V), Q_ref (VAR)]}}}

    Every (goal, policy) cell is ONE open-loop rollout (``PVDERVec.rollout``: the whole episode of
    ``episodes`` envs enqueued by one host call; per-step output blocks kept on the device, the
    return summed there over the steps before each env's ``done``).  ``random_actions`` ([K, episodes]
    integers) replaces the device-drawn actions of the 'random' policy (parity runs share them with
    the oracle)."""
    goals = list(goals) if goals is not None else list(PVDERVec.env_goal_spec.keys())      # :461
    out: Dict[str, Dict[str, Dict[str, object]]] = {}
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    from . import der_model
    for goal in goals:
        out[goal] = {}
        for name, fixed in ACTION_SPECS.items():
            env = PVDERVec(episodes, goals_list=[goal], device=device, auto_reset=False, seed=seed, **env_kwargs)
            env.reset()
            K = int(-(-env.max_sim_time * 60.0 // env.n_sim_time_steps_per_env_step)) + 1     # steps to `done`, + 1
            K = min(K, env.max_episode_steps + 1)
            if fixed is not None:
                acts = torch.full((K, episodes), fixed, dtype=torch.uint8, device=env.device)
            elif random_actions is not None:
                acts = torch.as_tensor(random_actions, device=env.device)[:K].to(torch.uint8).contiguous()
                K = int(acts.shape[0])
            else:
                acts = torch.randint(0, 5, (K, episodes), generator=gen, device=env.device, dtype=torch.uint8)
            blocks = torch.empty((K, 13, episodes), dtype=env.out.dtype, device=env.device)
            env.rollout(acts, out=blocks)
            done = blocks[:, 12] != 0                                            # [K, episodes]
            before = torch.cumsum(done.to(torch.int32), dim=0) - done.to(torch.int32) == 0   # steps up to and incl. the done one
            total = (blocks[:, 11].to(torch.float64) * before).sum(dim=0)
            out[goal][name] = {'return': float(total.mean()),                                 # :493
                               'ref': [float(env.Vdc_ref.double().mean()) * der_model.VBASE,
                                       float(env.Q_ref.double().mean()) * der_model.SBASE]}    # :494-495
    return out
