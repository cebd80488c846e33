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
                 **env_kwargs) -> Dict[str, Dict[str, Dict[str, object]]]:
    """{goal: {action_spec: {'return': mean episode return, 'ref': [Vdc_ref (V), Q_ref (VAR)]}}}"""
    goals = list(goals) if goals is not None else list(PVDERVec.env_goal_spec.keys())      # :461
    out: Dict[str, Dict[str, Dict[str, object]]] = {}
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for goal in goals:
        out[goal] = {}
        for name, fixed in ACTION_SPECS.items():
            env = PVDERVec(episodes, goals_list=[goal], device=device, auto_reset=False, seed=seed, **env_kwargs)
            env.reset()
            total = torch.zeros(episodes, dtype=torch.float64, device=env.device)
            alive = torch.ones(episodes, dtype=torch.bool, device=env.device)
            for _ in range(env.max_episode_steps + 1):
                if fixed is None:
                    a = torch.randint(0, 5, (episodes,), generator=gen, device=env.device, dtype=torch.uint8)
                else:
                    a = torch.full((episodes,), fixed, dtype=torch.uint8, device=env.device)
                _, r, d, _ = env.step(a)
                total += torch.where(alive, r.to(torch.float64), torch.zeros_like(total))
                alive &= ~d
                if not bool(alive.any()):
                    break
            from . import der_model
            out[goal][name] = {'return': float(total.mean()),                                 # :493
                               'ref': [float(env.Vdc_ref.double().mean()) * der_model.VBASE,
                                       float(env.Q_ref.double().mean()) * der_model.SBASE]}    # :494-495
    return out
