"""Domain randomisation over PARAMETER SETS ("optional per-env override").

The reference runs ONE config block (derId "50", /root/reference/config_der.json:2-21) for every env.  The
default integrator of this build folds the parameters into host-computed tables -- the steady state the
episodes start from, the Jacobian inside the matrix exponential, the coefficient sets -- so a parameter set
belongs to a launch, not to a lane: a batch is spread over G parameter sets as G ``PVDERVec`` shards, each
with its own plan and tables, stepped concurrently on G HIP streams (a shard of N/G envs leaves SIMDs idle;
the shards' kernels fill them side by side).  Every shard is the ordinary step path, parity-tested as such.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .vec_env import PVDERVec


class PVDERVecGroups:
    """``G`` parameter sets x ``envs_per_group[g]`` envs behind the batched step/reset API.

    ``der_configs[g]``: overrides of the config block for group g (``{}`` = the reference's values).
    Global env ids are contiguous over the groups (group g owns ids offset_g .. offset_g + n_g - 1; the
    event generator is keyed by them, as in the sharded multi-GPU env).  ``step`` returns concatenated
    ``(obs[N, 11], reward[N], done[N], info)`` tensors (copies: each shard keeps its own output block)."""

    def __init__(self, envs_per_group: Sequence[int], der_configs: Sequence[Optional[Dict[str, float]]],
                 device="cuda", env_id_base: int = 0, **kwargs):
        if len(envs_per_group) != len(der_configs) or not envs_per_group:
            raise ValueError("need one der_config per group")
        self.device = torch.device(device)
        self.groups: List[PVDERVec] = []
        off = int(env_id_base)
        for n, cfg in zip(envs_per_group, der_configs):
            self.groups.append(PVDERVec(int(n), device=device, env_id_base=off, der_config=cfg or None, **kwargs))
            off += int(n)
        self.num_envs = off - int(env_id_base)
        self.device = self.groups[0].device
        self._streams = [torch.cuda.Stream(self.device) for _ in self.groups]
        self._bounds = [0]
        for g in self.groups:
            self._bounds.append(self._bounds[-1] + g.num_envs)
        from .spaces import make_spaces
        self.action_space, self.observation_space = make_spaces()          # per env, as PVDER_env.py:50-51

    def _fan_out(self, fn):
        """Run ``fn(g, env)`` for every shard on its own stream, ordered after the caller's stream; the
        caller's stream then waits for all of them."""
        cur = torch.cuda.current_stream(self.device)
        outs = []
        for k, (env, st) in enumerate(zip(self.groups, self._streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn(k, env))
        for st in self._streams:
            cur.wait_stream(st)
        return outs

    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        b = self._bounds
        outs = self._fan_out(lambda k, env: env.reset(None if mask is None else mask[b[k]:b[k + 1]]))
        return torch.cat(outs, dim=0)

    def step(self, actions: torch.Tensor):
        if actions.shape != (self.num_envs,):
            raise ValueError("actions must have shape (%d,)" % self.num_envs)
        actions = actions.to(device=self.device, dtype=torch.uint8)
        b = self._bounds
        outs = self._fan_out(lambda k, env: env.step(actions[b[k]:b[k + 1]].contiguous()))
        return (torch.cat([o[0] for o in outs], dim=0), torch.cat([o[1] for o in outs], dim=0),
                torch.cat([o[2] for o in outs], dim=0), {})

    def step_stats(self) -> Dict[str, float]:
        tot: Dict[str, float] = {}
        for env in self.groups:
            for k, v in env.step_stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def close(self):
        for env in self.groups:
            env.close()
