"""Multi-GPU PVDER-v0: independent envs sharded over ranks, ONE all-gather per step.

The reference has no distributed code at all (one env, one Python thread).
Envs are fully independent (no coupling term anywhere in ``PVDER.step``,
/root/reference/gym_PVDER/envs/PVDER_env.py:138-196), so the path shards trivially: rank r owns the
contiguous block of global env ids [r*N_local, (r+1)*N_local).  The on-device event generator is
keyed by the GLOBAL env id, so results do not depend on the number of ranks.  The only
communication is the collation of the per-step output block ([13, N_local]: 11 obs + reward +
done) with ``all_gather_into_tensor`` -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the
CPU tests.  No collective sits on the data path of the integration itself.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global env id, number of local envs) of ``rank``; requires world | total_envs."""
    if total_envs % world != 0:
        raise ValueError("total_envs (%d) must be divisible by the world size (%d)" % (total_envs, world))
    n_local = total_envs // world
    return rank * n_local, n_local


def gather_outputs(out_local: torch.Tensor, buf: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather the [R, N_local] SoA output blocks of every rank; returns [world, R, N_local]
    (rank-major, so global env id g lives at [g // N_local, :, g % N_local])."""
    world = dist.get_world_size(group)
    if buf is None:
        buf = torch.empty((world,) + tuple(out_local.shape), dtype=out_local.dtype, device=out_local.device)
    # concatenation form [world*R, N_local] (accepted by RCCL and gloo alike) of the same memory
    dist.all_gather_into_tensor(buf.view(world * out_local.shape[0], out_local.shape[1]), out_local.contiguous(),
                                group=group)
    return buf


def global_view(gathered: torch.Tensor) -> torch.Tensor:
    """[world, R, N_local] -> [N_total, R] (a strided view, no copy): row g = global env g."""
    w, r, n = gathered.shape
    return gathered.permute(0, 2, 1).reshape(w * n, r)


class ShardedPVDERVec:
    """``total_envs`` PVDER-v0 envs over the ranks of an initialised process group.

    ``step(local_actions)`` advances the local shard with the fused HIP launch and returns the
    GLOBAL (obs[N_total, 11], reward[N_total], done[N_total]) collated by one all-gather.
    """

    def __init__(self, total_envs: int, device, group=None, **env_kwargs):
        from .vec_env import PVDERVec
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.first, self.n_local = shard_range(total_envs, self.rank, self.world)
        self.total_envs = total_envs
        self.env = PVDERVec(self.n_local, device=device, env_id_base=self.first, **env_kwargs)
        self._buf = torch.empty((self.world, 13, self.n_local), dtype=self.env.out.dtype, device=self.env.device)

    def reset(self):
        self.env.reset()
        return self._collate()[0]

    def _collate(self):
        g = global_view(gather_outputs(self.env.out, self._buf, self.group))
        return g[:, :11], g[:, 11], g[:, 12] != 0

    def step(self, local_actions: torch.Tensor):
        self.env.step(local_actions, return_tensors=False)
        return self._collate() + ({},)
