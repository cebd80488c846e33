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

import ctypes as C
import operator
import os
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


# What a step collates (DESIGN.md "Multi-GPU"): the whole [13, N_local] block is what the north_star asks for and what
# the default does; at 65 536 envs per GPU it is 3.4 MB per peer and step -- 44 us on one xGMI link against an 8 us step
# kernel -- so a consumer that evaluates its policy where the env lives can ask for less:
#   "full"         rows 0..12 every step
#   "reward_done"  rows 11, 12 (reward, done) every step: 0.5 MB per peer
#   "full_every_k" (reward, done) every step, the whole block every k-th step
#   "obs8"         rows 0-7 (the eight dynamic observations) + 11, 12 every step: rows 8-10 -- Vdc_ref, Q_ref, tStart / max_sim_time
#                  -- follow from the actions and the step count (PVDER_env.py:198-229, :540; `bookkeeping_rows` below), which a
#                  consumer that chose the actions holds already: 10 of 13 rows, 23 % less on the links
PAYLOADS = ("full", "reward_done", "full_every_k", "obs8")
REWARD_ROW = 11
OBS8 = -8                      # include/pvder_hip.h PVDER_GATHER_OBS8: (first row, rows) = (OBS8, 10)
OBS8_ROWS = 10


def check_payload(payload: str, every: int = 1) -> None:
    """ONE validator for every layer (ShardedPVDERVec, NativeStepGather, bench.py): "full_every_k" needs a period of at
    least 2 -- with 1 it IS "full", and the C path (full_every <= 1: never a whole block) and the Python path used to
    disagree about it."""
    if payload not in PAYLOADS:
        raise ValueError("payload must be one of %s" % (PAYLOADS,))
    if payload == "full_every_k":
        try:
            every = operator.index(every)             # (numpy / torch integer scalars are integers too: ADVICE r4)
        except TypeError:
            every = 0
    if payload == "full_every_k" and every < 2:
        raise ValueError("payload 'full_every_k' needs gather_every >= 2 (got %r); use payload='full' for every step" % (every,))


def payload_rows(payload: str, step_index: int, every: int = 1) -> Tuple[int, int]:
    """(first row, number of rows) of the [13, N] output block that step number ``step_index`` (0-based) collates."""
    check_payload(payload, every)
    if payload == "full" or (payload == "full_every_k" and (step_index + 1) % every == 0):
        return 0, 13
    if payload == "obs8":
        return OBS8, OBS8_ROWS
    return REWARD_ROW, 2


def pack_obs8(out_local: torch.Tensor) -> torch.Tensor:
    """[13, N] output block -> the [10, N] block the "obs8" payload sends (rows 0-7, 11, 12)."""
    return torch.cat([out_local[:8], out_local[REWARD_ROW:REWARD_ROW + 2]], dim=0)


def bookkeeping_rows(prev: torch.Tensor, actions: torch.Tensor, done: torch.Tensor, del_qref: float, del_vdcref: float,
                     vdc_ref0: float, n_sub: int, inv_max_sim_time: float, auto_reset: bool = True,
                     prev_done: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rows 8-10 of the output block -- Vdc_ref, Q_ref, tStart / max_sim_time -- after the step that took ``actions`` from a
    state whose rows 8-10 were ``prev`` ([3, N] fp32), recomputed where the actions were chosen instead of shipped ("obs8"):
    action_calc (PVDER_env.py:198-229) in the kernels' fp32 arithmetic (del_qref / del_vdcref: pvder_step_cfg, per unit), the
    clock in sim time steps (:155-160, :540) and, with ``auto_reset``, the same-step restart of the envs whose ``done`` flag
    the step returned (Vdc_ref0, 0, 0).  Without auto-reset pass ``prev_done``: a finished env returns the stale tuple
    (:145-152).  Bit-identical to what the kernel wrote (tests/test_distributed.py)."""
    f32, dev = torch.float32, prev.device
    c = lambda x: torch.tensor(x, dtype=f32, device=dev)
    zero = torch.zeros((), dtype=f32, device=dev)
    a = actions.to(device=dev, dtype=torch.int64)
    q = prev[1] + torch.where(a == 1, c(del_qref), zero) - torch.where(a == 2, c(del_qref), zero)
    vdc = prev[0] + torch.where(a == 3, c(del_vdcref), zero) - torch.where(a == 4, c(del_vdcref), zero)
    # the clock: prev[2] = fp32(ksub / 60 * inv_tmax) -> ksub back (exact: ksub <= 30 000), advanced by n_sub
    ksub = torch.round(prev[2].double() * 60.0 / inv_max_sim_time).to(torch.int64) + n_sub
    t = (ksub.to(f32) * c(1.0 / 60.0)) * c(inv_max_sim_time)
    if auto_reset:
        d = done.to(dev).bool()
        vdc, q, t = torch.where(d, c(vdc_ref0), vdc), torch.where(d, zero, q), torch.where(d, zero, t)
    elif prev_done is not None:
        s_ = prev_done.to(dev).bool()
        vdc, q, t = torch.where(s_, prev[0], vdc), torch.where(s_, prev[1], q), torch.where(s_, prev[2], t)
    return torch.stack([vdc, q, t])


def gather_payload(out_local: torch.Tensor, step_index: int, payload: str = "full", every: int = 1, group=None):
    """One step's collective with a payload choice: returns (gathered [world, rows, N_local], first row, rows)."""
    r0, nr = payload_rows(payload, step_index, every)
    if r0 == OBS8:
        return gather_outputs(pack_obs8(out_local), None, group), r0, nr
    return gather_outputs(out_local[r0:r0 + nr], None, group), r0, nr


def global_view(gathered: torch.Tensor) -> torch.Tensor:
    """[world, R, N_local] -> [N_total, R] (a strided view, no copy): row g = global env g."""
    w, r, n = gathered.shape
    return gathered.permute(0, 2, 1).reshape(w * n, r)


class ShardedPVDERVec:
    """``total_envs`` PVDER-v0 envs over the ranks of an initialised process group.

    ``step(local_actions)`` advances the local shard with the fused HIP launch and returns the
    GLOBAL (obs[N_total, 11], reward[N_total], done[N_total]) collated by one all-gather.
    """

    def __init__(self, total_envs: int, device, group=None, payload: str = "full", gather_every: int = 1, **env_kwargs):
        """``payload`` / ``gather_every``: what a step collates (PAYLOADS).  With "reward_done" / "full_every_k" a step
        returns the GLOBAL reward and done and, on the steps without the whole block, ``None`` for the observation
        (the local shard's is ``self.env.obs``); with "obs8" the [N_total, 8] DYNAMIC observations (entries 0-7: the set
        points and the clock follow from the actions, `bookkeeping_rows`)."""
        from .vec_env import PVDERVec
        check_payload(payload, gather_every)
        self.payload, self.gather_every, self._step_index = payload, int(gather_every), 0
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.first, self.n_local = shard_range(total_envs, self.rank, self.world)
        self.total_envs = total_envs
        self.env = PVDERVec(self.n_local, device=device, env_id_base=self.first, **env_kwargs)
        self._buf = torch.empty((self.world, 13, self.n_local), dtype=self.env.out.dtype, device=self.env.device)

    def reset(self):
        self.env.reset()
        self._step_index = 0             # the whole-block cadence of "full_every_k" restarts with the episode
        return self._collate()[0]

    def _collate(self):
        g = global_view(gather_outputs(self.env.out, self._buf, self.group))
        return g[:, :11], g[:, 11], g[:, 12] != 0

    def step(self, local_actions: torch.Tensor):
        self.env.step(local_actions, return_tensors=False)
        k = self._step_index
        self._step_index += 1
        r0, nr = payload_rows(self.payload, k, self.gather_every)
        if nr == 13:
            return self._collate() + ({},)
        if r0 == OBS8:
            # the eight dynamic observations + reward + done of every env; rows 8-10: `bookkeeping_rows` where the actions are
            g = global_view(gather_outputs(pack_obs8(self.env.out), None, self.group))     # [N_total, 10]
            return g[:, :8], g[:, 8], g[:, 9] != 0, {}
        g = global_view(gather_outputs(self.env.out[r0:r0 + nr], None, self.group))       # [N_total, 2]
        return None, g[:, 0], g[:, 1] != 0, {}


class RcclAllGather:
    """The per-step all-gather issued straight into RCCL (ctypes on the ``librccl.so`` torch ships) on a
    dedicated stream: the same collective as ``all_gather_into_tensor``, without c10d's per-call
    bookkeeping.  Measured at world size 1 (bench.py): the c10d call + staging copy cost 50 us of
    host time per step -- more than the 35 us the step kernel runs -- against ~10 us this way.

    The communicator is created once from a unique id that rank 0 draws and the existing process
    group broadcasts.  ``launch(send, recv)`` orders the gather behind everything already enqueued
    on the current stream and returns the event that marks its completion; the caller makes whatever
    stream reuses ``send`` / reads ``recv`` wait on it (``stream.wait_event``)."""

    NCCL_FLOAT32 = 7          # ncclDataType_t: ncclFloat32 / ncclFloat

    class _UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._nccl = C.CDLL(path)
        n = self._nccl
        n.ncclGetUniqueId.restype = C.c_int
        n.ncclGetUniqueId.argtypes = [C.POINTER(self._UniqueId)]
        n.ncclCommInitRank.restype = C.c_int
        n.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self._UniqueId, C.c_int]
        n.ncclAllGather.restype = C.c_int
        n.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        n.ncclCommDestroy.restype = C.c_int
        n.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = self._UniqueId()
        if self.rank == 0:
            self._check(n.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [bytes(bytearray(uid)) if self.rank == 0 else None]      # all 128 bytes (c_char arrays stop at NUL)
        dist.broadcast_object_list(box, src=0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(n.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
            self.stream = torch.cuda.Stream(self.device)
        self._ready = torch.cuda.Event()

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: ncclResult_t %d" % (what, rc))

    def launch(self, send: torch.Tensor, recv: torch.Tensor, done: Optional[torch.cuda.Event] = None) -> torch.cuda.Event:
        """recv[world * numel] <- all-gather(send[numel]) (fp32, contiguous), asynchronously."""
        assert send.dtype == torch.float32 and recv.dtype == torch.float32
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel()
        self._ready.record(torch.cuda.current_stream(self.device))      # the block is complete behind this point
        self.stream.wait_event(self._ready)
        self._check(self._nccl.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self.NCCL_FLOAT32,
                                             self._comm, self.stream.cuda_stream), "ncclAllGather")
        done = done if done is not None else torch.cuda.Event()
        done.record(self.stream)
        return done

    def native(self, env) -> "NativeStepGather":
        """The same collective driven from C together with the step launch (no Python per step)."""
        return NativeStepGather(self, env)

    def close(self):
        if getattr(self, "_comm", None):
            self.stream.synchronize()
            self._nccl.ncclCommDestroy(self._comm)
            self._comm = None


class NativeStepGather:
    """K env steps, each followed by the all-gather of its output block, from ONE host call
    (include/pvder_hip.h ``pvder_plan_rollout_gather``): the step kernel on torch's current stream, the
    gather on the C object's own stream, two output blocks alternating so that the gather of step t
    overlaps the kernel of step t+1.  Measured at world size 1 (bench.py, forced): the Python-level
    sequence (wait_event, step, record, wait, ncclAllGather via ctypes, record) costs ~29 us of host time
    per step against an 11 us kernel."""

    def __init__(self, rccl: RcclAllGather, env):
        from . import _native
        self._lib, self.env, self.world = _native.lib(), env, rccl.world
        fn = C.cast(rccl._nccl.ncclAllGather, C.c_void_p)
        self._h = C.c_void_p()
        with torch.cuda.device(env.device):
            _native.check(self._lib.pvder_gather_create(C.byref(self._h), fn, rccl._comm), "pvder_gather_create")
        n = env.num_envs
        # (the second block starts as a copy of the first: without auto-reset the first step carries the stale tuples
        # of finished envs over from "the other block", which must hold them)
        self.out = [env.out, env.out.clone()]
        self.recv = [torch.empty((self.world * 13, n), dtype=torch.float32, device=env.device) for _ in range(2)]
        self._k = 0

    payload, gather_every = "full", 1        # what a step collates (PAYLOADS); set before the first rollout

    def rollout(self, actions: torch.Tensor, sync_each: bool = False):
        """``actions``: [K, N] uint8 device tensor.  After ``drain()`` the collated block of the last
        step is ``self.recv[(steps_so_far - 1) & 1]`` (its first world * rows rows for a partial payload)."""
        from . import _native
        env = self.env
        assert actions.dtype == torch.uint8 and actions.is_contiguous() and actions.shape[1] == env.num_envs
        plan = env._plan or env._make_plan()
        K = int(actions.shape[0])
        check_payload(self.payload, self.gather_every)
        with env._device_ctx():
            r0, nr = payload_rows(self.payload, 0, 1) if self.payload != "full_every_k" else (REWARD_ROW, 2)
            rc = self._lib.pvder_plan_rollout_gather_rows(plan, self._h, env.fstate.data_ptr(), env.istate.data_ptr(),
                                                          env._ev_vg.data_ptr(), env._ev_si.data_ptr(), actions.data_ptr(), K,
                                                          self.out[0].data_ptr(), self.out[1].data_ptr(),
                                                          self.recv[0].data_ptr(), self.recv[1].data_ptr(),
                                                          env.stats.data_ptr(), env._stream(), self._k & 1, 1 if sync_each else 0,
                                                          r0, nr, self.gather_every if self.payload == "full_every_k" else 0,
                                                          self._k)
        _native.check(rc, "pvder_plan_rollout_gather")
        self._k += K
        if K:
            env.out = self.out[(self._k - 1) & 1]

    def drain(self):
        from . import _native
        with self.env._device_ctx():
            _native.check(self._lib.pvder_gather_drain(self._h, self.env._stream()), "pvder_gather_drain")

    def last(self) -> torch.Tensor:
        return self.recv[(self._k - 1) & 1]

    def close(self):
        if self._h:
            self._lib.pvder_gather_destroy(self._h)
            self._h = None
