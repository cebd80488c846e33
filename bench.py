#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched PVDER-v0 hot path on N MI355X (contract: task statement).

Default integrator: etdlin -- fp32 ETD3RK (Cox-Matthews exponential Runge-Kutta) on the FULL nonlinear model,
split as f(y) = L y + N(y) with L = its 11x11 Jacobian at the nominal operating point: L sits in the matrix
exponential (nothing of the model is approximated -- N = f - L y is integrated by the stages), so an env step
is an exponential-Euler tick of 2/1920 s behind the action / event jump + TWO equal ETD3RK steps over
the rest, with the dense matrix-vector products on the matrix cores (fp32 MFMA); `--integrator etdrk4`
is round 1's 16-step scheme, the literal classic-RK4 kernel is timed in the same run ("classic_rk4").
All meet the 1e-4 rel-L2 parity bar (tests/, -m gpu).

Workload (BASELINE.json configs[2], the configuration the north_star target is quoted on; with
--gpus N it is the per-GPU shard of configs[3] = weak scaling): 65 536 parallel PVDER-v0 envs per
GPU, PVDER-v0 default kwargs (n = 15 sim time steps of 1/60 s per env step, discrete
voltage_regulation reward), grid-voltage AND insolation perturbation events on, fp32, uniform random
actions resident on the device, same-step auto-reset.  A "step" is one batched env transition (one
fused HIP launch over all local envs); for N > 1 each step ends with ONE RCCL all-gather of the
[13, N_local] output block (11 obs + reward + done).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (fresh child
processes, one per GPU, before this process makes any GPU call) and relays rank 0's line; under
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0.  `value` counts env transitions of all ranks / max-over-ranks time of
exactly --steps steps; `sustained` repeats the measurement over >= 1 s of steps in the same process
(a 20-step window is 0.5 ms of GPU time); `roofline` prices the step kernel's algorithmic bytes
against HBM, `valu_roofline` its flops against the fp32 vector peak (the roof that binds this path).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))
sys.path.insert(0, ROOT)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory: -4 us per launch

HBM_PEAK = 8.0e12                 # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
VALU_PEAK_F32 = 157.3e12          # FLOP/s, MI355X_MICROARCH.md "Peak FP32 (vector)"
VALU_PEAK_F64 = 78.6e12           # FLOP/s, MI355X_MICROARCH.md "Peak FP64 (vector)"
BYTES_PER_ENV_STEP = {"f32": 269, "f64": 533}      # three-phase, carried words + obs + reward + done
# flop / instructions per integrator step, counted from the gfx950 ISA of the inner loops (DESIGN.md "Kernels")
# (etdlin: an ETD3RK step = 3 evaluations of f (~70 flop) + 6 dense 11x11 matrix-vector products on the
#  matrix cores (242 flop each); the tick = 1 evaluation + 1 product)
FLOP_PER_STEP = {"rk4": 1201, "etdrk4": 907, "etdlin": 3 * 70 + 6 * 242}
FLOP_PER_TICK = 70 + 242
FLOP_PER_K4 = 4 * 70 + 10 * 242       # (round 4: the quiet env step as ONE Krogstad step: 4 evaluations, 10 products, no tick)
INSTR_PER_STEP = {"etdrk4": 396}
DEFAULT_MICRO = {"etdlin": 32, "etdrk4": 32, "rk4": 4096, "rk45": 1}
# (the kernel a leg ran is asked of the plan itself: PVDERVec.kernel_name() -> pvder_plan_kernel_name, ABI 17)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--n-sub", type=int, default=15)
    ap.add_argument("--micro", type=int, default=None,
                    help="ETDRK4: first step behind an input jump = 1/(60*micro) s (default 32); classic RK4: steps per 1/60 s (4096)")
    ap.add_argument("--graded", type=int, default=1,
                    help="ETDRK4: 1 = two steps (first + remainder) in a sim time step that starts at a jump, 0 = micro uniform steps")
    ap.add_argument("--micro-coarse", type=int, default=None,
                    help="ETDRK4: steps per sim time step that does not start at an input jump (default 1)")
    ap.add_argument("--integrator", default="etdlin", choices=["etdlin", "etdrk4", "rk4", "rk45"])
    ap.add_argument("--classic-steps", type=int, default=3, help="classic-RK4 steps timed beside the main run (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget (0 = skip)")
    ap.add_argument("--fixed-policy", type=int, default=1, help="1: time the four fixed-action ramps of calc_returns' kind (0 = skip)")
    ap.add_argument("--rk45-steps", type=int, default=2, help="fp64 RK45 steps (BASELINE configs[4]) timed beside the main run (0 = skip)")
    ap.add_argument("--large-batch", type=int, default=1 << 20, help="envs of the large_batch leg (same kernel, one launch per env step; 0 = skip)")
    ap.add_argument("--side-legs", type=int, default=1, help="1: the configs1 (4096 envs, n = 4: etdlin and classic RK4) and desynchronised "
                                                             "(episode phases spread) legs (0 = skip)")
    ap.add_argument("--single-env-steps", type=int, default=1000, help="steps of the single_env leg: make('PVDER-v0'), random agent (0 = skip)")
    ap.add_argument("--time-every", type=int, default=64,
                    help="bracket every K-th step kernel with HIP events (clamped so that even a short run gets samples); 0 = none")
    ap.add_argument("--sustained-seconds", type=float, default=1.0,
                    help="length of the sustained leg that follows the --steps window (0 = skip)")
    ap.add_argument("--preroll-steps", type=int, default=20000,
                    help="untimed steps BEFORE the --warmup steps (0.2 s of GPU work: clocks, TLBs and code paths at their "
                         "steady state when the warm-up starts; a 5-step warm-up alone leaves a 20-step window ramping)")
    ap.add_argument("--graph-window", type=int, default=1,
                    help="1: a window shorter than 256 steps replays ONE HIP graph of its per-step launches (captured "
                         "before the window); 0: eager launches")
    ap.add_argument("--fused-seconds", type=float, default=0.5,
                    help="length of the separately labelled fused-rollout leg (0 = skip)")
    ap.add_argument("--fused-steps", type=int, default=160, help="env steps per fused launch in that leg")
    ap.add_argument("--collective", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: all-gather issued directly into RCCL (default) or through torch.distributed")
    ap.add_argument("--payload", default="full", choices=["full", "reward_done", "full_every_k", "obs8"],
                    help="N > 1: what a step collates -- the whole [13, N_local] block (default: what BASELINE configs[3] "
                         "names), (reward, done) only, or (reward, done) with the whole block every --gather-every steps")
    ap.add_argument("--gather-every", type=int, default=8)
    ap.add_argument("--sync-gather", type=int, default=0,
                    help="1: wait for each step's all-gather before the next step (no compute/comm overlap)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the rank launcher: every rank joins a gloo group, rank 0 prints ranks_seen")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# rank launcher: `python bench.py --gpus N` outside torchrun
# ---------------------------------------------------------------------------------------------------
def visible_gpu_count(sysfs: str = "/sys/class/kfd/kfd/topology/nodes") -> int:
    """GPUs this process would see, WITHOUT loading the HIP runtime (the parent that starts the ranks must not touch
    the GPU): the length of ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one is set, else the
    KFD topology nodes with compute units (simd_count > 0; CPU nodes report 0).  -1: cannot tell (no KFD sysfs)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        n = 0
        for node in os.listdir(sysfs):
            with open(os.path.join(sysfs, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except OSError:
        return -1


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Start n fresh child processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), relay
    their output (rank 0 prints the JSON line), return the worst exit code.  The parent makes no GPU
    call: a process that has initialised the GPU must not be replaced or forked into ranks."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PVDER_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:            # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def payload_rows_per_step(payload: str, gather_every: int) -> float:
    """Rows of the [13, N_local] block a step collates on average."""
    return {"full": 13.0, "reward_done": 2.0, "obs8": 10.0}.get(payload, 2.0 + 11.0 / max(gather_every, 2))


def expected_payload_efficiency(world: int, n_local: int, kernel_s: float, gather_every: int) -> dict:
    """What xGMI allows each payload BEFORE anything is measured (a single 8-GPU driver run should explain itself):
    a peer's block of `rows` rows crosses ONE link direction (76.5 GB/s, fully connected mesh, all peers in parallel), the
    gather of step t overlaps the kernel of step t+1, so a step cannot take less than max(kernel, link time): expected weak-
    scaling efficiency = kernel / max(kernel, link)."""
    out = {}
    for pl in ("full", "obs8", "reward_done", "full_every_k"):
        rows = payload_rows_per_step(pl, gather_every)
        link_s = rows * n_local * 4 / 76.5e9 if world > 1 else 0.0
        out[pl] = {"rows_per_step": rows, "link_us_per_step": 1e6 * link_s, "kernel_us_per_step": 1e6 * kernel_s,
                   "expected_efficiency": finite(kernel_s / max(kernel_s, link_s)) if kernel_s > 0 else None}
    return out


def scaling_note(world: int, value: float, legs: dict, expected: dict) -> str:
    """One sentence a reader of a lone N > 1 line needs: `value` is the payload configs[3] names ("full": the whole
    [13, N_local] block all-gathered every step); next to it the rate with no collation at all (`compute_only`: what the step path
    itself scales to -- envs are independent, no collective on the integration path) and, per payload, the efficiency xGMI allows
    at this kernel speed beside the one measured in this run (rate of the leg / compute_only rate)."""
    co = (legs.get("compute_only") or {}).get("value")
    parts = []
    for pl in ("full", "obs8", "full_every_k", "reward_done"):
        exp = (expected.get(pl) or {}).get("expected_efficiency")
        got = (legs.get(pl) or {}).get("value")
        if exp is None and got is None:
            continue
        parts.append("%s expected %s measured %s" % (pl, "%.2f" % exp if exp is not None else "n/a",
                                                       "%.2f" % (got / co) if (got and co) else "n/a"))
    return ("value = payload 'full' at %d GPUs (%.3g env-steps/s): every step all-gathers the whole [13, N_local] block, as BASELINE "
            "configs[3] names it; the step path alone (compute_only, no collation) runs %s env-steps/s at the same %d GPUs -- the "
            "figure to hold against N x the 1-GPU value for the scaling of the integration path.  A peer's block crosses one xGMI "
            "link direction (76.5 GB/s) and overlaps the next step's kernel, so a step takes max(kernel, link): efficiency against "
            "compute_only per payload -- %s." % (world, value, "%.3g" % co if co else "n/a", world, "; ".join(parts)))


def xgmi_roofline(world: int, n_local: int, rows: float, sec_per_step: float) -> dict:
    """What the collective must move and the bound xGMI puts on it (MI355X_MICROARCH.md: 7 links x ~153 GB/s
    bidirectional per GPU, i.e. ~77 GB/s into a GPU per link, all peers in parallel)."""
    recv = (world - 1) * n_local * 4 * rows
    return {"bound": "xgmi", "bytes_received_per_step_per_gpu": recv,
            "achieved": finite(recv / sec_per_step / 1e9), "peak": 7 * 76.5, "unit": "GB/s",
            "frac": finite(recv / sec_per_step / (7 * 76.5e9)),
            "frac_of_the_links_in_use": finite(recv / sec_per_step / (max(world - 1, 1) * 76.5e9)),
            "note": "bytes a GPU receives per step / step time, against 7 links x 76.5 GB/s into one GPU "
                    "(MI355X_MICROARCH.md); with fewer than 8 ranks only world-1 links carry data"}


def payload_leg(world: int, n_local: int, steps: int, seconds: float, payload: str, gather_every: int, host_s=None) -> dict:
    rows = payload_rows_per_step(payload, gather_every)
    d = {"value": world * n_local * steps / seconds, "unit": "env-steps/s", "steps": steps, "ms_per_step": 1e3 * seconds / steps,
         "rows_per_step": rows, "xgmi_roofline": xgmi_roofline(world, n_local, rows, seconds / steps)}
    if host_s is not None:
        d["host_enqueue_ms_per_step"] = 1e3 * host_s / steps
    return d


def selftest_rank():
    """--selftest-launcher: what a rank does, minus the GPU (gloo all-reduce of the rank ids)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = 1
    dist.all_reduce(t)
    # the N > 1 line's three legs -- no collation / the whole block / (reward, done) -- with a stand-in for the step launch
    # (gloo on the CPU: the STRUCTURE of the line, not a rate)
    from gym_PVDER_amd.distributed import gather_payload
    n_local, steps = 256, 4
    blk = torch.full((13, n_local), float(rank))
    legs = {}
    for name in ("compute_only", "full", "obs8", "reward_done"):
        dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            blk += 1.0                                   # "the step"
            if name != "compute_only":
                gather_payload(blk, k, name, 1)
        dist.barrier()
        el = time.perf_counter() - t0
        legs[name] = ({"value": world * n_local * steps / el, "unit": "env-steps/s", "steps": steps, "ms_per_step": 1e3 * el / steps}
                      if name == "compute_only" else payload_leg(world, n_local, steps, el, name, 1))
    if rank == 0:
        exp = expected_payload_efficiency(world, 65536, 8e-6, 8)
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_seen": int(t.sum()), "payload_legs": legs,
                          "expected_payload_efficiency": exp,
                          "scaling_note": scaling_note(world, legs["full"]["value"], legs, exp)}))
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------
# measurement helpers
# ---------------------------------------------------------------------------------------------------
def events_on_spec():
    from gym_PVDER_amd.vec_env import default_events_spec
    s = default_events_spec()
    s["insolation"]["ENABLE"] = True          # update_env_events([{'insolation': {'ENABLE': True}}])
    return s


def cpu_baseline(seconds: float, seed: int = 1):
    """The oracle (fp64 SciPy-odeint restatement of the reference's step path, "port") timed on
    one host core the way the reference times itself: wall time around step (PVDER_env.py:140,193)."""
    from oracle.cpu_worker import run
    v, n, used = run(seed, seconds)
    return {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d env steps of 1 env, PVDER-v0 defaults (n=15, LSODA rtol=atol=1e-4, hmax=1/120), "
                      "random actions, %.1f s of CPU" % (n, used)}


def cpu_baseline_all_cores(seconds: float):
    """BASELINE.md plan: one oracle env per process on every host core (the reference itself is
    single-threaded; this is the most its CPU path can do on this box).  Plain child processes,
    started before this process touches the GPU."""
    # the GPU box gives one GPU's job a 16-core CPU share (os.cpu_count() reports the whole host)
    n = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py")]
    procs = [subprocess.Popen(cmd + [str(100 + i), str(seconds)], stdout=subprocess.PIPE, text=True) for i in range(n)]
    vals = []
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 6 + 120)
        vals.append(float(out.split()[0]))
    return {"value": float(sum(vals)), "unit": "env-steps/s", "cores": n, "kind": "port",
            "sample": "%d processes x 1 env, %.1f s of CPU each, same settings as cpu_baseline" % (n, seconds)}


def classic_rk4_leg(torch, dev, N, n_sub, steps, micro=3750):
    """The literal fixed-step classic RK4 kernel on the same workload: `micro` steps per 1/60 s.  3750 (h = 4.44 us) puts
    |h lambda|max at 2.82 of the scheme's 2.95 along the fast poles' direction at the nominal operating point (2.94 where a
    policy has walked Vdc_ref up 9 %: the library's default stays 4096); parity at this step count:
    tests/test_gpu_golden.py::test_classic_rk4_at_3750_micro_steps_vs_golden."""
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, integrator="rk4", micro_steps=micro, seed=2024,
                   events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    acts = torch.randint(0, 5, (steps + 1, N), generator=gen, device=dev, dtype=torch.uint8)
    env.step(acts[0], return_tensors=False)
    env.kernel_timing(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        env.step(acts[i], return_tensors=False)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    k = env.kernel_times_ms()
    avg = sum(k) / len(k) * 1e-3
    fl = FLOP_PER_STEP["rk4"] * micro * n_sub * N / avg
    return {"value": N * steps / el, "unit": "env-steps/s", "steps": steps, "micro_steps": micro,
            "h_lambda_max": 2.582 * 4096.0 / micro, "h_lambda_bound": 2.95,
            "kernel": env.kernel_name(), "kernel_avg_ms": avg * 1e3,
            "valu_frac": fl / VALU_PEAK_F32, "hbm_frac": BYTES_PER_ENV_STEP["f32"] * N / avg / HBM_PEAK}


def fixed_policy_leg(torch, dev, N, n_sub):
    """The reference's evaluation driver holds ONE action for a whole episode (calc_returns, PVDER_env.py:480-487:
    'dec' -> action 1, 'no_change' -> action 2; actions 3 / 4 are the Vdc_ref counterparts): 160 env steps each,
    same envs and events as the headline, one launch per env step, timed in 10-step blocks with HIP events.
    Action 1 / 2 end at the inverter's current limit (careful pass), action 4 at the duty-cycle limit (23-state
    path on the switched operator sets): what a policy that holds an action costs the whole batch."""
    from gym_PVDER_amd.vec_env import PVDERVec
    out = {}
    for act, name in ((1, "dec: action 1, Q_ref +375 VAR per step"), (2, "no_change: action 2, Q_ref -375 VAR per step"),
                      (3, "action 3, Vdc_ref +0.3 V per step"), (4, "action 4, Vdc_ref -0.3 V per step")):
        env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, seed=2024, events_spec=events_on_spec(),
                       auto_reset=False)
        env.reset()
        acts = torch.full((10, N), act, dtype=torch.uint8, device=dev)
        n_blocks = 16 if n_sub == 15 else max(1, int(env.max_sim_time * 60 / n_sub) // 10)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_blocks + 1)]
        torch.cuda.synchronize(dev)
        ev[0].record()
        for b in range(n_blocks):
            env.rollout(acts, fused=False)
            ev[b + 1].record()
        torch.cuda.synchronize(dev)
        blocks = [ev[b].elapsed_time(ev[b + 1]) * 100.0 for b in range(n_blocks)]       # us per step
        st = env.step_stats()
        out["action_%d" % act] = {"policy": name, "steps": 10 * n_blocks, "mean_us_per_step": sum(blocks) / len(blocks),
                                  "worst_10_step_block_us_per_step": max(blocks),
                                  "us_per_step_by_10_step_block": [round(b, 1) for b in blocks],
                                  "n_nonfinite": st["n_nonfinite"], "Q_ref_end": float(env.Q_ref[0]), "Vdc_ref_end": float(env.Vdc_ref[0])}
    out["mean_us_per_step_all_four"] = sum(v["mean_us_per_step"] for v in out.values() if isinstance(v, dict) and "mean_us_per_step" in v) / 4.0
    # ... and what ONE env at the duty-cycle limit (Vdc_ref -9 %) costs the other N - 1, which run on random actions: a launch
    # lasts as long as its slowest workgroup (tools/one_saturated_env.py; round 2: 55 ms, round 3: 75 us)
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, seed=2, events_spec=events_on_spec(), auto_reset=False)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    acts = torch.randint(0, 5, (40, N), generator=gen, device=dev, dtype=torch.uint8)
    acts[:, 0] = 0
    env.fstate[env.ns + 1, 0] = float(env.Vdc_ref[0]) * (1.0 - 0.09)
    env.rollout(acts, fused=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(); env.rollout(acts, fused=False); e1.record()
    torch.cuda.synchronize(dev)
    f = env.fstate
    out["one_saturated_env"] = {"us_per_step": e0.elapsed_time(e1) * 1e3 / 40, "steps": 40,
                                "m_abs_of_that_env": float((torch.complex(f[2, 0], f[3, 0]) + 6000.0 * torch.complex(f[4, 0], f[5, 0])).abs()),
                                "workload": "%d envs on random actions, env 0 held at Vdc_ref -9 %% (duty cycle at its limit)" % N,
                                "n_nonfinite": env.step_stats()["n_nonfinite"]}
    out["note"] = ("%d envs, off_range='integrate' (the default: nothing terminates); parity of these ramps against the converged "
                   "bridge: tests/test_gpu_etdlin.py::test_fixed_action_ramps_vs_converged_bridge" % N)
    return out


def etdlin64_leg(torch, dev, N, n_sub, steps=40):
    """BASELINE configs[4] on the tolerance-match path that is not stability-bound: fp64 etdlin with step doubling on a
    dyadic menu as its error control (rtol 1e-6, atol 1e-8 in the RK45 kernel's norm), the RK45 kernel behind it."""
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, integrator="etdlin64", seed=2024,
                   events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(78)
    acts = torch.randint(0, 5, (steps + 4, N), generator=gen, device=dev, dtype=torch.uint8)
    for i in range(4):
        env.step(acts[i], return_tensors=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(4, steps + 4):
        env.step(acts[i], return_tensors=False)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    return {"value": N * steps / el, "unit": "env-steps/s", "steps": steps, "dtype": "f64", "rtol": env.rtol, "atol": env.atol,
            "workload": "BASELINE configs[4]: %d envs, fp64 etdlin with step doubling (tolerance-match), same events and random actions" % N,
            "kernel": "step_etdlin64_kernel (+ step_rk45_kernel<double,3> for envs handed over: none in this run)",
            "ms_per_step": 1e3 * el / steps, "hbm_frac": BYTES_PER_ENV_STEP["f64"] * N / (el / steps) / HBM_PEAK,
            "n_nonfinite": env.step_stats()["n_nonfinite"],
            "parity": "tests/test_gpu_fullsize.py::test_configs4_fp64_etdlin64_full_size_vs_converged_bridge (< 2e-6)"}


def rk45_leg(torch, dev, N, n_sub, steps=2):
    """BASELINE configs[4]: the same batch through the fp64 adaptive Dormand-Prince kernel (rtol 1e-6, atol 1e-8) --
    the tolerance-match path; stability-bound at h ~ 4 us."""
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, integrator="rk45", seed=2024,
                   events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    acts = torch.randint(0, 5, (steps + 1, N), generator=gen, device=dev, dtype=torch.uint8)
    env.step(acts[0], return_tensors=False)
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    r0.record()
    for i in range(1, steps + 1):
        env.step(acts[i], return_tensors=False)
    r1.record()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    h = env.hstep[env.hstep > 0]
    spe = float((n_sub / 60.0) / float(h.double().mean().item())) if h.numel() else None
    avg = r0.elapsed_time(r1) * 1e-3 / steps
    fl = (7 * 300 + 23 * 60) * spe * N / avg if spe else None
    return {"value": N * steps / el, "unit": "env-steps/s", "steps": steps, "dtype": "f64", "rtol": env.rtol, "atol": env.atol,
            "workload": "BASELINE configs[4]: %d envs, fp64 adaptive RK45, same events and random actions" % N,
            "kernel": env.kernel_name(), "ms_per_step": 1e3 * el / steps, "kernel_avg_ms": avg * 1e3,
            "integrator_steps_per_env_step": spe, "valu_f64_frac": finite(fl / VALU_PEAK_F64) if fl else None,
            "hbm_frac": BYTES_PER_ENV_STEP["f64"] * N / avg / HBM_PEAK, "n_nonfinite": env.step_stats()["n_nonfinite"]}


def single_env_leg(steps=1000):
    """BASELINE configs[0] on the PRODUCT: ``gym_PVDER_amd.make('PVDER-v0')`` (the drop-in's own single env = a PVDERVec of one
    env), random agent, `steps` steps with reset() at every episode end -- the loop of the reference's
    examples/gym_PVDER_environment_import_test.py:23-31.  One launch + one host read-back per step: a latency figure."""
    import gym_PVDER_amd as gp
    env = gp.make('PVDER-v0')
    env.reset()
    for _ in range(20):
        env.step(env.action_space.sample())
    env.reset()
    t0 = time.perf_counter()
    episodes = 0
    for _ in range(steps):
        _, _, done, _ = env.step(env.action_space.sample())
        if done:
            env.reset()
            episodes += 1
    el = time.perf_counter() - t0
    return {"value": steps / el, "unit": "env-steps/s", "steps": steps, "episodes_finished": episodes, "us_per_step": 1e6 * el / steps,
            "workload": "BASELINE configs[0]: 1 env, gym_PVDER_amd.make('PVDER-v0') defaults, random agent, reset() at every "
                        "episode end; one launch and one host read-back (obs, reward, done as Python objects) per step"}


def lane_params_leg(torch, dev, n_sub, sizes=(16384, 65536), steps=40):
    """The per-env parameter override (domain randomisation over C, the outer-loop gains, the PV array size and the
    insolation scale; PVDERVec(lane_params=...)): every env of the batch with its OWN set, one launch per env step -- on the
    matrix-core path (the default integrator, round 5: the group's exponential operator, the env's own constants in the right-hand
    side, two Krogstad steps per env step: step_etdlin_lane_kernel) and on the 23-state ETDRK4 path (integrator="etdrk4": the
    phase-per-lane kernel up to 32 768 envs, the lane-per-env kernel beyond)."""
    import numpy as np
    from gym_PVDER_amd import der_model
    from gym_PVDER_amd.vec_env import PVDERVec
    out = {}
    for N in sizes:
        rng = np.random.default_rng(1)
        base = der_model.DER_CONFIG["50"]
        sets = {"C_actual": base["C_actual"] * rng.uniform(0.7, 1.4, N), "Kp_DC": base["Kp_DC"] * rng.uniform(0.8, 1.25, N),
                "Ki_DC": base["Ki_DC"] * rng.uniform(0.8, 1.25, N), "Kp_Q": base["Kp_Q"] * rng.uniform(0.8, 1.25, N),
                "Ki_Q": base["Ki_Q"] * rng.uniform(0.8, 1.25, N), "Np": rng.integers(10, 12, N).astype(np.float64),
                "Sinsol_scale": rng.uniform(0.9, 1.0, N)}
        for integ in ("etdlin", "etdrk4"):
            env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, seed=2, events_spec=events_on_spec(), auto_reset=True,
                           integrator=integ, lane_params=sets)
            env.reset()
            gen = torch.Generator(device=dev)
            gen.manual_seed(9)
            a = torch.randint(0, 5, (steps, N), generator=gen, device=dev, dtype=torch.uint8)
            env.rollout(a, fused=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(); env.rollout(a, fused=False); e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / steps
            row = {"us_per_step": us, "value": N / us * 1e6, "unit": "env-steps/s", "distinct_parameter_sets": N,
                   "kernel": "step_etdlin_lane_kernel" if integ == "etdlin" else ("step_etdrk4_lane3_kernel" if N <= 32768 else "step_etdrk4_lane_kernel"),
                   "n_nonfinite": env.step_stats()["n_nonfinite"]}
            # (keys of rounds 3-5 -- envs_16384 / envs_65536 -- stay the ETDRK4 kernels; the matrix-core path beside them)
            out[("envs_%d" % N) if integ == "etdrk4" else ("etdlin_envs_%d" % N)] = row
    out["parity"] = ("tests/test_gpu_lane_params.py (64 sets vs the bridge integrating each: ETDRK4 lane kernels 3.4e-5, etdlin lane kernel "
                     "< 1e-4 -- see the test; the two ETDRK4 kernels bit for bit)")
    return out


def configs1_leg(torch, dev, steps=400, classic_steps=2):
    """BASELINE configs[1]: 4 096 parallel envs, n = 4 sim time steps per env step, the requested max_sim_time of 40 s clamped by
    the reference to 500 * n / 60 = 33.33 s (PVDER_env.py:569-571: 500 steps per episode, the TimeLimit of gym_PVDER/__init__.py:10),
    default events (voltage only, :60-61), random actions, auto-reset -- the default step path (etdlin) and the literal
    "fixed-step RK4 fp32" of that config line (3750 micro-steps per 1/60 s) side by side."""
    from gym_PVDER_amd.vec_env import PVDERVec
    N, n_sub = 4096, 4
    out = {"workload": "BASELINE configs[1]: 4096 envs, n=4, max_sim_time 40 -> 33.33 s (500 env steps per episode), voltage events, "
                       "random actions, auto-reset, one launch per env step"}
    gen = torch.Generator(device=dev)
    gen.manual_seed(41)
    acts = torch.randint(0, 5, (steps, N), generator=gen, device=dev, dtype=torch.uint8)
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, max_sim_time=40.0, seed=2024, auto_reset=True)
    assert abs(env.max_sim_time - 500 * n_sub / 60.0) < 1e-9
    env.reset()
    env.rollout(acts[:100], fused=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e0.record(); env.rollout(acts, fused=False); e1.record()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    out["etdlin"] = {"value": N * steps / el, "unit": "env-steps/s", "steps": steps, "us_per_step": 1e6 * el / steps,
                     "region_us_per_step": e0.elapsed_time(e1) * 1e3 / steps, "kernel": env.kernel_name(),
                     "n_nonfinite": env.step_stats()["n_nonfinite"], "max_sim_time": env.max_sim_time}
    del env
    rk = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, max_sim_time=40.0, integrator="rk4", micro_steps=3750, seed=2024,
                  auto_reset=True)
    rk.reset()
    rk.step(acts[0], return_tensors=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e0.record()
    for i in range(1, classic_steps + 1):
        rk.step(acts[i], return_tensors=False)
    e1.record()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    out["classic_rk4"] = {"value": N * classic_steps / el, "unit": "env-steps/s", "steps": classic_steps, "micro_steps": 3750,
                          "us_per_step": 1e6 * el / classic_steps, "region_us_per_step": e0.elapsed_time(e1) * 1e3 / classic_steps,
                          "kernel": rk.kernel_name(), "n_nonfinite": rk.step_stats()["n_nonfinite"],
                          "note": "4096 envs are 16 workgroups on 256 CUs: a latency figure (the 65 536-env rate is the classic_rk4 key)"}
    return out


def desynchronised_leg(torch, dev, N, n_sub, steps=640, warm=160):
    """The headline's workload with the envs' episode phases SPREAD: staggered masked resets leave env e at
    clock (e * 7919 mod 160) env steps into its episode, so that every wave mixes quiet lanes with lanes at an event row, one
    env in 160 ends its episode in any step, and the launch's clock hint (pvder_plan_hint: where a batch in lockstep IS)
    names the clock of 1 env in 160.  The per-lane schedule is tested for correctness
    (tests/test_gpu_etdlin.py::test_etdlin_schedule_is_per_lane); this is what it costs."""
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, seed=2024, events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    ep = int(round(env.max_sim_time * 60 / n_sub))                    # env steps per episode (160)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    acts = torch.randint(0, 5, (max(steps, ep), N), generator=gen, device=dev, dtype=torch.uint8)
    phase = (torch.arange(N, device=dev, dtype=torch.int64) * 7919) % ep
    for k in range(ep):               # after these ep steps env e is (ep - 1 - phase[e]) steps behind its last masked reset
        env.step(acts[k], return_tensors=False)
        env.reset(phase == k)
    def timed():
        env.rollout(acts[:warm], fused=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        e0.record(); env.rollout(acts[:steps], fused=False); e1.record()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3 / steps
    spread = int(torch.unique(env.istate[0]).numel())
    el_u, per_u = timed()                              # the envs where the resets left them: every wave mixes clocks
    env.sort_lanes(lane_io=True)                       # ... and re-ordered in the buffers by clock (round 6: PVDERVec.sort_lanes)
    el, per = timed()
    return {"value": N * steps / el, "unit": "env-steps/s", "steps": steps, "us_per_step": 1e6 * el / steps,
            "region_us_per_step": per * 1e6, "distinct_clocks_in_the_batch": spread, "episode_steps": ep,
            "hbm_frac": BYTES_PER_ENV_STEP["f32"] * N / per / HBM_PEAK, "kernel": env.kernel_name(),
            "n_nonfinite": env.step_stats()["n_nonfinite"],
            "unsorted": {"value": N * steps / el_u, "us_per_step": 1e6 * el_u / steps, "region_us_per_step": per_u * 1e6,
                         "note": "the same batch before sort_lanes(): waves mix quiet and event lanes, 64 event rows per wave and load"},
            "workload": "%d envs, events on, random actions, auto-reset, episode phases spread over all %d clocks by staggered masked "
                        "resets (a few envs end their episode in every step; the clock hint misses), the envs re-ordered in the buffers "
                        "by clock (PVDERVec.sort_lanes: homogeneous waves, one event row per wave; bit-identical per env id)" % (N, ep)}


def large_batch_leg(torch, dev, n_sub, N=1 << 20, steps=300, warm=60):
    """The same workload, kernel and launch-per-env-step at 1 048 576 envs per GPU: sixteen waves per SIMD's worth of envs
    instead of one -- the batch at which the step kernel's loads, arithmetic and stores overlap across waves."""
    from gym_PVDER_amd.vec_env import PVDERVec
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=n_sub, seed=2024, events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(55)
    acts = torch.randint(0, 5, (steps, N), generator=gen, device=dev, dtype=torch.uint8)
    env.rollout(acts[:warm], fused=False)
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    r0.record()
    env.rollout(acts, fused=False)
    r1.record()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    per = r0.elapsed_time(r1) * 1e-3 / steps
    return {"value": N * steps / el, "unit": "env-steps/s", "envs": N, "steps": steps, "ms_per_step": 1e3 * el / steps,
            "region_ms_per_step": per * 1e3, "hbm_frac": BYTES_PER_ENV_STEP["f32"] * N / per / HBM_PEAK,
            "achieved_GBps": BYTES_PER_ENV_STEP["f32"] * N / per / 1e9, "kernel": env.kernel_name(),
            "n_nonfinite": env.step_stats()["n_nonfinite"],
            "note": "one launch per env step, same events / random actions / auto-reset as the headline; NOT the headline (configs[2] is 65 536 envs)"}


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: a PMC summary under profiles/ is only valid for the code it
    was measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "gym-solarpvder-environment_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")) and f != "pvder_agent.hip":      # (the agent kernel is not on the step path)
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "pvder_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, n_env):
    """HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this very command, corrected as MI355X_MICROARCH.md prescribes; PMC counters
    cannot be read from inside the timed process).  Only a summary stamped with the hash of the
    kernel sources it was measured on counts: anything else -- stale code, other kernel, other size
    -- reports null."""
    import glob
    sha = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if (d.get("kernel") == kernel and int(d.get("envs", -1)) == n_env
                    and d.get("kernel_src_sha") == sha):
                return float(d["traffic_bytes_per_launch"]), os.path.basename(f)
        except Exception:
            continue
    return None, None


def finite(x):
    """JSON has no NaN/Infinity: unusable numbers become null."""
    return x if (x is not None and x == x and abs(x) != float("inf")) else None


def sample_period(time_every: int, steps: int) -> int:
    """Every how many launches a step kernel is bracketed with HIP events: the requested period,
    clamped so that even the shortest run records >= 4 samples (an event pair costs ~9 us of stream
    time, so short runs are not bracketed on every launch either)."""
    if time_every <= 0:
        return 0
    return max(1, min(time_every, steps // 4 if steps < 256 else steps // 8))


# ---------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    from gym_PVDER_amd.distributed import check_payload
    check_payload(args.payload, args.gather_every)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    warnings = []
    if local_rank == 0 and os.environ.get("PVDER_BENCH_CHILD") != "1":   # (the self-launcher built it already)
        from gym_PVDER_amd import _native        # before anything touches the GPU (hipcc is a child process)
        _native.build_native()       # no-op when the in-tree library is up to date (never a fallback: same HIP code)
    dist = None
    if world > 1 or os.environ.get("PVDER_BENCH_FORCE_DIST") == "1":   # (forced: exercises RCCL at world 1)
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if args.gpus != world:
        warnings.append("--gpus %d but WORLD_SIZE %d: measured %d rank(s)" % (args.gpus, world, world))
        if rank == 0:
            print("bench.py: warning: " + warnings[-1], file=sys.stderr)
    # CPU legs first: they start child processes, which must happen before this process touches the GPU
    cpu = {}
    if world == 1 and dist is None and args.cpu_seconds > 0:
        try:
            cpu["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            cpu["cpu_baseline_all_cores"] = cpu_baseline_all_cores(min(args.cpu_seconds, 8.0))
        except Exception as exc:      # a reported side number: never lose the GPU line over it
            cpu.setdefault("cpu_baseline", {"value": None})
            cpu["cpu_baseline"]["error"] = repr(exc)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    ranks_seen = 1
    if dist is not None:
        dist.barrier()               # rank 0 has (re)built the library, see above
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)        # through RCCL: how many GPUs really take part
        ranks_seen = int(seen.item())
    from gym_PVDER_amd.vec_env import PVDERVec
    N = args.envs
    if args.micro is None:
        args.micro = DEFAULT_MICRO[args.integrator]
    env = PVDERVec(N, device=dev, n_sim_time_steps_per_env_step=args.n_sub, micro_steps=args.micro,
                   micro_coarse=args.micro_coarse, graded=bool(args.graded), integrator=args.integrator, seed=2024,
                   env_id_base=rank * N, events_spec=events_on_spec(), auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    total = args.warmup + args.steps
    # (the pre-roll cycles through the pool too: a pool of a few dozen rows, repeated, is a RAMP -- every env drifts by the pattern's net
    #  action count per cycle, and a 20-step window behind it was measured with the duty-cycle-limit kernel following every launch)
    pool = max(total, 4096) if (args.sustained_seconds > 0 or args.preroll_steps > 0) else total + args.steps   # the sustained leg cycles through the pool
    #                                                                                  (+ steps: the window graph's warm twin)
    actions = torch.randint(0, 5, (pool, N), generator=gen, device=dev, dtype=torch.uint8)
    # ONE collective per step (RCCL over xGMI): all-gather of the [13, N_local] output block.  It is
    # double-buffered and asynchronous: the gather of step t runs on its own stream while the step
    # kernel of t+1 computes (the rollout does not consume the collated block before the next
    # launch); every gather has completed inside the timed region (waited for before the final sync).
    #   --collective rccl  (default): the env alternates between two output blocks and the gather is
    #       issued straight into RCCL (gym_PVDER_amd.distributed.RcclAllGather) -- no staging copy, no
    #       c10d bookkeeping on the host;
    #   --collective torch: torch.distributed.all_gather_into_tensor(async_op=True) on a staged copy
    #       (also the fallback if the direct communicator cannot be set up).
    collective = "none"
    direct = native = None
    if dist is not None:
        gathered = [torch.empty((world * 13, N), dtype=env.out.dtype, device=dev) for _ in range(2)]
        pending = [None, None]
        collective = "torch.distributed.all_gather_into_tensor (RCCL), async, staged copy"
        if args.collective == "rccl":
            try:
                from gym_PVDER_amd.distributed import RcclAllGather
                direct = RcclAllGather(dev)
                native = direct.native(env)          # step launch + gather enqueued from C (one host call per rollout)
                native.payload, native.gather_every = args.payload, args.gather_every
                collective = ("ncclAllGather (RCCL) issued by the C side right behind each step launch "
                              "(pvder_plan_rollout_gather), own stream, two output blocks")
            except Exception as exc:          # keep the run: fall back to the c10d path on every rank
                direct = native = None
                print("bench.py: direct RCCL all-gather unavailable (%r), using torch.distributed" % (exc,), file=sys.stderr)
            ok = torch.tensor([1 if direct is not None else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # all ranks take the same path
            if int(ok.item()) == 1:
                # one trial step + gather through the C path, checked against the rank's own block, before
                # the run depends on it (a failure here still falls back to c10d on every rank)
                try:
                    native.rollout(actions[:1])
                    native.drain()
                    torch.cuda.synchronize(dev)
                    mine = native.last()[rank * 13:(rank + 1) * 13]
                    good = bool(torch.equal(mine, env.out))
                except Exception as exc:
                    good = False
                    print("bench.py: C-side step+gather failed its trial (%r), using torch.distributed" % (exc,), file=sys.stderr)
                ok = torch.tensor([1 if good else 0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if native is not None:
                    env.out = native.out[0]
                direct = native = None
                collective = "torch.distributed.all_gather_into_tensor (RCCL), async, staged copy"
        if direct is None:
            stage = [torch.empty_like(env.out) for _ in range(2)]

    def one_step(i, gather=True):
        b = i & 1
        a = actions[i % pool]
        if dist is None or not gather:
            env.step(a, return_tensors=False)           # results land in env.out ([13, N])
            return
        if native is not None:
            native.rollout(actions[i % pool:i % pool + 1], sync_each=bool(args.sync_gather))
            return
        env.step(a, return_tensors=False)
        if pending[b] is not None:
            pending[b].wait()                        # stream dependency only, no host block
        stage[b].copy_(env.out)                      # 3.4 MB device copy: frees env.out for step t+1
        pending[b] = dist.all_gather_into_tensor(gathered[b], stage[b], async_op=True)
        if args.sync_gather:
            pending[b].wait()

    def drain():
        if native is not None:
            native.drain()
        elif dist is not None:
            for b in (0, 1):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

    def timed(first, count, every, gather=True, graph=None):
        """`count` steps starting at action row `first`, bracketed by barrier + synchronize on both
        sides; returns (max-over-ranks seconds, host enqueue seconds, per-launch event samples in ms,
        ms per step by ONE HIP-event pair around the whole region on the launch stream)."""
        env.kernel_timing(every > 0, max(1, every))
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        r0.record()
        if graph is not None:
            graph.replay()        # the same `count` per-step launches, captured before the window (see below)
        elif dist is None or not gather:
            # open-loop rollout: the launches of the stretch between two event-bracketed ones are enqueued
            # by ONE host call (env.rollout -> pvder_plan_rollout: the same per-step launch, K times)
            i, stop = first, first + count
            while i < stop:
                nxt = stop if every <= 0 else min(stop, i + every - 1 - (env._timing_k % every if env._timing else 0))
                j0, j1 = i % pool, i % pool + (nxt - i)
                if nxt > i and j1 <= pool:
                    env.rollout(actions[j0:j1], fused=False)       # (the headline: ONE launch per env step)
                    env._timing_k += nxt - i
                    i = nxt
                elif nxt > i:
                    env.rollout(actions[j0:pool], fused=False)
                    env._timing_k += pool - j0
                    i += pool - j0
                    continue
                if i < stop:
                    one_step(i, gather)           # the bracketed launch
                    i += 1
        elif native is not None:
            # step + gather pairs enqueued by the C side, the stretch between two bracketed launches per call
            i, stop = first, first + count
            while i < stop:
                j0 = i % pool
                n = min(stop - i, pool - j0)
                native.rollout(actions[j0:j0 + n], sync_each=bool(args.sync_gather))
                i += n
            drain()
        else:
            for i in range(first, first + count):
                one_step(i, gather)
            drain()
        r1.record()
        host = time.perf_counter() - t0        # host side of the loop (launches are asynchronous)
        # (rounds 2-4 polled the closing event ahead of synchronize(); on ROCm 7.2 the blocking wait returns 5 us EARLIER than
        #  the first successful query: tools/window_overhead.py -- 20-step graph 190 us with the poll, 185 without, 177 without
        #  the event pair, which stays: it is the roofline's region measurement)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        k = env.kernel_times_ms()
        env.kernel_timing(False)
        return el, host, k, r0.elapsed_time(r1) / count

    # pre-roll (reported as config.preroll_steps): more untimed steps of the same workload ahead of the warm-up
    preroll = max(0, args.preroll_steps) if args.integrator in ("etdlin", "etdrk4") else 0
    if preroll and dist is None:
        done_pre = 0
        while done_pre < preroll:
            n = min(preroll - done_pre, pool)
            env.rollout(actions[:n], fused=False)
            done_pre += n
    elif preroll:
        for i in range(min(preroll, 2000)):
            one_step(i)
        preroll = min(preroll, 2000)
    for i in range(args.warmup):
        one_step(i)
    drain()
    # Per-launch HIP-event brackets cost ~2 us of stream time apiece and split the rollout into several
    # host calls: a LONG window carries them (every 64th launch); a short one (the driver's 20 steps are
    # 0.25 ms of GPU time) is bracketed as a whole only, and individually bracketed launches follow right
    # BEHIND the window, outside the timed region -- `value` stays clean, the samples are still live.
    short = args.steps < 256
    every = 0 if short else sample_period(args.time_every, args.steps)
    n_post = 0 if (args.time_every <= 0 or not short) else 8
    env.reserve_timing_events((args.steps // max(every, 1) + 2 if every > 0 else 0) + n_post)   # created and warmed outside the window
    # A short window is launch-bound on the HOST: the first launches into an idle queue cost ~10 us apiece
    # (2.7 us once the queue is backed up), more than the 8 us kernel they start.  Its K per-step launches --
    # the very same kernels, action rows and order -- are therefore captured into ONE HIP graph before the
    # window and replayed inside it (the C entry points are asynchronous and allocation-free:
    # tests/test_gpu_parity.py::test_step_is_graph_capturable).  Long windows stay eager (GPU-bound).
    window_graph = None
    graph_warm_steps = 0
    if short and dist is None and args.graph_window and args.warmup + 2 * args.steps <= pool:
        try:
            # TWO graphs of K per-step launches each, captured back to back (the plan's prefetch hint advances with every
            # launch it enqueues -- captured or not -- so each graph carries the clock its replay will find): the first is
            # replayed UNTIMED (K more warm-up steps, reported as config.graph_warm_steps: the first launch of a graph pays
            # for its upload), the second is the timed window
            env.graph_hints(True)         # (opt in: these two graphs are replayed exactly where they were captured)
            warm_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(warm_graph):
                env.rollout(actions[args.warmup:args.warmup + args.steps], fused=False)
            window_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(window_graph):
                env.rollout(actions[args.warmup + args.steps:args.warmup + 2 * args.steps], fused=False)
            warm_graph.replay()
            torch.cuda.synchronize(dev)
            graph_warm_steps = args.steps
            env.graph_hints(False)
        except Exception as exc:
            window_graph = None
            warnings.append("HIP-graph capture of the window failed (%r): eager launches" % (exc,))
    elapsed, host_enqueue, kern_ms, region_ms = timed(args.warmup + graph_warm_steps, args.steps, every, graph=window_graph)
    total += graph_warm_steps
    st = env.step_stats()
    # was the duty-cycle-limit kernel following the step launches of this window?  (It does for 256 launches behind a sighting of an env
    # beyond 98.75 % of the limit -- once in ~15 000 steps of a random-action batch -- at +5-6 us per launch: a short window is in or out.)
    try:
        hyb_following = bool(env._lib.pvder_plan_hybrid_active(env._plan)) if getattr(env, "_plan", None) else False
    except Exception:
        hyb_following = None
    post_note = ""
    if n_post and dist is None:
        env.kernel_timing(True, 1)
        for i in range(total, total + n_post):
            one_step(i)
        kern_ms = env.kernel_times_ms()
        env.kernel_timing(False)
        post_note = " right behind the timed window"
        total += n_post

    # sustained leg: the same loop for >= sustained_seconds (a 20-step window is 0.5 ms of GPU time)
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = int(min(400000, max(args.steps, args.sustained_seconds / max(elapsed / args.steps, 1e-7))))
        el_s, host_s, k_s, region_s = timed(total, n_sus, sample_period(max(args.time_every, 1), n_sus))
        sustained = {"value": world * N * n_sus / el_s, "unit": "env-steps/s", "steps": n_sus, "seconds": el_s,
                     "ms_per_step": 1e3 * el_s / n_sus, "host_enqueue_ms_per_step": 1e3 * host_s / n_sus,
                     "kernel_avg_ms": finite(sum(k_s) / len(k_s)) if k_s else None, "kernel_samples": len(k_s),
                     "region_ms_per_step": region_s}
    # separately labelled leg: the same workload as open-loop rollouts with the env steps FUSED into one launch
    # (pvder_plan_rollout_fused: state in registers from step to step, bit-identical results; one call = one
    # 160-step episode) -- what calc_returns-style evaluation sweeps and recorded-action replays run at
    fused_leg = None
    if dist is None and args.integrator == "etdlin" and args.fused_seconds > 0 and env.ns == 23:
        Kf = max(1, min(args.fused_steps, pool))
        blocks = torch.empty((Kf, 13, N), dtype=env.out.dtype, device=dev) if Kf * 13 * N * 4 <= (4 << 30) else None
        env.rollout(actions[:Kf], out=blocks, fused=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        env.rollout(actions[:Kf], out=blocks, fused=True)
        torch.cuda.synchronize(dev)
        reps = int(max(2, min(2000, args.fused_seconds / max(time.perf_counter() - t0, 1e-6))))
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r0.record()
        for r in range(reps):
            j0 = (r * Kf) % max(1, pool - Kf + 1)
            env.rollout(actions[j0:j0 + Kf], out=blocks, fused=True)
        r1.record()
        torch.cuda.synchronize(dev)
        el_f = time.perf_counter() - t0
        per = el_f / (reps * Kf)
        fused_leg = {"value": N * reps * Kf / el_f, "unit": "env-steps/s", "steps_per_launch": Kf, "launches": reps,
                     "seconds": el_f, "ms_per_step": 1e3 * per, "region_ms_per_step": r0.elapsed_time(r1) / (reps * Kf),
                     "per_step_output_blocks": blocks is not None,
                     # what a step of a fused rollout really moves: its 13 output rows, its action byte, the two event
                     # inputs; the 25 state rows + counters are read and written once per LAUNCH
                     "bytes_per_env_step_actual": 13 * 4 + 1 + 8 + (2 * 25 * 4 + 5 * 4) / float(Kf),
                     "hbm_frac_actual": (13 * 4 + 1 + 8 + (2 * 25 * 4 + 5 * 4) / float(Kf)) * N / per / HBM_PEAK,
                     "nominal_frac_on_269B_per_env_step": BYTES_PER_ENV_STEP["f32"] * N / per / HBM_PEAK,
                     "nominal_frac_note": "269 B is what ONE LAUNCH PER STEP moves per env step; a fused rollout keeps the state on "
                                          "the chip, so this figure compares rates, it is not HBM utilisation (hbm_frac_actual is)",
                     "kernel": "step_etdlin_rollout_kernel<3> (+ step_etdlin_finish_kernel<3> for the step that ends an episode)",
                     "note": "same envs, events and random actions as the headline; results bit-identical to one launch per "
                             "step (tests/test_gpu_etdlin.py::test_fused_rollout_equals_single_steps); NOT the headline"}
    # ... and SERVED (pvder_plan_rollout_served, round 6): the same register-resident loop for a producer that does not have its
    # actions ahead -- every env step waits for its row to be posted; "ahead": posted before the launch, "depth_8": row k posted by
    # the host when every workgroup has reported step k - 8 done (the host in the loop, eight steps behind the env)
    served_leg = None
    if fused_leg is not None and hasattr(env, "rollout_served"):
        try:
            Ks = max(1, min(4 * args.fused_steps, pool))
            served_leg = {"steps_per_call": Ks, "unit": "env-steps/s"}
            for _ in range(4):       # (the duty-cycle-limit kernel follows 256 launches behind a sighting of an env near the limit --
                if not env._lib.pvder_plan_hybrid_active(env._plan or env._make_plan()):      # once in ~15 000 random-action steps)
                    break
                env.rollout(actions[:min(300, pool)], fused=False)
            for name, depth in (("ahead", None), ("depth_8", 8)):
                env.rollout_served(actions[:Ks], depth=depth)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                env.rollout_served(actions[:Ks], depth=depth)
                el_v = time.perf_counter() - t0
                served_leg[name] = {"value": N * Ks / el_v, "us_per_step": 1e6 * el_v / Ks}
            served_leg["kernel"] = "step_etdlin_served_kernel (+ step_etdlin_finish_kernel<3> for the step that ends an episode)"
            served_leg["note"] = ("bit-identical to one launch per step (tests/test_gpu_fullsize.py::test_served_rollout_is_bit_identical_"
                                  "and_waits_for_its_rows); NOT the headline: step() stays one launch per env step -- a producer one "
                                  "step behind the env pays the PCIe round trip per step (profiles/r06_ab_served_rollout.txt)")
        except (ValueError, RuntimeError, TimeoutError) as ex:      # (e.g. the duty-cycle-limit kernel is following the launches)
            served_leg = {"skipped": "%s: %s" % (type(ex).__name__, ex)}
    compute_only = None
    payload_legs = {}
    if dist is not None:        # the same steps without the collective: what the gather costs
        el_c, host_c, k_c, region_c = timed(total, args.steps, every, gather=False)
        compute_only = {"value": world * N * args.steps / el_c, "unit": "env-steps/s", "steps": args.steps,
                        "ms_per_step": 1e3 * el_c / args.steps, "host_enqueue_ms_per_step": 1e3 * host_c / args.steps,
                        "kernel_avg_ms": finite(min(sum(k_c) / len(k_c), region_c)) if k_c else finite(region_c),
                        "kernel_samples": len(k_c)}
        # ... and the same steps once per payload the consumer can choose (ONE run tells the whole story):
        # `value` stays the payload of the command line (default "full": what BASELINE configs[3] names)
        if native is not None:
            main_payload = args.payload
            for pl in ("full", "obs8", "reward_done"):
                if pl == main_payload:
                    continue
                native.payload = pl
                el_p, host_p, _, _ = timed(total, args.steps, 0)
                payload_legs[pl] = payload_leg(world, N, args.steps, el_p, pl, args.gather_every, host_p)
            native.payload = main_payload

    if rank == 0:
        prec = "f64" if args.integrator == "rk45" else "f32"
        value = world * N * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        # kernel duration: per-launch HIP-event samples of THIS window; a window too short to hold any
        # falls back to the sustained leg's samples (same process, same config) and says so
        src = ("hip events around %d launches%s" % (len(kern_ms), post_note)) if post_note else \
            "hip events around %d of the %d timed launches" % (len(kern_ms), args.steps)
        if not kern_ms and compute_only and compute_only["kernel_avg_ms"]:
            # with the collective on its own stream the step kernel shares the GPU with RCCL's kernels: its
            # duration is taken from the compute-only leg of this very process (same steps, no gather)
            kern_ms = [compute_only["kernel_avg_ms"]]
            src = "hip events in the compute-only leg (%d bracketed launches + the region pair)" % compute_only["kernel_samples"]
        elif not kern_ms and sustained and sustained["kernel_avg_ms"]:
            kern_ms = [sustained["kernel_avg_ms"]]
            src = "hip events in the sustained leg (%d samples)" % sustained["kernel_samples"]
        avg_kern_s = (sum(kern_ms) / len(kern_ms)) * 1e-3 if kern_ms else None
        # Both measurements bound the kernel's duration from ABOVE: a bracketed launch includes the two event
        # packets (~2 us next to an 11 us kernel), the region bracket includes the gaps between launches.
        # While the GPU is the bottleneck (no collective in the way) the smaller one is the better estimate --
        # rocprofv3's average (profiles/) is the cross-check.
        if region_ms and dist is None and (avg_kern_s is None or region_ms * 1e-3 < avg_kern_s):
            src = ("ONE hip-event pair around the %d back-to-back timed launches (%.4f ms per launch; the %d "
                   "individually bracketed launches average %s ms incl. their event packets)"
                   % (args.steps, region_ms, len(kern_ms), "%.4f" % (avg_kern_s * 1e3) if avg_kern_s else "n/a"))
            avg_kern_s = region_ms * 1e-3
        bytes_step = BYTES_PER_ENV_STEP[prec]
        achieved = bytes_step * N / avg_kern_s if avg_kern_s else None
        if avg_kern_s and world == 1 and avg_kern_s * 1e3 > ms_per_step * 1.15:
            warnings.append("kernel_avg_ms %.4f > 1.15 x ms_per_step %.4f: the event-bracketed launches run slower "
                            "than the un-bracketed step period" % (avg_kern_s * 1e3, ms_per_step))
        fps = FLOP_PER_STEP.get(args.integrator, 0)
        # integrator steps per env step: a sim time step that starts at an input jump (the first of each
        # env step + one event per second) takes 2 steps (graded) or `micro` uniform ones, the others micro_coarse
        if args.integrator == "etdlin":
            # per segment between input jumps: the tick + two equal ETD3RK steps (events on env-step boundaries
            # when n_sub divides 60: one segment per env step)
            segs = 1.0 + (0 if 60 % args.n_sub == 0 else (args.n_sub - 1) / 60.0)
            steps_per_env_step = 2.0 * segs
            # ... and an env step that does not start at an event row change (n_sub / 60 s <= 0.25 s: all but one per second)
            # is ONE Krogstad step without tick
            quiet_frac = 0.0
            if 60 % args.n_sub == 0 and args.n_sub <= 15 and not os.environ.get("PVDER_NO_K4"):
                quiet_frac = 1.0 - args.n_sub / 60.0
                steps_per_env_step = 2.0 * (1.0 - quiet_frac) + 1.0 * quiet_frac
        elif args.integrator == "etdrk4":
            n_fine = 1.0 + (args.n_sub - 1) / 60.0
            steps_per_env_step = n_fine * (2 if env.graded else args.micro) + (args.n_sub - n_fine) * env.micro_coarse \
                if env.micro_coarse > 0 else args.micro * args.n_sub
        else:
            steps_per_env_step = args.micro * args.n_sub
        if args.integrator == "rk45" and avg_kern_s and env.hstep is not None:
            # adaptive: integrator steps per env step estimated from the per-env step sizes the controller settled
            # on (stability-limited, nearly constant): 7 stages x (right-hand side ~300 flop + 23-state stage sums)
            h = env.hstep[env.hstep > 0]
            if h.numel():
                steps_per_env_step = float((args.n_sub / 60.0) / float(h.double().mean().item()))
                fps = 7 * 300 + 23 * 60
        flops = fps * steps_per_env_step * N / avg_kern_s if (fps and avg_kern_s) else None
        if flops and args.integrator == "etdlin":
            flops = ((1.0 - quiet_frac) * (2.0 * fps + FLOP_PER_TICK) * segs + quiet_frac * FLOP_PER_K4) * N / avg_kern_s
        traffic, traffic_src = pmc_traffic(env.kernel_name(), N)
        valu_peak = VALU_PEAK_F64 if prec == "f64" else VALU_PEAK_F32
        out = {
            "metric": "env-steps/sec (batched PVDER-v0)", "value": value, "unit": "env-steps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]" if (N, args.n_sub) == (65536, 15) else "variant of BASELINE configs[2]") +
                                   ": %d parallel PVDER-v0 envs per GPU, n=%d sim time steps "
                                   "per env step, voltage+insolation events on, %s, random actions, auto-reset"
                                   % (N, args.n_sub, {"etdlin": "fp32 ETD3RK on the full nonlinear model, f = L y + N(y) (exponential RK: the 11x11 Jacobian at the "
                                                                "operating point in the exponential, N in the stages; matrix-vector products on the matrix cores), "
                                                                "per segment between input jumps: exponential-Euler tick of 2/(60*%d) s + two "
                                                                "equal ETD3RK steps; an env step that does not start at an event row change "
                                                                "(three in four): ONE Krogstad ETDRK4 step, no tick" % args.micro,
                                                      "etdrk4": ("fp32 ETDRK4 (exponential RK, stiff order 4, 5 stages), sim time step behind an input jump: "
                                                                 + ("first step 1/(60*%d) s + one over the remainder" if env.graded else "%d uniform steps")
                                                                 + ", every other one: %d step(s)") % (args.micro, env.micro_coarse or args.micro),
                                                      "rk4": "fp32 fixed-step classic RK4, %d micro-steps per 1/60 s" % args.micro,
                                                      "rk45": "fp64 adaptive RK45"}[args.integrator]),
                       "envs_per_gpu": N, "n_sub": args.n_sub, "micro_steps": args.micro, "preroll_steps": preroll,
                       "graph_warm_steps": graph_warm_steps,
                       "launch": ("one HIP graph of the window's %d per-step launches, captured before the window" % args.steps)
                       if window_graph is not None else "eager: one launch per env step, enqueued from C",
                       "micro_coarse": env.micro_coarse, "graded": bool(env.graded),
                       "integrator": args.integrator,
                       "collective": ("all-gather of the [13, N_local] fp32 block per step: %s; %s"
                                      % (collective, "synchronous" if args.sync_gather else "overlapped with the next step"))
                       if dist is not None else "none"},
            "roofline": {"bound": "hbm", "achieved": finite(achieved / 1e9) if achieved else None, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": finite(achieved / HBM_PEAK) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": env.kernel_name(),
                         "kernel_avg_ms": finite(avg_kern_s * 1e3) if avg_kern_s else None,
                         "kernel_avg_source": src, "kernel_samples": len(kern_ms),
                         "region_ms_per_step": finite(region_ms),
                         "algorithmic_bytes_per_env_step": bytes_step, "kernel_src_sha": kernel_source_hash()},
            # the binding roof of this path is arithmetic latency, not HBM; fp32
            # vector and fp32 matrix pipes have the same 157.3 TFLOP/s peak (etdlin uses both)
            "valu_roofline": {"bound": ("valu+mfma_" if args.integrator == "etdlin" else "valu_") + prec, "achieved": finite(flops / 1e12), "peak": valu_peak / 1e12,
                              "unit": "TFLOP/s", "frac": finite(flops / valu_peak),
                              "flop_per_integrator_step": fps,
                              "integrator_steps_per_env_step": steps_per_env_step,
                              "instr_per_integrator_step": INSTR_PER_STEP.get(args.integrator),
                              "ns_per_instr_per_wave": (avg_kern_s * 1e9 / (steps_per_env_step * INSTR_PER_STEP[args.integrator])
                                                        if args.integrator in INSTR_PER_STEP else None)} if flops else None,
            "sustained": sustained,
            "fused_rollout": fused_leg,
            "served_rollout": served_leg,
            "ranks_seen": ranks_seen,
            "last_step_stats": st, "duty_cycle_limit_kernel_following": hyb_following,
        }
        if dist is not None:
            blk = 13 * N * 4
            out["compute_only"] = compute_only
            rows = 13.0
            if native is not None:
                rows = payload_rows_per_step(args.payload, args.gather_every)
            elif args.payload != "full":
                warnings.append("--payload %s needs the C-side step+gather path (unavailable in this run: c10d fallback): "
                                "the whole block was collated" % args.payload)
            out["config"]["payload"] = {"mode": args.payload if native is not None else "full", "rows_per_step": rows,
                                        "gather_every": args.gather_every if args.payload == "full_every_k" else None}
            out["xgmi_roofline"] = xgmi_roofline(world, N, rows, elapsed / args.steps)
            # the three legs side by side: no collation / the whole block / (reward, done) -- same process, same steps
            main = {"value": value, "unit": "env-steps/s", "steps": args.steps, "ms_per_step": ms_per_step, "rows_per_step": rows,
                    "xgmi_roofline": out["xgmi_roofline"]}
            legs = dict(payload_legs)
            legs[args.payload if native is not None else "full"] = main
            out["payload_legs"] = {"compute_only": compute_only, **{k: legs[k] for k in sorted(legs)}}
            kc = (compute_only or {}).get("kernel_avg_ms") or (compute_only or {}).get("ms_per_step")
            out["expected_payload_efficiency"] = expected_payload_efficiency(world, N, (kc or 0.0) * 1e-3, args.gather_every)
            out["scaling_note"] = scaling_note(world, value, out["payload_legs"], out["expected_payload_efficiency"])
            out["xgmi_bound"] = {"bytes_received_per_step_per_gpu": (world - 1) * blk,
                                 "link_GBps_one_direction": 76.5,
                                 "min_us_per_step": blk / 76.5e9 * 1e6 if world > 1 else 0.0,
                                 "note": "each peer's block arrives over its own link (fully connected mesh): the "
                                         "gather of one step cannot finish faster than one block over one link"}
        if world == 1 and args.classic_steps > 0 and args.integrator in ("etdrk4", "etdlin"):
            out["classic_rk4"] = classic_rk4_leg(torch, dev, N, args.n_sub, args.classic_steps)
        if world == 1 and dist is None and args.integrator == "etdlin":
            del env, actions                         # (the side legs build their own envs)
            try:
                if args.large_batch > 0:
                    out["large_batch"] = large_batch_leg(torch, dev, args.n_sub, N=args.large_batch)
                if args.single_env_steps > 0:
                    out["single_env"] = single_env_leg(args.single_env_steps)
                if args.side_legs:
                    out["configs1"] = configs1_leg(torch, dev)
                    out["desynchronised"] = desynchronised_leg(torch, dev, N, args.n_sub)
                if args.fixed_policy:
                    out["fixed_policy"] = fixed_policy_leg(torch, dev, N, args.n_sub)
                    out["lane_params"] = lane_params_leg(torch, dev, args.n_sub)
                if args.rk45_steps > 0:
                    out["rk45"] = rk45_leg(torch, dev, N, args.n_sub, args.rk45_steps)
                    out["etdlin64"] = etdlin64_leg(torch, dev, N, args.n_sub)
                    out["etdlin64"]["speedup_over_rk45"] = out["etdlin64"]["value"] / out["rk45"]["value"]
            except Exception as exc:                 # side legs: never lose the headline over them
                warnings.append("side leg failed: %r" % (exc,))
        out.update(cpu)
        if warnings:
            out["warnings"] = warnings
        print(json.dumps(out, allow_nan=False))
    if dist is not None:
        if native is not None:
            native.close()
        if direct is not None:
            direct.close()
        dist.destroy_process_group()


def main():
    args = parse()
    in_group = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.selftest_launcher:
        if in_group:
            selftest_rank()
            return 0
        return launch_ranks(max(1, args.gpus), sys.argv[1:])
    if not in_group and args.gpus > 1:
        # outside torchrun: start the ranks ourselves -- BEFORE this process makes any GPU call
        have = visible_gpu_count()                  # from the environment / KFD sysfs: no HIP call in this process
        if 0 <= have < args.gpus:
            print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
            return 2
        from gym_PVDER_amd import _native
        _native.build_native()                      # once, by the parent (hipcc is a child process)
        return launch_ranks(args.gpus, sys.argv[1:])
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
