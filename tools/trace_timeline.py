#!/usr/bin/env python
"""In-kernel timeline of step_etdrk4_bal_kernel (debug tool, GPU box only).

Builds a second copy of the library with -DPVDER_TRACE into gpurun_out/ (the shipped library has no
trace code), runs the default bench workload and prints, per trace slot, min / median / max over
workgroups of the 100 MHz wall clock relative to the earliest workgroup's kernel entry, next to the
HIP-event duration of the same launch.

  slot 0 entry | 1 coefficients staged in LDS | 2 state loads landed | 3+s start of sim time step s |
  20 integration done | 22 outputs and reward computed | 23 stores issued | 21 statistics reduced, end
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402
from gym_PVDER_amd import _native  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libpvder_hip_trace.so")
    srcs = [os.path.join(_native.CSRC, f) for f in sorted(os.listdir(_native.CSRC)) if f.endswith(".hip")]
    subprocess.check_call([_native.hipcc_path()] + _native.HIPCC_FLAGS + ["-fPIC", "-shared", "-DPVDER_TRACE",
                           "-I" + _native.INCLUDE, "-I" + _native.CSRC, "-o", lib] + srcs)
    _native.LIB_PATH = lib
    L = _native.lib()
    L.pvder_debug_trace.restype = C.c_int
    L.pvder_debug_trace.argtypes = [C.c_void_p, C.c_int]
    from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec
    spec = default_events_spec()
    spec["insolation"]["ENABLE"] = True
    dev = torch.device("cuda", 0)
    env = PVDERVec(N, device=dev, events_spec=spec, seed=2024, auto_reset=True)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    acts = torch.randint(0, 5, (400, N), generator=gen, device=dev, dtype=torch.uint8)
    for i in range(300):
        env.step(acts[i], return_tensors=False)
    torch.cuda.synchronize()
    nblk = min(4096, (N + 255) // 256)
    buf = np.zeros(4096 * 32, dtype=np.uint64)
    rows, evs = [], []
    for i in range(300, 340):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step(acts[i], return_tensors=False)
        e1.record()
        torch.cuda.synchronize()
        rc = L.pvder_debug_trace(buf.ctypes.data, buf.size)
        assert rc == 0, rc
        t = buf.reshape(4096, 32)[:nblk].astype(np.int64)
        t0 = t[:, 0].min()
        rows.append((t - t0) * 0.01)      # us
        evs.append(e0.elapsed_time(e1) * 1e3)
    if os.environ.get("PVDER_TRACE_BURST"):
        # steady state: a burst of un-synchronised launches, the trace of the LAST one only
        rows, evs = [], []
        for rep in range(10):
            for i in range(340, 400):
                env.step(acts[i], return_tensors=False)
            torch.cuda.synchronize()
            rc = L.pvder_debug_trace(buf.ctypes.data, buf.size)
            assert rc == 0, rc
            t = buf.reshape(4096, 32)[:nblk].astype(np.int64)
            rows.append((t - t[:, 0].min()) * 0.01)
            evs.append(float("nan"))
    if os.environ.get("PVDER_TRACE_ROLLOUT"):
        # fused rollout (pvder_plan_rollout_fused): timeline of the FOURTH step of the launch
        # slot 0 loop top | 1 next step's loads issued, state parked | 3.. integrator iterations |
        # 20 integration done | 2 workgroup agreed to go on | 22 outputs + reward | 21 output rows stored
        K = int(os.environ["PVDER_TRACE_ROLLOUT"])
        rows, evs = [], []
        for rep in range(10):
            env.rollout(acts[40:40 + K], fused=True)
            torch.cuda.synchronize()
            rc = L.pvder_debug_trace(buf.ctypes.data, buf.size)
            assert rc == 0, rc
            t = buf.reshape(4096, 32)[:nblk].astype(np.int64)
            rows.append((t - t[:, 0:1]) * 0.01)       # per workgroup, relative to ITS loop top
            evs.append(float("nan"))
    R = np.stack(rows)                     # [launch, block, slot]
    print("N=%d blocks=%d   event-timed step (both launches) median %.1f us" % (N, nblk, float(np.median(evs))))
    names = {0: "entry", 1: "lds staged", 2: "state loaded", 20: "integration done", 22: "outputs+reward done", 23: "stores issued", 21: "epilogue issued"}
    names.update({24: "it1: F1", 25: "it1: M21/31/41 F1", 26: "it1: F2", 27: "it1: M32/42 F2", 28: "it1: F3", 29: "it1: M43 F3"})
    names.update({18: "kernel arguments in", 19: "coefficient regs in"})
    for s in [0, 18, 19, 1, 2] + list(range(3, 18)) + [24, 25, 26, 27, 28, 29, 20, 22, 23, 21]:
        x = R[:, :, s]
        print("slot %2d %-18s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (
            s, names.get(s, "sub-step %d" % (s - 3)), np.median(x.min(1)), np.median(np.median(x, 1)),
            np.median(np.percentile(x, 90, axis=1)), np.median(x.max(1))))
    if os.environ.get("PVDER_TRACE_DUMP"):
        x = R[-1, :, 0]
        print("entry time by workgroup of the last launch (us), 16 per line:")
        for b in range(0, nblk, 16):
            print("  %4d: " % b + " ".join("%5.2f" % v for v in x[b:b + 16]))
    # per-block durations
    d = R[:, :, 21] - R[:, :, 0]
    print("per-block entry->end: med %.2f  max %.2f us;   entry spread (max slot0) %.2f us" % (
        np.median(d), np.median(d.max(1)), np.median(R[:, :, 0].max(1))))


if __name__ == "__main__":
    main()
