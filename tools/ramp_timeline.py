#!/usr/bin/env python3
"""GPU box: in-kernel timeline of the SECOND pass of step_etdlin_kernel (the round-3 body behind the lean pass: careful pass,
symmetrical components, 23-state tail) along a fixed-action episode -- where the 100-180 us of a step in the duty-cycle
regime go.  Builds a -DPVDER_TRACE -DPVDER_TRACE_COLD copy of the library into gpurun_out/.
    python tools/ramp_timeline.py [action] [step ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gym_PVDER_amd import _native
act = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = [int(x) for x in sys.argv[2:]] or [60, 61, 63, 84, 100, 130]
lib = os.path.join(ROOT, "gpurun_out", "libpvder_hip_trace_cold.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
srcs = [os.path.join(_native.CSRC, f) for f in sorted(os.listdir(_native.CSRC)) if f.endswith(".hip")]
subprocess.check_call([_native.hipcc_path()] + _native.HIPCC_FLAGS + ["-fPIC", "-shared", "-DPVDER_TRACE", "-DPVDER_TRACE_COLD",
                       "-I" + _native.INCLUDE, "-I" + _native.CSRC, "-o", lib] + srcs)
_native.LIB_PATH = lib
L = _native.lib()
L.pvder_debug_trace.restype = C.c_int
L.pvder_debug_trace.argtypes = [C.c_void_p, C.c_int]
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda", 0)
N = 65536
env = PVDERVec(N, device=dev, events_spec=both_events_spec(), seed=2024, auto_reset=False)
env.reset()
a = torch.full((N,), act, dtype=torch.uint8, device=dev)
buf = np.zeros(4096 * 32, dtype=np.uint64)
names = {0: "entry", 21: "lean pass done", 8: "2nd pass: entry", 9: "2nd pass: standard pass done", 10: "careful pass done", 11: "epilogue done", 12: "23-state tail done (wave 0)",
         13: "tail: operator sets staged (latest wave)", 14: "tail: state loaded (latest wave with lanes)", 15: "tail: sim time step 0 done (latest)", 16: "tail: sim time step 1 done", 17: "tail: sim time step 2 done", 18: "tail: last sim time step done (latest)"}
for t in range(max(steps) + 1):
    if t in steps:
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(a, return_tensors=False); e1.record()
    if t in steps:
        torch.cuda.synchronize()
        assert L.pvder_debug_trace(buf.ctypes.data, buf.size) == 0
        T = buf.reshape(4096, 32)[:256].astype(np.int64)
        t0 = T[:, 0].min()
        print("action %d step %d: launch %.1f us (events)" % (act, t, e0.elapsed_time(e1) * 1e3))
        for s in (0, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 12):
            x = (T[:, s] - t0) * 0.01
            ok = (T[:, s] >= t0) & (x < 1e4)                 # workgroups that stamped this slot in THIS launch
            if ok.any():
                print("   slot %2d %-30s workgroups %3d   min %7.2f  med %7.2f  max %7.2f us" % (s, names[s], int(ok.sum()), x[ok].min(), np.median(x[ok]), x[ok].max()))
        pk = T[:, 5]
        if (pk > 0).any():
            lanes, cy = (pk & 255)[pk > 0], (pk >> 8)[pk > 0]
            for lo, hi in ((1, 2), (3, 8), (9, 32), (33, 64)):
                m = (lanes >= lo) & (lanes <= hi)
                if m.any():
                    print("   slowest wave of a workgroup carried %2d..%2d lanes in %3d workgroups: loop %.0f .. %.0f kcycles (median %.0f)" % (lo, hi, int(m.sum()), cy[m].min() / 1e3, cy[m].max() / 1e3, np.median(cy[m]) / 1e3))
        cyc, wall = T[:, 30].astype(np.float64), T[:, 31].astype(np.float64)
        okc = wall > 0
        if okc.any():
            f = cyc[okc] / (wall[okc] * 10.0)               # cycles per ns = GHz (wall clock: 100 MHz)
            print("   tail loop (longest so far per workgroup): %.1f .. %.1f us, shader clock %.2f .. %.2f GHz (median %.2f)" % (
                wall[okc].min() * 0.01, wall[okc].max() * 0.01, f.min(), f.max(), np.median(f)))
