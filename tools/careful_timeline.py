#!/usr/bin/env python3
"""GPU box: in-kernel timeline (100 MHz wall clock) AND shader-clock cycles of the careful pass while every STRIDE-th env of
65 536 sits beyond the current limit (tools/sparse_careful_envs.py): one careful lane in ONE workgroup against one in EVERY
workgroup -- same instructions per wave, twice the time: where does it go (clock? memory? start skew?).
    python tools/careful_timeline.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gym_PVDER_amd import _native
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
buf = np.zeros(4096 * 32, dtype=np.uint64)
names = {0: "entry", 8: "2nd pass: entry", 9: "2nd pass: standard pass done", 10: "careful pass done", 11: "epilogue done"}
a = torch.zeros((N,), dtype=torch.uint8, device=dev)
for stride in (65536, 256, 64, 1):
    env = PVDERVec(N, device=dev, seed=2, events_spec=both_events_spec(), auto_reset=False)
    env.reset()
    env.fstate[23, ::stride] = 1.1
    for t in range(60):
        env.step(a, return_tensors=False)
    for rep in range(3):
        torch.cuda.synchronize()
        buf[:] = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(a, return_tensors=False); e1.record()
        torch.cuda.synchronize()
        assert L.pvder_debug_trace(buf.ctypes.data, buf.size) == 0
        T = buf.reshape(4096, 32)[:256].astype(np.int64)
        t0 = T[:, 0].min()
        print("every %d-th env careful, rep %d: launch %.1f us (events)" % (stride, rep, e0.elapsed_time(e1) * 1e3))
        for s in (0, 8, 9, 10, 11):
            x = (T[:, s] - t0) * 0.01
            ok = (T[:, s] >= t0) & (x < 1e4)
            if ok.any():
                print("   slot %2d %-30s workgroups %3d   min %7.2f  med %7.2f  max %7.2f us" % (s, names[s], int(ok.sum()), x[ok].min(), np.median(x[ok]), x[ok].max()))
        ok = (T[:, 8] >= t0) & (T[:, 10] > T[:, 8]) & (T[:, 27] > T[:, 26])
        if ok.any():
            wall = (T[ok, 10] - T[ok, 8]) * 10.0            # ns
            cyc = (T[ok, 27] - T[ok, 26]).astype(np.float64)
            print("   2nd pass entry -> careful done: %.1f .. %.1f us (median %.1f); %.0f .. %.0f kcycles (median %.0f); shader clock %.2f .. %.2f GHz (median %.2f)" % (
                wall.min() * 1e-3, wall.max() * 1e-3, np.median(wall) * 1e-3, cyc.min() / 1e3, cyc.max() / 1e3, np.median(cyc) / 1e3,
                (cyc / wall).min(), (cyc / wall).max(), np.median(cyc / wall)))
