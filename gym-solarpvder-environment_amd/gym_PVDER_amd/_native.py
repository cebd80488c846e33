"""ctypes binding of the C ABI in include/pvder_hip.h (libpvder_hip.so, built in-tree by hipcc).

There is NO fallback: if the shared library is missing or does not export every symbol the header
declares, importing the env's compute path raises.  PyTorch is only the owner of device memory
and streams; pointers cross the boundary as integers (``tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)                       # gym-solarpvder-environment_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpvder_hip.so")

EINVAL = -10001
ETIMEOUT = -10003             # include/pvder_hip.h PVDER_ETIMEOUT: the device never wrote the host mirror's sequence word
ABI_VERSION = 18

# every symbol include/pvder_hip.h declares (tests check the library exports exactly these)
SYMBOLS = ["pvder_version", "pvder_par_count", "pvder_block_size", "pvder_reset_masked_f32",
           "pvder_reset_masked_f64", "pvder_step_etdrk4_f32", "pvder_step_rk4_f32", "pvder_step_rk45_f64",
           "pvder_etdrk4_coef_count", "pvder_etdrk4_coefficients", "pvder_lin_set_floats", "pvder_lin_max_sets",
           "pvder_step_etdlin_f32", "pvder_plan_create", "pvder_plan_destroy", "pvder_plan_step",
           "pvder_plan_rollout", "pvder_gather_create", "pvder_gather_destroy", "pvder_gather_drain",
           "pvder_plan_rollout_gather", "pvder_plan_rollout_fused", "pvder_plan_rollout_gather_rows", "pvder_plan_hint",
           "pvder_fold_lane_params", "pvder_plan_lane_params", "pvder_reset_masked_lanes_f32", "pvder_fetch_env", "pvder_plan_step_fetch",
           "pvder_plan_step_mirror", "pvder_plan_graph_hints", "pvder_plan_kernel_name", "pvder_plan_hybrid_active",
           "pvder_complete_event_tables", "pvder_served_alloc", "pvder_served_free", "pvder_plan_rollout_served"]
MIRROR_WORDS = 18       # PVDER_MIRROR_WORDS (pvder_plan_step_mirror): 13 output rows + 4 counters + the sequence word, doubles
LPF_ROWS = 11          # enum PVDER_LPF_ROWS: 7 folded constants + the env's own steady state (i_a, m_a)
AGENT_SYMBOLS = ["pvder_qagent_step_f32", "pvder_plan_rollout_fused_agent"]            # include/pvder_agent.h

INTEGRATOR_IDS = {"etdlin": 0, "etdrk4": 1, "rk4": 2, "rk45": 3, "etdlin64": 4}       # enum pvder_integrator
LIN_MAX_SETS = 12


class StepCfg(C.Structure):
    """``struct pvder_step_cfg`` (include/pvder_hip.h)."""
    _fields_ = [("n_sub", C.c_int32), ("micro", C.c_int32), ("max_ksub", C.c_int32),
                ("max_steps", C.c_int32), ("goal", C.c_int32), ("discrete", C.c_int32),
                ("ev_s0", C.c_int32), ("ev_ds", C.c_int32), ("ev_count", C.c_int32),
                ("reserved", C.c_int32), ("del_qref", C.c_double), ("del_vdcref", C.c_double),
                ("inv_max_sim_time", C.c_double), ("p_target", C.c_double), ("q_target", C.c_double),
                ("rtol", C.c_double), ("atol", C.c_double), ("micro_coarse", C.c_int32), ("graded", C.c_int32),
                ("extra_rewards", C.c_int32), ("off_range", C.c_int32)]


class EventSpec(C.Structure):
    """``struct pvder_event_spec`` (include/pvder_hip.h)."""
    _fields_ = [("ev_count", C.c_int32), ("enable_voltage", C.c_int32), ("enable_insol", C.c_int32),
                ("reserved", C.c_int32), ("v_min", C.c_float), ("v_max", C.c_float),
                ("s_min", C.c_float), ("s_max", C.c_float)]


class AutoReset(C.Structure):
    """``struct pvder_autoreset`` (include/pvder_hip.h)."""
    _fields_ = [("enabled", C.c_int32), ("keep_events", C.c_int32), ("seed", C.c_uint64),
                ("env_id_base", C.c_int64), ("spec", EventSpec), ("y0", C.POINTER(C.c_double)), ("env_ids", C.c_void_p),
                ("lazy_events", C.c_int32), ("reserved0", C.c_int32)]


class QAgentCfg(C.Structure):
    """``struct pvder_qagent_cfg`` (include/pvder_agent.h)."""
    _fields_ = [("bins", C.c_int32), ("n_act", C.c_int32), ("first", C.c_int32), ("choose", C.c_int32),
                ("lr", C.c_float), ("gamma", C.c_float), ("epsilon", C.c_float), ("q_done", C.c_float),
                ("low", C.c_double), ("win", C.c_double), ("seed", C.c_uint64), ("counter", C.c_uint64),
                ("env_id_base", C.c_int64), ("policy", C.c_int32), ("hidden", C.c_int32)]


class LinCoef(C.Structure):
    """``struct pvder_lin_coef`` (include/pvder_hip.h)."""
    _fields_ = [("sets", C.c_void_p), ("n_sets", C.c_int32), ("len_ticks", C.c_int32 * LIN_MAX_SETS),
                ("sched", C.c_void_p), ("est_guard", C.c_float), ("n_sched_rows", C.c_int32), ("seq_prop", C.c_float * 18),
                ("k4", C.c_int32)]


# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs (the VALU code reads them without
# v_accvgpr_read copies); -amdgpu-kernarg-preload-count: the leading scalar kernel arguments (pointers and
# sizes a launch needs for its first loads) arrive in SGPRs with the wave instead of through an s_load
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
               "-mllvm", "-amdgpu-kernarg-preload-count=12"]


def hipcc_path() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into gym_PVDER_amd/lib/libpvder_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(INCLUDE, "pvder_hip.h"), os.path.join(INCLUDE, "pvder_agent.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH) for d in deps)):
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())          # build aside, then rename: never a half-written library
    cmd = [hipcc_path()] + HIPCC_FLAGS + ["-fPIC", "-shared", "-I" + INCLUDE, "-I" + CSRC, "-o", tmp] + srcs
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libpvder_hip.so (after torch, so that torch's libamdhip64 is the one HIP runtime of
    the process) and set the prototypes.  Raises if the library or a symbol is missing."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    # A/B runs of variant builds (tools/ab_*.sh) point PVDER_LIB_PATH at their own library instead of
    # copying it over the shipped one; the same C ABI, the same checks below
    LIB_PATH = os.environ.get("PVDER_LIB_PATH", LIB_PATH)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "native library %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for the PVDER step path"
            % LIB_PATH)
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so first: single HIP runtime)
    L = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS + AGENT_SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError("libpvder_hip.so lacks symbols: %s" % missing)
    if L.pvder_version() != ABI_VERSION:
        raise ImportError("libpvder_hip.so ABI %d != expected %d" % (L.pvder_version(), ABI_VERSION))
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    dp = C.POINTER(C.c_double)
    cfgp, arp = C.POINTER(StepCfg), C.POINTER(AutoReset)
    for name in ("pvder_reset_masked_f32", "pvder_reset_masked_f64"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [dp, cfgp, arp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.pvder_reset_masked_lanes_f32.restype = i32
    L.pvder_reset_masked_lanes_f32.argtypes = [dp, cfgp, arp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.pvder_plan_step_fetch.restype = i32
    L.pvder_plan_step_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.pvder_plan_step_mirror.restype = i32
    L.pvder_plan_step_mirror.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_uint64, vp]
    L.pvder_fetch_env.restype = i32
    L.pvder_fetch_env.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.pvder_step_rk4_f32.restype = i32
    L.pvder_step_rk4_f32.argtypes = [dp, cfgp, arp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.pvder_step_etdrk4_f32.restype = i32
    L.pvder_step_etdrk4_f32.argtypes = [dp, cfgp, arp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.pvder_etdrk4_coefficients.restype = i32
    L.pvder_etdrk4_coefficients.argtypes = [dp, cfgp, C.POINTER(C.c_float)]
    L.pvder_step_rk45_f64.restype = i32
    L.pvder_step_rk45_f64.argtypes = [dp, cfgp, arp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    linp = C.POINTER(LinCoef)
    L.pvder_step_etdlin_f32.restype = i32
    L.pvder_step_etdlin_f32.argtypes = [dp, cfgp, arp, vp, linp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.pvder_plan_create.restype = i32
    L.pvder_plan_create.argtypes = [C.POINTER(vp), i32, dp, cfgp, arp, vp, linp, i32]
    L.pvder_plan_destroy.restype = None
    L.pvder_plan_destroy.argtypes = [vp]
    L.pvder_plan_hint.restype = i32
    L.pvder_plan_hint.argtypes = [vp, i32]
    L.pvder_plan_graph_hints.restype = i32
    L.pvder_plan_graph_hints.argtypes = [vp, i32]
    L.pvder_plan_kernel_name.restype = C.c_char_p
    L.pvder_plan_kernel_name.argtypes = [vp]
    L.pvder_plan_hybrid_active.restype = i32
    L.pvder_plan_hybrid_active.argtypes = [vp]
    ipp = C.POINTER(C.POINTER(C.c_int))
    L.pvder_served_alloc.restype = i32
    L.pvder_served_alloc.argtypes = [i32, ipp, ipp, ipp]
    L.pvder_served_free.restype = i32
    L.pvder_served_free.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pvder_plan_rollout_served.restype = i32
    L.pvder_plan_rollout_served.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, C.c_size_t, vp, vp, vp, vp, i32, i32, vp]
    L.pvder_complete_event_tables.restype = i32
    L.pvder_complete_event_tables.argtypes = [C.POINTER(C.c_double), C.POINTER(StepCfg), C.POINTER(AutoReset), vp, vp, vp, i32, vp]
    L.pvder_fold_lane_params.restype = i32
    L.pvder_fold_lane_params.argtypes = [dp, vp, i32, vp]
    L.pvder_plan_lane_params.restype = i32
    L.pvder_plan_lane_params.argtypes = [vp, vp]
    L.pvder_plan_step.restype = i32
    L.pvder_plan_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pvder_plan_rollout.restype = i32
    L.pvder_plan_rollout.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, C.c_size_t, vp, vp, vp]
    L.pvder_plan_rollout_fused.restype = i32
    L.pvder_plan_rollout_fused.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, C.c_size_t, vp, vp, vp]
    L.pvder_qagent_step_f32.restype = i32
    L.pvder_qagent_step_f32.argtypes = [C.POINTER(QAgentCfg), vp, vp, vp, vp, vp, vp, i32, vp]
    L.pvder_plan_rollout_fused_agent.restype = i32
    L.pvder_plan_rollout_fused_agent.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, C.POINTER(QAgentCfg), vp, vp, vp, vp, vp, vp]
    L.pvder_gather_create.restype = i32
    L.pvder_gather_create.argtypes = [C.POINTER(vp), vp, vp]
    L.pvder_gather_destroy.restype = None
    L.pvder_gather_destroy.argtypes = [vp]
    L.pvder_gather_drain.restype = i32
    L.pvder_gather_drain.argtypes = [vp, vp]
    L.pvder_plan_rollout_gather.restype = i32
    L.pvder_plan_rollout_gather.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32]
    L.pvder_plan_rollout_gather_rows.restype = i32
    L.pvder_plan_rollout_gather_rows.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                                                 C.c_longlong]
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == EINVAL:
        raise ValueError("%s: invalid argument (PVDER_EINVAL)" % what)
    if rc == ETIMEOUT:
        raise TimeoutError("%s: the device did not deliver the step within the time limit (PVDER_ETIMEOUT)" % what)
    raise RuntimeError("%s: HIP error %d" % (what, -rc))
