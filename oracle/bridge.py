"""CPU ORACLE (test infrastructure only) -- ctypes front-end of oracle/pvder_bridge.c.

Builds ``oracle/_build/libpvder_bridge.so`` on demand (gcc, seconds).  Used by tests/, by
``__graft_entry__.smoke()`` and by bench.py's ``cpu_baseline`` leg only.  PARITY UNPINNED (see
oracle/pvder_model.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from . import pvder_model as pm

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libpvder_bridge.so")

PAR_NAMES = ["PHASES", "RF", "LF", "C", "KVG_RE", "KVG_IM", "KVI_RE", "KVI_IM", "KP_GCC", "KI_GCC",
             "KP_DC", "KI_DC", "KP_Q", "KI_Q", "WP", "KP_PLL", "KI_PLL", "M_LIMIT", "IREF_LIMIT",
             "WBASE", "VG_SCALE", "VDCBASE", "SBASE", "NP", "ISCR", "IRS", "PV_VT", "VDC_REF0", "PPV_AT_MPP"]

_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in ("pvder_bridge.c", "pvder_bridge_impl.h", "pvder_bridge_filippov.h", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        assert _lib.pvder_bridge_par_count() == len(PAR_NAMES)
    return _lib


def pack_params(m: pm.DERModel) -> np.ndarray:
    v = dict(PHASES=m.phases, RF=m.Rf, LF=m.Lf, C=m.C, KVG_RE=m.kv_g.real, KVG_IM=m.kv_g.imag,
             KVI_RE=m.kv_i.real, KVI_IM=m.kv_i.imag, KP_GCC=m.Kp_GCC, KI_GCC=m.Ki_GCC,
             KP_DC=m.Kp_DC, KI_DC=m.Ki_DC, KP_Q=m.Kp_Q, KI_Q=m.Ki_Q, WP=m.wp, KP_PLL=m.Kp_PLL,
             KI_PLL=m.Ki_PLL, M_LIMIT=m.m_limit, IREF_LIMIT=m.iref_limit, WBASE=pm.WBASE,
             VG_SCALE=m.vg_scale, VDCBASE=pm.VDCBASE, SBASE=pm.SBASE, NP=m.Np, ISCR=pm.ISCR,
             IRS=pm.IRS, PV_VT=m.pv_vt, VDC_REF0=m.Vdc_ref0, PPV_AT_MPP=0.0 if m.opts.ppv_uses_vdc else 1.0)
    return np.array([float(v[k]) for k in PAR_NAMES], dtype=np.float64)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _vec(x, N, dt):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=dt), (N,)))
    return a


def to_delta(y: np.ndarray, t) -> np.ndarray:
    """absolute PLL angle wte -> relative angle delta = wte - wbase*t (last state)."""
    y = np.array(y, dtype=np.float64, copy=True)
    y[..., -1] = y[..., -1] - pm.WBASE * np.asarray(t, dtype=np.float64)
    return y


def to_wte(y: np.ndarray, t) -> np.ndarray:
    y = np.array(y, dtype=np.float64, copy=True)
    y[..., -1] = y[..., -1] + pm.WBASE * np.asarray(t, dtype=np.float64)
    return y


def rk4_batch(m: pm.DERModel, Y: np.ndarray, t0, h: float, nsteps: int, vg, si, qref, vdcref,
              dtype=np.float64, comp: Optional[np.ndarray] = None) -> np.ndarray:
    """Y: [N, ns] in the delta form; returns the advanced copy (same dtype as requested).
    comp: optional [N, ns] Kahan compensation array of the same dtype, updated IN PLACE."""
    par = pack_params(m)
    Y = np.ascontiguousarray(Y, dtype=dtype).copy()
    N = Y.shape[0]
    if comp is not None:
        assert comp.dtype == dtype and comp.shape == Y.shape and comp.flags.c_contiguous
    cp64 = _p(comp, C.c_double) if (comp is not None and dtype == np.float64) else None
    cp32 = _p(comp, C.c_float) if (comp is not None and dtype == np.float32) else None
    t0 = _vec(t0, N, np.float64)
    if dtype == np.float64:
        a = [_vec(x, N, np.float64) for x in (vg, si, qref, vdcref)]
        lib().pvder_bridge_rk4_batch_f64(_p(par, C.c_double), C.c_int(N), _p(Y, C.c_double), cp64,
                                         _p(t0, C.c_double), C.c_double(h), C.c_long(nsteps),
                                         *[_p(x, C.c_double) for x in a])
    else:
        a = [_vec(x, N, np.float32) for x in (vg, si, qref, vdcref)]
        lib().pvder_bridge_rk4_batch_f32(_p(par, C.c_double), C.c_int(N), _p(Y, C.c_float), cp32,
                                         _p(t0, C.c_double), C.c_double(h), C.c_long(nsteps),
                                         *[_p(x, C.c_float) for x in a])
    return Y


def rk45_batch(m: pm.DERModel, Y: np.ndarray, t0, T: float, rtol: float, atol: float, hstep,
               vg, si, qref, vdcref):
    """Adaptive DP5(4) over [t0, t0+T]; returns (Y, hstep, nacc, nrej)."""
    par = pack_params(m)
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    N = Y.shape[0]
    t0 = _vec(t0, N, np.float64)
    hs = _vec(hstep, N, np.float64).copy()
    a = [_vec(x, N, np.float64) for x in (vg, si, qref, vdcref)]
    nacc = np.zeros(N, dtype=np.int64)
    nrej = np.zeros(N, dtype=np.int64)
    lib().pvder_bridge_rk45_batch_f64(_p(par, C.c_double), C.c_int(N), _p(Y, C.c_double),
                                      _p(t0, C.c_double), C.c_double(T), C.c_double(rtol),
                                      C.c_double(atol), _p(hs, C.c_double),
                                      *[_p(x, C.c_double) for x in a], _p(nacc, C.c_long),
                                      _p(nrej, C.c_long))
    return Y, hs, nacc, nrej


def filippov_batch(m: pm.DERModel, Y: np.ndarray, t0, T: float, rtol: float, atol: float, hstep, vg, si, qref, vdcref,
                   region: np.ndarray):
    """Event-locating DP5(4) over [t0, t0+T] following the sliding (Filippov) solution at the duty-cycle limit
    (pvder_bridge_filippov.h).  ``region``: [N, 3] int32, updated IN PLACE.  Returns (Y, hstep, nacc, nrej, nev)."""
    par = pack_params(m)
    Y = np.ascontiguousarray(Y, dtype=np.float64).copy()
    N = Y.shape[0]
    assert region.dtype == np.int32 and region.shape == (N, 3) and region.flags.c_contiguous
    t0 = _vec(t0, N, np.float64)
    hs = _vec(hstep, N, np.float64).copy()
    a = [_vec(x, N, np.float64) for x in (vg, si, qref, vdcref)]
    nacc, nrej, nev = (np.zeros(N, dtype=np.int64) for _ in range(3))
    lib().pvder_bridge_filippov_batch_f64(_p(par, C.c_double), C.c_int(N), _p(Y, C.c_double), _p(t0, C.c_double),
                                          C.c_double(T), C.c_double(rtol), C.c_double(atol), _p(hs, C.c_double),
                                          *[_p(x, C.c_double) for x in a], _p(region, C.c_int), _p(nacc, C.c_long),
                                          _p(nrej, C.c_long), _p(nev, C.c_long))
    return Y, hs, nacc, nrej, nev


def filippov_evaluate(m: pm.DERModel, y: np.ndarray, t: float, vg: float, si: float, qref: float, vdcref: float, region):
    """(dy, aux[3 phases][s, N, D]) of the event-locating integrator's right-hand side with the given regions."""
    par = pack_params(m)
    y = np.ascontiguousarray(y, dtype=np.float64)
    reg = np.ascontiguousarray(region, dtype=np.int32)
    dy, aux = np.zeros(m.n_states), np.zeros(9)
    lib().pvder_bridge_filippov_eval_f64(_p(par, C.c_double), _p(y, C.c_double), C.c_double(t), C.c_double(vg), C.c_double(si),
                                         C.c_double(qref), C.c_double(vdcref), _p(reg, C.c_int), _p(dy, C.c_double),
                                         _p(aux, C.c_double))
    return dy, aux.reshape(3, 3)


def evaluate(m: pm.DERModel, y: np.ndarray, t: float, vg: float, si: float, qref: float,
             vdcref: float, dtype=np.float64):
    """(dy, outs[16]) at one state in delta form."""
    par = pack_params(m)
    y = np.ascontiguousarray(y, dtype=dtype)
    dy = np.zeros(m.n_states, dtype=dtype)
    outs = np.zeros(16, dtype=dtype)
    if dtype == np.float64:
        lib().pvder_bridge_eval_f64(_p(par, C.c_double), _p(y, C.c_double), C.c_double(t),
                                    C.c_double(vg), C.c_double(si), C.c_double(qref),
                                    C.c_double(vdcref), _p(dy, C.c_double), _p(outs, C.c_double))
    else:
        lib().pvder_bridge_eval_f32(_p(par, C.c_double), _p(y, C.c_float), C.c_double(t),
                                    C.c_float(vg), C.c_float(si), C.c_float(qref),
                                    C.c_float(vdcref), _p(dy, C.c_float), _p(outs, C.c_float))
    return dy, outs
