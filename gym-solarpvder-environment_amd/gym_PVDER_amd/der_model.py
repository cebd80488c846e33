"""Host-side DER model set-up: per-unit parameter vector + steady-state initial state.

Host counterpart of what the reference does once per ``reset()`` through the third-party
``pvder`` package (``DERModel(..., derId, configFile, standAlone=True,
steadyStateInitialization=True)``, /root/reference/gym_PVDER/envs/PVDER_env.py:374-378): read the
``derId`` block of ``config_der.json`` (:2-21 for "50", :66-84 for "10"), per-unitise it with
pvder's base values, and find the operating point the ODE starts from.  The result is a flat
``double[PVDER_PAR_COUNT]`` (layout: include/pvder_hip.h ``enum pvder_par``) that the HIP
launchers fold into kernel-argument constants, and ``y0[ns]`` in the kernels' state order.

fp64, host only, runs once per env construction -- not on the hot path.
"""
from __future__ import annotations

import cmath
import math
from typing import Dict, Optional

import numpy as np

# ---- parameter blocks (values restated from the reference's config_der.json) ----------------
DER_CONFIG: Dict[str, Dict[str, float]] = {
    # derId "50" -- model_2 'ThreePhaseUnbalanced' (PVDER_env.py:57), config_der.json:2-21
    "50": dict(phases=3, Sinsol=100.0, Np=11, Ns=735, Vdcmpp0=550.0, Srated=50e3, Vdcrated=550.0,
               Ioverload=1.3, Vrmsrated=177.0, Rf_actual=0.002, Lf_actual=25.0e-6,
               C_actual=300.0e-6, R1_actual=0.0019, X1_actual=0.0561, Kp_GCC=6000.0, Ki_GCC=2000.0,
               Kp_DC=-2.0, Ki_DC=-10.0, Kp_Q=0.2, Ki_Q=10.0, wp=20e4, iaR0=1.0, iaI0=0.001),
    # derId "10" -- model_1 'SinglePhase' (PVDER_env.py:56), config_der.json:66-84
    "10": dict(phases=1, Sinsol=100.0, Np=2, Ns=1000, Vdcmpp0=750.0, Srated=10e3, Vdcrated=550.0,
               Ioverload=1.3, Vrmsrated=177.0, Rf_actual=0.002, Lf_actual=25.0e-6,
               C_actual=30.0e-6, R1_actual=0.0019, X1_actual=0.0561, Kp_GCC=12000.0, Ki_GCC=4000.0,
               Kp_DC=-4.0, Ki_DC=-20.0, Kp_Q=0.4, Ki_Q=20.0, wp=20e4, iaR0=0.5, iaI0=0.01),
}

# ---- base values / grid / PV-cell constants of the published pvder model ---------------------
VBASE = 500.0
SBASE = 50e3
WBASE = 2.0 * math.pi * 60.0
VDCBASE = VBASE
IBASE = SBASE / VBASE
ZBASE = VBASE ** 2 / SBASE
LBASE = ZBASE / WBASE
CBASE = 1.0 / (ZBASE * WBASE)
VGRIDRATED = 20415.0
Z2_ACTUAL = 1.61 + 5.54j
ZLOAD1_ACTUAL = 10e6 + 0j
KP_PLL, KI_PLL = 180.0, 320.0
ISCR, IRS, T_CELL, Q_E, K_B, A_IDEAL = 8.03, 1.2e-7, 298.15, 1.602e-19, 1.38e-23, 1.92
M_LIMIT = 1.0
WTE0 = 2.0 * math.pi

# index names of the parameter vector -- MUST match include/pvder_hip.h `enum pvder_par`
PAR = ["PHASES", "RF", "LF", "C", "KVG_RE", "KVG_IM", "KVI_RE", "KVI_IM", "KP_GCC", "KI_GCC",
       "KP_DC", "KI_DC", "KP_Q", "KI_Q", "WP", "KP_PLL", "KI_PLL", "M_LIMIT", "IREF_LIMIT", "WBASE",
       "VG_SCALE", "VDCBASE", "SBASE", "NP", "ISCR", "IRS", "PV_VT", "VRMS_REF", "VDC_REF0", "SINSOL0", "PPV_AT_MPP"]
# model options: the questions about the pvder model that /root/reference cannot answer and that DO move the observations
# (measured: profiles/r04_oracle_uncertainty.txt).  U5 "ppv_at_mpp": the PV array's power at the MPP
# voltage Vdcmpp0 instead of the instantaneous DC-link voltage (1e-4 ... 7e-4 in the observations).  U1 (winv = grid
# frequency) moves them by < 3e-6 and U7 (ratio Vgridrated / Vbase) has no steady state inside the duty-cycle limit: neither is
# a switch of the device model.
MODEL_OPTIONS = ("ppv_at_mpp",)
IDX = {k: i for i, k in enumerate(PAR)}


def der_config(der_id: str = "50", overrides: Optional[Dict[str, float]] = None) -> Dict[str, float]:
    """The config block of ``der_id`` (values of config_der.json) with ``overrides`` applied -- a
    parameter SET for domain randomisation: any key of the block except ``phases``."""
    if der_id not in DER_CONFIG:
        raise ValueError("unknown derId %r (available: %s)" % (der_id, sorted(DER_CONFIG)))
    c = dict(DER_CONFIG[der_id])
    for k, v in (overrides or {}).items():
        if k not in c or k == "phases":
            raise ValueError("der_config override %r: not a parameter of derId %s (%s)"
                             % (k, der_id, sorted(x for x in c if x != "phases")))
        if isinstance(v, bool) or not isinstance(v, (int, float)) or not math.isfinite(float(v)):
            raise ValueError("der_config override %r must be a finite number" % k)
        c[k] = float(v)
    return c


def param_vector(der_id: str = "50", overrides: Optional[Dict[str, float]] = None,
                 options: Optional[Dict[str, bool]] = None) -> np.ndarray:
    """Per-unit constants of one DER as the C ABI's ``double par[PVDER_PAR_COUNT]``; ``options``: MODEL_OPTIONS."""
    for k in (options or {}):
        if k not in MODEL_OPTIONS:
            raise ValueError("unknown model option %r (available: %s)" % (k, MODEL_OPTIONS))
    c = der_config(der_id, overrides)
    ph = int(c["phases"])
    a = VGRIDRATED / (c["Vrmsrated"] * math.sqrt(2.0))          # transformer turns ratio
    Z1 = complex(c["R1_actual"], c["X1_actual"]) / ZBASE
    Z2 = Z2_ACTUAL / ZBASE
    Zl = ZLOAD1_ACTUAL / ZBASE
    # stand-alone PCC-LV voltage: v = ((vg + (i/a) Z2)/a + i Z1) * Zl a^2/(a^2 (Z1+Zl) + Z2)
    kdiv = (Zl * a * a) / (a * a * (Z1 + Zl) + Z2)
    kvg = kdiv / a
    kvi = kdiv * (Z2 / (a * a) + Z1)
    irated = c["Srated"] / ((3.0 if ph == 3 else 1.0) * c["Vrmsrated"]) * math.sqrt(2.0)
    v = dict(PHASES=ph, RF=c["Rf_actual"] / ZBASE, LF=c["Lf_actual"] / LBASE, C=c["C_actual"] / CBASE,
             KVG_RE=kvg.real, KVG_IM=kvg.imag, KVI_RE=kvi.real, KVI_IM=kvi.imag,
             KP_GCC=c["Kp_GCC"], KI_GCC=c["Ki_GCC"], KP_DC=c["Kp_DC"], KI_DC=c["Ki_DC"],
             KP_Q=c["Kp_Q"], KI_Q=c["Ki_Q"], WP=c["wp"], KP_PLL=KP_PLL, KI_PLL=KI_PLL,
             M_LIMIT=M_LIMIT, IREF_LIMIT=irated / IBASE * c["Ioverload"], WBASE=WBASE,
             VG_SCALE=VGRIDRATED / VBASE, VDCBASE=VDCBASE, SBASE=SBASE, NP=c["Np"], ISCR=ISCR, IRS=IRS,
             PV_VT=Q_E * VDCBASE / (K_B * T_CELL * A_IDEAL * c["Ns"]),
             VRMS_REF=c["Vrmsrated"] / VBASE, VDC_REF0=c["Vdcmpp0"] / VDCBASE, SINSOL0=c["Sinsol"],
             PPV_AT_MPP=1.0 if (options or {}).get("ppv_at_mpp") else 0.0)
    return np.array([float(v[k]) for k in PAR], dtype=np.float64)


# Per-LANE overrides (include/pvder_hip.h PVDER_LP_*: the parameters outside the exponential operator of the 23-state ETDRK4
# path): config_der.json keys (:2-21) plus the insolation scale, in the row order of the C ABI
LANE_PARAMS = ("C_actual", "Kp_DC", "Ki_DC", "Kp_Q", "Ki_Q", "Np", "Sinsol_scale")


def lane_param_rows(lane_params: Dict[str, "np.ndarray"], num_envs: int) -> np.ndarray:
    """``{key: [N] values}`` (keys of LANE_PARAMS; a missing key or a NaN entry keeps the env at the group's value) ->
    ``double[PVDER_LP_ROWS][N]`` in the units of ``par`` (C per unit, the rest as configured)."""
    rows = np.full((len(LANE_PARAMS), int(num_envs)), np.nan, dtype=np.float64)
    for k, v in lane_params.items():
        if k not in LANE_PARAMS:
            raise ValueError("lane_params key %r is not one of %s (everything else belongs to the parameter group: "
                             "der_config / param_groups.PVDERVecGroups)" % (k, LANE_PARAMS))
        a = np.asarray(v, dtype=np.float64).reshape(-1)
        if a.shape != (int(num_envs),):
            raise ValueError("lane_params[%r] must hold one value per env (%d)" % (k, num_envs))
        rows[LANE_PARAMS.index(k)] = a / CBASE if k == "C_actual" else a
    return rows


def ppv(par: np.ndarray, vdc: float, sinsol: float) -> float:
    """PV array power in pu at DC-link voltage ``vdc`` (single-diode array model)."""
    vact = vdc * par[IDX["VDCBASE"]]
    iph = par[IDX["ISCR"]] * sinsol / 100.0
    p = par[IDX["NP"]] * vact * (iph - par[IDX["IRS"]] * (math.exp(par[IDX["PV_VT"]] * vdc) - 1.0))
    return max(0.0, p) / par[IDX["SBASE"]]


def steady_state(par: np.ndarray, der_id: str = "50", vgrid: float = 1.0, q_ref: float = 0.0) -> np.ndarray:
    """Operating point the episode starts from (steadyStateInitialization=True, PVDER_env.py:378):
    Re S = Ppv(Vdc_ref), Im S_PCC = Q_ref, di/dt = 0 at w = wbase; then x = m, u = 0,
    xDC + j xQ = ia, Vdc = Vdc_ref, xPLL = 0 and the PLL angle at its reset value.
    Two-unknown Newton iteration in fp64; returns y0 in the kernels' state order with the LAST
    entry in the relative-angle form  delta = wte - wbase*t = wte0 at t = 0."""
    ph = int(par[IDX["PHASES"]])
    kvg = complex(par[IDX["KVG_RE"]], par[IDX["KVG_IM"]])
    kvi = complex(par[IDX["KVI_RE"]], par[IDX["KVI_IM"]])
    zf = complex(par[IDX["RF"]], par[IDX["LF"]])
    vdc = par[IDX["VDC_REF0"]]
    p_pv = ppv(par, vdc, par[IDX["SINSOL0"]])
    vga = vgrid * par[IDX["VG_SCALE"]]

    def resid(z):
        ia = complex(z[0], z[1])
        va = kvg * vga + kvi * ia
        vta = va + zf * ia
        return np.array([(0.5 * ph * vta * ia.conjugate()).real - p_pv,
                         (0.5 * ph * va * ia.conjugate()).imag - q_ref])

    z = np.array([DER_CONFIG[der_id]["iaR0"], DER_CONFIG[der_id]["iaI0"]], dtype=np.float64)
    for _ in range(50):
        r = resid(z)
        J = np.empty((2, 2))
        for k in range(2):
            dz = np.zeros(2); dz[k] = 1e-7
            J[:, k] = (resid(z + dz) - resid(z - dz)) / 2e-7
        step = np.linalg.solve(J, r)
        z = z - step
        if np.abs(step).max() < 1e-15:
            break
    else:
        raise RuntimeError("steady-state Newton iteration did not converge")
    ia = complex(z[0], z[1])
    ma = (kvg * vga + kvi * ia + zf * ia) / (vdc / 2.0)
    rot = [1.0, cmath.exp(-2j * math.pi / 3.0), cmath.exp(2j * math.pi / 3.0)][:ph]
    y0 = []
    for r_ in rot:
        ip, mp = ia * r_, ma * r_
        y0 += [ip.real, ip.imag, mp.real, mp.imag, 0.0, 0.0]
    y0 += [vdc, ia.real, ia.imag, 0.0, WTE0]
    return np.asarray(y0, dtype=np.float64)


def physical_outputs(par: np.ndarray, y: np.ndarray, vgrid: np.ndarray, q_ref: np.ndarray,
                     vdc_ref: np.ndarray) -> dict:
    """Host-side (numpy, fp64) algebraic outputs of state rows ``y`` ([NS, ...], kernels' state
    order) in PHYSICAL units -- the quantities ``PVDER.render`` prints (PVDER_env.py:339-349) and the
    reference's ``SimulationResults`` plots (:359-364): duty cycle ma, currents, LV-side voltages,
    inverter / PCC powers, DC-link voltage, PV power.  Display only: nothing on the step path uses it."""
    ph = int(par[IDX["PHASES"]])
    kvg = complex(par[IDX["KVG_RE"]], par[IDX["KVG_IM"]])
    kvi = complex(par[IDX["KVI_RE"]], par[IDX["KVI_IM"]])
    y = np.asarray(y, dtype=np.float64)
    B = 6 * ph
    rot = [1.0, cmath.exp(-2j * math.pi / 3.0), cmath.exp(2j * math.pi / 3.0)][:ph]
    vdc = y[B]
    out = {"Vdc": vdc * VDCBASE, "Q_ref": np.asarray(q_ref) * SBASE, "Vdc_ref": np.asarray(vdc_ref) * VDCBASE}
    s_inv = 0.0
    s_pcc = 0.0
    for p, name in enumerate("abc"[:ph]):
        i_p = y[6 * p] + 1j * y[6 * p + 1]
        x_p = y[6 * p + 2] + 1j * y[6 * p + 3]
        u_p = y[6 * p + 4] + 1j * y[6 * p + 5]
        m_p = par[IDX["KP_GCC"]] * u_p + x_p
        vt_p = m_p * vdc / 2.0
        v_p = kvg * (np.asarray(vgrid) * par[IDX["VG_SCALE"]] * rot[p]) + kvi * i_p
        s_inv = s_inv + 0.5 * vt_p * np.conj(i_p)
        s_pcc = s_pcc + 0.5 * v_p * np.conj(i_p)
        out["m" + name], out["i" + name] = m_p, i_p * IBASE
        out["vt" + name], out["v" + name] = vt_p * VBASE, v_p * VBASE
    out["S_inverter"], out["S_PCC"] = s_inv * SBASE, s_pcc * SBASE
    return out
