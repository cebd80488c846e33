"""CPU ORACLE (test infrastructure only) -- dynamic-phasor PV-DER model restatement.

This file is *test infrastructure*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path (the HIP kernels behind
``include/pvder_hip.h``) never does.

PARITY UNPINNED.  The arithmetic of the hot path of the reference
(``/root/reference/gym_PVDER/envs/PVDER_env.py:166`` -> ``self.sim.run_simulation()``) lives in
the third-party package ``pvder`` (sibyjackgrove/SolarPV-DER-simulation-utility).  It is not
vendored in the reference, not pinned to any version (``/root/reference/setup.py:5``) and not
installed in the build container, and the reference holds no golden vectors
(``/root/reference/gym_PVDER/tests/test_gym_PVDER.py`` checks structure only).  What follows is
therefore a restatement of the *published* dynamic-phasor PV-DER model that ``pvder``
implements (``SolarPVDERThreePhase.ODE_model``, ``Grid.steady_state_model``,
``SimulationEvents.grid_events/solar_events``, ``PVModule.Ppv_calc``), anchored on

* the reference's call sites: ``PVDER_env.py:26-35`` (imports), ``:371-396`` (construction),
  ``:166-167`` (integration), ``:225-299`` and ``:535-540`` (attributes read back),
* the reference's parameter file ``config_der.json:2-21`` (derId "50") and ``:66-84`` ("10"),
* the circuit of ``PVDER_schematic.png`` (PV -> C -> inverter -> Lf,Rf -> PCC-LV || Zload1 ->
  transformer Z1 -> PCC-HV -> line Z2 -> grid source).

Every modelling choice that cannot be resolved from ``/root/reference`` is a named option in
:class:`ModelOptions` (U1..U7 of SURVEY.md Appendix A); DESIGN.md lists the values chosen.

All phasors are peak-value per-unit complex numbers; time is in seconds.
"""
from __future__ import annotations

import cmath
import math
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------------
# Parameter blocks restated from /root/reference/config_der.json (values only, data not code)
# --------------------------------------------------------------------------------------------

#: derId "50": config_der.json:2-21 (default model_2 of PVDER-v0, PVDER_env.py:57)
#: derId "10": config_der.json:66-84 (model_1, single phase, PVDER_env.py:56)
DER_CONFIG: Dict[str, Dict[str, float]] = {
    "50": dict(
        phases=3,
        Sinsol=100.0,                               # config_der.json:4
        Np=11, Ns=735, Vdcmpp0=550.0,               # :5
        Srated=50e3, Vdcrated=550.0, Ioverload=1.3, Vrmsrated=177.0,   # :7
        Rf_actual=0.002, Lf_actual=25.0e-6, C_actual=300.0e-6,         # :9
        R1_actual=0.0019, X1_actual=0.0561,                            # :10
        Kp_GCC=6000.0, Ki_GCC=2000.0,               # :12
        Kp_DC=-2.0, Ki_DC=-10.0,                    # :13
        Kp_Q=0.2, Ki_Q=10.0, wp=20e4,               # :14
        maR0=0.89, maI0=0.0, iaR0=1.0, iaI0=0.001,  # :16 (seeds of the steady-state solve)
    ),
    "10": dict(
        phases=1,
        Sinsol=100.0,                               # config_der.json:68
        Np=2, Ns=1000, Vdcmpp0=750.0,               # :69
        Srated=10e3, Vdcrated=550.0, Ioverload=1.3, Vrmsrated=177.0,   # :71
        Rf_actual=0.002, Lf_actual=25.0e-6, C_actual=30.0e-6,          # :73
        R1_actual=0.0019, X1_actual=0.0561,                            # :74
        Kp_GCC=12000.0, Ki_GCC=4000.0,              # :76
        Kp_DC=-4.0, Ki_DC=-20.0,                    # :77
        Kp_Q=0.4, Ki_Q=20.0, wp=20e4,               # :78
        maR0=0.7, maI0=0.0, iaR0=0.5, iaI0=0.01,    # :80
    ),
}

# Base values / grid constants of the pvder package (published model; not in /root/reference).
VBASE = 500.0                 # L-N peak
SBASE = 50e3
WBASE = 2.0 * math.pi * 60.0
VDCBASE = VBASE
IBASE = SBASE / VBASE
ZBASE = VBASE ** 2 / SBASE
LBASE = ZBASE / WBASE
CBASE = 1.0 / (ZBASE * WBASE)
VGRIDRATED = 20415.0          # L-N peak of a 25 kV L-L rms feeder
Z2_ACTUAL = 1.61 + 5.54j      # line impedance (ohm)
ZLOAD1_ACTUAL = 10e6 + 0j     # load at PCC-LV (open)
KP_PLL = 180.0
KI_PLL = 320.0
# PV module constants (single-diode model)
ISCR = 8.03
KV = 0.0017
T0_CELL = 298.15
IRS = 1.2e-7
Q_E = 1.602e-19
K_B = 1.38e-23
A_IDEALITY = 1.92

A120 = cmath.exp(2j * math.pi / 3.0)       # e^{+j120deg}
A240 = cmath.exp(-2j * math.pi / 3.0)      # e^{-j120deg}


@dataclass(frozen=True)
class ModelOptions:
    """Named resolutions of SURVEY.md Appendix A uncertainties (U1..U7)."""
    winv_is_pll: bool = True          # U1: cross-coupling frequency is the PLL frequency `we`
    m_limit: float = 1.0              # U2: anti-windup threshold on |m_p|
    refs_external: bool = True        # U3: Q_ref / Vdc_ref persist (env intent, PVDER_env.py:225-229,394)
    event_inclusive: bool = True      # U4: event at T is active for t >= T
    ppv_uses_vdc: bool = True         # U5: Ppv evaluated at the instantaneous Vdc (False: at the MPP voltage Vdcmpp0)
    ratio_is_vbase: bool = False      # U7: False: transformer ratio a = Vgridrated/(sqrt(2) Vrmsrated) (the choice consistent
    #                                   with Vdc = 550 V and |m| ~ 0.9); True: SURVEY Appendix A's Vgridrated/Vbase
    rtol: float = 1e-4                # U6: LSODA tolerances of the "reference-settings" run
    atol: float = 1e-4
    hmax: float = 1.0 / 120.0
    mxstep: int = 5000               # pvder passes mxstep=50 [recollection]; 50 aborts after some events
    wte0: float = 2.0 * math.pi       # PLL angle at reset


@dataclass(frozen=True)
class DERModel:
    """Per-unit constants of one DER (derived once on the host, fp64)."""
    der_id: str
    phases: int
    n_states: int
    Rf: float
    Lf: float
    C: float
    Z1: complex
    Z2: complex
    Zload1: complex
    a: float                      # transformer turns ratio Vgridrated/Varated
    Kp_GCC: float
    Ki_GCC: float
    Kp_DC: float
    Ki_DC: float
    Kp_Q: float
    Ki_Q: float
    wp: float
    Kp_PLL: float
    Ki_PLL: float
    m_limit: float
    iref_limit: float
    Np: float
    Ns: float
    Vdc_ref0: float
    Vrms_ref: float
    Sinsol0: float
    vg_scale: float               # Vgridrated/Vbase: grid event magnitude -> per-unit phasor
    # derived affine map  v_p = kv_g * vg_p + kv_i * i_p  (stand-alone PCC-LV voltage)
    kv_g: complex = 0j
    kv_i: complex = 0j
    pv_vt: float = 0.0            # q/(k T A Ns) * Vdcbase  -> exp(pv_vt * Vdc)
    seeds: Tuple[float, float, float, float] = (0.89, 0.0, 1.0, 0.001)
    opts: ModelOptions = field(default_factory=ModelOptions)


def build_model(der_id: str = "50", opts: Optional[ModelOptions] = None, overrides: Optional[dict] = None) -> DERModel:
    """Per-unitise a config block the way pvder's ``initialize_DER`` does (published model).
    ``overrides``: other values for keys of the block (a parameter set of a domain-randomised batch)."""
    opts = opts or ModelOptions()
    c = dict(DER_CONFIG[der_id])
    for k, v in (overrides or {}).items():
        assert k in c and k != "phases", k
        c[k] = float(v)
    phases = int(c["phases"])
    Varated = c["Vrmsrated"] * math.sqrt(2.0)
    a = VGRIDRATED / (VBASE if opts.ratio_is_vbase else Varated)
    Z1 = complex(c["R1_actual"], c["X1_actual"]) / ZBASE
    Z2 = Z2_ACTUAL / ZBASE
    Zl = ZLOAD1_ACTUAL / ZBASE
    if phases == 3:
        Iarated = (c["Srated"] / (3.0 * c["Vrmsrated"])) * math.sqrt(2.0)
    else:
        Iarated = (c["Srated"] / c["Vrmsrated"]) * math.sqrt(2.0)
    # v = ((vg + (i/a) Z2)/a + i Z1) * (Zl a^2 / (a^2 (Z1+Zl) + Z2))
    kdiv = (Zl * a * a) / (a * a * (Z1 + Zl) + Z2)
    kv_g = kdiv / a
    kv_i = kdiv * (Z2 / (a * a) + Z1)
    pv_vt = Q_E * VDCBASE / (K_B * T0_CELL * A_IDEALITY * c["Ns"])
    return DERModel(
        der_id=der_id, phases=phases, n_states=(23 if phases == 3 else 11),
        Rf=c["Rf_actual"] / ZBASE, Lf=c["Lf_actual"] / LBASE, C=c["C_actual"] / CBASE,
        Z1=Z1, Z2=Z2, Zload1=Zl, a=a,
        Kp_GCC=c["Kp_GCC"], Ki_GCC=c["Ki_GCC"], Kp_DC=c["Kp_DC"], Ki_DC=c["Ki_DC"],
        Kp_Q=c["Kp_Q"], Ki_Q=c["Ki_Q"], wp=c["wp"], Kp_PLL=KP_PLL, Ki_PLL=KI_PLL,
        m_limit=opts.m_limit, iref_limit=(Iarated / IBASE) * c["Ioverload"],
        Np=c["Np"], Ns=c["Ns"], Vdc_ref0=c["Vdcmpp0"] / VDCBASE,
        Vrms_ref=c["Vrmsrated"] / VBASE, Sinsol0=c["Sinsol"], vg_scale=VGRIDRATED / VBASE,
        kv_g=kv_g, kv_i=kv_i, pv_vt=pv_vt,
        seeds=(c["maR0"], c["maI0"], c["iaR0"], c["iaI0"]), opts=opts,
    )


# --------------------------------------------------------------------------------------------
# PV array, algebraic network, RHS  (reference style: Python complex scalars, one env)
# --------------------------------------------------------------------------------------------

def ppv_calc(m: DERModel, Vdc: float, Sinsol: float) -> float:
    """PV array power (pu).  pvder PVModule.Ppv_calc: single-diode array, clipped at zero."""
    Vdc_act = Vdc * VDCBASE
    Iph = (ISCR + KV * (T0_CELL - T0_CELL)) * (Sinsol / 100.0)
    p = m.Np * Iph * Vdc_act - m.Np * IRS * Vdc_act * (math.exp(m.pv_vt * Vdc) - 1.0)
    return max(0.0, p) / SBASE


def sign(x: float) -> int:
    return 1 if x > 0 else (-1 if x < 0 else 0)


class Algebraic:
    """Algebraic outputs of the model at a state (the attributes PVDER_env.py reads back)."""
    __slots__ = ("i", "x", "u", "m", "vt", "v", "S", "S_PCC", "Vrms", "Ppv", "iref", "vd", "we",
                 "Vdc", "xDC", "xQ", "xPLL", "wte")


def algebraic(m: DERModel, y: Sequence[float], t: float, vg_mag: float, Sinsol: float,
              Q_ref: float, Vdc_ref: float) -> Algebraic:
    """All algebraic quantities of pvder's update_voltages/update_power/update_RMS/update_iref/
    update_inverter_frequency chain (published model), for 1- or 3-phase."""
    P = m.phases
    o = Algebraic()
    if isinstance(y, np.ndarray):
        y = y.tolist()          # Python floats: ~5x faster scalar math than numpy scalars
    o.i = [complex(y[6 * p + 0], y[6 * p + 1]) for p in range(P)]
    o.x = [complex(y[6 * p + 2], y[6 * p + 3]) for p in range(P)]
    o.u = [complex(y[6 * p + 4], y[6 * p + 5]) for p in range(P)]
    base = 6 * P
    o.Vdc, o.xDC, o.xQ, o.xPLL, o.wte = (y[base + k] for k in range(5))
    vga = vg_mag * m.vg_scale + 0j
    vg = [vga, vga * A240, vga * A120][:P]
    o.m = [m.Kp_GCC * o.u[p] + o.x[p] for p in range(P)]
    o.vt = [o.m[p] * (o.Vdc / 2.0) for p in range(P)]
    o.v = [m.kv_g * vg[p] + m.kv_i * o.i[p] for p in range(P)]
    o.S = 0.5 * sum(o.vt[p] * o.i[p].conjugate() for p in range(P))
    o.S_PCC = 0.5 * sum(o.v[p] * o.i[p].conjugate() for p in range(P))
    o.Vrms = math.sqrt(sum(abs(v) ** 2 for v in o.v) / P) / math.sqrt(2.0)
    o.Ppv = ppv_calc(m, o.Vdc if m.opts.ppv_uses_vdc else m.Vdc_ref0, Sinsol)
    ia_ref = (o.xDC + m.Kp_DC * (Vdc_ref - o.Vdc)) + 1j * (o.xQ - m.Kp_Q * (Q_ref - o.S_PCC.imag))
    o.iref = [ia_ref, ia_ref * A240, ia_ref * A120][:P]
    # SRF-PLL: phasor -> time domain -> Park transform at angle wte (pvder utility_functions)
    wt = WBASE * t
    if P == 3:
        vt_dom = [(o.v[p] * cmath.exp(1j * wt)).real for p in range(3)]
        Us = (2.0 / 3.0) * (vt_dom[0] + vt_dom[1] * A120 + vt_dom[2] * A240) * cmath.exp(-1j * o.wte)
        o.vd = Us.real
    else:
        # single phase: alpha = v_a(t), beta = quadrature of the phasor (pvder single-phase PLL)
        valpha = (o.v[0] * cmath.exp(1j * wt)).real
        vbeta = (o.v[0] * cmath.exp(1j * wt)).imag
        o.vd = (complex(valpha, vbeta) * cmath.exp(-1j * o.wte)).real
    o.we = m.Kp_PLL * o.vd + o.xPLL + WBASE
    return o


def rhs(m: DERModel, y: Sequence[float], t: float, vg_mag: float, Sinsol: float,
        Q_ref: float, Vdc_ref: float) -> List[float]:
    """dy/dt -- restatement of pvder ``SolarPVDERThreePhase.ODE_model`` (published model).

    State order (pvder): per phase p: iR,iI,xR,xI,uR,uI ; then Vdc,xDC,xQ,xPLL,wte.
    """
    P = m.phases
    o = algebraic(m, y, t, vg_mag, Sinsol, Q_ref, Vdc_ref)
    winv = o.we if m.opts.winv_is_pll else WBASE
    wr = winv / WBASE
    out = [0.0] * m.n_states
    for p in range(P):
        i, x, u = o.i[p], o.x[p], o.u[p]
        di = (1.0 / m.Lf) * (-m.Rf * i - o.v[p] + o.vt[p]) - 1j * wr * i
        dx = m.Ki_GCC * u
        du = m.wp * (-u + o.iref[p] - i)
        if abs(o.m[p]) > m.m_limit:      # conditional-integration anti-windup
            dxR = 0.0 if sign(dx.real) == sign(x.real) else dx.real
            dxI = 0.0 if sign(dx.imag) == sign(x.imag) else dx.imag
            duR = 0.0 if sign(du.real) == sign(u.real) else du.real
            duI = 0.0 if sign(du.imag) == sign(u.imag) else du.imag
            dx = complex(dxR, dxI)
            du = complex(duR, duI)
        out[6 * p + 0], out[6 * p + 1] = di.real, di.imag
        out[6 * p + 2], out[6 * p + 3] = dx.real, dx.imag
        out[6 * p + 4], out[6 * p + 5] = du.real, du.imag
    base = 6 * P
    out[base + 0] = (o.Ppv - o.S.real) / (o.Vdc * m.C)
    dxDC = m.Ki_DC * (Vdc_ref - o.Vdc)
    dxQ = -m.Ki_Q * (Q_ref - o.S_PCC.imag)
    if abs(o.iref[0]) > m.iref_limit:
        if sign(dxDC) == sign(o.xDC):
            dxDC = 0.0
        if sign(dxQ) == sign(o.xQ):
            dxQ = 0.0
    out[base + 1] = dxDC
    out[base + 2] = dxQ
    out[base + 3] = m.Ki_PLL * o.vd
    out[base + 4] = o.we
    return out


def jac_fd(m: DERModel, y, t, vg_mag, Sinsol, Q_ref, Vdc_ref) -> np.ndarray:
    """Jacobian d rhs/dy by central differences (stand-in for pvder's analytic ``jac_ODE_model``;
    it only steers LSODA's Newton iteration, the converged solution does not depend on it)."""
    n = m.n_states
    J = np.empty((n, n))
    y = np.asarray(y, dtype=float)
    for k in range(n):
        h = 1e-7 * max(1.0, abs(y[k]))
        yp = y.copy(); yp[k] += h
        ym = y.copy(); ym[k] -= h
        J[:, k] = (np.asarray(rhs(m, yp, t, vg_mag, Sinsol, Q_ref, Vdc_ref))
                   - np.asarray(rhs(m, ym, t, vg_mag, Sinsol, Q_ref, Vdc_ref))) / (2 * h)
    return J


# --------------------------------------------------------------------------------------------
# Steady-state initialisation (steadyStateInitialization=True, PVDER_env.py:378)
# --------------------------------------------------------------------------------------------

def steady_state(m: DERModel, vg_mag: float = 1.0, Sinsol: Optional[float] = None,
                 Q_ref: float = 0.0) -> np.ndarray:
    """Operating point with Re S = Ppv(Vdc_ref), Im S_PCC = Q_ref, di/dt = 0 at w = wbase.

    pvder minimises the same residuals with Nelder-Mead from the ``steadystate_values`` seeds;
    here the zero is found with a Newton iteration in fp64 (documented deviation: exact root).
    Then x_p = m_p, u_p = 0, xDC = Re ia, xQ = Im ia, Vdc = Vdc_ref, xPLL = 0, wte = wte0.
    """
    from scipy.optimize import fsolve
    P = m.phases
    Sinsol = m.Sinsol0 if Sinsol is None else Sinsol
    Vdc = m.Vdc_ref0
    Ppv = ppv_calc(m, Vdc, Sinsol)
    vga = vg_mag * m.vg_scale + 0j

    def resid(z):
        ia = complex(z[0], z[1])
        va = m.kv_g * vga + m.kv_i * ia
        vta = va + (m.Rf + 1j * m.Lf) * ia      # di/dt = 0 with winv = wbase
        St = 0.5 * P * vta * ia.conjugate()
        Sp = 0.5 * P * va * ia.conjugate()
        return [St.real - Ppv, Sp.imag - Q_ref]

    z, info, ier, msg = fsolve(resid, [m.seeds[2], m.seeds[3]], full_output=True, xtol=1e-14)
    if ier != 1 and max(abs(r) for r in resid(z)) > 1e-13:       # (ier 3 "xtol too small" AT the root is not a failure)
        raise RuntimeError("steady-state solve failed: " + msg)
    ia = complex(z[0], z[1])
    va = m.kv_g * vga + m.kv_i * ia
    ma = (va + (m.Rf + 1j * m.Lf) * ia) / (Vdc / 2.0)
    rot = [1.0, A240, A120][:P]
    y0 = []
    for p in range(P):
        ip, mp = ia * rot[p], ma * rot[p]
        y0 += [ip.real, ip.imag, mp.real, mp.imag, 0.0, 0.0]
    y0 += [Vdc, ia.real, ia.imag, 0.0, m.opts.wte0]
    return np.asarray(y0, dtype=np.float64)


# --------------------------------------------------------------------------------------------
# Events  (SimulationEvents.create_random_events / grid_events / solar_events, published model;
#          spec and call site: PVDER_env.py:60-61, :400-411)
# --------------------------------------------------------------------------------------------

def splitmix64(x: int) -> int:
    """Counter-based hash shared bit-for-bit with the device event generator."""
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def counter_uniform(seed: int, env_id: int, episode: int, k: int, lane: int) -> float:
    """U[0,1) with 24 bits, keyed (seed, env, episode, event k, lane) -- same as the kernel."""
    key = splitmix64(seed & ((1 << 64) - 1))
    key = splitmix64(key ^ (env_id & 0xFFFFFFFF) ^ ((episode & 0xFFFFFFFF) << 32))
    key = splitmix64(key ^ ((k & 0xFFFFFFFF) << 8) ^ (lane & 0xFF))
    return (key >> 40) * (1.0 / 16777216.0)


@dataclass
class EventTable:
    """Piecewise-constant inputs: value ``vgrid[j]`` / ``sinsol[j]`` holds from event j's time
    (entry 0 = nominal before the first event).  Event j>=1 happens at ``t0 + (j-1)*dt``."""
    t0: float
    dt: float
    vgrid: np.ndarray      # [K+1]
    sinsol: np.ndarray     # [K+1]

    def index(self, t: float, inclusive: bool = True) -> int:
        K = len(self.vgrid) - 1
        if K == 0:
            return 0
        q = (t - self.t0) / self.dt
        j = math.floor(q + 1e-9) + 1 if inclusive else math.ceil(q - 1e-9)
        return max(0, min(K, j))

    def at(self, t: float, inclusive: bool = True) -> Tuple[float, float]:
        j = self.index(t, inclusive)
        return float(self.vgrid[j]), float(self.sinsol[j])


def event_times(spec: dict) -> np.ndarray:
    """np.arange(start, stop, step) on the *voltage* entry even for insolation events -- the
    quirk of PVDER_env.py:408-410."""
    v = spec["voltage"]
    return np.arange(v["t_events_start"], v["t_events_stop"], v["t_events_step"])


def make_event_table(spec: dict, sinsol0: float = 100.0, rng=None, counter_key=None) -> EventTable:
    """One event per grid time: type drawn from the ENABLEd list, value U[min,max]
    (create_random_events).  ``rng``: a ``random.Random`` (reference style, PVDER_env.py:15) or
    ``counter_key=(seed, env_id, episode)`` for the device-identical counter generator."""
    times = event_times(spec)
    K = len(times)
    enabled = [k for k in spec if spec[k]["ENABLE"]]      # dict order: insolation, voltage (:60-61)
    vg = np.ones(K + 1)
    si = np.full(K + 1, sinsol0)
    for j in range(1, K + 1):
        vg[j], si[j] = vg[j - 1], si[j - 1]
        if not enabled:
            continue
        if counter_key is not None:
            u_type = counter_uniform(*counter_key, j, 0)
            u_val = counter_uniform(*counter_key, j, 1)
            et = enabled[min(int(u_type * len(enabled)), len(enabled) - 1)]
            # fp32 arithmetic exactly as on the device
            lo, hi = np.float32(spec[et]["min"]), np.float32(spec[et]["max"])
            val = float(np.float32(lo + (hi - lo) * np.float32(u_val)))
        else:
            et = rng.choice(enabled)
            val = rng.uniform(spec[et]["min"], spec[et]["max"])
        if et == "voltage":
            vg[j] = val
        else:
            si[j] = val
    v = spec["voltage"]
    return EventTable(t0=float(v["t_events_start"]), dt=float(v["t_events_step"]), vgrid=vg, sinsol=si)


# --------------------------------------------------------------------------------------------
# Analytic Jacobian (the reference passes one to LSODA: jacFlag=True, PVDER_env.py:386)
# --------------------------------------------------------------------------------------------
_JAC_CACHE: Dict[str, object] = {}


def analytic_jacobian(m: DERModel):
    """d rhs/dy of the un-saturated model as a compiled function J(y, t, vg, si, qref, vdcref),
    derived symbolically (sympy) from the same equations as :func:`rhs` and lambdified with common
    sub-expression elimination.  pvder ships a hand-written ``jac_ODE_model``; inside an anti-windup
    branch callers fall back to :func:`jac_fd` (a Jacobian only steers LSODA's Newton iteration)."""
    if m.der_id in _JAC_CACHE:
        return _JAC_CACHE[m.der_id]
    import sympy as sp
    P = m.phases
    n = m.n_states
    ys = sp.symbols("y0:%d" % n, real=True)
    t, vgm, sinsol, qref, vdcref = sp.symbols("t vg si qref vdcref", real=True)
    b = 6 * P
    Vdc, xDC, xQ, xPLL, wte = ys[b], ys[b + 1], ys[b + 2], ys[b + 3], ys[b + 4]
    rot = [(1.0, 0.0), (-0.5, -math.sqrt(3) / 2), (-0.5, math.sqrt(3) / 2)][:P]
    vga = vgm * m.vg_scale
    kgR, kgI, kiR, kiI = m.kv_g.real, m.kv_g.imag, m.kv_i.real, m.kv_i.imag
    iR = [ys[6 * p] for p in range(P)]; iI = [ys[6 * p + 1] for p in range(P)]
    xR = [ys[6 * p + 2] for p in range(P)]; xI = [ys[6 * p + 3] for p in range(P)]
    uR = [ys[6 * p + 4] for p in range(P)]; uI = [ys[6 * p + 5] for p in range(P)]
    mR = [m.Kp_GCC * uR[p] + xR[p] for p in range(P)]; mI = [m.Kp_GCC * uI[p] + xI[p] for p in range(P)]
    vtR = [mR[p] * Vdc / 2 for p in range(P)]; vtI = [mI[p] * Vdc / 2 for p in range(P)]
    vR, vI = [], []
    for p in range(P):
        gR, gI = vga * rot[p][0], vga * rot[p][1]                      # vg_p
        vR.append(kgR * gR - kgI * gI + kiR * iR[p] - kiI * iI[p])
        vI.append(kgR * gI + kgI * gR + kiR * iI[p] + kiI * iR[p])
    ReS = sum(vtR[p] * iR[p] + vtI[p] * iI[p] for p in range(P)) / 2
    Qp = sum(vI[p] * iR[p] - vR[p] * iI[p] for p in range(P)) / 2
    Vact = Vdc * VDCBASE
    ppv = (m.Np * ISCR * (sinsol / 100.0) * Vact - m.Np * IRS * Vact * (sp.exp(m.pv_vt * Vdc) - 1)) / SBASE
    iarR = xDC + m.Kp_DC * (vdcref - Vdc)
    iarI = xQ - m.Kp_Q * (qref - Qp)
    wt = WBASE * t
    if P == 3:
        vt_dom = [vR[p] * sp.cos(wt) - vI[p] * sp.sin(wt) for p in range(3)]
        a120 = (-0.5, math.sqrt(3) / 2); a240 = (-0.5, -math.sqrt(3) / 2)
        UsR = (2.0 / 3.0) * (vt_dom[0] + vt_dom[1] * a120[0] + vt_dom[2] * a240[0])
        UsI = (2.0 / 3.0) * (vt_dom[1] * a120[1] + vt_dom[2] * a240[1])
        vd = UsR * sp.cos(wte) + UsI * sp.sin(wte)
    else:
        al = vR[0] * sp.cos(wt) - vI[0] * sp.sin(wt)
        be = vR[0] * sp.sin(wt) + vI[0] * sp.cos(wt)
        vd = al * sp.cos(wte) + be * sp.sin(wte)
    we = m.Kp_PLL * vd + xPLL + WBASE
    wr = we / WBASE if m.opts.winv_is_pll else 1.0
    f = [0] * n
    for p in range(P):
        rR = iarR * rot[p][0] - iarI * rot[p][1]
        rI = iarR * rot[p][1] + iarI * rot[p][0]
        f[6 * p] = (1.0 / m.Lf) * (-m.Rf * iR[p] - vR[p] + vtR[p]) + wr * iI[p]
        f[6 * p + 1] = (1.0 / m.Lf) * (-m.Rf * iI[p] - vI[p] + vtI[p]) - wr * iR[p]
        f[6 * p + 2] = m.Ki_GCC * uR[p]
        f[6 * p + 3] = m.Ki_GCC * uI[p]
        f[6 * p + 4] = m.wp * (-uR[p] + rR - iR[p])
        f[6 * p + 5] = m.wp * (-uI[p] + rI - iI[p])
    f[b] = (ppv - ReS) / (Vdc * m.C)
    f[b + 1] = m.Ki_DC * (vdcref - Vdc)
    f[b + 2] = -m.Ki_Q * (qref - Qp)
    f[b + 3] = m.Ki_PLL * vd
    f[b + 4] = we
    J = sp.Matrix(f).jacobian(sp.Matrix(ys))
    fn = sp.lambdify((ys, t, vgm, sinsol, qref, vdcref), J, modules="numpy", cse=True)

    def jac(y, tt, vg, si, q, v):
        return np.asarray(fn(tuple(float(v_) for v_ in y), float(tt), float(vg), float(si), float(q), float(v)),
                          dtype=np.float64)

    _JAC_CACHE[m.der_id] = jac
    return jac
