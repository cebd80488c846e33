// pvder_model.hpp -- device-side dynamic-phasor PV-DER model for gfx950 (one env per lane).
//
// Replaces, on the device, what the reference delegates to the third-party `pvder` package at
// gym_PVDER/envs/PVDER_env.py:166 (`self.sim.run_simulation()` -> SolarPVDERThreePhase.ODE_model,
// Grid.steady_state_model, PVModule.Ppv_calc) and the attribute reads of :238-299 / :535-540.
// Written from the model specification in DESIGN.md (section "Model"); real arithmetic, all
// constants folded on the host, no complex type, no library calls in the fp32 path.
//
// State order (pvder): per phase p: iR,iI,xR,xI,uR,uI ; then Vdc,xDC,xQ,xPLL,delta with
// delta = wte - wbase*t  (relative PLL angle: keeps fp32 exact for arbitrarily long episodes).
#pragma once
#include <hip/hip_runtime.h>

namespace pvder {

constexpr float kHalfSqrt3 = 0.86602540378443864676f;
constexpr double kTwoPi = 6.283185307179586476925;

// ---- per-launch constants: passed BY VALUE as kernel argument -> scalar registers (SGPRs);
// every lane of every wave reads the same values, so neither LDS nor VGPRs are spent on them.
template <typename T>
struct ModelK {
    T Rf, iLf, iC;            // filter R, 1/Lf, 1/C
    T kviR, kviI;             // v_p = A_p + kvi * i_p
    T kvi3R, kvi3I;           // kvi/3 (sequence components)
    T GR, GI;                 // A_a = G * vgrid_mag   (G = kvg * Vgridrated/Vbase)
    T KpG, KiG, KpDC, KiDC, KpQ, KiQ, wp, KpP, KiP;
    T mlim2, ilim2;           // limits squared
    T iwb;                    // 1/wbase
    T pvA;                    // Np*Iscr*Vdcbase/(100*Sbase): Iph term = pvA*Sinsol*Vdc
    T pvB;                    // Np*Irs*Vdcbase/Sbase
    T pvK;                    // q*Vdcbase/(k T A Ns)  [* log2(e) for the fp32 exp2 path]
                              // MODEL OPTION U5 (Ppv at the MPP voltage Vdcmpp0, PVDER_PAR_PPV_AT_MPP) is folded into these three
                              // on the host -- no further field, no further register: pvK = 0 flags it, and with
                              // pvA = -vdc0 pvA, pvB = -vdc0 pvB (1 - exp(pvK vdc0))  make_sub's  pv1 = pvA*Sinsol + pvB  IS
                              // -Ppv(Vdcmpp0, Sinsol), a constant per insolation level (ppv_of)
    T vrms_ref, vdc_ref0, sinsol0;
};

// ---- per sim-time-step (1/60 s) inputs: piecewise-constant events folded into constants
template <typename T, int PH>
struct SubK {
    T AR[PH], AI[PH];         // A_p = kvg*vg_p (grid event magnitude, 120-degree rotations)
    T pv1;                    // pvA*Sinsol + pvB
};

template <typename T> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ float exp_pv(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
    // angle given in turns (1 turn = 2 pi): v_sin_f32 / v_cos_f32 take turns directly
    static __device__ __forceinline__ void sincos_turns(float tr, float &s, float &c) {
        s = __builtin_amdgcn_sinf(tr);
        c = __builtin_amdgcn_cosf(tr);
    }
    static __device__ __forceinline__ float sqrt_(float x) { return __builtin_sqrtf(x); }
};
template <> struct Math<double> {
    static __device__ __forceinline__ double exp_pv(double x) { return exp(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ void sincos_turns(double tr, double &s, double &c) {
        sincospi(2.0 * tr, &s, &c);
    }
    static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
};

template <typename T, int PH>
__device__ __forceinline__ void make_sub(const ModelK<T> &P, T vgm, T sinsol, SubK<T, PH> &S) {
#pragma clang fp contract(off)       // (see outputs(): the same bits in every kernel)
    const T ar = P.GR * vgm, ai = P.GI * vgm;
    S.AR[0] = ar;
    S.AI[0] = ai;
    if constexpr (PH == 3) {
        const T h3 = (T)0.86602540378443864676;
        S.AR[1] = (T)-0.5 * ar + h3 * ai;   // A_a * e^{-j120}
        S.AI[1] = (T)-0.5 * ai - h3 * ar;
        S.AR[2] = (T)-0.5 * ar - h3 * ai;   // A_a * e^{+j120}
        S.AI[2] = (T)-0.5 * ai + h3 * ar;
    }
    S.pv1 = P.pvA * sinsol + P.pvB;
}

// PV array power (pu): max(0, Vdc*(pv1 - pvB*exp(pvK*Vdc)))
template <typename T>
__device__ __forceinline__ T ppv_of(const ModelK<T> &P, T pv1, T Vdc) {
#pragma clang fp contract(off)       // (reported as an observation: see outputs())
    const T e = Math<T>::exp_pv(P.pvK * Vdc);
    T p = Vdc * (pv1 - P.pvB * e);
    // model option U5 (wave-uniform, pvK == 0: see ModelK): pv1 = -Ppv(Vdcmpp0, Sinsol).  A select on a kernel argument that
    // is live here anyway -- the 23-state tail that calls this sits at the register limit
    p = P.pvK == (T)0 ? -pv1 : p;
    return p > (T)0 ? p : (T)0;
}

// dy/dt.  th2t = 2*wbase*t in turns (only matters when the currents are unbalanced).
// NPAR = number of identical phases the PH phases written out stand for: rhs<T, 1, 3> is the BALANCED
// three-phase model on its 11-state reduction (phase a + the shared states; phases b, c = phase a rotated)
template <typename T, int PH, int NPAR = 1>
__device__ __forceinline__ void rhs(const ModelK<T> &P, const SubK<T, PH> &S, const T *__restrict__ y,
                                    T th2t, T qref, T vdcref, T *__restrict__ dy) {
    constexpr int B = 6 * PH;
    const T Vdc = y[B], xDC = y[B + 1], xQ = y[B + 2], xPLL = y[B + 3], delta = y[B + 4];
    const T hV = (T)0.5 * Vdc;
    T vR[PH], vI[PH], mR[PH], mI[PH];
    T Pt = (T)0, Qp = (T)0;   // 2*Re(S)/Vdc... accumulated below
#pragma unroll
    for (int p = 0; p < PH; ++p) {
        const T iR = y[6 * p], iI = y[6 * p + 1];
        mR[p] = P.KpG * y[6 * p + 4] + y[6 * p + 2];
        mI[p] = P.KpG * y[6 * p + 5] + y[6 * p + 3];
        vR[p] = S.AR[p] + P.kviR * iR - P.kviI * iI;
        vI[p] = S.AI[p] + P.kviR * iI + P.kviI * iR;
        Pt += mR[p] * iR + mI[p] * iI;          // Re(m conj i)
        Qp += vI[p] * iR - vR[p] * iI;          // Im(v conj i)
    }
    const T ReS = (T)(0.5 * NPAR) * hV * Pt;    // Re S = 1/2 sum Re(vt conj i), vt = m Vdc/2
    const T Q = (T)(0.5 * NPAR) * Qp;           // Im S_PCC
    // current references
    const T iarR = xDC + P.KpDC * (vdcref - Vdc);
    const T qerr = qref - Q;
    const T iarI = xQ - P.KpQ * qerr;
    // SRF-PLL
    T vd;
    {
        T s1, c1;
        const T dturn = delta * (T)(1.0 / kTwoPi);
        Math<T>::sincos_turns(dturn, s1, c1);
        if constexpr (PH == 3) {
            const T sbR = y[6] + y[12], sbI = y[7] + y[13];
            const T dbR = y[6] - y[12], dbI = y[7] - y[13];
            const T bR = y[0] - (T)0.5 * sbR, bI = y[1] - (T)0.5 * sbI;
            const T h3 = (T)0.86602540378443864676;
            const T i1R = bR - h3 * dbI, i1I = bI + h3 * dbR;      // 3*I1
            const T i2R = bR + h3 * dbI, i2I = bI - h3 * dbR;      // 3*I2
            const T v1R = S.AR[0] + P.kvi3R * i1R - P.kvi3I * i1I;
            const T v1I = S.AI[0] + P.kvi3R * i1I + P.kvi3I * i1R;
            const T v2R = P.kvi3R * i2R - P.kvi3I * i2I;
            const T v2I = P.kvi3R * i2I + P.kvi3I * i2R;
            T s2, c2;
            Math<T>::sincos_turns(th2t + dturn, s2, c2);
            vd = v1R * c1 + v1I * s1 + v2R * c2 - v2I * s2;
        } else {
            vd = vR[0] * c1 + vI[0] * s1;
        }
    }
    const T dw = P.KpP * vd + xPLL;             // we - wbase
    const T wr = (T)1 + dw * P.iwb;             // winv/wbase
#pragma unroll
    for (int p = 0; p < PH; ++p) {
        const T iR = y[6 * p], iI = y[6 * p + 1];
        const T xR = y[6 * p + 2], xI = y[6 * p + 3];
        const T uR = y[6 * p + 4], uI = y[6 * p + 5];
        T rR, rI;                               // i_ref of this phase
        if (p == 0) {
            rR = iarR; rI = iarI;
        } else if (p == 1) {                    // * e^{-j120}
            rR = (T)-0.5 * iarR + (T)0.86602540378443864676 * iarI;
            rI = (T)-0.5 * iarI - (T)0.86602540378443864676 * iarR;
        } else {                                // * e^{+j120}
            rR = (T)-0.5 * iarR - (T)0.86602540378443864676 * iarI;
            rI = (T)-0.5 * iarI + (T)0.86602540378443864676 * iarR;
        }
        const T tR = mR[p] * hV - vR[p] - P.Rf * iR;
        const T tI = mI[p] * hV - vI[p] - P.Rf * iI;
        dy[6 * p] = P.iLf * tR + wr * iI;
        dy[6 * p + 1] = P.iLf * tI - wr * iR;
        T dxR = P.KiG * uR, dxI = P.KiG * uI;
        T duR = P.wp * (rR - iR - uR), duI = P.wp * (rI - iI - uI);
        if (mR[p] * mR[p] + mI[p] * mI[p] > P.mlim2) {   // conditional-integration anti-windup
            dxR = (dxR * xR > (T)0) ? (T)0 : dxR;
            dxI = (dxI * xI > (T)0) ? (T)0 : dxI;
            duR = (duR * uR > (T)0) ? (T)0 : duR;
            duI = (duI * uI > (T)0) ? (T)0 : duI;
        }
        dy[6 * p + 2] = dxR;
        dy[6 * p + 3] = dxI;
        dy[6 * p + 4] = duR;
        dy[6 * p + 5] = duI;
    }
    const T ppv = ppv_of(P, S.pv1, Vdc);
    dy[B] = (ppv - ReS) * Math<T>::rcp(Vdc) * P.iC;
    T dxDC = P.KiDC * (vdcref - Vdc);
    T dxQ = -P.KiQ * qerr;
    if (iarR * iarR + iarI * iarI > P.ilim2) {
        dxDC = (dxDC * xDC > (T)0) ? (T)0 : dxDC;
        dxQ = (dxQ * xQ > (T)0) ? (T)0 : dxQ;
    }
    dy[B + 1] = dxDC;
    dy[B + 2] = dxQ;
    dy[B + 3] = P.KiP * vd;
    dy[B + 4] = dw;
}

// Observation quantities at a state (PVDER.state, PVDER_env.py:531-542) + Vrms for the reward.
template <typename T, int PH>
__device__ __forceinline__ void outputs(const ModelK<T> &P, const SubK<T, PH> &S, const T *__restrict__ y,
                                        T &iaR, T &iaI, T &vaR, T &vaI, T &Ppcc, T &Qpcc, T &ppv,
                                        T &vrms) {
    // No fma contraction here: which product of  a*b + c*d  the compiler fuses depends on the code around
    // the call, and every kernel that reports an observation of the same state (per-step launch, fused
    // rollout, 23-state tail, reset) must report the same bits.
#pragma clang fp contract(off)
    constexpr int B = 6 * PH;
    T Pp = (T)0, Qp = (T)0, v2 = (T)0;
#pragma unroll
    for (int p = 0; p < PH; ++p) {
        const T iR = y[6 * p], iI = y[6 * p + 1];
        const T vR = S.AR[p] + P.kviR * iR - P.kviI * iI;
        const T vI = S.AI[p] + P.kviR * iI + P.kviI * iR;
        Pp += vR * iR + vI * iI;
        Qp += vI * iR - vR * iI;
        v2 += vR * vR + vI * vI;
        if (p == 0) { vaR = vR; vaI = vI; }
    }
    iaR = y[0];
    iaI = y[1];
    Ppcc = (T)0.5 * Pp;
    Qpcc = (T)0.5 * Qp;
    ppv = ppv_of(P, S.pv1, y[B]);
    vrms = Math<T>::sqrt_(v2 * (T)(1.0 / PH) * (T)0.5);
}

}  // namespace pvder
