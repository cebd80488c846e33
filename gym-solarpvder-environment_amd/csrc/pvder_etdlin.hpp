// pvder_etdlin.hpp -- device pieces of the linearised exponential integrator ("etdlin") for gfx950.
//
// dy/dt = f(y) = L y + N(y) with L the 11x11 Jacobian of the balanced model at its nominal operating
// point (host: gym_PVDER_amd/etd_linear.py).  Cox & Matthews' ETD3RK is evaluated in "f-form":
//     a  = y + M21 f(y)
//     b  = y + M31 f(y) + M32 f(a)
//     y+ = y + M41 f(y) + M42 f(a) + M43 f(b)
// -- an explicit Runge-Kutta step whose coefficients are DENSE 11x11 matrices (functions of hL,
// folded on the host in fp64).  The exponential carries the whole linear response (stiff current
// loop, PLL pole, DC link, outer PI loops), so TWO steps cover an env step: 1/(60*micro) s behind the
// input jump (new action / event), then one over the remainder (DESIGN.md "etdlin").
//
// This replaces, on the device, the LSODA + analytic-Jacobian integration the reference delegates to
// pvder/SciPy (gym_PVDER/envs/PVDER_env.py:166, :384-386) for envs in normal operation; envs near a
// limit (anti-windup) or off balance take the 23-state path (etdrk4_full_tail).
#pragma once
#include "pvder_model_pk.hpp"

namespace pvder {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f lds_cv4;

constexpr int kLinN = 11;                 // states of the balanced reduction
constexpr int kLinRows = 12;              // matrix rows padded to pairs
constexpr int kLinMatFloats = kLinN * kLinRows;          // 132: column-major, column k = 12 floats
constexpr int kLinMats = 6;               // M21 | M31 M32 | M41 M42 M43
constexpr int kLinSetFloats = kLinMats * kLinMatFloats;  // 792 floats = 3168 B per step size
constexpr int kLinMaxSets = 8;
enum { LM21 = 0, LM31, LM32, LM41, LM42, LM43 };

// Full right-hand side of the balanced 11-state model as six pairs (the state layout of the lin
// path): F[0] = di, F[1] = dx, F[2] = du, F[3] = (dVdc, dxDC), F[4] = (dxQ, dxPLL), F[5] = (ddelta, 0).
// Same equations as pvder_model.hpp::rhs for balanced phases, without the anti-windup branches
// (the caller defers envs whose |m| / |iref| come near a limit: m2 = |m_a|^2, ir2 = |iref_a|^2).
template <int PH>
__device__ __forceinline__ void bal_f(const ModelK<float> &P, const PkConst &K, const SubPk &S, const v2 *__restrict__ y,
                                      float qref, float vdcref, v2 kpo, v2 kio, v2 *__restrict__ F, float &m2,
                                      float &ir2) {
    const v2 ia = y[0], xa = y[1], ua = y[2];
    const float Vdc = y[3].x, xPLL = y[4].y, delta = y[5].x;
    const v2 ma = fma2(bc2(P.KpG), ua, xa);
    const float Pa = __builtin_fmaf(ma.y, ia.y, ma.x * ia.x);             // Re(m conj i) of one of PH identical phases
    const v2 tq = fma2(bc2((float)PH * K.qk), ia, bc2(0.5f * PH) * S.Asw[0]);
    const float Q = __builtin_fmaf(tq.y, ia.y, tq.x * ia.x);             // Im S_PCC
    // outer loops, packed: e = (Vdc_ref - Vdc, Q_ref - Q); iref_a = (xDC, xQ) + (Kp_DC, -Kp_Q) e
    const v2 e = mk2(vdcref - Vdc, qref - Q);
    const v2 iar = fma2(kpo, e, mk2(y[3].y, y[4].x));
    const v2 dpi = kio * e;                                               // (dxDC, dxQ)
    ir2 = __builtin_fmaf(iar.y, iar.y, iar.x * iar.x);
    float s1, c1;
    Math<float>::sincos_turns(delta * (float)(1.0 / kTwoPi), s1, c1);
    v2 V1 = fma2(bc2(P.kviR), ia, S.A0);
    V1 = fma2(mk2(-P.kviI, P.kviI), swp(ia), V1);                         // v_a (balanced: V2 = 0)
    const float vd = __builtin_fmaf(V1.y, s1, V1.x * c1);
    const float dw = __builtin_fmaf(P.KpP, vd, xPLL);                      // we - wbase
    // di/dt = cm m - A/Lf - Re(Z) i + (Im(Z) + winv/wbase) (iI, -iR),  Z = (kvi + Rf)/Lf
    const float zw = K.zi0 + 1.f + dw * P.iwb;
    v2 n = fma2(bc2(K.hiLf * Vdc), ma, S.D[0]);
    n = fma2(bc2(-K.ZR), ia, n);
    F[0] = fma2(mk2(zw, -zw), swp(ia), n);
    F[1] = bc2(P.KiG) * ua;
    F[2] = bc2(P.wp) * (iar - ia - ua);
    // dVdc = (Ppv - Re S)/(Vdc C), Ppv = max(0, Vdc g(Vdc)), Re S = Vdc PH Pa/4  ->  no division
    const float g = __builtin_fmaf(-P.pvB, Math<float>::exp_pv(P.pvK * Vdc), S.pv1);
    const float dVdc = __builtin_fmaf(-0.25f * PH * P.iC, Pa, P.iC * fmaxf(g, 0.f));
    F[3] = mk2(dVdc, dpi.x);
    F[4] = mk2(dpi.y, P.KiP * vd);
    F[5] = mk2(dw, 0.f);
    m2 = __builtin_fmaf(ma.y, ma.y, ma.x * ma.x);
}

// acc[j] += M_j f  for NM matrices that share the input vector f (six pairs; entry 11 is padding).
// `set` = LDS address of the lane's coefficient set; mat[j] = matrix number inside the set.  Per
// column k: 3 ds_read_b128 (12 rows) and 6 v_pk_fma_f32 per matrix, f_k broadcast to both halves by
// op_sel.  The accumulator chains of different rows / matrices are independent.
template <int NM>
__device__ __forceinline__ void lin_matvec(const lds_cv4 *set, const int (&mat)[NM], const v2 *__restrict__ f,
                                           v2 (&acc)[NM][6]) {
#pragma unroll
    for (int k = 0; k < kLinN; ++k) {
        const v2 fb = bc2((k & 1) ? f[k >> 1].y : f[k >> 1].x);
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const lds_cv4 *col = set + (mat[j] * kLinMatFloats + k * kLinRows) / 4;
            const v4f c0 = col[0], c1 = col[1], c2 = col[2];
            acc[j][0] = fma2(mk2(c0.x, c0.y), fb, acc[j][0]);
            acc[j][1] = fma2(mk2(c0.z, c0.w), fb, acc[j][1]);
            acc[j][2] = fma2(mk2(c1.x, c1.y), fb, acc[j][2]);
            acc[j][3] = fma2(mk2(c1.z, c1.w), fb, acc[j][3]);
            acc[j][4] = fma2(mk2(c2.x, c2.y), fb, acc[j][4]);
            acc[j][5] = fma2(mk2(c2.z, c2.w), fb, acc[j][5]);
        }
    }
}

}  // namespace pvder
