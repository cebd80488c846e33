// pvder_model_pk.hpp -- three-phase fp32 RHS written for gfx950's packed fp32 VALU
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 lanes per instruction).
//
// With 65 536 envs there is exactly ONE wave per SIMD, and a single wave issues one VALU
// instruction per ~4 cycles whatever its width (MI355X_MICROARCH.md, "vector-instruction ISSUE
// cost") -- so the lever is instructions per RK4 step, not flops.  Every complex quantity is a
// float2 (re, im); constants are folded so that the filter-current equation is three packed FMAs:
//
//   di_p/dt = cm*m_p + D_p - Z*i_p,   cm = Vdc/(2 Lf),  D_p = -A_p/Lf,  Z = (kvi+Rf)/Lf + j*winv/wbase
//   du_p/dt = wp*iref_p - wp*(i_p + u_p)
//   Im S_PCC = 1/2 sum Im(A_p conj i_p) + 1/2 Im(kvi) sum |i_p|^2        (v_p never materialised)
//   V1 = A_a + (kvi/3)(3 I1),  V2 = (kvi/3)(3 I2)                        (sequence components)
//
// Same model as pvder_model.hpp::rhs (which stays the generic fp64 / single-phase form); the two
// are checked against each other and against the oracle by the -m gpu tests.
#pragma once
#include "pvder_model.hpp"

namespace pvder {

typedef float v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2 mk2(float a, float b) { v2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2 bc2(float a) { v2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2 swp(v2 z) { return __builtin_shufflevector(z, z, 1, 0); }
// j*z = (-z.y, z.x)
__device__ __forceinline__ v2 jmul(v2 z) { return mk2(-z.y, z.x); }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }

// pair layout of the 23 states: 0..8 = (i,x,u) of phases a,b,c as (re,im); 9 = (Vdc,xDC);
// 10 = (xQ,xPLL); 11 = (delta, 0)
constexpr int NP = 12;

struct SubPk {
    v2 A0;          // A_a
    v2 Asw[3];      // (Im A_p, -Re A_p): dot with i_p gives Im(A_p conj i_p)
    v2 D[3];        // -A_p/Lf
    float pv1;
};

__device__ __forceinline__ void make_sub_pk(const ModelK<float> &P, float vgm, float sinsol, SubPk &S) {
    const float ar = P.GR * vgm, ai = P.GI * vgm;
    const float h3 = kHalfSqrt3;
    const float br = -0.5f * ar + h3 * ai, bi = -0.5f * ai - h3 * ar;   // A_a e^{-j120}
    const float cr = -0.5f * ar - h3 * ai, ci = -0.5f * ai + h3 * ar;   // A_a e^{+j120}
    S.A0 = mk2(ar, ai);
    S.Asw[0] = mk2(ai, -ar); S.Asw[1] = mk2(bi, -br); S.Asw[2] = mk2(ci, -cr);
    S.D[0] = mk2(-P.iLf * ar, -P.iLf * ai);
    S.D[1] = mk2(-P.iLf * br, -P.iLf * bi);
    S.D[2] = mk2(-P.iLf * cr, -P.iLf * ci);
    S.pv1 = P.pvA * sinsol + P.pvB;
}

struct PkConst {         // wave-uniform, derived once per launch from ModelK (SGPRs)
    float ZR, zi0, hiLf, qk;     // Re Z, iLf*kviI, iLf/2, kviI/2
};
__device__ __forceinline__ PkConst make_pkconst(const ModelK<float> &P) {
    PkConst c;
    c.ZR = P.iLf * (P.kviR + P.Rf);
    c.zi0 = P.iLf * P.kviI;
    c.hiLf = 0.5f * P.iLf;
    c.qk = 0.5f * P.kviI;
    return c;
}

__device__ __forceinline__ void rhs_pk(const ModelK<float> &P, const PkConst &K, const SubPk &S,
                                       const v2 *__restrict__ y, float th2t, float qref, float vdcref,
                                       v2 *__restrict__ dy) {
    const float Vdc = y[9].x, xDC = y[9].y, xQ = y[10].x, xPLL = y[10].y, delta = y[11].x;
    const v2 ia = y[0], xa = y[1], ua = y[2], ib = y[3], xb = y[4], ub = y[5], ic = y[6], xc = y[7], uc = y[8];
    const v2 kpg = bc2(P.KpG);
    const v2 ma = fma2(kpg, ua, xa), mb = fma2(kpg, ub, xb), mc = fma2(kpg, uc, xc);
    // packed accumulators: sum Re(m conj i), sum Im(A conj i), sum |i|^2 (lanes added at the end)
    v2 aP = ma * ia; aP = fma2(mb, ib, aP); aP = fma2(mc, ic, aP);
    v2 aQ = S.Asw[0] * ia; aQ = fma2(S.Asw[1], ib, aQ); aQ = fma2(S.Asw[2], ic, aQ);
    v2 aI = ia * ia; aI = fma2(ib, ib, aI); aI = fma2(ic, ic, aI);
    const float Pt = aP.x + aP.y;
    const float Q = 0.5f * (aQ.x + aQ.y) + K.qk * (aI.x + aI.y);       // Im S_PCC
    const float qerr = qref - Q;
    const float iarR = xDC + P.KpDC * (vdcref - Vdc);
    const float iarI = xQ - P.KpQ * qerr;
    // SRF-PLL on sequence components
    float s1, c1, s2, c2;
    const float dturn = delta * (float)(1.0 / kTwoPi);
    Math<float>::sincos_turns(dturn, s1, c1);
    Math<float>::sincos_turns(th2t + dturn, s2, c2);
    const v2 sb = ib + ic, db = ib - ic;
    const v2 base = fma2(bc2(-0.5f), sb, ia);
    const v2 jh3 = mk2(-kHalfSqrt3, kHalfSqrt3);      // j*s3 as a lane-sign pattern on the swapped operand
    const v2 rdb = swp(db) * jh3;
    const v2 i1 = base + rdb, i2 = base - rdb;                          // 3*I1, 3*I2
    const v2 jk3 = mk2(-P.kvi3I, P.kvi3I);
    v2 V1 = fma2(bc2(P.kvi3R), i1, S.A0); V1 = fma2(jk3, swp(i1), V1);
    // conj(V2) = conj(kvi/3) conj(3 I2), formed directly so that no lane needs a sign flip
    v2 V2c = mk2(P.kvi3R, -P.kvi3R) * i2; V2c = fma2(bc2(-P.kvi3I), swp(i2), V2c);
    const v2 t1 = V1 * mk2(c1, s1);
    const v2 t2 = fma2(V2c, mk2(c2, s2), t1);         // Re(V1 e^{-j d}) + Re(conj(V2) e^{-j(2th+d)})
    const float vd = t2.x + t2.y;
    const float dw = P.KpP * vd + xPLL;               // we - wbase
    const float wr = 1.0f + dw * P.iwb;
    // filter currents
    const v2 cm = bc2(K.hiLf * Vdc);
    const v2 zv = fma2(bc2(wr), mk2(1.f, -1.f), mk2(K.zi0, -K.zi0));   // (Im Z, -Im Z)
    const v2 nzr = bc2(-K.ZR);
    v2 da = fma2(cm, ma, S.D[0]); da = fma2(nzr, ia, da); da = fma2(zv, swp(ia), da);
    v2 dbb = fma2(cm, mb, S.D[1]); dbb = fma2(nzr, ib, dbb); dbb = fma2(zv, swp(ib), dbb);
    v2 dc = fma2(cm, mc, S.D[2]); dc = fma2(nzr, ic, dc); dc = fma2(zv, swp(ic), dc);
    // current controller
    const v2 wra = mk2(P.wp * iarR, P.wp * iarI);
    const v2 hw = bc2(-0.5f) * wra;
    const v2 jw = swp(wra) * jh3;                     // j*s3*wra
    const v2 wrb = hw - jw;                           // wra * e^{-j120}
    const v2 wrc = hw + jw;                           // wra * e^{+j120}
    const v2 nwp = bc2(-P.wp), kig = bc2(P.KiG);
    v2 dua = fma2(nwp, ia + ua, wra), dub = fma2(nwp, ib + ub, wrb), duc = fma2(nwp, ic + uc, wrc);
    v2 dxa = kig * ua, dxb = kig * ub, dxc = kig * uc;
    const v2 mma = ma * ma, mmb = mb * mb, mmc = mc * mc;
    const float na = mma.x + mma.y, nb = mmb.x + mmb.y, nc = mmc.x + mmc.y;
    if (fmaxf(fmaxf(na, nb), nc) > P.mlim2) {         // rare: conditional-integration anti-windup
        if (na > P.mlim2) {
            dxa.x = (dxa.x * xa.x > 0.f) ? 0.f : dxa.x; dxa.y = (dxa.y * xa.y > 0.f) ? 0.f : dxa.y;
            dua.x = (dua.x * ua.x > 0.f) ? 0.f : dua.x; dua.y = (dua.y * ua.y > 0.f) ? 0.f : dua.y;
        }
        if (nb > P.mlim2) {
            dxb.x = (dxb.x * xb.x > 0.f) ? 0.f : dxb.x; dxb.y = (dxb.y * xb.y > 0.f) ? 0.f : dxb.y;
            dub.x = (dub.x * ub.x > 0.f) ? 0.f : dub.x; dub.y = (dub.y * ub.y > 0.f) ? 0.f : dub.y;
        }
        if (nc > P.mlim2) {
            dxc.x = (dxc.x * xc.x > 0.f) ? 0.f : dxc.x; dxc.y = (dxc.y * xc.y > 0.f) ? 0.f : dxc.y;
            duc.x = (duc.x * uc.x > 0.f) ? 0.f : duc.x; duc.y = (duc.y * uc.y > 0.f) ? 0.f : duc.y;
        }
    }
    dy[0] = da; dy[1] = dxa; dy[2] = dua;
    dy[3] = dbb; dy[4] = dxb; dy[5] = dub;
    dy[6] = dc; dy[7] = dxc; dy[8] = duc;
    // DC link, outer PI loops, PLL
    const float ppv = ppv_of(P, S.pv1, Vdc);
    const float ReS = 0.25f * Vdc * Pt;               // 1/2 * Vdc/2 * sum Re(m conj i)
    const float dVdc = (ppv - ReS) * Math<float>::rcp(Vdc) * P.iC;
    float dxDC = P.KiDC * (vdcref - Vdc);
    float dxQ = -P.KiQ * qerr;
    if (iarR * iarR + iarI * iarI > P.ilim2) {
        dxDC = (dxDC * xDC > 0.f) ? 0.f : dxDC;
        dxQ = (dxQ * xQ > 0.f) ? 0.f : dxQ;
    }
    dy[9] = mk2(dVdc, dxDC);
    dy[10] = mk2(dxQ, P.KiP * vd);
    dy[11] = mk2(dw, 0.f);
}

}  // namespace pvder
