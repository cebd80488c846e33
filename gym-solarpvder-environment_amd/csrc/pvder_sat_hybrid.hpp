// pvder_sat_hybrid.hpp -- the 23-state model AT THE DUTY-CYCLE LIMIT, on the slow time scale (round 6).
//
// Reference path: PVDER.step -> action_calc lowers Vdc_ref by 0.02 n (PVDER_env.py:218-221, :229), held for a whole episode by the
// reference's evaluation driver (:480-487), integrated by sim.run_simulation() (:166): the duty cycle reaches m_limit and the
// current controller's conditional-integration anti-windup (pvder_model.hpp::rhs, the `mlim2` branch) takes over.  There the
// right-hand side is discontinuous and the converged solution (oracle/pvder_bridge_filippov.h) is a HYBRID one: per phase p and
// axis c the low-pass state u_c may only move towards zero while |m_p| >= m_limit -- with D_c = iref_c - i_c - u_c it is HELD while
// h_c = D_c sgn(u_c) > 0 --, the phase's current loop is open while it is held ("flight": rates of 1-10 /s, nothing stiff), and
// when h_c returns to zero the axis is released, the closed loop (omega_0 = 5.9e5 rad/s, zeta = 0.17) swings it inward for half a
// damped period (5.4 us) and the anti-windup holds it again at the first extremum: a bouncing ball with restitution 0.58 whose
// other axis follows the circle |m_p| = m_limit like a ratchet.  A step that holds "the mode found at its start" (rounds 3-5)
// does not converge to that solution under refinement (profiles/r05_saturated_step_prototypes.txt: 5e-2 ... 1e-1).
//
// This file is the statement of tools/sat_hybrid_proto.py::DevHybrid (fp64 prototype against the converged golden: <= 4.3e-3 on all
// 160 steps of the four action-4 episodes, medians <= 8e-5; profiles/r06_sat_hybrid_proto.txt) in C++ for host AND device:
//   * macro steps: Hochbruck-Ostermann ETDRK4 (5 stages) over a sim time step (the first one behind an action in two: 1/32 + 31/32);
//     the only stiff blocks are the CLOSED current loops, per axis the real 2 x 2 pair (i_c, u_c) with the constant
//     L2 = [[a, g0 K], [-wp, -wp]]; every phi_k(h L2) = c1 I + c2 L2 from ONE complex exponential -- exact for any h, no tables --,
//     which is what re-taking a step up to an event needs; held axes, the axes' cross coupling (7 /s) and the outer loops are in N;
//   * events inside a step: release of a held axis (h_c through zero: located on the Hermite cubic of h_c, two Newton corrections),
//     entry of a free phase (|m_p| through the limit: secant / bisection on values), a regulating axis turning outward or through zero;
//   * every discrete happening is RESOLVED by `settle`: a micro-integration (RK4, 0.25 us, finer while something moves fast) of the
//     phase's fast pair (D_p, u_p) -- four real states, the slow ones frozen, de/dt linearised in u -- under the oracle's own rules
//     (regions IN / OUT / SLIDE held over a micro step, G masks, Filippov's alpha, components put on zero) until nothing fast is left;
//   * modes are re-derived from the state at the start of an env step: no discrete state between launches.
// Arithmetic: fp64 (an event function is the difference iref_c - i_c - u_c of O(1) quantities at the 1e-9 level: fp32 rows in, fp32
// rows out, fp64 in between).  Used by the 23-state tail of the step kernels for envs at or near the limit (pvder_kernels.hip) and,
// compiled for the host, by tests/test_sat_hybrid_host.py against the same golden.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PVDER_HD __host__ __device__
#else
#define PVDER_HD
#endif
// The step functions below are CALLED on the device, not inlined: advance() holds six etd_step and four settle sites, env_step three
// advance sites -- inlined, env_step was 2 MB of code for a 64 KB instruction cache and ran at the speed of its instruction fetches.
#if defined(__HIP_DEVICE_COMPILE__)
#define PVDER_HYB_CALL __attribute__((noinline))
#else
#define PVDER_HYB_CALL
#endif

namespace pvder {
namespace hyb {

// (host builds of tools/hybrid_cost_model.py count calls: what an env step costs a lane is the number of these)
#if defined(PVDER_HYB_COUNT) && !defined(__HIP_DEVICE_COMPILE__)
extern long g_count[8];
#define PVDER_HYB_CNT(i) (++::pvder::hyb::g_count[i])
#else
#define PVDER_HYB_CNT(i) ((void)0)
#endif

// the model's constants (pvder_model.hpp::ModelK; pvK in the exp2 form).  TK = float on the device: they ARE the step kernels' fp32
// constants, and 23 doubles that every function reads are 184 bytes of private memory the kernel cannot afford (see Arena)
template <typename TK>
struct HybKT {
    TK Rf, iLf, iC, kviR, kviI, GR, GI, KpG, KiG, KpDC, KiDC, KpQ, KiQ, wp, KpP, KiP, mlim2, ilim2, iwb, pvA, pvB, pvK, vdc_ref0;
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef HybKT<float> HybK;
#else
typedef HybKT<double> HybK;
#endif

struct Sub {                   // per sim time step: grid voltage phasors and the PV constant
    double AR[3], AI[3], pv1;
};

struct Stats {                 // what an env step did (tests, counters)
    int settles, micro, events, capped;
};

// Every 23-vector the scheme needs at one time, as slots of ONE block the caller provides: on the device a lane's own stretch of LDS
// (step_sat_hybrid_kernel: the 64 envs of a block per workgroup -- on four 16-lane waves --, 2.4 KB each -- private arrays of this size would give the kernel a 6 KB scratch
// frame per lane, which the runtime can only serve by allocating scratch per dispatch: 0.2 ms per LAUNCH), on the host an array.
enum { A_N1 = 0, A_N2, A_N4, A_N5, A_YT, A_E2Y, A_EY, A_TMP, A_F0, A_F1, A_Y1, A_YT2, A_COUNT };
constexpr int kArenaDoubles = A_COUNT * 23;
struct Arena {
    double *base;
    PVDER_HD double *operator[](int k) const { return base + 23 * k; }
};

constexpr double kH3 = 0.86602540378443864676;
constexpr double kTwoPiD = 6.283185307179586476925;

PVDER_HD inline void make_sub(const HybK &P, double vgm, double sinsol, Sub &S) {
    const double ar = P.GR * vgm, ai = P.GI * vgm;
    S.AR[0] = ar; S.AI[0] = ai;
    S.AR[1] = -0.5 * ar + kH3 * ai; S.AI[1] = -0.5 * ai - kH3 * ar;
    S.AR[2] = -0.5 * ar - kH3 * ai; S.AI[2] = -0.5 * ai + kH3 * ar;
    S.pv1 = P.pvA * sinsol + P.pvB;
}

// The FREE field (no duty-cycle anti-windup; the current limit's freeze of the outer integrators stays): dy/dt of the 23 states in
// pvder order, th2t = 2 wbase t in turns.  Same algebra as pvder_model.hpp::rhs<T, 3>.  Also returns what the event functions need.
struct Alg {
    double vR[3], vI[3], iarR, iarI;
};
PVDER_HD PVDER_HYB_CALL inline void field(const HybK &P, const Sub &S, const double *y, double th2t, double qref, double vdcref, double *dy, Alg &A) {
    PVDER_HYB_CNT(1);
    const double Vdc = y[18], xDC = y[19], xQ = y[20], xPLL = y[21], delta = y[22];
    const double hV = 0.5 * Vdc;
    double mR[3], mI[3], Pt = 0.0, Qp = 0.0;
    for (int p = 0; p < 3; ++p) {
        const double iR = y[6 * p], iI = y[6 * p + 1];
        mR[p] = P.KpG * y[6 * p + 4] + y[6 * p + 2];
        mI[p] = P.KpG * y[6 * p + 5] + y[6 * p + 3];
        A.vR[p] = S.AR[p] + P.kviR * iR - P.kviI * iI;
        A.vI[p] = S.AI[p] + P.kviR * iI + P.kviI * iR;
        Pt += mR[p] * iR + mI[p] * iI;
        Qp += A.vI[p] * iR - A.vR[p] * iI;
    }
    const double ReS = 0.5 * hV * Pt, Q = 0.5 * Qp;
    const double iarR = xDC + P.KpDC * (vdcref - Vdc);
    const double qerr = qref - Q;
    const double iarI = xQ - P.KpQ * qerr;
    A.iarR = iarR; A.iarI = iarI;
    double vd;
    {
        const double ang = delta, s1 = sin(ang), c1 = cos(ang);
        const double sbR = y[6] + y[12], sbI = y[7] + y[13], dbR = y[6] - y[12], dbI = y[7] - y[13];
        const double bR = y[0] - 0.5 * sbR, bI = y[1] - 0.5 * sbI;
        const double i1R = bR - kH3 * dbI, i1I = bI + kH3 * dbR, i2R = bR + kH3 * dbI, i2I = bI - kH3 * dbR;
        const double k3R = P.kviR * (1.0 / 3.0), k3I = P.kviI * (1.0 / 3.0);
        const double v1R = S.AR[0] + k3R * i1R - k3I * i1I, v1I = S.AI[0] + k3R * i1I + k3I * i1R;
        const double v2R = k3R * i2R - k3I * i2I, v2I = k3R * i2I + k3I * i2R;
        const double a2 = kTwoPiD * th2t + ang, s2 = sin(a2), c2 = cos(a2);
        vd = v1R * c1 + v1I * s1 + v2R * c2 - v2I * s2;
    }
    const double dw = P.KpP * vd + xPLL, wr = 1.0 + dw * P.iwb;
    for (int p = 0; p < 3; ++p) {
        const double iR = y[6 * p], iI = y[6 * p + 1];
        double rR, rI;
        if (p == 0) { rR = iarR; rI = iarI; }
        else if (p == 1) { rR = -0.5 * iarR + kH3 * iarI; rI = -0.5 * iarI - kH3 * iarR; }
        else { rR = -0.5 * iarR - kH3 * iarI; rI = -0.5 * iarI + kH3 * iarR; }
        dy[6 * p] = P.iLf * (mR[p] * hV - A.vR[p] - P.Rf * iR) + wr * iI;
        dy[6 * p + 1] = P.iLf * (mI[p] * hV - A.vI[p] - P.Rf * iI) - wr * iR;
        dy[6 * p + 2] = P.KiG * y[6 * p + 4];
        dy[6 * p + 3] = P.KiG * y[6 * p + 5];
        dy[6 * p + 4] = P.wp * (rR - iR - y[6 * p + 4]);
        dy[6 * p + 5] = P.wp * (rI - iI - y[6 * p + 5]);
    }
    double ppv;
    {
        const double e = exp2(P.pvK * Vdc);
        ppv = Vdc * (S.pv1 - P.pvB * e);
        if (P.pvK == 0.0) ppv = -S.pv1;                 // model option U5 (see ModelK)
        ppv = ppv > 0.0 ? ppv : 0.0;
    }
    dy[18] = (ppv - ReS) / Vdc * P.iC;
    double dxDC = P.KiDC * (vdcref - Vdc), dxQ = -P.KiQ * qerr;
    if (iarR * iarR + iarI * iarI > P.ilim2) {
        dxDC = (dxDC * xDC > 0.0) ? 0.0 : dxDC;
        dxQ = (dxQ * xQ > 0.0) ? 0.0 : dxQ;
    }
    dy[19] = dxDC; dy[20] = dxQ; dy[21] = P.KiP * vd; dy[22] = dw;
}

// d/dt of e_p = iref_p - i_p with u_p held (2-vector), from the free field f at y: d iref_a/dt = (dxDC - KpDC dVdc) + j (dxQ + KpQ dQ),
// dQ = Im d(S_PCC)/dt, rotated to phase p, minus d i_p/dt.
PVDER_HD inline void demand_rate(const HybK &P, const Alg &A, const double *y, const double *f, int p, double *de) {
    double dQ = 0.0;
    for (int q = 0; q < 3; ++q) {
        const double diR = f[6 * q], diI = f[6 * q + 1], iR = y[6 * q], iI = y[6 * q + 1];
        const double dvR = P.kviR * diR - P.kviI * diI, dvI = P.kviR * diI + P.kviI * diR;
        // Im(dv conj(i)) + Im(v conj(di))
        dQ += (dvI * iR - dvR * iI) + (A.vI[q] * diR - A.vR[q] * diI);
    }
    dQ *= 0.5;
    const double daR = f[19] - P.KpDC * f[18], daI = f[20] + P.KpQ * dQ;
    double rR, rI;
    if (p == 0) { rR = daR; rI = daI; }
    else if (p == 1) { rR = -0.5 * daR + kH3 * daI; rI = -0.5 * daI - kH3 * daR; }
    else { rR = -0.5 * daR - kH3 * daI; rI = -0.5 * daI + kH3 * daR; }
    de[0] = rR - f[6 * p]; de[1] = rI - f[6 * p + 1];
}

// d(de_p/dt)/du_p (2 x 2, row-major Ge[2 r + c] = d de_r / d u_c): -g K on the diagonal, beside it what u does to the reference
// through the DC link (Re S) and through Q (the current's own derivative)
PVDER_HD inline void demand_gain(const HybK &P, const Alg &A, const double *y, int p, double *Ge) {
    const double K = P.KpG, gK = 0.5 * y[18] * P.iLf * K;
    const double iR = y[6 * p], iI = y[6 * p + 1];
    for (int c = 0; c < 2; ++c) {
        // d(di_p) = gK e_c  (e_0 = 1, e_1 = j)
        const double eR = c == 0 ? 1.0 : 0.0, eI = c == 0 ? 0.0 : 1.0;
        const double ddVdc = -(0.25 * K) * P.iC * (c == 0 ? iR : iI);
        // dQ = 1/2 gK Im[kvi e_c conj(i_p) + v_p conj(e_c)]
        const double kR = P.kviR * eR - P.kviI * eI, kI = P.kviR * eI + P.kviI * eR;
        const double ddQ = 0.5 * gK * ((kI * iR - kR * iI) + (A.vI[p] * eR - A.vR[p] * eI));
        const double daR = -P.KpDC * ddVdc, daI = P.KpQ * ddQ;
        double rR, rI;
        if (p == 0) { rR = daR; rI = daI; }
        else if (p == 1) { rR = -0.5 * daR + kH3 * daI; rI = -0.5 * daI - kH3 * daR; }
        else { rR = -0.5 * daR - kH3 * daI; rI = -0.5 * daI + kH3 * daR; }
        Ge[0 + c] = rR - gK * eR;
        Ge[2 + c] = rI - gK * eI;
    }
}

// ---- the closed loops' operator functions: phi_k(h L2) = c1[k] I + c2[k] L2, k = 0..3
struct Loop {
    double a, b, c, d;          // L2 = [[a, b], [c, d]]
    double al, be;              // eigenvalues al +- j be
};
struct Phi {
    double c1[4], c2[4];
};
PVDER_HD inline void make_loop(const HybK &P, Loop &L) {
    L.a = -(P.kviR + P.Rf) * P.iLf;
    L.b = 0.5 * P.vdc_ref0 * P.iLf * P.KpG;
    L.c = -P.wp;
    L.d = -P.wp;
    const double tr = L.a + L.d, det = L.a * L.d - L.b * L.c;
    L.al = 0.5 * tr;
    L.be = sqrt(det - 0.25 * tr * tr);
}
PVDER_HD inline void make_phi(const Loop &L, double h, Phi &F) {
    const double zr = h * L.al, zi = h * L.be;
    const double ex = exp(zr);
    double pr = ex * cos(zi), pi = ex * sin(zi);
    const double iz2 = 1.0 / (zr * zr + zi * zi);
    double fact = 1.0;
    for (int k = 0; k < 4; ++k) {
        const double c2 = pi / L.be;
        F.c2[k] = c2;
        F.c1[k] = pr - c2 * L.al;
        // phi_{k+1} = (phi_k - 1/k!) / z
        const double nr = pr - 1.0 / fact, ni = pi;
        pr = (nr * zr + ni * zi) * iz2;
        pi = (ni * zr - nr * zi) * iz2;
        fact *= (double)(k + 1);
    }
}

// state of the scheme across the steps of one env step
struct Run {
    int modes;                  // 2 bits per phase: bit 2p = u_R held, bit 2p+1 = u_I held
    int sat;                    // bit p: the phase is at the limit
    double t_settle[3];         // instant of the phase's last settle (no second release at the same instant)
    double t_rep[6]; int n_rep[6];
    Stats st;
};

PVDER_HD inline double s_of(const HybK &P, const double *y, int p) {
    const double mR = y[6 * p + 2] + P.KpG * y[6 * p + 4], mI = y[6 * p + 3] + P.KpG * y[6 * p + 5];
    return mR * mR + mI * mI - P.mlim2;
}
PVDER_HD inline double sgn(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// x_c of a saturated phase is held while it would grow (dx_c = Ki u_c has the sign of x_c)
PVDER_HD inline int xmask_of(const double *y, int sat) {
    int xm = 0;
    for (int p = 0; p < 3; ++p)
        for (int c = 0; c < 2; ++c)
            if (((sat >> p) & 1) && (y[6 * p + 4 + c] * y[6 * p + 2 + c] > 0.0 || y[6 * p + 2 + c] == 0.0)) xm |= 1 << (2 * p + c);
    return xm;
}

PVDER_HD inline void masked(const HybK &P, const Sub &S, const double *y, double th2t, double qref, double vdcref, int modes, int xmask,
                            double *f) {
    Alg A;
    field(P, S, y, th2t, qref, vdcref, f, A);
    for (int k = 0; k < 6; ++k) {
        const int p = k >> 1, c = k & 1;
        if ((modes >> k) & 1) f[6 * p + 4 + c] = 0.0;
        if ((xmask >> k) & 1) f[6 * p + 2 + c] = 0.0;
    }
}

// one Hochbruck-Ostermann ETDRK4 step of length h from (y, th0 = 2 theta in turns at its start) with the modes held.
// yn may be any slot but the eight this function works in (A_N1 .. A_TMP); y must not be one of them either.
PVDER_HD PVDER_HYB_CALL inline void etd_step(const HybK &P, const Sub &S, const Loop &L, const Arena &W, const double *y, double th0, double h, double qref,
                              double vdcref, int modes, int xmask, double *yn) {
    PVDER_HYB_CNT(0);
    Phi Fh, Fg;
    make_phi(L, h, Fh);
    make_phi(L, 0.5 * h, Fg);
    const double tq = 120.0 * h;
    double *N1 = W[A_N1], *N2 = W[A_N2], *N4 = W[A_N4], *N5 = W[A_N5], *yt = W[A_YT], *e2y = W[A_E2Y], *ey = W[A_EY], *tmp = W[A_TMP];
    // N(y) = F(y) - L y on the closed pairs
    auto Nof = [&](const double *yy, double th, double *N) {
        masked(P, S, yy, th, qref, vdcref, modes, xmask, N);
        for (int q = 0; q < 6; ++q) {
            if ((modes >> q) & 1) continue;
            const int p = q >> 1, c = q & 1, i0 = 6 * p + c, i1 = 6 * p + 4 + c;
            N[i0] -= L.a * yy[i0] + L.b * yy[i1];
            N[i1] -= L.c * yy[i0] + L.d * yy[i1];
        }
    };
    // A stage is  base + h sum_k w_k phi_k(.) N  over a few phi functions of (h L) and (h/2 L): on the components outside the closed
    // pairs every phi_k is a number (1, 1, 1/2, 1/6), on a closed pair c1_k I + c2_k L2 -- so the SUM is one number / one (C1, C2)
    // per source vector, and a stage is ONE pass per source vector (not one per phi function: the vectors live in LDS on the device).
    struct Comb { double s, C1, C2; };
    auto comb = [&](const double *wg, const double *wf) -> Comb {       // weights of g_k = phi_k(h/2 L) and f_k = phi_k(h L), k = 0..3
        const double sc[4] = {1.0, 1.0, 0.5, 1.0 / 6.0};
        Comb c = {0.0, 0.0, 0.0};
        for (int k = 0; k < 4; ++k) {
            c.s += (wg[k] + wf[k]) * sc[k];
            c.C1 += wg[k] * Fg.c1[k] + wf[k] * Fh.c1[k];
            c.C2 += wg[k] * Fg.c2[k] + wf[k] * Fh.c2[k];
        }
        return c;
    };
    // dst (=|+=) comb . v
    auto pass = [&](const Comb &c, const double *v, double *dst, bool add) {
        for (int j = 0; j < 23; ++j) {
            bool pair = false;
            if (j < 18) {
                const int p = j / 6, r = j % 6;
                if (r < 2) pair = !((modes >> (2 * p + r)) & 1);
                else if (r >= 4) pair = !((modes >> (2 * p + r - 4)) & 1);
            }
            if (pair) continue;
            const double t = c.s * v[j];
            dst[j] = add ? dst[j] + t : t;
        }
        for (int q = 0; q < 6; ++q) {
            if ((modes >> q) & 1) continue;
            const int p = q >> 1, cc = q & 1, i0 = 6 * p + cc, i1 = 6 * p + 4 + cc;
            const double v0 = v[i0], v1 = v[i1];
            const double t0 = c.C1 * v0 + c.C2 * (L.a * v0 + L.b * v1), t1 = c.C1 * v1 + c.C2 * (L.c * v0 + L.d * v1);
            dst[i0] = add ? dst[i0] + t0 : t0;
            dst[i1] = add ? dst[i1] + t1 : t1;
        }
    };
    const double Z[4] = {0, 0, 0, 0};
    Nof(y, th0, N1);
    {   // e2y = g0 y, ey = f0 y
        const double w0[4] = {1, 0, 0, 0};
        pass(comb(w0, Z), y, e2y, false);
        pass(comb(Z, w0), y, ey, false);
    }
    // U2 = e2y + h 0.5 g1 N1
    for (int j = 0; j < 23; ++j) yt[j] = e2y[j];
    { const double wg[4] = {0, 0.5 * h, 0, 0}; pass(comb(wg, Z), N1, yt, true); }
    Nof(yt, th0 + 0.5 * tq, N2);
    // U3 = e2y + h ((0.5 g1 - g2) N1 + g2 N2)
    for (int j = 0; j < 23; ++j) yt[j] = e2y[j];
    { const double wg[4] = {0, 0.5 * h, -h, 0}; pass(comb(wg, Z), N1, yt, true); }
    { const double wg[4] = {0, 0, h, 0}; pass(comb(wg, Z), N2, yt, true); }
    Nof(yt, th0 + 0.5 * tq, N5);                                     // N3 (in N5's slot)
    for (int j = 0; j < 23; ++j) N2[j] += N5[j];                     // N2 + N3
    // U4 = f0 y + h ((f1 - 2 f2) N1 + f2 (N2 + N3))
    for (int j = 0; j < 23; ++j) yt[j] = ey[j];
    { const double wf[4] = {0, h, -2.0 * h, 0}; pass(comb(Z, wf), N1, yt, true); }
    { const double wf[4] = {0, 0, h, 0}; pass(comb(Z, wf), N2, yt, true); }
    Nof(yt, th0 + tq, N4);
    // U5 = e2y + h (a51 N1 + a52 (N2 + N3) + a54 N4),  a52 = 0.5 g2 - 0.5 g3 + 0.25 f2 - f3,  a54 = 0.25 g2 - a52,
    // a51 = 0.5 g1 - 2 a52 - a54 = 0.5 g1 - 0.75 g2 + 0.5 g3 - 0.25 f2 + f3
    for (int j = 0; j < 23; ++j) yt[j] = e2y[j];
    { const double wg[4] = {0, 0.5 * h, -0.75 * h, 0.5 * h}, wf[4] = {0, 0, -0.25 * h, h}; pass(comb(wg, wf), N1, yt, true); }
    { const double wg[4] = {0, 0, 0.5 * h, -0.5 * h}, wf[4] = {0, 0, 0.25 * h, -h}; pass(comb(wg, wf), N2, yt, true); }
    { const double wg[4] = {0, 0, -0.25 * h, 0.5 * h}, wf[4] = {0, 0, -0.25 * h, h}; pass(comb(wg, wf), N4, yt, true); }
    Nof(yt, th0 + 0.5 * tq, N5);
    // y+ = f0 y + h ((f1 - 3 f2 + 4 f3) N1 + (-f2 + 4 f3) N4 + (4 f2 - 8 f3) N5)
    for (int j = 0; j < 23; ++j) yt[j] = ey[j];
    { const double wf[4] = {0, h, -3.0 * h, 4.0 * h}; pass(comb(Z, wf), N1, yt, true); }
    { const double wf[4] = {0, 0, -h, 4.0 * h}; pass(comb(Z, wf), N4, yt, true); }
    { const double wf[4] = {0, 0, 4.0 * h, -8.0 * h}; pass(comb(Z, wf), N5, yt, true); }
    for (int j = 0; j < 23; ++j) yn[j] = yt[j];
    for (int k = 0; k < 6; ++k) {          // held components are held exactly
        const int p = k >> 1, c = k & 1;
        if ((modes >> k) & 1) yn[6 * p + 4 + c] = y[6 * p + 4 + c];
        if ((xmask >> k) & 1) yn[6 * p + 2 + c] = y[6 * p + 2 + c];
    }
}

// ---- settle: the micro-integration of phase p's fast pair (see the header).  trust_u: the axes whose loop was closed over a large
// step carry their current to 1e-5 -- a demand of K 1e-5 = 0.06 in m: their current is put on the slow manifold (D_c = 0) first.
PVDER_HD PVDER_HYB_CALL inline void settle(const HybK &P, const Sub &S, const Arena &W, double *y, double th2t, double qref, double vdcref, int p,
                            bool trust_u, Run &R, double tnow) {
    const double K = P.KpG, wp = P.wp, lim2 = P.mlim2;
    double *f = W[A_F0];                 // (free here: a settle comes behind a committed state, never inside a step)
    Alg A;
    field(P, S, y, th2t, qref, vdcref, f, A);
    if (trust_u) {
        bool moved = false;
        for (int c = 0; c < 2; ++c)
            if (!((R.modes >> (2 * p + c)) & 1)) { y[6 * p + c] += f[6 * p + 4 + c] / wp; moved = true; }
        if (moved) field(P, S, y, th2t, qref, vdcref, f, A);
    }
    double de0[2], Ge[4];
    demand_rate(P, A, y, f, p, de0);
    demand_gain(P, A, y, p, Ge);
    const double x[2] = {y[6 * p + 2], y[6 * p + 3]};
    const double u0[2] = {y[6 * p + 4], y[6 * p + 5]};
    const double D0[2] = {f[6 * p + 4] / wp, f[6 * p + 5] / wp};
    double u[2] = {u0[0], u0[1]}, Dm[2] = {D0[0], D0[1]};
    const double hmic = 0.25e-6, tau_about = 20e-6;
    const int max_micro = 20000;

    // the region from the state alone (fil_decide) and the G masks
    auto decide = [&](const double *uu, const double *DD, bool *G) -> int {
        const double fu[2] = {wp * DD[0], wp * DD[1]};
        const double mm[2] = {x[0] + K * uu[0], x[1] + K * uu[1]};
        const double sv = mm[0] * mm[0] + mm[1] * mm[1] - lim2;
        for (int c = 0; c < 2; ++c) G[c] = (uu[c] == 0.0) || (fu[c] * uu[c] > 0.0);
        if (sv > 1e-11) return 1;
        if (sv < -1e-11) return 0;
        const double N = 2.0 * K * (mm[0] * fu[0] + mm[1] * fu[1]);
        double Dn = 0.0;
        for (int c = 0; c < 2; ++c) if (G[c]) Dn += 2.0 * K * mm[c] * fu[c];
        if (N <= 0.0) return 0;
        if (N - Dn >= 0.0) return 1;
        return 2;
    };
    // (u', D') with the region held over the step (as the oracle holds it: fil_rhs)
    auto fld = [&](const double *uu, const double *DD, int reg, double *up, double *Dp) {
        const double fu[2] = {wp * DD[0], wp * DD[1]};
        up[0] = fu[0]; up[1] = fu[1];
        if (reg != 0) {
            const double mm[2] = {x[0] + K * uu[0], x[1] + K * uu[1]};
            bool G[2];
            for (int c = 0; c < 2; ++c) G[c] = (uu[c] == 0.0) || (fu[c] * uu[c] > 0.0);
            if (reg == 1) {
                for (int c = 0; c < 2; ++c) if (G[c]) up[c] = 0.0;
            } else {
                const double N = 2.0 * K * (mm[0] * fu[0] + mm[1] * fu[1]);
                double Dn = 0.0;
                for (int c = 0; c < 2; ++c) if (G[c]) Dn += 2.0 * K * mm[c] * fu[c];
                double al = Dn > 0.0 ? N / Dn : 0.0;
                al = al < 0.0 ? 0.0 : (al > 1.0 ? 1.0 : al);
                for (int c = 0; c < 2; ++c) if (G[c]) up[c] *= (1.0 - al);
            }
        }
        const double du0 = uu[0] - u0[0], du1 = uu[1] - u0[1];
        Dp[0] = de0[0] + Ge[0] * du0 + Ge[1] * du1 - up[0];
        Dp[1] = de0[1] + Ge[2] * du0 + Ge[3] * du1 - up[1];
    };

    int n = 0, quiet = 0;
    double elapsed = 0.0;
    while (n < max_micro) {
        bool G[2];
        const int reg = decide(u, Dm, G);
        double up1[2], Dp1[2];
        fld(u, Dm, reg, up1, Dp1);
        // quiet: nothing moves fast, and no held component is about to be released (h_c = 0 and falling)
        const double upm = fmax(fabs(up1[0]), fabs(up1[1]));
        const bool fast = K * upm * 5e-6 >= 1e-7;
        bool about = false;
        for (int c = 0; c < 2; ++c)
            if (reg != 0 && G[c] && u[c] != 0.0) {
                const double sg = sgn(u[c]);
                if (Dp1[c] * sg < 0.0 && Dm[c] * sg < fmax(1e-10, -Dp1[c] * sg * tau_about)) about = true;
            }
        if (!fast && !about) {
            if (++quiet >= 2) break;
        } else {
            quiet = 0;
        }
        // a component that moves fast (a violent release: r ~ 50 /s) takes the step in pieces: K |du| <= 1e-3 per micro step
        const double h = hmic / fmax(1.0, ceil(K * upm * hmic / 1e-3));
        double ut[2], Dt[2], up2[2], Dp2[2], up3[2], Dp3[2], up4[2], Dp4[2];
        for (int c = 0; c < 2; ++c) { ut[c] = u[c] + 0.5 * h * up1[c]; Dt[c] = Dm[c] + 0.5 * h * Dp1[c]; }
        fld(ut, Dt, reg, up2, Dp2);
        for (int c = 0; c < 2; ++c) { ut[c] = u[c] + 0.5 * h * up2[c]; Dt[c] = Dm[c] + 0.5 * h * Dp2[c]; }
        fld(ut, Dt, reg, up3, Dp3);
        for (int c = 0; c < 2; ++c) { ut[c] = u[c] + h * up3[c]; Dt[c] = Dm[c] + h * Dp3[c]; }
        fld(ut, Dt, reg, up4, Dp4);
        double du[2], dD[2];
        for (int c = 0; c < 2; ++c) {
            du[c] = (h / 6.0) * (up1[c] + 2.0 * up2[c] + 2.0 * up3[c] + up4[c]);
            dD[c] = (h / 6.0) * (Dp1[c] + 2.0 * Dp2[c] + 2.0 * Dp3[c] + Dp4[c]);
        }
        double th = 1.0;
        const double mm0[2] = {x[0] + K * u[0], x[1] + K * u[1]};
        const double s0 = mm0[0] * mm0[0] + mm0[1] * mm0[1] - lim2;
        const double dmm[2] = {K * du[0], K * du[1]};
        const double qa = dmm[0] * dmm[0] + dmm[1] * dmm[1], qb = 2.0 * (mm0[0] * dmm[0] + mm0[1] * dmm[1]);
        const double s1 = s0 + qb + qa;
        if (qa > 0.0 && ((reg == 0 && s1 > 0.0) || (reg == 1 && s1 < 0.0))) {      // onto the circle (from inside / outside): stop ON it
            const double disc = qb * qb - 4.0 * qa * s0;
            if (disc >= 0.0) {
                const double sq = sqrt(disc), r1 = (-qb - sq) / (2.0 * qa), r2 = (-qb + sq) / (2.0 * qa);
                double best = 2.0;
                if (r1 > 1e-14 && r1 <= 1.0) best = r1;
                if (r2 > 1e-14 && r2 <= 1.0 && r2 < best) best = r2;
                if (best <= 1.0) th = best;
            }
        }
        int zc = -1;
        if (reg != 0)
            for (int c = 0; c < 2; ++c)                                         // a component through zero while not inside: onto zero
                if (u[c] != 0.0 && (u[c] + du[c]) * u[c] < 0.0) {
                    const double tz = u[c] / (-du[c]);
                    if (tz < th) { th = tz; zc = c; }
                }
        double un[2] = {u[0] + th * du[0], u[1] + th * du[1]};
        const double Dn2[2] = {Dm[0] + th * dD[0], Dm[1] + th * dD[1]};
        if (zc >= 0) un[zc] = 0.0;
        if (reg == 2) {                                                         // sliding: back onto the circle along what moved
            const double m1[2] = {x[0] + K * un[0], x[1] + K * un[1]};
            const double sv = m1[0] * m1[0] + m1[1] * m1[1] - lim2;
            const double g0 = du[0] != 0.0 ? 2.0 * K * m1[0] : 0.0, g1 = du[1] != 0.0 ? 2.0 * K * m1[1] : 0.0;
            const double gg = g0 * g0 + g1 * g1;
            if (gg > 0.0) { un[0] -= sv * g0 / gg; un[1] -= sv * g1 / gg; }
        }
        u[0] = un[0]; u[1] = un[1]; Dm[0] = Dn2[0]; Dm[1] = Dn2[1];
        ++n;
        elapsed += th * h;
    }
    // the current moved by what the demand did not account for: e = D + u, d i = d iref - d e
    for (int c = 0; c < 2; ++c) {
        const double de_total = (Dm[c] + u[c]) - (D0[c] + u0[c]);
        y[6 * p + c] += (de0[c] + f[6 * p + c]) * elapsed - de_total;
        y[6 * p + 4 + c] = u[c];
    }
    // classification for the macro step
    bool G[2];
    const int reg = decide(u, Dm, G);
    int md = 0;
    if (reg != 0) {
        double up1[2], Dp1[2];
        fld(u, Dm, reg, up1, Dp1);
        for (int c = 0; c < 2; ++c) {
            if (!G[c]) continue;
            // held -- unless it sits ON its release surface with the demand still falling: it regulates (the impacts' Zeno limit)
            const double sg = sgn(u[c]);
            if (u[c] != 0.0 && K * fabs(Dm[c]) < 1e-4 && Dp1[c] * sg < 0.0) continue;
            md |= 1 << c;
        }
    }
    R.modes = (R.modes & ~(3 << (2 * p))) | (md << (2 * p));
    R.sat = (R.sat & ~(1 << p)) | ((reg != 0 ? 1 : 0) << p);
    if (reg == 0) R.modes &= ~(3 << (2 * p));
    R.t_settle[p] = tnow;
    R.st.settles += 1;
    R.st.micro += n;
}

// modes and sat from the state alone (the oracle does the same at the start of every sim time step: fil_settle)
PVDER_HD PVDER_HYB_CALL inline void derive_modes(const HybK &P, const Sub &S, const Arena &W, const double *y, double th2t, double qref, double vdcref,
                                  Run &R) {
    double *f = W[A_F0];
    Alg A;
    field(P, S, y, th2t, qref, vdcref, f, A);
    R.modes = 0; R.sat = 0;
    for (int p = 0; p < 3; ++p) {
        if (s_of(P, y, p) > -1e-6) {
            int md = 0;
            for (int c = 0; c < 2; ++c) {
                const double u = y[6 * p + 4 + c];
                if (u == 0.0 || f[6 * p + 4 + c] * u > 0.0) md |= 1 << c;
            }
            R.modes |= md << (2 * p);
            if (md) R.sat |= 1 << p;
        }
    }
}

// at the start of a macro step: a held axis released by a jump of the inputs; a free phase whose closed loop is heading beyond the
// limit for good (an action near the limit, a grid-voltage event: the quasi-static duty cycle ends at or beyond it)
PVDER_HD PVDER_HYB_CALL inline void input_jump(const HybK &P, const Sub &S, const Arena &W, double *y, double th2t, double qref, double vdcref, Run &R,
                                double tnow) {
    const double K = P.KpG, jump_margin = 4e-3;
    for (int p = 0; p < 3; ++p) {
        double *f = W[A_F1];
        Alg A;
        field(P, S, y, th2t, qref, vdcref, f, A);
        const double D[2] = {f[6 * p + 4] / P.wp, f[6 * p + 5] / P.wp};
        if ((R.sat >> p) & 1) {
            bool rel = false;
            for (int c = 0; c < 2; ++c) {
                const double u = y[6 * p + 4 + c];
                if (((R.modes >> (2 * p + c)) & 1) && u != 0.0 && -D[c] * sgn(u) > 1e-9) rel = true;
            }
            if (rel) settle(P, S, W, y, th2t, qref, vdcref, p, false, R, tnow);
            continue;
        }
        double de0[2];
        demand_rate(P, A, y, f, p, de0);
        const bool jump = K * sqrt(D[0] * D[0] + D[1] * D[1]) > 0.5 || sqrt(de0[0] * de0[0] + de0[1] * de0[1]) > 0.5;
        if (!jump) continue;
        // where the phase ends once its current has caught up: the duty cycle that holds the current on its new reference
        const double g = 0.5 * y[18] * P.iLf;
        double *y2 = W[A_Y1], *f2 = W[A_YT2];
        for (int j = 0; j < 23; ++j) y2[j] = y[j];
        y2[6 * p] += D[0]; y2[6 * p + 1] += D[1];
        Alg A2;
        field(P, S, y2, th2t, qref, vdcref, f2, A2);
        const double mqR = y[6 * p + 2] + K * y[6 * p + 4] - f2[6 * p] / g, mqI = y[6 * p + 3] + K * y[6 * p + 5] - f2[6 * p + 1] / g;
        if (mqR * mqR + mqI * mqI <= P.mlim2 * (1.0 - jump_margin)) continue;
        settle(P, S, W, y, th2t, qref, vdcref, p, false, R, tnow);
    }
}

// first root in (0, 1) of the Hermite cubic with values a < 0 < b and slopes sa, sb (per unit of theta)
PVDER_HD inline double cubic_root(double a, double b, double sa, double sb, double th0) {
    const double c0 = a, c1 = sa, c2 = 3.0 * (b - a) - 2.0 * sa - sb, c3 = 2.0 * (a - b) + sa + sb;
    double lo = 0.0, hi = 1.0, th = th0;
    for (int it = 0; it < 20; ++it) {
        const double v = ((c3 * th + c2) * th + c1) * th + c0;
        if (v > 0.0) hi = th; else lo = th;
        const double d = (3.0 * c3 * th + 2.0 * c2) * th + c1;
        double tn = d > 0.0 ? th - v / d : 0.5 * (lo + hi);
        if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
        if (fabs(tn - th) < 1e-12) { th = tn; break; }
        th = tn;
    }
    return th;
}

enum { EV_NONE = 0, EV_ENTRY, EV_IMPACT, EV_CLAMP, EV_FREEZE };

// value / slope of an event function at a state (positive side = the event has happened)
PVDER_HD PVDER_HYB_CALL inline double event_value(const HybK &P, const Sub &S, const Arena &W, int kind, int p, int c, double sg0, const double *y, double th2t,
                                   double qref, double vdcref, double *slope) {
    double *f = W[A_F1];                 // (the field at the end of the full step is not needed any more while an event is located)
    Alg A;
    field(P, S, y, th2t, qref, vdcref, f, A);
    if (kind == EV_ENTRY) {
        if (slope) *slope = 0.0;
        return s_of(P, y, p);
    }
    if (kind == EV_IMPACT) {
        if (slope) {
            double de[2];
            demand_rate(P, A, y, f, p, de);
            *slope = -de[c] * sg0;
        }
        return -(f[6 * p + 4 + c] / P.wp) * sg0;
    }
    if (slope) *slope = -f[6 * p + 4 + c] * sg0;
    return -y[6 * p + 4 + c] * sg0;       // EV_CLAMP
}

// the circle constraint of saturated phases behind the slow motion of a step
PVDER_HD PVDER_HYB_CALL inline void after_step(const HybK &P, const Sub &S, const Arena &W, double *y, double th2t, double qref, double vdcref, Run &R,
                                double tnow) {
    for (int p = 0; p < 3; ++p)
        if (((R.sat >> p) & 1) && s_of(P, y, p) < -1e-12) settle(P, S, W, y, th2t, qref, vdcref, p, true, R, tnow);
}

// from (tnow, y, th0) over h with the events inside located and resolved; tnow is only a label (seconds within the env step)
PVDER_HD PVDER_HYB_CALL inline void advance(const HybK &P, const Sub &S, const Loop &L, const Arena &W, double *y, double th0, double h, double qref,
                             double vdcref, Run &R, double tnow) {
    input_jump(P, S, W, y, th0, qref, vdcref, R, tnow);
    double *f0 = W[A_F0], *f1 = W[A_F1], *y1 = W[A_Y1], *yt = W[A_YT2];
    double t = 0.0;                       // time into this macro step
    int guard = 0;
    while (t < h * (1.0 - 1e-13)) {
        const double hh = h - t, th_s = th0 + 120.0 * t;
        if (++guard > 48) {               // (never in the prototype's runs: the step is completed with what holds now)
            etd_step(P, S, L, W, y, th_s, hh, qref, vdcref, R.modes, xmask_of(y, R.sat), y1);
            for (int j = 0; j < 23; ++j) y[j] = y1[j];
            R.st.capped += 1;
            return;
        }
        const int modes0 = R.modes, xm = xmask_of(y, R.sat);
        Alg A0, A1;
        field(P, S, y, th_s, qref, vdcref, f0, A0);
        etd_step(P, S, L, W, y, th_s, hh, qref, vdcref, modes0, xm, y1);
        field(P, S, y1, th_s + 120.0 * hh, qref, vdcref, f1, A1);
        // the earliest event: (kind, p, c, value at the start a, at the end b); an event = a <= 0 < b
        int ek = EV_NONE, ep = 0, ec = 0;
        double ea = 0.0, eb = 0.0, eth = 2.0;
        for (int p = 0; p < 3; ++p) {
            if (!((R.sat >> p) & 1)) {
                const double s0 = s_of(P, y, p), s1 = s_of(P, y1, p);
                if (s1 > 0.0) {
                    const double th = s0 >= 0.0 ? 0.0 : s0 / (s0 - s1);
                    if (th < eth) { eth = th; ek = EV_ENTRY; ep = p; ec = -1; ea = s0; eb = s1; }
                }
                continue;
            }
            for (int c = 0; c < 2; ++c) {
                const double u0 = y[6 * p + 4 + c], u1 = y1[6 * p + 4 + c];
                if ((modes0 >> (2 * p + c)) & 1) {
                    if (u0 == 0.0) continue;                                    // clamped at zero: stays while saturated
                    const double sg = sgn(u0);
                    double a = -(f0[6 * p + 4 + c] / P.wp) * sg;
                    const double b = -(f1[6 * p + 4 + c] / P.wp) * sg;
                    const int q = 2 * p + c;
                    if (a >= 0.0 && (R.t_settle[p] == tnow + t)) {
                        if (b <= 0.0) continue;                                 // just settled: no second release at this instant
                        a = -1e-30;
                    }
                    (void)q;
                    if (b > 0.0 || a > 1e-9) {
                        if (a >= 0.0 && a <= 1e-9) {                            // ON the surface at the start: an impact only if falling
                            double de[2];
                            demand_rate(P, A0, y, f0, p, de);
                            if (-de[c] * sg <= 0.0) continue;
                        }
                        const double th = a >= 0.0 ? 0.0 : a / (a - b);
                        if (th < eth) { eth = th; ek = EV_IMPACT; ep = p; ec = c; ea = a; eb = b; }
                    }
                } else {
                    if (u1 * u0 < 0.0) {
                        const double a = -fabs(u0), b = fabs(u1), th = a / (a - b);
                        if (th < eth) { eth = th; ek = EV_CLAMP; ep = p; ec = c; ea = a; eb = b; }
                    } else if (fabs(u1) > fabs(u0) * (1.0 + 1e-9) + 1e-15) {
                        if (0.0 < eth) { eth = 0.0; ek = EV_FREEZE; ep = p; ec = c; ea = -1.0; eb = 1.0; }
                    }
                }
            }
        }
        if (ek == EV_NONE) {
            for (int j = 0; j < 23; ++j) y[j] = y1[j];
            t = h;
            after_step(P, S, W, y, th0 + 120.0 * h, qref, vdcref, R, tnow + h);
            continue;
        }
        R.st.events += 1;
        PVDER_HYB_CNT(3 + ek);            // (4: entry, 5: impact, 6: clamp, 7: freeze)
        double th = eth;
        const double sg0 = ec >= 0 ? sgn(y[6 * ep + 4 + ec]) : 0.0;
        if (ek == EV_IMPACT && th > 0.0) {
            double de0[2], de1[2];
            demand_rate(P, A0, y, f0, ep, de0);
            demand_rate(P, A1, y1, f1, ep, de1);
            th = cubic_root(ea, eb, -de0[ec] * sg0 * hh, -de1[ec] * sg0 * hh, th);
            for (int it = 0; it < 2; ++it) {
                etd_step(P, S, L, W, y, th_s, hh * th, qref, vdcref, modes0, xm, yt);
                double sl;
                const double val = event_value(P, S, W, ek, ep, ec, sg0, yt, th_s + 120.0 * hh * th, qref, vdcref, &sl);
                sl *= hh;
                if (sl <= 0.0) break;
                const double dth = -val / sl;
                if (fabs(dth) * hh < 5e-8) break;
                th = fmin(fmax(th + dth, 1e-9), 1.0);
            }
            th = fmin(th + 5e-8 / hh, 1.0);          // the event must have HAPPENED where it is handled
            etd_step(P, S, L, W, y, th_s, hh * th, qref, vdcref, modes0, xm, yt);
        } else if (ek != EV_FREEZE && th > 0.0) {
            double lo = 0.0, flo = ea, hi = 1.0, fhi = eb;
            for (int it = 0; it < 8; ++it) {                                    // secant and bisection in turns, values only
                etd_step(P, S, L, W, y, th_s, hh * th, qref, vdcref, modes0, xm, yt);
                const double val = event_value(P, S, W, ek, ep, ec, sg0, yt, th_s + 120.0 * hh * th, qref, vdcref, nullptr);
                if (val > 0.0) { hi = th; fhi = val; } else { lo = th; flo = val; }
                if ((hi - lo) * hh < 1e-7) break;
                th = (it & 1) ? lo + (hi - lo) * (-flo) / (fhi - flo) : 0.5 * (lo + hi);
                th = fmin(fmax(th, lo + 0.02 * (hi - lo)), hi - 0.02 * (hi - lo));
            }
            th = hi;
            etd_step(P, S, L, W, y, th_s, hh * th, qref, vdcref, modes0, xm, yt);
        } else {
            for (int j = 0; j < 23; ++j) yt[j] = y[j];
            th = 0.0;
        }
        for (int j = 0; j < 23; ++j) y[j] = yt[j];
        t += hh * th;
        const double th_e = th0 + 120.0 * t, te = tnow + t;
        if (ek == EV_FREEZE) {
            R.modes |= 1 << (2 * ep + ec);
        } else if (ek == EV_CLAMP) {
            y[6 * ep + 4 + ec] = 0.0;
            R.modes |= 1 << (2 * ep + ec);
        } else {
            // a release that repeats within 2 us of the axis' last one, again and again: the two time scales have met -- the axis
            // regulates for the rest of the step
            bool contact = false;
            if (ek == EV_IMPACT) {
                const int q = 2 * ep + ec;
                if (R.n_rep[q] > 0 && te - R.t_rep[q] < 2e-6) R.n_rep[q] += 1; else R.n_rep[q] = 1;
                R.t_rep[q] = te;
                if (R.n_rep[q] > 4) { R.modes &= ~(1 << q); contact = true; }
            }
            if (!contact) settle(P, S, W, y, th_e, qref, vdcref, ep, true, R, te);
        }
        after_step(P, S, W, y, th_e, qref, vdcref, R, te);
    }
}

// ONE env step of an env at / near the duty-cycle limit: ys[23] in pvder order with the PLL state in its relative form
// (delta = wte - wbase t), n_sub sim time steps of 1/60 s, the first in two macro steps (1/first_div, the rest).
// vg[s], si[s]: the inputs of sim time step s.
// arena: kArenaDoubles doubles of working memory (see Arena).
template <typename TV>
PVDER_HD inline void env_step(const HybK &P, double *y, double qref, double vdcref, const TV *vg, const TV *si, int n_sub, int first_div,
                              Stats *st_out, double *arena) {
    Arena W;
    W.base = arena;
    Loop L;
    make_loop(P, L);
    Run R;
    R.modes = 0; R.sat = 0;
    for (int p = 0; p < 3; ++p) R.t_settle[p] = -1.0;
    for (int q = 0; q < 6; ++q) { R.t_rep[q] = -1.0; R.n_rep[q] = 0; }
    R.st.settles = R.st.micro = R.st.events = R.st.capped = 0;
    Sub S;
    const double dt = 1.0 / 60.0;
    for (int s = 0; s < n_sub; ++s) {
        make_sub(P, (double)vg[s], (double)si[s], S);
        if (s == 0) derive_modes(P, S, W, y, 0.0, qref, vdcref, R);
        const double t0 = s * dt;
        const bool jump = s == 0 || vg[s] != vg[s - 1] || si[s] != si[s - 1];      // a new action / a new event row: the graded pair
        if (jump && first_div > 1) {
            const double h1 = dt / first_div;
            advance(P, S, L, W, y, 0.0, h1, qref, vdcref, R, t0);
            advance(P, S, L, W, y, 120.0 * h1, dt - h1, qref, vdcref, R, t0 + h1);
        } else {
            advance(P, S, L, W, y, 0.0, dt, qref, vdcref, R, t0);
        }
    }
    if (st_out) *st_out = R.st;
}

}  // namespace hyb
}  // namespace pvder
