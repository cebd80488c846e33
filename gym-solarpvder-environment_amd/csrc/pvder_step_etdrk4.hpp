// pvder_step_etdrk4.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// ETDRK4 (Hochbruck-Ostermann): the 23-state tail in its two forms, the lane kernels, the duty-cycle-limit follow-up kernel, the balanced fast path.
#pragma once

namespace {

// =============================================================================================
// ETDRK4: explicit exponential Runge-Kutta integration with the stiff current loop in the exponential.
//
// Per phase the pair z_p = (i_p, u_p) obeys  dz_p/dt = Lz z_p + N_p(y)  with the CONSTANT complex
// 2x2 matrix  Lz = [[-Z0, c0 Kp_GCC], [-wp, -wp]]  (Z0 = (kvi+Rf)/Lf + j, c0 = Vdc_ref0/(2 Lf)):
// its eigenvalues -1e5 +- j5.8e5 1/s are what forces classic RK4 down to h ~ 4 us.  Everything
// else (x_p, Vdc, xDC, xQ, xPLL, delta: poles <= 90 1/s) has Lz = 0.  With
// N_i = cm x + D + (cm-c0) Kp_GCC u - j (winv/wbase - 1) i,  N_u = wp iref_p  the scheme of
// Hochbruck & Ostermann (SIAM J. Numer. Anal. 43, 2005, eq. 5.19; 5 stages, c = 0, 1/2, 1/2, 1, 1/2;
// stiff order 4 -- Krogstad's 4-stage ETDRK4, used first, lost order on this slaved fast block)
//   U2 = E2 z + A21 N1;   U3 = E2 z + A31 N1 + A32 N2;   U4 = E z + A41 N1 + A42 (N2+N3)
//   U5 = E2 z + A51 N1 + A52 (N2+N3) + A54 N4;           z+ = E z + B1 N1 + B4 N4 + B5 N5
// propagates the fast loop exactly (E = exp(h Lz), phi-function matrices folded on the host in
// fp64); for the Lz = 0 states the same tableau is a 5-stage 4th-order Runge-Kutta method,
// y+ = y + h (N1 + N4)/6 + 2h N5/3.  h is set by the slow dynamics: 16 steps per env step (n = 15)
// instead of 61 440 (DESIGN.md "Step schedule").
// While a duty cycle is near its limit the conditional-integration anti-windup changes the fast
// dynamics; such (rare) steps are redone by classic RK4 micro-steps from the saved state.
// =============================================================================================
// The coefficient matrices themselves (3 step sizes x 13 matrices x 8 floats: set 0 for
// h = 1/(60*micro), the first step behind an input jump, set 1 for h = 1/(60*micro_coarse), set 2 for
// the remainder (1 - 1/micro)/60 of the graded schedule;
// order E2,E,P1,A1,A2,B1,B3,C1,C2,C4; per complex entry (re, im) for 11,12,21,22) live in a
// small device buffer filled once by the caller from pvder_etdrk4_coefficients(): as a by-value
// kernel argument they made the kernarg segment 1.9 KB, which every workgroup re-reads at launch.
constexpr int kMatFloats = 8;                      // 2x2 complex: (re, im) of entries 11, 12, 21, 22
constexpr int kMatCount = 13;
constexpr int kCoefSets = 3;                       // step sizes: first behind a jump / coarse / remainder
constexpr int kCoefFloats = kCoefSets * kMatCount * kMatFloats;
struct EtdK {
    float c0kp;             // c0 * Kp_GCC  (for N_i)
    float sat2_event;       // guard on the step that follows an input jump:  (0.97 *m_limit)^2
    float sat2_smooth;      // guard on all other steps:                       (0.998*m_limit)^2
    float sat2_flag;        // (0.9875 m_limit)^2: an env whose duty cycle STARTS a step beyond it leaves the lean pass (same arithmetic in the
                            // second pass, which raises the plan's hybrid flag: the hybrid kernel is in place 14 env steps of the
                            // reference's Vdc_ref ramp before the duty cycle saturates).  Ordinary PVDER-v0 rollouts -- 65 536 envs,
                            // random actions, both event kinds, three episodes -- reach 0.9817 at worst (tools/m_distribution.py): a
                            // lower threshold has some env leave the lean pass in one step of three, 2.2 us on EVERY launch
    int rk4_sub[3];         // classic RK4 micro-steps replacing one ETD step in the fallback (per set)
    float hset[3];          // step sizes [s]: 1/(60 micro), 1/(60 micro_coarse), (1 - 1/micro)/60
    float i2_to_m2;         // balanced kernel: |iref|^2 * i2_to_m2 > guard  <=>  |iref| within 10 % of its limit
};
// matrices of one step size, in the order they are used: those applied to (z, N1) first
enum { EM_E2 = 0, EM_E, EM_A21, EM_A31, EM_A41, EM_A51, EM_B1, EM_A32, EM_A42, EM_A52, EM_A54, EM_B4, EM_B5, EM_COUNT };

// ---- the switched block: the same 13 matrices for the (i_p, u_p) pair of ONE phase as REAL 4x4 matrices on
// (iR, iI, uR, uI), once per anti-windup MODE of the pair.  While |m_p| > m_limit the reference model freezes the
// components of du_p/dt (and dx_p/dt) that point outward (pvder_model.hpp::rhs, the `mlim2` branch): component-wise,
// so the pair no longer obeys a complex 2x2 system but one of four real LINEAR ones -- mode bit 0: uR frozen (its row
// of Lz zeroed), bit 1: uI frozen; mode 0 is the complex system itself.  Every ETDRK4 step of the 23-state path is taken with,
// per pair, the operator set of the mode found at its start (etd_switched_step) -- where a step that met the limit used
// to be redone by 61 440 classic-RK4 steps per env step (round 2), then by a second, switched step (round 3).
// Layout: [step size][mode][matrix][16 floats]; a matrix as column pairs (M0k, M1k) k = 0..3, then (M2k, M3k).
constexpr int kSwModes = 4;
constexpr int kSwMatFloats = 16;
constexpr int kSwFloats = kCoefSets * kSwModes * kMatCount * kSwMatFloats;       // 2496 floats = 9984 B
constexpr int kCoefTotal = kCoefFloats + kSwFloats;

// The ONE by-value argument of the ETDRK4 kernel (kernarg offset 0).  A single struct so that the
// 23-state tail (etdrk4_full_tail) can re-read every argument through a laundered copy of the kernarg
// pointer instead of keeping ~130 scalars of the fast path alive across the hot loop.
struct EtdArgs {
    ModelK<float> P;
    StepK<float> C;
    ResetK R;
    EtdK Ec;
    const float *coef;
    float *fstate;
    int32_t *istate;
    float *ev_vg, *ev_si;
    const uint8_t *action;
    float *out;
    float *stats;
    int stats_fresh;         // 1: this launch starts the step's statistics (overwrites its slots), 0: adds
    int n_env;
    int force_full;          // A/B and tests: every env takes the 23-state path
    int rk4_fallback;        // A/B and tests: steps at the duty-cycle limit by classic RK4 micro-steps (round 2's way)
    int sat_hybrid;          // 1 (set per launch): step_sat_hybrid_kernel follows this launch -- the 23-state tail hands it the envs at /
                             // near the duty-cycle limit (status kDeferred) instead of stepping them on the held operator sets
    int *hyb_flag;           // device-addressable HOST words: [0] written 1 by a workgroup that meets an env beyond kHybFlag2 (auto mode);
                             // [1] the sequence number of the last step launch that STARTED (hyb_progress: the host keeps <= 16 in flight)
    int hyb_seq;             // this launch's sequence number (0: do not report -- a captured launch)
    int ph1_micro;           // single phase: classic RK4 micro-steps per sim time step of the envs the fast path hands over
                             // (8192: poles at 1.0e6 rad/s) -- the tail of the step kernel (rk4_tail_1ph)
};

typedef __attribute__((address_space(3))) v4 lds_v4;     // LDS-resident coefficient entries

// one complex matrix entry c = (re, im):  acc += c * z  -- two v_pk_fma_f32: the broadcast of re / im,
// the swap of z and the sign of the low product are operand modifiers (op_sel, op_sel_hi, neg_lo), so
// an entry costs 2 floats of LDS and no moves
// (hipcc folds the broadcast itself; the second one it would build with v_xor + v_mov, hence the asm:
//  lo += -c.hi * z.hi,  hi += c.hi * z.lo)
__device__ __forceinline__ v2 cmac(v2 acc, v2 c, v2 z) {
    acc = pvder::fma2(pvder::bc2(c.x), z, acc);
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(c), "v"(z));
    return acc;
}
struct Mat4 { v4 lo, hi; };    // lo = entries 11, 12;  hi = entries 21, 22
__device__ __forceinline__ Mat4 load_mat(const lds_v4 *M, int k) {
    Mat4 m;
    m.lo = M[2 * k + 0]; m.hi = M[2 * k + 1];        // ds_read_b128 x2
    return m;
}
// (oi, ou) += M * (ni, nu)
__device__ __forceinline__ void matvec(const Mat4 &M, v2 ni, v2 nu, v2 &oi, v2 &ou) {
    oi = cmac(oi, pvder::mk2(M.lo.x, M.lo.y), ni);
    oi = cmac(oi, pvder::mk2(M.lo.z, M.lo.w), nu);
    ou = cmac(ou, pvder::mk2(M.hi.x, M.hi.y), ni);
    ou = cmac(ou, pvder::mk2(M.hi.z, M.hi.w), nu);
}

__device__ __forceinline__ float dot2(v2 a, v2 b) { return __builtin_fmaf(a.y, b.y, a.x * b.x); }

// N(y): fast part nf[2p] = N_i of phase p, nf[2p+1] = N_u; slow part ns[0..2] = dx_p,
// ns[3] = d(Vdc,xDC), ns[4] = d(xQ,xPLL), ns[5] = d(delta,0).  Returns max_p |m_p|^2.
__device__ __forceinline__ float etd_eval(const ModelK<float> &P, const pvder::PkConst &K, const EtdK &Ec,
                                          const pvder::SubPk &S, const v2 *__restrict__ y, float th2t,
                                          float qref, float vdcref, v2 *__restrict__ nf, v2 *__restrict__ ns) {
    // (contraction is spelled out below: this function and etd_eval3 must round alike -- an env's result does not depend on which
    //  of the two forms of the 23-state step advances it, bit for bit; left to hipcc the two contexts contract differently)
#pragma clang fp contract(off)
    using namespace pvder;
    const float Vdc = y[9].x, xDC = y[9].y, xQ = y[10].x, xPLL = y[10].y, delta = y[11].x;
    const v2 ia = y[0], xa = y[1], ua = y[2], ib = y[3], xb = y[4], ub = y[5], ic = y[6], xc = y[7], uc = y[8];
    const v2 kpg = bc2(P.KpG);
    const v2 ma = fma2(kpg, ua, xa), mb = fma2(kpg, ub, xb), mc = fma2(kpg, uc, xc);
    // the three power sums, phase by phase and then (a + b) + c: the order in which the phase-per-lane form (etd_eval3) adds
    // what it collects across a quad -- an env's result does not depend on which of the two forms advances it, bit for bit
    const float Pt = (dot2(ma, ia) + dot2(mb, ib)) + dot2(mc, ic);
    const float Qs = (dot2(S.Asw[0], ia) + dot2(S.Asw[1], ib)) + dot2(S.Asw[2], ic);
    const float Is = (dot2(ia, ia) + dot2(ib, ib)) + dot2(ic, ic);
    const float Q = __builtin_fmaf(K.qk, Is, 0.5f * Qs);
    const float qerr = qref - Q;
    const float iarR = __builtin_fmaf(P.KpDC, vdcref - Vdc, xDC);
    const float iarI = __builtin_fmaf(-P.KpQ, qerr, xQ);
    float s1, c1, s2, c2;
    const float dturn = delta * (float)(1.0 / kTwoPi);
    Math<float>::sincos_turns(dturn, s1, c1);
    Math<float>::sincos_turns(th2t + dturn, s2, c2);
    const v2 sb = ib + ic, db = ib - ic;
    const v2 base = fma2(bc2(-0.5f), sb, ia);
    const v2 jh3 = mk2(-kHalfSqrt3, kHalfSqrt3);
    const v2 rdb = swp(db) * jh3;
    const v2 i1 = base + rdb, i2 = base - rdb;
    const v2 jk3 = mk2(-P.kvi3I, P.kvi3I);
    v2 V1 = fma2(bc2(P.kvi3R), i1, S.A0); V1 = fma2(jk3, swp(i1), V1);
    v2 V2c = mk2(P.kvi3R, -P.kvi3R) * i2; V2c = fma2(bc2(-P.kvi3I), swp(i2), V2c);
    const v2 t1 = V1 * mk2(c1, s1);
    const v2 t2 = fma2(V2c, mk2(c2, s2), t1);
    const float vd = t2.x + t2.y;
    const float dw = __builtin_fmaf(P.KpP, vd, xPLL);
    const float dwr = dw * P.iwb;                               // winv/wbase - 1
    // N_i = cm x + D + (cm - c0) Kp u - j dwr i
    const float cmf = K.hiLf * Vdc;
    const v2 cm = bc2(cmf);
    const v2 dck = bc2(__builtin_fmaf(cmf, P.KpG, -Ec.c0kp));
    const v2 jd = mk2(dwr, -dwr);                               // -j*dwr*(iR + j iI) = (dwr iI, -dwr iR)
    v2 n;
    n = fma2(cm, xa, S.D[0]); n = fma2(dck, ua, n); nf[0] = fma2(jd, swp(ia), n);
    n = fma2(cm, xb, S.D[1]); n = fma2(dck, ub, n); nf[2] = fma2(jd, swp(ib), n);
    n = fma2(cm, xc, S.D[2]); n = fma2(dck, uc, n); nf[4] = fma2(jd, swp(ic), n);
    // N_u = wp * iref_p
    const v2 wra = mk2(P.wp * iarR, P.wp * iarI);
    const v2 hw = bc2(-0.5f) * wra;
    const v2 jw = swp(wra) * jh3;
    nf[1] = wra; nf[3] = hw - jw; nf[5] = hw + jw;
    const v2 kig = bc2(P.KiG);
    ns[0] = kig * ua; ns[1] = kig * ub; ns[2] = kig * uc;
    const float ppv = ppv_of(P, S.pv1, Vdc);
    const float dVdc = __builtin_fmaf(-0.25f * Vdc, Pt, ppv) * Math<float>::rcp(Vdc) * P.iC;        // (Ppv - Re S)/(Vdc C)
    float dxDC = P.KiDC * (vdcref - Vdc);
    float dxQ = -P.KiQ * qerr;
    if (__builtin_fmaf(iarI, iarI, iarR * iarR) > P.ilim2) {
        dxDC = (dxDC * xDC > 0.f) ? 0.f : dxDC;
        dxQ = (dxQ * xQ > 0.f) ? 0.f : dxQ;
    }
    ns[3] = mk2(dVdc, dxDC);
    ns[4] = mk2(dxQ, P.KiP * vd);
    ns[5] = mk2(dw, 0.f);
    return fmaxf(fmaxf(dot2(ma, ma), dot2(mb, mb)), dot2(mc, mc));
}

__device__ __forceinline__ float max_m2(const ModelK<float> &P, const v2 *y) {
    using namespace pvder;
    const v2 kpg = bc2(P.KpG);
    const v2 ma = fma2(kpg, y[2], y[1]), mb = fma2(kpg, y[5], y[4]), mc = fma2(kpg, y[8], y[7]);
    const v2 a = ma * ma, b = mb * mb, c = mc * mc;
    return fmaxf(fmaxf(a.x + a.y, b.x + b.y), c.x + c.y);
}

// ---- one ETDRK4 step of the 23-state model with the (i_p, u_p) pairs on the operator sets of their anti-windup
// modes (see kSwModes): THE step of etdrk4_full_tail.  Hochbruck-Ostermann's tableau, Kahan-compensated slow states;
// returns the largest |m_p|^2 at the start and at the end of the step.  On top of the plain scheme:
//   * per phase, the mode found at the START of the step (|m_p| > m_limit: the components of du_p/dt that point
//     outward are frozen) selects the real 4x4 operator set of the pair and blanks the frozen rows of N_u;
//   * dx_p/dt carries the model's sign tests at every stage.
// The mode is held over the step: a pair that leaves the limit inside the step is released at the next one
// (steps of 1/1920 .. 1/60 s against the 5 us the controller needs to pull |m| back), and in the regime this serves
// the duty cycle sits AT its limit for seconds -- stuck, not chattering.  With a duty cycle at its limit the
// trajectory is not reproducible beyond 1e-2 .. 1e-1 by ANY integrator (fp64 RK45 at 1e-8 vs 1e-10: 3e-2 .. 9e-2,
// profiles/r03_saturation_floor.txt), so this path is checked for what is robust: the duty cycle pinned at the limit,
// the DC link and the powers at the converged solution's level, finite states (tests/test_gpu_parity.py).
struct Mat16 { v4 a, b, c, d; };     // (col 0, col 1 of rows i), (col 2, col 3 of rows i), the same for rows u
// LDS layout of the switched block (etdrk4_full_tail stages it that way): [step size][matrix][16-byte word j = 0..3][mode] --
// the four MODES of a word side by side.  A lane addresses its pair's operator set by its own mode: lanes of one mode read
// one address (a broadcast), lanes of different modes read ADJACENT 16-byte words, i.e. different banks -- conflict-free
// whatever the mix.  (Round 3 laid the block out [mode][matrix][word]: modes 832 B apart, and a wave whose envs sat at
// different modes -- every wave while a batch APPROACHES the duty-cycle limit -- serialised each of its 2500 operand reads
// per env step: the 23-state path took 140 us there against 63 us once the whole batch had saturated,
// profiles/r04_ramp4_timeline.txt.)
__device__ __forceinline__ Mat16 load_mat16(const lds_v4 *M /* + mode */, int k) {
    Mat16 m;
    m.a = M[(4 * k + 0) * kSwModes]; m.b = M[(4 * k + 1) * kSwModes]; m.c = M[(4 * k + 2) * kSwModes]; m.d = M[(4 * k + 3) * kSwModes];
    return m;
}
__device__ __forceinline__ void matvec16(const Mat16 &M, v2 ni, v2 nu, v2 &oi, v2 &ou) {
    using namespace pvder;
    oi = fma2(mk2(M.a.x, M.a.y), bc2(ni.x), oi); oi = fma2(mk2(M.a.z, M.a.w), bc2(ni.y), oi);
    oi = fma2(mk2(M.b.x, M.b.y), bc2(nu.x), oi); oi = fma2(mk2(M.b.z, M.b.w), bc2(nu.y), oi);
    ou = fma2(mk2(M.c.x, M.c.y), bc2(ni.x), ou); ou = fma2(mk2(M.c.z, M.c.w), bc2(ni.y), ou);
    ou = fma2(mk2(M.d.x, M.d.y), bc2(nu.x), ou); ou = fma2(mk2(M.d.z, M.d.w), bc2(nu.y), ou);
}
// the model's sign tests on dx_p/dt at a stage state yt (ns[0..2] = dx_p), and the frozen rows of N_u
__device__ __forceinline__ void sw_mask(const ModelK<float> &P, const v2 *__restrict__ yt, int modes /* 2 bits per phase */,
                                        v2 *__restrict__ nf, v2 *__restrict__ ns) {
    using namespace pvder;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const v2 x = yt[3 * p + 1], m = fma2(bc2(P.KpG), yt[3 * p + 2], x);
        if (__builtin_fmaf(m.y, m.y, m.x * m.x) > P.mlim2) {
            ns[p].x = ns[p].x * x.x > 0.f ? 0.f : ns[p].x;
            ns[p].y = ns[p].y * x.y > 0.f ? 0.f : ns[p].y;
        }
        if (modes & (1 << (2 * p))) nf[2 * p + 1].x = 0.f;
        if (modes & (2 << (2 * p))) nf[2 * p + 1].y = 0.f;
    }
}
__device__ __forceinline__ float etd_switched_step(const ModelK<float> &P, const pvder::PkConst &K, const EtdK &Ec,
                                                   const pvder::SubPk &S, v2 *__restrict__ y, v2 *__restrict__ comp,
                                                   const lds_v4 *Mset, float h, float t0, float tq, float qref, float vdcref) {
    using namespace pvder;
    constexpr int SL[6] = {1, 4, 7, 9, 10, 11};
    const v2 h2 = bc2(0.5f * h), h4 = bc2(0.25f * h), h8 = bc2(0.125f * h), h6 = bc2(h * (float)(1.0 / 6.0)),
             h23 = bc2(h * (float)(2.0 / 3.0));
    const float t1 = t0 + 0.5f * tq, t2 = t0 + tq;
    v2 n1f[6], nf[6], ns1[6], nsl[6], s23[6], k14[6], yt[NP], n23[6];
    v2 u3[6], u4[6], u5[6], zn[6];
    const float m2a = etd_eval(P, K, Ec, S, y, t0, qref, vdcref, n1f, ns1);                // N1
    int mode = 0;        // 2 bits per phase
    // (the operator set of a pair is addressed from its mode wherever it is read: three pointers less to keep)
#define PVDER_MP(p) (Mset + ((mode >> (2 * (p))) & 3))
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const v2 u = y[3 * p + 2], m = fma2(bc2(P.KpG), u, y[3 * p + 1]);
        const v2 du = fma2(bc2(-P.wp), y[3 * p] + u, n1f[2 * p + 1]);                      // wp (iref_p - i_p - u_p)
        mode |= (__builtin_fmaf(m.y, m.y, m.x * m.x) > P.mlim2 ? ((du.x * u.x > 0.f ? 1 : 0) | (du.y * u.y > 0.f ? 2 : 0)) : 0) << (2 * p);
    }
    sw_mask(P, y, mode, n1f, ns1);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const v2 zi = y[3 * p], zu = y[3 * p + 2];
        const v2 n1i = n1f[2 * p], n1u = n1f[2 * p + 1];
        v2 e2i = bc2(0.f), e2u = bc2(0.f), ei = bc2(0.f), eu = bc2(0.f);
        matvec16(load_mat16(PVDER_MP(p), EM_E2), zi, zu, e2i, e2u);
        matvec16(load_mat16(PVDER_MP(p), EM_E), zi, zu, ei, eu);
        v2 ai = e2i, au = e2u;
        matvec16(load_mat16(PVDER_MP(p), EM_A21), n1i, n1u, ai, au);                             // U2
        yt[3 * p] = ai; yt[3 * p + 2] = au;
        u3[2 * p] = e2i; u3[2 * p + 1] = e2u;
        matvec16(load_mat16(PVDER_MP(p), EM_A31), n1i, n1u, u3[2 * p], u3[2 * p + 1]);
        u4[2 * p] = ei; u4[2 * p + 1] = eu;
        matvec16(load_mat16(PVDER_MP(p), EM_A41), n1i, n1u, u4[2 * p], u4[2 * p + 1]);
        u5[2 * p] = e2i; u5[2 * p + 1] = e2u;
        matvec16(load_mat16(PVDER_MP(p), EM_A51), n1i, n1u, u5[2 * p], u5[2 * p + 1]);
        zn[2 * p] = ei; zn[2 * p + 1] = eu;
        matvec16(load_mat16(PVDER_MP(p), EM_B1), n1i, n1u, zn[2 * p], zn[2 * p + 1]);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) yt[SL[q]] = fma2(h2, ns1[q], y[SL[q]]);
    etd_eval(P, K, Ec, S, yt, t1, qref, vdcref, nf, nsl);                                  // N2
    sw_mask(P, yt, mode, nf, nsl);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        n23[2 * p] = nf[2 * p]; n23[2 * p + 1] = nf[2 * p + 1];
        matvec16(load_mat16(PVDER_MP(p), EM_A32), nf[2 * p], nf[2 * p + 1], u3[2 * p], u3[2 * p + 1]);       // U3
        yt[3 * p] = u3[2 * p]; yt[3 * p + 2] = u3[2 * p + 1];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) { s23[q] = nsl[q]; yt[SL[q]] = fma2(h2, nsl[q], y[SL[q]]); }
    etd_eval(P, K, Ec, S, yt, t1, qref, vdcref, nf, nsl);                                  // N3
    sw_mask(P, yt, mode, nf, nsl);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        n23[2 * p] += nf[2 * p]; n23[2 * p + 1] += nf[2 * p + 1];
        matvec16(load_mat16(PVDER_MP(p), EM_A42), n23[2 * p], n23[2 * p + 1], u4[2 * p], u4[2 * p + 1]);     // U4
        matvec16(load_mat16(PVDER_MP(p), EM_A52), n23[2 * p], n23[2 * p + 1], u5[2 * p], u5[2 * p + 1]);
        yt[3 * p] = u4[2 * p]; yt[3 * p + 2] = u4[2 * p + 1];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) { s23[q] += nsl[q]; yt[SL[q]] = fma2(h2, s23[q], y[SL[q]]); }
    etd_eval(P, K, Ec, S, yt, t2, qref, vdcref, nf, nsl);                                  // N4
    sw_mask(P, yt, mode, nf, nsl);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        matvec16(load_mat16(PVDER_MP(p), EM_A54), nf[2 * p], nf[2 * p + 1], u5[2 * p], u5[2 * p + 1]);       // U5
        matvec16(load_mat16(PVDER_MP(p), EM_B4), nf[2 * p], nf[2 * p + 1], zn[2 * p], zn[2 * p + 1]);
        yt[3 * p] = u5[2 * p]; yt[3 * p + 2] = u5[2 * p + 1];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        k14[q] = ns1[q] + nsl[q];
        yt[SL[q]] = fma2(h8, s23[q], fma2(h4, ns1[q], y[SL[q]]));
    }
    etd_eval(P, K, Ec, S, yt, t1, qref, vdcref, nf, nsl);                                  // N5
    sw_mask(P, yt, mode, nf, nsl);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        matvec16(load_mat16(PVDER_MP(p), EM_B5), nf[2 * p], nf[2 * p + 1], zn[2 * p], zn[2 * p + 1]);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int g = SL[q];
        const v2 inc = fma2(h23, nsl[q], fma2(h6, k14[q], -comp[g]));
        const v2 tsum = y[g] + inc;
        comp[g] = (tsum - y[g]) - inc;
        y[g] = tsum;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) { y[3 * p] = zn[2 * p]; y[3 * p + 2] = zn[2 * p + 1]; }
    return fmaxf(m2a, max_m2(P, y));          // |m|^2 at the start and at the end of the step
#undef PVDER_MP
}

// =============================================================================================
// The 23-state ETDRK4 step with an env's three PHASES ON THREE LANES (round 4; step_etdrk4_lane_kernel).  The (i_p, x_p, u_p)
// triple of phase p lives in lane p of a QUAD (lane 3 mirrors phase a and is left out of every sum); the five shared states
// (Vdc, xDC, xQ, xPLL, delta) are carried by all four lanes, which compute them from the same operands in the same order --
// bit-equal by construction.  What couples the phases crosses the quad by DPP (quad_perm broadcasts: no LDS, no barrier): the
// three power sums and the phase currents the PLL's sequence components need.  A wave's instruction stream per step drops
// from the three phases' 39 operator products + five full right-hand sides to one phase's 13 + five reduced ones (~0.6 x),
// and a batch gives four times the waves: the lone wave per SIMD of the one-lane-per-env form becomes four.
// Same scheme, same operator sets, same guards as etd_switched_step / etd_eval (which stay the tail of the step kernels).
// =============================================================================================
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {       // lane K of the quad, to all four
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ v2 quad_bcast2(v2 v) { return pvder::mk2(quad_bcast<K>(v.x), quad_bcast<K>(v.y)); }
// sum / max over the three phase lanes, the same value (bit for bit) in every lane of the quad
__device__ __forceinline__ float quad_sum3(float v) { return (quad_bcast<0>(v) + quad_bcast<1>(v)) + quad_bcast<2>(v); }
__device__ __forceinline__ float quad_max3(float v) { return fmaxf(fmaxf(quad_bcast<0>(v), quad_bcast<1>(v)), quad_bcast<2>(v)); }

struct Phase3 {             // what a lane holds of its env
    v2 i, x, u;             // its phase
    v2 s0, s1, s2;          // (Vdc, xDC), (xQ, xPLL), (delta, 0): shared
};
struct PhaseK {             // per-lane constants of its phase
    v2 Asw, D;              // SubPk::Asw[p], D[p]
    v2 rot;                 // .y = 0 / -1 / +1: phase a / b / c (the rotation of iref_a into the phase: 0, -120, +120 degrees)
};

// N(y) of one phase + the shared slow derivatives: ni / nu = N_i / N_u of the lane's phase, nx = dx_p/dt,
// n0 = d(Vdc, xDC), n1 = d(xQ, xPLL), n2 = d(delta, 0).  Returns the lane's |m_p|^2.
__device__ __forceinline__ float etd_eval3(const ModelK<float> &P, const pvder::PkConst &K, const EtdK &Ec, const pvder::SubPk &S,
                                           const PhaseK &Q, const Phase3 &y, float th2t, float qref, float vdcref,
                                           v2 &ni, v2 &nu, v2 &nx, v2 &n0, v2 &n1, v2 &n2) {
#pragma clang fp contract(off)          // (see etd_eval)
    using namespace pvder;
    const float Vdc = y.s0.x, xDC = y.s0.y, xQ = y.s1.x, xPLL = y.s1.y, delta = y.s2.x;
    const v2 m = fma2(bc2(P.KpG), y.u, y.x);
    const float Pt = quad_sum3(dot2(m, y.i));
    const float Qs = quad_sum3(dot2(Q.Asw, y.i));
    const float Is = quad_sum3(dot2(y.i, y.i));
    const float Qpcc = __builtin_fmaf(K.qk, Is, 0.5f * Qs);
    const float qerr = qref - Qpcc;
    const float iarR = __builtin_fmaf(P.KpDC, vdcref - Vdc, xDC);
    const float iarI = __builtin_fmaf(-P.KpQ, qerr, xQ);
    float s1, c1, s2, c2;
    const float dturn = delta * (float)(1.0 / kTwoPi);
    Math<float>::sincos_turns(dturn, s1, c1);
    Math<float>::sincos_turns(th2t + dturn, s2, c2);
    // the PLL sees the positive- and negative-sequence components of the PCC voltage: all three phase currents
    const v2 ia = quad_bcast2<0>(y.i), ib = quad_bcast2<1>(y.i), ic = quad_bcast2<2>(y.i);
    const v2 sb = ib + ic, db = ib - ic;
    const v2 base = fma2(bc2(-0.5f), sb, ia);
    const v2 jh3 = mk2(-kHalfSqrt3, kHalfSqrt3);
    const v2 rdb = swp(db) * jh3;
    const v2 i1 = base + rdb, i2 = base - rdb;
    const v2 jk3 = mk2(-P.kvi3I, P.kvi3I);
    v2 V1 = fma2(bc2(P.kvi3R), i1, S.A0); V1 = fma2(jk3, swp(i1), V1);
    v2 V2c = mk2(P.kvi3R, -P.kvi3R) * i2; V2c = fma2(bc2(-P.kvi3I), swp(i2), V2c);
    const v2 t1 = V1 * mk2(c1, s1);
    const v2 t2 = fma2(V2c, mk2(c2, s2), t1);
    const float vd = t2.x + t2.y;
    const float dw = __builtin_fmaf(P.KpP, vd, xPLL);
    const float dwr = dw * P.iwb;
    const float cmf = K.hiLf * Vdc;
    v2 n = fma2(bc2(cmf), y.x, Q.D);
    n = fma2(bc2(__builtin_fmaf(cmf, P.KpG, -Ec.c0kp)), y.u, n);
    ni = fma2(mk2(dwr, -dwr), swp(y.i), n);
    // N_u = wp * iref_p,  iref_p = iref_a rotated into the lane's phase
    const v2 wra = mk2(P.wp * iarR, P.wp * iarI);
    {   // (phase a: wra; b, c: -wra/2 -/+ j sqrt(3)/2 wra -- the operations of etd_eval, so that the two forms agree bit for bit;
        //  Q.rot.y = 0, -1, +1 selects)
        const v2 hw = bc2(-0.5f) * wra;
        const v2 jw = swp(wra) * jh3;
        nu = Q.rot.y == 0.f ? wra : (Q.rot.y < 0.f ? hw - jw : hw + jw);
    }
    nx = bc2(P.KiG) * y.u;
    const float ppv = ppv_of(P, S.pv1, Vdc);
    const float dVdc = __builtin_fmaf(-0.25f * Vdc, Pt, ppv) * Math<float>::rcp(Vdc) * P.iC;
    float dxDC = P.KiDC * (vdcref - Vdc);
    float dxQ = -P.KiQ * qerr;
    if (__builtin_fmaf(iarI, iarI, iarR * iarR) > P.ilim2) {
        dxDC = (dxDC * xDC > 0.f) ? 0.f : dxDC;
        dxQ = (dxQ * xQ > 0.f) ? 0.f : dxQ;
    }
    n0 = mk2(dVdc, dxDC);
    n1 = mk2(dxQ, P.KiP * vd);
    n2 = mk2(dw, 0.f);
    return dot2(m, m);
}

// the model's sign tests on dx_p/dt at a stage state, and the frozen rows of N_u (the lane's phase; `mode`: its two bits)
__device__ __forceinline__ void sw_mask3(const ModelK<float> &P, v2 x, v2 u, int mode, v2 &nu, v2 &nx) {
    using namespace pvder;
    const v2 m = fma2(bc2(P.KpG), u, x);
    if (__builtin_fmaf(m.y, m.y, m.x * m.x) > P.mlim2) {
        nx.x = nx.x * x.x > 0.f ? 0.f : nx.x;
        nx.y = nx.y * x.y > 0.f ? 0.f : nx.y;
    }
    if (mode & 1) nu.x = 0.f;
    if (mode & 2) nu.y = 0.f;
}

// one ETDRK4 step (etd_switched_step, a phase per lane); comp: Kahan compensation of (x, s0, s1, s2).  Returns the largest
// |m_p|^2 of the ENV at the start and at the end of the step (the same in the quad's lanes).
__device__ __forceinline__ float etd_switched_step3(const ModelK<float> &P, const pvder::PkConst &K, const EtdK &Ec,
                                                    const pvder::SubPk &S, const PhaseK &Q, Phase3 &y, v2 (&comp)[4],
                                                    const lds_v4 *Mset, float h, float t0, float tq, float qref, float vdcref) {
    using namespace pvder;
    const v2 h2 = bc2(0.5f * h), h4 = bc2(0.25f * h), h8 = bc2(0.125f * h), h6 = bc2(h * (float)(1.0 / 6.0)),
             h23 = bc2(h * (float)(2.0 / 3.0));
    const float t1 = t0 + 0.5f * tq, t2 = t0 + tq;
    v2 n1i, n1u, ni, nu;
    v2 ns1[4], nsl[4], s23[4], k14[4];        // slow derivatives: x, s0, s1, s2
    const float m2a = etd_eval3(P, K, Ec, S, Q, y, t0, qref, vdcref, n1i, n1u, ns1[0], ns1[1], ns1[2], ns1[3]);       // N1
    int mode = 0;
    {
        const v2 m = fma2(bc2(P.KpG), y.u, y.x);
        const v2 du = fma2(bc2(-P.wp), y.i + y.u, n1u);                                    // wp (iref_p - i_p - u_p)
        mode = __builtin_fmaf(m.y, m.y, m.x * m.x) > P.mlim2 ? ((du.x * y.u.x > 0.f ? 1 : 0) | (du.y * y.u.y > 0.f ? 2 : 0)) : 0;
    }
    const lds_v4 *Mp = Mset + mode;
    sw_mask3(P, y.x, y.u, mode, n1u, ns1[0]);
    v2 e2i = bc2(0.f), e2u = bc2(0.f), ei = bc2(0.f), eu = bc2(0.f);
    matvec16(load_mat16(Mp, EM_E2), y.i, y.u, e2i, e2u);
    matvec16(load_mat16(Mp, EM_E), y.i, y.u, ei, eu);
    Phase3 yt;
    yt.i = e2i; yt.u = e2u;
    matvec16(load_mat16(Mp, EM_A21), n1i, n1u, yt.i, yt.u);                               // U2
    v2 u3i = e2i, u3u = e2u, u4i = ei, u4u = eu, u5i = e2i, u5u = e2u, zi = ei, zu = eu;
    matvec16(load_mat16(Mp, EM_A31), n1i, n1u, u3i, u3u);
    matvec16(load_mat16(Mp, EM_A41), n1i, n1u, u4i, u4u);
    matvec16(load_mat16(Mp, EM_A51), n1i, n1u, u5i, u5u);
    matvec16(load_mat16(Mp, EM_B1), n1i, n1u, zi, zu);
    yt.x = fma2(h2, ns1[0], y.x); yt.s0 = fma2(h2, ns1[1], y.s0); yt.s1 = fma2(h2, ns1[2], y.s1); yt.s2 = fma2(h2, ns1[3], y.s2);
    etd_eval3(P, K, Ec, S, Q, yt, t1, qref, vdcref, ni, nu, nsl[0], nsl[1], nsl[2], nsl[3]);                        // N2
    sw_mask3(P, yt.x, yt.u, mode, nu, nsl[0]);
    v2 n23i = ni, n23u = nu;
    matvec16(load_mat16(Mp, EM_A32), ni, nu, u3i, u3u);                                   // U3
    yt.i = u3i; yt.u = u3u;
#pragma unroll
    for (int q = 0; q < 4; ++q) s23[q] = nsl[q];
    yt.x = fma2(h2, nsl[0], y.x); yt.s0 = fma2(h2, nsl[1], y.s0); yt.s1 = fma2(h2, nsl[2], y.s1); yt.s2 = fma2(h2, nsl[3], y.s2);
    etd_eval3(P, K, Ec, S, Q, yt, t1, qref, vdcref, ni, nu, nsl[0], nsl[1], nsl[2], nsl[3]);                        // N3
    sw_mask3(P, yt.x, yt.u, mode, nu, nsl[0]);
    n23i += ni; n23u += nu;
    matvec16(load_mat16(Mp, EM_A42), n23i, n23u, u4i, u4u);                               // U4
    matvec16(load_mat16(Mp, EM_A52), n23i, n23u, u5i, u5u);
    yt.i = u4i; yt.u = u4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) s23[q] += nsl[q];
    yt.x = fma2(h2, s23[0], y.x); yt.s0 = fma2(h2, s23[1], y.s0); yt.s1 = fma2(h2, s23[2], y.s1); yt.s2 = fma2(h2, s23[3], y.s2);
    etd_eval3(P, K, Ec, S, Q, yt, t2, qref, vdcref, ni, nu, nsl[0], nsl[1], nsl[2], nsl[3]);                        // N4
    sw_mask3(P, yt.x, yt.u, mode, nu, nsl[0]);
    matvec16(load_mat16(Mp, EM_A54), ni, nu, u5i, u5u);                                   // U5
    matvec16(load_mat16(Mp, EM_B4), ni, nu, zi, zu);
    yt.i = u5i; yt.u = u5u;
#pragma unroll
    for (int q = 0; q < 4; ++q) k14[q] = ns1[q] + nsl[q];
    yt.x = fma2(h8, s23[0], fma2(h4, ns1[0], y.x)); yt.s0 = fma2(h8, s23[1], fma2(h4, ns1[1], y.s0));
    yt.s1 = fma2(h8, s23[2], fma2(h4, ns1[2], y.s1)); yt.s2 = fma2(h8, s23[3], fma2(h4, ns1[3], y.s2));
    etd_eval3(P, K, Ec, S, Q, yt, t1, qref, vdcref, ni, nu, nsl[0], nsl[1], nsl[2], nsl[3]);                        // N5
    sw_mask3(P, yt.x, yt.u, mode, nu, nsl[0]);
    matvec16(load_mat16(Mp, EM_B5), ni, nu, zi, zu);
    // Kahan-compensated slow update:  y+ = y + h (N1 + N4)/6 + 2h N5/3
    auto kahan = [&](v2 &yq, v2 &cq, v2 n5, v2 n14) {
        const v2 inc = fma2(h23, n5, fma2(h6, n14, -cq));
        const v2 tsum = yq + inc;
        cq = (tsum - yq) - inc;
        yq = tsum;
    };
    kahan(y.x, comp[0], nsl[0], k14[0]); kahan(y.s0, comp[1], nsl[1], k14[1]);
    kahan(y.s1, comp[2], nsl[2], k14[2]); kahan(y.s2, comp[3], nsl[3], k14[3]);
    y.i = zi; y.u = zu;
    const v2 mn = fma2(bc2(P.KpG), y.u, y.x);
    return quad_max3(fmaxf(m2a, __builtin_fmaf(mn.y, mn.y, mn.x * mn.x)));
}

// Which env steps the hybrid kernel takes (while it follows the launches: EtdArgs.sat_hybrid).  An env step that never comes within
// 0.4 % of the duty-cycle limit has no anti-windup event in it and no held mode (the switched sets start at 0.998: sat2_smooth; an
// input jump counts as an entry beyond 1 - 4e-3: hyb::input_jump) -- the 23-state tail's own result stands.  So the tail hands over
// an env that STARTS the step within 0.4 % and, AFTER the step, one whose duty cycle came that close during it (the step is
// discarded: nothing of it was stored); every other env near the limit costs what it cost before round 6 (42 us, not milliseconds).
constexpr float kHybTouch2 = 0.996f * 0.996f;
// the phase-per-lane form (etd_tail3) has no such check: a workgroup with an env beyond this takes the lane-per-env form (one env step
// does not go from here to the limit without an event -- and an event's own step is on the held sets wherever the kernel is not yet
// following: the same bound as EtdK.sat2_flag)
constexpr float kHybNear2 = 0.9875f * 0.9875f;
constexpr int kHybMaxSub = 16;                      // sim time steps per env step the hybrid kernel takes (PVDER-v0: 15)
constexpr int kHybHoldLaunches = 256;               // auto mode: the hybrid kernel follows this many step launches behind the last sighting
// The env transition of the quads named by (e_quad, mine) on the phase-per-lane form; Msw: the staged operator sets;
// LP: per-lane parameters (lane_par); stats / stats_fresh as epilogue takes them; act / out: the step's action row and
// output block.  Lane 0 of a quad is the env for everything outside the integration (epilogue).
template <bool LP>
__device__ __forceinline__ void etd_tail3(const EtdArgs &A, const float *Msw, int e_quad, bool mine, const float *__restrict__ lane_par,
                                          float *stats, int stats_fresh, const uint8_t *__restrict__ action, float *__restrict__ out,
                                          bool stats_local = false) {
    using namespace pvder;
    const StepK<float> &C = A.C;
    const EtdK &Ec = A.Ec;
    const int n_env = A.n_env;
    const size_t N = (size_t)n_env;
    constexpr int NS = 23;
    const int ph = threadIdx.x & 3;                      // 0, 1, 2: phases a, b, c; 3: mirrors phase a (left out of every sum)
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(C, A.fstate, A.istate, action, n_env, L, ys, 0, e_quad);
    L.live = L.live && mine;
    L.run = L.run && mine;
    ModelK<float> Pl = A.P;
    if constexpr (LP) {
        const float *__restrict__ lp = lane_par + L.ec;
        Pl.iC = lp[(size_t)PVDER_LPF_IC * N];
        Pl.KpDC = lp[(size_t)PVDER_LPF_KP_DC * N]; Pl.KiDC = lp[(size_t)PVDER_LPF_KI_DC * N];
        Pl.KpQ = lp[(size_t)PVDER_LPF_KP_Q * N];   Pl.KiQ = lp[(size_t)PVDER_LPF_KI_Q * N];
        Pl.pvA = lp[(size_t)PVDER_LPF_PVA * N];    Pl.pvB = lp[(size_t)PVDER_LPF_PVB * N];
    }
    const ModelK<float> &P = LP ? Pl : A.P;
    if (__any((int)L.run)) {                             // (every lane of a quad runs or none: no DPP partner is ever masked off)
        const int p3 = ph == 3 ? 0 : ph;
        Phase3 y;
        y.i = mk2(p3 == 0 ? ys[0] : (p3 == 1 ? ys[6] : ys[12]), p3 == 0 ? ys[1] : (p3 == 1 ? ys[7] : ys[13]));
        y.x = mk2(p3 == 0 ? ys[2] : (p3 == 1 ? ys[8] : ys[14]), p3 == 0 ? ys[3] : (p3 == 1 ? ys[9] : ys[15]));
        y.u = mk2(p3 == 0 ? ys[4] : (p3 == 1 ? ys[10] : ys[16]), p3 == 0 ? ys[5] : (p3 == 1 ? ys[11] : ys[17]));
        y.s0 = mk2(ys[18], ys[19]); y.s1 = mk2(ys[20], ys[21]); y.s2 = mk2(ys[22], 0.f);
        v2 comp[4] = {bc2(0.f), bc2(0.f), bc2(0.f), bc2(0.f)};
        const PkConst K = make_pkconst(P);
        PhaseK Q;
        Q.rot = p3 == 0 ? mk2(1.f, 0.f) : (p3 == 1 ? mk2(-0.5f, -1.f) : mk2(-0.5f, 1.f));      // (.y: which of phases a / b / c)
        SubPk S;
        bool fail = false;
        float m2_peak = 0.f;                              // max |m_p|^2 of the env over the env step (the lane form keeps the same)
        int prev_row = L.ksub > 0 ? event_row(L.ksub - 1, C.ev_s0, C.ev_ds, C.ev_count) : 0;
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            const bool input_jump = row != prev_row;
            prev_row = row;
            const bool lane_fine = (s == 0) || input_jump;        // two-rate schedule, per env (etdrk4_full_tail)
            make_sub_pk(P, A.ev_vg[(size_t)row * N + L.ec], A.ev_si[(size_t)row * N + L.ec], S);
            Q.Asw = p3 == 0 ? S.Asw[0] : (p3 == 1 ? S.Asw[1] : S.Asw[2]);
            Q.D = p3 == 0 ? S.D[0] : (p3 == 1 ? S.D[1] : S.D[2]);
            for (int pass = 0; pass < 2; ++pass) {
                const bool fine = pass == 0;
                // (a quad's lanes share their env's schedule; the lanes of a wave whose envs take the other pass sit this one out
                //  TOGETHER with their quad: the DPP exchanges stay inside quads)
                if (!__builtin_amdgcn_readfirstlane(__any((int)(lane_fine == fine)))) continue;
                const int nst = fine ? (C.graded ? 2 : C.micro) : C.micro2;
                float t0 = 0.f;
                for (int j = 0; j < nst; ++j) {
                    const int set = fine ? ((C.graded && j > 0) ? 2 : 0) : 1;
                    const float h = set == 0 ? Ec.hset[0] : (set == 1 ? Ec.hset[1] : Ec.hset[2]);
                    const float tq = 120.f * h;
                    const float guard = (j == 0 && input_jump) ? Ec.sat2_event : Ec.sat2_smooth;
                    const lds_v4 *Ms = (const lds_v4 *)Msw + (kSwModes * kMatCount * 4) * set;
                    Phase3 yn = y;
                    v2 cn[4] = {comp[0], comp[1], comp[2], comp[3]};
                    const float mm2 = etd_switched_step3(P, K, Ec, S, Q, yn, cn, Ms, h, t0, tq, L.qref, L.vdcref);
                    if (lane_fine == fine) {
                        y = yn;
                        comp[0] = cn[0]; comp[1] = cn[1]; comp[2] = cn[2]; comp[3] = cn[3];
                        m2_peak = fmaxf(m2_peak, mm2);
                        if (mm2 > guard && C.off_range == PVDER_OFF_RANGE_TERMINATE) fail = true;
                    }
                    t0 += tq;
                }
            }
        }
        // the env's 23 states back in the order of the state rows (every lane of the quad holds them; lane 0 stores)
        const v2 ib = quad_bcast2<1>(y.i), xb = quad_bcast2<1>(y.x), ub = quad_bcast2<1>(y.u);
        const v2 ic = quad_bcast2<2>(y.i), xc = quad_bcast2<2>(y.x), uc = quad_bcast2<2>(y.u);
        const v2 ia = quad_bcast2<0>(y.i), xa = quad_bcast2<0>(y.x), ua = quad_bcast2<0>(y.u);
        if (L.run) {
            ys[0] = ia.x; ys[1] = ia.y; ys[2] = xa.x; ys[3] = xa.y; ys[4] = ua.x; ys[5] = ua.y;
            ys[6] = ib.x; ys[7] = ib.y; ys[8] = xb.x; ys[9] = xb.y; ys[10] = ub.x; ys[11] = ub.y;
            ys[12] = ic.x; ys[13] = ic.y; ys[14] = xc.x; ys[15] = xc.y; ys[16] = uc.x; ys[17] = uc.y;
            ys[18] = y.s0.x; ys[19] = y.s0.y; ys[20] = y.s1.x; ys[21] = y.s1.y; ys[22] = y.s2.x;
            L.ksub += C.n_sub;
            L.steps += 1;
            L.fail = fail;
            // (the same hand-over as the lane-per-env form, on the same number: see kHybTouch2)
            if (A.sat_hybrid && m2_peak > kHybTouch2 * P.mlim2) {
                if (ph == 0) A.istate[(size_t)PVDER_I_STATUS * (size_t)n_env + L.e] = kDeferred;
                L.run = false;
                L.live = false;
            } else if (A.hyb_flag && !A.sat_hybrid && m2_peak > Ec.sat2_flag && ph == 0) {
                *(volatile int *)A.hyb_flag = 1;
            }
        }
    }
    L.live = L.live && ph == 0;
    L.run = L.run && ph == 0;
    epilogue<float, 3>(P, C, A.R, L, ys, A.fstate, A.istate, A.ev_vg, A.ev_si, out, stats, n_env, stats_fresh, false, 0.f, 0.f,
                       LP ? lane_par : nullptr, stats_local);
}

// The 23-state ETDRK4 path, the tail of step_etdrk4_bal_kernel<3>: its workgroups run it for the lanes
// they must defer (`defer`; their state in memory is untouched, every other lane of the workgroup is
// finished).  It takes a LAUNDERED copy of the kernarg address and re-reads every argument through
// it, so that nothing of the fast path's register state stays live across the hot loop on its
// behalf (sharing the arguments cost the hot loop AGPR copies; a real function call costs the kernel
// a scratch allocation -- AMDGPU callees save callee-saved VGPRs even when they never return -- and
// with scratch enabled the workgroups start ~2 us later).
typedef __attribute__((address_space(4))) const EtdArgs *KernargPtr;
// Auto mode of the duty-cycle-limit kernel (launch_plan): the host must see an env's flag within a few launches of its being raised,
// so it keeps at most 16 step launches in flight -- by this word, which workgroup 0 of every step launch writes as it STARTS (one
// posted store behind two scalar loads, while the wave waits for its state rows anyway).  Round 6's first form, a stream marker
// (hipEventRecord) every 8 launches, cost the headline 0.4-0.5 us per launch on the GPU's side of the queue.
template <int KOFF>
__device__ __forceinline__ void hyb_progress() {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    typedef __attribute__((address_space(4))) const char *KernargBytes;
    KernargPtr ka = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
    asm volatile("" : "+s"(ka));
    int *const w = ka->hyb_flag;
    const int seq = ka->hyb_seq;
    if (w != nullptr && seq != 0) __hip_atomic_store(w + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// LP (step_etdrk4_lane_kernel): every env with its OWN values of the parameters that stay outside the exponential operator
// (lane_par: [PVDER_LPF_ROWS][n_env] folded constants, pvder_fold_lane_params) -- the lane's ModelK is the launch's with
// those seven fields replaced (VGPRs instead of SGPRs); stats_fresh: this pass starts the step's statistics.
// ---- envs at / near the duty-cycle limit (round 6): the hybrid scheme of pvder_sat_hybrid.hpp, in a kernel of its own behind the step
// kernel -- fp64, 6 KB of private arrays per lane: inside the step kernels it would cost every launch their occupancy and a scratch ring.

template <bool LP = false, bool T3 = false>
__device__ __forceinline__ void etdrk4_full_tail(const EtdArgs &A, bool defer, const uint8_t *act = nullptr,
                                                 long long out_off = 0, const float *__restrict__ lane_par = nullptr,
                                                 int stats_fresh = 0) {
    const int run_status = 0;
    const ModelK<float> &P0 = A.P;
    const StepK<float> &C = A.C;
    const ResetK &R = A.R;
    const EtdK &Ec = A.Ec;
    const float *__restrict__ coef = A.coef;
    float *__restrict__ fstate = A.fstate;
    int32_t *__restrict__ istate = A.istate;
    float *__restrict__ ev_vg = A.ev_vg;
    float *__restrict__ ev_si = A.ev_si;
    const uint8_t *__restrict__ action = act ? act : A.action;       // (rollouts: the step's own action row / output block)
    float *__restrict__ out = A.out + out_off;
    float *__restrict__ stats = A.stats;
    const int n_env = A.n_env;
    constexpr int NS = 23, NP = pvder::NP;
    const size_t N = (size_t)n_env;
    // The 10 coefficient matrices (160 floats) do not fit the scalar register file next to the
    // model constants (they would be spilled to VGPR lanes and read back with v_readlane in the
    // hot loop): stage the block once in LDS and read it with broadcast ds_read (every lane reads
    // the same address: conflict-free, no VALU issue slot).
#ifdef PVDER_RK4_FALLBACK
    __shared__ __attribute__((aligned(16))) float Mlds[kCoefFloats];
    for (int k = threadIdx.x; k < kCoefFloats; k += kBlock) Mlds[k] = coef[k];
#endif
    __shared__ __attribute__((aligned(16))) float Msw[kSwFloats];          // the operator sets per anti-windup mode (mode 0: the complex block)
    for (int k = threadIdx.x; k < kSwFloats; k += kBlock) {
        // buffer: [step size][mode][matrix][word j][4 floats]  ->  LDS: [step size][matrix][word j][mode][4 floats] (load_mat16)
        const int c = k & 3, j = (k >> 2) & 3, mat = (k >> 4) % kMatCount, sm = (k >> 4) / kMatCount, mode = sm % kSwModes, set = sm / kSwModes;
        Msw[((((set * kMatCount + mat) * 4 + j) * kSwModes + mode) << 2) + c] = coef[kCoefFloats + k];
    }
    // PACK the workgroup's deferred envs into its first lanes (deterministic order: by thread index).  The step loop below is
    // a single wave's dependent chain, and what it costs depends on how the deferred lanes are spread: ONE lane in each of a
    // workgroup's four waves -- what a batch looks like while it approaches the duty-cycle limit, a few envs per wave behind
    // every insolation event -- runs the loop in 150 us, the same four envs in ONE wave in 63 us, and so do 256 of them
    // (tools/sparse_saturated_envs.py, profiles/r04_sparse_saturated_envs.txt: 93 us per batched step with one deferred env per
    // workgroup, 155 with one per wave, 95 with every env deferred).  The states come from memory anyway.
    __shared__ int wave_cnt[kBlock / 64];
    __shared__ int packed[kBlock];
    {
        const unsigned long long bal = __ballot((int)defer);
        if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (int)__popcll(bal);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) base += w < (int)(threadIdx.x >> 6) ? wave_cnt[w] : 0;
        if (defer) packed[base + (int)__popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int)threadIdx.x;
        __syncthreads();
    }
    int n_def = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) n_def += wave_cnt[w];
    // T3 (step_etdlin_kernel_w1: the step kernel of batches that give a SIMD one wave): a workgroup with at most 64 deferred envs
    // -- one env at the duty-cycle limit among 65 536, a Vdc_ref ramp approaching the limit, an env behind an event -- takes them
    // through the phase-per-lane form (etd_tail3: a quad of lanes per env, 0.6 x the instructions of the wave the batch waits
    // for): that env costs the batch 54 instead of 85 us per step, the action-4 ramp 60-93 instead of 74-124 us while it
    // approaches the limit; more than 64: the lane-per-env form below (the throughput form, 97 against 193 us with every env at
    // the limit).  The two forms agree bit for bit (tools/lane_forms_bitwise.py): which one advances an env is a question of
    // time only.  (Side by side they need 16 registers more than the 256 two waves per SIMD leave each -- spilled into the
    // kernel's entry block, 7.65 -> 8.16 us on the standard path: hence a kernel of its own, one wave per SIMD, where they
    // sit in AGPRs: profiles/r04_ab_tail3_in_step_kernel.txt.)
    // (sat_hybrid: an env at / near the duty-cycle limit is NOT stepped here -- step_sat_hybrid_kernel, which follows this launch,
    //  takes it: marked kDeferred by the lane-per-env form below, which a workgroup that holds such an env therefore takes)
    bool any_hyb = false;
    if constexpr (T3 && !LP) {
        if (A.sat_hybrid && n_def <= kBlock / 4) {
            const int q = (int)(threadIdx.x >> 2), ph = (int)(threadIdx.x & 3);
            bool near = false;
            if (q < n_def && ph < 3) {
                const size_t e = (size_t)(blockIdx.x * kBlock) + (size_t)packed[q];
                if (e < N && istate[(size_t)PVDER_I_STATUS * N + e] == 0) {
                    const float xr = fstate[(size_t)(6 * ph + 2) * N + e], xi = fstate[(size_t)(6 * ph + 3) * N + e];
                    const float ur = fstate[(size_t)(6 * ph + 4) * N + e], ui = fstate[(size_t)(6 * ph + 5) * N + e];
                    const float mr = __builtin_fmaf(P0.KpG, ur, xr), mi = __builtin_fmaf(P0.KpG, ui, xi);
                    near = __builtin_fmaf(mi, mi, mr * mr) > kHybNear2 * P0.mlim2;
                }
            }
            any_hyb = __syncthreads_or((int)near) != 0;
        }
    }
    if constexpr (T3 && !LP) {
        if (n_def <= kBlock / 4 && !any_hyb) {
            const int q = (int)(threadIdx.x >> 2);
            etd_tail3<false>(A, Msw, (int)(blockIdx.x * kBlock) + (q < n_def ? packed[q] : (int)threadIdx.x), q < n_def, nullptr, stats,
                             stats_fresh, action, out);
            return;
        }
    }
    const bool mine = (int)threadIdx.x < n_def;
    PVDER_TRACE_COLD_MAX(13, true);
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(C, fstate, istate, action, n_env, L, ys, run_status,
                        (int)(blockIdx.x * kBlock) + (mine ? packed[threadIdx.x] : (int)threadIdx.x));
    L.live = L.live && mine;                       // everyone else in the workgroup is done already
    L.run = L.run && mine;
    ModelK<float> Pl = P0;
    if constexpr (LP) {
        const float *__restrict__ lp = lane_par + L.ec;
        Pl.iC = lp[(size_t)PVDER_LPF_IC * N];
        Pl.KpDC = lp[(size_t)PVDER_LPF_KP_DC * N]; Pl.KiDC = lp[(size_t)PVDER_LPF_KI_DC * N];
        Pl.KpQ = lp[(size_t)PVDER_LPF_KP_Q * N];   Pl.KiQ = lp[(size_t)PVDER_LPF_KI_Q * N];
        Pl.pvA = lp[(size_t)PVDER_LPF_PVA * N];    Pl.pvB = lp[(size_t)PVDER_LPF_PVB * N];
    }
    const ModelK<float> &P = LP ? Pl : P0;
    PVDER_TRACE_COLD_MAX(14, __any((int)L.run));
#if defined(PVDER_TRACE) && defined(PVDER_TRACE_COLD)
    const unsigned long long trc_c0 = clock64(), trc_w0 = wall_clock64();      // (shader cycles / 100 MHz wall clock at the loop's start)
#endif
    if (A.sat_hybrid && L.run) {
        float m2 = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float mr = __builtin_fmaf(P.KpG, ys[6 * p + 4], ys[6 * p + 2]), mi = __builtin_fmaf(P.KpG, ys[6 * p + 5], ys[6 * p + 3]);
            m2 = fmaxf(m2, __builtin_fmaf(mi, mi, mr * mr));
        }
        if (m2 > kHybTouch2 * P.mlim2) {
            istate[(size_t)PVDER_I_STATUS * N + L.e] = kDeferred;        // step_sat_hybrid_kernel advances it (this launch stores nothing of it)
            L.run = false;
            L.live = false;
        }
    } else if (A.hyb_flag && L.run) {
        // (the hybrid kernel does not follow this launch, yet here is an env that wants it -- an event took it to the limit
        //  in one step: this one step on the held operator sets, and the flag)
        float m2 = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float mr = __builtin_fmaf(P.KpG, ys[6 * p + 4], ys[6 * p + 2]), mi = __builtin_fmaf(P.KpG, ys[6 * p + 5], ys[6 * p + 3]);
            m2 = fmaxf(m2, __builtin_fmaf(mi, mi, mr * mr));
        }
        if (m2 > Ec.sat2_flag) *(volatile int *)A.hyb_flag = 1;
    }
    if (L.run) {
        v2 y[NP], comp[NP];
        to_pairs(ys, y);
#pragma unroll
        for (int k = 0; k < NP; ++k) comp[k] = pvder::bc2(0.f);
        const pvder::PkConst K = pvder::make_pkconst(P);
        // slow pairs of the packed layout: x_a, x_b, x_c, (Vdc,xDC), (xQ,xPLL), (delta,0)
        constexpr int SL[6] = {1, 4, 7, 9, 10, 11};
        pvder::SubPk S;
        int prev_row = L.ksub > 0 ? event_row(L.ksub - 1, C.ev_s0, C.ev_ds, C.ev_count) : 0;
        float m2_peak = 0.f;                                  // max |m_p|^2 over the env step (every step's start and end)
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            const bool input_jump = row != prev_row;          // a grid-voltage / insolation event starts here
            prev_row = row;
            // two-rate schedule: the sim time step that starts at an input jump (new action: s == 0;
            // new event row) takes C.micro steps, every other one C.micro2 (DESIGN.md "Step schedule").
            // PER LANE: the lanes whose env has a jump here take the fine pass, the others the coarse one (a wave
            // whose envs run in step executes one of the two) -- an env's trajectory does not depend on which envs
            // share its wave (round 1 took the fine rate for the whole wave if ANY lane had a jump)
            const bool lane_fine = (s == 0) || input_jump;
            pvder::make_sub_pk(P, ev_vg[(size_t)row * N + L.ec], ev_si[(size_t)row * N + L.ec], S);
            for (int pass = 0; pass < 2; ++pass) {
            const bool fine = pass == 0;
            if (lane_fine != fine) continue;
            const int nst = fine ? (C.graded ? 2 : C.micro) : C.micro2;
            float t0 = 0.f;                                   // 2*theta (turns) since the start of this sim time step
            for (int j = 0; j < nst; ++j) {
                // graded schedule: first step 1/(60 micro), then ONE step over the remainder
                const int set = fine ? ((C.graded && j > 0) ? 2 : 0) : 1;
                const float h = set == 0 ? Ec.hset[0] : (set == 1 ? Ec.hset[1] : Ec.hset[2]);
                const v2 h2 = pvder::bc2(0.5f * h), h4 = pvder::bc2(0.25f * h), h8 = pvder::bc2(0.125f * h),
                         h6 = pvder::bc2(h * (float)(1.0 / 6.0)), h23 = pvder::bc2(h * (float)(2.0 / 3.0));
                const float tq = 120.f * h;                   // 2*theta advances 2 turns per 1/60 s
                // (the guard from a SCALAR load, selected by value: left to itself hipcc selects the ADDRESS by the per-lane
                // condition and fetches the guard with a vector load from the kernel-argument segment -- a memory round trip
                // inside every step of this latency-bound loop)
                float g_sm = Ec.sat2_smooth;
                asm volatile("" : "+s"(g_sm));
                // (sat2_event / sat2_smooth = (0.97 / 0.998)^2: one scalar instead of two -- this path has no SGPR to spare)
                const float guard = (j == 0 && input_jump) ? g_sm * 0.94467492f : g_sm;
#ifndef PVDER_RK4_FALLBACK
                // ONE step function for every lane of this path: the (i_p, u_p) pairs on the real 4x4 operator set of the
                // anti-windup mode each is in at the start of the step -- mode 0 (below the limit: every pair of an env that
                // is merely off balance) is the complex 2x2 system itself.  (Until round 3 a step was first taken on the
                // complex block and, if the duty cycle met the guard, taken AGAIN on the switched sets: an env at the limit
                // -- what this path mostly serves since slightly off-balance envs stay on the matrix-core path -- paid
                // for every step twice.)
                const lds_v4 *Ms = (const lds_v4 *)Msw + (kSwModes * kMatCount * 4) * set;
                asm volatile("" : "+v"(Ms));
                const float t2 = t0 + tq;
                const float mm2 = etd_switched_step(P, K, Ec, S, y, comp, Ms, h, t0, tq, L.qref, L.vdcref);
                m2_peak = fmaxf(m2_peak, mm2);
                if (mm2 > guard && C.off_range == PVDER_OFF_RANGE_TERMINATE) L.fail = true;     // the episode ends here (epilogue)
#else
                // opaque per-iteration copy of the LDS base: the 160 coefficients are re-read from
                // LDS every step (40 ds_read_b128) instead of being hoisted into 160 registers
                const lds_v4 *Mq = (const lds_v4 *)Mlds + 2 * kMatCount * set;
                asm volatile("" : "+v"(Mq));
                const float t1 = t0 + 0.5f * tq, t2 = t0 + tq;
                v2 n1f[6], nf[6], ns1[6], nsl[6], s23[6], k14[6], yt[NP], n23[6];
                v2 u3[6], u4[6], u5[6], zn[6];    // running sums of U3, U4, U5, z+   (index 2p = i_p, 2p+1 = u_p)
                const float m2a = etd_eval(P, K, Ec, S, y, t0, L.qref, L.vdcref, n1f, ns1);      // N1
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const v2 zi = y[3 * p], zu = y[3 * p + 2];
                    const v2 n1i = n1f[2 * p], n1u = n1f[2 * p + 1];
                    v2 e2i = pvder::bc2(0.f), e2u = pvder::bc2(0.f), ei = pvder::bc2(0.f), eu = pvder::bc2(0.f);
                    matvec(load_mat(Mq, EM_E2), zi, zu, e2i, e2u);                    // E2 z
                    matvec(load_mat(Mq, EM_E), zi, zu, ei, eu);                       // E z
                    v2 ai = e2i, au = e2u;
                    matvec(load_mat(Mq, EM_A21), n1i, n1u, ai, au);                   // U2
                    yt[3 * p] = ai; yt[3 * p + 2] = au;
                    u3[2 * p] = e2i; u3[2 * p + 1] = e2u;
                    matvec(load_mat(Mq, EM_A31), n1i, n1u, u3[2 * p], u3[2 * p + 1]);
                    u4[2 * p] = ei; u4[2 * p + 1] = eu;
                    matvec(load_mat(Mq, EM_A41), n1i, n1u, u4[2 * p], u4[2 * p + 1]);
                    u5[2 * p] = e2i; u5[2 * p + 1] = e2u;
                    matvec(load_mat(Mq, EM_A51), n1i, n1u, u5[2 * p], u5[2 * p + 1]);
                    zn[2 * p] = ei; zn[2 * p + 1] = eu;
                    matvec(load_mat(Mq, EM_B1), n1i, n1u, zn[2 * p], zn[2 * p + 1]);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) yt[SL[q]] = pvder::fma2(h2, ns1[q], y[SL[q]]);
                etd_eval(P, K, Ec, S, yt, t1, L.qref, L.vdcref, nf, nsl);             // N2
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    n23[2 * p] = nf[2 * p]; n23[2 * p + 1] = nf[2 * p + 1];
                    matvec(load_mat(Mq, EM_A32), nf[2 * p], nf[2 * p + 1], u3[2 * p], u3[2 * p + 1]);   // U3
                    yt[3 * p] = u3[2 * p]; yt[3 * p + 2] = u3[2 * p + 1];
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) { s23[q] = nsl[q]; yt[SL[q]] = pvder::fma2(h2, nsl[q], y[SL[q]]); }
                etd_eval(P, K, Ec, S, yt, t1, L.qref, L.vdcref, nf, nsl);             // N3
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    n23[2 * p] += nf[2 * p]; n23[2 * p + 1] += nf[2 * p + 1];
                    matvec(load_mat(Mq, EM_A42), n23[2 * p], n23[2 * p + 1], u4[2 * p], u4[2 * p + 1]);   // U4
                    matvec(load_mat(Mq, EM_A52), n23[2 * p], n23[2 * p + 1], u5[2 * p], u5[2 * p + 1]);
                    yt[3 * p] = u4[2 * p]; yt[3 * p + 2] = u4[2 * p + 1];
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) { s23[q] += nsl[q]; yt[SL[q]] = pvder::fma2(h2, s23[q], y[SL[q]]); }
                etd_eval(P, K, Ec, S, yt, t2, L.qref, L.vdcref, nf, nsl);             // N4
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    matvec(load_mat(Mq, EM_A54), nf[2 * p], nf[2 * p + 1], u5[2 * p], u5[2 * p + 1]);   // U5
                    matvec(load_mat(Mq, EM_B4), nf[2 * p], nf[2 * p + 1], zn[2 * p], zn[2 * p + 1]);
                    yt[3 * p] = u5[2 * p]; yt[3 * p + 2] = u5[2 * p + 1];
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    k14[q] = ns1[q] + nsl[q];
                    yt[SL[q]] = pvder::fma2(h8, s23[q], pvder::fma2(h4, ns1[q], y[SL[q]]));
                }
                etd_eval(P, K, Ec, S, yt, t1, L.qref, L.vdcref, nf, nsl);             // N5
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    matvec(load_mat(Mq, EM_B5), nf[2 * p], nf[2 * p + 1], zn[2 * p], zn[2 * p + 1]);
                    yt[3 * p] = zn[2 * p]; yt[3 * p + 2] = zn[2 * p + 1];
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)                                            // x_p candidate
                    yt[SL[q]] = pvder::fma2(h23, nsl[q], pvder::fma2(h6, k14[q], y[SL[q]]));
                const float m2b = max_m2(P, yt);
                if (fmaxf(m2a, m2b) > guard && C.off_range == PVDER_OFF_RANGE_TERMINATE) {
                    L.fail = true;                 // the episode ends here (epilogue); the state no longer matters
                } else if (fmaxf(m2a, m2b) > guard) {
                    // duty cycle at / near its limit: the anti-windup may have switched the fast dynamics -> this
                    // step again on the operator sets of the pairs' anti-windup modes (random-action PVDER-v0
                    // episodes never get here; a policy that walks Vdc_ref down 6 % does, and among 65 536 envs
                    // some do after a +2 % grid-voltage event on top of a few per cent)
#ifdef PVDER_RK4_FALLBACK
                    if (A.rk4_fallback) {
                        const int rk4_sub = set == 0 ? Ec.rk4_sub[0] : (set == 1 ? Ec.rk4_sub[1] : Ec.rk4_sub[2]);
                        rk4_pk_steps(P, K, S, y, comp, rk4_sub, h / (float)rk4_sub, t0, tq / (float)rk4_sub, L.qref, L.vdcref);
                    } else
#endif
                    {
                        const lds_v4 *Ms = (const lds_v4 *)Msw + (kSwModes * kMatCount * 4) * set;
                        asm volatile("" : "+v"(Ms));
                        etd_switched_step(P, K, Ec, S, y, comp, Ms, h, t0, tq, L.qref, L.vdcref);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) { y[3 * p] = yt[3 * p]; y[3 * p + 2] = yt[3 * p + 2]; }
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const int g = SL[q];
                        const v2 inc = pvder::fma2(h23, nsl[q], pvder::fma2(h6, k14[q], -comp[g]));
                        const v2 tsum = y[g] + inc;
                        comp[g] = (tsum - y[g]) - inc;
                        y[g] = tsum;
                    }
                }
#endif
                t0 = t2;
            }
            }
            PVDER_TRACE_COLD_MAX(15 + (s < 3 ? s : 3), true);
        }
#if defined(PVDER_TRACE) && defined(PVDER_TRACE_COLD)
        if (blockIdx.x < 4096) {       // the loop's length in shader cycles (slot 30) and in wall-clock ticks (slot 31): the clock it ran at
            // (slot 5: cycles << 8 | lanes of THIS wave in the loop -- the wave with the longest loop and how many lanes it carried)
            atomicMax(&g_trace[blockIdx.x * 32 + 5], ((unsigned long long)(clock64() - trc_c0) << 8) | (unsigned long long)__popcll(__ballot(1)));
            atomicMax(&g_trace[blockIdx.x * 32 + 30], (unsigned long long)(clock64() - trc_c0));
            atomicMax(&g_trace[blockIdx.x * 32 + 31], (unsigned long long)(wall_clock64() - trc_w0));
        }
#endif
        from_pairs(y, ys);
        L.ksub += C.n_sub;
        L.steps += 1;
#ifndef PVDER_RK4_FALLBACK
        if (A.sat_hybrid && m2_peak > kHybTouch2 * P.mlim2) {
            // the duty cycle came within 0.4 % of its limit DURING the step: the step is the hybrid kernel's (nothing stored here)
            istate[(size_t)PVDER_I_STATUS * N + L.e] = kDeferred;
            L.run = false;
            L.live = false;
        } else if (A.hyb_flag && !A.sat_hybrid && m2_peak > Ec.sat2_flag) {
            *(volatile int *)A.hyb_flag = 1;       // (it came near during the step: the kernel follows from the next launch on)
        }
#endif
    }
    epilogue<float, 3>(P, C, R, L, ys, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh, false, 0.f, 0.f,
                       LP ? lane_par : nullptr);
}

// Single phase: the envs the fast path hands over (estimate guard, duty cycle at its limit -- the single-phase inverter
// never gets there in PVDER-v0: |m| <= 0.73 along all four fixed-action episodes) are redone by classic RK4 micro-steps
// right here, by the workgroup that holds them, like the three-phase kernels run their 23-state path: no second
// launch (until late in round 3 every single-phase step paid for one, 4 us of 13, to pick up envs that are never there).
__device__ __forceinline__ void rk4_tail_1ph(const EtdArgs &A, bool defer, const uint8_t *act = nullptr, long long out_off = 0) {
    rk4_body<float, 1>(A.P, A.C, A.ph1_micro, A.R, A.fstate, A.istate, A.ev_vg, A.ev_si, act ? act : A.action, A.out + out_off,
                       A.stats, A.n_env, 0, 0, true, defer);
}

// The follow-up launch of the step kernels (etdlin, ETDRK4; three-phase): advances the envs their 23-state tail marked kDeferred --
// envs at / near the duty-cycle limit -- by ONE env step of the hybrid scheme (pvder_sat_hybrid.hpp::env_step: fp64 between the fp32
// rows), then the same epilogue as every step kernel (reward, done, observation, same-step auto-reset, statistics added to the
// step's).  A 64-env block per workgroup (kHybSub active lanes in each of its waves), ONE env per lane, the scheme's twelve 23-vectors + the state in the lane's own 2.4 KB of LDS
// (153 KB of the CU's 160: as private arrays they make a 6 KB scratch frame per lane, which the runtime serves by allocating scratch
// per DISPATCH -- 0.2 ms per launch, measured -- and every access a memory round trip).  Its workgroups take the blocks hybrid_scan_kernel listed.
constexpr int kHybBlock = 64;                                                     // envs per listed block
#ifndef PVDER_HYB_SUB
#define PVDER_HYB_SUB 16
#endif
// A listed block is taken in QUARTERS: kHybSub = 16 envs, in the first lanes of ONE wave that is a workgroup of its own, with 16 arenas
// (39 KB of LDS: four such workgroups per CU, one per SIMD -- the kernel holds 416 registers per lane).  Round 6, second session, in
// three steps (profiles/r06_hybrid_sub_wave.txt): one 64-lane wave per block kept one SIMD of four busy; four 16-lane waves in ONE
// workgroup used the four SIMDs but ended together -- and a batch at the limit waits for its slowest ENV (the cost of an env step is
// heavy-tailed: tools/hybrid_cost_model.py), so a block cost what the slowest of its 64 envs did; as workgroups of their own the
// quarters end by themselves and the hardware hands the SIMD its next quarter.  (8 / 4 envs per wave -- two / four waves per SIMD at
// 256 / 128 registers, 1.6 / 2.2 KB of spills per lane -- were measured in the one-workgroup form and bought nothing there.)
constexpr int kHybSub = PVDER_HYB_SUB;
constexpr int kHybQuarters = kHybBlock / kHybSub;                                 // items per listed block
constexpr int kHybThreads = 64;
static_assert(kHybSub * kHybQuarters == kHybBlock && kHybSub <= 64, "a quarter is one wave");
constexpr int kHybWavesPerSimd = kHybQuarters > 4 ? kHybQuarters / 4 : 1;
// workgroups of the kernel (they loop over the items of the list).  The hardware hands a SIMD its next workgroup when one ends: with
// one item per workgroup a batch of 65 536 saturated envs (items of very unequal length) is dealt greedily (measured in the one-workgroup-
// per-block form: 279-394 ms per batched step with 512 workgroups of two blocks each, 240-350 with 1024, 253-344 with 2048; a followed
// launch that finds nothing: 15.7-15.9 / 15.9-16.4 / 16.0 us)
#ifndef PVDER_HYB_GRID
#define PVDER_HYB_GRID 4096
#endif
constexpr int kHybGrid = PVDER_HYB_GRID;
constexpr int kHybLaneDoubles = pvder::hyb::kArenaDoubles + 23;                   // odd: lanes on distinct LDS banks at the same index
constexpr size_t kHybLdsBytes = (size_t)kHybSub * (kHybLaneDoubles * sizeof(double) + 2 * kHybMaxSub * sizeof(float));
// WHICH 64-env blocks hold an env marked kDeferred: one wave per block, no LDS -- whole-chip occupancy.  (The step kernel below
// reserves 153 KB of LDS per workgroup: launched over every block it ran ONE workgroup per CU at a time, and a batch of 1 048 576 envs
// paid 60 us per step for 16 384 workgroups that found nothing.)  list[parity] = number of blocks found, list[16 + i] = their indices;
// the OTHER parity's counter is cleared here for the next launch (nobody reads or writes it during this one).
// (n_blk: the list's capacity.  A HIP graph that captured these launches replays them with ONE parity: its counter is then cleared by a
//  memset node in front of the scan -- launch_plan -- and, whatever a caller does, no index beyond the capacity is written or read.)
__global__ __launch_bounds__(kBlock) void hybrid_scan_kernel(const int32_t *__restrict__ istate, int n_env, int *__restrict__ list, int parity, int n_blk) {
    const int e = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (e == 0) list[1 - parity] = 0;
    const bool mine = e < n_env && istate[(size_t)PVDER_I_STATUS * (size_t)n_env + (size_t)e] == kDeferred;
    if (__any((int)mine) && (threadIdx.x & 63) == 0) {
        const int i = atomicAdd(&list[parity], 1);
        if (i < n_blk) list[16 + i] = e >> 6;
    }
}
__global__ __launch_bounds__(kHybThreads) __attribute__((amdgpu_waves_per_eu(kHybWavesPerSimd, kHybWavesPerSimd))) void step_sat_hybrid_kernel(const EtdArgs A, const float *__restrict__ lane_par,
                                                                                                              const int *__restrict__ list, int parity, int n_blk) {
    using namespace pvder::hyb;
    extern __shared__ __attribute__((aligned(16))) double hyb_lds[];
    const int n_env = A.n_env;
    const size_t N = (size_t)n_env;
    const int n_found = list[parity] < n_blk ? list[parity] : n_blk;
    for (int it = (int)blockIdx.x; it < n_found * kHybQuarters; it += (int)gridDim.x) {
    const int blk = list[16 + it / kHybQuarters];
    const int ln = (int)threadIdx.x, al = ln < kHybSub ? ln : 0;                 // (al: the lane's arena)
    const int slot = (it % kHybQuarters) * kHybSub + al;                         // the env's place in its block
    const int e_mine = ln < kHybSub ? blk * kHybBlock + slot : n_env;
    Lane<float> L;
    float ys[23];
    prologue<float, 23>(A.C, A.fstate, A.istate, A.action, n_env, L, ys, kDeferred, e_mine < n_env ? e_mine : n_env - 1);
    L.live = L.live && e_mine < n_env;
    L.run = L.run && e_mine < n_env;
    ModelK<float> P = A.P;
    if (lane_par) {
        const float *__restrict__ lp = lane_par + L.ec;
        P.iC = lp[(size_t)PVDER_LPF_IC * N];
        P.KpDC = lp[(size_t)PVDER_LPF_KP_DC * N]; P.KiDC = lp[(size_t)PVDER_LPF_KI_DC * N];
        P.KpQ = lp[(size_t)PVDER_LPF_KP_Q * N];   P.KiQ = lp[(size_t)PVDER_LPF_KI_Q * N];
        P.pvA = lp[(size_t)PVDER_LPF_PVA * N];    P.pvB = lp[(size_t)PVDER_LPF_PVB * N];
    }
    if (L.run) {
        const StepK<float> &C = A.C;
        HybK H;
        H.Rf = P.Rf; H.iLf = P.iLf; H.iC = P.iC; H.kviR = P.kviR; H.kviI = P.kviI; H.GR = P.GR; H.GI = P.GI; H.KpG = P.KpG; H.KiG = P.KiG;
        H.KpDC = P.KpDC; H.KiDC = P.KiDC; H.KpQ = P.KpQ; H.KiQ = P.KiQ; H.wp = P.wp; H.KpP = P.KpP; H.KiP = P.KiP; H.mlim2 = P.mlim2;
        H.ilim2 = P.ilim2; H.iwb = P.iwb; H.pvA = P.pvA; H.pvB = P.pvB; H.pvK = P.pvK; H.vdc_ref0 = P.vdc_ref0;
        double *arena = hyb_lds + (size_t)al * kHybLaneDoubles;
        double *y = arena + kArenaDoubles;
        float *vg = (float *)(hyb_lds + (size_t)kHybSub * kHybLaneDoubles) + (size_t)al * (2 * kHybMaxSub), *si = vg + kHybMaxSub;
        for (int k = 0; k < 23; ++k) y[k] = (double)ys[k];
        const int n_sub = C.n_sub < kHybMaxSub ? C.n_sub : kHybMaxSub;      // (plans with longer env steps never set sat_hybrid)
        for (int s = 0; s < n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            vg[s] = A.ev_vg[(size_t)row * N + L.ec];
            si[s] = A.ev_si[(size_t)row * N + L.ec];
        }
        env_step(H, y, (double)L.qref, (double)L.vdcref, vg, si, n_sub, C.graded ? C.micro : 1, (Stats *)nullptr, arena);
        for (int k = 0; k < 23; ++k) ys[k] = (float)y[k];
        L.ksub += C.n_sub;
        L.steps += 1;
    }
    // (block blk IS wave e / 64 of the step kernels' numbering: its statistics slot -- which the block's quarters add to as they finish)
    epilogue<float, 3>(P, A.C, A.R, L, ys, A.fstate, A.istate, A.ev_vg, A.ev_si, A.out, A.stats ? A.stats + 4 * (size_t)blk : nullptr,
                       n_env, 0, false, 0.f, 0.f, lane_par, true, -1, -1, -1, true);
    }
}

// =============================================================================================
// ETDRK4, balanced fast path.  PVDER-v0 feeds the 'ThreePhaseUnbalanced' model a balanced grid
// (unbalance ratios 1.0) from a balanced steady state, and the ODE maps balanced states to balanced
// states as long as no anti-windup sign test fires: phases b and c are phase a rotated by -/+120
// degrees (this is pvder's own 'SolarPVDERThreePhaseBalanced' reduction).  This kernel checks that
// at entry, integrates phase a + the 5 shared states (11 of 23) with the same ETDRK4 scheme, and
// writes b and c back by rotation.  An env that is NOT balanced on entry, or whose duty cycle
// reaches the saturation guard at any step, is left untouched and marked kDeferred: the full
// 23-state path (etdrk4_full_tail, at the end of this kernel) advances it.
// =============================================================================================
template <int PH>
__device__ __forceinline__ float bal_eval(const ModelK<float> &P, const pvder::PkConst &K, const EtdK &Ec,
                                          const pvder::SubPk &S, v2 ia, v2 xa, v2 ua, v2 pa, v2 pb, float delta,
                                          float qref, float vdcref, v2 kpo, v2 kio, v2 &ni, v2 &nu,
                                          v2 *__restrict__ ns, float &ir2) {
    using namespace pvder;
    // slow pairs: pa = (xDC, xQ) -- the two outer PI integrators, pb = (Vdc, xPLL)
    const float Vdc = pb.x;
    const v2 ma = fma2(bc2(P.KpG), ua, xa);
    // dot products as mul + fma (written as sums of a packed product hipcc pairs them up through
    // three v_mov); Q = Im S_PCC = PH (Asw/2 + qk i) . i
    const float Pa = __builtin_fmaf(ma.y, ia.y, ma.x * ia.x);                // Re(m conj i) of one of PH identical phases
    const v2 tq = fma2(bc2((float)PH * K.qk), ia, bc2(0.5f * PH) * S.Asw[0]);
    const float Q = __builtin_fmaf(tq.y, ia.y, tq.x * ia.x);
    // packed outer loops: e = (Vdc_ref - Vdc, Q_ref - Q);  iref_a = (xDC, xQ) + (Kp_DC, -Kp_Q) e
    const v2 e = mk2(vdcref - Vdc, qref - Q);
    const v2 iar = fma2(kpo, e, pa);
    const v2 dpi = kio * e;                                              // (dxDC, dxQ)
    // no current-limit anti-windup here: the caller hands an env whose |iref| comes within 10 % of
    // the limit over to the 23-state path (ir2 = |iref|^2)
    ir2 = __builtin_fmaf(iar.y, iar.y, iar.x * iar.x);
    float s1, c1;
    Math<float>::sincos_turns(delta * (float)(1.0 / kTwoPi), s1, c1);
    v2 V1 = fma2(bc2(P.kviR), ia, S.A0); V1 = fma2(mk2(-P.kviI, P.kviI), swp(ia), V1);   // v_a (V2 = 0)
    const float vd = __builtin_fmaf(V1.y, s1, V1.x * c1);
    const float dw = P.KpP * vd + pb.y;
    const float dwr = dw * P.iwb;
    const float cmf = K.hiLf * Vdc;
    v2 n = fma2(bc2(cmf), xa, S.D[0]);
    n = fma2(bc2(cmf * P.KpG - Ec.c0kp), ua, n);
    ni = fma2(mk2(dwr, -dwr), swp(ia), n);
    nu = bc2(P.wp) * iar;
    ns[0] = bc2(P.KiG) * ua;
    // dVdc = (Ppv - Re S)/(Vdc C) with Ppv = max(0, Vdc g(Vdc)), Re S = Vdc PH Pa/4  (Vdc > 0):
    //      = (max(0, g) - PH Pa/4)/C         -- no division
    float g = __builtin_fmaf(-P.pvB, Math<float>::exp_pv(P.pvK * Vdc), S.pv1);
    if (P.pvK == 0.f) g = -S.pv1 * pvder::Math<float>::rcp(Vdc);       // (model option U5, wave-uniform: see ModelK)
    const float dVdc = __builtin_fmaf(-0.25f * PH * P.iC, Pa, P.iC * fmaxf(g, 0.f));
    ns[1] = dpi;
    ns[2] = mk2(dVdc, P.KiP * vd);
    ns[3] = mk2(dw, 0.f);
    return __builtin_fmaf(ma.y, ma.y, ma.x * ma.x);
}

// |z_b - z_a e^{-j120}|^2 + |z_c - z_a e^{+j120}|^2
__device__ __forceinline__ float unbalance2(v2 za, v2 zb, v2 zc) {
    using namespace pvder;
    const v2 hz = bc2(-0.5f) * za, jz = swp(za) * mk2(-kHalfSqrt3, kHalfSqrt3);
    const v2 db = zb - (hz - jz), dc = zc - (hz + jz);
    const v2 s = db * db + dc * dc;
    return s.x + s.y;
}

template <int PH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void step_etdrk4_bal_kernel(const EtdArgs A) {
    hyb_progress<0>();
    const ModelK<float> &P = A.P;
    const StepK<float> &C = A.C;
    const ResetK &R = A.R;
    const EtdK &Ec = A.Ec;
    const float *__restrict__ coef = A.coef;
    float *__restrict__ fstate = A.fstate;
    int32_t *__restrict__ istate = A.istate;
    float *__restrict__ ev_vg = A.ev_vg;
    float *__restrict__ ev_si = A.ev_si;
    const uint8_t *__restrict__ action = A.action;
    float *__restrict__ out = A.out;
    float *__restrict__ stats = A.stats;
    const int stats_fresh = A.stats_fresh;
    const int n_env = A.n_env;
    constexpr int NS = 6 * PH + 5, B = 6 * PH;      // PH = 1: the single-phase model is this reduction itself
    const size_t N = (size_t)n_env;
    PVDER_TRACE_AT(0);
    __shared__ __attribute__((aligned(16))) float Mlds[kCoefFloats];
    for (int k = threadIdx.x; k < kCoefFloats; k += kBlock) Mlds[k] = coef[k];
    __syncthreads();
    PVDER_TRACE_AT(1);
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(C, fstate, istate, action, n_env, L, ys);
    bool defer = PH == 3 && A.force_full != 0;       // (A/B and tests: everything through the 23-state tail)
    if (L.run) {
        v2 ia = pvder::mk2(ys[0], ys[1]), xa = pvder::mk2(ys[2], ys[3]), ua = pvder::mk2(ys[4], ys[5]);
        v2 pa = pvder::mk2(ys[B + 1], ys[B + 2]), pb = pvder::mk2(ys[B], ys[B + 3]), dl = pvder::mk2(ys[B + 4], 0.f);   // (xDC,xQ), (Vdc,xPLL)
        if constexpr (PH == 3) {   // balance check (tolerances: 2e-5 relative on i and x, 2e-9 absolute on u ~ 1e-6)
            const float ei = unbalance2(ia, pvder::mk2(ys[6], ys[7]), pvder::mk2(ys[12], ys[13]));
            const float ex = unbalance2(xa, pvder::mk2(ys[8], ys[9]), pvder::mk2(ys[14], ys[15]));
            const float eu = unbalance2(ua, pvder::mk2(ys[10], ys[11]), pvder::mk2(ys[16], ys[17]));
            const v2 i2 = ia * ia, x2 = xa * xa;
            defer |= ei > 4e-10f * (i2.x + i2.y) + 1e-20f || ex > 4e-10f * (x2.x + x2.y) + 1e-20f || eu > 4e-18f;
        }
        v2 cx = pvder::bc2(0.f), cv = pvder::bc2(0.f), cq = pvder::bc2(0.f), cd = pvder::bc2(0.f);   // Kahan
        const pvder::PkConst K = pvder::make_pkconst(P);
        // outer-loop gains (Kp_DC, -Kp_Q), (Ki_DC, -Ki_Q), pinned to SGPRs.  Left to itself hipcc
        // merges the adjacent kernarg fields into one 12-byte memcpy to a private slot, promotes that
        // slot to LDS and -- to index it by flat work-item id -- makes every wave read the AQL
        // dispatch packet (host memory, over PCIe) at kernel entry: workgroup starts then trickle in
        // at ~100 ns each, 25 us for 256 workgroups (tools/trace_timeline.py, tools/dispatch_probe.hip).
        float g_kpdc = P.KpDC, g_kpq = -P.KpQ, g_kidc = P.KiDC, g_kiq = -P.KiQ;
        asm volatile("" : "+s"(g_kpdc), "+s"(g_kpq), "+s"(g_kidc), "+s"(g_kiq));
        const v2 kpo = pvder::mk2(g_kpdc, g_kpq), kio = pvder::mk2(g_kidc, g_kiq);
        pvder::SubPk S;
        int prev_row = L.ksub > 0 ? event_row(L.ksub - 1, C.ev_s0, C.ev_ds, C.ev_count) : 0;
#ifdef PVDER_TRACE
        if (threadIdx.x == 0 && blockIdx.x < 4096)      // forces the state loads to have landed
            g_trace[blockIdx.x * 32 + 2] = wall_clock64() + (pb.x > 1e30f ? 1 : 0);
#endif
        // `defer` only accumulates: a deferred lane keeps computing (its results are discarded -- the
        // 23-state path restarts it from the state in memory), so the loops carry no per-lane exit and
        // the state update needs no mask; a wave whose lanes are ALL deferred leaves
        float i2m = Ec.i2_to_m2;
        asm volatile("" : "+v"(i2m));          // keep it in a VGPR (it would be re-read from the kernarg segment every step)
        // event row of sim time step ksub + s, tracked incrementally (event_row() is an integer division:
        // ~40 issue slots of a lone wave, per sim time step): `until` = sim time steps to the next row change
        int row = event_row(L.ksub, C.ev_s0, C.ev_ds, C.ev_count);
        int until = 0x7fffffff;
        if (C.ev_count > 0) {
            const int d = L.ksub - C.ev_s0;
            until = d < 0 ? -d : C.ev_ds - (d - (d / C.ev_ds) * C.ev_ds);
        }
        for (int s = 0; s < C.n_sub && !__builtin_amdgcn_readfirstlane(__all((int)defer)); ++s) {
            if (s < 16) PVDER_TRACE_AT(3 + s);
            const bool input_jump = row != prev_row;
            prev_row = row;
            // two-rate schedule, per lane (see etdrk4_full_tail): fine pass for the lanes with a jump here, coarse
            // pass for the others; envs in step (PVDER-v0 rollouts) execute one of the two
            const bool lane_fine = (s == 0) || input_jump;
            pvder::make_sub_pk(P, ev_vg[(size_t)row * N + L.ec], ev_si[(size_t)row * N + L.ec], S);
            for (int pass = 0; pass < 2; ++pass) {
            const bool fine = pass == 0;
            const bool mine = lane_fine == fine;          // (every lane computes; the lanes of this pass commit: no exec games)
            if (!__builtin_amdgcn_readfirstlane(__any((int)mine))) continue;
            const int nst = fine ? (C.graded ? 2 : C.micro) : C.micro2;
            for (int j = 0; j < nst; ++j) {
                // graded schedule: first step 1/(60 micro), then ONE step over the remainder
                const int set = fine ? ((C.graded && j > 0) ? 2 : 0) : 1;
                const float h = set == 0 ? Ec.hset[0] : (set == 1 ? Ec.hset[1] : Ec.hset[2]);
                const v2 h2 = pvder::bc2(0.5f * h), h4 = pvder::bc2(0.25f * h), h8 = pvder::bc2(0.125f * h),
                         h6 = pvder::bc2(h * (float)(1.0 / 6.0)), h23 = pvder::bc2(h * (float)(2.0 / 3.0));
                const float guard = (j == 0 && input_jump) ? Ec.sat2_event : Ec.sat2_smooth;
                const lds_v4 *Mq = (const lds_v4 *)Mlds + 2 * kMatCount * set;
                asm volatile("" : "+v"(Mq));
                v2 n1i, n1u, ni, nu, ns1[4], ns[4], s23[4];
                float ir2a, ir2x;
                // software prefetch: issue the LDS reads of the seven matrices applied to (z, N1) now,
                // the model evaluation below hides their latency (one wave per SIMD: nobody else will)
                const Mat4 mE2 = load_mat(Mq, EM_E2), mE = load_mat(Mq, EM_E), mA21 = load_mat(Mq, EM_A21);
                const Mat4 mA31 = load_mat(Mq, EM_A31), mA41 = load_mat(Mq, EM_A41), mA51 = load_mat(Mq, EM_A51),
                           mB1 = load_mat(Mq, EM_B1);
                __builtin_amdgcn_sched_barrier(0);     // keep the reads up here (hipcc would sink them to their uses)
                const float m2a = bal_eval<PH>(P, K, Ec, S, ia, xa, ua, pa, pb, dl.x, L.qref, L.vdcref, kpo, kio, n1i, n1u, ns1, ir2a);
                v2 e2i = pvder::bc2(0.f), e2u = pvder::bc2(0.f), ei = pvder::bc2(0.f), eu = pvder::bc2(0.f);
                matvec(mE2, ia, ua, e2i, e2u);
                matvec(mE, ia, ua, ei, eu);
                v2 ti = e2i, tu = e2u;                                          // U2
                matvec(mA21, n1i, n1u, ti, tu);
                v2 u3i = e2i, u3u = e2u, u4i = ei, u4u = eu, u5i = e2i, u5u = e2u, zi = ei, zu = eu;
                matvec(mA31, n1i, n1u, u3i, u3u);
                matvec(mA41, n1i, n1u, u4i, u4u);
                matvec(mA51, n1i, n1u, u5i, u5u);
                matvec(mB1, n1i, n1u, zi, zu);
                const Mat4 mA32 = load_mat(Mq, EM_A32), mA42 = load_mat(Mq, EM_A42), mA52 = load_mat(Mq, EM_A52),
                           mA54 = load_mat(Mq, EM_A54), mB4 = load_mat(Mq, EM_B4), mB5 = load_mat(Mq, EM_B5);
                __builtin_amdgcn_sched_barrier(0);                              // prefetched behind evaluations 2-5
                bal_eval<PH>(P, K, Ec, S, ti, pvder::fma2(h2, ns1[0], xa), tu, pvder::fma2(h2, ns1[1], pa),
                         pvder::fma2(h2, ns1[2], pb), dl.x + 0.5f * h * ns1[3].x, L.qref, L.vdcref, kpo, kio, ni, nu, ns, ir2x);   // N2
                v2 n23i = ni, n23u = nu;
                matvec(mA32, ni, nu, u3i, u3u);                                 // U3
#pragma unroll
                for (int q = 0; q < 4; ++q) s23[q] = ns[q];
                bal_eval<PH>(P, K, Ec, S, u3i, pvder::fma2(h2, ns[0], xa), u3u, pvder::fma2(h2, ns[1], pa),
                         pvder::fma2(h2, ns[2], pb), dl.x + 0.5f * h * ns[3].x, L.qref, L.vdcref, kpo, kio, ni, nu, ns, ir2x);   // N3
                n23i += ni; n23u += nu;
                matvec(mA42, n23i, n23u, u4i, u4u);                             // U4
                matvec(mA52, n23i, n23u, u5i, u5u);
#pragma unroll
                for (int q = 0; q < 4; ++q) s23[q] += ns[q];
                bal_eval<PH>(P, K, Ec, S, u4i, pvder::fma2(h2, s23[0], xa), u4u, pvder::fma2(h2, s23[1], pa),
                         pvder::fma2(h2, s23[2], pb), dl.x + 0.5f * h * s23[3].x, L.qref, L.vdcref, kpo, kio, ni, nu, ns, ir2x);   // N4
                matvec(mA54, ni, nu, u5i, u5u);                                 // U5
                matvec(mB4, ni, nu, zi, zu);
                v2 k14[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) k14[q] = ns1[q] + ns[q];
                bal_eval<PH>(P, K, Ec, S, u5i, pvder::fma2(h8, s23[0], pvder::fma2(h4, ns1[0], xa)), u5u,
                         pvder::fma2(h8, s23[1], pvder::fma2(h4, ns1[1], pa)), pvder::fma2(h8, s23[2], pvder::fma2(h4, ns1[2], pb)),
                         dl.x + h * (0.25f * ns1[3].x + 0.125f * s23[3].x), L.qref, L.vdcref, kpo, kio, ni, nu, ns, ir2x);         // N5
                matvec(mB5, ni, nu, zi, zu);
                // Kahan-compensated slow update:  y+ = y + h (N1 + N4)/6 + 2h N5/3
                v2 inc, tsum;
                v2 ncx, ncv, ncq, ncd;
                inc = pvder::fma2(h23, ns[0], pvder::fma2(h6, k14[0], -cx)); tsum = xa + inc; ncx = (tsum - xa) - inc; const v2 xn = tsum;
                inc = pvder::fma2(h23, ns[1], pvder::fma2(h6, k14[1], -cv)); tsum = pa + inc; ncv = (tsum - pa) - inc; const v2 vn = tsum;
                inc = pvder::fma2(h23, ns[2], pvder::fma2(h6, k14[2], -cq)); tsum = pb + inc; ncq = (tsum - pb) - inc; const v2 qn = tsum;
                inc = pvder::fma2(h23, ns[3], pvder::fma2(h6, k14[3], -cd)); tsum = dl + inc; ncd = (tsum - dl) - inc; const v2 dn = tsum;
                const v2 mn = pvder::fma2(pvder::bc2(P.KpG), zu, xn);
                // near duty-cycle saturation or the current limit: the 23-state path redoes this env step
                defer |= mine && fmaxf(fmaxf(m2a, __builtin_fmaf(mn.y, mn.y, mn.x * mn.x)), ir2a * i2m) > guard;
                ia = mine ? zi : ia; ua = mine ? zu : ua; xa = mine ? xn : xa; pa = mine ? vn : pa; pb = mine ? qn : pb; dl = mine ? dn : dl;
                cx = mine ? ncx : cx; cv = mine ? ncv : cv; cq = mine ? ncq : cq; cd = mine ? ncd : cd;
            }
            }
            {   // branch-free (a per-lane branch here costs an exec-mask save in an SGPR-starved kernel)
                const bool chg = --until == 0;
                row += (chg && row < C.ev_count) ? 1 : 0;
                until = chg ? C.ev_ds : until;
            }
        }
        if (!defer) {
            // phases b, c = phase a rotated by -/+ 120 degrees
            const v2 jh3 = pvder::mk2(-pvder::kHalfSqrt3, pvder::kHalfSqrt3);
            const v2 z[3] = {ia, xa, ua};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ys[2 * k] = z[k].x; ys[2 * k + 1] = z[k].y;
                if constexpr (PH == 3) {
                    const v2 hz = pvder::bc2(-0.5f) * z[k], jz = pvder::swp(z[k]) * jh3;
                    const v2 zb = hz - jz, zc = hz + jz;
                    ys[6 + 2 * k] = zb.x; ys[6 + 2 * k + 1] = zb.y;
                    ys[12 + 2 * k] = zc.x; ys[12 + 2 * k + 1] = zc.y;
                }
            }
            ys[B] = pb.x; ys[B + 1] = pa.x; ys[B + 2] = pa.y; ys[B + 3] = pb.y; ys[B + 4] = dl.x;
            L.ksub += C.n_sub;
            L.steps += 1;
        }
    }
    PVDER_TRACE_AT(20);
    if (PH == 1 && defer && C.off_range == PVDER_OFF_RANGE_TERMINATE) {
        L.fail = L.run;     // single phase: the hand-over would be classic RK4 for the whole env step -- the episode ends instead
    } else if (defer) {
        // hand the env over untouched: nothing stored, nothing added to the stats
        L.run = false;
        L.live = false;
    }
    epilogue<float, PH>(P, C, R, L, ys, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh);
    PVDER_TRACE_AT(21);
    // the deferred envs of this workgroup (none in PVDER-v0 rollouts) take the 23-state path (three-phase) / classic RK4
    // micro-steps (single phase) right here -- no second launch
    defer = defer && !L.fail;
    if (__syncthreads_or((int)defer)) {
        KernargPtr ka = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        if constexpr (PH == 3) etdrk4_full_tail(*(const EtdArgs *)ka, defer);
        else rk4_tail_1ph(*(const EtdArgs *)ka, defer);
    }
}

// Per-LANE parameter override ("optional per-env override [P][N] for domain randomisation"; config_der.json:2-21):
// every env of the batch on the 23-state ETDRK4 path with its own DC-link capacitance, outer-loop gains, PV array size and
// insolation scale -- the parameters that stay OUTSIDE the exponential operator (the stiff (i_p, u_p) pairs: Rf, Lf, Kp_GCC,
// wp, and the operator sets of the anti-windup modes, belong to the parameter GROUP: param_groups.PVDERVecGroups).  No
// balanced reduction here: an env with its own PV array is not at the group's operating point, and the reduction's guards
// are folded for that one.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1))) void step_etdrk4_lane_kernel(const EtdArgs A, const float *lane_par) {
    hyb_progress<0>();
    etdrk4_full_tail<true>(A, true, nullptr, 0, lane_par, A.stats_fresh);
}

// ... and the same transition with an env's three phases on three lanes of a quad (etd_switched_step3): grid = 4 n_env / 256.
// 16 envs per wave, four times the waves, 0.6 x the instructions per wave and step.  Measured (tools/lane_params_rate.py, us
// per batched step, this kernel / a lane per env): 16 384 envs 45 / 71, 32 768 envs 65 / 71, 49 152 envs 89 / 72, 65 536 envs
// 115 / 72 -- the operand reads of the stiff pairs (LDS) and the issue slots are shared by the SIMD's waves, so the form that
// wins a lone wave's latency loses the batch's throughput: launch_plan picks by batch size.
constexpr int kLane3MaxEnvs = 32768;
__global__ __launch_bounds__(kBlock) void step_etdrk4_lane3_kernel(const EtdArgs A, const float *lane_par) {
    hyb_progress<0>();
    __shared__ __attribute__((aligned(16))) float Msw[kSwFloats];          // (as etdrk4_full_tail stages them)
    __shared__ __attribute__((aligned(16))) float wstats[4 * (kBlock / 64)];
    for (int k = threadIdx.x; k < kSwFloats; k += kBlock) {
        const int c = k & 3, j = (k >> 2) & 3, mat = (k >> 4) % kMatCount, sm = (k >> 4) / kMatCount, mode = sm % kSwModes, set = sm / kSwModes;
        Msw[((((set * kMatCount + mat) * 4 + j) * kSwModes + mode) << 2) + c] = A.coef[kCoefFloats + k];
    }
    __syncthreads();
    // the wave's statistics go to LDS first -- the caller's slots are per 64 envs (include/pvder_hip.h): a workgroup here
    float *wst = A.stats ? (float *)wstats : nullptr;
    etd_tail3<true>(A, Msw, (int)((blockIdx.x * (unsigned)kBlock + threadIdx.x) >> 2), true, lane_par, wst, 1, A.action, A.out, true);
    if (A.stats) {
        __syncthreads();
        if (threadIdx.x == 0) {
            v4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) t += ((const v4 *)wstats)[w];
            v4 *slot = (v4 *)A.stats + blockIdx.x;
            if (!A.stats_fresh) t += *slot;
            *slot = t;
        }
    }
}

}  // namespace
