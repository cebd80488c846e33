// pvder_kernels.hip -- hand-written gfx950 kernels + C ABI (include/pvder_hip.h) of the batched
// PVDER-v0 hot path.  One env per lane; structure-of-arrays state in HBM (env index fastest, so
// every global access of a wave is one fully coalesced 256-byte row segment); model constants in
// SGPRs (kernel argument); the whole env transition
//     action_calc -> integrate n_sub sim time steps -> reward_calc -> done -> state [-> auto-reset]
// (PVDER_env.py:138-196, :198-229, :231-301, :531-542, :316-334) is ONE launch with the states held
// in registers for the entire step; wave-level (DPP) reduction of the reward-sum / done-count
// statistics into one 16-byte store per wave, no atomics, no barrier (env_utilities.py:32-38 aggregate).
//
// Integrators (fixed per launch, selected by the entry point / plan):
//   step_etdlin_kernel<PH>     fp32 ETD3RK (Cox-Matthews) on dy/dt = L y + N(y), L = the full 11x11
//                              Jacobian at the nominal operating point: a tick + two steps per env step,
//                              the dense 11x11 matrix-vector products on the matrix cores (fp32 MFMA
//                              4x4x1, coefficient operands packed sixteen to a register); balanced
//                              11-state fast path + 23-state tail in ONE launch (default)
//   step_etdlin_rollout_kernel K env steps of an open-loop action sequence FUSED into one launch, state
//   + ..._finish_kernel        in registers from step to step (pvder_plan_rollout_fused); bit-identical
//   step_etdrk4_bal_kernel<PH> fp32 exponential Runge-Kutta method of stiff order 4 (Hochbruck-Ostermann,
//                              5 stages): only the stiff current loop (i_p,u_p) is in the exponential;
//                              balanced 11-state fast path + 23-state tail (etdrk4_full_tail) in ONE launch
//   step_rk4_pk_kernel   fp32  classic RK4 micro-stepping, packed-VALU three-phase form
//   step_rk4_kernel<T,PH>      classic RK4 micro-stepping, generic (single phase / A-B reference)
//   step_rk45_kernel<T,PH> fp64 adaptive Dormand-Prince 5(4)
//
// The right-hand side itself has no dense contraction (per-env nonlinear ODE: VALU, packed fp32); the
// matrix cores enter through the exponential integrator, whose coefficients are dense matrices shared
// by all envs -- a GEMM with one operand per parameter set and one column per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>
#include <time.h>
#include <chrono>
#include <complex>
#include <new>

#include "pvder_hip.h"
#include "pvder_model.hpp"
#include "pvder_model_pk.hpp"
#include "pvder_etdlin.hpp"
#include "pvder_sat_hybrid.hpp"
#include "pvder_agent.h"
#include "pvder_agent_dev.hpp"

// ---- the device side, one part per integrator (anonymous namespace each: one translation unit) ----
#include "pvder_step_common.hpp"
#include "pvder_step_rk4.hpp"
#include "pvder_step_etdrk4.hpp"
#include "pvder_step_etdlin.hpp"
#include "pvder_step_etdlin_kernels.hpp"
#include "pvder_step_rollout.hpp"
#include "pvder_step_rk45.hpp"
#include "pvder_step_reset.hpp"

namespace {

// =============================================================================================
// host side: parameter folding + launchers
// =============================================================================================
template <typename T>
ModelK<T> fold_params(const double *par, bool exp2_path) {
    ModelK<T> P;
    P.Rf = (T)par[PVDER_PAR_RF];
    P.iLf = (T)(1.0 / par[PVDER_PAR_LF]);
    P.iC = (T)(1.0 / par[PVDER_PAR_C]);
    P.kviR = (T)par[PVDER_PAR_KVI_RE];
    P.kviI = (T)par[PVDER_PAR_KVI_IM];
    P.kvi3R = (T)(par[PVDER_PAR_KVI_RE] / 3.0);
    P.kvi3I = (T)(par[PVDER_PAR_KVI_IM] / 3.0);
    P.GR = (T)(par[PVDER_PAR_KVG_RE] * par[PVDER_PAR_VG_SCALE]);
    P.GI = (T)(par[PVDER_PAR_KVG_IM] * par[PVDER_PAR_VG_SCALE]);
    P.KpG = (T)par[PVDER_PAR_KP_GCC];
    P.KiG = (T)par[PVDER_PAR_KI_GCC];
    P.KpDC = (T)par[PVDER_PAR_KP_DC];
    P.KiDC = (T)par[PVDER_PAR_KI_DC];
    P.KpQ = (T)par[PVDER_PAR_KP_Q];
    P.KiQ = (T)par[PVDER_PAR_KI_Q];
    P.wp = (T)par[PVDER_PAR_WP];
    P.KpP = (T)par[PVDER_PAR_KP_PLL];
    P.KiP = (T)par[PVDER_PAR_KI_PLL];
    P.mlim2 = (T)(par[PVDER_PAR_M_LIMIT] * par[PVDER_PAR_M_LIMIT]);
    P.ilim2 = (T)(par[PVDER_PAR_IREF_LIMIT] * par[PVDER_PAR_IREF_LIMIT]);
    P.iwb = (T)(1.0 / par[PVDER_PAR_WBASE]);
    const double scale = par[PVDER_PAR_VDCBASE] / par[PVDER_PAR_SBASE];
    P.pvA = (T)(par[PVDER_PAR_NP] * par[PVDER_PAR_ISCR] * scale / 100.0);
    P.pvB = (T)(par[PVDER_PAR_NP] * par[PVDER_PAR_IRS] * scale);
    P.pvK = (T)(par[PVDER_PAR_PV_VT] * (exp2_path ? 1.4426950408889634074 : 1.0));
    P.vrms_ref = (T)par[PVDER_PAR_VRMS_REF];
    P.vdc_ref0 = (T)par[PVDER_PAR_VDC_REF0];
    P.sinsol0 = (T)par[PVDER_PAR_SINSOL0];
    if (par[PVDER_PAR_PPV_AT_MPP] != 0.0) {
        // model option U5: the array's power at the MPP voltage -- folded into (pvA, pvB, pvK), see ModelK
        const double v0 = par[PVDER_PAR_VDC_REF0], e0 = exp(par[PVDER_PAR_PV_VT] * v0);
        const double a = par[PVDER_PAR_NP] * par[PVDER_PAR_ISCR] * scale / 100.0, b = par[PVDER_PAR_NP] * par[PVDER_PAR_IRS] * scale;
        P.pvA = (T)(-v0 * a);
        P.pvB = (T)(-v0 * b * (1.0 - e0));
        P.pvK = (T)0;
    }
    return P;
}

template <typename T>
StepK<T> fold_cfg(const pvder_step_cfg *c) {
    StepK<T> C;
    C.n_sub = c->n_sub; C.micro = c->micro; C.max_ksub = c->max_ksub;
    C.micro2 = c->micro_coarse > 0 ? c->micro_coarse : c->micro; C.max_steps = c->max_steps;
    C.graded = (c->graded != 0 && c->micro > 1) ? 1 : 0;
    C.goal = c->goal; C.discrete = c->discrete; C.ev_s0 = c->ev_s0; C.ev_ds = c->ev_ds;
    C.ev_count = c->ev_count;
    C.max_it = c->reserved > 0 ? c->reserved : 20000000;   // RK45: attempts per sim time step
    C.extra = c->extra_rewards;
    C.off_range = c->off_range;
    C.dq = (T)c->del_qref; C.dv = (T)c->del_vdcref; C.inv_tmax = (T)c->inv_max_sim_time;
    C.p_t = (T)c->p_target; C.q_t = (T)c->q_target; C.rtol = (T)c->rtol; C.atol = (T)c->atol;
    return C;
}

int fold_reset(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, ResetK &R) {
    R.enabled = 0; R.keep_events = 1; R.seed = 0; R.env_id_base = 0; R.env_ids = nullptr; R.lazy = 0;
    R.spec = pvder_event_spec{};
    for (int k = 0; k < 23; ++k) R.y0[k] = 0.0;
    if (!ar) return 0;
    if (!ar->y0) return PVDER_EINVAL;
    if (ar->spec.ev_count != cfg->ev_count) return PVDER_EINVAL;
    const int ns = PVDER_NS((int)par[PVDER_PAR_PHASES]);
    R.enabled = ar->enabled; R.keep_events = ar->keep_events; R.seed = ar->seed; R.env_id_base = ar->env_id_base; R.env_ids = ar->env_ids;
    R.lazy = ar->lazy_events != 0 && !ar->keep_events;
    R.spec = ar->spec;
    for (int k = 0; k < ns; ++k) R.y0[k] = ar->y0[k];
    return 0;
}

int check_common(const double *par, const pvder_step_cfg *cfg, int n_env) {
    if (!par || !cfg || n_env <= 0) return PVDER_EINVAL;
    const int ph = (int)par[PVDER_PAR_PHASES];
    if (ph != 1 && ph != 3) return PVDER_EINVAL;
    if (cfg->n_sub < 1 || cfg->micro < 1 || cfg->ev_count < 0) return PVDER_EINVAL;
    if (cfg->ev_count > 0 && cfg->ev_ds < 1) return PVDER_EINVAL;
    if (cfg->goal < 0 || cfg->goal > 2) return PVDER_EINVAL;
    return 0;
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }

// ---- ETDRK4 coefficient matrices: functions of the constant 2x2 complex Lz, by the Lagrange
// (Sylvester) formula  f(L) = f(l1)(L - l2 I)/(l1-l2) + f(l2)(L - l1 I)/(l2-l1)   (fp64, host)
typedef std::complex<double> cd;
struct M2 { cd a, b, c, d; };
M2 fun2(const M2 &L, cd l1, cd l2, cd f1, cd f2) {
    const cd w1 = f1 / (l1 - l2), w2 = f2 / (l2 - l1);
    return M2{w1 * (L.a - l2) + w2 * (L.a - l1), (w1 + w2) * L.b, (w1 + w2) * L.c,
              w1 * (L.d - l2) + w2 * (L.d - l1)};
}
cd phi(int k, cd z) {
    if (k == 0) return std::exp(z);
    if (k == 1) return (std::exp(z) - 1.0) / z;
    if (k == 2) return (std::exp(z) - 1.0 - z) / (z * z);
    return (std::exp(z) - 1.0 - z - 0.5 * z * z) / (z * z * z);
}
M2 lin(double s1, const M2 &A, double s2, const M2 &B, double s3, const M2 &Cm) {
    return M2{s1 * A.a + s2 * B.a + s3 * Cm.a, s1 * A.b + s2 * B.b + s3 * Cm.b, s1 * A.c + s2 * B.c + s3 * Cm.c,
              s1 * A.d + s2 * B.d + s3 * Cm.d};
}
void pack_m2(const M2 &M, float *o) {
    const cd e[4] = {M.a, M.b, M.c, M.d};
    for (int k = 0; k < 4; ++k) { o[2 * k] = (float)e[k].real(); o[2 * k + 1] = (float)e[k].imag(); }
}
// ---- the switched block (kSwModes): phi-functions of the REAL 4x4 operator of a pair's anti-windup mode, by the
// exponential of the augmented matrix [[hL, I, 0, 0], [0, 0, I, 0], [0, 0, 0, I], [0, 0, 0, 0]] (first block row =
// phi_0 .. phi_3), Taylor series with scaling and squaring in long double (|hL| reaches 1.5e4: 15 squarings)
typedef long double lreal;
struct R4 { lreal a[4][4]; };
struct R16 { lreal a[16][16]; };
R16 mul16(const R16 &A, const R16 &B) {
    R16 Cm;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            lreal t = 0;
            for (int k = 0; k < 16; ++k) t += A.a[i][k] * B.a[k][j];
            Cm.a[i][j] = t;
        }
    return Cm;
}
void phis4(const R4 &L, double h, R4 (&out)[4]) {
    R16 M{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) M.a[i][j] = (lreal)h * L.a[i][j];
    for (int b = 0; b < 3; ++b)
        for (int i = 0; i < 4; ++i) M.a[4 * b + i][4 * (b + 1) + i] = 1;
    lreal norm = 0;
    for (int j = 0; j < 16; ++j) {
        lreal c = 0;
        for (int i = 0; i < 16; ++i) c += fabsl(M.a[i][j]);
        norm = c > norm ? c : norm;
    }
    int sq = 0;
    while (norm > 0.25L) { norm *= 0.5L; ++sq; }
    const lreal sc = ldexpl(1.0L, -sq);
    for (auto &row : M.a)
        for (auto &v : row) v *= sc;
    R16 T{}, term{};
    for (int i = 0; i < 16; ++i) T.a[i][i] = term.a[i][i] = 1;
    for (int k = 1; k <= 22; ++k) {
        term = mul16(term, M);
        for (auto &row : term.a)
            for (auto &v : row) v /= (lreal)k;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) T.a[i][j] += term.a[i][j];
    }
    // (the identity blocks were scaled with the rest: T = exp(sc W), and squaring sq times gives exp(W) itself)
    for (int q = 0; q < sq; ++q) T = mul16(T, T);
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[k].a[i][j] = T.a[i][4 * k + j];
}
R4 lin4(double s1, const R4 &A, double s2, const R4 &B, double s3, const R4 &Cm) {
    R4 o;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o.a[i][j] = (lreal)s1 * A.a[i][j] + (lreal)s2 * B.a[i][j] + (lreal)s3 * Cm.a[i][j];
    return o;
}
void pack_r4(const R4 &M, float *o) {           // column pairs of rows (0, 1), then of rows (2, 3)
    for (int k = 0; k < 4; ++k) { o[2 * k] = (float)M.a[0][k]; o[2 * k + 1] = (float)M.a[1][k]; }
    for (int k = 0; k < 4; ++k) { o[8 + 2 * k] = (float)M.a[2][k]; o[8 + 2 * k + 1] = (float)M.a[3][k]; }
}
void make_switched(const cd Z0, double c0kp, double wp, const double (&hs)[kCoefSets], float *sw /* kSwFloats */) {
    for (int set = 0; set < kCoefSets; ++set)
        for (int mode = 0; mode < kSwModes; ++mode) {
            // rows / columns (iR, iI, uR, uI) of  Lz = [[-Z0, c0 Kp], [-wp, -wp]];  a frozen component of u: its row zeroed
            R4 L{};
            L.a[0][0] = -Z0.real(); L.a[0][1] = Z0.imag(); L.a[0][2] = c0kp;
            L.a[1][0] = -Z0.imag(); L.a[1][1] = -Z0.real(); L.a[1][3] = c0kp;
            if (!(mode & 1)) { L.a[2][0] = -wp; L.a[2][2] = -wp; }
            if (!(mode & 2)) { L.a[3][1] = -wp; L.a[3][3] = -wp; }
            const double h = hs[set];
            R4 ph[4], pf[4];
            phis4(L, 0.5 * h, ph);
            phis4(L, h, pf);
            const R4 Z{};
            const R4 a52 = lin4(0.5, ph[2], -1.0, pf[3], 0.25, pf[2]);
            const R4 a52f = lin4(1.0, a52, -0.5, ph[3], 0, Z);
            const R4 a54 = lin4(0.25, ph[2], -1.0, a52f, 0, Z);
            const R4 a51 = lin4(0.5, ph[1], -2.0, a52f, -1.0, a54);
            float(*m)[kSwMatFloats] = reinterpret_cast<float(*)[kSwMatFloats]>(sw + ((size_t)set * kSwModes + mode) * kMatCount * kSwMatFloats);
            pack_r4(ph[0], m[EM_E2]);
            pack_r4(pf[0], m[EM_E]);
            pack_r4(lin4(0.5 * h, ph[1], 0, Z, 0, Z), m[EM_A21]);
            pack_r4(lin4(0.5 * h, ph[1], -h, ph[2], 0, Z), m[EM_A31]);
            pack_r4(lin4(h, ph[2], 0, Z, 0, Z), m[EM_A32]);
            pack_r4(lin4(h, pf[1], -2 * h, pf[2], 0, Z), m[EM_A41]);
            pack_r4(lin4(h, pf[2], 0, Z, 0, Z), m[EM_A42]);
            pack_r4(lin4(h, a51, 0, Z, 0, Z), m[EM_A51]);
            pack_r4(lin4(h, a52f, 0, Z, 0, Z), m[EM_A52]);
            pack_r4(lin4(h, a54, 0, Z, 0, Z), m[EM_A54]);
            pack_r4(lin4(h, pf[1], -3 * h, pf[2], 4 * h, pf[3]), m[EM_B1]);
            pack_r4(lin4(-h, pf[2], 4 * h, pf[3], 0, Z), m[EM_B4]);
            pack_r4(lin4(4 * h, pf[2], -8 * h, pf[3], 0, Z), m[EM_B5]);
        }
}

EtdK make_etd(const double *par, int micro, int micro_coarse, int graded, float *coef /* host, kCoefTotal, may be NULL */) {
    const double iLf = 1.0 / par[PVDER_PAR_LF];
    const cd kvi(par[PVDER_PAR_KVI_RE], par[PVDER_PAR_KVI_IM]);
    const cd Z0 = iLf * (kvi + par[PVDER_PAR_RF]) + cd(0.0, 1.0);
    const double c0 = iLf * par[PVDER_PAR_VDC_REF0] / 2.0;
    const double wp = par[PVDER_PAR_WP], kp = par[PVDER_PAR_KP_GCC];
    const M2 L{-Z0, cd(c0 * kp, 0), cd(-wp, 0), cd(-wp, 0)};
    const cd tr = L.a + L.d, det = L.a * L.d - L.b * L.c;
    const cd disc = std::sqrt(tr * tr - 4.0 * det);
    const cd l1 = 0.5 * (tr + disc), l2 = 0.5 * (tr - disc);
    EtdK E;
    float(*m)[kMatCount][kMatFloats] = reinterpret_cast<float(*)[kMatCount][kMatFloats]>(coef);
    const M2 Z{};
    // step sizes of the schedule: first step behind a jump, coarse step, remainder of the jump's sim time step
    const double hs[kCoefSets] = {1.0 / (60.0 * micro), 1.0 / (60.0 * micro_coarse),
                                  (graded && micro > 1) ? (1.0 - 1.0 / micro) / 60.0 : 1.0 / (60.0 * micro)};
    for (int set = 0; set < kCoefSets; ++set) {
        const double h = hs[set];
        E.rk4_sub[set] = (int)std::ceil(4096.0 * 60.0 * h - 1e-9);      // (23-state tail only)
        if (E.rk4_sub[set] < 1) E.rk4_sub[set] = 1;
        E.hset[set] = (float)h;
        if (!coef) continue;              // per-launch call: only the scalars are needed (the phi matrices
                                          // cost ~3 us of complex exponentials; they live in the caller's buffer)
        M2 ph[4], pf[4];      // phi_k(h L/2), phi_k(h L)
        for (int k = 0; k < 4; ++k) {
            ph[k] = fun2(L, l1, l2, phi(k, 0.5 * h * l1), phi(k, 0.5 * h * l2));
            pf[k] = fun2(L, l1, l2, phi(k, h * l1), phi(k, h * l2));
        }
        // Hochbruck-Ostermann (2005, eq. 5.19) coefficient functions, h folded in
        const M2 a52 = lin(0.5, ph[2], -1.0, pf[3], 0.25, pf[2]);           // phi2,5/2 - phi3,4 + phi2,4/4 ...
        const M2 a52f = lin(1.0, a52, -0.5, ph[3], 0, Z);                    // ... - phi3,5/2
        const M2 a54 = lin(0.25, ph[2], -1.0, a52f, 0, Z);
        const M2 a51 = lin(0.5, ph[1], -2.0, a52f, -1.0, a54);
        pack_m2(ph[0], m[set][EM_E2]);
        pack_m2(pf[0], m[set][EM_E]);
        pack_m2(lin(0.5 * h, ph[1], 0, Z, 0, Z), m[set][EM_A21]);
        pack_m2(lin(0.5 * h, ph[1], -h, ph[2], 0, Z), m[set][EM_A31]);
        pack_m2(lin(h, ph[2], 0, Z, 0, Z), m[set][EM_A32]);
        pack_m2(lin(h, pf[1], -2 * h, pf[2], 0, Z), m[set][EM_A41]);
        pack_m2(lin(h, pf[2], 0, Z, 0, Z), m[set][EM_A42]);
        pack_m2(lin(h, a51, 0, Z, 0, Z), m[set][EM_A51]);
        pack_m2(lin(h, a52f, 0, Z, 0, Z), m[set][EM_A52]);
        pack_m2(lin(h, a54, 0, Z, 0, Z), m[set][EM_A54]);
        pack_m2(lin(h, pf[1], -3 * h, pf[2], 4 * h, pf[3]), m[set][EM_B1]);
        pack_m2(lin(-h, pf[2], 4 * h, pf[3], 0, Z), m[set][EM_B4]);
        pack_m2(lin(4 * h, pf[2], -8 * h, pf[3], 0, Z), m[set][EM_B5]);
        // classic RK4 needs h <~ 4.1 us on this spectrum: 4096 micro-steps per 1/60 s
    }
    if (coef) make_switched(Z0, c0 * kp, wp, hs, coef + kCoefFloats);
    E.c0kp = (float)(c0 * kp);
    // Anti-windup switches the fast dynamics only while |m_p| > m_limit.  On a step that starts at
    // an input jump the under-damped current loop (zeta ~ 0.17) overshoots, so |m| may cross the
    // limit inside the step although both ends are below it: keep 3 % clearance there; between
    // jumps m moves by < 1e-3 per step and 0.2 % is ample.
    const double ge = 0.97 * par[PVDER_PAR_M_LIMIT], gs = 0.998 * par[PVDER_PAR_M_LIMIT];
    E.sat2_event = (float)(ge * ge);
    E.sat2_smooth = (float)(gs * gs);
    E.sat2_flag = (float)(0.9875 * 0.9875 * par[PVDER_PAR_M_LIMIT] * par[PVDER_PAR_M_LIMIT]);
    E.i2_to_m2 = (float)(ge * ge / (0.81 * par[PVDER_PAR_IREF_LIMIT] * par[PVDER_PAR_IREF_LIMIT]));
    return E;
}

template <typename T>
int reset_impl(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, T *fstate,
               int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *mask, T *out, int n_env,
               void *stream, const float *lane_par = nullptr) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (!ar || !fstate || !istate || !ev_vg || !ev_si) return PVDER_EINVAL;
    ResetK R;
    rc = fold_reset(par, cfg, ar, R);
    if (rc) return rc;
    const int ph = (int)par[PVDER_PAR_PHASES];
    const ModelK<T> P = fold_params<T>(par, sizeof(T) == 4);
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (ph == 3)
        hipLaunchKernelGGL((reset_kernel<T, 3>), grid, block, 0, st, P, R, fstate, istate, ev_vg, ev_si, mask, out,
                           n_env, lane_par);
    else
        hipLaunchKernelGGL((reset_kernel<T, 1>), grid, block, 0, st, P, R, fstate, istate, ev_vg, ev_si, mask, out,
                           n_env, lane_par);
    return hip_rc(hipGetLastError());
}

// every row of the generator's tables for the episode each env is in (pvder_autoreset.lazy_events leaves the rows beyond an env's
// clock to the steps that come to them: a reader that wants whole tables -- a test, a checkpoint meant to be read -- asks for them)
__global__ __launch_bounds__(kBlock) void complete_events_kernel(const ResetK R, float sinsol0, const int32_t *__restrict__ istate,
                                                                 float *__restrict__ ev_vg, float *__restrict__ ev_si, int n_env) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_env) return;
    const size_t N = (size_t)n_env;
    fill_events(R.spec, sinsol0, R.seed, (uint64_t)(R.env_id_base + (R.env_ids ? (int64_t)R.env_ids[e] : (int64_t)e)),
                (uint64_t)istate[PVDER_I_EPISODE * N + e], ev_vg, ev_si, N, e);
}

// A finished env returns the stale tuple (PVDER_env.py:145-152) and the step kernels leave its rows of the
// output block alone.  When a step writes ANOTHER block than the previous one (rollouts with per-step
// blocks, the two alternating blocks of the step + all-gather path) the tuple is carried over first.
// Only enqueued without auto-reset (with it no env is ever finished on entry).
template <typename T>
__global__ __launch_bounds__(kBlock) void carry_stale_kernel(const int32_t *__restrict__ istate, const T *__restrict__ prev,
                                                             T *__restrict__ out, int n_env) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_env) return;
    const size_t N = (size_t)n_env;
    if (istate[PVDER_I_STATUS * N + e] == 0) return;
#pragma unroll
    for (int q = 0; q < PVDER_OROWS; ++q) out[(size_t)q * N + e] = prev[(size_t)q * N + e];
}

// ---- plans: everything a step launch needs that does not change from step to step, folded once
// (parameter folding, phi-function scalars, reset arguments).  The per-call entry points below build
// one on the stack; pvder_plan_create() keeps one on the heap for callers that step many times.
}  // namespace

// The single env's read-back without a copy packet (pvder_plan_step_mirror): output rows and counters of one env -- and the
// trajectory record, a plain copy of the state rows + output block -- written by ONE wave behind the step kernel, the sequence
// word last (every lane's stores released at system scope before the barrier; `host` is fine-grained pinned host memory).
template <typename T>
__global__ __launch_bounds__(64) void mirror_env_kernel(const T *__restrict__ out, const int32_t *__restrict__ istate, int n_env, int env,
                                                        T *__restrict__ snap, const T *__restrict__ snap_src, int snap_n,
                                                        double *__restrict__ host, unsigned long long seq) {
    const int t = threadIdx.x;
    if (t < PVDER_OROWS) host[t] = (double)out[(size_t)t * n_env + env];
    else if (t < PVDER_OROWS + 4) host[t] = (double)istate[(size_t)(t - PVDER_OROWS) * n_env + env];
    for (int k = t; k < snap_n; k += 64) snap[k] = snap_src[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (t == 0) __hip_atomic_store((unsigned long long *)(host + PVDER_OROWS + 4), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct pvder_plan {
    int integrator, ph, n_env;
    int episode_steps;               // env steps of an episode (max_sim_time, TimeLimit): bounds the episode ends inside a rollout
    LinArgs lin;                     // etdlin; lin.A alone is the ETDRK4 argument block
    ModelK<float> Pf;                // classic RK4 / single-phase follow-up
    StepK<float> Cf, C4;
    ModelK<double> Pd;               // RK45 / etdlin64
    StepK<double> Cd;
    Lin64K l64;                      // etdlin64
    ResetK R;
    const float *lane_par;           // etdrk4 / etdlin: [PVDER_LPF_ROWS][n_env] folded per-lane constants (pvder_plan_lane_params), or NULL
    const float *k4lane = nullptr;   // etdlin: the two Krogstad blocks of step_etdlin_lane_kernel (pvder_lin_coef.k4 & 2), or NULL
    int lane_steps = 2;              // ... and the Krogstad steps per env step they were folded for (pvder_lin_coef.k4 bits 8-15; 0 = 2)
    int hyb_mode = 0;                // pvder_step_cfg.off_range: 0 = the hybrid kernel when an env comes near the duty-cycle limit (auto),
                                     // 1 = terminate, 2 = held operator sets (rounds 3-5), 3 = the hybrid kernel behind EVERY step launch
    int *hyb_flag_host = nullptr;    // auto: pinned host word the step kernels write when an env comes near the limit
    mutable int hyb_hold = 0;        // auto: step launches the hybrid kernel still follows (re-armed whenever the flag is seen)
    // auto: the host must SEE the flag before the duty cycle gets from 0.9875 to its limit (14 env steps of the reference's ramp):
    // never more than 16 step launches of this plan in flight (an event every 8 launches; the one recorded 16 launches ago is
    // waited for -- by polling, the queue holds 8 launches of work meanwhile)
    int *hyb_list = nullptr;         // device: [0], [1] = blocks found (by launch parity), [16 + i] = 64-env blocks with an env for the hybrid kernel
    mutable unsigned hyb_launches = 0;
    mutable unsigned thr_n = 0;      // auto: step launches numbered (EtdArgs.hyb_seq; hyb_flag_host[1] = the last one that started)
    mutable bool thr_off = false;
    bool graph_hints = false;        // pvder_plan_graph_hints: a CAPTURED pvder_plan_rollout keeps the clock hints it was captured with
    mutable int ksub_hint;           // etdlin: the clock (ksub) most envs of the batch are at when the NEXT step starts, -1 = unknown
                                     // (pvder_plan_hint; advanced by every step this plan enqueues: a prefetch hint, never a result)
    int n_sub;
    ~pvder_plan() {
        if (hyb_flag_host) (void)hipHostFree(hyb_flag_host);
        if (hyb_list) (void)hipFree(hyb_list);
    }
};

namespace {

int fill_plan(pvder_plan &p, int integrator, const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
              const float *coef, const pvder_lin_coef *lin, int n_env) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (integrator < PVDER_INT_ETDLIN || integrator > PVDER_INT_ETDLIN64) return PVDER_EINVAL;
    p.integrator = integrator;
    p.ph = (int)par[PVDER_PAR_PHASES];
    p.n_env = n_env;
    p.episode_steps = (cfg->max_ksub + cfg->n_sub - 1) / cfg->n_sub;
    if (cfg->max_steps > 0 && cfg->max_steps < p.episode_steps) p.episode_steps = cfg->max_steps;
    p.ksub_hint = -1;
    p.lane_par = nullptr;
    p.n_sub = cfg->n_sub;
    rc = fold_reset(par, cfg, ar, p.R);
    if (rc) return rc;
    if (integrator == PVDER_INT_RK45 || integrator == PVDER_INT_ETDLIN64) {
        if (!(cfg->rtol > 0.0) || !(cfg->atol > 0.0)) return PVDER_EINVAL;
        p.Pd = fold_params<double>(par, false);
        p.Cd = fold_cfg<double>(cfg);
        if (integrator == PVDER_INT_ETDLIN64) {
            // three-phase, events on env-step boundaries (an env step = one segment); lin->sets: DOUBLES
            if (p.ph != 3 || !lin || !lin->sets || lin->n_sets < 3 || lin->n_sets > 1 + kL64MaxLevels) return PVDER_EINVAL;
            if (cfg->ev_count > 0 && (cfg->ev_ds % cfg->n_sub != 0 || cfg->ev_s0 % cfg->n_sub != 0)) return PVDER_EINVAL;
            p.l64.coef = (const double *)lin->sets;
            p.l64.levels = lin->n_sets - 1;
            p.l64.accept = kL64Accept;
            p.l64.min_level = kL64MinLevel;
            if (const char *v = getenv("PVDER_L64_ACCEPT")) p.l64.accept = atof(v);          // (A/B knobs)
            if (const char *v = getenv("PVDER_L64_MINLEVEL")) p.l64.min_level = atoi(v);
            if (p.l64.min_level < 1) p.l64.min_level = 1;
            const double gs = 0.998 * par[PVDER_PAR_M_LIMIT];
            p.l64.sat2 = gs * gs;
            p.l64.ir2_guard = 0.81 * par[PVDER_PAR_IREF_LIMIT] * par[PVDER_PAR_IREF_LIMIT];
        }
        return 0;
    }
    p.Pf = fold_params<float>(par, true);
    p.Cf = fold_cfg<float>(cfg);
    p.C4 = p.Cf;
    p.C4.micro = 8192;               // single-phase follow-up: classic RK4, poles at 1.0e6 rad/s
    if (integrator == PVDER_INT_RK4) return 0;
    if (!coef) return PVDER_EINVAL;
    EtdArgs &A = p.lin.A;
    A.P = p.Pf;
    A.C = p.Cf;
    A.R = p.R;
    A.Ec = make_etd(par, cfg->micro, cfg->micro_coarse > 0 ? cfg->micro_coarse : cfg->micro, cfg->graded, nullptr);
    A.coef = coef;
    A.n_env = n_env;
    // PVDER_ETD_VARIANT=full / cfg->reserved bit 0 send every env down the 23-state path (A/B, tests)
    p.hyb_mode = cfg->off_range;
    if (const char *v = getenv("PVDER_SAT_HYBRID")) {                  // (A/B: 0 = held operator sets, 3 = the hybrid kernel always)
        if (v[0] == '0' && p.hyb_mode != PVDER_OFF_RANGE_TERMINATE) p.hyb_mode = 2;
        if (v[0] == '3' && p.hyb_mode != PVDER_OFF_RANGE_TERMINATE) p.hyb_mode = 3;
    }
    A.hyb_flag = nullptr;
    if (p.hyb_mode == 0 && p.ph == 3 && cfg->n_sub <= kHybMaxSub) {
        void *dev = nullptr;
        if (hipHostMalloc((void **)&p.hyb_flag_host, 64, hipHostMallocMapped) == hipSuccess && p.hyb_flag_host
            && hipHostGetDevicePointer(&dev, p.hyb_flag_host, 0) == hipSuccess && dev) {
            p.hyb_flag_host[0] = 0; p.hyb_flag_host[1] = 0;
            A.hyb_flag = (int *)dev;
        } else {
            // (no device: plans are still folded and validated -- tests/test_host.py; nothing can be launched there anyway)
            (void)hipGetLastError();
            if (p.hyb_flag_host) (void)hipHostFree(p.hyb_flag_host);
            p.hyb_flag_host = nullptr;
        }
    }
    if ((p.hyb_mode == 0 || p.hyb_mode == 3) && p.ph == 3 && cfg->n_sub <= kHybMaxSub) {
        const size_t words = 16 + (size_t)((n_env + kHybBlock - 1) / kHybBlock);
        if (hipMalloc((void **)&p.hyb_list, words * sizeof(int)) != hipSuccess || hipMemset(p.hyb_list, 0, words * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();         // (no device: see above -- without the list the hybrid kernel is never launched)
            if (p.hyb_list) (void)hipFree(p.hyb_list);
            p.hyb_list = nullptr;
        }
    }
    static const bool full_env = []() { const char *v = getenv("PVDER_ETD_VARIANT"); return v && v[0] == 'f'; }();
    A.force_full = (p.ph == 3 && (full_env || (cfg->reserved & 1))) ? 1 : 0;
    A.rk4_fallback = (cfg->reserved & 2) ? 1 : 0;
    A.sat_hybrid = 0;                // (set per launch: launch_plan)
    A.ph1_micro = p.C4.micro;
    if (integrator == PVDER_INT_ETDLIN) {
        if (!lin || !lin->sets || !lin->sched || lin->n_sets < 2 || lin->n_sets > pvder::kLinMaxSets) return PVDER_EINVAL;
        if (lin->len_ticks[0] < 1 || !(lin->est_guard > 0.f)) return PVDER_EINVAL;
        if (lin->n_sched_rows < 2 * cfg->n_sub || lin->n_sched_rows > 2 * cfg->n_sub + 128) return PVDER_EINVAL;
        for (int s = 1; s < lin->n_sets; ++s)
            if (lin->len_ticks[s] < 1) return PVDER_EINVAL;
        LinK &K = p.lin.Lk;
        K.lcoef = lin->sets;
        K.sched = (const unsigned long long *)lin->sched;
        K.est_guard = lin->est_guard;
        // the careful pass takes over where |iref| comes within 5 % of the current limit
        K.ir2_careful = (float)(0.95 * 0.95 * par[PVDER_PAR_IREF_LIMIT] * par[PVDER_PAR_IREF_LIMIT]);
        K.sat2_hard = (float)(0.9995 * 0.9995 * par[PVDER_PAR_M_LIMIT] * par[PVDER_PAR_M_LIMIT]);
        // the propagator of a small imbalance: its (x, x) entry is exp(-T Ki/Kp) -- an all-zero block (a caller written
        // against ABI 10) would silently drop what an anti-windup episode leaves in the controller's integrators
        if (p.ph == 3 && !(lin->seq_prop[8] > 0.f && lin->seq_prop[8] <= 1.f)) return PVDER_EINVAL;
        for (int k = 0; k < 18; ++k) K.seqE[k] = lin->seq_prop[k];
        // the Krogstad block of the quiet env step sits behind the sets (pvder_lin_coef.k4; PVDER_NO_K4=1: A/B runs without it)
        static const bool no_k4 = []() { const char *v = getenv("PVDER_NO_K4"); return v && v[0] == '1'; }();
        K.k4coef = ((lin->k4 & 1) != 0 && !no_k4) ? lin->sets + pvder::kLinMatFloats + (size_t)(lin->n_sets - 1) * pvder::kLinSetFloats : nullptr;
        // ... and behind it the two blocks of the per-lane parameter path (half a quiet env step, half of what a tick leaves)
        p.k4lane = ((lin->k4 & 3) == 3 && p.ph == 3)
                       ? lin->sets + pvder::kLinMatFloats + (size_t)(lin->n_sets - 1) * pvder::kLinSetFloats + pvder::kLinK4Floats : nullptr;
        p.lane_steps = (lin->k4 >> 8) & 0xff;
        if (p.lane_steps == 0) p.lane_steps = 2;
        if (p.k4lane && p.lane_steps > 16) return PVDER_EINVAL;
    }
    return 0;
}

// the hint `steps` env steps later: envs in lockstep finish together -- and restart together (same-step auto-reset)
void advance_hint(const pvder_plan &p, int steps) {
    if (p.ksub_hint < 0 || p.n_sub <= 0 || p.episode_steps <= 0) return;
    long long k = (long long)(p.ksub_hint / p.n_sub) + steps;
    if (p.R.enabled) k %= p.episode_steps;
    else if (k > p.episode_steps) k = p.episode_steps;        // (finished, not running: whatever)
    p.ksub_hint = (int)k * p.n_sub;
}

// the hint as the step kernel takes it: the clock, and whether the env step that starts there is a quiet one (one segment, no
// event row change at its start: lin_first_segment) -- which coefficient registers the launch loads first
int hint_word(const pvder_plan &p) {
    const int k = p.ksub_hint;
    if (k < 0) return -1;
    const StepK<float> &C = p.lin.A.C;
    bool at_event = false;
    int until = 0x3fffffff;
    if (C.ev_count > 0) {
        const int d = k - C.ev_s0;
        if (d < 0) until = -d;
        else {
            const int qd = d / C.ev_ds, rd = d - qd * C.ev_ds;
            until = qd + 1 < C.ev_count ? C.ev_ds - rd : 0x3fffffff;
            at_event = rd == 0 && qd < C.ev_count;
        }
    }
    const bool quiet = p.lin.Lk.k4coef != nullptr && !at_event && until >= C.n_sub;
    return quiet ? (k | kLinHintQuiet) : k;
}

int launch_plan(const pvder_plan &p, void *fstate, int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *action,
                void *out, double *hstep, float *stats, int zero_stats, void *stream, const void *out_prev = nullptr,
                bool hint_in_graphs = false) {
    if (!fstate || !istate || !ev_vg || !ev_si || !action || !out) return PVDER_EINVAL;
    const int n_env = p.n_env;
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    // (ONE query per launch: the clock hint and the launch throttle below both want to know)
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = st != nullptr && hipStreamIsCapturing(st, &cap_status) == hipSuccess && cap_status == hipStreamCaptureStatusActive;
    if (out_prev && out_prev != out && !p.R.enabled) {       // stale tuples of finished envs follow the output block
        if (p.integrator >= PVDER_INT_RK45)
            hipLaunchKernelGGL(carry_stale_kernel<double>, grid, block, 0, st, istate, (const double *)out_prev, (double *)out, n_env);
        else
            hipLaunchKernelGGL(carry_stale_kernel<float>, grid, block, 0, st, istate, (const float *)out_prev, (float *)out, n_env);
    }
    const int fresh = zero_stats != 0 ? 1 : 0;       // first launch of the step: overwrite the statistics slots
    switch (p.integrator) {
    case PVDER_INT_ETDLIN:
    case PVDER_INT_ETDRK4: {
        LinArgs G = p.lin;
        EtdArgs &A = G.A;
        A.fstate = (float *)fstate; A.istate = istate; A.ev_vg = ev_vg; A.ev_si = ev_si;
        A.action = action; A.out = (float *)out; A.stats = stats; A.stats_fresh = fresh;
        // the hybrid kernel behind this launch? (hyb_mode 3: always; 0: while an env is, or lately was, near the duty-cycle limit)
        bool hyb = false;
        if (p.ph == 3 && p.n_sub <= kHybMaxSub && p.hyb_list && (p.hyb_mode == 0 || p.hyb_mode == 3)) {
            if (p.hyb_mode == 3) hyb = true;
            else {
                if (p.hyb_flag_host && __atomic_load_n(p.hyb_flag_host, __ATOMIC_RELAXED) != 0) {
                    __atomic_store_n(p.hyb_flag_host, 0, __ATOMIC_RELAXED);
                    p.hyb_hold = kHybHoldLaunches;
                }
                hyb = p.hyb_hold > 0;
                if (p.hyb_hold > 0) --p.hyb_hold;
            }
        }
        A.sat_hybrid = hyb ? 1 : 0;
        // auto mode: at most 16 step launches in flight (the flag of an env that comes near the limit is seen within 16 steps of its
        // being raised; hyb_progress): the word the kernels write as they start, no packet on the stream
        A.hyb_seq = 0;
        if (p.hyb_mode == 0 && p.hyb_flag_host && !capturing && !p.thr_off) {
            if (++p.thr_n == 0u) ++p.thr_n;                    // (0 = "do not report")
            const int seq = (int)p.thr_n;
            volatile int *const started = p.hyb_flag_host + 1;
            auto ahead = [&]() { const unsigned d = p.thr_n - (unsigned)*started; return d > 16u && d < 0x80000000u; };
            if (ahead()) {
                const auto t0 = std::chrono::steady_clock::now();
                unsigned spins = 0;
                while (ahead()) {
                    if ((++spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                        p.thr_off = true;          // (the word does not move: a kernel that does not report -- never wait again)
                        break;
                    }
                }
            }
            A.hyb_seq = seq;
        }
        // ONE launch -- the fast path, whose workgroups themselves run the 23-state path (three-phase) / classic RK4
        // micro-steps, 8192 per 1/60 s (single phase: the "balanced" kernel IS the model) for the envs they must defer
        // (none in PVDER-v0 rollouts).
        if (p.integrator == PVDER_INT_ETDLIN) {
            static const bool no_w1 = []() { const char *v = getenv("PVDER_NO_W1"); return v && v[0] == '1'; }();      // (A/B)
            // The hint is a kernel ARGUMENT: a launch captured into a HIP graph replays the clock and the quiet bit it was
            // captured with while the plan's own hint moves on -- every replay at another clock would fetch the event rows twice
            // and preload the wrong coefficient block (slower than no hint at all; results unaffected).  A captured step is
            // therefore taken with the hint cleared, and leaves the plan's clock alone -- per-step launches and rollouts
            // alike.  A caller that replays its K-step rollout graph exactly where it was captured (bench.py's window)
            // opts in with pvder_plan_graph_hints (hint_in_graphs).
            const int hint = (capturing && !hint_in_graphs) ? -1 : hint_word(p);
            if (p.ph == 3 && p.lane_par)
                hipLaunchKernelGGL(step_etdlin_lane_kernel, grid, block, 0, st, G, p.lane_par, p.k4lane, p.lane_steps);
            else if (p.ph == 3 && n_env <= kOneWavePerSimdEnvs && !no_w1)
                hipLaunchKernelGGL(step_etdlin_kernel_w1<3>, grid, block, 0, st, A.fstate, A.istate, A.action, G.Lk.lcoef, n_env, hint, G.Lk.k4coef, G);
            else if (p.ph == 3)
                hipLaunchKernelGGL(step_etdlin_kernel<3>, grid, block, 0, st, A.fstate, A.istate, A.action, G.Lk.lcoef, n_env, hint, G.Lk.k4coef, G);
            else
                hipLaunchKernelGGL(step_etdlin_kernel<1>, grid, block, 0, st, A.fstate, A.istate, A.action, G.Lk.lcoef, n_env, hint, G.Lk.k4coef, G);
            if (!capturing || hint_in_graphs) advance_hint(p, 1);
        } else {
            // per-lane parameters: a phase per lane (four times the waves, 0.6 x the instructions per wave: the LATENCY form) while
            // that leaves one or two waves per SIMD -- 45 us per batched step up to 16 384 envs, 65 up to 32 768 --, a lane per env
            // (72 us whatever the batch up to 65 536: the THROUGHPUT form) beyond; PVDER_LANE1=1 / PVDER_LANE3=1 force one (A/B)
            static const int lane_form = []() { const char *a = getenv("PVDER_LANE1"), *b = getenv("PVDER_LANE3");
                                                return (a && a[0] == '1') ? 1 : ((b && b[0] == '1') ? 3 : 0); }();
            const bool lane1 = lane_form == 1 || (lane_form == 0 && n_env > kLane3MaxEnvs);
            if (p.ph == 3 && p.lane_par && lane1) hipLaunchKernelGGL(step_etdrk4_lane_kernel, grid, block, 0, st, A, p.lane_par);
            else if (p.ph == 3 && p.lane_par)
                hipLaunchKernelGGL(step_etdrk4_lane3_kernel, dim3((unsigned)(((size_t)n_env * 4 + kBlock - 1) / kBlock)), block, 0, st, A, p.lane_par);
            else if (p.ph == 3) hipLaunchKernelGGL(step_etdrk4_bal_kernel<3>, grid, block, 0, st, A);
            else hipLaunchKernelGGL(step_etdrk4_bal_kernel<1>, grid, block, 0, st, A);
        }
        if (hyb) {
            static size_t lds_ok[64] = {0};          // (beyond 64 KB of dynamic LDS the kernel has to be told: once per device)
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PVDER_EINVAL;
            if (lds_ok[dev] < kHybLdsBytes) {
                const hipError_t e = hipFuncSetAttribute((const void *)step_sat_hybrid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHybLdsBytes);
                if (e != hipSuccess) return hip_rc(e);
                lds_ok[dev] = kHybLdsBytes;
            }
            const int n_blk = (n_env + kHybBlock - 1) / kHybBlock;
            const int parity = (int)(p.hyb_launches++ & 1u);
            // (a captured launch is replayed with THIS parity every time: its counter is cleared inside the graph)
            if (capturing && hipMemsetAsync(p.hyb_list + parity, 0, sizeof(int), st) != hipSuccess) return hip_rc(hipGetLastError());
            hipLaunchKernelGGL(hybrid_scan_kernel, grid, block, 0, st, (const int32_t *)istate, n_env, p.hyb_list, parity, n_blk);
            hipLaunchKernelGGL(step_sat_hybrid_kernel, dim3(n_blk * kHybQuarters < kHybGrid ? n_blk * kHybQuarters : kHybGrid), dim3(kHybThreads), kHybLdsBytes, st, A, p.lane_par,
                               (const int *)p.hyb_list, parity, n_blk);
        }
        break;
    }
    case PVDER_INT_RK4: {
        // PVDER_RK4_VARIANT=generic selects the un-packed template (A/B runs); default = packed kernel
        static const bool generic = []() { const char *v = getenv("PVDER_RK4_VARIANT"); return v && v[0] == 'g'; }();
        if (p.ph == 3 && !generic)
            hipLaunchKernelGGL(step_rk4_pk_kernel, grid, block, 0, st, p.Pf, p.Cf, p.R, (float *)fstate, istate, ev_vg,
                               ev_si, action, (float *)out, stats, n_env, fresh);
        else if (p.ph == 3)
            hipLaunchKernelGGL((step_rk4_kernel<float, 3>), grid, block, 0, st, p.Pf, p.Cf, p.R, (float *)fstate, istate,
                               ev_vg, ev_si, action, (float *)out, stats, n_env, fresh, 0);
        else
            hipLaunchKernelGGL((step_rk4_kernel<float, 1>), grid, block, 0, st, p.Pf, p.Cf, p.R, (float *)fstate, istate,
                               ev_vg, ev_si, action, (float *)out, stats, n_env, fresh, 0);
        break;
    }
    default: {
        if (!hstep) return PVDER_EINVAL;
        int rk_fresh = fresh, rk_status = 0;
        if (p.integrator == PVDER_INT_ETDLIN64) {
            // the tolerance-match path on the balanced reduction; the RK45 kernel behind it takes the envs it hands over
            const size_t lds = sizeof(double) * ((size_t)(1 + p.l64.levels) * kL64Set + 11 * (size_t)kBlock);
            // beyond 64 KB of dynamic LDS the kernel has to be told -- once per size AND per device (the attribute is the device's:
            // a process that steps on a second GPU needs it there too; gfx950's 160 KB per CU is what this kernel is sized for)
            static size_t lds_allowed[64] = {0};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PVDER_EINVAL;
            if (lds > lds_allowed[dev]) {
                const hipError_t e = hipFuncSetAttribute((const void *)step_etdlin64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return hip_rc(e);
                lds_allowed[dev] = lds;
            }
            hipLaunchKernelGGL(step_etdlin64_kernel, grid, block, lds, st, p.Pd, p.Cd, p.R, p.l64, (double *)fstate, istate, ev_vg,
                               ev_si, action, (double *)out, stats, n_env, fresh);
            rk_fresh = 0;
            rk_status = (int)kDeferred;
        }
        if (p.ph == 3)
            hipLaunchKernelGGL((step_rk45_kernel<double, 3>), grid, block, 0, st, p.Pd, p.Cd, p.R, (double *)fstate, istate,
                               ev_vg, ev_si, action, (double *)out, hstep, stats, n_env, rk_fresh, rk_status);
        else
            hipLaunchKernelGGL((step_rk45_kernel<double, 1>), grid, block, 0, st, p.Pd, p.Cd, p.R, (double *)fstate, istate,
                               ev_vg, ev_si, action, (double *)out, hstep, stats, n_env, rk_fresh, rk_status);
        if (p.R.enabled) {      // same-step auto-reset as a second pass (see step_rk45_kernel)
            if (p.ph == 3)
                hipLaunchKernelGGL((reset_kernel<double, 3, true>), grid, block, 0, st, p.Pd, p.R, (double *)fstate, istate,
                                   ev_vg, ev_si, (const uint8_t *)nullptr, (double *)out, n_env);
            else
                hipLaunchKernelGGL((reset_kernel<double, 1, true>), grid, block, 0, st, p.Pd, p.R, (double *)fstate, istate,
                                   ev_vg, ev_si, (const uint8_t *)nullptr, (double *)out, n_env);
        }
    }
    }
    return hip_rc(hipGetLastError());
}


// The step kernel launch_plan picks for this plan (same conditions, same order) as rocprofv3 prints it.
const char *plan_kernel_name(const pvder_plan &p) {
    static const bool no_w1 = []() { const char *v = getenv("PVDER_NO_W1"); return v && v[0] == '1'; }();
    static const int lane_form = []() { const char *a = getenv("PVDER_LANE1"), *b = getenv("PVDER_LANE3");
                                        return (a && a[0] == '1') ? 1 : ((b && b[0] == '1') ? 3 : 0); }();
    static const bool generic = []() { const char *v = getenv("PVDER_RK4_VARIANT"); return v && v[0] == 'g'; }();
    switch (p.integrator) {
    case PVDER_INT_ETDLIN:
        if (p.ph == 3 && p.lane_par) return "step_etdlin_lane_kernel";
        if (p.ph == 3 && p.n_env <= kOneWavePerSimdEnvs && !no_w1) return "step_etdlin_kernel_w1<3>";
        return p.ph == 3 ? "step_etdlin_kernel<3>" : "step_etdlin_kernel<1>";
    case PVDER_INT_ETDRK4:
        if (p.ph == 3 && p.lane_par)
            return (lane_form == 1 || (lane_form == 0 && p.n_env > kLane3MaxEnvs)) ? "step_etdrk4_lane_kernel" : "step_etdrk4_lane3_kernel";
        return p.ph == 3 ? "step_etdrk4_bal_kernel<3>" : "step_etdrk4_bal_kernel<1>";
    case PVDER_INT_RK4:
        if (p.ph == 3 && !generic) return "step_rk4_pk_kernel";
        return p.ph == 3 ? "step_rk4_kernel<float, 3>" : "step_rk4_kernel<float, 1>";
    case PVDER_INT_ETDLIN64: return "step_etdlin64_kernel";
    default: return p.ph == 3 ? "step_rk45_kernel<double, 3>" : "step_rk45_kernel<double, 1>";
    }
}

}  // namespace

extern "C" {

int pvder_version(void) { return PVDER_ABI_VERSION; }
int pvder_par_count(void) { return PVDER_PAR_COUNT; }
int pvder_block_size(void) { return kBlock; }

int pvder_reset_masked_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                           float *fstate, int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *mask,
                           float *out, int n_env, void *stream) {
    return reset_impl<float>(par, cfg, ar, fstate, istate, ev_vg, ev_si, mask, out, n_env, stream);
}

int pvder_reset_masked_lanes_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                                 float *fstate, int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *mask,
                                 float *out, int n_env, const float *lane_folded, void *stream) {
    if (lane_folded && par && (int)par[PVDER_PAR_PHASES] != 3) return PVDER_EINVAL;      // (lane parameters: three-phase model)
    return reset_impl<float>(par, cfg, ar, fstate, istate, ev_vg, ev_si, mask, out, n_env, stream, lane_folded);
}

int pvder_reset_masked_f64(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                           double *fstate, int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *mask,
                           double *out, int n_env, void *stream) {
    return reset_impl<double>(par, cfg, ar, fstate, istate, ev_vg, ev_si, mask, out, n_env, stream);
}

int pvder_complete_event_tables(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, const int32_t *istate,
                                float *ev_vg, float *ev_si, int n_env, void *stream) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (!ar || !istate || !ev_vg || !ev_si) return PVDER_EINVAL;
    ResetK R;
    rc = fold_reset(par, cfg, ar, R);
    if (rc) return rc;
    if (R.keep_events) return 0;                           // (explicit tables: nothing is generated, nothing to complete)
    const ModelK<float> P = fold_params<float>(par, true);
    hipLaunchKernelGGL(complete_events_kernel, dim3((n_env + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, R, (float)P.sinsol0,
                       istate, ev_vg, ev_si, n_env);
    return hip_rc(hipGetLastError());
}

int pvder_fetch_env(const void *out, const int32_t *istate, int n_env, int env, int fp64, void *host_out, int32_t *host_ist,
                    void *stream) {
    if (!out || !istate || !host_out || !host_ist || n_env < 1 || env < 0 || env >= n_env) return PVDER_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t es = fp64 ? sizeof(double) : sizeof(float);
    hipError_t e = hipMemcpy2DAsync(host_out, es, (const char *)out + (size_t)env * es, (size_t)n_env * es, es, PVDER_OROWS,
                                    hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(host_ist, sizeof(int32_t), istate + env, (size_t)n_env * sizeof(int32_t), sizeof(int32_t), 4,
                             hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return hip_rc(e);
}

int pvder_etdrk4_coef_count(void) { return kCoefTotal; }
int pvder_lin_set_floats(void) { return pvder::kLinSetFloats; }
int pvder_lin_max_sets(void) { return pvder::kLinMaxSets; }

int pvder_etdrk4_coefficients(const double *par, const pvder_step_cfg *cfg, float *coef_host) {
    if (!par || !cfg || !coef_host || cfg->micro < 1) return PVDER_EINVAL;
    make_etd(par, cfg->micro, cfg->micro_coarse > 0 ? cfg->micro_coarse : cfg->micro, cfg->graded, coef_host);
    return 0;
}

// ---- per-call entry points: fold, launch, forget
int pvder_step_etdlin_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                          const float *coef, const pvder_lin_coef *lin, float *fstate, int32_t *istate,
                          float *ev_vg, float *ev_si, const uint8_t *action, float *out, float *stats,
                          int zero_stats, int n_env, void *stream) {
    pvder_plan p;
    const int rc = fill_plan(p, PVDER_INT_ETDLIN, par, cfg, ar, coef, lin, n_env);
    return rc ? rc : launch_plan(p, fstate, istate, ev_vg, ev_si, action, out, nullptr, stats, zero_stats, stream);
}

int pvder_step_etdrk4_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                          const float *coef, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                          const uint8_t *action, float *out, float *stats, int zero_stats, int n_env,
                          void *stream) {
    pvder_plan p;
    const int rc = fill_plan(p, PVDER_INT_ETDRK4, par, cfg, ar, coef, nullptr, n_env);
    return rc ? rc : launch_plan(p, fstate, istate, ev_vg, ev_si, action, out, nullptr, stats, zero_stats, stream);
}

int pvder_step_rk4_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, float *fstate,
                       int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *action, float *out,
                       float *stats, int zero_stats, int n_env, void *stream) {
    pvder_plan p;
    const int rc = fill_plan(p, PVDER_INT_RK4, par, cfg, ar, nullptr, nullptr, n_env);
    return rc ? rc : launch_plan(p, fstate, istate, ev_vg, ev_si, action, out, nullptr, stats, zero_stats, stream);
}

int pvder_step_rk45_f64(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, double *fstate,
                        int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *action, double *out,
                        double *hstep, float *stats, int zero_stats, int n_env, void *stream) {
    if (!hstep) return PVDER_EINVAL;
    pvder_plan p;
    const int rc = fill_plan(p, PVDER_INT_RK45, par, cfg, ar, nullptr, nullptr, n_env);
    return rc ? rc : launch_plan(p, fstate, istate, ev_vg, ev_si, action, out, hstep, stats, zero_stats, stream);
}

// ---- plan handles: fold once, step many times
int pvder_plan_create(pvder_plan **plan, int integrator, const double *par, const pvder_step_cfg *cfg,
                      const pvder_autoreset *ar, const float *coef, const pvder_lin_coef *lin, int n_env) {
    if (!plan) return PVDER_EINVAL;
    *plan = nullptr;
    pvder_plan *p = new (std::nothrow) pvder_plan;
    if (!p) return -(int)hipErrorOutOfMemory;
    const int rc = fill_plan(*p, integrator, par, cfg, ar, coef, lin, n_env);
    if (rc) { delete p; return rc; }
    *plan = p;
    return 0;
}

void pvder_plan_destroy(pvder_plan *plan) { delete plan; }

int pvder_fold_lane_params(const double *par, const double *lane, int n_env, float *folded) {
    if (!par || !lane || !folded || n_env <= 0) return PVDER_EINVAL;
    static const int idx[PVDER_LP_ROWS - 1] = {PVDER_PAR_C, PVDER_PAR_KP_DC, PVDER_PAR_KI_DC, PVDER_PAR_KP_Q, PVDER_PAR_KI_Q, PVDER_PAR_NP};
    const size_t N = (size_t)n_env;
    for (int e = 0; e < n_env; ++e) {
        double q[PVDER_PAR_COUNT];
        for (int k = 0; k < PVDER_PAR_COUNT; ++k) q[k] = par[k];
        for (int r = 0; r < PVDER_LP_ROWS - 1; ++r) {
            const double v = lane[(size_t)r * N + e];
            if (v == v) q[idx[r]] = v;                       // (NaN: the group's value)
        }
        double sc = lane[(size_t)PVDER_LP_SINSOL_SCALE * N + e];
        if (!(sc == sc)) sc = 1.0;
        if (!(q[PVDER_PAR_C] > 0.0) || !(q[PVDER_PAR_NP] >= 0.0) || !(sc >= 0.0) || !(sc < 1e30)) return PVDER_EINVAL;
        for (int r = 0; r < PVDER_LP_ROWS - 1; ++r)
            if (!(fabs(q[idx[r]]) < 1e30)) return PVDER_EINVAL;
        const ModelK<float> P = fold_params<float>(q, true);
        folded[(size_t)PVDER_LPF_IC * N + e] = P.iC;
        folded[(size_t)PVDER_LPF_KP_DC * N + e] = P.KpDC; folded[(size_t)PVDER_LPF_KI_DC * N + e] = P.KiDC;
        folded[(size_t)PVDER_LPF_KP_Q * N + e] = P.KpQ;   folded[(size_t)PVDER_LPF_KI_Q * N + e] = P.KiQ;
        // the insolation scale multiplies the photo-current term only:  pv1 = pvA * Sinsol + pvB  (make_sub; with model option
        // U5 pvA carries -Vdcmpp0 times it, the same factor)
        folded[(size_t)PVDER_LPF_PVA * N + e] = (float)((double)P.pvA * sc);
        folded[(size_t)PVDER_LPF_PVB * N + e] = P.pvB;
        // the env's own steady state (steadyStateInitialization=True, PVDER_env.py:374-378; the host mirror of
        // gym_PVDER_amd/der_model.py::steady_state): Newton root in fp64 of  Re S = Ppv(Vdc_ref0),  Im S_PCC = 0  with
        // di/dt = 0 at w = wbase, i.e. vt = v + (Rf + j Lf) i,  v = kvg vg + kvi i
        const double ph = q[PVDER_PAR_PHASES], vdc = q[PVDER_PAR_VDC_REF0];
        const double vact = vdc * q[PVDER_PAR_VDCBASE];
        double ppv = q[PVDER_PAR_NP] * vact * (q[PVDER_PAR_ISCR] * (q[PVDER_PAR_SINSOL0] * sc) / 100.0
                                               - q[PVDER_PAR_IRS] * (exp(q[PVDER_PAR_PV_VT] * vdc) - 1.0)) / q[PVDER_PAR_SBASE];
        if (ppv < 0.0) ppv = 0.0;
        const double gR = q[PVDER_PAR_KVG_RE] * q[PVDER_PAR_VG_SCALE], gI = q[PVDER_PAR_KVG_IM] * q[PVDER_PAR_VG_SCALE];
        const double kR = q[PVDER_PAR_KVI_RE], kI = q[PVDER_PAR_KVI_IM], zR = kR + q[PVDER_PAR_RF], zI = kI + q[PVDER_PAR_LF];
        // Re S   = ph/2 Re[(g + z i) conj(i)] = ph/2 (gR iR + gI iI + zR |i|^2)
        // Im Spcc = ph/2 Im[(g + k i) conj(i)] = ph/2 (gI iR - gR iI + kI |i|^2)
        double iR = 2.0 * ppv / (ph * gR), iI = 0.0;
        int it = 0;
        for (; it < 60; ++it) {
            const double i2 = iR * iR + iI * iI;
            const double r0 = 0.5 * ph * (gR * iR + gI * iI + zR * i2) - ppv, r1 = 0.5 * ph * (gI * iR - gR * iI + kI * i2);
            const double a = 0.5 * ph * (gR + 2.0 * zR * iR), b = 0.5 * ph * (gI + 2.0 * zR * iI);
            const double c = 0.5 * ph * (gI + 2.0 * kI * iR), d = 0.5 * ph * (-gR + 2.0 * kI * iI);
            const double det = a * d - b * c;
            if (!(fabs(det) > 1e-300)) return PVDER_EINVAL;
            const double dR = (d * r0 - b * r1) / det, dI = (a * r1 - c * r0) / det;
            iR -= dR; iI -= dI;
            if (fabs(dR) + fabs(dI) < 1e-15) break;
        }
        if (it == 60 || !(fabs(iR) + fabs(iI) < 1e6)) return PVDER_EINVAL;
        const double mR = (gR + zR * iR - zI * iI) / (0.5 * vdc), mI = (gI + zR * iI + zI * iR) / (0.5 * vdc);
        folded[(size_t)PVDER_LPF_IA_RE * N + e] = (float)iR; folded[(size_t)PVDER_LPF_IA_IM * N + e] = (float)iI;
        folded[(size_t)PVDER_LPF_MA_RE * N + e] = (float)mR; folded[(size_t)PVDER_LPF_MA_IM * N + e] = (float)mI;
    }
    return 0;
}

int pvder_plan_lane_params(pvder_plan *plan, const float *lane_folded) {
    if (!plan) return PVDER_EINVAL;
    if (lane_folded && plan->ph != 3) return PVDER_EINVAL;
    if (lane_folded && plan->integrator == PVDER_INT_ETDLIN && !plan->k4lane) return PVDER_EINVAL;
    if (lane_folded && plan->integrator != PVDER_INT_ETDRK4 && plan->integrator != PVDER_INT_ETDLIN) return PVDER_EINVAL;
    plan->lane_par = lane_folded;
    return 0;
}


int pvder_plan_hint(pvder_plan *plan, int ksub) {
    if (!plan) return -1;
    if (ksub >= -1) plan->ksub_hint = ksub;
    return plan->ksub_hint;
}

int pvder_plan_graph_hints(pvder_plan *plan, int on) {
    if (!plan) return PVDER_EINVAL;
    const int was = plan->graph_hints ? 1 : 0;
    if (on >= 0) plan->graph_hints = on != 0;
    return was;
}

const char *pvder_plan_kernel_name(const pvder_plan *plan) { return plan ? plan_kernel_name(*plan) : ""; }

int pvder_plan_hybrid_active(const pvder_plan *plan) {
    if (!plan || plan->ph != 3 || plan->n_sub > kHybMaxSub) return 0;
    if (plan->hyb_mode == 3) return 1;
    if (plan->hyb_mode != 0) return 0;
    return (plan->hyb_hold > 0 || (plan->hyb_flag_host && __atomic_load_n(plan->hyb_flag_host, __ATOMIC_RELAXED) != 0)) ? 1 : 0;
}

int pvder_plan_step(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                    const uint8_t *action, void *out, double *hstep, float *stats, void *stream) {
    if (!plan) return PVDER_EINVAL;
    return launch_plan(*plan, fstate, istate, ev_vg, ev_si, action, out, hstep, stats, 1, stream);
}

int pvder_plan_step_fetch(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                          const uint8_t *action, void *out, double *hstep, float *stats, void *snapshot, const void *snapshot_src,
                          size_t snapshot_bytes, void *host_out, int32_t *host_ist, void *stream) {
    if (!plan || plan->n_env != 1) return PVDER_EINVAL;
    int rc = launch_plan(*plan, fstate, istate, ev_vg, ev_si, action, out, hstep, stats, 1, stream);
    if (rc) return rc;
    if (snapshot) {
        if (!snapshot_src) return PVDER_EINVAL;
        const hipError_t e = hipMemcpyAsync(snapshot, snapshot_src, snapshot_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) return hip_rc(e);
    }
    return pvder_fetch_env(out, istate, 1, 0, plan->integrator >= PVDER_INT_RK45 ? 1 : 0, host_out, host_ist, stream);
}

int pvder_plan_step_mirror(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                           const uint8_t *action, void *out, double *hstep, float *stats, void *snapshot, const void *snapshot_src,
                           size_t snapshot_bytes, double *host_block, uint64_t seq, void *stream) {
    if (!plan || plan->n_env != 1 || !host_block || (snapshot && !snapshot_src)) return PVDER_EINVAL;
    volatile uint64_t *word = (volatile uint64_t *)(host_block + PVDER_OROWS + 4);
    if (*word == seq) return PVDER_EINVAL;
    void *dev_block = nullptr;
    if (hipHostGetDevicePointer(&dev_block, host_block, 0) != hipSuccess || !dev_block) { (void)hipGetLastError(); return PVDER_EINVAL; }
    int rc = launch_plan(*plan, fstate, istate, ev_vg, ev_si, action, out, hstep, stats, 1, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (plan->integrator >= PVDER_INT_RK45)
        hipLaunchKernelGGL(mirror_env_kernel<double>, dim3(1), dim3(64), 0, st, (const double *)out, istate, 1, 0, (double *)snapshot,
                           (const double *)snapshot_src, (int)(snapshot ? snapshot_bytes / sizeof(double) : 0), (double *)dev_block,
                           (unsigned long long)seq);
    else
        hipLaunchKernelGGL(mirror_env_kernel<float>, dim3(1), dim3(64), 0, st, (const float *)out, istate, 1, 0, (float *)snapshot,
                           (const float *)snapshot_src, (int)(snapshot ? snapshot_bytes / sizeof(float) : 0), (double *)dev_block,
                           (unsigned long long)seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_rc(e);
    // the word arrives over the bus ~1 us behind the kernel's store: a spin on it instead of the completion signal's interrupt
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n((const uint64_t *)word, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 0x3ffu) == 0x3ffu) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double ms = (t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
            if (ms > (double)PVDER_MIRROR_TIMEOUT_MS) break;
            if (ms > 0.2) {                    // a slow (fp64) step: stop burning the core, look again every 20 us
                const timespec nap = {0, 20000};
                nanosleep(&nap, nullptr);
            }
        }
    }
    e = hipStreamSynchronize(st);             // (a fault in the step kernel surfaces here)
    if (e != hipSuccess) return hip_rc(e);
    return __atomic_load_n((const uint64_t *)word, __ATOMIC_ACQUIRE) == seq ? 0 : PVDER_ETIMEOUT;
}

int pvder_plan_rollout(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                       const uint8_t *actions, int n_steps, void *out, size_t out_step_stride, double *hstep,
                       float *stats, void *stream) {
    if (!plan || n_steps < 0 || !actions || !out) return PVDER_EINVAL;
    const size_t esz = plan->integrator >= PVDER_INT_RK45 ? sizeof(double) : sizeof(float);
    for (int k = 0; k < n_steps; ++k) {
        char *blk = (char *)out + (size_t)k * out_step_stride * esz;
        const int rc = launch_plan(*plan, fstate, istate, ev_vg, ev_si, actions + (size_t)k * plan->n_env, blk, hstep, stats, 1,
                                   stream, k > 0 ? blk - out_step_stride * esz : nullptr, plan->graph_hints);
        if (rc) return rc;
    }
    return 0;
}

// Open-loop rollout with the env steps FUSED into one launch (etdlin, three-phase): same results as
// pvder_plan_rollout, bit for bit.  The fused kernel (state in registers from step to step) runs until a
// workgroup meets a step it does not carry -- the one that ends an episode, an env off balance / near a
// limit --, the finish kernel takes those workgroups through that ONE step (per-step body, through
// memory), the fused kernel resumes; one such pair per episode end the rollout can contain, and a last
// finish pass for whatever is left.
int pvder_plan_rollout_fused(const pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                             const uint8_t *actions, int n_steps, float *out, size_t out_step_stride, float *stats,
                             int32_t *progress, void *stream) {
    if (!plan || n_steps < 0 || !actions || !out || !progress || !fstate || !istate || !ev_vg || !ev_si) return PVDER_EINVAL;
    if (plan->integrator != PVDER_INT_ETDLIN || plan->ph != 3 || plan->lane_par) return PVDER_EINVAL;     // (per-lane sets: pvder_plan_rollout)
    if (n_steps == 0) return 0;
    const int n_env = plan->n_env;
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    RollArgs Q;
    Q.G = plan->lin;
    EtdArgs &A = Q.G.A;
    RollK &Rk = Q.Rk;
    A.fstate = fstate; A.istate = istate; A.ev_vg = ev_vg; A.ev_si = ev_si;
    A.action = actions; A.out = out; A.stats = stats; A.stats_fresh = 1;
    Rk.actions = actions; Rk.n_steps = n_steps; Rk.out_stride = (long long)out_step_stride; Rk.progress = progress;
    Rk.ag = pvder_agent::AgentK{}; Rk.q = nullptr; Rk.ag_state = nullptr; Rk.ag_action = nullptr; Rk.ag_alive = nullptr; Rk.ag_ret = nullptr;
    Rk.mailbox = nullptr; Rk.done = nullptr; Rk.sv_dev = nullptr; Rk.sv_paced = 0;
    if (A.R.lazy) {
        // pvder_autoreset.lazy_events: the fused loop reads its event rows one env step AHEAD of the step that would leave them
        // behind -- it runs on whole tables (completed here, one launch per rollout), and so do the auto-resets of its finish kernel
        hipLaunchKernelGGL(complete_events_kernel, grid, block, 0, st, A.R, (float)A.P.sinsol0, istate, ev_vg, ev_si, n_env);
        A.R.lazy = 0;
    }
    // (fused kernel, finish kernel for ONE step) once per episode end a workgroup can meet inside the rollout,
    // then the finish kernel for whatever is left (workgroups with deferred / finished envs)
    const int ep = plan->episode_steps > 0 ? plan->episode_steps : 1;
    const int pairs = n_steps / ep + 2;
    for (int r = 0; r < pairs; ++r) {
        Rk.resume = r > 0 ? 1 : 0;
        Rk.max_finish = r + 1 < pairs ? 1 : n_steps;
        static const bool no_cp = []() { const char *v = getenv("PVDER_NO_CP"); return v && v[0] == '1'; }();      // (A/B)
        if (n_env <= kOneWavePerSimdEnvs && !no_cp) hipLaunchKernelGGL(step_etdlin_rollout_cp_kernel, grid, block, 0, st, Q);
        else hipLaunchKernelGGL(step_etdlin_rollout_kernel<3>, grid, block, 0, st, Q);
        hipLaunchKernelGGL(step_etdlin_finish_kernel<3>, grid, block, 0, st, Q);
    }
    advance_hint(*plan, n_steps);
    return hip_rc(hipGetLastError());
}

// SERVED rollout (round 6; profiles/r06_ab_served_rollout.txt): pvder_plan_rollout_fused whose env steps wait for their action rows.
// The caller posts "rows 0 .. p-1 are there" into a host-coherent word; ONE thread of the launch (the relay workgroup) polls it over
// PCIe and fans it out to a device word per workgroup; a workgroup reports the steps whose outputs have left its CU to a device word
// the relay reduces to (min, max) for the host.  4.9 us per env step at 65 536 envs with the rows posted ahead (one launch per step:
// 8.5), bit-identical; a producer that posts row k + 1 only when step k - D is reported done pays the PCIe round trip: 5.2 us at
// D = 8, 27 us at D = 1.  Every wait is bounded (0.2 s): a caller that stops posting gets its launch back, with `progress` short.
// host-coherent words: [0] = the mailbox, [16] = min, [17] = max over the workgroups of the steps done; *host and *dev name the same memory; *scratch: 16 * (4 + 2 n_wg) device words
int pvder_served_alloc(int n_env, int **host, int **dev, int **scratch) {
    if (n_env < 1 || !host || !dev || !scratch) return PVDER_EINVAL;
    const int n_wg = (n_env + kBlock - 1) / kBlock;
    int *h = nullptr;
    void *d = nullptr;
    int *s = nullptr;
    hipError_t e = hipMalloc((void **)&s, 64 * (size_t)(4 + 2 * n_wg));
    if (e == hipSuccess) e = hipMemset(s, 0, 64 * (size_t)(4 + 2 * n_wg));
    if (e == hipSuccess) e = hipHostMalloc((void **)&h, sizeof(int) * 32, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {           // (nothing is handed out, nothing is kept)
        if (h) (void)hipHostFree(h);
        if (s) (void)hipFree(s);
        return hip_rc(e);
    }
    for (int i = 0; i < 32; ++i) h[i] = 0;
    *host = h; *dev = (int *)d; *scratch = s;
    return 0;
}
int pvder_served_free(int *host, int *scratch) {
    const hipError_t a = host ? hipHostFree(host) : hipSuccess, b = scratch ? hipFree(scratch) : hipSuccess;
    return hip_rc(a != hipSuccess ? a : b);
}

// pvder_plan_rollout_fused with its steps gated by the mailbox (same launches: served kernel + finish kernel once per episode end
// the rollout can contain; the finish kernel is not gated -- the step it takes is one whose action row the served kernel had)
int pvder_plan_rollout_served(const pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                              const uint8_t *actions, int n_steps, float *out, size_t out_step_stride, float *stats,
                              int32_t *progress, int *words_dev, int *scratch_dev, int paced, int resume, void *stream) {
    const int *mailbox_dev = words_dev;
    int *done_dev = words_dev ? words_dev + 16 : nullptr;
    if (!plan || n_steps < 0 || !actions || !out || !progress || !fstate || !istate || !ev_vg || !ev_si || !mailbox_dev || !done_dev || !scratch_dev) return PVDER_EINVAL;
    if (plan->integrator != PVDER_INT_ETDLIN || plan->ph != 3 || plan->lane_par || (resume && !paced)) return PVDER_EINVAL;
    if (plan->n_env > kOneWavePerSimdEnvs) return PVDER_EINVAL;       // (every workgroup must be RESIDENT while it waits: one wave per SIMD)
    if (n_steps == 0) return 0;
    const int n_env = plan->n_env;
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    RollArgs Q;
    Q.G = plan->lin;
    EtdArgs &A = Q.G.A;
    RollK &Rk = Q.Rk;
    A.fstate = fstate; A.istate = istate; A.ev_vg = ev_vg; A.ev_si = ev_si;
    A.action = actions; A.out = out; A.stats = stats; A.stats_fresh = 1;
    Rk.actions = actions; Rk.n_steps = n_steps; Rk.out_stride = (long long)out_step_stride; Rk.progress = progress;
    Rk.ag = pvder_agent::AgentK{}; Rk.q = nullptr; Rk.ag_state = nullptr; Rk.ag_action = nullptr; Rk.ag_alive = nullptr; Rk.ag_ret = nullptr;
    Rk.mailbox = mailbox_dev; Rk.done = done_dev; Rk.sv_dev = scratch_dev; Rk.sv_paced = paced ? 1 : 0;
    const dim3 grid_sv(grid.x + 1);                    // + the relay workgroup
    if (!resume) {
        if (pvder_plan_hybrid_active(plan)) return PVDER_EINVAL;        // (an env at the duty-cycle limit: its steps belong to the follow-up kernel -- pvder_plan_rollout)
        if (hipMemsetAsync(scratch_dev, 0, 64 * (size_t)(4 + 2 * grid.x), st) != hipSuccess) return hip_rc(hipErrorUnknown);
        if (A.R.lazy)
            hipLaunchKernelGGL(complete_events_kernel, grid, block, 0, st, A.R, (float)A.P.sinsol0, istate, ev_vg, ev_si, n_env);
    }
    A.R.lazy = 0;                                      // (whole tables: see pvder_plan_rollout_fused)
    if (paced) {
        // ONE pair: the served loop until a workgroup meets a step it cannot carry (then every workgroup leaves: kServedStop), the
        // finish kernel for that one step of every workgroup that has steps left.  The caller looks at `progress` and calls again.
        Rk.resume = resume ? 1 : 0;
        Rk.max_finish = 1;
        hipLaunchKernelGGL(step_etdlin_served_kernel, grid_sv, block, 0, st, Q);
        hipLaunchKernelGGL(step_etdlin_finish_kernel<3>, grid, block, 0, st, Q);
    } else {
        // every row is posted already (the caller's word says n_steps): the pairs of pvder_plan_rollout_fused, nobody asks anybody to
        // leave, and the last finish pass -- which does not wait for rows -- takes what is left
        const int ep = plan->episode_steps > 0 ? plan->episode_steps : 1;
        const int pairs = n_steps / ep + 2;
        for (int r = 0; r < pairs; ++r) {
            Rk.resume = r > 0 ? 1 : 0;
            Rk.max_finish = r + 1 < pairs ? 1 : n_steps;
            hipLaunchKernelGGL(step_etdlin_served_kernel, grid_sv, block, 0, st, Q);
            hipLaunchKernelGGL(step_etdlin_finish_kernel<3>, grid, block, 0, st, Q);
        }
    }
    if (!resume) advance_hint(*plan, n_steps);
    return hip_rc(hipGetLastError());
}

// Closed-loop training episode in a handful of launches: n_steps x (env step, agent step) with the env state AND the
// agent's registers on the chip from step to step (step_etdlin_rollout_agent_kernel); the steps a workgroup cannot carry
// -- the one that ends the episode, an env off balance / at a limit -- are taken, env step and agent step, by the
// finish kernel.  Same results as n_steps x (pvder_plan_step, pvder_qagent_step_f32) wherever that sequence is itself
// deterministic (envs that do not race for the same table entries; one wave of envs always).
int pvder_plan_rollout_fused_agent(const pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                                   int n_steps, float *out, float *stats, int32_t *progress, const pvder_qagent_cfg *cfg,
                                   float *q_table, int64_t *state, uint8_t *action, uint8_t *alive, double *ep_reward,
                                   void *stream) {
    if (!plan || n_steps < 0 || !out || !progress || !fstate || !istate || !ev_vg || !ev_si) return PVDER_EINVAL;
    if (!cfg || !q_table || !state || !action || !alive || !ep_reward) return PVDER_EINVAL;
    if (plan->integrator != PVDER_INT_ETDLIN || plan->ph != 3 || plan->R.enabled || plan->lane_par) return PVDER_EINVAL;   // (episodes are the caller's)
    if (cfg->n_act < 1 || cfg->n_act > 5 || cfg->first || !cfg->choose) return PVDER_EINVAL;
    if (cfg->policy == PVDER_POLICY_QNET) {
        if (cfg->hidden < 1 || cfg->hidden > pvder_agent::kQnetMaxHidden) return PVDER_EINVAL;
    } else if (cfg->policy != PVDER_POLICY_TABLE || cfg->bins < 1 || !(cfg->win > 0.0)) {
        return PVDER_EINVAL;
    }
    if (n_steps == 0) return 0;
    const int n_env = plan->n_env;
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    RollArgs Q;
    Q.G = plan->lin;
    EtdArgs &A = Q.G.A;
    RollK &Rk = Q.Rk;
    A.fstate = fstate; A.istate = istate; A.ev_vg = ev_vg; A.ev_si = ev_si;
    A.action = action; A.out = out; A.stats = stats; A.stats_fresh = 1;
    Rk.actions = nullptr; Rk.n_steps = n_steps; Rk.out_stride = 0; Rk.progress = progress;
    pvder_agent::AgentK &K = Rk.ag;
    K.bins = cfg->bins; K.n_act = cfg->n_act; K.first = 0; K.choose = 1;
    K.one_minus_lr = (float)(1.0 - (double)cfg->lr); K.lr = cfg->lr; K.gamma = cfg->gamma;
    K.epsilon = cfg->epsilon; K.q_done = cfg->q_done; K.low = cfg->low; K.win = cfg->win;
    K.seed = cfg->seed; K.counter = cfg->counter; K.env_id_base = cfg->env_id_base;
    K.policy = cfg->policy; K.hidden = cfg->hidden;
    Rk.q = q_table; Rk.ag_state = (long long *)state; Rk.ag_action = action; Rk.ag_alive = alive; Rk.ag_ret = ep_reward;
    Rk.mailbox = nullptr; Rk.done = nullptr; Rk.sv_dev = nullptr; Rk.sv_paced = 0;
    const int ep = plan->episode_steps > 0 ? plan->episode_steps : 1;
    const int pairs = n_steps / ep + 2;
    for (int r = 0; r < pairs; ++r) {
        Rk.resume = r > 0 ? 1 : 0;
        Rk.max_finish = r + 1 < pairs ? 1 : n_steps;
        hipLaunchKernelGGL(step_etdlin_rollout_agent_kernel<3>, grid, block, 0, st, Q);
        hipLaunchKernelGGL((step_etdlin_finish_kernel<3, true>), grid, block, 0, st, Q);
    }
    advance_hint(*plan, n_steps);
    return hip_rc(hipGetLastError());
}

// ---- the per-step collective of the sharded env, enqueued from C ------------------------------------
// One all-gather of the [13, N_local] output block per env step (RCCL over xGMI): the step kernel runs
// on the caller's stream, the gather on this object's own stream, two output blocks alternate so that
// the gather of step t overlaps the kernel of step t+1.  Everything a step needs -- wait for the gather
// that last read the block, launch, record, cross-stream wait, ncclAllGather, record -- is issued from
// ONE host call per rollout instead of six Python-level calls per step.  RCCL is not linked: the
// caller passes the address of ncclAllGather from the librccl it already uses.
typedef int (*nccl_all_gather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
struct pvder_gather {
    nccl_all_gather_fn all_gather;
    void *comm;
    hipStream_t stream;
    hipEvent_t ready[2], done[2];
    bool pending[2];
    float *stage[2];         // "obs8" payload: the ten rows a step sends, packed (allocated at the first such call)
    int stage_envs;
};

// the "obs8" payload (PVDER_GATHER_OBS8): rows 0-7 (the dynamic observations) and 11-12 (reward, done) of an output block side
// by side -- rows 8-10 (Vdc_ref, Q_ref, t / max_sim_time) follow from the actions and the step count, which whoever chose the
// actions holds already (PVDER_env.py:198-229, :540): 10 of 13 rows on the links
__global__ __launch_bounds__(kBlock) void pack_obs8_kernel(const float *__restrict__ out, float *__restrict__ stage, int n_env) {
    const size_t N = (size_t)n_env;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)PVDER_GATHER_OBS8_ROWS * N) return;
    const size_t r = i / N, e = i - r * N;
    stage[i] = out[(r < 8 ? r : r + 3) * N + e];
}

int pvder_gather_create(pvder_gather **out, void *nccl_all_gather, void *comm) {
    if (!out || !nccl_all_gather || !comm) return PVDER_EINVAL;
    *out = nullptr;
    pvder_gather *g = new (std::nothrow) pvder_gather;
    if (!g) return -(int)hipErrorOutOfMemory;
    g->all_gather = (nccl_all_gather_fn)nccl_all_gather;
    g->comm = comm;
    g->pending[0] = g->pending[1] = false;
    g->stage[0] = g->stage[1] = nullptr;
    g->stage_envs = 0;
    g->stream = nullptr;
    g->ready[0] = g->ready[1] = g->done[0] = g->done[1] = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipEventCreateWithFlags(&g->ready[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) {                       // partial initialisation: release what was created
        for (int b = 0; b < 2; ++b) {
            if (g->ready[b]) (void)hipEventDestroy(g->ready[b]);
            if (g->done[b]) (void)hipEventDestroy(g->done[b]);
        }
        if (g->stream) (void)hipStreamDestroy(g->stream);
        delete g;
        return hip_rc(e);
    }
    *out = g;
    return 0;
}

void pvder_gather_destroy(pvder_gather *g) {
    if (!g) return;
    (void)hipStreamSynchronize(g->stream);
    for (int b = 0; b < 2; ++b) { (void)hipEventDestroy(g->ready[b]); (void)hipEventDestroy(g->done[b]); }
    (void)hipStreamDestroy(g->stream);
    for (int b = 0; b < 2; ++b) if (g->stage[b]) (void)hipFree(g->stage[b]);
    delete g;
}

// make `stream` wait for the gathers still in flight (their results are then visible to work enqueued on it)
int pvder_gather_drain(pvder_gather *g, void *stream) {
    if (!g) return PVDER_EINVAL;
    for (int b = 0; b < 2; ++b)
        if (g->pending[b]) {
            const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, g->done[b], 0);
            if (e != hipSuccess) return hip_rc(e);
            g->pending[b] = false;
        }
    return 0;
}

// n_steps transitions, each followed by the all-gather of its output block: step k writes out[k & 1]
// ([13, n_env] floats) and gathers it into recv[k & 1] ([world * 13, n_env]).  sync_each != 0: the next
// step waits for the gather (no overlap).  first_block: parity of the first step (continue a rollout).
int pvder_plan_rollout_gather(const pvder_plan *plan, pvder_gather *g, void *fstate, int32_t *istate, float *ev_vg,
                              float *ev_si, const uint8_t *actions, int n_steps, float *out0, float *out1, float *recv0,
                              float *recv1, float *stats, void *stream, int first_block, int sync_each) {
    return pvder_plan_rollout_gather_rows(plan, g, fstate, istate, ev_vg, ev_si, actions, n_steps, out0, out1, recv0, recv1,
                                          stats, stream, first_block, sync_each, 0, PVDER_OROWS, 1, 0);
}

// The same with a PAYLOAD choice: step k collates rows [row0, row0 + n_rows) of its output block -- (reward, done) =
// rows 11, 12: 0.5 MB per rank and step at 65 536 envs instead of 3.4 -- except every `full_every`-th step (counted
// from `step_index0`), which collates the whole block (full_every <= 1: the given rows always).  A partial gather
// fills the first world * n_rows rows of its recv block.
int pvder_plan_rollout_gather_rows(const pvder_plan *plan, pvder_gather *g, void *fstate, int32_t *istate, float *ev_vg,
                                   float *ev_si, const uint8_t *actions, int n_steps, float *out0, float *out1, float *recv0,
                                   float *recv1, float *stats, void *stream, int first_block, int sync_each, int row0,
                                   int n_rows, int full_every, long long step_index0) {
    if (!plan || !g || n_steps < 0 || !actions || !out0 || !out1 || !recv0 || !recv1) return PVDER_EINVAL;
    if (plan->integrator >= PVDER_INT_RK45) return PVDER_EINVAL;     // (fp32 blocks only)
    const bool obs8 = row0 == PVDER_GATHER_OBS8;
    if (!obs8 && (row0 < 0 || n_rows < 1 || row0 + n_rows > PVDER_OROWS)) return PVDER_EINVAL;
    if (obs8 && g->stage_envs != plan->n_env) {        // (first obs8 call, or another batch: the staging blocks; not under capture)
        for (int b = 0; b < 2; ++b) {
            if (g->stage[b]) (void)hipFree(g->stage[b]);
            g->stage[b] = nullptr;
            const hipError_t e = hipMalloc((void **)&g->stage[b], sizeof(float) * PVDER_GATHER_OBS8_ROWS * (size_t)plan->n_env);
            if (e != hipSuccess) { g->stage_envs = 0; return hip_rc(e); }
        }
        g->stage_envs = plan->n_env;
    }
    hipStream_t st = (hipStream_t)stream;
    float *outs[2] = {out0, out1}, *recvs[2] = {recv0, recv1};
    for (int k = 0; k < n_steps; ++k) {
        const int b = (first_block + k) & 1;
        hipError_t e = hipSuccess;
        if (g->pending[b]) e = hipStreamWaitEvent(st, g->done[b], 0);        // gather t-2 has read this block
        if (e != hipSuccess) return hip_rc(e);
        const int rc = launch_plan(*plan, fstate, istate, ev_vg, ev_si, actions + (size_t)k * plan->n_env, outs[b],
                                   nullptr, stats, 1, stream, outs[b ^ 1]);
        if (rc) return rc;
        e = hipEventRecord(g->ready[b], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(g->stream, g->ready[b], 0);
        if (e != hipSuccess) return hip_rc(e);
        const bool full = full_every > 1 && (step_index0 + k + 1) % full_every == 0;
        const int r0 = full ? 0 : row0, nr = full ? PVDER_OROWS : (obs8 ? PVDER_GATHER_OBS8_ROWS : n_rows);
        const size_t count = (size_t)nr * plan->n_env;
        const float *send = outs[b] + (size_t)(obs8 && !full ? 0 : r0) * plan->n_env;
        if (obs8 && !full) {                          // pack on the gather's own stream: off the step kernels' critical path
            hipLaunchKernelGGL(pack_obs8_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, g->stream,
                               outs[b], g->stage[b], plan->n_env);
            send = g->stage[b];
        }
        const int nrc = g->all_gather(send, recvs[b], count, 7 /* ncclFloat32 */, g->comm, g->stream);
        if (nrc != 0) return -20000 - nrc;
        e = hipEventRecord(g->done[b], g->stream);
        if (e != hipSuccess) return hip_rc(e);
        g->pending[b] = true;
        if (sync_each) {
            e = hipStreamWaitEvent(st, g->done[b], 0);
            if (e != hipSuccess) return hip_rc(e);
        }
    }
    return 0;
}

#ifdef PVDER_DEBUG_COUNTERS
int pvder_debug_counters(unsigned long long *host, int clear) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && clear) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z));
    }
    return -(int)e;
}
#endif
#ifdef PVDER_TRACE
int pvder_debug_trace(unsigned long long *host, int n_words) {
    return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * (size_t)n_words);
}
#endif

}  // extern "C"
