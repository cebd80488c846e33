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
#include <complex>
#include <new>

#include "pvder_hip.h"
#include "pvder_model.hpp"
#include "pvder_model_pk.hpp"
#include "pvder_etdlin.hpp"
#include "pvder_sat_hybrid.hpp"
#include "pvder_agent.h"
#include "pvder_agent_dev.hpp"

namespace {
// Debug-only in-kernel timeline (tools/trace_timeline.py builds a separate library with
// -DPVDER_TRACE; the shipped library compiles these to nothing).  Slot = 100 MHz wall clock of wave 0.
#ifdef PVDER_DEBUG_COUNTERS
// debug build only (tools/diag_counters.py): why envs leave the standard path -- lanes counted per reason
__device__ unsigned long long g_dbg[16];
#define PVDER_DBG_COUNT(slot, cond) do { if (cond) atomicAdd(&g_dbg[slot], 1ull); } while (0)
#else
#define PVDER_DBG_COUNT(slot, cond) do { } while (0)
#endif
#ifdef PVDER_TRACE
__device__ unsigned long long g_trace[4096 * 32];
#define PVDER_TRACE_AT(slot)                                                                   \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 32 + (slot)] = wall_clock64(); \
    } while (0)
// ... after `val` (a float) has been computed
#define PVDER_TRACE_DEP(slot, val)                                                             \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                             \
            g_trace[blockIdx.x * 32 + (slot)] = wall_clock64() + ((val) > 1e30f ? 1 : 0);      \
    } while (0)
#else
#define PVDER_TRACE_AT(slot) do { } while (0)
#define PVDER_TRACE_DEP(slot, val) do { } while (0)
#endif
// (-DPVDER_TRACE -DPVDER_TRACE_COLD: the stamps of the second pass of step_etdlin_kernel -- lin_step_body behind the lean
//  pass -- instead of the lean pass's own: slots 8 entry | 9 standard pass done | 10 careful pass done | 11 epilogue done |
//  12 23-state tail done; tools/ramp_timeline.py)
#if defined(PVDER_TRACE) && defined(PVDER_TRACE_COLD)
#define PVDER_TRACE_COLD_AT(slot) PVDER_TRACE_AT(slot)
// ... the shader clock (cycles) at the same point: slot 26 second pass entry | 27 careful pass done (tools/careful_timeline.py)
#define PVDER_TRACE_COLD_CYC(slot)                                                              \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 32 + (slot)] = clock64(); \
    } while (0)
// ... the LATEST wave of the workgroup to get there (slots 13 tail: operator sets staged | 14 state loaded | 15 + s: sim time step s done)
#define PVDER_TRACE_COLD_MAX(slot, cond)                                                                          \
    do {                                                                                                           \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && (cond)) atomicMax(&g_trace[blockIdx.x * 32 + (slot)], (unsigned long long)wall_clock64()); \
    } while (0)
#else
#define PVDER_TRACE_COLD_AT(slot) do { } while (0)
#define PVDER_TRACE_COLD_CYC(slot) do { } while (0)
#define PVDER_TRACE_COLD_MAX(slot, cond) do { } while (0)
#endif

// A/B switches of the kernarg laundering (tools/ab_variant.sh): -DPVDER_NO_LAUNDER_EPI / _ROLL let hipcc
// keep the kernel arguments live instead of re-reading them
#ifdef PVDER_NO_LAUNDER_EPI
#define PVDER_LAUNDER_EPI(p) do { } while (0)
#else
#define PVDER_LAUNDER_EPI(p) asm volatile("" : "+s"(p))
#endif
#ifdef PVDER_NO_LAUNDER_ROLL
#define PVDER_LAUNDER_ROLL(p) do { } while (0)
#else
#define PVDER_LAUNDER_ROLL(p) asm volatile("" : "+s"(p))
#endif

typedef float v4 __attribute__((ext_vector_type(4)));

using pvder::ModelK;
using pvder::SubK;
using pvder::v2;

#ifndef PVDER_BLOCK
#define PVDER_BLOCK 256
#endif
constexpr int kBlock = PVDER_BLOCK;   // 4 waves
#ifdef PVDER_NO_HYB_FLAG            // (A/B: the lean pass without the hybrid flag's term)
#define PVDER_HYB_LEAN(x)
#else
#define PVDER_HYB_LEAN(x) x
#endif: one per SIMD of a CU
constexpr int kDeferred = -1; // transient istate status (etdlin64 only): "advance me in the follow-up launch" (step_rk45_kernel)

template <typename T>
struct StepK {
    int n_sub, micro, micro2, graded, max_ksub, max_steps, goal, discrete, ev_s0, ev_ds, ev_count, max_it, extra;
    int off_range;          // pvder_step_cfg.off_range: 1 = an env that needs the RK4 fallback ends its episode instead
    T dq, dv, inv_tmax, p_t, q_t, rtol, atol;
};

// in-kernel (auto-)reset arguments (by value; y0 in double so both precisions share it)
struct ResetK {
    int enabled, keep_events;
    uint64_t seed;
    int64_t env_id_base;
    const int32_t *env_ids;      // device, or NULL: the env a lane carries (pvder_autoreset.env_ids: a batch whose envs were re-ordered)
    int lazy;                    // pvder_autoreset.lazy_events: an auto-reset writes the rows its first step reads, every step the rows its successor reads
    pvder_event_spec spec;
    double y0[23];
};

// ---- counter-based event generator: bit-identical to oracle/pvder_model.py::counter_uniform
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float counter_uniform(uint64_t seed, uint64_t env, uint64_t episode, uint32_t k,
                                                 uint32_t lane) {
    uint64_t key = splitmix64(seed);
    key = splitmix64(key ^ (env & 0xFFFFFFFFull) ^ ((episode & 0xFFFFFFFFull) << 32));
    key = splitmix64(key ^ ((uint64_t)k << 8) ^ (uint64_t)(lane & 0xFF));
    return (float)(key >> 40) * (1.0f / 16777216.0f);
}

// lo + (hi-lo)*u with NO fma contraction: event values must equal the oracle's fp32 arithmetic
// bit for bit (random.uniform(min, max) of create_random_events)
__device__ __forceinline__ float lerp_unfused(float lo, float hi, float u) {
#pragma clang fp contract(off)
    const float d = hi - lo;
    const float p = d * u;
    return lo + p;
}

// create_random_events (PVDER_env.py:400-411): one event per grid time, type drawn from the ENABLEd
// list (dict order: insolation, voltage -- :60-61), value U[min,max]; rows k = 0..K of env e
// rows j0+1 .. j1 of env e from the values of row j0 (an event changes ONE of the two inputs: the other is carried)
__device__ __forceinline__ void fill_event_rows(const pvder_event_spec &spec, uint64_t seed, uint64_t gid, uint64_t episode,
                                                float *__restrict__ ev_vg, float *__restrict__ ev_si, size_t N, int e, int j0, int j1,
                                                float vg, float si) {
    const int n_en = (spec.enable_insol != 0) + (spec.enable_voltage != 0);
    if (j1 > spec.ev_count) j1 = spec.ev_count;
    for (int k = j0 + 1; k <= j1; ++k) {
        if (n_en > 0) {
            const float ut = counter_uniform(seed, gid, episode, k, 0);
            const float uv = counter_uniform(seed, gid, episode, k, 1);
            int pick = (int)(ut * (float)n_en);
            if (pick > n_en - 1) pick = n_en - 1;
            const bool is_insol = spec.enable_insol && (pick == 0);
            if (is_insol) si = lerp_unfused(spec.s_min, spec.s_max, uv);
            else vg = lerp_unfused(spec.v_min, spec.v_max, uv);
        }
        ev_vg[(size_t)k * N + e] = vg;
        ev_si[(size_t)k * N + e] = si;
    }
}
__device__ __forceinline__ void fill_events(const pvder_event_spec &spec, float sinsol0, uint64_t seed,
                                         uint64_t gid, uint64_t episode, float *__restrict__ ev_vg,
                                         float *__restrict__ ev_si, size_t N, int e) {
    ev_vg[e] = 1.0f;
    ev_si[e] = sinsol0;
    fill_event_rows(spec, seed, gid, episode, ev_vg, ev_si, N, e, 0, spec.ev_count, 1.0f, sinsol0);
}

// event-table row active at sim-time-step g (events at ksub = s0 + j*ds, j = 0..K-1; an event
// at exactly g is already active: grid_events(t) returns the latest event with T <= t)
__device__ __forceinline__ int event_row(int g, int s0, int ds, int K) {
    if (K <= 0 || g < s0) return 0;
    int j = (g - s0) / ds + 1;
    return j > K ? K : j;
}

// ---- wave64 sum, total in lane 63, by DPP row operations (quad swaps, half-row / row mirror, then
// row_bcast:15 / row_bcast:31 across the four rows): six VALU instructions with DPP operands -- no
// LDS crossbar round trips (six dependent ds_bpermute of __shfl_xor cost a lone wave ~0.3 us per sum)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);        // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);        // quad_perm:[2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);       // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);       // row_mirror: every lane holds its row's sum
    v = dpp_add<0x142, 0xA>(v);       // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xC>(v);       // row_bcast:31 into rows 2, 3: lane 63 holds the wave's sum
    return v;
}

template <typename T>
__device__ __forceinline__ T ladder(T err, T lo, T hi) { return err <= lo ? (T)1 : (err >= hi ? (T)-5 : (T)-1); }

// one reward type of reward_calc (PVDER_env.py:249-299): the +1 / -1 / -5 ladder or the negative squared error
enum { kRewVoltage = 0, kRewPower, kRewQ, kRewVdc };
template <typename T>
__device__ __forceinline__ T reward_term(int type, const StepK<T> &C, const ModelK<T> &P, T vrms, T Ppcc, T Qpcc, T Vdc,
                                         T vdcref, T qref) {
#pragma clang fp contract(off)       // (the same bits in every kernel: see pvder::outputs)
    T err, lo = (T)0.01, hi = (T)0.05, cont;
    if (type == kRewVoltage) {                     // voltage_error :276-287
        const T d = vrms - P.vrms_ref;
        err = fabs(d) / fabs(P.vrms_ref); cont = -(d * d);
    } else if (type == kRewPower) {                // power_error :290-299
        const T d = Ppcc - C.p_t;
        err = fabs(d) / C.p_t; hi = (T)0.03; cont = -(d * d);
    } else if (type == kRewQ) {                    // Q_error :263-275
        // Qtarget = PV_model.Q_ref under voltage_regulation (:238), the spec's Q_ref / Sbase otherwise (:241)
        T qt = C.goal == PVDER_GOAL_VOLTAGE ? qref : C.q_t;
        const T d = Qpcc - qt;
        cont = -(d * d);
        if (qt == (T)0) qt = (T)1e-6;              // :266-267 (discrete branch only)
        err = fabs(Qpcc - qt) / fabs(qt);
    } else {                                       // Vdc_error :251-262 (Vdctarget = Vdc_ref, :243)
        const T d = vdcref - Vdc;
        err = fabs(d) / vdcref; lo = (T)0.02; cont = -(d * d);
    }
    return C.discrete ? ladder(err, lo, hi) : cont;
}

template <typename T>
__device__ __forceinline__ T reward_of(const StepK<T> &C, const ModelK<T> &P, T vrms, T Ppcc, T Qpcc, T Vdc, T vdcref, T qref) {
    // PVDER_env.py:231-301 -- the goal's 'required' term (:78-93 via :448-452) ...
    T r = reward_term<T>(C.goal == PVDER_GOAL_VOLTAGE ? kRewVoltage : (C.goal == PVDER_GOAL_POWER ? kRewPower : kRewQ), C, P,
                         vrms, Ppcc, Qpcc, Vdc, vdcref, qref);
    // ... plus the optional terms of my_spec (:249), summed (:301): Vdc_error (its target is defined for
    // power_regulation only, :243), Q_error under voltage_regulation, voltage_error under the other two goals
    if (C.extra != 0) {
        if ((C.extra & PVDER_REWARD_VDC_ERROR) && C.goal == PVDER_GOAL_POWER)
            r += reward_term<T>(kRewVdc, C, P, vrms, Ppcc, Qpcc, Vdc, vdcref, qref);
        if ((C.extra & PVDER_REWARD_Q_ERROR) && C.goal == PVDER_GOAL_VOLTAGE)
            r += reward_term<T>(kRewQ, C, P, vrms, Ppcc, Qpcc, Vdc, vdcref, qref);
        if ((C.extra & PVDER_REWARD_VOLTAGE_ERROR) && C.goal != PVDER_GOAL_VOLTAGE)
            r += reward_term<T>(kRewVoltage, C, P, vrms, Ppcc, Qpcc, Vdc, vdcref, qref);
    }
    return r;
}

// ---- per-lane bookkeeping shared by every step kernel
template <typename T>
struct Lane {
    int e, ec;
    bool live, run;
    int ksub, steps, status;
    T qref, vdcref;
    float bad_action;
    bool fail;              // left the range the fast integrators cover and cfg->off_range says: end the episode
};

// State rows are read once and result rows written once per launch: non-temporal accesses keep them
// from lingering in L2 as dirty lines that the end-of-kernel release must write back (10.9 MB per
// launch) -- measured 24.55 -> 23.83 us per step (tools/ab_nt.sh; -DPVDER_PLAIN_MEM builds the A side).
template <typename T>
__device__ __forceinline__ T ld(const T *p) {
#ifdef PVDER_PLAIN_MEM
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
template <typename T>
__device__ __forceinline__ void st(T *p, T v) {
#ifdef PVDER_PLAIN_MEM
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

template <typename T, int NS>
__device__ __forceinline__ void prologue(const StepK<T> &C, const T *__restrict__ fstate,
                                         const int32_t *__restrict__ istate,
                                         const uint8_t *__restrict__ action, int n_env, Lane<T> &L, T *y,
                                         int run_status = 0, int e_own = -1) {
    const size_t N = (size_t)n_env;
    // (e_own: the env this lane works on when it is not the one its thread index names -- the 23-state tail packs the
    //  workgroup's deferred envs into its first lanes)
    L.e = e_own >= 0 ? e_own : (int)(blockIdx.x * kBlock + threadIdx.x);
    L.live = L.e < n_env;
    L.ec = L.live ? L.e : n_env - 1;               // clamp: tail lanes recompute env n-1, never store
    L.ksub = ld(&istate[PVDER_I_KSUB * N + L.ec]);
    L.steps = ld(&istate[PVDER_I_STEPS * N + L.ec]);
    L.status = ld(&istate[PVDER_I_STATUS * N + L.ec]);
    // finished envs (status 1/2) return the stale tuple (:145-152).  run_status = kDeferred selects
    // the envs a preceding fast-path kernel handed over to this launch.
    L.run = L.live && L.status == run_status;
    L.fail = false;
    if (L.run) L.status = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) y[k] = ld(&fstate[(size_t)k * N + L.ec]);
    L.qref = ld(&fstate[(size_t)NS * N + L.ec]);
    L.vdcref = ld(&fstate[(size_t)(NS + 1) * N + L.ec]);
    // ---- action_calc (PVDER_env.py:198-229)
    const int a = ld(&action[L.ec]);
    L.bad_action = 0.f;
    if (a == 1) L.qref += C.dq;
    else if (a == 2) L.qref -= C.dq;
    else if (a == 3) L.vdcref += C.dv;
    else if (a == 4) L.vdcref -= C.dv;
    else if (a != 0) L.bad_action = 1.f;
    // a follow-up launch (run_status != 0) counts invalid actions only for the lanes it advances: the
    // launch that handed them over has counted everyone else already
    if (run_status != 0 && !L.run) L.bad_action = 0.f;
}

// y0 of an env with its own PV array (pvder_plan_lane_params): i_p, x_p = m_p, xDC + j xQ = i_a from the lane's rows
// PVDER_LPF_IA_RE .. PVDER_LPF_MA_IM; every other entry keeps the group's value (u_p = 0, Vdc = Vdc_ref0, xPLL = 0, delta)
template <typename T, int PH>
__device__ __forceinline__ void lane_steady_state(const float *__restrict__ lane_par, size_t N, int e, T *y) {
    const T iR = (T)lane_par[(size_t)PVDER_LPF_IA_RE * N + e], iI = (T)lane_par[(size_t)PVDER_LPF_IA_IM * N + e];
    const T mR = (T)lane_par[(size_t)PVDER_LPF_MA_RE * N + e], mI = (T)lane_par[(size_t)PVDER_LPF_MA_IM * N + e];
    const T h3 = (T)0.86602540378443864676;
    y[0] = iR; y[1] = iI; y[2] = mR; y[3] = mI;
    if (PH == 3) {
        // phase b = phase a e^{-j120}, phase c = phase a e^{+j120}
        y[6] = (T)-0.5 * iR + h3 * iI;  y[7] = (T)-0.5 * iI - h3 * iR;  y[8] = (T)-0.5 * mR + h3 * mI;  y[9] = (T)-0.5 * mI - h3 * mR;
        y[12] = (T)-0.5 * iR - h3 * iI; y[13] = (T)-0.5 * iI + h3 * iR; y[14] = (T)-0.5 * mR - h3 * mI; y[15] = (T)-0.5 * mI + h3 * mR;
    }
    y[6 * PH + 1] = iR; y[6 * PH + 2] = iI;
}

template <typename T>
__device__ __forceinline__ void write_obs(T *__restrict__ out, size_t N, int e, T iaR, T iaI, T vaR, T vaI,
                                          T Ppcc, T Qpcc, T Vdc, T ppv, T vdcref, T qref, T tfrac) {
    st(&out[0 * N + e], iaR);   st(&out[1 * N + e], iaI);
    st(&out[2 * N + e], vaR);   st(&out[3 * N + e], vaI);
    st(&out[4 * N + e], Ppcc);  st(&out[5 * N + e], Qpcc);
    st(&out[6 * N + e], Vdc);   st(&out[7 * N + e], ppv);
    st(&out[8 * N + e], vdcref); st(&out[9 * N + e], qref);
    st(&out[10 * N + e], tfrac);                   // tStart/max_sim_time (:540)
}

// epilogue: reward_calc / done / state at (y(tStop), tStop); optional same-step auto-reset of the
// envs that just finished (PVDER.reset, :316-334); stores; wave-reduced statistics
template <typename T, int PH, bool AR = true>
__device__ __forceinline__ void epilogue(const ModelK<T> &P, const StepK<T> &C, const ResetK &R, Lane<T> &L,
                                         T *y, T *__restrict__ fstate, int32_t *__restrict__ istate,
                                         float *__restrict__ ev_vg, float *__restrict__ ev_si,
                                         T *__restrict__ out, float *__restrict__ stats, int n_env,
                                         int stats_fresh = 0, bool have_inputs = false, float vg_stop = 0.f,
                                         float si_stop = 0.f, const float *__restrict__ lane_par = nullptr,
                                         bool stats_local = false, int gen_j0 = -1, int gen_j1 = -1, int episode_now = -1) {
    // have_inputs: the caller fetched the event inputs in force at tStop ahead of the integration (same table
    // entries; a lone wave per SIMD would otherwise sit out a memory round trip right here)
    constexpr int NS = 6 * PH + 5;
    const size_t N = (size_t)n_env;
    const int ec = L.ec;
    SubK<T, PH> S;
    if (!have_inputs) {
        const int row = event_row(L.ksub, C.ev_s0, C.ev_ds, C.ev_count);
        vg_stop = ev_vg[(size_t)row * N + ec]; si_stop = ev_si[(size_t)row * N + ec];
    }
    pvder::make_sub<T, PH>(P, (T)vg_stop, (T)si_stop, S);
    T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
    pvder::outputs<T, PH>(P, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
    T reward = reward_of<T>(C, P, vrms, Ppcc, Qpcc, y[6 * PH], L.vdcref, L.qref);
#ifdef PVDER_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 32 + 22] = wall_clock64() + (reward > (T)1e30 ? 1 : 0);
#endif
    T chk = (T)0;
#pragma unroll
    for (int q = 0; q < NS; ++q) chk += fabs(y[q]);
    float nonfinite = 0.f, done_f = 0.f;
    if (L.run) {
        if (!(chk < (T)1e30) || L.fail) {          // NaN/Inf (or off range, by request): CONVERGENCE_FAILURE branch (:169-172)
            L.status = 2; reward = (T)-100; nonfinite = 1.f;
        } else if (L.ksub >= C.max_ksub || (C.max_steps > 0 && L.steps >= C.max_steps)) {
            L.status = 1;                          // tStop >= max_sim_time (:183) / TimeLimit
        }
        const int e = L.e;
        done_f = L.status != 0 ? 1.f : 0.f;
        T Vdc = y[6 * PH];
        T tfrac = (T)L.ksub * (T)(1.0 / 60.0) * C.inv_tmax;
        const bool restart = AR && R.enabled && L.status != 0;
        if (R.enabled && !R.keep_events && (restart || R.lazy)) {
            // ---- the event tables of the generator (create_random_events, PVDER_env.py:400-411).  Whole at the auto-reset (the
            // round-1 scheme: 38 rows x 2 counter hashes per env, serial in the lane -- 4.6 us for a wave whose envs end together,
            // profiles/r06_desync_breakdown.txt), or AS THE EPISODE ADVANCES (lazy): a step leaves behind the rows its successor
            // reads -- rows <= event_row(tStop + n_sub) --, carried from the tStop row it holds anyway; the auto-reset writes
            // row 0 (and what the first step reads).  Rows beyond that are the previous episode's until their turn comes
            // (pvder_complete_event_tables for a reader that wants them all).
            // (gen_j0, gen_j1, episode_now: a caller that knew the clock at the start of the launch has them ready -- two integer
            // divisions and a dependent load less behind the integration: lin_step_lean)
            int j0 = 0, j1 = 0;
            float cvg = 1.0f, csi = (float)P.sinsol0;
            if (restart) {
                j1 = R.lazy ? event_row(C.n_sub, C.ev_s0, C.ev_ds, C.ev_count) : R.spec.ev_count;
            } else {
                if (gen_j0 >= 0) { j0 = gen_j0; j1 = gen_j1; }
                else { j0 = event_row(L.ksub, C.ev_s0, C.ev_ds, C.ev_count); j1 = event_row(L.ksub + C.n_sub, C.ev_s0, C.ev_ds, C.ev_count); }
                cvg = vg_stop; csi = si_stop;
            }
            if (restart || j1 > j0) {
                const int episode = (!restart && episode_now >= 0) ? episode_now : istate[PVDER_I_EPISODE * N + e] + (restart ? 1 : 0);
                if (restart) {
                    istate[PVDER_I_EPISODE * N + e] = episode;
                    ev_vg[e] = cvg; ev_si[e] = csi;
                }
                fill_event_rows(R.spec, R.seed, (uint64_t)(R.env_id_base + (R.env_ids ? (int64_t)R.env_ids[e] : (int64_t)e)), (uint64_t)episode,
                                ev_vg, ev_si, N, e, j0, j1, cvg, csi);
            }
        } else if (restart) {
            istate[PVDER_I_EPISODE * N + e] += 1;
        }
        if (restart) {
            // same-step auto-reset: reward/done rows keep the transition that ended the episode,
            // the observation rows and the carried state restart from the steady state
#pragma unroll
            for (int q = 0; q < NS; ++q) y[q] = (T)R.y0[q];
            if (lane_par) lane_steady_state<T, PH>(lane_par, N, e, y);      // the env's OWN operating point (PVDER_env.py:374-378)
            L.qref = (T)0;
            L.vdcref = P.vdc_ref0;
            L.ksub = 0; L.steps = 0; L.status = 0;
            // (row 0 of the tables: the generator's 1.0 pu / nominal insolation, or what an explicit table holds)
            if (R.keep_events) pvder::make_sub<T, PH>(P, (T)ev_vg[e], (T)ev_si[e], S);
            else pvder::make_sub<T, PH>(P, (T)1.0f, (T)(float)P.sinsol0, S);
            pvder::outputs<T, PH>(P, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
            Vdc = y[6 * PH];
            tfrac = (T)0;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) st(&fstate[(size_t)q * N + e], y[q]);
        st(&fstate[(size_t)NS * N + e], L.qref);
        st(&fstate[(size_t)(NS + 1) * N + e], L.vdcref);
        st(&istate[PVDER_I_KSUB * N + e], (int32_t)L.ksub);
        st(&istate[PVDER_I_STEPS * N + e], (int32_t)L.steps);
        st(&istate[PVDER_I_STATUS * N + e], (int32_t)L.status);
        write_obs<T>(out, N, e, iaR, iaI, vaR, vaI, Ppcc, Qpcc, Vdc, ppv, L.vdcref, L.qref, tfrac);
        st(&out[PVDER_O_REWARD * N + e], reward);
        st(&out[PVDER_O_DONE * N + e], (T)done_f);
    }
    PVDER_TRACE_AT(23);
    // ---- statistics: DPP wave reduction, then ONE 16-byte store by lane 63 into THIS WAVE's slot
    // stats[4 * wave + k] -- no atomics (256 workgroups finishing together on four same-address float atomics,
    // ~10 ns each, serialised at the memory side, held every launch back by 2.7 us), and no LDS / barrier
    // either (combining the four waves of a workgroup cost a __syncthreads at the very end of the launch:
    // every wave waited for the slowest of its workgroup).  stats_fresh: overwrite the slot (first launch of
    // an env step) or add to it (a follow-up launch / the 23-state tail of the same step).
    if (stats) {
        const float r = wave_sum_to_lane63(L.run ? (float)reward : 0.f);
        // the three counters are 0 / 1 per lane: a ballot and a population count each (scalar), not a DPP tree
        const float d = (float)__popcll(__ballot((int)(done_f != 0.f)));
        const float nf = (float)__popcll(__ballot((int)(nonfinite != 0.f)));
        const float ba = (float)__popcll(__ballot((int)(L.live && L.bad_action != 0.f)));
        // (lane id from mbcnt, not threadIdx.x & 63: a value the compiler would otherwise keep -- or spill -- from the first
        // instruction of the kernel, where the coefficient loads use it, to this last one)
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 63u) {
            // (stats_local: `stats` IS this workgroup's own block of kBlock / 64 slots -- LDS of step_etdrk4_lane3_kernel)
            v4 *slot = (v4 *)stats + ((stats_local ? (size_t)0 : (size_t)blockIdx.x * (kBlock / 64)) + (threadIdx.x >> 6));
            v4 t = {r, d, nf, ba};
            if (!stats_fresh) t += *slot;
            *slot = t;
        }
    }
}

// =============================================================================================
// classic RK4 micro-stepping, generic template (single phase; A/B reference of the packed form)
// =============================================================================================
// The body of step_rk4_kernel; also the tail of the single-phase etdlin / ETDRK4 step kernels (only_mine: the lanes
// the fast path handed over, `mine`; micro: classic RK4 steps per sim time step).
template <typename T, int PH>
__device__ __forceinline__ void rk4_body(
    const ModelK<T> &P, const StepK<T> &C, int micro, const ResetK &R, T *__restrict__ fstate, int32_t *__restrict__ istate,
    float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ action,
    T *__restrict__ out, float *__restrict__ stats, int n_env, int stats_fresh, int run_status, bool only_mine, bool mine) {
    constexpr int NS = 6 * PH + 5;
    const size_t N = (size_t)n_env;
    Lane<T> L;
    T y[NS], comp[NS];
    prologue<T, NS>(C, fstate, istate, action, n_env, L, y, run_status);
    if (only_mine) L.live = L.run = L.run && mine;
#pragma unroll
    for (int k = 0; k < NS; ++k) comp[k] = (T)0;
    const T h = (T)(1.0 / 60.0) / (T)micro;
    const T h2 = (T)0.5 * h, h6 = h * (T)(1.0 / 6.0);
    const T tq = (T)2 / (T)micro;                  // 2*theta advances 2 turns per 1/60 s
    if (L.run) {
        SubK<T, PH> S;
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            pvder::make_sub<T, PH>(P, (T)ev_vg[(size_t)row * N + L.ec], (T)ev_si[(size_t)row * N + L.ec], S);
            for (int j = 0; j < micro; ++j) {
                T k[NS], ks[NS], yt[NS];
                const T t0 = (T)j * tq, t1 = ((T)j + (T)0.5) * tq, t2 = ((T)j + (T)1) * tq;
                pvder::rhs<T, PH>(P, S, y, t0, L.qref, L.vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] = k[q]; yt[q] = y[q] + h2 * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t1, L.qref, L.vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] += (T)2 * k[q]; yt[q] = y[q] + h2 * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t1, L.qref, L.vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] += (T)2 * k[q]; yt[q] = y[q] + h * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t2, L.qref, L.vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    // Kahan: with h ~ 4 us the increment is below ulp(y)/2 in fp32 and plain
                    // summation would stagnate (DESIGN.md "fp32 hazards")
                    const T inc = h6 * (ks[q] + k[q]) - comp[q];
                    const T tsum = y[q] + inc;
                    comp[q] = (tsum - y[q]) - inc;
                    y[q] = tsum;
                }
            }
        }
        L.ksub += C.n_sub;
        L.steps += 1;
    }
    epilogue<T, PH>(P, C, R, L, y, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh);
}
template <typename T, int PH>
__global__ __launch_bounds__(kBlock) void step_rk4_kernel(
    const ModelK<T> P, const StepK<T> C, const ResetK R, T *__restrict__ fstate, int32_t *__restrict__ istate,
    float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ action,
    T *__restrict__ out, float *__restrict__ stats, int n_env, int stats_fresh,
    int run_status) {
    rk4_body<T, PH>(P, C, C.micro, R, fstate, istate, ev_vg, ev_si, action, out, stats, n_env, stats_fresh, run_status, false, false);
}

// ---- packed pair layout <-> pvder state order
__device__ __forceinline__ void to_pairs(const float *ys, v2 *y) {
#pragma unroll
    for (int k = 0; k < 11; ++k) y[k] = pvder::mk2(ys[2 * k], ys[2 * k + 1]);
    y[11] = pvder::mk2(ys[22], 0.f);
}
__device__ __forceinline__ void from_pairs(const v2 *y, float *ys) {
#pragma unroll
    for (int k = 0; k < 11; ++k) { ys[2 * k] = y[k].x; ys[2 * k + 1] = y[k].y; }
    ys[22] = y[11].x;
}

// nsteps classic RK4 steps of size h on the packed state (Kahan-compensated, u pairs plain);
// 2*theta (turns) at step j, stage c is  tq0 + (j + c)*tq
__device__ __forceinline__ void rk4_pk_steps(const ModelK<float> &P, const pvder::PkConst &K,
                                             const pvder::SubPk &S, v2 *y, v2 *comp, int nsteps, float h,
                                             float tq0, float tq, float qref, float vdcref) {
    constexpr int NP = pvder::NP;
    const v2 h2 = pvder::bc2(0.5f * h), h1 = pvder::bc2(h), h6 = pvder::bc2(h * (float)(1.0 / 6.0));
    const v2 two = pvder::bc2(2.f);
    for (int j = 0; j < nsteps; ++j) {
        v2 k[NP], ks[NP], yt[NP];
        const float t0 = tq0 + (float)j * tq, t1 = tq0 + ((float)j + 0.5f) * tq, t2 = tq0 + ((float)j + 1.f) * tq;
        pvder::rhs_pk(P, K, S, y, t0, qref, vdcref, k);
#pragma unroll
        for (int q = 0; q < NP; ++q) { ks[q] = k[q]; yt[q] = pvder::fma2(h2, k[q], y[q]); }
        pvder::rhs_pk(P, K, S, yt, t1, qref, vdcref, k);
#pragma unroll
        for (int q = 0; q < NP; ++q) { ks[q] = pvder::fma2(two, k[q], ks[q]); yt[q] = pvder::fma2(h2, k[q], y[q]); }
        pvder::rhs_pk(P, K, S, yt, t1, qref, vdcref, k);
#pragma unroll
        for (int q = 0; q < NP; ++q) { ks[q] = pvder::fma2(two, k[q], ks[q]); yt[q] = pvder::fma2(h1, k[q], y[q]); }
        pvder::rhs_pk(P, K, S, yt, t2, qref, vdcref, k);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (q == 2 || q == 5 || q == 8) {
                // u_p: |u| ~ 1e-6 and du/dt is large -> increments are far above ulp(u)/2
                y[q] = pvder::fma2(h6, ks[q] + k[q], y[q]);
            } else {
                const v2 inc = pvder::fma2(h6, ks[q] + k[q], -comp[q]);
                const v2 tsum = y[q] + inc;
                comp[q] = (tsum - y[q]) - inc;
                y[q] = tsum;
            }
        }
    }
}

// =============================================================================================
// classic RK4 micro-stepping, three-phase fp32, packed-VALU formulation
// =============================================================================================
__global__ __launch_bounds__(kBlock) void step_rk4_pk_kernel(
    const ModelK<float> P, const StepK<float> C, const ResetK R, float *__restrict__ fstate,
    int32_t *__restrict__ istate, float *__restrict__ ev_vg, float *__restrict__ ev_si,
    const uint8_t *__restrict__ action, float *__restrict__ out, float *__restrict__ stats, int n_env, int stats_fresh) {
    constexpr int NS = 23, NP = pvder::NP;
    const size_t N = (size_t)n_env;
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(C, fstate, istate, action, n_env, L, ys);
    if (L.run) {
        v2 y[NP], comp[NP];
        to_pairs(ys, y);
#pragma unroll
        for (int k = 0; k < NP; ++k) comp[k] = pvder::bc2(0.f);
        const float h = (float)(1.0 / 60.0) / (float)C.micro;
        const float tq = 2.f / (float)C.micro;
        const pvder::PkConst K = pvder::make_pkconst(P);
        pvder::SubPk S;
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            pvder::make_sub_pk(P, ev_vg[(size_t)row * N + L.ec], ev_si[(size_t)row * N + L.ec], S);
            rk4_pk_steps(P, K, S, y, comp, C.micro, h, 0.f, tq, L.qref, L.vdcref);
        }
        from_pairs(y, ys);
        L.ksub += C.n_sub;
        L.steps += 1;
    }
    epilogue<float, 3>(P, C, R, L, ys, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh);
}

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
    int *hyb_flag;           // device-addressable HOST word: written 1 by a workgroup that meets an env beyond kHybFlag2 (auto mode)
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
// LP (step_etdrk4_lane_kernel): every env with its OWN values of the parameters that stay outside the exponential operator
// (lane_par: [PVDER_LPF_ROWS][n_env] folded constants, pvder_fold_lane_params) -- the lane's ModelK is the launch's with
// those seven fields replaced (VGPRs instead of SGPRs); stats_fresh: this pass starts the step's statistics.
// ---- envs at / near the duty-cycle limit (round 6): the hybrid scheme of pvder_sat_hybrid.hpp, in a kernel of its own behind the step
// kernel -- fp64, 6 KB of private arrays per lane: inside the step kernels it would cost every launch their occupancy and a scratch ring.
constexpr float kHybNear2 = 0.95f * 0.95f;          // an env of the 23-state tail with |m_p|^2 beyond this fraction of the limit's: hybrid
constexpr int kHybMaxSub = 16;                      // sim time steps per env step the hybrid kernel takes (PVDER-v0: 15)
constexpr int kHybHoldLaunches = 256;               // auto mode: the hybrid kernel follows this many step launches behind the last sighting

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
        if (m2 > kHybNear2 * P.mlim2) {
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
// step's).  64 lanes per workgroup, ONE env per lane, the scheme's twelve 23-vectors + the state in the lane's own 2.4 KB of LDS
// (153 KB of the CU's 160: as private arrays they make a 6 KB scratch frame per lane, which the runtime serves by allocating scratch
// per DISPATCH -- 0.2 ms per launch, measured -- and every access a memory round trip).  A workgroup without such an env returns at once.
constexpr int kHybBlock = 64;
constexpr int kHybLaneDoubles = pvder::hyb::kArenaDoubles + 23;                   // odd: lanes on distinct LDS banks at the same index
constexpr size_t kHybLdsBytes = (size_t)kHybBlock * (kHybLaneDoubles * sizeof(double) + 2 * kHybMaxSub * sizeof(float));
__global__ __launch_bounds__(kHybBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_sat_hybrid_kernel(const EtdArgs A, const float *__restrict__ lane_par) {
    using namespace pvder::hyb;
    extern __shared__ __attribute__((aligned(16))) double hyb_lds[];
    const int n_env = A.n_env;
    const size_t N = (size_t)n_env;
    const int e_mine = (int)(blockIdx.x * kHybBlock + threadIdx.x);
    {
        const bool mine = e_mine < n_env && A.istate[(size_t)PVDER_I_STATUS * N + (size_t)e_mine] == kDeferred;
        if (!__syncthreads_or((int)mine)) return;
    }
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
        double *arena = hyb_lds + (size_t)threadIdx.x * kHybLaneDoubles;
        double *y = arena + kArenaDoubles;
        float *vg = (float *)(hyb_lds + (size_t)kHybBlock * kHybLaneDoubles) + (size_t)threadIdx.x * (2 * kHybMaxSub), *si = vg + kHybMaxSub;
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
    // (this workgroup IS wave e / 64 of the step kernels' numbering: its statistics slot)
    epilogue<float, 3>(P, A.C, A.R, L, ys, A.fstate, A.istate, A.ev_vg, A.ev_si, A.out, A.stats ? A.stats + 4 * (size_t)blockIdx.x : nullptr,
                       n_env, 0, false, 0.f, 0.f, lane_par, true);
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
    etdrk4_full_tail<true>(A, true, nullptr, 0, lane_par, A.stats_fresh);
}

// ... and the same transition with an env's three phases on three lanes of a quad (etd_switched_step3): grid = 4 n_env / 256.
// 16 envs per wave, four times the waves, 0.6 x the instructions per wave and step.  Measured (tools/lane_params_rate.py, us
// per batched step, this kernel / a lane per env): 16 384 envs 45 / 71, 32 768 envs 65 / 71, 49 152 envs 89 / 72, 65 536 envs
// 115 / 72 -- the operand reads of the stiff pairs (LDS) and the issue slots are shared by the SIMD's waves, so the form that
// wins a lone wave's latency loses the batch's throughput: launch_plan picks by batch size.
constexpr int kLane3MaxEnvs = 32768;
__global__ __launch_bounds__(kBlock) void step_etdrk4_lane3_kernel(const EtdArgs A, const float *lane_par) {
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

// =============================================================================================
// etdlin: linearised exponential integrator, balanced fast path (pvder_etdlin.hpp).  Same frame as
// step_etdrk4_bal_kernel -- balance check on entry, 11 of 23 states integrated, phases b/c written
// back by rotation, envs near a limit or whose embedded error indicator trips are left untouched
// and advanced by the 23-state path at the end of the kernel -- but the integrator is ETD3RK on
// dy/dt = L y + N(y) with L the FULL 11x11 Jacobian at the nominal operating point: the exponential
// carries the whole linear response, so a segment between two input jumps (new action, new event
// row) takes an exponential-Euler tick of 1-2/(60*micro) s and then the ETD3RK steps the host's
// schedule table lists for a segment of that length -- for PVDER-v0 (events on env-step boundaries)
// two equal steps.  The matrix-vector products run on the matrix cores (MFMA, operands resident in
// registers); the VALU only evaluates f.
//
// The step schedule is PER LANE (each env's own jump positions and table row); only the trip count
// is wave-uniform (lanes that have arrived idle) and a wave whose lanes need different coefficient
// sets makes one pass per distinct set, so an env's trajectory does not depend on which envs share
// its wave (bit for bit: every lane's arithmetic is the same whatever its neighbours do).
// =============================================================================================
struct LinK {
    const float *lcoef;              // device: tick matrix (kLinMatFloats) + (n_sets-1) * kLinSetFloats floats, read as
                                     // MFMA A operands straight from memory (pvder_etdlin.hpp)
    const unsigned long long *sched; // device: row L-1 = the ETD3RK sets (one byte each, 0 = end) a segment of L sim
                                     // time steps takes behind its tick
    float est_guard;                 // a step whose indicator max |y+ - b| over (i_a, Vdc) exceeds this hands the env
                                     // to the 23-state path
    float ir2_careful;               // (0.95 iref_limit)^2: an env whose |iref_a|^2 exceeds it at the start of the tick or of an ETD3RK
                                     // step takes the careful pass (current limit: hybrid steps with the outer loops' anti-windup,
                                     // finer schedule).  Below the limit the standard pass IS the model; the 5 % cover what |iref|
                                     // does between two samples 1/8 s apart and comes back from (the outer loops' overshoot behind an
                                     // event: 2-3 %); an excursion that stays shows in the next sample.  (0.9 until late in round 3.)
    float sat2_hard;                 // |m|^2 beyond which an env leaves the balanced reduction for the 23-state path.  Between
                                     // the smooth-step guard (0.998 m_limit: what two samples per 0.125 s step can vouch
                                     // for) and this one (0.9995: what the careful pass's 24 samples can) it takes the
                                     // careful pass -- the anti-windup does not engage below m_limit, the reduction is exact
    float seqE[18];                  // exp(A0 T): what an env step does to a negative- / zero-sequence component (i, x, u) of an
                                     // env slightly off balance (lin_seq_split); complex 3x3, row-major (re, im)
    const float *k4coef;             // device: the ten f-form matrices of ONE Krogstad ETDRK4 step over a whole env step (behind the
                                     // sets in the coefficient block; pvder_lin_coef.k4), or NULL: every segment by tick + ETD3RK steps
};
// The coefficient registers of the matrix-core path and what they hold (wave-uniform): the tick matrix, ONE ETD3RK set
// (`cur`: its index, 0 = none) and the Krogstad block.  A launch loads what the step it expects needs (step_etdlin_kernel:
// the plan's hint says whether the batch's env step starts at an event row change); whoever needs the rest loads it.
struct LinRegs {
    pvder::LinMatRegs At;
    pvder::LinSetRegs Ac;
    pvder::LinK4Regs Ak;
    int cur;
    bool at_ok, k4_ok;
};
struct LinArgs {
    EtdArgs A;                       // at kernarg offset 0: the 23-state tail re-reads it from there
    LinK Lk;
};
// open-loop rollouts (pvder_plan_rollout): the arguments that move from step to step
struct RollK {
    const uint8_t *actions;          // [n_steps][n_env]
    int n_steps;
    long long out_stride;            // floats between the output blocks of consecutive steps (0: one block)
    int *progress;                   // [n_wg]: steps of the rollout the workgroup has behind it
    int resume;                      // fused kernel: 0 = start at step 0, 1 = start at progress[wg]
    int max_finish;                  // finish kernel: at most this many steps per workgroup
    // closed loop (pvder_plan_rollout_fused_agent): the agent step between two env steps of the launch
    pvder_agent::AgentK ag;          // ag.counter = exploration key of the agent call behind step 0 (+ k behind step k)
    float *q;                        // the table
    long long *ag_state;             // [n_env] table row of the observation the pending action was chosen in
    uint8_t *ag_action;              // [n_env] the pending action (in: chosen behind the reset / the previous rollout)
    uint8_t *ag_alive;               // [n_env]
    double *ag_ret;                  // [n_env] return of the running episode
};

// first segment of an env step that starts at sim time step ksub: ONE integer division gives the event
// row in force, whether an event starts exactly there, and `until` = sim time steps to the next
// event-row change; the segment's inputs and schedule row are requested right away
struct LinSeg {
    int row, until, seg;
    bool at_event;                   // an event row starts exactly at the segment's first sim time step (an input jump whatever the action)
    float in_vg, in_si;
    unsigned long long sch;
};
__device__ __forceinline__ LinSeg lin_first_segment(const StepK<float> &C, const LinK &Lk, int ksub, const float *ev_vg,
                                                    const float *ev_si, size_t N, int ec) {
    constexpr int kBig = 0x3fffffff;
    LinSeg s;
    s.row = 0; s.until = kBig; s.at_event = false;
    if (C.ev_count > 0) {
        const int d = ksub - C.ev_s0;
        if (d < 0) {
            s.until = -d;
        } else {
            const int qd = d / C.ev_ds, rd = d - qd * C.ev_ds;
            s.row = qd + 1 < C.ev_count ? qd + 1 : C.ev_count;
            s.until = qd + 1 < C.ev_count ? C.ev_ds - rd : kBig;
            s.at_event = rd == 0 && qd < C.ev_count;
        }
    }
    s.seg = s.until < C.n_sub ? s.until : C.n_sub;
    s.in_vg = ev_vg[(size_t)s.row * N + ec];
    s.in_si = ev_si[(size_t)s.row * N + ec];
    s.sch = Lk.sched[s.seg - 1];
    return s;
}

// balance check of a three-phase state (tolerances: 2e-5 relative on i and x, 2e-9 absolute on u ~ 1e-6)
__device__ __forceinline__ bool lin_unbalanced(const float *ys) {
    using namespace pvder;
    const v2 ia = mk2(ys[0], ys[1]), xa = mk2(ys[2], ys[3]), ua = mk2(ys[4], ys[5]);
    const float ei = unbalance2(ia, mk2(ys[6], ys[7]), mk2(ys[12], ys[13]));
    const float ex = unbalance2(xa, mk2(ys[8], ys[9]), mk2(ys[14], ys[15]));
    const float eu = unbalance2(ua, mk2(ys[10], ys[11]), mk2(ys[16], ys[17]));
    const v2 i2 = ia * ia, x2 = xa * xa;
    return ei > 4e-10f * (i2.x + i2.y) + 1e-20f || ex > 4e-10f * (x2.x + x2.y) + 1e-20f || eu > 4e-18f;
}

// ---- envs slightly OFF BALANCE stay on the matrix-core path: symmetrical components -----------------------------------
// z+ = (z_a + a z_b + a^2 z_c)/3, z- = (z_a + a^2 z_b + a z_c)/3, z0 = (z_a + z_b + z_c)/3 of z = i, x, u (a = e^{+j120}).
// Below the duty-cycle limit the phase equations are complex-linear in (i_p, x_p, u_p) given the shared states, and the
// forcing (grid voltage, current reference) is positive-sequence: z+ with the shared states obeys the BALANCED model --
// up to terms quadratic in z-, z0 in the power sums and a 2wt ripple K_vi i- in the PLL input --, z- and z0 decay by
// themselves, z(T) = exp(A0 T) z(0) (LinK::seqE; host: gym_PVDER_amd/etd_linear.py::seq_propagator).  What survives an
// anti-windup episode for seconds is the integrator of the current controller (x' = -(Ki/Kp) x, 1/3 per second): a
// 1e-3 imbalance that kept an env on the 23-state path for ~50 env steps and its whole workgroup waiting for it.
// An off-balance env whose z-, z0 are small (the neglected terms below 1e-5 relative, at the start and at the end of the
// env step) runs the standard / careful pass on z+; its duty-cycle guards are tightened by (1 + r)^2, r = 1.6 x the
// share of |m-| + |m0| in |m+| (each PHASE must stay clear of the limit; 1.6: the ringing of the fast modes).
__device__ __forceinline__ v2 seq_rot(v2 z, float sgn) {            // z e^{sgn j120}
    using namespace pvder;
    return bc2(-0.5f) * z + swp(z) * mk2(-sgn * kHalfSqrt3, sgn * kHalfSqrt3);
}
// (z-, z0) of one quantity from its three phase values; zp: the positive-sequence component
__device__ __forceinline__ void seq_of(v2 za, v2 zb, v2 zc, v2 &zp, v2 &zn, v2 &z0) {
    using namespace pvder;
    const v2 third = bc2(1.f / 3.f);
    zp = third * (za + seq_rot(zb, 1.f) + seq_rot(zc, -1.f));
    zn = third * (za + seq_rot(zb, -1.f) + seq_rot(zc, 1.f));
    z0 = third * (za + zb + zc);
}
__device__ __forceinline__ v2 cmul(float er, float ei, v2 z) { return pvder::mk2(er * z.x - ei * z.y, er * z.y + ei * z.x); }
__device__ __forceinline__ float abs2(v2 z) { return __builtin_fmaf(z.y, z.y, z.x * z.x); }
// the components an env step later: zo = E z
__device__ __forceinline__ void seq_advance(const float (&E)[18], const v2 (&z)[3], v2 (&zo)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
        zo[r] = cmul(E[6 * r], E[6 * r + 1], z[0]) + cmul(E[6 * r + 2], E[6 * r + 3], z[1]) + cmul(E[6 * r + 4], E[6 * r + 5], z[2]);
}
constexpr float kSeqI2Max = 1e-5f, kSeqMI2Max = 1e-10f, kSeqRMax = 0.05f;
// Split the 18 phase rows in ys: returns whether the env qualifies; then ys[0..5] hold z+ (the rows lin_from_rows reads)
// and msc = (1 + r)^2.  ys[6..17] are left alone (dead: lin_seq_merge re-reads the rows it needs from memory, so that no
// register stays live across the integration for the sake of an env in ten thousand).
__device__ __forceinline__ bool lin_seq_split(const ModelK<float> &P, const LinK &Lk, float *ys, float &msc) {
    using namespace pvder;
    v2 zp[3], zn[3], z0[3], en[3], e0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        seq_of(mk2(ys[2 * k], ys[2 * k + 1]), mk2(ys[6 + 2 * k], ys[7 + 2 * k]), mk2(ys[12 + 2 * k], ys[13 + 2 * k]), zp[k], zn[k], z0[k]);
    seq_advance(Lk.seqE, zn, en);
    seq_advance(Lk.seqE, z0, e0);
    const v2 kp = bc2(P.KpG);
    const float ip2 = abs2(zp[0]), mp2 = abs2(fma2(kp, zp[2], zp[1]));
    const float i2 = abs2(zn[0]) + abs2(z0[0]), m2 = abs2(fma2(kp, zn[2], zn[1])) + abs2(fma2(kp, z0[2], z0[1]));
    const float j2 = abs2(en[0]) + abs2(e0[0]), n2 = abs2(fma2(kp, en[2], en[1])) + abs2(fma2(kp, e0[2], e0[1]));
    const float r = 1.6f * __builtin_sqrtf(2.f * fmaxf(m2, n2) / mp2);
    const bool ok = i2 <= kSeqI2Max * ip2 && m2 * i2 <= kSeqMI2Max * mp2 * ip2 && j2 <= kSeqI2Max * ip2 &&
                    n2 * j2 <= kSeqMI2Max * mp2 * ip2 && r <= kSeqRMax;      // (NaN: not ok)
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { ys[2 * k] = zp[k].x; ys[2 * k + 1] = zp[k].y; }
        msc = (1.f + r) * (1.f + r);
    }
    return ok;
}
// ys: the 23 rows lin_to_rows wrote from z+ (phases b, c = rotations); adds the negative- and zero-sequence components
// the env step leaves of the ones the env STARTED with (rows 0..17 of fstate, still the state before the step)
__device__ __forceinline__ void lin_seq_merge(const LinK &Lk, const float *__restrict__ fstate, size_t N, int e, float *ys) {
    using namespace pvder;
    v2 zp, zn[3], z0[3], en[3], e0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *p = fstate + (size_t)(2 * k) * N + e;
        seq_of(mk2(p[0], p[N]), mk2(p[6 * N], p[7 * N]), mk2(p[12 * N], p[13 * N]), zp, zn[k], z0[k]);
    }
    seq_advance(Lk.seqE, zn, en);
    seq_advance(Lk.seqE, z0, e0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const v2 da = en[k] + e0[k], db = seq_rot(en[k], 1.f) + e0[k], dc = seq_rot(en[k], -1.f) + e0[k];
        ys[2 * k] += da.x; ys[2 * k + 1] += da.y;
        ys[6 + 2 * k] += db.x; ys[7 + 2 * k] += db.y;
        ys[12 + 2 * k] += dc.x; ys[13 + 2 * k] += dc.y;
    }
}

// 23-state rows <-> the accumulator layout of the 11-state reduction (rows 0-3, 4-7, 8-11)
template <int PH>
__device__ __forceinline__ void lin_from_rows(const float *ys, pvder::v4f (&Y)[3]) {
    constexpr int B = 6 * PH;
    Y[0] = {ys[0], ys[1], ys[2], ys[3]};
    Y[1] = {ys[4], ys[5], ys[B], ys[B + 1]};
    Y[2] = {ys[B + 2], ys[B + 3], ys[B + 4], 0.f};
}
template <int PH>
__device__ __forceinline__ void lin_to_rows(const pvder::v4f (&Y)[3], float *ys) {   // phases b, c = phase a rotated by -/+ 120 degrees
    using namespace pvder;
    constexpr int B = 6 * PH;
    const v2 jh3 = mk2(-kHalfSqrt3, kHalfSqrt3);
    const v2 z3[3] = {mk2(Y[0].x, Y[0].y), mk2(Y[0].z, Y[0].w), mk2(Y[1].x, Y[1].y)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ys[2 * k] = z3[k].x; ys[2 * k + 1] = z3[k].y;
        if constexpr (PH == 3) {
            const v2 hz = bc2(-0.5f) * z3[k], jz = swp(z3[k]) * jh3;
            const v2 zb = hz - jz, zc = hz + jz;
            ys[6 + 2 * k] = zb.x; ys[6 + 2 * k + 1] = zb.y;
            ys[12 + 2 * k] = zc.x; ys[12 + 2 * k + 1] = zc.y;
        }
    }
    ys[B] = Y[1].z; ys[B + 1] = Y[1].w; ys[B + 2] = Y[2].x; ys[B + 3] = Y[2].y; ys[B + 4] = Y[2].z;
}

// One ETD3RK step of every lane from Y with the coefficient set in Ac: Yn = y+, returns whether the lane
// must be handed to the 23-state path (|m| guard at the start, the middle -- stage a --, the end of the step;
// embedded error indicator, |m| beyond the hard guard); near_ilim: |iref| at the start of the step is within 5 % of the
// current limit, or |m| beyond the smooth-step guard at the start, middle or end (careful pass)
template <int PH, bool LEAN = false>
__device__ __forceinline__ bool lin_etd3rk(const LinArgs &G, const pvder::LinSetRegs &Ac, const pvder::v4f (&Y)[3],
                                           const pvder::SubPk &S, float qref, float vdcref, v2 kpo, v2 kio, float irg,
                                           float msc, pvder::v4f (&Yn)[3], bool traced, bool &near_ilim) {
    using namespace pvder;
    const ModelK<float> &P = G.A.P;
    const EtdK &Ec = G.A.Ec;
    const PkConst K = make_pkconst(P);
    v2 yp[6], F[6];
    float m2a, ir2a, m2x, ir2x, m2b;
    lin_pairs(Y, yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a);
    if (traced) PVDER_TRACE_DEP(24, F[0].x + F[2].y + F[3].x + F[5].x);
    v4f acc3[3][3] = {{Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}};
    {
        const int m3[3] = {LM21, LM31, LM41};
        lin_mfma<3>(Ac, m3, F, acc3);                     // a = y + M21 F1; b, y+ started
    }
    if (traced) PVDER_TRACE_DEP(25, acc3[0][0].x + acc3[1][1].x + acc3[2][2].x);
    lin_pairs(acc3[0], yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2x, ir2x);
    if (traced) PVDER_TRACE_DEP(26, F[0].x + F[2].y + F[3].x + F[5].x);
    v4f acc2[2][3] = {{acc3[1][0], acc3[1][1], acc3[1][2]}, {acc3[2][0], acc3[2][1], acc3[2][2]}};
    {
        const int m2[2] = {LM32, LM42};
        lin_mfma<2>(Ac, m2, F, acc2);                     // b complete
    }
    if (traced) PVDER_TRACE_DEP(27, acc2[0][0].x + acc2[1][1].x);
    lin_pairs(acc2[0], yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2b, ir2x);
    if (traced) PVDER_TRACE_DEP(28, F[0].x + F[2].y + F[3].x + F[5].x);
    Yn[0] = acc2[1][0]; Yn[1] = acc2[1][1]; Yn[2] = acc2[1][2];
    lin_mfma1(Ac[LM43], F, Yn);                           // y+
    if (traced) PVDER_TRACE_DEP(29, Yn[0].x + Yn[2].x);
    // embedded indicator: 3rd-order y+ against the 2nd-order stage value b
    const float est = fmaxf(fmaxf(fabsf(Yn[0].x - acc2[0][0].x), fabsf(Yn[0].y - acc2[0][0].y)),
                            fabsf(Yn[1].z - acc2[0][1].z));
    const v2 mn = fma2(bc2(P.KpG), mk2(Yn[1].x, Yn[1].y), mk2(Yn[0].z, Yn[0].w));
    // |m| at the start, the middle (stage a), the end (stage b and y+) of the step
    // (msc: 1, or the (1 + r)^2 of an env carried by its positive-sequence component, lin_seq_split)
    const float mmax = msc * fmaxf(fmaxf(m2a, m2x), fmaxf(m2b, __builtin_fmaf(mn.y, mn.y, mn.x * mn.x)));
    // the current reference or the duty cycle comes near its limit: the careful pass redoes the env step
    // (LEAN: the duty cycle at the START of the step beyond 97 % of its limit sends the env to the second pass, which raises the plan's
    //  hybrid flag -- not the peak, which every action's transient lifts)
    near_ilim = ir2a > irg || mmax > Ec.sat2_smooth PVDER_HYB_LEAN(|| (LEAN && msc * m2a > Ec.sat2_flag));
    return mmax > G.Lk.sat2_hard || est > G.Lk.est_guard;
}

// One Krogstad ETDRK4 step of every lane over a WHOLE env step (round 4: the env step behind a mere action -- no event row
// change at its start, so no input jump the stiff loop would ring from and no tick: tools/scheme_proto.py, fp32 on the
// committed episodes: 7.0 -> 4.7 right-hand sides per env step of a PVDER-v0 rollout at unchanged accuracy).  Same guards as
// lin_etd3rk: |m| at the start, the two middle stages, the c = 1 stage and the end; |iref| at the start and the middle (the
// careful pass's threshold); embedded indicator y+ against the stage value U4.
// EARLY (the second pass of the step kernels; `mine`: the lanes the step is for): when ALL of them are beyond the careful
// pass's thresholds at the START of the step -- a batch held at a limit; the single-phase unit whenever Q_ref moves -- the step
// is cut short behind its first right-hand side (`cut`, wave-uniform; Yn is then meaningless): the careful pass redoes those
// lanes whatever the other three right-hand sides would say.  (Not in the lean first pass: a ballot and a branch between the
// first right-hand side and its products cost the standard path 0.18 us, profiles/r04_ab_k4_early_out.txt.)
template <int PH, bool EARLY>
__device__ __forceinline__ bool lin_k4(const LinArgs &G, const pvder::LinK4Regs &Ak, const pvder::v4f (&Y)[3],
                                       const pvder::SubPk &S, float qref, float vdcref, v2 kpo, v2 kio, float irg,
                                       float msc, pvder::v4f (&Yn)[3], bool &near_ilim, bool mine, bool &cut,
                                       const ModelK<float> *lane_model = nullptr) {
    using namespace pvder;
    const ModelK<float> &P = lane_model ? *lane_model : G.A.P;        // (step_etdlin_lane_kernel: the env's own constants in f)
    const EtdK &Ec = G.A.Ec;
    const PkConst K = make_pkconst(P);
    v2 yp[6], F[6];
    float m2a, ir2a, m2x, ir2x, m2y, ir2y, m2b, ir2b;
    lin_pairs(Y, yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a);
    cut = false;
    if constexpr (EARLY) {
        const bool flagged = ir2a > irg || msc * m2a > Ec.sat2_smooth;
        if (__builtin_amdgcn_readfirstlane(!__any((int)(mine && !flagged)))) {
            near_ilim = flagged;
            cut = true;
            return msc * m2a > G.Lk.sat2_hard;
        }
    }
    v4f acc4[4][3] = {{Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}};
    {
        const int m4[4] = {KM21, KM31, KM41, KB1};
        lin_mfma<4>(Ak, m4, F, acc4);                     // U2 = y + M21 F1; U3, U4, y+ started
    }
    lin_pairs(acc4[0], yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2x, ir2x);
    v4f acc3[3][3] = {{acc4[1][0], acc4[1][1], acc4[1][2]}, {acc4[2][0], acc4[2][1], acc4[2][2]}, {acc4[3][0], acc4[3][1], acc4[3][2]}};
    {
        const int m3[3] = {KM32, KM42, KB2};
        lin_mfma<3>(Ak, m3, F, acc3);                     // U3 complete
    }
    lin_pairs(acc3[0], yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2y, ir2y);
    v4f acc2[2][3] = {{acc3[1][0], acc3[1][1], acc3[1][2]}, {acc3[2][0], acc3[2][1], acc3[2][2]}};
    {
        const int m2[2] = {KM43, KB3};
        lin_mfma<2>(Ak, m2, F, acc2);                     // U4 complete
    }
    lin_pairs(acc2[0], yp);
    bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2b, ir2b);
    Yn[0] = acc2[1][0]; Yn[1] = acc2[1][1]; Yn[2] = acc2[1][2];
    lin_mfma1(Ak[KB4], F, Yn);                            // y+
    const float est = fmaxf(fmaxf(fabsf(Yn[0].x - acc2[0][0].x), fabsf(Yn[0].y - acc2[0][0].y)),
                            fabsf(Yn[1].z - acc2[0][1].z));
    const v2 mn = fma2(bc2(P.KpG), mk2(Yn[1].x, Yn[1].y), mk2(Yn[0].z, Yn[0].w));
    const float mmax = msc * fmaxf(fmaxf(fmaxf(m2a, m2x), fmaxf(m2y, m2b)), __builtin_fmaf(mn.y, mn.y, mn.x * mn.x));
    near_ilim = ir2a > irg || ir2y > irg || mmax > Ec.sat2_smooth PVDER_HYB_LEAN(|| (!EARLY && msc * m2a > Ec.sat2_flag));     // (EARLY = the second pass)
    return mmax > G.Lk.sat2_hard || est > G.Lk.est_guard;
}

// the tick matrix and an ETD3RK set in registers (whoever steps by tick + ETD3RK steps: a launch that expected a quiet
// env step has loaded the Krogstad block only)
__device__ __forceinline__ void lin_need_t2(const LinArgs &G, int lane, LinRegs &R) {
    if (!R.at_ok) {
        pvder::lin_load_mat(G.Lk.lcoef, lane, R.At);
        R.at_ok = true;
    }
    if (R.cur == 0) {
        pvder::lin_load_set(G.Lk.lcoef + pvder::kLinMatFloats, lane, R.Ac);
        R.cur = 1;
    }
}

// Does the duty cycle stay clear of its limit over the TICK behind an input jump (Y -> acc)?  The jump kicks the
// under-damped current loop (poles -1.0e5 +- j5.8e5 1/s, zeta = 0.17): m = Kp_GCC u + x rings around its new value --
// reached well inside the tick, 20 decay times -- with a first overshoot of exp(-pi zeta / sqrt(1 - zeta^2)) = 0.58
// of the jump.  Peak = |m_new| + 0.6 |m_new - m_old| against the guards (lin_integrate): the clearance an env needs
// follows the jump it actually sees (an action alone: none; a 2 % grid-voltage event: 1.2 %) where a blanket 3 % sent
// every env beyond |m| = 0.97 to the 23-state path at every event (among 65 536 envs on a Vdc_ref ramp: thousands).
__device__ __forceinline__ float lin_tick_peak2(const ModelK<float> &P, const pvder::v4f (&Y)[3],
                                                const pvder::v4f (&acc)[3], float m2a) {
    using namespace pvder;
    const v2 m0 = fma2(bc2(P.KpG), mk2(Y[1].x, Y[1].y), mk2(Y[0].z, Y[0].w));
    const v2 mn = fma2(bc2(P.KpG), mk2(acc[1].x, acc[1].y), mk2(acc[0].z, acc[0].w));
    const v2 dm = mn - m0;
    const float n2 = __builtin_fmaf(mn.y, mn.y, mn.x * mn.x), d2 = __builtin_fmaf(dm.y, dm.y, dm.x * dm.x);
    const float peak2 = n2 + 1.2f * __builtin_sqrtf(n2 * d2) + 0.36f * d2;          // (|m_new| + 0.6 |dm|)^2
    return fmaxf(m2a, peak2);
}

// next byte of a lane's schedule row; a byte with bit 7 set continues the row in another one (rows behind the
// 2 * n_sub regular ones: segments that need more than seven steps)
__device__ __forceinline__ unsigned long long lin_sched_next(const LinArgs &G, unsigned long long sch) {
    sch >>= 8;
#ifndef PVDER_NO_CONT
    if (sch & 0x80ull) sch = G.Lk.sched[2 * G.A.C.n_sub + (int)(sch & 0x7full)];
#endif
    return sch;
}

// One env step of the lanes with `run` set, state Y in registers: per-lane schedule, wave-uniform trip
// count.  Returns the lanes that must be handed to the 23-state path (their Y is then meaningless);
// `careful`: the lanes whose current reference comes near its limit -- the careful pass (lin_integrate_careful)
// redoes their env step from the state they started with (their Y is meaningless too).
// SPLIT = false (the fused rollout kernel, which hands both kinds to the per-step body): one flag, returned.
template <int PH, bool SPLIT = true>
__device__ __forceinline__ bool lin_integrate(const LinArgs &G, int lane, LinRegs &R, pvder::v4f (&Y)[3], float qref,
                                              float vdcref, int ec, LinSeg sg, bool run_all, bool defer, bool &careful,
                                              int trace_step, float msc = 1.f) {
    using namespace pvder;
    constexpr int kBig = 0x3fffffff;
    const ModelK<float> &P = G.A.P;
    const StepK<float> &C = G.A.C;
    const EtdK &Ec = G.A.Ec;
    const size_t N = (size_t)G.A.n_env;
    const PkConst K = make_pkconst(P);
    // outer-loop gains pinned to SGPRs (see step_etdrk4_bal_kernel: the dispatch-packet trap)
    float g_kpdc = P.KpDC, g_kpq = -P.KpQ, g_kidc = P.KiDC, g_kiq = -P.KiQ;
    asm volatile("" : "+s"(g_kpdc), "+s"(g_kpq), "+s"(g_kidc), "+s"(g_kiq));
    const v2 kpo = mk2(g_kpdc, g_kpq), kio = mk2(g_kidc, g_kiq);
    float irg = G.Lk.ir2_careful;
    asm volatile("" : "+v"(irg));
    careful = false;
    SubPk S;
    // ---- the QUIET env step: one segment, no event row change at its start (a new action at most) -- ONE Krogstad step, no
    // tick (lin_k4).  Per lane, as everything here: the lanes of a wave whose envs start at an event row take the tick + ETD3RK
    // path below, in the same launch (envs in lockstep -- PVDER-v0 rollouts -- agree).
    bool quiet = false;
#ifndef PVDER_NO_WAVE_BAIL
    if constexpr (!SPLIT) {          // (an env that leaves from the start -- off balance, forced: see the wave bail below)
        if (__builtin_amdgcn_readfirstlane(__any((int)(run_all && defer)))) return true;
    }
#endif
#ifndef PVDER_NO_K4
    quiet = sg.seg >= C.n_sub && !sg.at_event && G.Lk.k4coef != nullptr;
    if (__builtin_amdgcn_readfirstlane(__any((int)(run_all && !defer && quiet)))) {
        if (!R.k4_ok) {
            lin_load_k4(G.Lk.k4coef, lane, R.Ak);
            R.k4_ok = true;
        }
        if (trace_step == 0) PVDER_TRACE_AT(3);
        make_sub_pk(P, sg.in_vg, sg.in_si, S);
        v4f Yn[3];
        bool near, cut;
        const bool trip = lin_k4<PH, SPLIT>(G, R.Ak, Y, S, qref, vdcref, kpo, kio, irg, msc, Yn, near, run_all && !defer && quiet, cut);
        if (run_all && quiet) {
            if constexpr (SPLIT) { careful |= near; defer |= trip && !careful; } else defer |= trip || near;
            if (!cut) { Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2]; }
        }
    }
#endif
    const bool run = run_all && !quiet;          // (the lanes the tick + ETD3RK schedule below is for)
    if (!__builtin_amdgcn_readfirstlane(__any((int)(run && !defer)))) {
#ifndef PVDER_NO_WAVE_BAIL
        if constexpr (!SPLIT) {          // (the lean first pass gives the WAVE up as soon as one of its envs leaves: see below)
            if (__builtin_amdgcn_readfirstlane(__any((int)(run_all && defer)))) return true;
        }
#endif
        return defer;
    }
    lin_need_t2(G, lane, R);
    const LinMatRegs &At = R.At;
    LinSetRegs &Ac = R.Ac;
    int &cur = R.cur;
    int spos = 0;                        // sim time steps of the env step behind this lane
    bool start = true;                   // this lane is at the start of a segment (its tick is due)
    bool done = !run;
    cur = __builtin_amdgcn_readfirstlane(cur);           // (wave-uniform by construction: tell hipcc)
    // ---- the standard env step, straight-line: every lane that runs has ONE segment (no event row change
    // inside the env step) covered by the tick and two steps of the resident set -- PVDER-v0 with its envs
    // in step.  Same arithmetic, same commits as the general loop below, without its bookkeeping (at one
    // wave per SIMD every instruction is 2-3 ns: the loop's ballots, schedule bytes and segment counters were
    // a quarter of an env step's integration time).
    {
        const bool standard = sg.seg >= C.n_sub && (sg.sch & 0xffffffull) == (unsigned long long)(cur | (cur << 8));
        if (__builtin_amdgcn_readfirstlane(__any((int)(run && !defer && !standard))) == 0) {
            if (!__builtin_amdgcn_readfirstlane(__any((int)(run && !defer)))) return defer;
#ifndef PVDER_NO_WAVE_BAIL
            if constexpr (!SPLIT) {          // (an env that leaves from the start -- off balance, forced: see the wave bail below)
                if (__builtin_amdgcn_readfirstlane(__any((int)(run && defer)))) return true;
            }
#endif
            if (trace_step == 0) PVDER_TRACE_AT(3);
            v2 yp[6], F[6];
            float m2a, ir2a;
            make_sub_pk(P, sg.in_vg, sg.in_si, S);
            lin_pairs(Y, yp);
            bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a);
            v4f acc[3] = {Y[0], Y[1], Y[2]};
            lin_mfma1(At, F, acc);
            const float pk2 = msc * lin_tick_peak2(P, Y, acc, m2a);
            bool trip = pk2 > G.Lk.sat2_hard;
            if (run) {
                if constexpr (SPLIT) { careful |= ir2a > irg || pk2 > Ec.sat2_smooth; defer |= trip && !careful; }
                else defer |= pk2 > Ec.sat2_smooth PVDER_HYB_LEAN(|| msc * m2a > Ec.sat2_flag) || ir2a > irg;
                Y[0] = acc[0]; Y[1] = acc[1]; Y[2] = acc[2];
            }
            // (a wave whose envs all sit at the current limit -- a policy that holds one Q_ref action -- goes
            // straight to the careful pass)
#ifndef PVDER_NO_EARLY
            if (!__builtin_amdgcn_readfirstlane(__any((int)(run && !defer && !careful)))) return defer;
#endif
#ifndef PVDER_NO_WAVE_BAIL
            // The lean first pass (SPLIT = false: step_etdlin_kernel's, the fused rollout's) gives the WAVE up as soon as one
            // of its envs leaves: the workgroup goes through the second pass anyway (the fused rollout: to the finish kernel),
            // which redoes this wave's envs from the state in memory with the same arithmetic -- the two ETD3RK steps here
            // would be 4 us spent on results nobody needs at every step of a batch held at a limit (fixed-action policies:
            // PVDER_env.py:480-487).  Nothing of a wave that bails is stored: bit-identical results.
            if constexpr (!SPLIT) {
                if (__builtin_amdgcn_readfirstlane(__any((int)(run && defer)))) return true;
            }
#endif
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (trace_step == 0 && st == 1) PVDER_TRACE_AT(4);
                v4f Yn[3];
                bool near;
                trip = lin_etd3rk<PH, !SPLIT>(G, Ac, Y, S, qref, vdcref, kpo, kio, irg, msc, Yn, trace_step == 0 && st == 1, near);
                if (run) {
                    if constexpr (SPLIT) { careful |= near; defer |= trip && !careful; } else defer |= trip || near;
                    Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2];
                }
            }
            return defer;
        }
    }
    for (int it = 0;; ++it) {
        if (!__builtin_amdgcn_readfirstlane(__any((int)(!done && !defer && !careful)))) break;
        if (trace_step == 0 && it < 16) PVDER_TRACE_AT(3 + it);
        v2 yp[6], F[6];
        float m2a, ir2a;
        if (__builtin_amdgcn_readfirstlane(__any((int)(start && !done)))) {
            // ---- a segment starts (new action at the start of the env step, or a new event row): the
            // TICK behind the jump, exponential Euler  y+ = y + h phi1(hL) f(y).  The jump kicks the stiff
            // loop; the exponential carries that transient exactly, and over one tick the rest of f
            // hardly moves (fp64 prototype: same trajectory error as an ETD3RK tick).  The jump may make
            // |m| overshoot inside the tick: lin_tick_trip.
            // (lanes in the middle of a segment recompute S from the inputs they hold: same values)
            make_sub_pk(P, sg.in_vg, sg.in_si, S);
            lin_pairs(Y, yp);
            bal_f<PH>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a);
            v4f acc[3] = {Y[0], Y[1], Y[2]};
            lin_mfma1(At, F, acc);
            const float pk2 = msc * lin_tick_peak2(P, Y, acc, m2a);
            const bool trip = pk2 > G.Lk.sat2_hard;
            if (start && !done) {
                if constexpr (SPLIT) { careful |= ir2a > irg || pk2 > Ec.sat2_smooth; defer |= trip && !careful; }
                else defer |= pk2 > Ec.sat2_smooth PVDER_HYB_LEAN(|| msc * m2a > Ec.sat2_flag) || ir2a > irg;
                Y[0] = acc[0]; Y[1] = acc[1]; Y[2] = acc[2];
                start = false;
            }
        }
        // ---- the segment's next ETD3RK step (schedule row: one set index per byte, 0 = segment finished)
        const int sidx = (int)(sg.sch & 0xffull);
        bool todo = !done && sidx != 0;
        // One pass per DISTINCT set among the wave's lanes.  Lockstep (every lane that has a step to take
        // names the resident set -- always, while the envs of the wave share their schedule): ONE pass with
        // the registers as they are; the general loop reloads sets and is entered only by a wave whose envs
        // have drifted apart.  Same arithmetic either way: a lane's result does not depend on its neighbours.
        if (__builtin_amdgcn_readfirstlane(__any((int)(todo && sidx != cur))) == 0) {
            v4f Yn[3];
            bool near;
            const bool trip = lin_etd3rk<PH, !SPLIT>(G, Ac, Y, S, qref, vdcref, kpo, kio, irg, msc, Yn, trace_step == 0 && it == 1, near);
            if (todo) {
                if constexpr (SPLIT) { careful |= near; defer |= trip && !careful; } else defer |= trip || near;
                Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2];
                sg.sch = lin_sched_next(G, sg.sch);
            }
        } else {
            for (;;) {
                const unsigned long long tm = __ballot((int)todo);
                if (tm == 0) break;
                const int s_now = __builtin_amdgcn_readlane(sidx, (int)__builtin_ctzll(tm));
                if (s_now != cur) {
                    lin_load_set(G.Lk.lcoef + kLinMatFloats + (size_t)(s_now - 1) * kLinSetFloats, lane, Ac);
                    cur = s_now;
                }
                v4f Yn[3];
                bool near;
                const bool trip = lin_etd3rk<PH, !SPLIT>(G, Ac, Y, S, qref, vdcref, kpo, kio, irg, msc, Yn, false, near);
                if (todo && sidx == s_now) {
                    if constexpr (SPLIT) { careful |= near; defer |= trip && !careful; } else defer |= trip || near;
                    Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2];
                    sg.sch = lin_sched_next(G, sg.sch);
                    todo = false;
                }
            }
        }
        // ---- segment finished?  Then the next one (if any) starts at an event row change
        const bool fin = !done && !start && (sg.sch & 0xffull) == 0;
        if (__builtin_amdgcn_readfirstlane(__any((int)fin))) {
            if (fin) {
                spos += sg.seg;
                sg.until -= sg.seg;
                done = spos >= C.n_sub;
                if (!done) {                         // an event row starts here
                    sg.row += 1;
                    sg.until = sg.row < C.ev_count ? C.ev_ds : kBig;
                    const int left = C.n_sub - spos;
                    sg.seg = sg.until < left ? sg.until : left;
                    sg.in_vg = G.A.ev_vg[(size_t)sg.row * N + ec];
                    sg.in_si = G.A.ev_si[(size_t)sg.row * N + ec];
                    sg.sch = G.Lk.sched[sg.seg - 1];
                    start = true;
                }
            }
        }
    }
    return defer;
}

// =============================================================================================
// The careful pass: envs at the CURRENT LIMIT on the matrix-core path.
// Beyond |iref| = iref_limit the reference model integrates its outer PI loops conditionally -- xDC / xQ stop
// whenever they would move outward (pvder_model.hpp::rhs, the `ilim2` branch).  That logic is common to the three
// phases (iref_a is), so the balanced reduction keeps it; what it breaks is the smoothness the big ETD3RK steps
// rely on.  With s = |iref|^2 - iref_limit^2 the env is FREE (s <= 0), FROZEN (s > 0) or SLIDING along s = 0
// (free dynamics push s up, frozen dynamics pull it down), and where it enters the limit the frozen integrators
// remember the instant they stopped: a chattering explicit scheme needs 1/240 s steps for 3e-4 per env step
// (fp64 prototype, tools/limit_proto.py).  The hybrid step below takes the SAME big step up to three times with
// the outward-moving integrators advancing at alpha times their rate (bal_f<PH, true>):
//   1. alpha = 1 from a free state / 0 from a frozen one; accepted if s keeps its sign over the step;
//   2. otherwise the crossing is located (quadratic through s at the start, stage a, the end) and alpha = the
//      fraction of the step spent free; accepted if the env then stays on the far side (stuck behind the crossing);
//   3. otherwise it slides: alpha by the secant rule such that the step ends ON the limit, s = 0.
// Schedule: the careful rows of the table (n_sub .. 2 n_sub - 1): a standard env step = tick + EIGHT equal steps
// (fp64 prototype along the reference's fixed-action ramps to Q_ref = +-1.2 pu, eight envs, worst env step against the
// converged solution: four steps 3.5e-4, eight 1.1e-4; the chattering scheme with 60 steps 5.8e-4).
// =============================================================================================
template <int PH>
__device__ __forceinline__ bool lin_etd3rk_aw(const LinArgs &G, const pvder::LinSetRegs &Ac, const pvder::v4f (&Y)[3],
                                              const pvder::SubPk &S, float qref, float vdcref, v2 kpo, v2 kio, float alpha,
                                              float msc, pvder::v4f (&Yn)[3], float &ir2_mid, float &ir2_end) {
    using namespace pvder;
    const ModelK<float> &P = G.A.P;
    const EtdK &Ec = G.A.Ec;
    const PkConst K = make_pkconst(P);
    v2 yp[6], F[6];
    float m2a, ir2a, m2x, m2b, ir2b;
    lin_pairs(Y, yp);
    bal_f<PH, true>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a, alpha);
    bool fz_dc = F[3].y == 0.f, fz_q = F[4].x == 0.f;        // (alpha = 0: the outward-moving integrators came out as +-0)
    v4f acc3[3][3] = {{Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}, {Y[0], Y[1], Y[2]}};
    {
        const int m3[3] = {LM21, LM31, LM41};
        lin_mfma<3>(Ac, m3, F, acc3);
    }
    lin_pairs(acc3[0], yp);
    bal_f<PH, true>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2x, ir2_mid, alpha);
    fz_dc = fz_dc && F[3].y == 0.f; fz_q = fz_q && F[4].x == 0.f;
    v4f acc2[2][3] = {{acc3[1][0], acc3[1][1], acc3[1][2]}, {acc3[2][0], acc3[2][1], acc3[2][2]}};
    {
        const int m2[2] = {LM32, LM42};
        lin_mfma<2>(Ac, m2, F, acc2);
    }
    lin_pairs(acc2[0], yp);
    bal_f<PH, true>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2b, ir2b, alpha);
    fz_dc = fz_dc && F[3].y == 0.f; fz_q = fz_q && F[4].x == 0.f;
    Yn[0] = acc2[1][0]; Yn[1] = acc2[1][1]; Yn[2] = acc2[1][2];
    lin_mfma1(Ac[LM43], F, Yn);
    // A FROZEN outer integrator stays where it is -- exactly (round 4).  Its derivative is zero at the three stages, but the
    // f-form matrices are dense functions of hL and L couples the integrators to the currents (d xQ'/d i != 0): what the
    // currents do inside the step leaks into the "frozen" state at second order -- 1e-5 per env step along the reference's
    // Q_ref ramps, most of it in the tick behind the action -- and an env stuck behind the current limit stays there for the
    // rest of its episode: nothing ever pulls that back (x_Q 1.6e-4 off after 16 env steps, fp64 prototype; the converged
    // bridge holds both integrators to the last bit).  Frozen at all three stages: an integrator whose sign test releases it
    // inside the step moves, as the model says.  etd_linear.hybrid_step states the same.
    if (alpha == 0.f) {
        if (fz_dc) Yn[1].w = Y[1].w;
        if (fz_q) Yn[2].x = Y[2].x;
    }
    ir2_end = lin_iref2<PH>(K, S, Yn, qref, vdcref, kpo);
    const float est = fmaxf(fmaxf(fabsf(Yn[0].x - acc2[0][0].x), fabsf(Yn[0].y - acc2[0][0].y)),
                            fabsf(Yn[1].z - acc2[0][1].z));
    const v2 mn = fma2(bc2(P.KpG), mk2(Yn[1].x, Yn[1].y), mk2(Yn[0].z, Yn[0].w));
    const float mmax = msc * fmaxf(fmaxf(m2a, m2x), fmaxf(m2b, __builtin_fmaf(mn.y, mn.y, mn.x * mn.x)));
    return mmax > G.Lk.sat2_hard || est > G.Lk.est_guard;
}

// where the quadratic through (0, s0), (1/2, sm), (1, s1) crosses zero, as a fraction of the step (s0, s1 of
// opposite sign); the chord's crossing when the parabola has none inside
__device__ __forceinline__ float lin_cross_frac(float s0, float sm, float s1) {
    const float cq = 2.f * (s1 - 2.f * sm + s0), bq = s1 - s0 - cq;
    const float chord = s0 / (s0 - s1);
    const float disc = bq * bq - 4.f * cq * s0;
    float t = chord;
    if (fabsf(cq) > 1e-12f && disc >= 0.f) {
        const float r = __builtin_sqrtf(disc);
        // the numerically stable pair of roots:  q = -(b + sign(b) r)/2,  t1 = q/c,  t2 = s0/q
        const float q = -0.5f * (bq + (bq >= 0.f ? r : -r));
        const float t1 = q / cq, t2 = q != 0.f ? s0 / q : -1.f;
        if (t1 >= 0.f && t1 <= 1.f) t = t1;
        else if (t2 >= 0.f && t2 <= 1.f) t = t2;
    }
    return fminf(1.f, fmaxf(0.f, t));
}

// What a hybrid step that SLID leaves for the lane's next one (round 5): the alpha it ended with and dy+/dalpha between its last
// two evaluations.  A lane in six slides through most of its env step once the whole batch sits at the current limit
// (profiles/r05_careful_counters.txt), three evaluations a step, and a wave waits for its slowest lane: 24 evaluations per env
// step where fifty-nine lanes need eight.  The next step of such a lane starts AT the alpha the last one found: ONE evaluation,
// then along the carried derivative onto s = 0 while that moves alpha by less than kSlideThr -- otherwise a second evaluation one
// Newton step further and the secant through the two.  3 + 7 x (1 .. 2) evaluations, same 7e-5 against the converged bridge
// (etd_linear.hybrid_step states the same; fp64 prototype: tools/careful_schemes_proto.py).
struct LinSlide {
    bool on = false;
    float a = 0.f;
    pvder::v4f D[3];
};
constexpr float kSlideThr = 0.03f;

// t with s(Yb + t D) = 0, s = |iref|^2 - iref_limit^2 (quadratic in the state): slope and curvature along D from s(Yb + D),
// s(Yb - D), two Newton steps
template <int PH>
__device__ __forceinline__ float lin_slide_along(const pvder::PkConst &K, const pvder::SubPk &S, const pvder::v4f (&Yb)[3], float sb,
                                                 const pvder::v4f (&D)[3], float qref, float vdcref, v2 kpo, float lim) {
    using namespace pvder;
    v4f Yp[3], Ym[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { Yp[g] = Yb[g] + D[g]; Ym[g] = Yb[g] - D[g]; }
    const float sp = lin_iref2<PH>(K, S, Yp, qref, vdcref, kpo) - lim, sm = lin_iref2<PH>(K, S, Ym, qref, vdcref, kpo) - lim;
    const float gs = 0.5f * (sp - sm), q = 0.5f * (sp + sm) - sb;
    if (gs == 0.f) return 0.f;
    const float t = -sb / gs;
    const float d = gs + 2.f * q * t;
    return d != 0.f ? t - (sb + gs * t + q * t * t) / d : t;
}

// one hybrid step of every lane (`act`: the lanes that will commit it); Yn = y+; returns the hand-over flag
template <int PH>
__device__ __forceinline__ bool lin_hybrid_step(const LinArgs &G, const pvder::LinSetRegs &Ac, const pvder::v4f (&Y)[3],
                                                const pvder::SubPk &S, float qref, float vdcref, v2 kpo, v2 kio, bool act,
                                                float msc, LinSlide &C, pvder::v4f (&Yn)[3]) {
    using namespace pvder;
    const float lim = G.A.P.ilim2;
    const PkConst K = make_pkconst(G.A.P);
    const float s0 = lin_iref2<PH>(K, S, Y, qref, vdcref, kpo) - lim;
    const bool from_free = s0 <= 0.f;
    const bool cont = act && C.on;               // behind a step that slid: from where that one ended
    const float a1 = cont ? C.a : (from_free ? 1.f : 0.f);
    float irm, ire;
    bool trip = lin_etd3rk_aw<PH>(G, Ac, Y, S, qref, vdcref, kpo, kio, a1, msc, Yn, irm, ire);
    const float s1 = ire - lim;
    PVDER_DBG_COUNT(6, act);
    float b = a1;
    bool two = false;
    if (__builtin_amdgcn_readfirstlane(__any((int)cont))) {
        const float t = lin_slide_along<PH>(K, S, Yn, s1, C.D, qref, vdcref, kpo, lim);
        const float at = a1 + t;
        if (cont) {
            if (fabsf(t) < kSlideThr && at >= 0.f && at <= 1.f) {
#pragma unroll
                for (int g = 0; g < 3; ++g) Yn[g] = Yn[g] + t * C.D[g];
                C.a = at;
            } else {
                b = fminf(1.f, fmaxf(0.f, at));
                two = b != a1;
                if (!two) C.on = false;            // (pinned at an end of [0, 1] and asked to go further: the step as it is)
            }
        }
    }
    const bool need2 = act && !cont && (from_free ? s1 > 0.f : s1 < 0.f);
    PVDER_DBG_COUNT(7, need2 || two);
    if (__builtin_amdgcn_readfirstlane(__any((int)(need2 || two)))) {
        // the limit is crossed inside the step: the integrators run free for the fraction `tau` (1 - tau) of it
        const float tau = lin_cross_frac(s0, irm - lim, s1);
        const float a2 = need2 ? (from_free ? tau : 1.f - tau) : b;
        v4f Y2[3];
        const bool trip2 = lin_etd3rk_aw<PH>(G, Ac, Y, S, qref, vdcref, kpo, kio, a2, msc, Y2, irm, ire);
        const float s2 = ire - lim;
        if (__builtin_amdgcn_readfirstlane(__any((int)two))) {
            // a continued slide, second evaluation: the slide ends inside this step (free / frozen integrators are what holds from
            // here on), or the secant through the two
            v4f Dn[3];
            const float rd = two ? 1.f / (b - a1) : 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) Dn[g] = (Y2[g] - Yn[g]) * rd;
            float t = lin_slide_along<PH>(K, S, Y2, s2, Dn, qref, vdcref, kpo, lim);
            if (two) {
                const bool ends = (b == 1.f && s2 <= 0.f) || (b == 0.f && s2 >= 0.f);
                t = (ends || fabsf(t) > 1.f) ? 0.f : t;
                float bt = b + t;
                const bool leaves = bt < 0.f || bt > 1.f;
                if (leaves) { bt = fminf(1.f, fmaxf(0.f, bt)); t = bt - b; }
#pragma unroll
                for (int g = 0; g < 3; ++g) { Yn[g] = Y2[g] + t * Dn[g]; C.D[g] = Dn[g]; }
                C.a = bt;
                C.on = !(ends || leaves);
                trip = trip2;
            }
        }
        if (need2) { Yn[0] = Y2[0]; Yn[1] = Y2[1]; Yn[2] = Y2[2]; trip = trip2; }
        const bool need3 = need2 && (from_free ? s2 < 0.f : s2 > 0.f);
        PVDER_DBG_COUNT(8, need3);
        if (__builtin_amdgcn_readfirstlane(__any((int)need3))) {
            // it does not stay on the far side: sliding along the limit -- the step ends on s = 0 (secant between
            // the two evaluations that bracket it)
            const float a3 = need3 ? (from_free ? a2 - s2 * (1.f - a2) / (s1 - s2) : a2 * s1 / (s1 - s2)) : a2;
            const bool trip3 = lin_etd3rk_aw<PH>(G, Ac, Y, S, qref, vdcref, kpo, kio, a3, msc, Y2, irm, ire);
            if (need3) {
                // the third evaluation lands NEAR the limit (the secant between two evaluations far apart); sliding means ON
                // it.  Between two close evaluations the state is all but linear in alpha: one more secant step through
                // (a2, s2), (a3, s3), taken on the STATES -- no further evaluation (round 4: the converged bridge slides at
                // |iref| = iref_limit to 1e-8, these steps used to end 1.6e-4 above it, 1.4e-4 in the observations of that
                // env step; etd_linear.hybrid_step states the same)
                const float s3 = ire - lim;
                float w = -s3 / (s3 - s2);
                const bool ok = s3 != s2 && fabsf(a3 - a2) > 1e-4f && fabsf(w) <= 1.f;
                w = ok ? w : 0.f;
                // ... and what the lane's next step starts from
                const float as = a3 + w * (a3 - a2);
                C.on = ok && as >= 0.f && as <= 1.f;
                C.a = as;
                const float rd = ok ? 1.f / (a3 - a2) : 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const v4f dY = Y2[g] - Yn[g];
                    C.D[g] = dY * rd;
                    Yn[g] = Y2[g] + w * dY;
                }
                trip = trip3;
            }
        }
    }
    return trip;
}

// The env step of the lanes with `run` set (the careful lanes of lin_integrate), from the state they started the
// env step with.  Same frame as lin_integrate's general loop -- per-lane schedule rows (the careful ones), one pass
// per distinct coefficient set, wave-uniform trip counts, a lane's arithmetic independent of its neighbours.
template <int PH>
__device__ __forceinline__ bool lin_integrate_careful(const LinArgs &G, int lane, LinRegs &R, pvder::v4f (&Y)[3], float qref,
                                                      float vdcref, int ec, LinSeg sg, bool run, bool defer, float msc) {
    using namespace pvder;
    lin_need_t2(G, lane, R);
    const LinMatRegs &At = R.At;
    LinSetRegs &Ac = R.Ac;
    int &cur = R.cur;
    constexpr int kBig = 0x3fffffff;
    const ModelK<float> &P = G.A.P;
    const StepK<float> &C = G.A.C;
    const EtdK &Ec = G.A.Ec;
    const size_t N = (size_t)G.A.n_env;
    const PkConst K = make_pkconst(P);
    float g_kpdc = P.KpDC, g_kpq = -P.KpQ, g_kidc = P.KiDC, g_kiq = -P.KiQ;
    asm volatile("" : "+s"(g_kpdc), "+s"(g_kpq), "+s"(g_kidc), "+s"(g_kiq));
    const v2 kpo = mk2(g_kpdc, g_kpq), kio = mk2(g_kidc, g_kiq);
    int spos = 0;
    bool start = true;
    bool done = !run;
    SubPk S;
    LinSlide Cs;
#pragma unroll
    for (int g = 0; g < 3; ++g) Cs.D[g] = v4f{0.f, 0.f, 0.f, 0.f};
    cur = __builtin_amdgcn_readfirstlane(cur);
    sg.sch = G.Lk.sched[C.n_sub + sg.seg - 1];           // the careful row of the first segment
    for (;;) {
        if (!__builtin_amdgcn_readfirstlane(__any((int)(!done && !defer)))) break;
        if (__builtin_amdgcn_readfirstlane(__any((int)(start && !done)))) {
            // the tick behind the jump: frozen or free by the side of the limit the env starts on
            v2 yp[6], F[6];
            float m2a, ir2a;
            make_sub_pk(P, sg.in_vg, sg.in_si, S);
            lin_pairs(Y, yp);
            bal_f<PH, true>(P, K, S, yp, qref, vdcref, kpo, kio, F, m2a, ir2a, -1.f);
            v4f acc[3] = {Y[0], Y[1], Y[2]};
            lin_mfma1(At, F, acc);
            if (ir2a > P.ilim2) {                  // frozen integrators are HELD over the tick too (see lin_etd3rk_aw):
                if (F[3].y == 0.f) acc[1].w = Y[1].w;     // the jump kicks the current loop, and h phi1(hL) is dense
                if (F[4].x == 0.f) acc[2].x = Y[2].x;
            }
            const bool trip = msc * lin_tick_peak2(P, Y, acc, m2a) > G.Lk.sat2_hard;
            PVDER_DBG_COUNT(4, start && !done && !defer && trip);
            if (start && !done) {
                defer |= trip;
                Y[0] = acc[0]; Y[1] = acc[1]; Y[2] = acc[2];
                start = false;
                Cs.on = false;                     // (the jump moves |iref|: whatever slid starts over)
            }
        }
        const int sidx = (int)(sg.sch & 0xffull);
        bool todo = !done && !defer && sidx != 0;
        for (;;) {
            const unsigned long long tm = __ballot((int)todo);
            if (tm == 0) break;
            const int s_now = __builtin_amdgcn_readlane(sidx, (int)__builtin_ctzll(tm));
            if (s_now != cur) {
                lin_load_set(G.Lk.lcoef + kLinMatFloats + (size_t)(s_now - 1) * kLinSetFloats, lane, Ac);
                cur = s_now;
            }
            v4f Yn[3];
            const bool mine = todo && sidx == s_now;
            const bool trip = lin_hybrid_step<PH>(G, Ac, Y, S, qref, vdcref, kpo, kio, mine, msc, Cs, Yn);
            PVDER_DBG_COUNT(5, mine && trip);
            if (mine) {
                defer |= trip;
                Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2];
                sg.sch = lin_sched_next(G, sg.sch);
                todo = false;
            }
        }
        const bool fin = !done && !start && (sg.sch & 0xffull) == 0;
        if (__builtin_amdgcn_readfirstlane(__any((int)fin))) {
            if (fin) {
                spos += sg.seg;
                sg.until -= sg.seg;
                done = spos >= C.n_sub;
                if (!done) {
                    sg.row += 1;
                    sg.until = sg.row < C.ev_count ? C.ev_ds : kBig;
                    const int left = C.n_sub - spos;
                    sg.seg = sg.until < left ? sg.until : left;
                    sg.in_vg = G.A.ev_vg[(size_t)sg.row * N + ec];
                    sg.in_si = G.A.ev_si[(size_t)sg.row * N + ec];
                    sg.sch = G.Lk.sched[C.n_sub + sg.seg - 1];
                    start = true;
                }
            }
        }
    }
    return defer;
}

// One env transition of the workgroup's envs through memory -- the body of step_etdlin_kernel, also
// run step after step by the rollout's finish kernel.  act / out_off select the action row and the
// output block of the step (rollouts); the kernel arguments are re-read through laundered copies of
// the kernarg pointer where keeping them live across the step loop would not fit the scalar
// register file (hipcc then spills SGPRs and gives the kernel a private segment).
// KOFF: byte offset of G in the kernel-argument segment; fstate / istate / n_env: the copies the caller
// holds (step_etdlin_kernel: preloaded into SGPRs by the dispatcher, see there).
template <int PH, int KOFF, bool T3 = false>
__device__ __forceinline__ void lin_step_body(const LinArgs &G, int lane, LinRegs &R,
                                              const uint8_t *__restrict__ act, long long out_off, int stats_fresh,
                                              const float *__restrict__ fstate, const int32_t *__restrict__ istate, int n_env,
                                              bool only = true) {
    using namespace pvder;
    const EtdArgs &A = G.A;
    constexpr int NS = 6 * PH + 5;
    const size_t N = (size_t)n_env;
    typedef __attribute__((address_space(4))) const char *KernargBytes;
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(A.C, fstate, istate, act, n_env, L, ys);
    // `only`: the lanes this pass advances (step_etdlin_kernel: the envs its lean first pass, lin_step_lean, left alone;
    // everyone else of the workgroup has stored its result -- and been counted -- already)
    if (!only) { L.run = false; L.live = false; L.bad_action = 0.f; }
    PVDER_TRACE_COLD_AT(8);
    PVDER_TRACE_COLD_CYC(26);
    // ksub is the first state word to arrive: the event inputs and the schedule row of the first segment
    // are requested while the other state rows are still in flight
    const LinSeg sg = lin_first_segment(A.C, G.Lk, L.ksub, A.ev_vg, A.ev_si, N, L.ec);
    // ... and so are the inputs in force at tStop (the observation is taken there): the row the first segment
    // runs on, the next one if it changes exactly at tStop, a second division only if an event cuts the env step
    int row_stop = sg.row;
    if (sg.until == A.C.n_sub) row_stop = sg.row + 1 < A.C.ev_count ? sg.row + 1 : A.C.ev_count;
    else if (sg.until < A.C.n_sub) row_stop = event_row(L.ksub + A.C.n_sub, A.C.ev_s0, A.C.ev_ds, A.C.ev_count);
    const float vg_stop = A.ev_vg[(size_t)row_stop * N + L.ec], si_stop = A.ev_si[(size_t)row_stop * N + L.ec];
    PVDER_TRACE_AT(1);
    if constexpr (PH == 3) {
        // an env whose duty cycle is beyond 98.75 % of its limit: the plan's hybrid flag (a word of pinned HOST memory; the next step
        // launches are followed by step_sat_hybrid_kernel -- in place well before an env saturates: sat2_flag)
        if (A.hyb_flag && L.run) {
            float m2 = 0.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float mr = __builtin_fmaf(A.P.KpG, ys[6 * p + 4], ys[6 * p + 2]), mi = __builtin_fmaf(A.P.KpG, ys[6 * p + 5], ys[6 * p + 3]);
                m2 = fmaxf(m2, __builtin_fmaf(mi, mi, mr * mr));
            }
            if (m2 > A.Ec.sat2_flag) *(volatile int *)A.hyb_flag = 1;
        }
    }
    bool defer = PH == 3 && A.force_full != 0;
    bool seq = false;                   // off balance, carried by its positive-sequence component (lin_seq_split)
    float msc = 1.f;
    if constexpr (PH == 3) {
        const bool unb = L.run && !defer && lin_unbalanced(ys);
        if (__builtin_amdgcn_readfirstlane(__any((int)unb))) {
            // (rare: the propagator is read from the kernel-argument segment right here, not kept in SGPRs for the launch)
            KernargPtr kq = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
            asm volatile("" : "+s"(kq));
            if (unb) seq = lin_seq_split(A.P, ((const LinArgs *)kq)->Lk, ys, msc);
            defer |= unb && !seq;
        }
    }
    {
        // every lane enters (a lane that does not run is `done` from the start): the coefficient registers are
        // packed ACROSS the lanes of the wave -- a set reloaded inside the loop must be loaded by all 64
        v4f Y[3];
        lin_from_rows<PH>(ys, Y);
        PVDER_TRACE_AT(2);
        bool careful;
        const bool defer0 = defer;          // (off balance / forced: the 23-state path whatever the standard pass finds)
        defer = lin_integrate<PH>(G, lane, R, Y, L.qref, L.vdcref, L.ec, sg, L.run, defer, careful, 0, msc);
        PVDER_DBG_COUNT(0, L.run);
        PVDER_DBG_COUNT(1, L.run && defer && !careful);
        PVDER_DBG_COUNT(9, L.run && defer && careful);
        // an env the standard pass gave up on in one step AND flagged for the careful pass in a later one (it crossed the
        // 0.9 |iref| / 0.998 |m| threshold inside the env step: what tripped was the standard pass's own coarse, limit-blind
        // step) belongs to the careful pass, which decides with its own guards
        careful = careful && L.run && !defer0;
        defer = defer0 || (defer && !careful);
        PVDER_TRACE_COLD_AT(9);
        PVDER_DBG_COUNT(2, careful);
        if (__builtin_amdgcn_readfirstlane(__any((int)careful))) {
            // envs at the current limit: their env step again, from the state it started with, by the careful pass
            v4f Yc[3];
            lin_from_rows<PH>(ys, Yc);
            if (careful) { Y[0] = Yc[0]; Y[1] = Yc[1]; Y[2] = Yc[2]; }
            const bool dc = lin_integrate_careful<PH>(G, lane, R, Y, L.qref, L.vdcref, L.ec, sg, careful, false, msc);
            defer |= careful && dc;
            PVDER_DBG_COUNT(3, careful && dc);
        }
        PVDER_TRACE_COLD_AT(10);
        PVDER_TRACE_COLD_CYC(27);
        if (L.run && !defer) {
            lin_to_rows<PH>(Y, ys);
            L.ksub += A.C.n_sub;
            L.steps += 1;
        }
        if constexpr (PH == 3) {
            seq = seq && L.run && !defer;
            PVDER_DBG_COUNT(12, seq);
            if (__builtin_amdgcn_readfirstlane(__any((int)seq))) {
                KernargPtr kq = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
                asm volatile("" : "+s"(kq));
                if (seq) lin_seq_merge(((const LinArgs *)kq)->Lk, fstate, N, L.e, ys);
            }
        }
    }
    PVDER_TRACE_AT(20);
    KernargPtr ka = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
    PVDER_LAUNDER_EPI(ka);
    const EtdArgs &E = *(const EtdArgs *)ka;
    if (PH == 1 && defer && E.C.off_range == PVDER_OFF_RANGE_TERMINATE) {
        L.fail = L.run;     // single phase: the hand-over would be classic RK4 for the whole env step -- the episode ends instead
    } else if (defer) {
        L.run = false;
        L.live = false;
    }
    epilogue<float, PH>(E.P, E.C, E.R, L, ys, E.fstate, E.istate, E.ev_vg, E.ev_si, E.out + out_off, E.stats, E.n_env,
                        stats_fresh, true, vg_stop, si_stop);
    PVDER_TRACE_AT(21);
    {
        // the deferred envs of this workgroup (none in PVDER-v0 rollouts) take the 23-state path (three-phase) / classic
        // RK4 micro-steps (single phase) right here -- no second launch
        defer = defer && !L.fail && only;
        PVDER_DBG_COUNT(10, defer);
        if (__syncthreads_or((int)defer)) {
            PVDER_DBG_COUNT(11, threadIdx.x == 0);
            KernargPtr kb = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
            asm volatile("" : "+s"(kb));
            PVDER_TRACE_COLD_AT(11);
            if constexpr (PH == 3) etdrk4_full_tail<false, T3>(*(const EtdArgs *)kb, defer, act, out_off);
            else rk4_tail_1ph(*(const EtdArgs *)kb, defer, act, out_off);
            PVDER_TRACE_COLD_AT(12);
        }
    }
}

// The leading scalar arguments -- what the first loads of a launch need -- are PRELOADED into SGPRs by the
// dispatcher (-mllvm -amdgpu-kernarg-preload-count: gfx950 kernel-argument preload; the code object keeps
// the s_load prologue for firmware without it): the wave does not sit out a scalar-memory round trip
// (~0.3 us from kernel entry, tools/trace_timeline.py slot 18) before it can issue its first vector load.
// They repeat G.A.fstate / .istate / .action / G.Lk.lcoef / G.A.n_env; G itself follows at kLinHeadBytes.
struct LinHead {
    float *fstate;
    int32_t *istate;
    const uint8_t *action;
    const float *lcoef;
    int n_env, ksub_hint;            // ksub_hint: -1 = none; else the clock, | kLinHintQuiet when the env step that starts there is a QUIET one
    const float *k4coef;
};
constexpr int kLinHintQuiet = 1 << 30;
constexpr int kLinHeadBytes = (int)sizeof(LinHead);
// lin_step_body / the 23-state tail re-read G (and G.A at its start) from the kernel-argument segment at this byte
// offset: the by-value LinArgs must follow the head without padding, and the blocks the tails re-read sit at offset 0
static_assert(alignof(LinArgs) <= 8 && sizeof(LinHead) % alignof(LinArgs) == 0, "LinArgs must start right behind LinHead");
static_assert(offsetof(LinArgs, A) == 0, "the 23-state tail reads EtdArgs at the start of LinArgs");

// The LEAN first pass of step_etdlin_kernel: the env transition of every env that stays on the standard path -- balanced,
// clear of the current limit and of the duty-cycle guards: every env of a PVDER-v0 rollout under a random or a
// learning policy -- with nothing else in its live ranges.  It is the step of the fused rollout kernel (lin_integrate
// with ONE hand-over flag), through memory.  An env that needs anything more (off balance: symmetrical components; at
// the current limit: the careful pass; at the duty-cycle limit: the 23-state path) is left untouched -- nothing of it is
// stored or counted -- and the workgroup that holds one takes it through lin_step_body from the state still in memory
// (round 3 ran that body for everyone: the careful flag, the guard scale of the symmetrical-component carry and the 23
// state rows kept for the careful pass's restart rode along on the standard path -- 8.3 -> 8.8 us per step).  Same
// arithmetic for the envs that stay (msc = 1 is exact), same body for the ones that leave: bit-identical results.
template <int PH, int KOFF>
__device__ __forceinline__ bool lin_step_lean(const LinArgs &G, int lane, LinRegs &R,
                                              const uint8_t *__restrict__ act, int stats_fresh,
                                              const float *__restrict__ fstate, const int32_t *__restrict__ istate, int n_env,
                                              int ksub_hint) {
    using namespace pvder;
    const EtdArgs &A = G.A;
    constexpr int NS = 6 * PH + 5;
    const size_t N = (size_t)n_env;
    typedef __attribute__((address_space(4))) const char *KernargBytes;
    Lane<float> L;
    float ys[NS];
    // ---- the event inputs, SPECULATIVELY.  What a step reads from the event tables depends on the env's clock
    // (ksub -> event row): a dependent fetch behind the state rows, one more memory round trip (0.6-0.8 us with 1024 waves
    // asking at once) before the first flop.  The caller tells where the envs of the batch ARE (ksub_hint: a kernel
    // argument, there with the wave; pvder_plan_hint -- envs stepped in lockstep since a common reset, the benchmark and
    // every vectorised training loop), the rows that clock names are requested with the state rows, and a lane whose own
    // clock says otherwise (an env reset on its own) falls back to the dependent fetch -- with its whole wave, same values:
    // a wrong or missing hint (-1) costs time, never a result.
    const int ec0 = min((int)(blockIdx.x * kBlock + threadIdx.x), n_env - 1);
    LinSeg sg;
    float vg_stop = 0.f, si_stop = 0.f;
    const bool hinted = ksub_hint >= 0;
    if (hinted) {
        sg = lin_first_segment(A.C, G.Lk, ksub_hint, A.ev_vg, A.ev_si, N, ec0);
        int row_stop = sg.row;                 // (the inputs in force at tStop: see lin_step_body)
        if (sg.until == A.C.n_sub) row_stop = sg.row + 1 < A.C.ev_count ? sg.row + 1 : A.C.ev_count;
        else if (sg.until < A.C.n_sub) row_stop = event_row(ksub_hint + A.C.n_sub, A.C.ev_s0, A.C.ev_ds, A.C.ev_count);
        vg_stop = A.ev_vg[(size_t)row_stop * N + ec0]; si_stop = A.ev_si[(size_t)row_stop * N + ec0];
    }
    // (pvder_autoreset.lazy_events: the table rows this step leaves behind for its successor -- rows gen_j0+1 .. gen_j1, one step in
    //  four at PVDER-v0's grid -- and the episode word their hash is keyed with, asked for here with everything else)
    int gen_j0 = -1, gen_j1 = -1, episode_now = -1;
    auto rows_to_leave = [&](const LinSeg &s0, int ec) {
        if (!A.R.lazy) return;
        const int n = A.C.n_sub, K = A.C.ev_count;
        auto within = [&](int span) { return s0.until <= span ? 1 + (span - s0.until >= A.C.ev_ds ? (span - s0.until) / A.C.ev_ds : 0) : 0; };
        gen_j0 = min(s0.row + within(n), K); gen_j1 = min(s0.row + within(2 * n), K);
        if (gen_j1 > gen_j0) episode_now = istate[PVDER_I_EPISODE * N + ec];
    };
    if (hinted) rows_to_leave(sg, ec0);
    prologue<float, NS>(A.C, fstate, istate, act, n_env, L, ys);
    if (__builtin_amdgcn_readfirstlane(__any((int)(!hinted || (L.run && L.ksub != ksub_hint))))) {
        sg = lin_first_segment(A.C, G.Lk, L.ksub, A.ev_vg, A.ev_si, N, L.ec);
        int row_stop = sg.row;
        if (sg.until == A.C.n_sub) row_stop = sg.row + 1 < A.C.ev_count ? sg.row + 1 : A.C.ev_count;
        else if (sg.until < A.C.n_sub) row_stop = event_row(L.ksub + A.C.n_sub, A.C.ev_s0, A.C.ev_ds, A.C.ev_count);
        vg_stop = A.ev_vg[(size_t)row_stop * N + L.ec]; si_stop = A.ev_si[(size_t)row_stop * N + L.ec];
        episode_now = -1;
        rows_to_leave(sg, L.ec);
    }
    PVDER_TRACE_AT(1);
    bool leave = PH == 3 && A.force_full != 0;
    if constexpr (PH == 3) leave |= lin_unbalanced(ys);
    leave = leave && L.run;
    {
        v4f Y[3];
        lin_from_rows<PH>(ys, Y);
        PVDER_TRACE_AT(2);
        bool careful;
        leave = lin_integrate<PH, false>(G, lane, R, Y, L.qref, L.vdcref, L.ec, sg, L.run, leave, careful, 0) && L.run;
        if (L.run && !leave) {
            lin_to_rows<PH>(Y, ys);
            L.ksub += A.C.n_sub;
            L.steps += 1;
        }
    }
    PVDER_TRACE_AT(20);
    KernargPtr ka = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
    PVDER_LAUNDER_EPI(ka);
    const EtdArgs &E = *(const EtdArgs *)ka;
    if (leave) { L.run = false; L.live = false; }
    // (a use the compiler cannot move: the episode word was asked for at the START of the launch and stays asked for there --
    //  left alone, its load sinks into the one block that reads it, behind the integration: a memory round trip per generated row)
    asm volatile("" : "+v"(episode_now));
    epilogue<float, PH>(E.P, E.C, E.R, L, ys, E.fstate, E.istate, E.ev_vg, E.ev_si, E.out, E.stats, E.n_env, stats_fresh, true,
                        vg_stop, si_stop, nullptr, false, gen_j0, gen_j1, episode_now);
    PVDER_TRACE_AT(21);
    return leave;
}

// CHAIN (A/B experiment of round 5; PVDER_CHAIN=1): launches of consecutive env steps on two alternating streams,
// ordered per WORKGROUP instead of per launch -- workgroup j of step t+1 reads only what workgroup j of step t wrote, so it may
// load its kernel arguments and coefficient registers while step t is still running and waits (bounded) for chain[j] >= seq
// before it touches the state; at its end it publishes chain[j] = seq + 1 (release, agent scope: the two workgroups may sit on
// different XCDs).  A wait that times out sets chain[gridDim.x] (a flagged wrong answer, never a hung GPU).
constexpr int kChainSpinLimit = 1 << 22;            // x ~0.1 us of s_sleep: ~0.4 s
template <int PH, bool T3, bool CHAIN = false>
__device__ __forceinline__ void step_etdlin_entry(float *fstate, int32_t *istate, const uint8_t *action, const float *lcoef, int n_env,
                                                  int ksub_word, const float *k4coef, const LinArgs &G, int32_t *chain = nullptr,
                                                  int seq = 0) {
    using namespace pvder;
    PVDER_TRACE_AT(0);
    // ---- A operands of the matrix-core products (pvder_etdlin.hpp): the tick matrix and the launch's
    // first ETD3RK set stay in registers for the whole launch; another set replaces the latter only when
    // an env needs it (an env off the event grid).  Issued FIRST (they depend on nothing but the kernel
    // arguments), the state rows right behind them: one memory latency.
    const int lane = threadIdx.x & 63;
    // (round 4) ... or, where the plan expects a QUIET env step -- no event row change at its start: three of four PVDER-v0
    // steps --, the Krogstad block alone (lin_k4); what a lane needs beyond that is loaded where it is needed
    const int ksub_hint = ksub_word < 0 ? -1 : (ksub_word & (kLinHintQuiet - 1));
    LinRegs R;
    PVDER_TRACE_DEP(18, (float)n_env);                  // (trace builds: the leading kernel arguments are there)
    if (ksub_word >= 0 && (ksub_word & kLinHintQuiet) != 0 && k4coef != nullptr) {
        lin_load_k4(k4coef, lane, R.Ak);
        R.k4_ok = true; R.at_ok = false; R.cur = 0;
        PVDER_TRACE_DEP(19, R.Ak[0][0] + R.Ak[9][2]);
    } else {
        lin_load_mat(lcoef, lane, R.At);
        lin_load_set(lcoef + kLinMatFloats, lane, R.Ac);
        R.k4_ok = false; R.at_ok = true; R.cur = 1;
        PVDER_TRACE_DEP(19, R.At[0] + R.Ac[5][2]);       // (... the coefficient registers are loaded)
    }
    if constexpr (CHAIN) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&chain[blockIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > kChainSpinLimit) { __hip_atomic_store(&chain[gridDim.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        // (the acquire of thread 0 invalidated the caches this workgroup reads through -- its CU's L1, its XCD's L2: one
        //  invalidation per workgroup, not per wave -- PVDER_CHAIN_FENCE_ALL=1 at build time: every wave, +8 us per step)
#ifdef PVDER_CHAIN_FENCE_ALL
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
#ifdef PVDER_NO_LEAN
    lin_step_body<PH, kLinHeadBytes, T3>(G, lane, R, action, 0, G.A.stats_fresh, fstate, istate, n_env);
#else
    // (single phase: the 10 kVA unit of model_1 sits within 5 % of its current limit as soon as Q_ref moves -- its env steps ARE
    //  careful-pass steps, and a lean first pass in front of them is 8 us of 42 spent on results nobody uses: the body at once)
    if constexpr (PH == 1) {
        lin_step_body<PH, kLinHeadBytes, T3>(G, lane, R, action, 0, G.A.stats_fresh, fstate, istate, n_env);
        return;
    }
    const bool leave = lin_step_lean<PH, kLinHeadBytes>(G, lane, R, action, G.A.stats_fresh, fstate, istate, n_env, ksub_hint);
    if (__syncthreads_or((int)leave)) {
        // (rare: every argument again through a laundered copy of the kernarg pointer -- nothing of this pass stays
        // live across the lean one)
        typedef __attribute__((address_space(4))) const char *KernargBytes;
        typedef __attribute__((address_space(4))) const LinArgs *LinKernargPtr;
        LinKernargPtr kc = (LinKernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + kLinHeadBytes);
        asm volatile("" : "+s"(kc));
        const LinArgs &Gc = *(const LinArgs *)kc;
        LinRegs Rc;                      // (its coefficient registers too: loaded where the pass needs them)
        Rc.at_ok = false; Rc.cur = 0; Rc.k4_ok = false;
        lin_step_body<PH, kLinHeadBytes, T3>(Gc, lane, Rc, Gc.A.action, 0, 0, Gc.A.fstate, Gc.A.istate, Gc.A.n_env, leave);
    }
#endif
    if constexpr (CHAIN) {
#ifdef PVDER_CHAIN_FENCE_ALL
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // every wave: its rows are out before the workgroup says so
#else
        __builtin_amdgcn_s_waitcnt(0);                            // every wave: its stores have left the CU (one write-back per workgroup below)
#endif
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&chain[blockIdx.x], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Two kernels, one body.  step_etdlin_kernel: two waves per SIMD (256 VGPRs) -- the batches beyond 65 536 envs live on that.
// step_etdlin_kernel_w1: ONE wave per SIMD, which is all a batch of up to 65 536 envs (1024 waves) gives the chip anyway -- the
// register file of the second wave holds what the two forms of the 23-state tail need side by side (T3: etdrk4_full_tail).
// launch_plan picks by batch size; same arithmetic, same results.
template <int PH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void step_etdlin_kernel(float *fstate, int32_t *istate, const uint8_t *action,
                                                             const float *lcoef, int n_env, int ksub_word, const float *k4coef, const LinArgs G) {
    step_etdlin_entry<PH, false>(fstate, istate, action, lcoef, n_env, ksub_word, k4coef, G);
}
template <int PH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_etdlin_kernel_w1(float *fstate, int32_t *istate, const uint8_t *action,
                                                             const float *lcoef, int n_env, int ksub_word, const float *k4coef, const LinArgs G) {
    step_etdlin_entry<PH, PH == 3>(fstate, istate, action, lcoef, n_env, ksub_word, k4coef, G);
}
#ifdef PVDER_CHAIN_EXPERIMENT
// (A/B, -DPVDER_CHAIN_EXPERIMENT builds + PVDER_CHAIN=1: see step_etdlin_entry; NOT in the shipped library -- measured and not kept,
//  profiles/r05_ab_workgroup_chained_launches.txt) the two-waves-per-SIMD kernel, so that a workgroup of step t+1 finds room beside one of step t
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void step_etdlin_kernel_chain(float *fstate, int32_t *istate, const uint8_t *action,
                                                             const float *lcoef, int n_env, int ksub_word, const float *k4coef, const LinArgs G,
                                                             int32_t *chain, int seq) {
    step_etdlin_entry<3, false, true>(fstate, istate, action, lcoef, n_env, ksub_word, k4coef, G, chain, seq);
}
#endif
// ---- per-LANE parameter sets on the matrix-core path (round 5; pvder_plan_lane_params on an etdlin plan) ------------------------
// Domain randomisation (an optional per-env override of config_der.json:2-21) went through the 23-state ETDRK4 lane
// kernels: 48-78 us per batched step against 7.8.  The exponential integrator does not need the env's own operator: f = L y + N(y)
// holds for ANY L, the f-form steps evaluate f alone -- with the env's OWN constants (the seven folded rows of lane_par: 1/C, the
// outer PI gains, the PV array) -- and L, the GROUP's Jacobian at its operating point, stays the one shared coefficient block.  What
// the mismatch costs is accuracy per step (the outer loops' part of N is no longer small): TWO Krogstad steps per env step instead
// of one hold 6e-5 of the converged per-set bridge on the 64-set golden (C x 0.7-1.4, gains x 0.8-1.25, 10 / 11 strings; one step:
// 2.7e-4, the env's own operator: 1.7e-5 -- tools/lane_etdlin_proto.py); n_k4 = 6 holds 5e-5 at C x 0.5-2 / gains x 0.67-1.5, 8
// hold 8e-5 at C x 0.33-3 / gains x 0.5-2 (tools/lane_etdlin_envelope.py).  k4lane: two Krogstad blocks behind the plan's own
// (pvder_lin_coef.k4 & 2): a quiet env step / n_k4, and what a tick leaves of an env step that starts at an event row / n_k4.
// An env that needs more than that -- off balance, near the current or the duty-cycle limit, an event row change INSIDE its env
// step -- takes the 23-state lane path in the same launch (etdrk4_full_tail<LP>), as the lockstep kernels do.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void step_etdlin_lane_kernel(const LinArgs G, const float *__restrict__ lane_par,
                                                                                                          const float *__restrict__ k4lane, int n_k4) {
    using namespace pvder;
    const EtdArgs &A = G.A;
    constexpr int NS = 23;
    const size_t N = (size_t)A.n_env;
    const int lane = threadIdx.x & 63;
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(A.C, A.fstate, A.istate, A.action, A.n_env, L, ys);
    ModelK<float> Pl = A.P;
    {
        const float *__restrict__ lp = lane_par + L.ec;
        Pl.iC = lp[(size_t)PVDER_LPF_IC * N];
        Pl.KpDC = lp[(size_t)PVDER_LPF_KP_DC * N]; Pl.KiDC = lp[(size_t)PVDER_LPF_KI_DC * N];
        Pl.KpQ = lp[(size_t)PVDER_LPF_KP_Q * N];   Pl.KiQ = lp[(size_t)PVDER_LPF_KI_Q * N];
        Pl.pvA = lp[(size_t)PVDER_LPF_PVA * N];    Pl.pvB = lp[(size_t)PVDER_LPF_PVB * N];
    }
    const LinSeg sg = lin_first_segment(A.C, G.Lk, L.ksub, A.ev_vg, A.ev_si, N, L.ec);
    int row_stop = sg.row;                 // (the inputs in force at tStop: see lin_step_body)
    if (sg.until == A.C.n_sub) row_stop = sg.row + 1 < A.C.ev_count ? sg.row + 1 : A.C.ev_count;
    else if (sg.until < A.C.n_sub) row_stop = event_row(L.ksub + A.C.n_sub, A.C.ev_s0, A.C.ev_ds, A.C.ev_count);
    const float vg_stop = A.ev_vg[(size_t)row_stop * N + L.ec], si_stop = A.ev_si[(size_t)row_stop * N + L.ec];
    // one segment per env step here; anything else, and whatever the guards below find, is the 23-state path's
    bool leave = (A.force_full != 0 || lin_unbalanced(ys) || sg.seg < A.C.n_sub) && L.run;
    {
        v4f Y[3];
        lin_from_rows<3>(ys, Y);
        const v2 kpo = mk2(Pl.KpDC, -Pl.KpQ), kio = mk2(Pl.KiDC, -Pl.KiQ);
        const float irg = G.Lk.ir2_careful;
        SubPk S;
        make_sub_pk(Pl, sg.in_vg, sg.in_si, S);
        LinK4Regs Ak;
#pragma unroll 1
        for (int kind = 0; kind < 2; ++kind) {          // 0: quiet env steps; 1: env steps that start at an event row (tick first)
            const bool mine = L.run && !leave && (sg.at_event ? kind == 1 : kind == 0);
            if (!__builtin_amdgcn_readfirstlane(__any((int)mine))) continue;
            lin_load_k4(k4lane + (size_t)kind * kLinK4Floats, lane, Ak);
            if (kind == 1) {
                LinMatRegs At;
                lin_load_mat(G.Lk.lcoef, lane, At);
                const PkConst K = make_pkconst(Pl);
                v2 yp[6], F[6];
                float m2a, ir2a;
                lin_pairs(Y, yp);
                bal_f<3>(Pl, K, S, yp, L.qref, L.vdcref, kpo, kio, F, m2a, ir2a);
                v4f acc[3] = {Y[0], Y[1], Y[2]};
                lin_mfma1(At, F, acc);
                const float pk2 = lin_tick_peak2(Pl, Y, acc, m2a);
                if (mine) {
                    leave |= pk2 > A.Ec.sat2_smooth PVDER_HYB_LEAN(|| m2a > A.Ec.sat2_flag) || ir2a > irg;
                    Y[0] = acc[0]; Y[1] = acc[1]; Y[2] = acc[2];
                }
            }
#pragma unroll 1
            for (int st = 0; st < n_k4; ++st) {
                v4f Yn[3];
                bool near, cut;
                const bool trip = lin_k4<3, false>(G, Ak, Y, S, L.qref, L.vdcref, kpo, kio, irg, 1.f, Yn, near, mine, cut, &Pl);
                if (mine && !leave) {
                    leave |= trip || near;
                    Y[0] = Yn[0]; Y[1] = Yn[1]; Y[2] = Yn[2];
                }
            }
        }
        if (L.run && !leave) {
            lin_to_rows<3>(Y, ys);
            L.ksub += A.C.n_sub;
            L.steps += 1;
        }
    }
    if (leave) { L.run = false; L.live = false; }
    epilogue<float, 3>(Pl, A.C, A.R, L, ys, A.fstate, A.istate, A.ev_vg, A.ev_si, A.out, A.stats, A.n_env, A.stats_fresh, true, vg_stop,
                       si_stop, lane_par);
    if (__syncthreads_or((int)leave)) {
        typedef __attribute__((address_space(4))) const char *KernargBytes;
        KernargPtr kb = (KernargPtr)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr());
        asm volatile("" : "+s"(kb));
        etdrk4_full_tail<true>(*(const EtdArgs *)kb, leave, nullptr, 0, lane_par, 0);
    }
}

constexpr int kOneWavePerSimdEnvs = 65536;       // 256 CUs x 4 SIMDs x 64 lanes (MI355X)

// =============================================================================================
// Open-loop rollouts (pvder_plan_rollout_fused; etdlin, three-phase): n_steps env steps in ONE launch, the
// 11-state reduction in registers from the first step to the last.  A per-step launch spends 2 us waiting
// for its state rows, 1.7 us on outputs and the write-back of 41 rows and 1.2 us between launches around
// 3.3 us of integration; the fused loop reads the state once, writes it once, and per step only computes and
// stores the 13 output rows -- with the SAME code the per-step launch runs (lin_integrate; rotation into the
// 23-row form; outputs / reward_of), so the results are bit-identical to n_steps single launches.
// What the loop does NOT carry is everything rare: a workgroup leaves the loop -- its state as of the
// start of that step restored -- at the first step in which one of its envs is off balance or near a
// limit (23-state path), would finish its episode (done / auto-reset logic, known BEFORE the step: time
// and step counters), is not running, or comes out non-finite.  step_etdlin_finish_kernel then takes
// the workgroup through that step (per-step body, through memory) and the fused kernel resumes.
// =============================================================================================
struct RollArgs {
    LinArgs G;                       // at kernarg offset 0 (the 23-state tail of the finish kernel re-reads G.A from there)
    RollK Rk;
};
static_assert(offsetof(RollArgs, G) == 0, "the finish kernel's tail reads LinArgs / EtdArgs at kernarg offset 0");
typedef __attribute__((address_space(4))) const RollArgs *RollKernargPtr;
// the kernel arguments through a laundered copy of the kernarg pointer: hipcc must re-read what it needs
// (scalar loads, cache hits) instead of keeping hundreds of loop-invariant values -- parameters copied
// into VGPRs, packed constants -- live across the step loop, where every register is taken
__device__ __forceinline__ const RollArgs &roll_args() {
    RollKernargPtr ka = (RollKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    PVDER_LAUNDER_ROLL(ka);
    return *(const RollArgs *)ka;
}

// AG = true: CLOSED LOOP -- the tabular Q-learning step of the reference's training loop (agents/qagent.py:56-105:
// discretise the observation, update the table with the transition that just ended, choose the next action
// epsilon-greedily; pvder_agent_dev.hpp) runs between env step k and k + 1 with the env state still in registers;
// the action row of a step is what the agent chose behind the previous one.
template <int PH, bool AG>
__device__ __forceinline__ void rollout_body(const RollArgs &Q) {
    using namespace pvder;
    constexpr int NS = 6 * PH + 5;
    if (Q.Rk.resume && Q.Rk.progress[blockIdx.x] >= Q.Rk.n_steps) return;      // (workgroup-uniform) nothing left
    const int lane = threadIdx.x & 63;
    LinRegs R;                        // tick matrix, first ETD3RK set and Krogstad block resident for the whole rollout
    lin_load_mat(Q.G.Lk.lcoef, lane, R.At);
    lin_load_set(Q.G.Lk.lcoef + kLinMatFloats, lane, R.Ac);
    R.at_ok = true; R.cur = 1; R.k4_ok = false;
    if (Q.G.Lk.k4coef != nullptr) {
        lin_load_k4(Q.G.Lk.k4coef, lane, R.Ak);
        R.k4_ok = true;
    }
    const int e = blockIdx.x * kBlock + threadIdx.x;
    v4f Y[3];
    float qref, vdcref;
    int ksub, steps;
    bool bail;
    {
        const EtdArgs &A = Q.G.A;
        const size_t N = (size_t)A.n_env;
        const int ec = e < A.n_env ? e : A.n_env - 1;
        float ys[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) ys[q] = ld(&A.fstate[(size_t)q * N + ec]);
        qref = ld(&A.fstate[(size_t)NS * N + ec]);
        vdcref = ld(&A.fstate[(size_t)(NS + 1) * N + ec]);
        ksub = ld(&A.istate[PVDER_I_KSUB * N + ec]);
        steps = ld(&A.istate[PVDER_I_STEPS * N + ec]);
        const int status = ld(&A.istate[PVDER_I_STATUS * N + ec]);
        bail = (PH == 3 && A.force_full != 0) || status != 0;
        if constexpr (PH == 3) bail |= lin_unbalanced(ys);
        lin_from_rows<PH>(ys, Y);
    }
    float last_reward = 0.f, bad = 0.f;
    const int k_first = Q.Rk.resume ? Q.Rk.progress[blockIdx.x] : 0;
    int k = k_first;
    // the agent's registers of this env (closed loop)
    long long ag_s = 0;
    int ag_a = 0;
    bool ag_alive = false;
    double ag_ret = 0.0;
    __shared__ __attribute__((aligned(16))) float qnet[AG ? pvder_agent::kQnetLdsFloats : 4];       // (PVDER_POLICY_QNET: the network's weights)
    if constexpr (AG) {
        const int ec = e < Q.G.A.n_env ? e : Q.G.A.n_env - 1;
        ag_s = Q.Rk.ag_state[ec]; ag_a = Q.Rk.ag_action[ec]; ag_alive = Q.Rk.ag_alive[ec] != 0; ag_ret = Q.Rk.ag_ret[ec];
        if (Q.Rk.ag.policy == 1) {
            pvder_agent::qnet_stage(Q.Rk.q, Q.Rk.ag.hidden, Q.Rk.ag.n_act, qnet, (int)threadIdx.x, kBlock);
            __syncthreads();
        }
    }
    if (!__syncthreads_or((int)bail) && k < Q.Rk.n_steps) {
        // what a step reads from memory -- its action, the event inputs and the schedule row of its first
        // segment -- is requested ONE STEP AHEAD (the addresses only depend on the step counter): a lone wave
        // per SIMD would otherwise sit out two memory round trips per env step
        int a_now;
        LinSeg sg;
        {
            const EtdArgs &A = Q.G.A;
            const size_t N = (size_t)A.n_env;
            const int ec = e < A.n_env ? e : A.n_env - 1;
            if constexpr (AG) a_now = ag_a;
            else a_now = ld(&Q.Rk.actions[(size_t)k * N + ec]);
            sg = lin_first_segment(A.C, Q.G.Lk, ksub, A.ev_vg, A.ev_si, N, ec);
        }
        for (;; ++k) {
            const RollArgs &Q1 = roll_args();
            if (k >= Q1.Rk.n_steps) break;
            const EtdArgs &A = Q1.G.A;
            const size_t N = (size_t)A.n_env;
            const int ec = e < A.n_env ? e : A.n_env - 1;
            const bool traced = k == k_first + 3;       // (trace builds: timeline of the fourth step)
            if (traced) PVDER_TRACE_AT(0);
            int a_next = 0;
            if constexpr (!AG) {
                const int k_next = k + 1 < Q1.Rk.n_steps ? k + 1 : k;
                a_next = ld(&Q1.Rk.actions[(size_t)k_next * N + ec]);
            }
            // (the row in force at the end of this step is the one the next step starts with: its inputs
            // also serve the observation at tStop)
            const LinSeg sg_next = lin_first_segment(A.C, Q1.G.Lk, ksub + A.C.n_sub, A.ev_vg, A.ev_si, N, ec);
            // the state at the start of the step, should the workgroup leave (13 registers: there are enough)
            const v4f Y0 = Y[0], Y1 = Y[1], Y2 = Y[2];
            const float qref0 = qref, vdcref0 = vdcref;
            // the step that would END the episode is not taken here (done / auto-reset live in the per-step body)
            bool leave = ksub + A.C.n_sub >= A.C.max_ksub || (A.C.max_steps > 0 && steps + 1 >= A.C.max_steps);
            // action_calc (PVDER_env.py:198-229)
            if (traced) PVDER_TRACE_AT(1);
            const int a = a_now;
            bad = 0.f;
            if (a == 1) qref += A.C.dq;
            else if (a == 2) qref -= A.C.dq;
            else if (a == 3) vdcref += A.C.dv;
            else if (a == 4) vdcref -= A.C.dv;
            else if (a != 0) bad = 1.f;
            if (!__builtin_amdgcn_readfirstlane(__any((int)leave))) {
                bool careful;          // (an env at the current limit leaves too: its step belongs to the per-step body's careful pass)
                leave = lin_integrate<PH, false>(Q1.G, lane, R, Y, qref, vdcref, ec, sg, true, false, careful, traced ? 0 : 1);
            }
            if (traced) PVDER_TRACE_DEP(20, Y[0].x + Y[2].z);
            float chk = fabsf(Y[0].x) + fabsf(Y[0].y) + fabsf(Y[0].z) + fabsf(Y[0].w) + fabsf(Y[1].x) + fabsf(Y[1].y) +
                        fabsf(Y[1].z) + fabsf(Y[1].w) + fabsf(Y[2].x) + fabsf(Y[2].y) + fabsf(Y[2].z);
            leave |= !(chk < 1e30f);
            if (__syncthreads_or((int)leave)) {
                Y[0] = Y0; Y[1] = Y1; Y[2] = Y2;
                qref = qref0; vdcref = vdcref0;
                break;
            }
            // observation and reward at (y(tStop), tStop), exactly as the per-step epilogue computes them
            if (traced) PVDER_TRACE_AT(2);
            const RollArgs &Q2 = roll_args();
            const EtdArgs &B = Q2.G.A;
            ksub += B.C.n_sub;
            steps += 1;
            float ys[NS];
            lin_to_rows<PH>(Y, ys);
            SubK<float, PH> S;
            make_sub<float, PH>(B.P, sg_next.in_vg, sg_next.in_si, S);
            float iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
            outputs<float, PH>(B.P, S, ys, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
            last_reward = reward_of<float>(B.C, B.P, vrms, Ppcc, Qpcc, ys[6 * PH], vdcref, qref);
            if (traced) PVDER_TRACE_DEP(22, last_reward + ppv);
            if (e < B.n_env) {
                float *o = B.out + k * Q2.Rk.out_stride;
                write_obs<float>(o, N, e, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ys[6 * PH], ppv, vdcref, qref,
                                 (float)ksub * (float)(1.0 / 60.0) * B.C.inv_tmax);
                st(&o[PVDER_O_REWARD * N + e], last_reward);
                st(&o[PVDER_O_DONE * N + e], 0.f);
            }
            if (traced) PVDER_TRACE_AT(21);
            if constexpr (AG) {
                // the agent step behind env step k (never an episode-ending one here: done = false)
                const float obs[pvder_agent::kObs] = {iaR, iaI, vaR, vaI, Ppcc, Qpcc, ys[6 * PH], ppv, vdcref, qref,
                                                      (float)ksub * (float)(1.0 / 60.0) * B.C.inv_tmax};
                if (e < B.n_env)
                    pvder_agent::agent_step(Q2.Rk.ag, Q2.Rk.q, obs, last_reward, false, Q2.Rk.ag.env_id_base + e,
                                            Q2.Rk.ag.counter + (uint64_t)k, ag_s, ag_a, ag_alive, ag_ret, qnet);
                a_next = ag_a;
            }
            a_now = a_next;
            sg = sg_next;
        }
    }
    // state -> memory (as of the last completed step), statistics of the last step if it was completed here
    const RollArgs &Q3 = roll_args();
    const EtdArgs &A = Q3.G.A;
    const size_t N = (size_t)A.n_env;
    const bool live = e < A.n_env;
    if (live && k > k_first) {
        float ys[NS];
        lin_to_rows<PH>(Y, ys);
#pragma unroll
        for (int q = 0; q < NS; ++q) st(&A.fstate[(size_t)q * N + e], ys[q]);
        st(&A.fstate[(size_t)NS * N + e], qref);
        st(&A.fstate[(size_t)(NS + 1) * N + e], vdcref);
        st(&A.istate[PVDER_I_KSUB * N + e], (int32_t)ksub);
        st(&A.istate[PVDER_I_STEPS * N + e], (int32_t)steps);
    }
    if (k == Q3.Rk.n_steps && k > k_first && A.stats) {
        const float r = wave_sum_to_lane63(live ? last_reward : 0.f);
        const float ba = wave_sum_to_lane63(live ? bad : 0.f);
        if ((threadIdx.x & 63) == 63) {              // (#done = #non-finite = 0 in a step taken here)
            v4 *slot = (v4 *)A.stats + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
            *slot = v4{r, 0.f, 0.f, ba};
        }
    }
    if constexpr (AG) {
        if (live) {
            Q3.Rk.ag_state[e] = ag_s; Q3.Rk.ag_action[e] = (uint8_t)ag_a; Q3.Rk.ag_alive[e] = ag_alive ? 1 : 0; Q3.Rk.ag_ret[e] = ag_ret;
        }
    }
    if (threadIdx.x == 0) Q3.Rk.progress[blockIdx.x] = k;
}

template <int PH>
__global__ __launch_bounds__(kBlock)
#ifdef PVDER_ROLLOUT_OCC3
__attribute__((amdgpu_waves_per_eu(3)))      // 168 VGPRs: three waves per SIMD for batches beyond 131 072 envs (rounds 2-3; with the
#endif                                       // Krogstad block resident next to the ETD3RK set the loop needs ~200: two waves per SIMD)
void step_etdlin_rollout_kernel(const RollArgs Q) { rollout_body<PH, false>(Q); }

template <int PH>
__global__ __launch_bounds__(kBlock) void step_etdlin_rollout_agent_kernel(const RollArgs Q) { rollout_body<PH, true>(Q); }

// the steps the fused kernel left to a workgroup, one at a time through memory (usually: the step that
// ends an episode, once per episode)
template <int PH, bool AG = false>
__global__ __launch_bounds__(kBlock) void step_etdlin_finish_kernel(const RollArgs Q) {
    using namespace pvder;
    const LinArgs &G = Q.G;
    const RollK &Rk = Q.Rk;
    const int k0 = Rk.progress[blockIdx.x];
    if (k0 >= Rk.n_steps) return;
    const int k1 = Rk.n_steps - k0 > Rk.max_finish ? k0 + Rk.max_finish : Rk.n_steps;
    const int lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) float qnet[AG ? pvder_agent::kQnetLdsFloats : 4];       // (PVDER_POLICY_QNET: the network's weights)
    if constexpr (AG) {
        if (Rk.ag.policy == 1) {
            pvder_agent::qnet_stage(Rk.q, Rk.ag.hidden, Rk.ag.n_act, qnet, (int)threadIdx.x, kBlock);
            __syncthreads();
        }
    }
    for (int k = k0; k < k1; ++k) {
        if (k > 0 && Rk.out_stride != 0 && !G.A.R.enabled) {      // stale tuples follow the output block (carry_stale_kernel)
            const int e = blockIdx.x * kBlock + threadIdx.x;
            const size_t N = (size_t)G.A.n_env;
            if (e < G.A.n_env && G.A.istate[PVDER_I_STATUS * N + e] != 0) {
                float *o = G.A.out + k * Rk.out_stride;
#pragma unroll
                for (int q = 0; q < PVDER_OROWS; ++q) o[(size_t)q * N + e] = (o - Rk.out_stride)[(size_t)q * N + e];
            }
        }
        LinRegs R;
        R.at_ok = false; R.cur = 0; R.k4_ok = false;       // (loaded where the step needs them)
        lin_step_body<PH, 0, PH == 3>(G, lane, R, AG ? Rk.ag_action : Rk.actions + (size_t)k * G.A.n_env, k * Rk.out_stride, 1,
                             G.A.fstate, G.A.istate, G.A.n_env);
        if constexpr (AG) {
            // the agent step behind this env step, from the output block the step has just written (this lane's own
            // rows); the transition that ENDS an episode is among the steps taken here
            const RollArgs &Qa = roll_args();
            const int e = blockIdx.x * kBlock + threadIdx.x;
            if (e < Qa.G.A.n_env)
                pvder_agent::agent_step_block(Qa.Rk.ag, Qa.Rk.q, Qa.G.A.out + k * Qa.Rk.out_stride, (size_t)Qa.G.A.n_env, e,
                                              Qa.Rk.ag.counter + (uint64_t)k, Qa.Rk.ag_state, Qa.Rk.ag_action, Qa.Rk.ag_alive,
                                              Qa.Rk.ag_ret, qnet);
        }
        __threadfence();                 // the next step of this workgroup reads what this one wrote
        __syncthreads();
    }
    if (threadIdx.x == 0) Rk.progress[blockIdx.x] = k1;
}

#ifdef PVDER_RK45_LAMBDA
// REPRODUCER BUILD ONLY (tools/hipcc_rk45_lambda_repro.sh): one sim time step of the adaptive pair factored into a helper that
// takes the right-hand side as a lambda -- the arithmetic of the written-out loop in step_rk45_kernel, line for line.
// hipcc (ROCm 7.2) compiled the 23-state instance of this to code that returns NaN states in round 2.
template <typename T, int NS, typename F>
__device__ __forceinline__ void rk45_sim_step(F rhsf, T *y, T &hn, int max_it, T rtol, T atol) {
    constexpr T c2 = 1. / 5, c3 = 3. / 10, c4 = 4. / 5, c5 = 8. / 9;
    constexpr T a21 = 1. / 5, a31 = 3. / 40, a32 = 9. / 40, a41 = 44. / 45, a42 = -56. / 15, a43 = 32. / 9,
                a51 = 19372. / 6561, a52 = -25360. / 2187, a53 = 64448. / 6561, a54 = -212. / 729,
                a61 = 9017. / 3168, a62 = -355. / 33, a63 = 46732. / 5247, a64 = 49. / 176,
                a65 = -5103. / 18656, b1 = 35. / 384, b3 = 500. / 1113, b4 = 125. / 192,
                b5 = -2187. / 6784, b6 = 11. / 84;
    constexpr T e1 = 71. / 57600, e3 = -71. / 16695, e4 = 71. / 1920, e5 = -17253. / 339200,
                e6 = 22. / 525, e7 = -1. / 40;
    const T Tend = (T)(1.0 / 60.0);
    const T tq = (T)120;
    T t = (T)0;
    T k1[NS];
    rhsf(y, (T)0, k1);
    // bounded: every lane leaves after at most 2e7 attempts per 1/60 s (h >= ~1e-9 s)
    for (int it = 0; it < max_it && t < Tend * (T)(1.0 - 1e-14); ++it) {
        T h = hn;
        bool last = false;
        if (t + h >= Tend) { h = Tend - t; last = true; }
        T k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], yt[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a21 * k1[q]);
        rhsf(yt, (t + c2 * h) * tq, k2);
#pragma unroll
        for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a31 * k1[q] + a32 * k2[q]);
        rhsf(yt, (t + c3 * h) * tq, k3);
#pragma unroll
        for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a41 * k1[q] + a42 * k2[q] + a43 * k3[q]);
        rhsf(yt, (t + c4 * h) * tq, k4);
#pragma unroll
        for (int q = 0; q < NS; ++q)
            yt[q] = y[q] + h * (a51 * k1[q] + a52 * k2[q] + a53 * k3[q] + a54 * k4[q]);
        rhsf(yt, (t + c5 * h) * tq, k5);
#pragma unroll
        for (int q = 0; q < NS; ++q)
            yt[q] = y[q] + h * (a61 * k1[q] + a62 * k2[q] + a63 * k3[q] + a64 * k4[q] + a65 * k5[q]);
        rhsf(yt, (t + h) * tq, k6);
        T errv[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            yt[q] = y[q] + h * (b1 * k1[q] + b3 * k3[q] + b4 * k4[q] + b5 * k5[q] + b6 * k6[q]);
            errv[q] = e1 * k1[q] + e3 * k3[q] + e4 * k4[q] + e5 * k5[q] + e6 * k6[q];
        }
        rhsf(yt, (t + h) * tq, k2);   // k7 (reuses k2)
        T err = (T)0;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const T ek = h * (errv[q] + e7 * k2[q]);
            const T ay = fmax(fabs(y[q]), fabs(yt[q]));
            const T sc = atol + rtol * ay;
            const T r = ek / sc;
            err += r * r;
        }
        err = sqrt(err * (T)(1.0 / NS));
        const bool ok = !(err > (T)1) || h < (T)1e-12;   // NaN err: accept, caught by the epilogue
        if (ok) {
            t += h;
#pragma unroll
            for (int q = 0; q < NS; ++q) { y[q] = yt[q]; k1[q] = k2[q]; }   // FSAL
        }
        T fac = (T)0.9 * pow(fmax(err, (T)1e-30), (T)-0.2);
        fac = fmin((T)5, fmax((T)0.2, fac));
        if (!ok) hn = h * fac;
        else if (!last || h * fac < hn) hn = h * fac;
        if (ok && last) break;
    }
}
#endif

// =============================================================================================
// adaptive Dormand-Prince RK45 (per-lane step size; fp64 tolerance-match path)
// =============================================================================================
template <typename T, int PH>
__global__ __launch_bounds__(kBlock) void step_rk45_kernel(
    const ModelK<T> P, const StepK<T> C, const ResetK R, T *__restrict__ fstate, int32_t *__restrict__ istate,
    float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ action,
    T *__restrict__ out, T *__restrict__ hstep, float *__restrict__ stats, int n_env, int stats_fresh, int run_status) {
    constexpr int NS = 6 * PH + 5;
    const size_t N = (size_t)n_env;
    Lane<T> L;
    T y[NS];
    prologue<T, NS>(C, fstate, istate, action, n_env, L, y, run_status);     // (kDeferred: the envs etdlin64 handed over)
    T hn = hstep[L.ec];
    if (!(hn > (T)0)) hn = (T)1e-6;

    // Dormand-Prince 5(4) tableau
    constexpr T c2 = 1. / 5, c3 = 3. / 10, c4 = 4. / 5, c5 = 8. / 9;
    constexpr T a21 = 1. / 5, a31 = 3. / 40, a32 = 9. / 40, a41 = 44. / 45, a42 = -56. / 15, a43 = 32. / 9,
                a51 = 19372. / 6561, a52 = -25360. / 2187, a53 = 64448. / 6561, a54 = -212. / 729,
                a61 = 9017. / 3168, a62 = -355. / 33, a63 = 46732. / 5247, a64 = 49. / 176,
                a65 = -5103. / 18656, b1 = 35. / 384, b3 = 500. / 1113, b4 = 125. / 192,
                b5 = -2187. / 6784, b6 = 11. / 84;
    constexpr T e1 = 71. / 57600, e3 = -71. / 16695, e4 = 71. / 1920, e5 = -17253. / 339200,
                e6 = 22. / 525, e7 = -1. / 40;
    const T Tend = (T)(1.0 / 60.0);
    const T tq = (T)120;                            // 2*theta in turns = 120 t

    if (L.run) {
        SubK<T, PH> S;
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(L.ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            pvder::make_sub<T, PH>(P, (T)ev_vg[(size_t)row * N + L.ec], (T)ev_si[(size_t)row * N + L.ec], S);
#ifdef PVDER_RK45_LAMBDA
            rk45_sim_step<T, NS>([&](const T *yy, T tau, T *dy) { pvder::rhs<T, PH>(P, S, yy, tau, L.qref, L.vdcref, dy); },
                                 y, hn, C.max_it, C.rtol, C.atol);
#else
            T t = (T)0;
            T k1[NS];
            pvder::rhs<T, PH>(P, S, y, (T)0, L.qref, L.vdcref, k1);
            // bounded: every lane leaves after at most 2e7 attempts per 1/60 s (h >= ~1e-9 s)
            for (int it = 0; it < C.max_it && t < Tend * (T)(1.0 - 1e-14); ++it) {
                T h = hn;
                bool last = false;
                if (t + h >= Tend) { h = Tend - t; last = true; }
                T k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], yt[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a21 * k1[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c2 * h) * tq, L.qref, L.vdcref, k2);
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a31 * k1[q] + a32 * k2[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c3 * h) * tq, L.qref, L.vdcref, k3);
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a41 * k1[q] + a42 * k2[q] + a43 * k3[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c4 * h) * tq, L.qref, L.vdcref, k4);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    yt[q] = y[q] + h * (a51 * k1[q] + a52 * k2[q] + a53 * k3[q] + a54 * k4[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c5 * h) * tq, L.qref, L.vdcref, k5);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    yt[q] = y[q] + h * (a61 * k1[q] + a62 * k2[q] + a63 * k3[q] + a64 * k4[q] + a65 * k5[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + h) * tq, L.qref, L.vdcref, k6);
                T errv[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    yt[q] = y[q] + h * (b1 * k1[q] + b3 * k3[q] + b4 * k4[q] + b5 * k5[q] + b6 * k6[q]);
                    errv[q] = e1 * k1[q] + e3 * k3[q] + e4 * k4[q] + e5 * k5[q] + e6 * k6[q];
                }
                pvder::rhs<T, PH>(P, S, yt, (t + h) * tq, L.qref, L.vdcref, k2);   // k7 (reuses k2)
                T err = (T)0;
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const T ek = h * (errv[q] + e7 * k2[q]);
                    const T ay = fmax(fabs(y[q]), fabs(yt[q]));
                    const T sc = C.atol + C.rtol * ay;
                    const T r = ek / sc;
                    err += r * r;
                }
                err = sqrt(err * (T)(1.0 / NS));
                const bool ok = !(err > (T)1) || h < (T)1e-12;   // NaN err: accept, caught by the epilogue
                if (ok) {
                    t += h;
#pragma unroll
                    for (int q = 0; q < NS; ++q) { y[q] = yt[q]; k1[q] = k2[q]; }   // FSAL
                }
                T fac = (T)0.9 * pow(fmax(err, (T)1e-30), (T)-0.2);
                fac = fmin((T)5, fmax((T)0.2, fac));
                if (!ok) hn = h * fac;
                else if (!last || h * fac < hn) hn = h * fac;
                if (ok && last) break;
            }
#endif
        }
        L.ksub += C.n_sub;
        L.steps += 1;
    }
    const bool ran = L.run;
    // no in-kernel auto-reset here: this kernel sits at the 256-VGPR + AGPR limit -- hipcc has miscompiled it
    // twice when its code was rearranged (auto-reset inlined into the epilogue, round 1; the sim time step
    // factored into a helper taking the right-hand side as a lambda, round 2: NaN states; both caught by the
    // parity tests) -- so its loop stays written out and the launcher enqueues
    // reset_kernel<..., true> (reset where status != 0) right behind it instead
    epilogue<T, PH, false>(P, C, R, L, y, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh);
    if (ran) hstep[L.e] = (L.status != 0) ? (T)0 : hn;       // a finished env restarts the controller
}

// =============================================================================================
// etdlin64 -- the tolerance-match path that is not stability-bound (BASELINE configs[4]).
// The explicit RK45 pair above resolves the stiff current loop: 62 000 accepted steps per env step, 0.55 s per batched
// step.  This kernel runs the etdlin scheme in DOUBLE on the balanced 11-state reduction -- the exponential carries the
// stiff loop exactly -- with STEP DOUBLING on a dyadic menu as its error control: level j takes a Krogstad ETDRK4 tick behind
// the input jump and 2^(j+1) equal Krogstad steps over the rest of the env step; the result of level j is accepted
// when it differs from that of level j-1 by less than the tolerance in the RK45 kernel's norm.  PER LANE: a wave climbs the menu until
// each of its lanes has accepted; a lane's result depends on its own values only.  PVDER-v0 steps resolve at level 1
// (tick + 2, tick + 4 steps), a few per cent at level 2, the step behind a reset -- the PLL swings a quarter turn into lock -- at 128.
// Envs the reduction does not cover -- off balance, near the duty-cycle or current limit, an event inside the env
// step, nothing accepted at the finest pair -- are marked kDeferred and taken by step_rk45_kernel in the follow-up
// launch.  Dense 11x11 products on the fp64 VALU in ROLLED column loops, operands broadcast from LDS (8 sets x 10 matrices:
// 84 KB, dynamic), the lane's right-hand side staged in LDS beside them: 256 VGPRs + 66 AGPRs, NO scratch.
// fp64 prototype on the goldens (etd_linear.lin64_reference_step): <= 1.5e-6 of the converged solution.
// Round 5: 2.83 -> 0.20 ms per batched step at 65 536 envs (worst step 6.1 -> 0.33 ms), 1.03e-6 of the bridge
// (profiles/r05_l64_accept_sweep.txt): Krogstad steps (the tail of the level distribution is what a launch waits for) and the
// rolled products (14 us per step of four right-hand sides and ten products; round 4: 30 us per ETD3RK step, 99 spilled VGPRs).
// =============================================================================================
constexpr int kL64Cols = 12, kL64Mat = 11 * kL64Cols, kL64Set = 10 * kL64Mat, kL64MaxLevels = 9;
// acceptance: two consecutive levels agree to kL64Accept x the tolerance -- no Richardson credit: behind an input jump the
// schemes lose order on the slaved current loop (ETD3RK to ONE: true error, in units of the tolerance, 3.9 2.5 1.5 0.73 0.35
// for 2 .. 32 steps; Krogstad to ~3), where the error is close to the difference: factor 1
constexpr double kL64Accept = 1.0;
constexpr int kL64MinLevel = 1;
struct Lin64K {
    const double *coef;              // device: (1 + levels) sets of kL64Set doubles (set 0: the tick)
    int levels;
    int min_level;                   // the first level whose result may be accepted (against the level below it)
    double accept;                   // two consecutive levels must agree to `accept` x the tolerance
    double sat2, ir2_guard;          // hand-over guards: |m|^2, |iref|^2
};

// acc[j] += M_mat[j] f  for NM matrices of one set at once.  A matrix sits in LDS COLUMN by column ([11 columns][12 doubles]:
// column k = the 11 entries that multiply f[k], read as six broadcast 16-byte words); the lane's f sits in LDS too
// (Fl[k * kBlock]: written by the caller behind each right-hand side).  Round 5: a ROLLED loop over the columns -- the body is
// 6 NM operand reads, one read of f[k] and 11 NM fused multiply-adds with static register indices; nothing for hipcc to hoist
// (round 4 unrolled the 121 x NM products and pinned a depth-one pipeline with scheduling barriers: 24 operand registers in
// flight, and still 99 spilled registers around it).
template <int NM>
__device__ __forceinline__ void mv64(const double *__restrict__ Mset, const int (&mat)[NM], const double *__restrict__ Fl,
                                     double (&acc)[NM][11]) {
#pragma unroll 1
    for (int k = 0; k < 11; ++k) {
        double2 col[NM][6];
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const double2 *c = (const double2 *)(Mset + mat[j] * kL64Mat + k * kL64Cols);
#pragma unroll
            for (int q = 0; q < 6; ++q) col[j][q] = c[q];
        }
        const double fk = Fl[k * kBlock];
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[j][2 * q] = fma(col[j][q].x, fk, acc[j][2 * q]);
                if (2 * q + 1 < 11) acc[j][2 * q + 1] = fma(col[j][q].y, fk, acc[j][2 * q + 1]);
            }
    }
}
// |m_a|^2 and |iref_a|^2 at a balanced state
__device__ __forceinline__ void guards64(const ModelK<double> &P, const SubK<double, 1> &S, const double (&y)[11], double qref,
                                         double vdcref, double &m2, double &ir2) {
    const double mR = P.KpG * y[4] + y[2], mI = P.KpG * y[5] + y[3];
    m2 = mR * mR + mI * mI;
    const double vR = S.AR[0] + P.kviR * y[0] - P.kviI * y[1], vI = S.AI[0] + P.kviR * y[1] + P.kviI * y[0];
    const double Q = 1.5 * (vI * y[0] - vR * y[1]);
    const double rR = y[7] + P.KpDC * (vdcref - y[6]), rI = y[8] - P.KpQ * (qref - Q);
    ir2 = rR * rR + rI * rI;
}
// `nsteps` Krogstad ETDRK4 steps in f-form (etd_linear.k4_step; matrices KM21 ... KB4 of pvder_etdlin.hpp: four right-hand sides,
// ten products, one matrix at a time) from u, in place; trip: a guard (|m|^2, |iref|^2) was met on the way.
// Inlined at its ONE call site: the kernel runs the tick and every level through the same loop nest (round 4 had the step as a
// real function, called per step, its array arguments in scratch memory: 784 B per lane; inlined at TWO call sites -- tick and
// level loop -- the body kept 1100 values alive and spilled 600 of them).
__device__ __forceinline__ void rhs_to_lds(const ModelK<double> &P, const SubK<double, 1> &S, const double (&y)[11], double qref,
                                           double vdcref, double *__restrict__ Fl) {
    double F[11];
    pvder::rhs<double, 1, 3>(P, S, y, 0.0, qref, vdcref, F);
#pragma unroll
    for (int i = 0; i < 11; ++i) Fl[i * kBlock] = F[i];
}
__device__ __forceinline__ bool run64(const ModelK<double> &P, const SubK<double, 1> &S, const double *__restrict__ Mset, int nsteps,
                                      double (&u)[11], double qref, double vdcref, double sat2, double ir2_guard,
                                      double *__restrict__ Fl) {
    bool trip = false;
#pragma unroll 1
    for (int st = 0; st < nsteps; ++st) {
        double m2a, m2b, ir2, irb;
        guards64(P, S, u, qref, vdcref, m2a, ir2);
        // stage values as running sums  w[0] = U2, w[1] = U3, w[2] = U4, w[3] = y+  (all start from y)
        double w[4][11];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 11; ++i) w[j][i] = u[i];
        rhs_to_lds(P, S, u, qref, vdcref, Fl);                                    // F1
        { const int m[4] = {pvder::KM21, pvder::KM31, pvder::KM41, pvder::KB1}; mv64<4>(Mset, m, Fl, w); }
        rhs_to_lds(P, S, w[0], qref, vdcref, Fl);                                 // F2 = f(U2)
        {
            const int m[3] = {pvder::KM32, pvder::KM42, pvder::KB2};
            double v[3][11];
#pragma unroll
            for (int i = 0; i < 11; ++i) { v[0][i] = w[1][i]; v[1][i] = w[2][i]; v[2][i] = w[3][i]; }
            mv64<3>(Mset, m, Fl, v);
#pragma unroll
            for (int i = 0; i < 11; ++i) { w[1][i] = v[0][i]; w[2][i] = v[1][i]; w[3][i] = v[2][i]; }
        }
        rhs_to_lds(P, S, w[1], qref, vdcref, Fl);                                 // F3 = f(U3)
        {
            const int m[2] = {pvder::KM43, pvder::KB3};
            double v[2][11];
#pragma unroll
            for (int i = 0; i < 11; ++i) { v[0][i] = w[2][i]; v[1][i] = w[3][i]; }
            mv64<2>(Mset, m, Fl, v);
#pragma unroll
            for (int i = 0; i < 11; ++i) { w[2][i] = v[0][i]; w[3][i] = v[1][i]; }
        }
        rhs_to_lds(P, S, w[2], qref, vdcref, Fl);                                 // F4 = f(U4)
        {
            const int m[1] = {pvder::KB4};
            double v[1][11];
#pragma unroll
            for (int i = 0; i < 11; ++i) v[0][i] = w[3][i];
            mv64<1>(Mset, m, Fl, v);
#pragma unroll
            for (int i = 0; i < 11; ++i) u[i] = v[0][i];
        }
        guards64(P, S, u, qref, vdcref, m2b, irb);
        trip |= fmax(m2a, m2b) > sat2 || ir2 > ir2_guard;
    }
    return trip;
}

__global__ __launch_bounds__(kBlock) void step_etdlin64_kernel(
    const ModelK<double> P, const StepK<double> C, const ResetK R, const Lin64K K, double *__restrict__ fstate,
    int32_t *__restrict__ istate, float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ action,
    double *__restrict__ out, float *__restrict__ stats, int n_env, int stats_fresh) {
    constexpr int NS = 23;
    const size_t N = (size_t)n_env;
    extern __shared__ __attribute__((aligned(16))) double Ms[];       // (1 + levels) x kL64Set doubles + 11 x kBlock (launch_plan sizes it)
    for (int k = threadIdx.x; k < (1 + K.levels) * kL64Set; k += kBlock) Ms[k] = K.coef[k];
    double *Fl = Ms + (size_t)(1 + K.levels) * kL64Set + threadIdx.x;     // this lane's right-hand side: Fl[k * kBlock]
    __syncthreads();
    Lane<double> L;
    double ys[NS];
    prologue<double, NS>(C, fstate, istate, action, n_env, L, ys);
    bool defer = false;
    if (L.run) {
        // balance (phases b, c = phase a rotated by -/+ 120 degrees) to 1e-9 relative: what a balanced env carries in
        // double is rounding; anything more is a real imbalance (23-state model: the RK45 kernel)
        double ub = 0.0, nrm = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double aR = ys[2 * k], aI = ys[2 * k + 1];
            const double bR = -0.5 * aR + 0.86602540378443864676 * aI, bI = -0.5 * aI - 0.86602540378443864676 * aR;
            const double cR = -0.5 * aR - 0.86602540378443864676 * aI, cI = -0.5 * aI + 0.86602540378443864676 * aR;
            const double d = fabs(ys[6 + 2 * k] - bR) + fabs(ys[7 + 2 * k] - bI) + fabs(ys[12 + 2 * k] - cR) + fabs(ys[13 + 2 * k] - cI);
            const double sc = (k == 2) ? 1e-6 : fabs(aR) + fabs(aI);       // (u ~ 1e-6: absolute)
            ub = fmax(ub, d / (sc + 1e-300));
            nrm = fmax(nrm, sc);
        }
        defer = !(ub <= 1e-9) || !(nrm > 0.0);
    }
    // the env step must be ONE segment: the event row in force does not change inside it
    int row = 0;
    {
        int until = 0x3fffffff;
        if (C.ev_count > 0) {
            const int d = L.ksub - C.ev_s0;
            if (d < 0) {
                until = -d;
            } else {
                const int qd = d / C.ev_ds, rd = d - qd * C.ev_ds;
                row = qd + 1 < C.ev_count ? qd + 1 : C.ev_count;
                until = qd + 1 < C.ev_count ? C.ev_ds - rd : 0x3fffffff;
            }
        }
        defer |= until < C.n_sub;
    }
    SubK<double, 1> S;
    pvder::make_sub<double, 1>(P, (double)ev_vg[(size_t)row * N + L.ec], (double)ev_si[(size_t)row * N + L.ec], S);
    double y0[11] = {ys[0], ys[1], ys[2], ys[3], ys[4], ys[5], ys[18], ys[19], ys[20], ys[21], ys[22]};
    bool active = L.run && !defer;
    if (__syncthreads_or((int)active)) {           // (workgroup-uniform; every lane of an entering wave computes)
        double yt[11], prev[11], u[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) { yt[i] = y0[i]; prev[i] = y0[i]; }
        bool accepted = false;
        // lvl = -1: the tick (set 0, one step, from y0); lvl >= 0: 2^(lvl+1) steps from the tick's result -- ONE loop nest, so
        // that the step body exists once
        for (int lvl = -1; lvl < K.levels; ++lvl) {
            if (!__builtin_amdgcn_readfirstlane(__any((int)(active && !defer && !accepted)))) break;
            const double *Mset = Ms + (size_t)(1 + lvl) * kL64Set;
#pragma unroll
            for (int i = 0; i < 11; ++i) u[i] = yt[i];
            const bool trip = run64(P, S, Mset, lvl < 0 ? 1 : (2 << lvl), u, L.qref, L.vdcref, K.sat2, K.ir2_guard, Fl);
            if (lvl < 0) {
#pragma unroll
                for (int i = 0; i < 11; ++i) yt[i] = u[i];
                defer |= active && trip;
                continue;
            }
            bool ok = false;
            if (lvl >= K.min_level) {
                double e2 = 0.0;
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    const double sc = K.accept * (C.atol + C.rtol * fmax(fabs(u[i]), fabs(prev[i])));
                    const double r = (u[i] - prev[i]) / sc;
                    e2 += r * r;
                }
                ok = e2 * (1.0 / 11.0) <= 1.0;
            }
            if (active && !defer && !accepted) {
                defer |= trip;
                accepted = ok;
#pragma unroll
                for (int i = 0; i < 11; ++i) prev[i] = u[i];          // the level's result: the candidate / the accepted one
            }
        }
        defer |= active && !accepted;              // nothing accepted at the finest pair: the RK45 kernel decides
        if (active && !defer) {
            // phases b, c = phase a rotated by -/+ 120 degrees
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double aR = prev[2 * k], aI = prev[2 * k + 1];
                ys[2 * k] = aR; ys[2 * k + 1] = aI;
                ys[6 + 2 * k] = -0.5 * aR + 0.86602540378443864676 * aI; ys[7 + 2 * k] = -0.5 * aI - 0.86602540378443864676 * aR;
                ys[12 + 2 * k] = -0.5 * aR - 0.86602540378443864676 * aI; ys[13 + 2 * k] = -0.5 * aI + 0.86602540378443864676 * aR;
            }
            ys[18] = prev[6]; ys[19] = prev[7]; ys[20] = prev[8]; ys[21] = prev[9]; ys[22] = prev[10];
            L.ksub += C.n_sub;
            L.steps += 1;
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) ys[q] = 1.0;      // (a lane that stores nothing: keeps ys out of the loop's registers)
        }
    } else {
#pragma unroll
        for (int q = 0; q < NS; ++q) ys[q] = 1.0;
    }
    if (L.run && defer) {
        istate[PVDER_I_STATUS * N + L.e] = kDeferred;      // marker for the follow-up launch (step_rk45_kernel)
        L.run = false;
        L.live = false;
    }
    // (same-step auto-reset: the reset pass behind the RK45 follow-up, as for that kernel)
    epilogue<double, 3, false>(P, C, R, L, ys, fstate, istate, ev_vg, ev_si, out, stats, n_env, stats_fresh);
}

// =============================================================================================
// masked reset: steady state broadcast + event-table refill + reset observation
// =============================================================================================
// FINISHED = true: auto-reset pass behind a step kernel -- resets the envs whose status != 0 and
// leaves the reward/done rows (the transition that ended the episode) untouched
template <typename T, int PH, bool FINISHED = false>
__global__ __launch_bounds__(kBlock) void reset_kernel(
    const ModelK<T> P, const ResetK R, T *__restrict__ fstate, int32_t *__restrict__ istate,
    float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ mask,
    T *__restrict__ out, int n_env, const float *__restrict__ lane_par = nullptr) {
    constexpr int NS = 6 * PH + 5;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_env) return;
    const size_t N = (size_t)n_env;
    if (FINISHED) {
        if (istate[PVDER_I_STATUS * N + e] == 0) return;
    } else if (mask && mask[e] == 0) {
        return;
    }
    T y[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) y[k] = (T)R.y0[k];
    ModelK<T> Pl = P;
    if (lane_par) {
        // per-lane parameters: the env's own steady state, and the PV constants its reset observation (Ppv) is taken with
        lane_steady_state<T, PH>(lane_par, N, e, y);
        Pl.pvA = (T)lane_par[(size_t)PVDER_LPF_PVA * N + e];
        Pl.pvB = (T)lane_par[(size_t)PVDER_LPF_PVB * N + e];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) fstate[(size_t)k * N + e] = y[k];
    const T qref = (T)0, vdcref = P.vdc_ref0;
    fstate[(size_t)NS * N + e] = qref;
    fstate[(size_t)(NS + 1) * N + e] = vdcref;
    const int episode = istate[PVDER_I_EPISODE * N + e] + 1;
    istate[PVDER_I_KSUB * N + e] = 0;
    istate[PVDER_I_STEPS * N + e] = 0;
    istate[PVDER_I_STATUS * N + e] = 0;
    istate[PVDER_I_EPISODE * N + e] = episode;
    if (!R.keep_events)
        fill_events(R.spec, (float)P.sinsol0, R.seed, (uint64_t)(R.env_id_base + (R.env_ids ? (int64_t)R.env_ids[e] : (int64_t)e)), (uint64_t)episode, ev_vg,
                    ev_si, N, e);
    if (out) {
        SubK<T, PH> S;
        pvder::make_sub<T, PH>(Pl, (T)ev_vg[e], (T)ev_si[e], S);
        T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
        pvder::outputs<T, PH>(Pl, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
        write_obs<T>(out, N, e, iaR, iaI, vaR, vaI, Ppcc, Qpcc, y[6 * PH], ppv, vdcref, qref, (T)0);
        if (!FINISHED) {
            out[PVDER_O_REWARD * N + e] = (T)0;
            out[PVDER_O_DONE * N + e] = (T)0;
        }
    }
}

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
    mutable hipEvent_t thr_ev[2] = {nullptr, nullptr};
    mutable bool thr_set[2] = {false, false};
    mutable unsigned thr_n = 0;
    bool graph_hints = false;        // pvder_plan_graph_hints: a CAPTURED pvder_plan_rollout keeps the clock hints it was captured with
    mutable int ksub_hint;           // etdlin: the clock (ksub) most envs of the batch are at when the NEXT step starts, -1 = unknown
                                     // (pvder_plan_hint; advanced by every step this plan enqueues: a prefetch hint, never a result)
    int n_sub;
    // A/B experiment (PVDER_CHAIN=1, pvder_plan_rollout): per-workgroup chained launches on two streams
    mutable int32_t *chain = nullptr;          // device: int32[n_wg + 1] (progress per workgroup, [n_wg] = a wait timed out)
    mutable hipStream_t chain_stream = nullptr;
    mutable hipEvent_t chain_fork = nullptr, chain_join = nullptr;
    ~pvder_plan() {
        if (hyb_flag_host) (void)hipHostFree(hyb_flag_host);
        for (int k = 0; k < 2; ++k) if (thr_ev[k]) (void)hipEventDestroy(thr_ev[k]);
        if (chain) (void)hipFree(chain);
        if (chain_fork) (void)hipEventDestroy(chain_fork);
        if (chain_join) (void)hipEventDestroy(chain_join);
        if (chain_stream) (void)hipStreamDestroy(chain_stream);
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
            *p.hyb_flag_host = 0;
            A.hyb_flag = (int *)dev;
        } else {
            // (no device: plans are still folded and validated -- tests/test_host.py; nothing can be launched there anyway)
            (void)hipGetLastError();
            if (p.hyb_flag_host) (void)hipHostFree(p.hyb_flag_host);
            p.hyb_flag_host = nullptr;
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
        if (p.ph == 3 && p.n_sub <= kHybMaxSub && (p.hyb_mode == 0 || p.hyb_mode == 3)) {
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
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool capturing = st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
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
            hipLaunchKernelGGL(step_sat_hybrid_kernel, dim3((n_env + kHybBlock - 1) / kHybBlock), dim3(kHybBlock), kHybLdsBytes, st, A, p.lane_par);
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
    if (p.hyb_mode == 0 && p.hyb_flag_host) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
        if (!capturing && ((++p.thr_n) & 7u) == 0u) {
            const int slot = (int)((p.thr_n >> 3) & 1u);
            if (!p.thr_ev[slot] && hipEventCreateWithFlags(&p.thr_ev[slot], hipEventDisableTiming) != hipSuccess) p.thr_ev[slot] = nullptr;
            if (p.thr_ev[slot]) {
                if (p.thr_set[slot]) while (hipEventQuery(p.thr_ev[slot]) == hipErrorNotReady) { }
                p.thr_set[slot] = hipEventRecord(p.thr_ev[slot], st) == hipSuccess;
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

// (A/B experiment) 1 if a chained launch gave up waiting for its predecessor since the last chained rollout began (host sync)
extern "C" int pvder_debug_chain_timeout(const pvder_plan *plan) {
    if (!plan || !plan->chain) return 0;
    int32_t v = 0;
    const int n_wg = (plan->n_env + kBlock - 1) / kBlock;
    if (hipMemcpy(&v, plan->chain + n_wg, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
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
#ifdef PVDER_CHAIN_EXPERIMENT
    static const bool chained = []() { const char *v = getenv("PVDER_CHAIN"); return v && v[0] == '1'; }();
    if (chained && plan->integrator == PVDER_INT_ETDLIN && plan->ph == 3 && plan->R.enabled && n_steps > 1) {
        // ---- A/B experiment: consecutive steps on two alternating streams, ordered per workgroup (step_etdlin_entry)
        const pvder_plan &p = *plan;
        const int n_env = p.n_env, n_wg = (n_env + kBlock - 1) / kBlock;
        hipStream_t sa = (hipStream_t)stream;
        hipError_t e = hipSuccess;
        if (!p.chain) {
            e = hipMalloc((void **)&p.chain, sizeof(int32_t) * (size_t)(n_wg + 1));
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.chain_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.chain_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.chain_join, hipEventDisableTiming);
            if (e != hipSuccess) return hip_rc(e);
        }
        e = hipMemsetAsync(p.chain, 0, sizeof(int32_t) * (size_t)(n_wg + 1), sa);
        if (e == hipSuccess) e = hipEventRecord(p.chain_fork, sa);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.chain_stream, p.chain_fork, 0);
        if (e != hipSuccess) return hip_rc(e);
        const dim3 grid(n_wg), block(kBlock);
        for (int k = 0; k < n_steps; ++k) {
            LinArgs G = p.lin;
            EtdArgs &A = G.A;
            A.fstate = (float *)fstate; A.istate = istate; A.ev_vg = ev_vg; A.ev_si = ev_si;
            A.action = actions + (size_t)k * n_env; A.out = (float *)out + (size_t)k * out_step_stride; A.stats = stats; A.stats_fresh = 1;
            static const bool one_stream = []() { const char *v = getenv("PVDER_CHAIN_ONE_STREAM"); return v && v[0] == '1'; }();
            hipLaunchKernelGGL(step_etdlin_kernel_chain, grid, block, 0, ((k & 1) && !one_stream) ? p.chain_stream : sa, A.fstate, A.istate, A.action,
                               G.Lk.lcoef, n_env, hint_word(p), G.Lk.k4coef, G, p.chain, k);
            advance_hint(p, 1);
        }
        e = hipEventRecord(p.chain_join, p.chain_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(sa, p.chain_join, 0);
        return e != hipSuccess ? hip_rc(e) : hip_rc(hipGetLastError());
    }
#endif
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
        hipLaunchKernelGGL(step_etdlin_rollout_kernel<3>, grid, block, 0, st, Q);
        hipLaunchKernelGGL(step_etdlin_finish_kernel<3>, grid, block, 0, st, Q);
    }
    advance_hint(*plan, n_steps);
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
