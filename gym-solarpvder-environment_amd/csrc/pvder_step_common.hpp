// pvder_step_common.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// What every step kernel shares: kernel-argument structs, the counter-based event generator, prologue (action_calc) and epilogue (outputs, reward, done, auto-reset, statistics).
#pragma once

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
constexpr int kBlock = PVDER_BLOCK;   // 4 waves: one per SIMD of a CU
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

// a store that is THROUGH the XCD's L2 when it is acknowledged (agent scope): what a kernel still running publishes to kernels that
// start on other streams meanwhile -- the served rollout's output rows (a launch boundary does it for every other kernel)
template <typename T>
__device__ __forceinline__ void st_through(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T, bool THROUGH = false>
__device__ __forceinline__ void st_out(T *p, T v) {
    if constexpr (THROUGH) st_through(p, v); else st(p, v);
}
template <typename T, bool THROUGH = false>
__device__ __forceinline__ void write_obs(T *__restrict__ out, size_t N, int e, T iaR, T iaI, T vaR, T vaI,
                                          T Ppcc, T Qpcc, T Vdc, T ppv, T vdcref, T qref, T tfrac) {
    st_out<T, THROUGH>(&out[0 * N + e], iaR);   st_out<T, THROUGH>(&out[1 * N + e], iaI);
    st_out<T, THROUGH>(&out[2 * N + e], vaR);   st_out<T, THROUGH>(&out[3 * N + e], vaI);
    st_out<T, THROUGH>(&out[4 * N + e], Ppcc);  st_out<T, THROUGH>(&out[5 * N + e], Qpcc);
    st_out<T, THROUGH>(&out[6 * N + e], Vdc);   st_out<T, THROUGH>(&out[7 * N + e], ppv);
    st_out<T, THROUGH>(&out[8 * N + e], vdcref); st_out<T, THROUGH>(&out[9 * N + e], qref);
    st_out<T, THROUGH>(&out[10 * N + e], tfrac);                   // tStart/max_sim_time (:540)
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
                                         bool stats_local = false, int gen_j0 = -1, int gen_j1 = -1, int episode_now = -1,
                                         bool stats_atomic = false) {
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
            if (stats_atomic) {
                // (step_sat_hybrid_kernel only: the four quarters of a 64-env block are workgroups of their own and add to the block's
                //  slot as they finish -- the three counters are exact whatever the order, the reward sum to the last bit of a float)
                float *f = (float *)slot;
                atomicAdd(f, r); atomicAdd(f + 1, d); atomicAdd(f + 2, nf); atomicAdd(f + 3, ba);
            } else {
                if (!stats_fresh) t += *slot;
                *slot = t;
            }
        }
    }
}

}  // namespace
