// pvder_kernels.hip -- hand-written gfx950 kernels + C ABI (include/pvder_hip.h) of the batched
// PVDER-v0 hot path.  One env per lane; structure-of-arrays state in HBM (env index fastest, so
// every global access of a wave is one fully coalesced 256-byte row segment); model constants in
// SGPRs (kernel argument); the whole env transition
//     action_calc -> n_sub x micro RK4 steps -> reward_calc -> done -> state
// (PVDER_env.py:138-196, :198-229, :231-301, :531-542) is ONE launch with the 23 states held in
// registers for the entire step; wave-level DPP reduction + one atomic per wave for the
// reward-sum / done-count statistics (env_utilities.py:32-38 aggregate).
//
// No MFMA: there is no dense contraction in this path (per-env 23-state nonlinear ODE).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include "pvder_hip.h"
#include "pvder_model.hpp"
#include "pvder_model_pk.hpp"

namespace {

using pvder::ModelK;
using pvder::SubK;

constexpr int kBlock = 256;   // 4 waves: one per SIMD of a CU

template <typename T>
struct StepK {
    int n_sub, micro, max_ksub, max_steps, goal, discrete, ev_s0, ev_ds, ev_count;
    T dq, dv, inv_tmax, p_t, q_t, rtol, atol;
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

// event-table row active at sim-time-step g (events at ksub = s0 + j*ds, j = 0..K-1; an event
// at exactly g is already active: grid_events(t) returns the latest event with T <= t)
__device__ __forceinline__ int event_row(int g, int s0, int ds, int K) {
    if (K <= 0 || g < s0) return 0;
    int j = (g - s0) / ds + 1;
    return j > K ? K : j;
}

// ---- wave64 sum via DPP row/bank shifts (no LDS), result valid in lane 63 -> broadcast by readlane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ T reward_of(const StepK<T> &C, const ModelK<T> &P, T vrms, T Ppcc, T Qpcc) {
    // PVDER_env.py:231-301 -- one 'required' term per goal (:78-93 via :448-452)
    T err, lo, hi, cont;
    if (C.goal == PVDER_GOAL_VOLTAGE) {            // voltage_error :276-287
        const T d = vrms - P.vrms_ref;
        err = fabs(d) / fabs(P.vrms_ref); lo = (T)0.01; hi = (T)0.05; cont = -(d * d);
    } else if (C.goal == PVDER_GOAL_POWER) {       // power_error :290-299
        const T d = Ppcc - C.p_t;
        err = fabs(d) / C.p_t; lo = (T)0.01; hi = (T)0.03; cont = -(d * d);
    } else {                                       // Q_error :263-275
        T qt = C.q_t;
        const T d = Qpcc - qt;
        cont = -(d * d);
        if (qt == (T)0) qt = (T)1e-6;              // :266-267 (discrete branch only)
        err = fabs(Qpcc - qt) / fabs(qt); lo = (T)0.01; hi = (T)0.05;
    }
    if (!C.discrete) return cont;
    return err <= lo ? (T)1 : (err >= hi ? (T)-5 : (T)-1);
}

// =============================================================================================
// fused env transition, fixed-step classic RK4 with Kahan-compensated state update
// =============================================================================================
template <typename T, int PH>
__global__ __launch_bounds__(kBlock) void step_rk4_kernel(
    const ModelK<T> P, const StepK<T> C, T *__restrict__ fstate, int32_t *__restrict__ istate,
    const float *__restrict__ ev_vg, const float *__restrict__ ev_si,
    const uint8_t *__restrict__ action, T *__restrict__ out, float *__restrict__ stats, int n_env) {
    constexpr int NS = 6 * PH + 5;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    const bool live = e < n_env;
    const int ec = live ? e : n_env - 1;           // clamp: tail lanes recompute env n-1, never store
    const size_t N = (size_t)n_env;

    int ksub = istate[PVDER_I_KSUB * N + ec];
    int steps = istate[PVDER_I_STEPS * N + ec];
    int status = istate[PVDER_I_STATUS * N + ec];
    const bool run = live && status == 0;          // finished envs return the stale tuple (:145-152)

    T y[NS], comp[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        y[k] = fstate[(size_t)k * N + ec];
        comp[k] = (T)0;
    }
    T qref = fstate[(size_t)NS * N + ec];
    T vdcref = fstate[(size_t)(NS + 1) * N + ec];

    // ---- action_calc (PVDER_env.py:198-229)
    const int a = action[ec];
    float bad_action = 0.f;
    if (a == 1) qref += C.dq;
    else if (a == 2) qref -= C.dq;
    else if (a == 3) vdcref += C.dv;
    else if (a == 4) vdcref -= C.dv;
    else if (a != 0) bad_action = 1.f;

    const T h = (T)(1.0 / 60.0) / (T)C.micro;
    const T h2 = (T)0.5 * h, h6 = h * (T)(1.0 / 6.0);
    const T tq = (T)2 / (T)C.micro;                // 2*theta advances 2 turns per 1/60 s

    SubK<T, PH> S;
    if (run) {
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            const T vgm = (T)ev_vg[(size_t)row * N + ec];
            const T sin_ = (T)ev_si[(size_t)row * N + ec];
            pvder::make_sub<T, PH>(P, vgm, sin_, S);
            for (int j = 0; j < C.micro; ++j) {
                T k[NS], ks[NS], yt[NS];
                const T t0 = (T)j * tq, t1 = ((T)j + (T)0.5) * tq, t2 = ((T)j + (T)1) * tq;
                pvder::rhs<T, PH>(P, S, y, t0, qref, vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] = k[q]; yt[q] = y[q] + h2 * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t1, qref, vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] += (T)2 * k[q]; yt[q] = y[q] + h2 * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t1, qref, vdcref, k);
#pragma unroll
                for (int q = 0; q < NS; ++q) { ks[q] += (T)2 * k[q]; yt[q] = y[q] + h * k[q]; }
                pvder::rhs<T, PH>(P, S, yt, t2, qref, vdcref, k);
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
        ksub += C.n_sub;
        steps += 1;
    }

    // ---- epilogue: reward_calc / done / state at (y(tStop), tStop)
    {
        const int row = event_row(ksub, C.ev_s0, C.ev_ds, C.ev_count);
        const T vgm = (T)ev_vg[(size_t)row * N + ec];
        const T sin_ = (T)ev_si[(size_t)row * N + ec];
        pvder::make_sub<T, PH>(P, vgm, sin_, S);
    }
    T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
    pvder::outputs<T, PH>(P, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
    T reward = reward_of<T>(C, P, vrms, Ppcc, Qpcc);
    T chk = (T)0;
#pragma unroll
    for (int q = 0; q < NS; ++q) chk += fabs(y[q]);
    float nonfinite = 0.f;
    if (run) {
        if (!(chk < (T)1e30)) {                    // NaN/Inf: CONVERGENCE_FAILURE branch (:169-172)
            status = 2; reward = (T)-100; nonfinite = 1.f;
        } else if (ksub >= C.max_ksub || (C.max_steps > 0 && steps >= C.max_steps)) {
            status = 1;                            // tStop >= max_sim_time (:183) / TimeLimit
        }
    }
    if (run) {
#pragma unroll
        for (int q = 0; q < NS; ++q) fstate[(size_t)q * N + e] = y[q];
        fstate[(size_t)NS * N + e] = qref;
        fstate[(size_t)(NS + 1) * N + e] = vdcref;
        istate[PVDER_I_KSUB * N + e] = ksub;
        istate[PVDER_I_STEPS * N + e] = steps;
        istate[PVDER_I_STATUS * N + e] = status;
        out[0 * N + e] = iaR;   out[1 * N + e] = iaI;
        out[2 * N + e] = vaR;   out[3 * N + e] = vaI;
        out[4 * N + e] = Ppcc;  out[5 * N + e] = Qpcc;
        out[6 * N + e] = y[6 * PH];  out[7 * N + e] = ppv;
        out[8 * N + e] = vdcref;     out[9 * N + e] = qref;
        out[10 * N + e] = (T)ksub * (T)(1.0 / 60.0) * C.inv_tmax;      // tStart/max_sim_time (:540)
        out[PVDER_O_REWARD * N + e] = reward;
        out[PVDER_O_DONE * N + e] = status != 0 ? (T)1 : (T)0;
    }
    // ---- wave-level reduction of the step statistics, one atomic per wave
    if (stats) {
        const float r = wave_sum(run ? (float)reward : 0.f);
        const float d = wave_sum(run && status != 0 ? 1.f : 0.f);
        const float nf = wave_sum(nonfinite);
        const float ba = wave_sum(live ? bad_action : 0.f);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats[0], r);
            if (d != 0.f) atomicAdd(&stats[1], d);
            if (nf != 0.f) atomicAdd(&stats[2], nf);
            if (ba != 0.f) atomicAdd(&stats[3], ba);
        }
    }
}

// =============================================================================================
// fused env transition, fixed-step RK4, three-phase fp32, packed-VALU formulation (the hot kernel)
// =============================================================================================
__global__ __launch_bounds__(kBlock) void step_rk4_pk_kernel(
    const ModelK<float> P, const StepK<float> C, float *__restrict__ fstate, int32_t *__restrict__ istate,
    const float *__restrict__ ev_vg, const float *__restrict__ ev_si,
    const uint8_t *__restrict__ action, float *__restrict__ out, float *__restrict__ stats, int n_env) {
    using pvder::v2;
    constexpr int NS = 23, NP = pvder::NP;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    const bool live = e < n_env;
    const int ec = live ? e : n_env - 1;
    const size_t N = (size_t)n_env;

    int ksub = istate[PVDER_I_KSUB * N + ec];
    int steps = istate[PVDER_I_STEPS * N + ec];
    int status = istate[PVDER_I_STATUS * N + ec];
    const bool run = live && status == 0;

    v2 y[NP], comp[NP];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        y[k] = pvder::mk2(fstate[(size_t)(2 * k) * N + ec], fstate[(size_t)(2 * k + 1) * N + ec]);
        comp[k] = pvder::bc2(0.f);
    }
    y[11] = pvder::mk2(fstate[(size_t)22 * N + ec], 0.f);
    comp[11] = pvder::bc2(0.f);
    float qref = fstate[(size_t)NS * N + ec];
    float vdcref = fstate[(size_t)(NS + 1) * N + ec];

    const int a = action[ec];
    float bad_action = 0.f;
    if (a == 1) qref += C.dq;
    else if (a == 2) qref -= C.dq;
    else if (a == 3) vdcref += C.dv;
    else if (a == 4) vdcref -= C.dv;
    else if (a != 0) bad_action = 1.f;

    const float h = (float)(1.0 / 60.0) / (float)C.micro;
    const v2 h2 = pvder::bc2(0.5f * h), h1 = pvder::bc2(h), h6 = pvder::bc2(h * (float)(1.0 / 6.0));
    const v2 two = pvder::bc2(2.f);
    const float tq = 2.f / (float)C.micro;
    const pvder::PkConst K = pvder::make_pkconst(P);

    pvder::SubPk S;
    if (run) {
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            pvder::make_sub_pk(P, ev_vg[(size_t)row * N + ec], ev_si[(size_t)row * N + ec], S);
            for (int j = 0; j < C.micro; ++j) {
                v2 k[NP], ks[NP], yt[NP];
                const float t0 = (float)j * tq, t1 = ((float)j + 0.5f) * tq, t2 = ((float)j + 1.f) * tq;
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
                        // Kahan-compensated update (DESIGN.md "fp32 hazards")
                        const v2 inc = pvder::fma2(h6, ks[q] + k[q], -comp[q]);
                        const v2 tsum = y[q] + inc;
                        comp[q] = (tsum - y[q]) - inc;
                        y[q] = tsum;
                    }
                }
            }
        }
        ksub += C.n_sub;
        steps += 1;
    }

    // ---- epilogue (scalar form, once per step)
    float ys[NS];
#pragma unroll
    for (int k = 0; k < 11; ++k) { ys[2 * k] = y[k].x; ys[2 * k + 1] = y[k].y; }
    ys[22] = y[11].x;
    SubK<float, 3> S1;
    {
        const int row = event_row(ksub, C.ev_s0, C.ev_ds, C.ev_count);
        pvder::make_sub<float, 3>(P, ev_vg[(size_t)row * N + ec], ev_si[(size_t)row * N + ec], S1);
    }
    float iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
    pvder::outputs<float, 3>(P, S1, ys, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
    float reward = reward_of<float>(C, P, vrms, Ppcc, Qpcc);
    float chk = 0.f;
#pragma unroll
    for (int q = 0; q < NS; ++q) chk += fabsf(ys[q]);
    float nonfinite = 0.f;
    if (run) {
        if (!(chk < 1e30f)) {
            status = 2; reward = -100.f; nonfinite = 1.f;
        } else if (ksub >= C.max_ksub || (C.max_steps > 0 && steps >= C.max_steps)) {
            status = 1;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) fstate[(size_t)q * N + e] = ys[q];
        fstate[(size_t)NS * N + e] = qref;
        fstate[(size_t)(NS + 1) * N + e] = vdcref;
        istate[PVDER_I_KSUB * N + e] = ksub;
        istate[PVDER_I_STEPS * N + e] = steps;
        istate[PVDER_I_STATUS * N + e] = status;
        out[0 * N + e] = iaR;   out[1 * N + e] = iaI;
        out[2 * N + e] = vaR;   out[3 * N + e] = vaI;
        out[4 * N + e] = Ppcc;  out[5 * N + e] = Qpcc;
        out[6 * N + e] = ys[18];     out[7 * N + e] = ppv;
        out[8 * N + e] = vdcref;     out[9 * N + e] = qref;
        out[10 * N + e] = (float)ksub * (float)(1.0 / 60.0) * C.inv_tmax;
        out[PVDER_O_REWARD * N + e] = reward;
        out[PVDER_O_DONE * N + e] = status != 0 ? 1.f : 0.f;
    }
    if (stats) {
        const float r = wave_sum(run ? reward : 0.f);
        const float d = wave_sum(run && status != 0 ? 1.f : 0.f);
        const float nf = wave_sum(nonfinite);
        const float ba = wave_sum(live ? bad_action : 0.f);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats[0], r);
            if (d != 0.f) atomicAdd(&stats[1], d);
            if (nf != 0.f) atomicAdd(&stats[2], nf);
            if (ba != 0.f) atomicAdd(&stats[3], ba);
        }
    }
}

// =============================================================================================
// fused env transition, adaptive Dormand-Prince RK45 (per-lane step size; fp64 tolerance-match path)
// =============================================================================================
template <typename T, int PH>
__global__ __launch_bounds__(kBlock) void step_rk45_kernel(
    const ModelK<T> P, const StepK<T> C, T *__restrict__ fstate, int32_t *__restrict__ istate,
    const float *__restrict__ ev_vg, const float *__restrict__ ev_si,
    const uint8_t *__restrict__ action, T *__restrict__ out, T *__restrict__ hstep,
    float *__restrict__ stats, int n_env) {
    constexpr int NS = 6 * PH + 5;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    const bool live = e < n_env;
    const int ec = live ? e : n_env - 1;
    const size_t N = (size_t)n_env;

    int ksub = istate[PVDER_I_KSUB * N + ec];
    int steps = istate[PVDER_I_STEPS * N + ec];
    int status = istate[PVDER_I_STATUS * N + ec];
    const bool run = live && status == 0;

    T y[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) y[k] = fstate[(size_t)k * N + ec];
    T qref = fstate[(size_t)NS * N + ec];
    T vdcref = fstate[(size_t)(NS + 1) * N + ec];
    T hn = hstep[ec];
    if (!(hn > (T)0)) hn = (T)1e-6;

    const int a = action[ec];
    float bad_action = 0.f;
    if (a == 1) qref += C.dq;
    else if (a == 2) qref -= C.dq;
    else if (a == 3) vdcref += C.dv;
    else if (a == 4) vdcref -= C.dv;
    else if (a != 0) bad_action = 1.f;

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

    SubK<T, PH> S;
    if (run) {
        for (int s = 0; s < C.n_sub; ++s) {
            const int row = event_row(ksub + s, C.ev_s0, C.ev_ds, C.ev_count);
            pvder::make_sub<T, PH>(P, (T)ev_vg[(size_t)row * N + ec], (T)ev_si[(size_t)row * N + ec], S);
            T t = (T)0;
            T k1[NS];
            pvder::rhs<T, PH>(P, S, y, (T)0, qref, vdcref, k1);
            // bounded: a lane can take at most kMaxIter attempts per 1/60 s (h >= ~1e-9 s)
            for (int it = 0; it < 20000000 && t < Tend * (T)(1.0 - 1e-14); ++it) {
                T h = hn;
                bool last = false;
                if (t + h >= Tend) { h = Tend - t; last = true; }
                T k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], yt[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a21 * k1[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c2 * h) * tq, qref, vdcref, k2);
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a31 * k1[q] + a32 * k2[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c3 * h) * tq, qref, vdcref, k3);
#pragma unroll
                for (int q = 0; q < NS; ++q) yt[q] = y[q] + h * (a41 * k1[q] + a42 * k2[q] + a43 * k3[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c4 * h) * tq, qref, vdcref, k4);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    yt[q] = y[q] + h * (a51 * k1[q] + a52 * k2[q] + a53 * k3[q] + a54 * k4[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + c5 * h) * tq, qref, vdcref, k5);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    yt[q] = y[q] + h * (a61 * k1[q] + a62 * k2[q] + a63 * k3[q] + a64 * k4[q] + a65 * k5[q]);
                pvder::rhs<T, PH>(P, S, yt, (t + h) * tq, qref, vdcref, k6);
                T errv[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    yt[q] = y[q] + h * (b1 * k1[q] + b3 * k3[q] + b4 * k4[q] + b5 * k5[q] + b6 * k6[q]);
                    errv[q] = e1 * k1[q] + e3 * k3[q] + e4 * k4[q] + e5 * k5[q] + e6 * k6[q];
                }
                pvder::rhs<T, PH>(P, S, yt, (t + h) * tq, qref, vdcref, k2);   // k7 (reuses k2)
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
        }
        ksub += C.n_sub;
        steps += 1;
    }

    {
        const int row = event_row(ksub, C.ev_s0, C.ev_ds, C.ev_count);
        pvder::make_sub<T, PH>(P, (T)ev_vg[(size_t)row * N + ec], (T)ev_si[(size_t)row * N + ec], S);
    }
    T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
    pvder::outputs<T, PH>(P, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
    T reward = reward_of<T>(C, P, vrms, Ppcc, Qpcc);
    T chk = (T)0;
#pragma unroll
    for (int q = 0; q < NS; ++q) chk += fabs(y[q]);
    float nonfinite = 0.f;
    if (run) {
        if (!(chk < (T)1e30)) {
            status = 2; reward = (T)-100; nonfinite = 1.f;
        } else if (ksub >= C.max_ksub || (C.max_steps > 0 && steps >= C.max_steps)) {
            status = 1;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) fstate[(size_t)q * N + e] = y[q];
        fstate[(size_t)NS * N + e] = qref;
        fstate[(size_t)(NS + 1) * N + e] = vdcref;
        hstep[e] = hn;
        istate[PVDER_I_KSUB * N + e] = ksub;
        istate[PVDER_I_STEPS * N + e] = steps;
        istate[PVDER_I_STATUS * N + e] = status;
        out[0 * N + e] = iaR;   out[1 * N + e] = iaI;
        out[2 * N + e] = vaR;   out[3 * N + e] = vaI;
        out[4 * N + e] = Ppcc;  out[5 * N + e] = Qpcc;
        out[6 * N + e] = y[6 * PH];  out[7 * N + e] = ppv;
        out[8 * N + e] = vdcref;     out[9 * N + e] = qref;
        out[10 * N + e] = (T)ksub * (T)(1.0 / 60.0) * C.inv_tmax;
        out[PVDER_O_REWARD * N + e] = reward;
        out[PVDER_O_DONE * N + e] = status != 0 ? (T)1 : (T)0;
    }
    if (stats) {
        const float r = wave_sum(run ? (float)reward : 0.f);
        const float d = wave_sum(run && status != 0 ? 1.f : 0.f);
        const float nf = wave_sum(nonfinite);
        const float ba = wave_sum(live ? bad_action : 0.f);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats[0], r);
            if (d != 0.f) atomicAdd(&stats[1], d);
            if (nf != 0.f) atomicAdd(&stats[2], nf);
            if (ba != 0.f) atomicAdd(&stats[3], ba);
        }
    }
}

// =============================================================================================
// masked reset: steady state broadcast + event-table refill + reset observation
// =============================================================================================
template <int NSMAX>
struct Y0K { double v[NSMAX]; };

template <typename T, int PH>
__global__ __launch_bounds__(kBlock) void reset_kernel(
    const ModelK<T> P, const StepK<T> C, const Y0K<23> Y0, const pvder_event_spec spec,
    T *__restrict__ fstate, int32_t *__restrict__ istate, float *__restrict__ ev_vg,
    float *__restrict__ ev_si, const uint8_t *__restrict__ mask, T *__restrict__ out, uint64_t seed,
    int64_t env_id_base, int keep_events, int n_env) {
    constexpr int NS = 6 * PH + 5;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_env) return;
    if (mask && mask[e] == 0) return;
    const size_t N = (size_t)n_env;
    T y[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        y[k] = (T)Y0.v[k];
        fstate[(size_t)k * N + e] = y[k];
    }
    const T qref = (T)0, vdcref = P.vdc_ref0;
    fstate[(size_t)NS * N + e] = qref;
    fstate[(size_t)(NS + 1) * N + e] = vdcref;
    const int episode = istate[PVDER_I_EPISODE * N + e] + 1;
    istate[PVDER_I_KSUB * N + e] = 0;
    istate[PVDER_I_STEPS * N + e] = 0;
    istate[PVDER_I_STATUS * N + e] = 0;
    istate[PVDER_I_EPISODE * N + e] = episode;
    if (!keep_events) {
        // create_random_events (PVDER_env.py:400-411): one event per grid time, type drawn from
        // the ENABLEd list (dict order: insolation, voltage -- :60-61), value U[min,max]
        float vg = 1.0f, si = (float)P.sinsol0;
        ev_vg[e] = vg;
        ev_si[e] = si;
        const int n_en = (spec.enable_insol != 0) + (spec.enable_voltage != 0);
        for (int k = 1; k <= spec.ev_count; ++k) {
            if (n_en > 0) {
                const float ut = counter_uniform(seed, (uint64_t)(env_id_base + e), (uint64_t)episode, k, 0);
                const float uv = counter_uniform(seed, (uint64_t)(env_id_base + e), (uint64_t)episode, k, 1);
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
    if (out) {
        SubK<T, PH> S;
        pvder::make_sub<T, PH>(P, (T)ev_vg[e], (T)ev_si[e], S);
        T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
        pvder::outputs<T, PH>(P, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
        out[0 * N + e] = iaR;   out[1 * N + e] = iaI;
        out[2 * N + e] = vaR;   out[3 * N + e] = vaI;
        out[4 * N + e] = Ppcc;  out[5 * N + e] = Qpcc;
        out[6 * N + e] = y[6 * PH];  out[7 * N + e] = ppv;
        out[8 * N + e] = vdcref;     out[9 * N + e] = qref;
        out[10 * N + e] = (T)0;
        out[PVDER_O_REWARD * N + e] = (T)0;
        out[PVDER_O_DONE * N + e] = (T)0;
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
    return P;
}

template <typename T>
StepK<T> fold_cfg(const pvder_step_cfg *c) {
    StepK<T> C;
    C.n_sub = c->n_sub; C.micro = c->micro; C.max_ksub = c->max_ksub; C.max_steps = c->max_steps;
    C.goal = c->goal; C.discrete = c->discrete; C.ev_s0 = c->ev_s0; C.ev_ds = c->ev_ds;
    C.ev_count = c->ev_count;
    C.dq = (T)c->del_qref; C.dv = (T)c->del_vdcref; C.inv_tmax = (T)c->inv_max_sim_time;
    C.p_t = (T)c->p_target; C.q_t = (T)c->q_target; C.rtol = (T)c->rtol; C.atol = (T)c->atol;
    return C;
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

template <typename T>
int reset_impl(const double *par, const double *y0, const pvder_step_cfg *cfg, const pvder_event_spec *spec,
               T *fstate, int32_t *istate, float *ev_vg, float *ev_si, const uint8_t *mask, T *out,
               uint64_t seed, int64_t env_id_base, int keep_events, int n_env, void *stream) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (!y0 || !spec || !fstate || !istate || !ev_vg || !ev_si) return PVDER_EINVAL;
    if (spec->ev_count != cfg->ev_count) return PVDER_EINVAL;
    const int ph = (int)par[PVDER_PAR_PHASES];
    const ModelK<T> P = fold_params<T>(par, sizeof(T) == 4);
    const StepK<T> C = fold_cfg<T>(cfg);
    Y0K<23> Y0;
    for (int k = 0; k < 23; ++k) Y0.v[k] = k < PVDER_NS(ph) ? y0[k] : 0.0;
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (ph == 3)
        hipLaunchKernelGGL((reset_kernel<T, 3>), grid, block, 0, st, P, C, Y0, *spec, fstate, istate, ev_vg,
                           ev_si, mask, out, seed, env_id_base, keep_events, n_env);
    else
        hipLaunchKernelGGL((reset_kernel<T, 1>), grid, block, 0, st, P, C, Y0, *spec, fstate, istate, ev_vg,
                           ev_si, mask, out, seed, env_id_base, keep_events, n_env);
    return hip_rc(hipGetLastError());
}

}  // namespace

extern "C" {

int pvder_version(void) { return PVDER_ABI_VERSION; }
int pvder_par_count(void) { return PVDER_PAR_COUNT; }
int pvder_block_size(void) { return kBlock; }

int pvder_reset_masked_f32(const double *par, const double *y0, const pvder_step_cfg *cfg,
                           const pvder_event_spec *spec, float *fstate, int32_t *istate, float *ev_vg,
                           float *ev_si, const uint8_t *mask, float *out, uint64_t seed,
                           int64_t env_id_base, int keep_events, int n_env, void *stream) {
    return reset_impl<float>(par, y0, cfg, spec, fstate, istate, ev_vg, ev_si, mask, out, seed, env_id_base,
                             keep_events, n_env, stream);
}

int pvder_reset_masked_f64(const double *par, const double *y0, const pvder_step_cfg *cfg,
                           const pvder_event_spec *spec, double *fstate, int32_t *istate, float *ev_vg,
                           float *ev_si, const uint8_t *mask, double *out, uint64_t seed,
                           int64_t env_id_base, int keep_events, int n_env, void *stream) {
    return reset_impl<double>(par, y0, cfg, spec, fstate, istate, ev_vg, ev_si, mask, out, seed, env_id_base,
                              keep_events, n_env, stream);
}

int pvder_step_rk4_f32(const double *par, const pvder_step_cfg *cfg, float *fstate, int32_t *istate,
                       const float *ev_vg, const float *ev_si, const uint8_t *action, float *out,
                       float *stats, int zero_stats, int n_env, void *stream) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (!fstate || !istate || !ev_vg || !ev_si || !action || !out) return PVDER_EINVAL;
    const int ph = (int)par[PVDER_PAR_PHASES];
    const ModelK<float> P = fold_params<float>(par, true);
    const StepK<float> C = fold_cfg<float>(cfg);
    hipStream_t st = (hipStream_t)stream;
    if (stats && zero_stats) {
        hipError_t e = hipMemsetAsync(stats, 0, 4 * sizeof(float), st);
        if (e != hipSuccess) return hip_rc(e);
    }
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    // PVDER_RK4_VARIANT=generic selects the un-packed template (A/B runs); default = packed kernel
    static const bool generic = []() { const char *v = getenv("PVDER_RK4_VARIANT"); return v && v[0] == 'g'; }();
    if (ph == 3 && !generic)
        hipLaunchKernelGGL(step_rk4_pk_kernel, grid, block, 0, st, P, C, fstate, istate, ev_vg, ev_si, action, out,
                           stats, n_env);
    else if (ph == 3)
        hipLaunchKernelGGL((step_rk4_kernel<float, 3>), grid, block, 0, st, P, C, fstate, istate, ev_vg, ev_si,
                           action, out, stats, n_env);
    else
        hipLaunchKernelGGL((step_rk4_kernel<float, 1>), grid, block, 0, st, P, C, fstate, istate, ev_vg, ev_si,
                           action, out, stats, n_env);
    return hip_rc(hipGetLastError());
}

int pvder_step_rk45_f64(const double *par, const pvder_step_cfg *cfg, double *fstate, int32_t *istate,
                        const float *ev_vg, const float *ev_si, const uint8_t *action, double *out,
                        double *hstep, float *stats, int zero_stats, int n_env, void *stream) {
    int rc = check_common(par, cfg, n_env);
    if (rc) return rc;
    if (!fstate || !istate || !ev_vg || !ev_si || !action || !out || !hstep) return PVDER_EINVAL;
    if (!(cfg->rtol > 0.0) || !(cfg->atol > 0.0)) return PVDER_EINVAL;
    const int ph = (int)par[PVDER_PAR_PHASES];
    const ModelK<double> P = fold_params<double>(par, false);
    const StepK<double> C = fold_cfg<double>(cfg);
    hipStream_t st = (hipStream_t)stream;
    if (stats && zero_stats) {
        hipError_t e = hipMemsetAsync(stats, 0, 4 * sizeof(float), st);
        if (e != hipSuccess) return hip_rc(e);
    }
    const dim3 grid((n_env + kBlock - 1) / kBlock), block(kBlock);
    if (ph == 3)
        hipLaunchKernelGGL((step_rk45_kernel<double, 3>), grid, block, 0, st, P, C, fstate, istate, ev_vg, ev_si,
                           action, out, hstep, stats, n_env);
    else
        hipLaunchKernelGGL((step_rk45_kernel<double, 1>), grid, block, 0, st, P, C, fstate, istate, ev_vg, ev_si,
                           action, out, hstep, stats, n_env);
    return hip_rc(hipGetLastError());
}

}  // extern "C"
