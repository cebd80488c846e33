// pvder_step_rk4.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// Classic RK4 micro-stepping (generic template and the packed three-phase form).
#pragma once

namespace {

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

}  // namespace
