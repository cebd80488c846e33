// pvder_step_rk45.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// The fp64 tolerance-match paths: adaptive Dormand-Prince RK45 and etdlin64.
#pragma once

namespace {

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

}  // namespace
