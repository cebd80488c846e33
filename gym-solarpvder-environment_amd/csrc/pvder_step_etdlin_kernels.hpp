// pvder_step_etdlin_kernels.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// Etdlin: the careful pass (envs at the current limit), the lean and the full per-step bodies, the step kernels, per-lane parameter sets.
#pragma once

namespace {

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
    if constexpr (PH == 3) {
        // auto mode of the duty-cycle-limit kernel: an env that STARTS a step beyond 98.75 % of the limit is reported (EtdK.sat2_flag;
        // the host lets step_sat_hybrid_kernel follow the launches from there on).  The env itself stays here: it leaves this pass
        // when its duty cycle PEAKS beyond 99.8 % (sat2_smooth), as ever -- the 23-state tail decides about the hand-over (kHybTouch2).
        // (the first form sent every env beyond 98.75 % to the tail: 83 -> 141 us per step at 1 048 576 envs, where some always are)
        const float mr = __builtin_fmaf(A.P.KpG, ys[4], ys[2]), mi = __builtin_fmaf(A.P.KpG, ys[5], ys[3]);
        if (L.run && __builtin_fmaf(mi, mi, mr * mr) > A.Ec.sat2_flag && A.hyb_flag != nullptr) *(volatile int *)A.hyb_flag = 1;
    }
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

template <int PH, bool T3>
__device__ __forceinline__ void step_etdlin_entry(float *fstate, int32_t *istate, const uint8_t *action, const float *lcoef, int n_env,
                                                  int ksub_word, const float *k4coef, const LinArgs &G) {
    using namespace pvder;
    hyb_progress<kLinHeadBytes>();
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
    hyb_progress<0>();
    const EtdArgs &A = G.A;
    constexpr int NS = 23;
    const size_t N = (size_t)A.n_env;
    const int lane = threadIdx.x & 63;
    Lane<float> L;
    float ys[NS];
    prologue<float, NS>(A.C, A.fstate, A.istate, A.action, A.n_env, L, ys);
    {   // (auto mode of the duty-cycle-limit kernel: as lin_step_lean)
        const float mr = __builtin_fmaf(A.P.KpG, ys[4], ys[2]), mi = __builtin_fmaf(A.P.KpG, ys[5], ys[3]);
        if (L.run && __builtin_fmaf(mi, mi, mr * mr) > A.Ec.sat2_flag && A.hyb_flag != nullptr) *(volatile int *)A.hyb_flag = 1;
    }
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
                    leave |= pk2 > A.Ec.sat2_smooth || ir2a > irg;
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

}  // namespace
