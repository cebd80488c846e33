// pvder_step_etdlin.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// Etdlin: the linearised exponential integrator on the matrix cores (segments, tick + ETD3RK sets, the Krogstad block).
#pragma once

namespace {

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
    // served rollouts (rollout_body<PH, false, true>: pvder_plan_rollout_served)
    const int *mailbox;              // [1], host-coherent: action rows 0 .. *mailbox - 1 are there to be read (polled by the RELAY only)
    int *done;                       // [2], host-coherent: min / max over the workgroups of the steps whose outputs have left their CU (the relay's)
    int *sv_dev;                     // device words: see served_relay (csrc/pvder_step_rollout.hpp)
    int sv_paced;                    // 1: a workgroup that leaves the loop for the finish kernel asks every workgroup to leave (kServedStop)
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
    near_ilim = ir2a > irg || mmax > Ec.sat2_smooth;
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
    near_ilim = ir2a > irg || ir2y > irg || mmax > Ec.sat2_smooth;     // (EARLY = the second pass)
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
                else defer |= pk2 > Ec.sat2_smooth || ir2a > irg;
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
                else defer |= pk2 > Ec.sat2_smooth || ir2a > irg;
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

}  // namespace
