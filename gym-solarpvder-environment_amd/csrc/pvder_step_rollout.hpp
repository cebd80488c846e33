// pvder_step_rollout.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// Fused rollouts: the env steps of a window inside one kernel + the finish kernel.
#pragma once

namespace {

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
// SV = true (pvder_plan_rollout_served; round 6, VERDICT r5 #5 option B -- a resident step server): the fused loop takes step k only once
// the caller has POSTED its action row (*Rk.mailbox > k: a host-coherent word, polled one step ahead so that the poll's round trip
// sits behind the integration), and tells the caller how far its outputs are (Rk.done[wg]).  A wait is bounded (kServedWaitTicks
// of the 100 MHz clock): on a timeout the WORKGROUP leaves the loop as it leaves for a step it cannot carry -- its state as of the
// last step taken -- and the kernel ends: never a hung GPU.  No agent-scope fence anywhere.
// The mailbox is polled over PCIe by ONE thread of the launch -- a relay workgroup without envs (the last of the grid), which
// copies what it reads into a device word (Rk.sv_dev[0]) the env workgroups poll: 1024 waves reading a host word every step are
// 1024 PCIe reads per step, 82 us of them (measured: profiles/r06_ab_served_rollout.txt).
// STOP (Rk.sv_paced: a caller that posts row by row, against the reports): a workgroup that leaves the loop for a step it cannot
// carry (an episode end, an env off the standard path) needs the finish kernel, which runs when THIS launch has ended -- and the
// other workgroups would sit in their waits for rows the caller posts only when EVERY workgroup has reported.  So it says so (Rk.sv_dev[2]), the relay passes it on in bit 30 of every workgroup's word, and everybody leaves at
// their next barrier: the launch ends, the finish kernel takes each workgroup through ONE step, the caller launches the pair again.
constexpr unsigned long long kServedWaitTicks = 20000000ull;            // 0.2 s
constexpr int kServedStop = 1 << 30;
__device__ __forceinline__ int served_posted(const int *word) {
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true: rows 0 .. need - 1 are there; false: a stop was asked for, or the wait ran out
__device__ __forceinline__ bool served_wait(const int *word, int need) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const int w = served_posted(word);
        if (w & kServedStop) return false;
        if (w >= need) return true;
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > kServedWaitTicks) return false;
    }
}

// sv_dev (device words, a cache line each): [1] = workgroups that left; [16 (2 + 2 wg)] = the relay's copy of *mailbox for workgroup
// wg (1024 waves polling ONE device word every step cost 10 us per step: the same-address serialisation the statistics' atomics once
// showed); [16 (3 + 2 wg)] = steps whose outputs have left workgroup wg's CU.  The relay reduces those to Rk.done[0] = min, [1] = max
// (host words): a workgroup that reported to the host itself waited, every step, for its own PCIe write to be acknowledged (+9.5 us).
__device__ __forceinline__ const int *served_word(const int *sv_dev) { return sv_dev + 16 * (2 + 2 * (int)blockIdx.x); }
__device__ __forceinline__ int *served_done(int *sv_dev) { return sv_dev + 16 * (3 + 2 * (int)blockIdx.x); }
__device__ __forceinline__ void served_relay(const RollK &Rk) {
    __shared__ int relay_p, relay_stop, relay_min, relay_max, relay_ask;
    unsigned long long tlast = wall_clock64();
    int last = -1;
    const int n_wg = (int)gridDim.x - 1;
    for (;;) {
        if (threadIdx.x == 0) {
            const int p = __hip_atomic_load(Rk.mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p != last) { last = p; tlast = wall_clock64(); }
            relay_p = p < kServedStop ? p : kServedStop - 1; relay_min = 0x7fffffff; relay_max = 0;
            relay_ask = __hip_atomic_load(&Rk.sv_dev[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 ? kServedStop : 0;
            relay_stop = (__hip_atomic_load(&Rk.sv_dev[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= n_wg      // everyone has left
                          || wall_clock64() - tlast > 4 * kServedWaitTicks) ? 1 : 0;  // nobody posts and somebody stays: they time out, so do we
        }
        __syncthreads();
        const int p = relay_p, stop = relay_stop;
        for (int w = (int)threadIdx.x; w < n_wg; w += kBlock) {
            __hip_atomic_store(&Rk.sv_dev[16 * (2 + 2 * w)], p | relay_ask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const int d = __hip_atomic_load(&Rk.sv_dev[16 * (3 + 2 * w)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            atomicMin(&relay_min, d); atomicMax(&relay_max, d);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&Rk.done[0], relay_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&Rk.done[1], relay_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (stop) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) {          // (the next launch counts from 0, and nobody has asked it to stop)
        __hip_atomic_store(&Rk.sv_dev[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&Rk.sv_dev[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int w = (int)threadIdx.x; w < n_wg; w += kBlock)
        __hip_atomic_store(&Rk.sv_dev[16 * (2 + 2 * w)], relay_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// CP = true (round 6, VERDICT r5 #2b): an env at the CURRENT limit stays in the loop -- its env step is redone from the state it started
// with by the careful pass (lin_integrate_careful: the very call of the per-step body, same inputs, same bits) instead of sending its
// whole workgroup through the finish kernel, step after step, for as long as the env sits there (a held Q_ref action: the last 60
// steps of the episode).  The careful pass's registers make this a one-wave-per-SIMD kernel: batches up to 65 536 envs
// (step_etdlin_rollout_cp_kernel; beyond that the two-wave kernel without it).
template <int PH, bool AG, bool SV = false, bool CP = false>
__device__ __forceinline__ void rollout_body(const RollArgs &Q) {
    using namespace pvder;
    constexpr int NS = 6 * PH + 5;
    if constexpr (SV) {
        if (blockIdx.x == gridDim.x - 1) { served_relay(Q.Rk); return; }
        if (Q.Rk.resume && Q.Rk.progress[blockIdx.x] >= Q.Rk.n_steps) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&Q.Rk.sv_dev[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
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
    bool sv_timeout = false;         // (SV) a wait for the caller ran out: the workgroup leaves at its next barrier
    if constexpr (SV) {
        // (a resumed launch: the steps before k were taken by earlier launches -- the finish kernel among them -- and are complete)
        if (threadIdx.x == 0) __hip_atomic_store(served_done(Q.Rk.sv_dev), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (k < Q.Rk.n_steps) {
            if (bail) {                  // (this workgroup needs the finish kernel before anything else: everybody out)
                if (threadIdx.x == 0 && Q.Rk.sv_paced) __hip_atomic_store(&Q.Rk.sv_dev[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (!served_wait(served_word(Q.Rk.sv_dev), k + 1)) {
                bail = true;
            }
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
            int posted = 0;
            if constexpr (SV) {
                posted = served_posted(served_word(Q1.Rk.sv_dev));        // (asked for here, looked at behind the integration)
            } else if constexpr (!AG) {
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
                bool careful;
                if constexpr (CP) {
                    // the standard pass as the plain loop runs it (one flag, wave-level early outs) ...
                    leave = lin_integrate<PH, false>(Q1.G, lane, R, Y, qref, vdcref, ec, sg, true, false, careful, traced ? 0 : 1);
                }
                // (a lane that leaves with its duty cycle beyond 95 % of its limit at the START of the step will most likely trip the careful
                //  pass's own guard too -- the Q_ref-up ramp from step 110 --: its workgroup leaves at once, nothing is spent on it here.
                //  Time only: the per-step body redoes the step from the same state either way.)
                bool crowded = false;
                if constexpr (CP) {
                    const v2 m0 = fma2(bc2(Q1.G.A.P.KpG), mk2(Y1.x, Y1.y), mk2(Y0.z, Y0.w));
                    crowded = leave && __builtin_fmaf(m0.y, m0.y, m0.x * m0.x) > 0.9025f * Q1.G.A.P.mlim2;
                }
                if (CP && __builtin_amdgcn_readfirstlane(__any((int)leave)) && !__builtin_amdgcn_readfirstlane(__any((int)crowded))) {
                    // ... and, where a lane leaves, once more from the state the step started with, as lin_step_body runs it: it
                    // names the lanes for the 23-state path and the lanes for the careful pass (same arithmetic for everybody else)
                    Y[0] = Y0; Y[1] = Y1; Y[2] = Y2;
                    bool defer = lin_integrate<PH, true>(Q1.G, lane, R, Y, qref, vdcref, ec, sg, true, false, careful, 1);
                    defer = defer && !careful;
                    if (__builtin_amdgcn_readfirstlane(__any((int)careful))) {
                        if (careful) { Y[0] = Y0; Y[1] = Y1; Y[2] = Y2; }
                        const bool dc = lin_integrate_careful<PH>(Q1.G, lane, R, Y, qref, vdcref, ec, sg, careful, false, 1.f);
                        defer |= careful && dc;
                    }
                    leave = defer;
                }
                if constexpr (!CP) {
                    // (an env at the current limit leaves too: its step belongs to the per-step body's careful pass)
                    leave = lin_integrate<PH, false>(Q1.G, lane, R, Y, qref, vdcref, ec, sg, true, false, careful, traced ? 0 : 1);
                }
            }
            if (traced) PVDER_TRACE_DEP(20, Y[0].x + Y[2].z);
            float chk = fabsf(Y[0].x) + fabsf(Y[0].y) + fabsf(Y[0].z) + fabsf(Y[0].w) + fabsf(Y[1].x) + fabsf(Y[1].y) +
                        fabsf(Y[1].z) + fabsf(Y[1].w) + fabsf(Y[2].x) + fabsf(Y[2].y) + fabsf(Y[2].z);
            leave |= !(chk < 1e30f);
            bool asked = false;              // (SV) this workgroup leaves because ANOTHER one asked, or because the caller is gone
            if constexpr (SV) {
                asked = sv_timeout || (posted & kServedStop) != 0;
                posted &= kServedStop - 1;
                __builtin_amdgcn_s_waitcnt(0);                // this wave's stores of the steps before this one have left the CU
            }
            if (__syncthreads_or((int)(leave || asked))) {
                if constexpr (SV) {              // its OWN reason: the others must let the finish kernel run
                    if (__syncthreads_or((int)leave) && threadIdx.x == 0 && roll_args().Rk.sv_paced)
                        __hip_atomic_store(&roll_args().Rk.sv_dev[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                Y[0] = Y0; Y[1] = Y1; Y[2] = Y2;
                qref = qref0; vdcref = vdcref0;
                break;
            }
            bool sv_wait = false;
            if constexpr (SV) {
                if (threadIdx.x == 0) __hip_atomic_store(served_done(Q1.Rk.sv_dev), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (k + 1 < Q1.Rk.n_steps) {
                    // the next step's action row: there already (the caller is ahead), or waited for behind this step's outputs
                    if (posted >= k + 2) a_next = ld(&Q1.Rk.actions[(size_t)(k + 1) * N + ec]);
                    else sv_wait = true;
                }
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
                // (SV: written THROUGH the L2 -- the report that follows says "a kernel that starts now, on any stream, reads them")
                write_obs<float, SV>(o, N, e, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ys[6 * PH], ppv, vdcref, qref,
                                     (float)ksub * (float)(1.0 / 60.0) * B.C.inv_tmax);
                st_out<float, SV>(&o[PVDER_O_REWARD * N + e], last_reward);
                st_out<float, SV>(&o[PVDER_O_DONE * N + e], 0.f);
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
            if constexpr (SV) {
                if (sv_wait) {
                    const RollArgs &Q4 = roll_args();
                    if (served_wait(served_word(Q4.Rk.sv_dev), k + 2)) a_next = ld(&Q4.Rk.actions[(size_t)(k + 1) * (size_t)Q4.G.A.n_env + (e < Q4.G.A.n_env ? e : Q4.G.A.n_env - 1)]);
                    else sv_timeout = true;
                }
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
    if constexpr (SV) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(served_done(Q3.Rk.sv_dev), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(&Q3.Rk.sv_dev[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int PH>
__global__ __launch_bounds__(kBlock)
#ifdef PVDER_ROLLOUT_OCC3
__attribute__((amdgpu_waves_per_eu(3)))      // 168 VGPRs: three waves per SIMD for batches beyond 131 072 envs (rounds 2-3; with the
#endif                                       // Krogstad block resident next to the ETD3RK set the loop needs ~200: two waves per SIMD)
void step_etdlin_rollout_kernel(const RollArgs Q) { rollout_body<PH, false>(Q); }

template <int PH>
__global__ __launch_bounds__(kBlock) void step_etdlin_rollout_agent_kernel(const RollArgs Q) { rollout_body<PH, true>(Q); }

// ... with the careful pass inside the loop (batches up to 65 536 envs: one wave per SIMD is all they give the chip anyway)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void step_etdlin_rollout_cp_kernel(const RollArgs Q) {
    rollout_body<3, false, false, true>(Q);
}

// the fused loop SERVED: every env step waits for its action row to be posted (pvder_plan_rollout_served; grid = workgroups + the relay)
// (WITHOUT the careful pass: its registers would make this a one-wave-per-SIMD kernel, and the relay workgroup needs a place next to
//  256 resident env workgroups)
__global__ __launch_bounds__(kBlock) void step_etdlin_served_kernel(const RollArgs Q) { rollout_body<3, false, true, false>(Q); }

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

}  // namespace
