/* pvder_hip.h -- C ABI of the MI355X (gfx950) batched PVDER-v0 hot path.
 *
 * The reference (sibyjackgrove/gym-SolarPVDER-environment) is pure Python and has no FFI of
 * its own; the boundary it offers is the Gym Env API of `PVDER-v0`
 * (gym_PVDER/envs/PVDER_env.py:138 step, :316 reset) and, inside it, the `self.sim` /
 * `self.sim.PV_model` attribute surface of the third-party `pvder` package.
 * This header is what a ctypes stub inside the reference's `PVDER` class binds in place of
 * `self.sim.run_simulation()` + `reward_calc()` + `state`.
 *
 * Conventions
 *   - plain C, no torch types: device pointers are raw `void*`-compatible pointers obtained from
 *     `tensor.data_ptr()`; `stream` is a `hipStream_t` passed as `void*` (NULL = default stream);
 *   - every entry point is asynchronous on `stream`, performs no allocation and no host sync
 *     (graph-capturable), and returns 0 or a negative `hipError_t`; argument errors return
 *     PVDER_EINVAL before anything is launched;
 *   - the caller owns every buffer.  All per-env arrays are structure-of-arrays with the env
 *     index fastest:  row r of an [R][n_env] array starts at  base + r*n_env.
 */
#ifndef PVDER_HIP_H
#define PVDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVDER_ABI_VERSION 18
#define PVDER_EINVAL (-10001)

/* ---- model parameter vector (host, double): per-unit constants of one DER ------------------
 * Filled by the host from config_der.json (reference: config_der.json:2-21 for derId "50",
 * :66-84 for "10") with pvder's base values; shared by all envs of a launch (kernel argument
 * -> SGPRs). */
enum pvder_par {
    PVDER_PAR_PHASES = 0, /* 3 (ThreePhaseUnbalanced, PVDER_env.py:57) or 1 (SinglePhase, :56) */
    PVDER_PAR_RF, PVDER_PAR_LF, PVDER_PAR_C,
    PVDER_PAR_KVG_RE, PVDER_PAR_KVG_IM, /* v_p = kvg*vg_p + kvi*i_p (stand-alone PCC-LV voltage) */
    PVDER_PAR_KVI_RE, PVDER_PAR_KVI_IM,
    PVDER_PAR_KP_GCC, PVDER_PAR_KI_GCC, PVDER_PAR_KP_DC, PVDER_PAR_KI_DC, PVDER_PAR_KP_Q,
    PVDER_PAR_KI_Q, PVDER_PAR_WP, PVDER_PAR_KP_PLL, PVDER_PAR_KI_PLL, PVDER_PAR_M_LIMIT,
    PVDER_PAR_IREF_LIMIT, PVDER_PAR_WBASE, PVDER_PAR_VG_SCALE, PVDER_PAR_VDCBASE, PVDER_PAR_SBASE,
    PVDER_PAR_NP, PVDER_PAR_ISCR, PVDER_PAR_IRS, PVDER_PAR_PV_VT,
    PVDER_PAR_VRMS_REF,   /* Vrmsrated/Vbase (reward, PVDER_env.py:280) */
    PVDER_PAR_VDC_REF0,   /* Vdcmpp0/Vdcbase: Vdc_ref after reset */
    PVDER_PAR_SINSOL0,    /* insolation before the first solar event */
    PVDER_PAR_PPV_AT_MPP, /* model option (/root/reference cannot decide it): 0 (default) = the PV
                             array's power at the INSTANTANEOUS DC-link voltage, != 0 = at the MPP voltage Vdcmpp0 (a constant
                             per insolation level).  Moves the observations by 1e-4 ... 7e-4 (profiles/r04_oracle_uncertainty.txt) */
    PVDER_PAR_COUNT
};

/* ---- float state rows  [PVDER_FROWS(ns)][n_env] ------------------------------------------------
 * rows 0..ns-1 : ODE states in pvder order (per phase iR,iI,xR,xI,uR,uI; Vdc,xDC,xQ,xPLL,delta)
 *                where delta = wte - wbase*t is the PLL angle relative to the grid angle;
 * row  ns      : Q_ref   (PV_model.Q_ref,   PVDER_env.py:225)
 * row  ns+1    : Vdc_ref (PV_model.Vdc_ref, PVDER_env.py:229)
 * ns = 23 (three phase) or 11 (single phase). */
#define PVDER_NS(phases) (6 * (phases) + 5)
#define PVDER_FROWS(phases) (PVDER_NS(phases) + 2)

/* ---- int32 state rows  [PVDER_IROWS][n_env] */
enum pvder_irow {
    PVDER_I_KSUB = 0, /* sim time steps since reset: tStart = ksub/60 s (PVDER_env.py:63,180) */
    PVDER_I_STEPS,    /* env.steps (env_utilities.py:48-52) */
    PVDER_I_EPISODE,  /* episodes started (keys the event generator) */
    PVDER_I_STATUS,   /* 0 running, 1 time limit reached (done), 2 non-finite state (done, reward -100) */
    PVDER_IROWS
};

/* ---- output block  [PVDER_OROWS][n_env] float ---------------------------------------------------
 * rows 0..10 = the 11 observations of PVDER.state (PVDER_env.py:531-542, names :44-49),
 * row 11 = reward (PVDER_env.py:231-301), row 12 = done as 0.0/1.0.  One contiguous block so a
 * rank's whole per-step output is one RCCL all-gather message. */
#define PVDER_OBS_DIM 11
#define PVDER_O_REWARD 11
#define PVDER_O_DONE 12
#define PVDER_OROWS 13

enum pvder_goal { /* goals_list[0], PVDER_env.py:78-93, :234 */
    PVDER_GOAL_VOLTAGE = 0, /* voltage_regulation -> voltage_error */
    PVDER_GOAL_POWER = 1,   /* power_regulation   -> power_error   */
    PVDER_GOAL_Q = 2        /* Q_regulation       -> Q_error       */
};

/* ---- per-launch step configuration (host struct, passed by pointer, copied at launch) */
typedef struct pvder_step_cfg {
    int32_t n_sub;        /* n_sim_time_steps_per_env_step (PVDER_env.py:603-620) */
    int32_t micro;        /* integrator steps per 1/60 s sim time step (h = 1/(60*micro)): classic RK4
                             needs ~4096 (stiff current loop); ETDRK4: see `graded`; ignored by RK45 */
    int32_t max_ksub;     /* done when ksub >= max_ksub  (tStop >= max_sim_time, :183) */
    int32_t max_steps;    /* TimeLimit max_episode_steps (gym_PVDER/__init__.py:10); 0 = none */
    int32_t goal;         /* enum pvder_goal */
    int32_t discrete;     /* DISCRETE_REWARD (:253, :265, :278, :291) */
    int32_t ev_s0;        /* ksub of the first event  (t_events_start*60) */
    int32_t ev_ds;        /* ksub between events      (t_events_step*60)  */
    int32_t ev_count;     /* K events; event tables have K+1 rows (row 0 = nominal) */
    int32_t reserved;     /* integrator-specific test knob, 0 in production: RK45 = cap on attempts per
                             sim time step; ETDRK4 / etdlin bit 0 = run the full 23-state kernel only (bit 1 is read by
                             A/B builds with -DPVDER_RK4_FALLBACK only -- steps at the duty-cycle limit by classic RK4
                             micro-steps; the shipped library ignores it) */
    double del_qref;      /* _delQref/Sbase   per env step (:617, :225) */
    double del_vdcref;    /* _delVdcref/Vdcbase per env step (:618, :229) */
    double inv_max_sim_time; /* obs[10] = tStart/max_sim_time (:540) */
    double p_target;      /* P_ref/Sbase (:245) */
    double q_target;      /* Q_ref/Sbase (:241) */
    double rtol, atol;    /* RK45 only */
    int32_t micro_coarse; /* ETDRK4 step schedule: a sim time step that starts at an input jump (new
                             action or new event row) takes `micro` uniform steps, every other one
                             `micro_coarse` (0 = same as micro) ... */
    int32_t graded;       /* ... or, with graded != 0, the sim time step that starts at a jump takes TWO
                             steps: h = 1/(60*micro) -- the size of the FIRST step behind a jump is
                             what sets the error -- then the remainder (1 - 1/micro)/60.
                             PVDER defaults: micro 32, micro_coarse 1, graded 1 */
    int32_t extra_rewards; /* reward types of env_goal_spec[goal]['reward']['my_spec'] beyond the goal's
                             required one (:249), a mask: PVDER_REWARD_VDC_ERROR adds the Vdc_error term
                             (:251-262; power_regulation only, as its target is defined at :243-245),
                             PVDER_REWARD_Q_ERROR the Q_error term under voltage_regulation (:263-275 with
                             Qtarget = PV_model.Q_ref, :238 -- 1e-6 when it is 0 in the discrete branch, :266-267),
                             PVDER_REWARD_VOLTAGE_ERROR the voltage_error term (:276-287) under the other goals */
    int32_t off_range;    /* what happens to an env whose duty cycle |m| reaches its limit, where the reference model's component-wise
                             anti-windup switches the stiff dynamics (an env at the CURRENT limit is in range since round 3: etdlin
                             keeps it on the matrix-core path, careful pass, and never terminates it):
                             PVDER_OFF_RANGE_INTEGRATE (0, default): it is integrated through, as the reference does
                             (PVDER_env.py:166) -- by the hybrid scheme of csrc/pvder_sat_hybrid.hpp (round 6: flight / impact /
                             contact on the slow time scale, events located inside the steps, fp64 between the fp32 rows; <= 6e-3
                             per env step, median 1.5e-5, of the converged event-locating solution along the reference's
                             Vdc_ref-down ramp) in a kernel of its own, step_sat_hybrid_kernel, which FOLLOWS the step launches
                             while an env is, or within the last 256 launches was, beyond 98.75 % of the limit (a word of pinned
                             host memory the step kernels write; at most 16 step launches of such a plan are in flight, so that
                             the host sees it in time): nothing on any launch until then, ~4 us per launch while it follows,
                             milliseconds per env step for the envs it advances;
                             PVDER_OFF_RANGE_TERMINATE (1): the episode ends (status 2, done, reward -100: the reference's failure
                             branch, :169-172) -- for training runs that must not stall on a saturated converter;
                             PVDER_OFF_RANGE_HELD_MODES (2): rounds 3-5's step -- the 23-state path on the operator sets of the
                             anti-windup modes, the mode of a phase found at the START of an integrator step and held over it:
                             ~55 us per env step, 1e-2 ... 1e-1 off once the duty cycle saturates (A/B, tests);
                             PVDER_OFF_RANGE_HYBRID_ALWAYS (3): as 0 with the hybrid kernel behind EVERY step launch (+4 us each):
                             deterministic whatever the host's timing (tests).
                             fp32 exponential integrators only (etdlin, ETDRK4); three-phase model; n_sub <= 16. */
} pvder_step_cfg;
#define PVDER_REWARD_VDC_ERROR 1
#define PVDER_REWARD_Q_ERROR 2
#define PVDER_REWARD_VOLTAGE_ERROR 4
#define PVDER_OFF_RANGE_INTEGRATE 0
#define PVDER_OFF_RANGE_TERMINATE 1
#define PVDER_OFF_RANGE_HELD_MODES 2
#define PVDER_OFF_RANGE_HYBRID_ALWAYS 3

/* ---- event generator spec for reset (PVDER_env.py:60-61, :400-411) */
typedef struct pvder_event_spec {
    int32_t ev_count;        /* K */
    int32_t enable_voltage;  /* env_events_spec['voltage']['ENABLE'] */
    int32_t enable_insol;    /* env_events_spec['insolation']['ENABLE'] */
    int32_t reserved;
    float v_min, v_max;      /* U[min,max] pu */
    float s_min, s_max;      /* U[min,max] insolation */
} pvder_event_spec;

/* ---- reset arguments, also used for the in-kernel same-step auto-reset of finished envs
 * (PVDER.reset / setup_PVDER_simulation, PVDER_env.py:316-334, :366-398) */
typedef struct pvder_autoreset {
    int32_t enabled;         /* step entry points: reset envs that finish in this step (old-gym
                                VectorEnv behaviour); ignored by pvder_reset_masked_* */
    int32_t keep_events;     /* != 0: keep the event tables (explicit tables of a parity run) */
    uint64_t seed;           /* counter generator key: (seed, env_id_base + e, episode, k) */
    int64_t env_id_base;     /* global id of env 0 of this shard */
    pvder_event_spec spec;
    const double *y0;        /* host, PVDER_NS doubles: steady state in the kernels' state order
                                (last entry = delta = wte - wbase*t at t = 0) */
    const int32_t *env_ids;  /* (ABI 17) device, int32[n_env], or NULL = identity: the env that column e of every buffer carries.  The
                                envs of a batch are independent (PVDER_env.py:138-196) and a caller may ORDER them as it likes -- by
                                clock, so that the lanes of a wave run one schedule and read one event row
                                (gym_PVDER_amd.PVDERVec.sort_lanes: a desynchronised batch at the lockstep rate) --; what follows an
                                env's identity is the event generator's key (seed, env_id_base + env_ids[e], episode, k). */
    int32_t lazy_events;     /* (ABI 18) != 0: the generator's tables are written AS THE EPISODE ADVANCES.  A same-step auto-reset writes
                                row 0 and the rows the first env step reads; every env step leaves behind the rows its successor reads
                                (rows <= event_row(tStop + n_sub), carried from the tStop row the step holds anyway: one hash pair, one
                                step in four at PVDER-v0's grid).  Same rows, same values, same results bit for bit -- but the rows
                                BEYOND an env's clock are the previous episode's until their turn comes, and every step of an episode
                                must use the same n_sub.  Why: generating an episode's 38 rows at once is 4.6 us of serial hashing in
                                the step that ends an episode -- once per 160 steps in lockstep, in EVERY step of a desynchronised
                                batch (profiles/r06_desync_breakdown.txt).  pvder_reset_masked_* always write whole tables;
                                pvder_complete_event_tables does it on request.  0 (and callers of ABI <= 17): whole tables at the
                                auto-reset.  Ignored with keep_events. */
    int32_t reserved0;
} pvder_autoreset;

int pvder_version(void);
int pvder_par_count(void);
/* Threads per workgroup of the step kernels (bench.py reports it with the roofline). */
int pvder_block_size(void);

/* (ABI 18) Every row of the generator's event tables for the episode each env is in (istate row PVDER_I_EPISODE): what
 * pvder_autoreset.lazy_events leaves unwritten beyond an env's clock.  Idempotent; no-op with keep_events.
 * Replaces nothing in the reference (its create_random_events, PVDER_env.py:400-411, builds the whole list at reset()). */
int pvder_complete_event_tables(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar, const int32_t *istate,
                                float *ev_vg, float *ev_si, int n_env, void *stream);

/* Masked reset.  For every env with mask[e] != 0 (mask == NULL: all):
 *   fstate <- ar->y0, Q_ref <- 0, Vdc_ref <- par; istate: ksub = steps = status = 0, episode += 1;
 *   event tables ev_vg/ev_si [K+1][n_env] refilled from the counter generator unless
 *   ar->keep_events; out block <- reset observation (reward 0, done 0) when out != NULL. */
int pvder_reset_masked_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                           float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                           const uint8_t *mask, float *out, int n_env, void *stream);
/* ... with per-LANE parameters (pvder_plan_lane_params below): lane_folded = DEVICE float[PVDER_LPF_ROWS][n_env]; every reset env
 * starts from ITS OWN steady state and its reset observation (Ppv) uses its own PV constants.  NULL: pvder_reset_masked_f32. */
int pvder_reset_masked_lanes_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                                 float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                                 const uint8_t *mask, float *out, int n_env, const float *lane_folded, void *stream);
int pvder_reset_masked_f64(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                           double *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                           const uint8_t *mask, double *out, int n_env, void *stream);

/* One env transition for n_env envs (PVDER.step, PVDER_env.py:138-196): action_calc ->
 * integrate n_sub sim time steps -> reward_calc -> done -> state, ONE fused launch.
 *   action: [n_env] uint8 in 0..4 (values > 4 are treated as 0 and counted in stats[3]).
 *   Envs whose status != 0 on entry are not advanced (stale tuple, :145-152).
 *   ar (may be NULL): when ar->enabled, envs that finish in this step are reset in the same
 *     launch: obs rows = reset observation, reward/done rows = the transition that ended.
 *   stats (device, may be NULL, 16-byte aligned): float[4 * n_waves], n_waves = ceil(n_env / pvder_block_size())
 *     * (pvder_block_size() / 64): slot [4*w + k] holds wave w's partial of [k=0] sum of rewards, [1] #done,
 *     [2] #non-finite, [3] #invalid actions (wave-level DPP reduction, one 16-byte store per wave, no
 *     atomics, no barrier); the caller sums the slots when it wants the totals.  zero_stats != 0: this
 *     call starts the statistics of an env step (every slot is overwritten); 0: it adds to them.
 *
 * pvder_step_etdrk4_f32: fp32 fixed-step ETDRK4 (explicit exponential Runge-Kutta method of stiff
 *   order 4: Hochbruck & Ostermann 2005, five stages): the stiff filter-current/controller pair of
 *   each phase is propagated with the exact matrix exponential, the other 11 states by the
 *   scheme's non-stiff limit (a 5-stage 4th-order Runge-Kutta method); schedule: cfg->micro (32),
 *   cfg->micro_coarse (1), cfg->graded (1) -- see pvder_step_cfg.
 * pvder_step_rk4_f32: fp32 fixed-step classic RK4 micro-stepping (cfg->micro ~ 4096),
 *   Kahan-compensated state update. */
int pvder_step_etdrk4_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                          const float *coef, float *fstate, int32_t *istate, float *ev_vg,
                          float *ev_si, const uint8_t *action, float *out, float *stats,
                          int zero_stats, int n_env, void *stream);

/* ETDRK4 coefficient block: pvder_etdrk4_coef_count() floats -- the phi-function matrices of the constant
 * stiff 2x2 complex block for the step sizes of the schedule (1/(60*micro), 1/(60*micro_coarse) and
 * (1 - 1/micro)/60), followed by the SWITCHED block: the same matrices for the (i_p, u_p) pair of one phase as real
 * 4x4 operators, once per anti-windup mode of the pair (which components of du_p/dt the reference model's
 * conditional integration freezes while |m_p| > m_limit) -- a step of the 23-state path that meets the duty-cycle
 * limit is taken on the operator set of the mode found at its start.  All folded in fp64 / long double.
 * pvder_etdrk4_coefficients() computes the block on the HOST into coef_host; the caller uploads it once
 * to a device buffer and passes that buffer as `coef` to every pvder_step_etdrk4_f32() call with the
 * same par / micro / micro_coarse / graded (kept out of the kernel-argument segment, which every workgroup
 * re-reads at launch). */
/* One env's output words and counters on the HOST -- what the reference's step() hands back as Python objects (PVDER_env.py:196):
 * out[r * n_env + env] (r = 0..12, float or double) -> host_out[13], istate[k * n_env + env] (k = 0..3) -> host_ist[4], two
 * asynchronous copies behind whatever `stream` holds, then ONE wait for the stream.  host_* should be pinned memory. */
int pvder_fetch_env(const void *out, const int32_t *istate, int n_env, int env, int fp64, void *host_out, int32_t *host_ist,
                    void *stream);
int pvder_etdrk4_coef_count(void);
int pvder_etdrk4_coefficients(const double *par, const pvder_step_cfg *cfg, float *coef_host);
int pvder_step_rk4_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                       float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                       const uint8_t *action, float *out, float *stats, int zero_stats, int n_env,
                       void *stream);

/* Same transition, fp64 adaptive Dormand-Prince RK45 (cfg->rtol/atol), per-env step size kept
 * in hstep [n_env] doubles (in/out; 0 = pick an initial step). */
int pvder_step_rk45_f64(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                        double *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                        const uint8_t *action, double *out, double *hstep, float *stats,
                        int zero_stats, int n_env, void *stream);

/* ---- etdlin: linearised exponential integrator (the default of the Python host) -------------------
 * Same transition, fp32, ETD3RK (Cox & Matthews 2002) on dy/dt = L y + N(y) with L the 11x11
 * Jacobian of the balanced model at its nominal operating point -- the device-side counterpart of
 * the Jacobian-driven LSODA integration the reference runs (PVDER_env.py:384-386, jacFlag).  The
 * scheme is evaluated in f-form, U_i = y + sum_j M_ij f(U_j): six dense 11x11 matrices per step
 * size ("coefficient set", PVDER_LIN_SET_FLOATS floats: M21, M31, M32, M41, M42, M43, each
 * PVDER_LIN_MAT_FLOATS = 12 x 12 floats, row-major with the 11 rows / columns padded to 12 -- the
 * kernels read them as MFMA A operands, 16 bytes at a time),
 * computed by the caller in fp64 -- the Python host does it in gym_PVDER_amd/etd_linear.py
 * (phi-functions of hL by an augmented-matrix exponential) -- and uploaded once to a device buffer.
 *   sets         : [tick matrix (PVDER_LIN_MAT_FLOATS)] [set 1] ... [set n_sets-1].  The tick matrix is
 *                  h phi1(hL) for h = len_ticks[0] ticks, one tick = 1/(60 * cfg->micro) s: the first
 *                  step behind an input jump (new action / new event row) is an exponential-Euler
 *                  step  y+ = y + h phi1(hL) f(y)  of that length (1 or 2 ticks);
 *   set s >= 1   : ETD3RK over len_ticks[s] ticks;
 *   sched        : device, n_sched_rows rows of 8 bytes.  A row lists, one set number per byte and terminated
 *                  by 0, the ETD3RK steps an env takes behind the tick to cover a SEGMENT (the stretch up to the
 *                  next event row change or the end of the env step); the tick and the listed lengths must add
 *                  up to the segment.  Rows 0 .. n_sub-1: row L-1 = a segment of L sim time steps.  The Python
 *                  host lists, for the segment lengths PVDER-v0 meets, two steps of half of what the tick leaves
 *                  (an env step = tick + two equal ETD3RK steps with ONE coefficient set) and tiles any other
 *                  length with whole multiples of 1/60 s.  Rows n_sub .. 2 n_sub-1: the same for the CAREFUL
 *                  pass (below; the host lists eight equal steps where the regular row lists two halves).
 *                  A byte with bit 7 set ends a row that needs more than seven steps: the list continues in
 *                  row 2 n_sub + (byte & 0x7f);
 *   est_guard    : an env whose embedded error indicator  max |y+ - b|  over (i_a, Vdc) exceeds it in
 *                  any ETD3RK step -- hopelessly far from the operating point L was taken at -- is
 *                  advanced by the 23-state ETDRK4 path instead (`coef`, as for
 *                  pvder_step_etdrk4_f32), like the envs that come near the duty-cycle limit or are not balanced.
 * The careful pass: an env whose current reference comes within 5 % of iref_limit (a policy that holds one
 * Q_ref action walks there, PVDER_env.py:214-217, :480-487) stays on this path; its env step is taken by hybrid
 * ETD3RK steps that carry the reference model's conditional integration of the outer PI loops -- free, frozen,
 * or sliding along |iref| = iref_limit, with the crossing located inside the step (csrc: lin_integrate_careful). */
#define PVDER_LIN_MAT_FLOATS 144
#define PVDER_LIN_SET_FLOATS 864
#define PVDER_LIN_MAX_SETS 12
typedef struct pvder_lin_coef {
    const float *sets;       /* device, PVDER_LIN_MAT_FLOATS + (n_sets - 1) * PVDER_LIN_SET_FLOATS floats */
    int32_t n_sets;          /* 2 .. PVDER_LIN_MAX_SETS */
    int32_t len_ticks[PVDER_LIN_MAX_SETS];
    const uint8_t *sched;    /* device, 8 * n_sched_rows bytes */
    float est_guard;
    int32_t n_sched_rows;    /* >= 2 * cfg->n_sub (regular rows, careful rows, continuation rows) */
    float seq_prop[18];      /* exp(A0 n_sub / 60): complex 3x3, row-major (re, im) -- what an env step leaves of a negative- /
                              * zero-sequence component (i, x, u) of an env slightly off balance, which the kernel carries on
                              * the matrix-core path by its positive-sequence component (etd_linear.py::seq_propagator);
                              * all zero: such components are dropped at the end of the step -- do not */
    int32_t k4;              /* (ABI 14) != 0: 10 * PVDER_LIN_MAT_FLOATS more floats follow the sets in `sets` -- the f-form matrices
                              * M21 | M31 M32 | M41 M42 M43 | B1 B2 B3 B4 of ONE Krogstad ETDRK4 step of n_sub / 60 s
                              * (etd_linear.py::k4_f_matrices).  An env step that is one segment and does not start at an event row
                              * change (a new action at most: no input jump the stiff loop rings from) is then taken as that one
                              * step, without tick: 4 right-hand sides instead of 7 at unchanged accuracy (tools/scheme_proto.py).
                              * 0: every segment by tick + ETD3RK steps (the schedule table).
                              * Bit 1 (ABI 16, with bit 0: k4 & 3 = 3): TWO more blocks of the same kind follow -- Krogstad steps of
                              * (n_sub / 60) / K s and of (n_sub / 60 - len_ticks[0] ticks) / K, K = bits 8-15 of k4 (0: K = 2; at
                              * most 16) -- for pvder_plan_lane_params on an etdlin plan, which takes K such steps per env step
                              * (etd_linear.py::pack_block(lane_k4=K)) */
} pvder_lin_coef;
int pvder_lin_set_floats(void);
int pvder_lin_max_sets(void);
int pvder_step_etdlin_f32(const double *par, const pvder_step_cfg *cfg, const pvder_autoreset *ar,
                          const float *coef, const pvder_lin_coef *lin, float *fstate, int32_t *istate,
                          float *ev_vg, float *ev_si, const uint8_t *action, float *out, float *stats,
                          int zero_stats, int n_env, void *stream);

/* ---- plans: fold once, step many times ----------------------------------------------------------
 * The per-call entry points above re-derive the kernel arguments (parameter folding, reset block,
 * phi-function scalars) on every call -- a few microseconds of host time, which is as long as the
 * etdlin kernel runs.  A plan holds them; it references `coef` / `lin->sets` (device buffers owned
 * by the caller, which must outlive the plan) and copies everything else.  Rebuild the plan when
 * par / cfg / ar change.  pvder_plan_step == the matching pvder_step_* call with zero_stats = 1;
 * hstep is only read for PVDER_INT_RK45 (fstate/out are then double).
 * pvder_plan_rollout: n_steps transitions from ONE host call -- the same per-step launch, enqueued
 * n_steps times, with action row k = actions + k*n_env and output block k = out + k*out_step_stride
 * elements (0: every step overwrites the same block).  A finished env (status != 0 on entry, no
 * auto-reset) returns the stale tuple: the step kernels leave its rows alone, and block k > 0 receives
 * them from block k-1; block 0 must hold them on entry (copy the block of the preceding step there). */
/* PVDER_INT_ETDLIN64 (plans only): the tolerance-match path of BASELINE configs[4] that is not stability-bound -- the
 * etdlin scheme in fp64 on the balanced reduction with step doubling on a dyadic menu as its error control (cfg->rtol /
 * cfg->atol in the RK45 kernel's norm), the adaptive RK45 kernel behind it for the envs the reduction does not cover
 * (fstate / out double, hstep as for RK45).  `lin->sets`: device DOUBLES, (1 + levels) sets of 10 x [11 columns][12] f-form
 * matrices of a Krogstad ETDRK4 step (KM21 ... KB4, column by column; ABI 15 -- ABI 14: six ETD3RK matrices) -- set 0 for the tick, set 1 + j for (n_sub / 60 - tick) / 2^(j+1) s (gym_PVDER_amd/etd_linear.py::
 * pack_block64); lin->n_sets = 1 + levels (3 .. 10); sched / len_ticks / est_guard unused.  Three-phase model, events on
 * env-step boundaries (PVDER_EINVAL otherwise). */
enum pvder_integrator { PVDER_INT_ETDLIN = 0, PVDER_INT_ETDRK4 = 1, PVDER_INT_RK4 = 2, PVDER_INT_RK45 = 3, PVDER_INT_ETDLIN64 = 4 };
typedef struct pvder_plan pvder_plan;
int pvder_plan_create(pvder_plan **plan, int integrator, const double *par, const pvder_step_cfg *cfg,
                      const pvder_autoreset *ar, const float *coef, const pvder_lin_coef *lin, int n_env);
void pvder_plan_destroy(pvder_plan *plan);
/* Where the envs of the batch ARE (etdlin plans): `ksub` = the sim-time-step counter (row PVDER_I_KSUB) most envs hold when
 * the NEXT step starts -- 0 behind a reset of the whole batch (PVDER.reset, PVDER_env.py:316-334), -1 = unknown (the state of
 * a new plan), -2 = query only.  Returns the hint in force.  The plan advances it by itself with every step it enqueues
 * (n_sub per step, back to 0 where a batch with same-step auto-reset ends its episode).  The step launch requests the event
 * rows that clock names together with the state rows instead of behind them (one memory round trip less before the first
 * flop); an env whose own counter says otherwise takes the dependent fetch as before.  A prefetch hint: a wrong or missing
 * one costs time, never a result. */
int pvder_plan_hint(pvder_plan *plan, int ksub);
/* (ABI 17) The clock hint is a kernel ARGUMENT: a launch captured into a HIP graph replays the hint it was captured with.  By
 * default every captured launch -- pvder_plan_step and pvder_plan_rollout alike -- is therefore taken with the hint cleared and
 * leaves the plan's clock alone (a replay at another clock with a stale hint is slower than no hint; results never depend on it).
 * on = 1: a captured pvder_plan_rollout keeps and advances its hints -- for a caller that replays the graph exactly where it was
 * captured (bench.py's timed window).  on < 0 = query only.  Returns the previous setting, PVDER_EINVAL without a plan. */
int pvder_plan_graph_hints(pvder_plan *plan, int on);
/* (ABI 17) The step kernel a launch of this plan runs, as rocprofv3 prints it without its namespace and argument list (e.g.
 * "step_etdlin_kernel_w1<3>": the one-wave-per-SIMD form a batch of up to 65 536 three-phase envs takes): what a measurement
 * names as "the kernel that ran".  Static storage; "" without a plan. */
const char *pvder_plan_kernel_name(const pvder_plan *plan);
/* (ABI 17) 1 while step_sat_hybrid_kernel follows this plan's step launches -- off_range PVDER_OFF_RANGE_HYBRID_ALWAYS, or
 * PVDER_OFF_RANGE_INTEGRATE while an env is, or lately was, beyond 98.75 % of the duty-cycle limit (see pvder_step_cfg.off_range) --,
 * else 0.  The fused rollouts (pvder_plan_rollout_fused*) step such envs on the held operator sets: a caller that wants the
 * hybrid step takes pvder_plan_rollout while this returns 1 (gym_PVDER_amd.PVDERVec.rollout does). */
int pvder_plan_hybrid_active(const pvder_plan *plan);

/* Per-LANE parameter override (PVDER_INT_ETDRK4, three-phase model): every env of the batch with its own values of the
 * parameters that stay outside the exponential operator of the 23-state ETDRK4 path -- domain randomisation over the
 * DC-link capacitance, the outer PI loops, the PV array size and the insolation scale (config_der.json:2-21: C, Kp_DC,
 * Ki_DC, Kp_Q, Ki_Q, Np; SURVEY.md a10 "optional per-env override [P][N]").  Everything inside the operator (Rf, Lf,
 * Kp_GCC, Ki_GCC, wp, the ratings and limits) stays per parameter GROUP: one plan per set (param_groups.PVDERVecGroups).
 *   lane rows, in the units of par[] (PVDER_PAR_C ... are per-unit values): NaN = keep par[]'s value for that env. */
enum {
    PVDER_LP_C = 0, PVDER_LP_KP_DC, PVDER_LP_KI_DC, PVDER_LP_KP_Q, PVDER_LP_KI_Q, PVDER_LP_NP,
    PVDER_LP_SINSOL_SCALE, /* factor on the insolation the env sees (event table and reset value alike); NaN = 1 */
    PVDER_LP_ROWS
};
/*   folded rows (what the kernel reads; float32): 1/C, the four gains, and the two constants of the PV array model
 *   pv1 = pvA * Sinsol + pvB (pvder_model.hpp ModelK; pvA carries Np and the insolation scale, pvB Np). */
/*   ... and (ABI 15) the env's OWN steady state -- the reference builds every DER with steadyStateInitialization=True
 *   (PVDER_env.py:374-378), so an env with its own PV array starts at ITS operating point: phase a's current i_a and duty cycle
 *   m_a (Newton root of  Re S = Ppv(Vdc_ref0; Np, scale),  Im S_PCC = 0,  di/dt = 0, solved per env in fp64 by the fold);
 *   the other entries of y0 follow from them (i_b,c = i_a e^{-/+j120}, x_p = m_p, u_p = 0, xDC + j xQ = i_a) or do not depend
 *   on the lane's parameters (Vdc = Vdc_ref0, xPLL = 0, delta). */
enum { PVDER_LPF_IC = 0, PVDER_LPF_KP_DC, PVDER_LPF_KI_DC, PVDER_LPF_KP_Q, PVDER_LPF_KI_Q, PVDER_LPF_PVA, PVDER_LPF_PVB,
       PVDER_LPF_IA_RE, PVDER_LPF_IA_IM, PVDER_LPF_MA_RE, PVDER_LPF_MA_IM, PVDER_LPF_ROWS };
/* pvder_fold_lane_params (host, no GPU): lane[PVDER_LP_ROWS][n_env] doubles -> folded[PVDER_LPF_ROWS][n_env] floats, by
 * the SAME folding as the plan's own constants (model option PVDER_PAR_PPV_AT_MPP included).  PVDER_EINVAL for a
 * non-positive C, a negative Np or scale, or a non-finite value. */
int pvder_fold_lane_params(const double *par, const double *lane, int n_env, float *folded);
/* pvder_plan_lane_params: lane_folded = DEVICE pointer to float[PVDER_LPF_ROWS][n_env] (kept by the plan, not copied;
 * NULL returns the plan to its own constants).  While set, pvder_plan_step / pvder_plan_rollout advance EVERY env on the
 * 23-state path (no balanced reduction: an env with its own PV array is not at the group's operating point) -- batches up to
 * 32 768 envs with an env's three phases on three lanes (step_etdrk4_lane3_kernel: 45-65 us per batched step), larger ones
 * with a lane per env (step_etdrk4_lane_kernel: 72 us at 65 536 envs); same scheme, results within fp32 rounding of each other.  The same-step auto-reset restarts an env from ITS OWN steady state (rows PVDER_LPF_IA_RE ..), and so does
 * pvder_reset_masked_lanes_f32; the observation's Ppv and the reward use the env's own constants.
 * PVDER_EINVAL unless the plan is a three-phase PVDER_INT_ETDRK4 plan -- or (ABI 16) a three-phase PVDER_INT_ETDLIN plan whose
 * coefficient block carries the two lane blocks (pvder_lin_coef.k4 = 3): the batch then stays on the matrix-core path, the
 * exponential operator the GROUP's (one shared coefficient block), each env's own constants in the right-hand side alone and two
 * Krogstad steps per env step (step_etdlin_lane_kernel: 6e-5 of the converged per-set bridge on the 64-set golden in fp64,
 * tools/lane_etdlin_proto.py; ~12 us per batched step at 65 536 distinct sets); an env off balance, near a limit or with an event row
 * change inside its env step takes the 23-state lane path in the same launch.  Fused rollouts refuse such a plan. */
int pvder_plan_lane_params(pvder_plan *plan, const float *lane_folded);
int pvder_plan_step(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                    const uint8_t *action, void *out, double *hstep, float *stats, void *stream);
/* The single env's whole step from ONE host call (the drop-in's `PVDER.step`, PVDER_env.py:138-196): pvder_plan_step, then --
 * snapshot != NULL -- one device-to-device copy of snapshot_bytes from snapshot_src (the trajectory record the reference keeps
 * of every solution row, :389), then pvder_fetch_env of env 0 (n_env = 1).  Three Python-level calls of ~10 us each become one. */
int pvder_plan_step_fetch(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                          const uint8_t *action, void *out, double *hstep, float *stats, void *snapshot, const void *snapshot_src,
                          size_t snapshot_bytes, void *host_out, int32_t *host_ist, void *stream);
/* ... and without a copy packet or a stream wait (ABI 16): behind the step launch ONE small kernel writes the env's output block
 * and counters -- and, snapshot != NULL, the trajectory record -- and then a sequence number into `host_block`, PVDER_MIRROR_WORDS
 * doubles of PINNED host memory the device can address (hipHostMalloc / torch pin_memory: fine-grained, written over the bus as the
 * kernel stores); the host spins on the sequence word.  host_block[0..12] = the 13 output rows (fp32 rows widened: exact),
 * [13..16] = istate rows ksub / steps / done / status, [17] = seq as written LAST (system-scope release).  seq must differ from the
 * block's current [17] (callers count up).  The wait is bounded (PVDER_MIRROR_TIMEOUT_MS, then hipStreamSynchronize decides):
 * PVDER_EINVAL when host_block is not device-addressable or n_env != 1, PVDER_ETIMEOUT when the word never arrives. */
#define PVDER_MIRROR_WORDS 18
#define PVDER_MIRROR_TIMEOUT_MS 2000
#define PVDER_ETIMEOUT (-10003)
int pvder_plan_step_mirror(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                           const uint8_t *action, void *out, double *hstep, float *stats, void *snapshot, const void *snapshot_src,
                           size_t snapshot_bytes, double *host_block, uint64_t seq, void *stream);
int pvder_plan_rollout(const pvder_plan *plan, void *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                       const uint8_t *actions, int n_steps, void *out, size_t out_step_stride, double *hstep,
                       float *stats, void *stream);
/* pvder_plan_rollout_fused: the same open-loop rollout (calc_returns' fixed policies, PVDER_env.py:458-497;
 * recorded or synthetic action sequences) with the env steps FUSED into one launch -- PVDER_INT_ETDLIN,
 * three-phase model only (PVDER_EINVAL otherwise).  The 11-state reduction of a balanced env stays in
 * registers from the first step to the last: the state rows are read once and written once, and per step
 * only the 13 output rows are stored.  Results are bit-identical to pvder_plan_rollout: the steps a
 * workgroup cannot carry in registers (the one that ends an episode -- done / same-step auto-reset --, an
 * env that is unbalanced, near a limit, not running or non-finite) are taken by the per-step body, through
 * memory, by a follow-up launch; 2 * (n_steps / episode length + 2) launches per call.
 *   progress : device scratch, int32[ceil(n_env / pvder_block_size())] (contents need not be initialised).
 *   stats    : as pvder_plan_step; on return the slots hold the statistics of the LAST step. */
int pvder_plan_rollout_fused(const pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                             const uint8_t *actions, int n_steps, float *out, size_t out_step_stride, float *stats,
                             int32_t *progress, void *stream);

/* (ABI 18) SERVED rollout: pvder_plan_rollout_fused whose env steps wait for their action rows -- the register-resident state of the
 * fused loop for a producer that does NOT know its actions ahead (a policy a few steps behind the env, a recorded stream read as
 * it arrives).  Same restrictions and launches as pvder_plan_rollout_fused, same results bit for bit (PVDER_env.py:138-196 has no
 * notion of WHEN an action arrives).  Protocol, on words pvder_served_alloc returns (host[i] and dev[i] are the same memory):
 *   host[0]  (caller writes)  p: action rows 0 .. p-1 of `actions` are in device memory, visible to the device.  Monotonic.
 *   host[16], host[17]  (the launch writes)  min / max over the workgroups of d = "the outputs of steps < d have left the CU"
 *                        (a workgroup reports step k while it takes step k+1: a producer can be at most ONE step behind the env).
 * One thread of the launch (a relay workgroup) polls host[0] over PCIe and fans it out on the device; every wait is BOUNDED
 * (0.2 s): a caller that stops posting gets its launches back, `progress` short.
 * paced = 0: the caller has posted EVERY row already (host[0] = n_steps before the call): one call enqueues the whole rollout, as
 * pvder_plan_rollout_fused does (its last finish pass does not wait for rows).
 * paced = 1: rows are posted against the reports.  ONE call = ONE pair of launches: the served loop until a workgroup meets a step
 * it cannot carry (an episode end, an env off the standard path: that workgroup asks EVERY workgroup to leave, so that the launch
 * ends), then the finish kernel, which takes every workgroup that has steps left through one step (whose action row the served loop
 * had).  The caller reads `progress` (device, int32 per workgroup of 256 envs: steps behind it) when the pair is complete and calls
 * again with resume = 1 until every entry is n_steps -- posting meanwhile; PVDERVec.rollout_served is that loop.  A batch with an
 * env that stays off the standard path for many steps pays a pair (and a host round trip) per step, like a fused rollout does.
 * resume = 0 starts a rollout (clears `scratch`, completes lazy event tables; PVDER_EINVAL while the duty-cycle-limit kernel follows
 * the plan's launches: pvder_plan_hybrid_active).
 * The served loop writes its output rows THROUGH the L2 (agent scope) before it reports them: a kernel that starts on another
 * stream once host[16] says so reads them whole.
 * MI355X, 65 536 envs: 4.9 us per env step with the rows posted ahead (one launch per step 8.1, open-loop fused 4.8), 5.5 us with
 * row k+1 posted when step k-8 is reported, 27 us when step k-1 is (the PCIe round trip of a host in the loop):
 * profiles/r06_ab_served_rollout.txt.  `scratch`: device words of the protocol. */
int pvder_served_alloc(int n_env, int **host, int **dev, int **scratch);
int pvder_served_free(int *host, int *scratch);
int pvder_plan_rollout_served(const pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                              const uint8_t *actions, int n_steps, float *out, size_t out_step_stride, float *stats,
                              int32_t *progress, int *words_dev, int *scratch_dev, int paced, int resume, void *stream);

/* ---- the per-step collective of the sharded env (multi-GPU) -------------------------
 * The reference has no distributed code; the build shards independent envs over ranks and collates
 * the [13, n_env] output block of every step with ONE all-gather (RCCL over xGMI).  These entry points
 * enqueue step kernel + gather from C -- one host call per rollout instead of six Python-level calls
 * per step: the kernel on the caller's stream, the gather on the object's own stream, two output
 * blocks alternating (step k uses out[(first_block + k) & 1] / recv[...]; the stale tuples of finished
 * envs are carried from one block to the other) so that the gather of step t overlaps the kernel of
 * step t+1.  RCCL is not linked: `nccl_all_gather` is the address of
 * ncclAllGather in the librccl the caller uses, `comm` its ncclComm_t.  recv blocks are
 * [world * 13, n_env] floats.  pvder_gather_drain makes `stream` wait for the gathers in flight. */
typedef struct pvder_gather pvder_gather;
int pvder_gather_create(pvder_gather **g, void *nccl_all_gather, void *comm);
void pvder_gather_destroy(pvder_gather *g);
int pvder_gather_drain(pvder_gather *g, void *stream);
int pvder_plan_rollout_gather(const pvder_plan *plan, pvder_gather *g, void *fstate, int32_t *istate, float *ev_vg,
                              float *ev_si, const uint8_t *actions, int n_steps, float *out0, float *out1, float *recv0,
                              float *recv1, float *stats, void *stream, int first_block, int sync_each);
/* The same with a choice of PAYLOAD: a step collates rows [row0, row0 + n_rows) of its output block -- (reward, done)
 * = rows 11, 12: 0.5 MB per rank and step at 65 536 envs instead of 3.4 MB, the xGMI link time of a step drops from
 * 44 us to 7 us -- except every `full_every`-th step (counted from step_index0), which collates the whole block
 * (full_every <= 1: never).  A partial gather fills the first world * n_rows rows of its recv block.  The consumer
 * decides what it needs every step: a policy evaluated where its env lives needs the peers' rewards and dones (for
 * the return statistics / the episode bookkeeping), not their observations. */
/* row0 = PVDER_GATHER_OBS8 (n_rows ignored): the "obs8" payload -- rows 0-7 (the eight DYNAMIC observations) and 11-12 (reward,
 * done), packed side by side by a small kernel on the gather's stream and collated as [world * 10, n_env]: rows 8-10 (Vdc_ref,
 * Q_ref, tStart / max_sim_time) follow from the actions and the step count (PVDER_env.py:198-229, :540), which a consumer that
 * chose the actions holds already -- 10 of 13 rows, 23 % less on the links than "full".  The staging blocks (2 x 10 x n_env
 * floats) are allocated at the first such call (not under stream capture). */
#define PVDER_GATHER_OBS8 (-8)
#define PVDER_GATHER_OBS8_ROWS 10
int pvder_plan_rollout_gather_rows(const pvder_plan *plan, pvder_gather *g, void *fstate, int32_t *istate, float *ev_vg,
                                   float *ev_si, const uint8_t *actions, int n_steps, float *out0, float *out1, float *recv0,
                                   float *recv1, float *stats, void *stream, int first_block, int sync_each, int row0,
                                   int n_rows, int full_every, long long step_index0);

#ifdef __cplusplus
}
#endif
#endif /* PVDER_HIP_H */
