/* pvder_agent.h -- C ABI of the device-resident tabular Q-learning step.
 *
 * Replaces the body of the reference's training loop, /root/reference/gym_PVDER/agents/qagent.py:
 * discretise the observation (get_discrete_state, :43-45), update the table with the transition that just
 * ended (:83-103), choose the next action epsilon-greedily (:63-68) -- for n_env environments learning into
 * ONE table in lockstep, in ONE launch per env step next to the step launch of pvder_hip.h; observations,
 * actions, rewards and the table never leave the GPU and the host never synchronises inside an episode.
 * Same conventions as pvder_hip.h: plain pointers, a hipStream_t, asynchronous, allocation-free;
 * 0 / PVDER_EINVAL (-10001) / -hipError_t. */
#ifndef PVDER_AGENT_H
#define PVDER_AGENT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvder_qagent_cfg {
    int32_t bins;       /* bins per observation dimension (reference: 6, DISCRETE_OS_SIZE :35); bins^11 table rows */
    int32_t n_act;      /* actions per row (5, PVDER_env.py:50) */
    int32_t first;      /* 1: first call of an episode -- discretise the reset observation, no table update */
    int32_t choose;     /* 1: write the next epsilon-greedy action into `action`; 0: leave it to the caller */
    float lr;           /* LEARNING_RATE (:14) */
    float gamma;        /* DISCOUNT (:15) */
    float epsilon;      /* exploration probability of this call (:27, decayed per episode by the caller) */
    float q_done;       /* value the entry of the transition that ENDS an episode gets (0, :100-103) */
    double low;         /* observation_space.low (-10, PVDER_env.py:51) */
    double win;         /* discrete_os_win_size = (high - low) / bins (:38) */
    uint64_t seed;      /* exploration stream: counter_uniform(seed, global env id, counter, 0, lane) -- the */
    uint64_t counter;   /*   generator of the event tables; lane 0: explore?, lane 1: which action */
    int64_t env_id_base;
    int32_t policy;     /* (ABI 15) PVDER_POLICY_TABLE = 0: the tabular agent above; PVDER_POLICY_QNET = 1: a Q-NETWORK as the action
                           source -- the QNetwork of the reference's DQN demo (examples/gym_PVDER_environment_tf_agents_DQN_demo.ipynb
                           cell 17: observation(11) -> Dense(hidden, relu) -> Dense(n_act)), INFERENCE only: epsilon-greedy on its
                           outputs, no update on the device (the trainer owns the weights: gym_PVDER_amd/agents/dqn.py); bins / lr /
                           gamma / q_done / low / win unused, `state` left alone, `alive` / `ep_reward` kept as above */
    int32_t hidden;     /* PVDER_POLICY_QNET: width of the hidden layer (100: fc_layer_params = (100,), cell 12); 1 .. 128 */
} pvder_qagent_cfg;
enum { PVDER_POLICY_TABLE = 0, PVDER_POLICY_QNET = 1 };
/* PVDER_POLICY_QNET: `q_table` = device float[hidden * 11 + hidden + n_act * hidden + n_act]: W1 [hidden][11] row-major, b1 [hidden],
 * W2 [n_act][hidden] row-major, b2 [n_act] (torch.nn.Linear layouts); n_act <= 8.  Every lane evaluates the SAME network on its
 * own observation: a launch stages the weights in LDS once per workgroup (broadcast reads), the products run on the vector ALUs
 * in the order k = 0..10 / j = 0..hidden-1 (fused multiply-adds): within fp32 rounding of a torch forward pass, the same argmax
 * except on ties. */

/* One agent step for n_env envs.
 *   q_table : device float[bins^11 * n_act], row-major (row = flat index of the 11 bin numbers, first
 *             observation component most significant -- numpy's ravel_multi_index of the reference's tuple)
 *   out     : the env's output block [13][n_env] of the step that just ran (rows 0..10 observation, 11 reward,
 *             12 done), or of the reset (cfg->first)
 *   state   : int64[n_env] in/out -- table row of the observation the pending action was chosen in
 *   action  : uint8[n_env] in: the action that led to `out`; out (cfg->choose): the next one
 *   alive   : uint8[n_env] in/out -- 1 while the env's episode runs (a finished env returns stale tuples,
 *             PVDER_env.py:145-152, and must not learn from them); cfg->first sets it to 1
 *   ep_reward : double[n_env] in/out -- return of the running episode (cfg->first zeroes it)
 * Update rule, per env with alive != 0 (qagent.py:83-103):
 *   done ? Q[s][a] = q_done : Q[s][a] = (1 - lr) Q[s][a] + lr (r + gamma max_a' Q[s'][a'])
 * When several envs hit the same entry in one call, one of their updates wins (plain stores). */
int pvder_qagent_step_f32(const pvder_qagent_cfg *cfg, float *q_table, const float *out, int64_t *state,
                          uint8_t *action, uint8_t *alive, double *ep_reward, int n_env, void *stream);

/* The closed loop without the host in it: n_steps x (env step, agent step) of a training episode -- qagent.py:56-105 --
 * in a handful of launches, the env state and the agent's per-env registers on the chip from step to step (the fused
 * rollout of pvder_hip.h, pvder_plan_rollout_fused, with the agent step between two env steps).  `plan`: a
 * PVDER_INT_ETDLIN three-phase plan WITHOUT auto-reset (struct pvder_plan of pvder_hip.h); fstate .. progress as for
 * pvder_plan_rollout_fused, `out` ONE [13][n_env] block (every step overwrites it: the agent consumes it on the chip).
 * `action` holds, on entry, the action chosen behind the reset (a pvder_qagent_step_f32 call with cfg->first = 1);
 * on return state / action / alive / ep_reward are what n_steps two-launch steps would have left.  cfg: first = 0,
 * choose = 1; cfg->counter keys the agent call behind env step 0, + k behind step k.
 * CONCURRENCY: this is ASYNCHRONOUS (Hogwild-style) Q-learning across workgroups.  The two-launch loop puts a kernel
 * boundary behind every agent step: what env A writes to a table entry at step k, env B reads at step k + 1.  Inside a fused
 * launch only the waves of ONE workgroup advance in step (workgroup votes); workgroups run through their env steps
 * independently, so an agent step k of one workgroup interleaves with steps k +- m of another on the shared table, the table
 * is read with plain loads through per-CU caches that are not coherent within a launch (another workgroup's update may be
 * seen late -- at the latest in the next launch, i.e. after at most n_steps steps -- or lost to a read-modify-write race),
 * and which of several simultaneous updates of one entry wins differs from the two-launch order.  Results therefore equal
 * the two-launch sequence bit for bit only where no two workgroups touch the same entry (always: a batch of <= one
 * workgroup with the envs of a wave in lockstep -- tests at 1, 8, 64 envs); for larger batches they agree STATISTICALLY
 * (returns, visited entries, values of often-visited entries: tests/test_gpu_golden.py::
 * test_closed_loop_fused_training_statistics_at_many_workgroups).  Callers that need the two-launch semantics exactly use
 * pvder_plan_step + pvder_qagent_step_f32. */
struct pvder_plan;
int pvder_plan_rollout_fused_agent(const struct pvder_plan *plan, float *fstate, int32_t *istate, float *ev_vg, float *ev_si,
                                   int n_steps, float *out, float *stats, int32_t *progress, const pvder_qagent_cfg *cfg,
                                   float *q_table, int64_t *state, uint8_t *action, uint8_t *alive, double *ep_reward,
                                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PVDER_AGENT_H */
