// pvder_agent.hip -- device-resident tabular Q-learning step (include/pvder_agent.h): the body of the
// reference's QAgent.train loop (gym_PVDER/agents/qagent.py:56-105) for n_env envs in ONE launch per env
// step.  HBM-bound gather / scatter on a bins^11 x 5 fp32 table (6 bins: 7.3 GB): per env 13 coalesced
// reads of the output block, two 20-byte row gathers, one 4-byte scatter, a handful of integer ops.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pvder_agent.h"

namespace {

constexpr int kAgentBlock = 256;
constexpr int kEinval = -10001;            // PVDER_EINVAL (pvder_hip.h)
constexpr int kObs = 11, kRewardRow = 11, kDoneRow = 12;

// counter-based generator of pvder_kernels.hip / oracle/pvder_model.py::counter_uniform (bit-identical)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float counter_uniform(uint64_t seed, uint64_t env, uint64_t episode, uint32_t k, uint32_t lane) {
    uint64_t key = splitmix64(seed);
    key = splitmix64(key ^ (env & 0xFFFFFFFFull) ^ ((episode & 0xFFFFFFFFull) << 32));
    key = splitmix64(key ^ ((uint64_t)k << 8) ^ (uint64_t)(lane & 0xFF));
    return (float)(key >> 40) * (1.0f / 16777216.0f);
}

struct AgentK {
    int bins, n_act, first, choose;
    float one_minus_lr, lr, gamma, epsilon, q_done;
    double low, win;
    uint64_t seed, counter;
    long long env_id_base;
};

__global__ __launch_bounds__(kAgentBlock) void qagent_step_kernel(const AgentK K, float *__restrict__ q,
                                                                   const float *__restrict__ out, long long *__restrict__ state,
                                                                   uint8_t *__restrict__ action, uint8_t *__restrict__ alive,
                                                                   double *__restrict__ ep_reward, int n_env) {
    const int e = blockIdx.x * kAgentBlock + threadIdx.x;
    if (e >= n_env) return;
    const size_t N = (size_t)n_env;
    // get_discrete_state (qagent.py:43-45): (state - low) / win in float64, truncated; the Box bound itself
    // (+10) falls into the last bin
    long long s2 = 0;
#pragma unroll
    for (int k = 0; k < kObs; ++k) {
        long long d = (long long)(((double)out[(size_t)k * N + e] - K.low) / K.win);
        d = d < 0 ? 0 : (d > K.bins - 1 ? K.bins - 1 : d);
        s2 = s2 * K.bins + d;
    }
    const float *row2 = q + (size_t)s2 * K.n_act;
    if (K.first) {
        alive[e] = 1;
        ep_reward[e] = 0.0;
    } else if (alive[e]) {
        const float r = out[(size_t)kRewardRow * N + e];
        const bool done = out[(size_t)kDoneRow * N + e] != 0.f;
        float *entry = q + (size_t)state[e] * K.n_act + action[e];
        float nq = K.q_done;                                           // :100-103
        if (!done) {                                                   // :83-95
            float mx = row2[0];
            for (int j = 1; j < K.n_act; ++j) mx = fmaxf(mx, row2[j]);
            nq = K.one_minus_lr * *entry + K.lr * (r + K.gamma * mx);
        }
        *entry = nq;
        ep_reward[e] += (double)r;
        alive[e] = done ? 0 : 1;
    }
    state[e] = s2;
    if (K.choose) {                                                    // :63-68
        int best = 0;
        float bq = row2[0];
        for (int j = 1; j < K.n_act; ++j) {                            // np.argmax: the first maximum
            const float v = row2[j];
            if (v > bq) { bq = v; best = j; }
        }
        const uint64_t gid = (uint64_t)(K.env_id_base + e);
        const float u = counter_uniform(K.seed, gid, K.counter, 0, 0);
        if (u <= K.epsilon) {                                          // np.random.random() > epsilon -> greedy
            int a = (int)(counter_uniform(K.seed, gid, K.counter, 0, 1) * (float)K.n_act);
            best = a > K.n_act - 1 ? K.n_act - 1 : a;
        }
        action[e] = (uint8_t)best;
    }
}

}  // namespace

extern "C" int pvder_qagent_step_f32(const pvder_qagent_cfg *cfg, float *q_table, const float *out, int64_t *state,
                                     uint8_t *action, uint8_t *alive, double *ep_reward, int n_env, void *stream) {
    if (!cfg || !q_table || !out || !state || !action || !alive || !ep_reward || n_env < 1) return kEinval;
    if (cfg->bins < 1 || cfg->n_act < 1 || cfg->n_act > 255 || !(cfg->win > 0.0)) return kEinval;
    double rows = 1.0;
    for (int k = 0; k < kObs; ++k) rows *= cfg->bins;
    if (rows * cfg->n_act > 4.0e12) return kEinval;                    // (a table nobody can allocate)
    AgentK K;
    K.bins = cfg->bins; K.n_act = cfg->n_act; K.first = cfg->first; K.choose = cfg->choose;
    K.one_minus_lr = (float)(1.0 - (double)cfg->lr); K.lr = cfg->lr; K.gamma = cfg->gamma;
    K.epsilon = cfg->epsilon; K.q_done = cfg->q_done; K.low = cfg->low; K.win = cfg->win;
    K.seed = cfg->seed; K.counter = cfg->counter; K.env_id_base = cfg->env_id_base;
    const dim3 grid((n_env + kAgentBlock - 1) / kAgentBlock), block(kAgentBlock);
    hipLaunchKernelGGL(qagent_step_kernel, grid, block, 0, (hipStream_t)stream, K, q_table, out, (long long *)state, action,
                       alive, ep_reward, n_env);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
