// pvder_agent_dev.hpp -- device side of the tabular Q-learning step (include/pvder_agent.h), shared by the
// stand-alone agent kernel (pvder_agent.hip: one launch behind each step launch) and by the closed-loop fused
// rollout (pvder_kernels.hip: the agent step between two env steps of ONE launch, the env state still in
// registers).  The body of the reference's QAgent.train loop, /root/reference/gym_PVDER/agents/qagent.py:56-105.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pvder_agent {

constexpr int kObs = 11;

struct AgentK {
    int bins, n_act, first, choose;
    float one_minus_lr, lr, gamma, epsilon, q_done;
    double low, win;
    uint64_t seed, counter;
    long long env_id_base;
    int policy, hidden;             // pvder_qagent_cfg.policy / .hidden (PVDER_POLICY_QNET: q = the network's weights)
};
constexpr int kQnetMaxAct = 8;
constexpr int kQnetMaxHidden = 128;
// LDS image of the network (staged once per launch by qnet_stage): W1 rows padded to 12 floats, W2 TRANSPOSED with rows of 8
// (three + two 16-byte broadcast reads per hidden unit), b1, b2 -- at fixed offsets
constexpr int kQnetW1 = 0, kQnetW2 = kQnetMaxHidden * 12, kQnetB1 = kQnetW2 + kQnetMaxHidden * 8, kQnetB2 = kQnetB1 + kQnetMaxHidden,
              kQnetLdsFloats = kQnetB2 + 8;

// w (global, include/pvder_agent.h PVDER_POLICY_QNET): W1 [H][11], b1 [H], W2 [A][H], b2 [A]  ->  the LDS image; all threads of the
// workgroup call it, the caller synchronises
__device__ __forceinline__ void qnet_stage(const float *__restrict__ w, int H, int A, float *__restrict__ lds, int tid, int nthreads) {
    const float *__restrict__ b1 = w + (size_t)H * kObs, *__restrict__ W2 = b1 + H, *__restrict__ b2 = W2 + (size_t)A * H;
    for (int i = tid; i < H * 12; i += nthreads) { const int j = i / 12, k = i - 12 * j; lds[kQnetW1 + i] = k < kObs ? w[j * kObs + k] : 0.f; }
    for (int i = tid; i < H * 8; i += nthreads) { const int j = i >> 3, a = i & 7; lds[kQnetW2 + i] = a < A ? W2[a * H + j] : 0.f; }
    for (int i = tid; i < H; i += nthreads) lds[kQnetB1 + i] = b1[i];
    if (tid < 8) lds[kQnetB2 + tid] = tid < A ? b2[tid] : -3.0e38f;
}

// The Q-network of the reference's DQN demo (examples/gym_PVDER_environment_tf_agents_DQN_demo.ipynb cell 17: QNetwork with
// fc_layer_params = (100,)) as the greedy action of one env: relu(W1 obs + b1) -> W2 h + b2 -> first maximum.  Every lane reads
// the same LDS words (broadcast reads, no bank conflicts); H (11 + A) fused multiply-adds per env in the order k = 0..10, j = 0..H-1.
// (Round 5, first version: the weights through the table pointer -- a non-const global pointer the tabular agent writes
//  through, so hipcc issued 1600 per-lane vector loads per env step: 45 us; from LDS: see profiles/r05_agent_rate.txt)
typedef float v4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int qnet_greedy(const float *__restrict__ lds, int H, int A, const float (&obs)[kObs], float (&qout)[kQnetMaxAct]) {
#pragma unroll
    for (int a = 0; a < kQnetMaxAct; ++a) qout[a] = lds[kQnetB2 + a];
    for (int j = 0; j < H; ++j) {
        const v4q *r1 = (const v4q *)(lds + kQnetW1 + 12 * j);
        const v4q w0 = r1[0], w1 = r1[1], w2 = r1[2];
        float h = lds[kQnetB1 + j];
        h = __builtin_fmaf(w0.x, obs[0], h); h = __builtin_fmaf(w0.y, obs[1], h); h = __builtin_fmaf(w0.z, obs[2], h); h = __builtin_fmaf(w0.w, obs[3], h);
        h = __builtin_fmaf(w1.x, obs[4], h); h = __builtin_fmaf(w1.y, obs[5], h); h = __builtin_fmaf(w1.z, obs[6], h); h = __builtin_fmaf(w1.w, obs[7], h);
        h = __builtin_fmaf(w2.x, obs[8], h); h = __builtin_fmaf(w2.y, obs[9], h); h = __builtin_fmaf(w2.z, obs[10], h);
        h = fmaxf(h, 0.f);
        const v4q *r2 = (const v4q *)(lds + kQnetW2 + 8 * j);
        const v4q u0 = r2[0], u1 = r2[1];
        qout[0] = __builtin_fmaf(u0.x, h, qout[0]); qout[1] = __builtin_fmaf(u0.y, h, qout[1]); qout[2] = __builtin_fmaf(u0.z, h, qout[2]);
        qout[3] = __builtin_fmaf(u0.w, h, qout[3]); qout[4] = __builtin_fmaf(u1.x, h, qout[4]); qout[5] = __builtin_fmaf(u1.y, h, qout[5]);
        qout[6] = __builtin_fmaf(u1.z, h, qout[6]); qout[7] = __builtin_fmaf(u1.w, h, qout[7]);
    }
    int best = 0;
    float bq = qout[0];
#pragma unroll
    for (int a = 1; a < kQnetMaxAct; ++a)
        if (a < A && qout[a] > bq) { bq = qout[a]; best = a; }
    return best;
}

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

// get_discrete_state (qagent.py:43-45): (state - low) / win in float64, truncated; the Box bound itself (+10)
// falls into the last bin.  Returns the flat table row (first observation component most significant).
__device__ __forceinline__ long long discretise(const AgentK &K, const float (&obs)[kObs]) {
    long long s = 0;
#pragma unroll
    for (int k = 0; k < kObs; ++k) {
        long long d = (long long)(((double)obs[k] - K.low) / K.win);
        d = d < 0 ? 0 : (d > K.bins - 1 ? K.bins - 1 : d);
        s = s * K.bins + d;
    }
    return s;
}

// One agent step of env `e` on the transition (state, action) -> obs with reward r / done: table update
// (qagent.py:83-103), bookkeeping, next epsilon-greedy action (:63-68).  `counter`: exploration key of this call.
// state / action / alive / ret are the env's agent-side registers (or its array slots).
__device__ __forceinline__ void agent_step(const AgentK &K, float *__restrict__ q, const float (&obs)[kObs], float r, bool done,
                                           long long gid, uint64_t counter, long long &state, int &action, bool &alive,
                                           double &ret, const float *__restrict__ qnet_lds = nullptr) {
    if (K.policy == 1) {
        // PVDER_POLICY_QNET: the network chooses, nobody learns here (the trainer holds the weights); the episode
        // bookkeeping of the tabular agent stays (return of the running episode, alive)
        if (K.first) {
            alive = true;
            ret = 0.0;
        } else if (alive) {
            ret += (double)r;
            alive = !done;
        }
        if (K.choose) {
            float qv[kQnetMaxAct];
            int best = qnet_greedy(qnet_lds, K.hidden, K.n_act, obs, qv);
            const float u = counter_uniform(K.seed, (uint64_t)gid, counter, 0, 0);
            if (u <= K.epsilon) {
                int a = (int)(counter_uniform(K.seed, (uint64_t)gid, counter, 0, 1) * (float)K.n_act);
                best = a > K.n_act - 1 ? K.n_act - 1 : a;
            }
            action = best;
        }
        return;
    }
    const long long s2 = discretise(K, obs);
    const float *row2 = q + (size_t)s2 * K.n_act;
    if (K.first) {
        alive = true;
        ret = 0.0;
    } else if (alive) {
        float *entry = q + (size_t)state * K.n_act + action;
        float nq = K.q_done;                                           // :100-103
        if (!done) {                                                   // :83-95
            float mx = row2[0];
            for (int j = 1; j < K.n_act; ++j) mx = fmaxf(mx, row2[j]);
            nq = K.one_minus_lr * *entry + K.lr * (r + K.gamma * mx);
        }
        *entry = nq;
        ret += (double)r;
        alive = !done;
    }
    state = s2;
    if (K.choose) {                                                    // :63-68
        int best = 0;
        float bq = row2[0];
        for (int j = 1; j < K.n_act; ++j) {                            // np.argmax: the first maximum
            const float v = row2[j];
            if (v > bq) { bq = v; best = j; }
        }
        const float u = counter_uniform(K.seed, (uint64_t)gid, counter, 0, 0);
        if (u <= K.epsilon) {                                          // np.random.random() > epsilon -> greedy
            int a = (int)(counter_uniform(K.seed, (uint64_t)gid, counter, 0, 1) * (float)K.n_act);
            best = a > K.n_act - 1 ? K.n_act - 1 : a;
        }
        action = best;
    }
}

// the same on array slots, observation / reward / done read from an output block [13][n_env]
__device__ __forceinline__ void agent_step_block(const AgentK &K, float *__restrict__ q, const float *__restrict__ out, size_t N,
                                                 int e, uint64_t counter, long long *__restrict__ state,
                                                 uint8_t *__restrict__ action, uint8_t *__restrict__ alive,
                                                 double *__restrict__ ep_reward, const float *__restrict__ qnet_lds = nullptr) {
    float obs[kObs];
#pragma unroll
    for (int k = 0; k < kObs; ++k) obs[k] = out[(size_t)k * N + e];
    long long s = state[e];
    int a = action[e];
    bool al = alive[e] != 0;
    double ret = ep_reward[e];
    const float r = out[(size_t)11 * N + e];
    const bool done = out[(size_t)12 * N + e] != 0.f;
    agent_step(K, q, obs, r, done, K.env_id_base + e, counter, s, a, al, ret, qnet_lds);
    state[e] = s;
    if (K.choose) action[e] = (uint8_t)a;
    alive[e] = al ? 1 : 0;
    ep_reward[e] = ret;
}

}  // namespace pvder_agent
