// pvder_agent.hip -- device-resident tabular Q-learning step (include/pvder_agent.h): the body of the
// reference's QAgent.train loop (gym_PVDER/agents/qagent.py:56-105) for n_env envs in ONE launch per env
// step.  HBM-bound gather / scatter on a bins^11 x 5 fp32 table (6 bins: 7.3 GB): per env 13 coalesced
// reads of the output block, two 20-byte row gathers, one 4-byte scatter, a handful of integer ops.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pvder_agent.h"
#include "pvder_agent_dev.hpp"

namespace {

using pvder_agent::AgentK;
constexpr int kAgentBlock = 256;
constexpr int kEinval = -10001;            // PVDER_EINVAL (pvder_hip.h)
constexpr int kObs = pvder_agent::kObs;

__global__ __launch_bounds__(kAgentBlock) void qagent_step_kernel(const AgentK K, float *__restrict__ q,
                                                                   const float *__restrict__ out, long long *__restrict__ state,
                                                                   uint8_t *__restrict__ action, uint8_t *__restrict__ alive,
                                                                   double *__restrict__ ep_reward, int n_env) {
    __shared__ __attribute__((aligned(16))) float qnet[pvder_agent::kQnetLdsFloats];
    if (K.policy == 1) {             // (wave-uniform) the Q-network's weights: once per workgroup into LDS
        pvder_agent::qnet_stage(q, K.hidden, K.n_act, qnet, (int)threadIdx.x, kAgentBlock);
        __syncthreads();
    }
    const int e = blockIdx.x * kAgentBlock + threadIdx.x;
    if (e >= n_env) return;
    pvder_agent::agent_step_block(K, q, out, (size_t)n_env, e, K.counter, state, action, alive, ep_reward, qnet);
}

}  // namespace

extern "C" int pvder_qagent_step_f32(const pvder_qagent_cfg *cfg, float *q_table, const float *out, int64_t *state,
                                     uint8_t *action, uint8_t *alive, double *ep_reward, int n_env, void *stream) {
    if (!cfg || !q_table || !out || !state || !action || !alive || !ep_reward || n_env < 1) return kEinval;
    if (cfg->policy == PVDER_POLICY_QNET) {
        if (cfg->n_act < 1 || cfg->n_act > pvder_agent::kQnetMaxAct || cfg->hidden < 1 || cfg->hidden > pvder_agent::kQnetMaxHidden) return kEinval;
    } else {
        if (cfg->policy != PVDER_POLICY_TABLE) return kEinval;
        if (cfg->bins < 1 || cfg->n_act < 1 || cfg->n_act > 255 || !(cfg->win > 0.0)) return kEinval;
        double rows = 1.0;
        for (int k = 0; k < kObs; ++k) rows *= cfg->bins;
        if (rows * cfg->n_act > 4.0e12) return kEinval;                    // (a table nobody can allocate)
    }
    AgentK K;
    K.bins = cfg->bins; K.n_act = cfg->n_act; K.first = cfg->first; K.choose = cfg->choose;
    K.one_minus_lr = (float)(1.0 - (double)cfg->lr); K.lr = cfg->lr; K.gamma = cfg->gamma;
    K.epsilon = cfg->epsilon; K.q_done = cfg->q_done; K.low = cfg->low; K.win = cfg->win;
    K.seed = cfg->seed; K.counter = cfg->counter; K.env_id_base = cfg->env_id_base;
    K.policy = cfg->policy; K.hidden = cfg->hidden;
    const dim3 grid((n_env + kAgentBlock - 1) / kAgentBlock), block(kAgentBlock);
    hipLaunchKernelGGL(qagent_step_kernel, grid, block, 0, (hipStream_t)stream, K, q_table, out, (long long *)state, action,
                       alive, ep_reward, n_env);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
