// pvder_step_reset.hpp -- part of pvder_kernels.hip (ONE translation unit: included there, in order; not a header of its own).
// Masked reset, event-table completion.
#pragma once

namespace {

// =============================================================================================
// masked reset: steady state broadcast + event-table refill + reset observation
// =============================================================================================
// FINISHED = true: auto-reset pass behind a step kernel -- resets the envs whose status != 0 and
// leaves the reward/done rows (the transition that ended the episode) untouched
template <typename T, int PH, bool FINISHED = false>
__global__ __launch_bounds__(kBlock) void reset_kernel(
    const ModelK<T> P, const ResetK R, T *__restrict__ fstate, int32_t *__restrict__ istate,
    float *__restrict__ ev_vg, float *__restrict__ ev_si, const uint8_t *__restrict__ mask,
    T *__restrict__ out, int n_env, const float *__restrict__ lane_par = nullptr) {
    constexpr int NS = 6 * PH + 5;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_env) return;
    const size_t N = (size_t)n_env;
    if (FINISHED) {
        if (istate[PVDER_I_STATUS * N + e] == 0) return;
    } else if (mask && mask[e] == 0) {
        return;
    }
    T y[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) y[k] = (T)R.y0[k];
    ModelK<T> Pl = P;
    if (lane_par) {
        // per-lane parameters: the env's own steady state, and the PV constants its reset observation (Ppv) is taken with
        lane_steady_state<T, PH>(lane_par, N, e, y);
        Pl.pvA = (T)lane_par[(size_t)PVDER_LPF_PVA * N + e];
        Pl.pvB = (T)lane_par[(size_t)PVDER_LPF_PVB * N + e];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) fstate[(size_t)k * N + e] = y[k];
    const T qref = (T)0, vdcref = P.vdc_ref0;
    fstate[(size_t)NS * N + e] = qref;
    fstate[(size_t)(NS + 1) * N + e] = vdcref;
    const int episode = istate[PVDER_I_EPISODE * N + e] + 1;
    istate[PVDER_I_KSUB * N + e] = 0;
    istate[PVDER_I_STEPS * N + e] = 0;
    istate[PVDER_I_STATUS * N + e] = 0;
    istate[PVDER_I_EPISODE * N + e] = episode;
    if (!R.keep_events)
        fill_events(R.spec, (float)P.sinsol0, R.seed, (uint64_t)(R.env_id_base + (R.env_ids ? (int64_t)R.env_ids[e] : (int64_t)e)), (uint64_t)episode, ev_vg,
                    ev_si, N, e);
    if (out) {
        SubK<T, PH> S;
        pvder::make_sub<T, PH>(Pl, (T)ev_vg[e], (T)ev_si[e], S);
        T iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms;
        pvder::outputs<T, PH>(Pl, S, y, iaR, iaI, vaR, vaI, Ppcc, Qpcc, ppv, vrms);
        write_obs<T>(out, N, e, iaR, iaI, vaR, vaI, Ppcc, Qpcc, y[6 * PH], ppv, vdcref, qref, (T)0);
        if (!FINISHED) {
            out[PVDER_O_REWARD * N + e] = (T)0;
            out[PVDER_O_DONE * N + e] = (T)0;
        }
    }
}

}  // namespace
