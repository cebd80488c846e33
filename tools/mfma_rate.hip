// Probe (GPU box): issue rate of v_mfma_f32_4x4x1_16b_f32 for a lone wave per SIMD, with the operand
// pattern of step_etdlin_kernel (9 / 6 / 3 independent accumulators, A broadcast from block 0).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC, int CBSZ>
__global__ __launch_bounds__(256) void rate(float *out, int iters, long long *cyc) {
    v4f acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 11; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], CBSZ, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int CBSZ>
void run(const char *name, int grid) {
    float *out; long long *cyc, h;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    rate<NACC, CBSZ><<<grid, 256>>>(out, iters, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate<NACC, CBSZ><<<grid, 256>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 11 * NACC;
    printf("%-28s grid %4d: %.1f clock64 ticks / MFMA, %.2f ns / MFMA (event)\n", name, grid, h / n, ms * 1e6 / n);
}
int main() {
    run<9, 0>("9 accumulators, cbsz 0", 256);
    run<9, 4>("9 accumulators, cbsz 4", 256);
    run<6, 4>("6 accumulators, cbsz 4", 256);
    run<3, 4>("3 accumulators, cbsz 4", 256);
    run<1, 4>("1 accumulator, cbsz 4", 256);
    run<9, 4>("9 accumulators, cbsz 4", 1);
    return 0;
}
