// VALU issue-rate probe at one wave per SIMD (debug tool, GPU box only): cycles per instruction for
// dependent and independent chains of the instructions the ETDRK4 inner loop is made of.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
    v2 a = {1.0001f + threadIdx.x * 1e-7f, 0.9999f}, b = {1e-6f, -1e-6f}, c = {0.5f, 0.25f};
    v2 a1 = a + b, a2 = a - b, a3 = a * c;
    float s = a.x, s1 = a.y, s2 = b.x, s3 = c.y;
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.f; lds[threadIdx.x + 512] = 2.f; lds[threadIdx.x + 768] = 3.f;
    __syncthreads();
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                               : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) }
        if (MODE == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(s2), "v"(s3));) }
        if (MODE == 3) { REP8(REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                               : "+v"(s), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b.x), "v"(c.x));)) }
        if (MODE == 4) { REP64(asm volatile("v_sin_f32 %0, %0" : "+v"(s));) }
        if (MODE == 5) { REP8(REP8(asm volatile("v_sin_f32 %0, %0\n v_exp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(s), "+v"(s1), "+v"(s2), "+v"(s3));)) }
        if (MODE == 6) {   // trans op followed by independent pk_fma work (does the trans unit overlap?)
            REP8(REP8(asm volatile("v_sin_f32 %0, %0\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                   : "+v"(s), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) }
        if (MODE == 7) {   // ds_read_b128 + dependent use
            float4 q; int addr = (threadIdx.x & 15) * 16;
            REP64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
                  asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s) : "v"(q.x));)
        }
        if (MODE == 8) {   // pk_fma with an SGPR-pair operand
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(b), "v"(c));)
        }
        if (MODE == 9) {   // dependent pk_mul -> pk_fma alternating two chains
            REP8(REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                                   : "+v"(a), "+v"(a1) : "v"(b), "v"(c));)) }
        if (MODE == 10) {  // v_mov_b32 chain
            REP64(asm volatile("v_mov_b32 %0, %1\n" : "=v"(s) : "v"(s2));) }
        if (MODE == 11) {  // dependent v_mul_f32
            REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s) : "v"(s3));) }
    }
    const unsigned long long t1 = clock64();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a.x + a1.x + a2.y + a3.x + s + s1 + s2 + s3 + a.y;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int MODE>
void run(const char *name, int n_instr_per_iter) {
    const int grid = 256, iters = 200;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 16);
    std::vector<unsigned long long> h(grid * 2);
    for (int r = 0; r < 3; ++r) { probe<MODE><<<grid, 256>>>(out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h.data(), cyc, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> c, w;
    for (int b = 0; b < grid; ++b) { c.push_back((double)h[2 * b]); w.push_back((double)h[2 * b + 1]); }
    std::sort(c.begin(), c.end()); std::sort(w.begin(), w.end());
    const double n = (double)n_instr_per_iter * iters;
    printf("%-44s %7.2f clk64-ticks/instr  %7.3f ns/instr\n", name, c[grid / 2] / n, w[grid / 2] * 10.0 / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_pk_fma_f32 dependent chain", 64);
    run<1>("v_pk_fma_f32 4 independent chains", 256);
    run<9>("v_pk_fma_f32 2 alternating chains", 256);
    run<8>("v_pk_fma_f32 dependent, SGPR-pair operand", 64);
    run<2>("v_fma_f32 dependent chain", 64);
    run<3>("v_fma_f32 4 independent chains", 256);
    run<11>("v_mul_f32 dependent chain", 64);
    run<10>("v_mov_b32 (independent)", 64);
    run<4>("v_sin_f32 dependent chain", 64);
    run<5>("sin/exp/rcp/cos 4 independent", 256);
    run<6>("v_sin + 3 independent pk_fma", 256);
    run<7>("ds_read_b128 + wait + use", 192);
    return 0;
}
