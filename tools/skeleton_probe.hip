// Memory-skeleton probe (debug tool, GPU box only): what does ONE LAUNCH PER ENV STEP cost at 65 536 envs (one wave per
// SIMD, 256 workgroups) BEFORE any arithmetic -- the launch-to-launch period of dependent launches that only read R rows
// of a structure-of-arrays state, wait, optionally make a dependent second fetch, spin for a given time in place of the
// integration, and write W rows?  The floor of every per-step kernel design for this batch size.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/skeleton_probe tools/skeleton_probe.hip && gpurun_out/skeleton_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kBlock = 256;
static int kN = 65536;        // envs per launch (argv[1])

template <int R, int W, bool DEP, bool NT>
__global__ __launch_bounds__(kBlock) void skel(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ tab,
                                               const int *__restrict__ idx, int spin_ticks, int n) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    float acc = 0.f;
    float v[R > 0 ? R : 1];
    int k = 0;
    if (DEP) k = NT ? __builtin_nontemporal_load(&idx[e]) : idx[e];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = NT ? __builtin_nontemporal_load(&in[(size_t)r * n + e]) : in[(size_t)r * n + e];
    if (DEP) acc += tab[(size_t)(k & 31) * n + e];            // the dependent fetch: an event-table row named by a loaded word
#pragma unroll
    for (int r = 0; r < R; ++r) acc += v[r];
    if (spin_ticks > 0) {
        // "integration": dependent arithmetic for about spin_ticks/100 us (depends on the loaded data)
        const unsigned long long t0 = wall_clock64();
        while ((long long)(wall_clock64() - t0) < spin_ticks) acc = __builtin_fmaf(acc, 0.999f, 1e-3f);
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (NT) __builtin_nontemporal_store(acc + (float)w, &out[(size_t)w * n + e]);
        else out[(size_t)w * n + e] = acc + (float)w;
    }
}

// The same rows moved with 16-byte accesses: wave w of the workgroup fetches rows w, w + 4, ... for ALL 256 envs of the workgroup
// (one dwordx4 per lane = 1 KB per wave-instruction instead of 256 B), parks them in LDS, and every lane picks its own env's
// words up from there (and the reverse for the stores).  The layout in HBM is unchanged.
template <int R, int W>
__global__ __launch_bounds__(kBlock) void skelv(const float *__restrict__ in, float *__restrict__ out, int spin_ticks, int n) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int RM = R > W ? R : W;
    __shared__ __attribute__((aligned(16))) float lds[RM * kBlock];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * kBlock;
#pragma unroll
    for (int r = wave; r < R; r += 4) {
        const f4 v = __builtin_nontemporal_load((const f4 *)(in + (size_t)r * n + base) + lane);
        *((f4 *)(lds + r * kBlock) + lane) = v;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += lds[r * kBlock + threadIdx.x];
    if (spin_ticks > 0) {
        const unsigned long long t0 = wall_clock64();
        while ((long long)(wall_clock64() - t0) < spin_ticks) acc = __builtin_fmaf(acc, 0.999f, 1e-3f);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w) lds[w * kBlock + threadIdx.x] = acc + (float)w;
    __syncthreads();
#pragma unroll
    for (int w = wave; w < W; w += 4) {
        const f4 v = *((const f4 *)(lds + w * kBlock) + lane);
        __builtin_nontemporal_store(v, (f4 *)(out + (size_t)w * n + base) + lane);
    }
}

template <int R, int W>
double runv(const char *name, int spin_us_x100, float *a, float *b) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 2000;
    std::vector<double> t;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < K; ++k) skelv<R, W><<<kN / kBlock, kBlock, 0, 0>>>(k & 1 ? b : a, k & 1 ? a : b, spin_us_x100, kN);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms * 1e3 / K);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s R %2d W %2d 16-byte accesses through LDS spin %4.2f us: %6.2f us per launch (best %5.2f)\n", name, R, W,
           spin_us_x100 * 0.01, t[t.size() / 2], t[0]);
    return t[t.size() / 2];
}

template <int R, int W, bool DEP, bool NT>
double run(const char *name, int spin_us_x100, float *a, float *b, const float *tab, const int *idx) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int K = 2000;
    std::vector<double> t;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < K; ++k) {          // dependent launches: each reads what the previous one wrote (ping-pong)
            skel<R, W, DEP, NT><<<kN / kBlock, kBlock, 0, 0>>>(k & 1 ? b : a, k & 1 ? a : b, tab, idx, spin_us_x100, kN);
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms * 1e3 / K);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s R %2d W %2d dep %d nt %d spin %4.2f us: %6.2f us per launch (best %5.2f)\n", name, R, W, (int)DEP, (int)NT,
           spin_us_x100 * 0.01, t[t.size() / 2], t[0]);
    return t[t.size() / 2];
}

int main(int argc, char **argv) {
    if (argc > 1) kN = atoi(argv[1]);
    printf("# %d envs per launch = %d workgroups of %d\n", kN, kN / kBlock, kBlock);
    float *a, *b, *tab;
    int *idx;
    hipMalloc(&a, sizeof(float) * 48 * kN);
    hipMalloc(&b, sizeof(float) * 48 * kN);
    hipMalloc(&tab, sizeof(float) * 32 * kN);
    hipMalloc(&idx, sizeof(int) * kN);
    hipMemset(a, 0, sizeof(float) * 48 * kN);
    hipMemset(b, 0, sizeof(float) * 48 * kN);
    hipMemset(tab, 0, sizeof(float) * 32 * kN);
    hipMemset(idx, 0, sizeof(int) * kN);
    // (eager launches from one host thread cost ~3 us apiece: without arithmetic every variant reads as the HOST's launch rate;
    //  with 2 us of stand-in arithmetic every variant is GPU-bound and the differences are the memory skeleton's)
    for (int spin : {0, 200}) {
        run<0, 1, false, true>("empty", spin, a, b, tab, idx);
        run<1, 1, false, true>("1 row in, 1 row out", spin, a, b, tab, idx);
        run<13, 1, false, true>("13 rows in", spin, a, b, tab, idx);
        run<29, 1, false, true>("29 rows in", spin, a, b, tab, idx);
        run<29, 1, true, true>("29 rows in + dependent fetch", spin, a, b, tab, idx);
        run<29, 13, false, true>("29 in, 13 out", spin, a, b, tab, idx);
        run<29, 29, false, true>("29 in, 29 out", spin, a, b, tab, idx);
        run<29, 41, false, true>("29 in, 41 out", spin, a, b, tab, idx);
        run<13, 41, false, true>("13 in, 41 out", spin, a, b, tab, idx);
        run<29, 41, true, true>("29 in + dep, 41 out (the step's skeleton)", spin, a, b, tab, idx);
        run<29, 41, true, false>("the same, plain loads / stores", spin, a, b, tab, idx);
        run<29, 41, false, false>("29 in, 41 out, plain", spin, a, b, tab, idx);
        run<13, 13, false, true>("13 in, 13 out", spin, a, b, tab, idx);
        runv<29, 1>("29 in (16 B, via LDS)", spin, a, b);
        runv<29, 41>("29 in, 41 out (16 B, via LDS)", spin, a, b);
        runv<13, 13>("13 in, 13 out (16 B, via LDS)", spin, a, b);
    }
    run<29, 41, true, true>("skeleton + 3.3 us of arithmetic", 330, a, b, tab, idx);
    run<0, 1, false, true>("empty + 3.3 us of arithmetic", 330, a, b, tab, idx);
    return 0;
}
