// Workgroup-dispatch probe (debug tool, GPU box only): how long does the dispatcher take to get all
// workgroups of a one-wave-per-SIMD grid running, as a function of VGPR allocation, LDS, kernarg
// size and block shape?   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dispatch_probe tools/dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Big { float v[120]; };

template <int VG, int LDSB, int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe(unsigned long long *t, int spin_ticks, Big big) {
    const unsigned long long t0 = wall_clock64();
    __shared__ float lds[LDSB / 4 > 0 ? LDSB / 4 : 1];
    if (LDSB) {
        lds[threadIdx.x % (LDSB / 4 > 0 ? LDSB / 4 : 1)] = big.v[threadIdx.x % 120];
        __syncthreads();
    }
    if (VG >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VG >= 200) asm volatile("v_mov_b32 v199, 0" ::: "v199");
    if (VG >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    while ((long long)(wall_clock64() - t0) < spin_ticks) {}
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = t0;
        t[blockIdx.x * 2 + 1] = wall_clock64() + (LDSB ? (lds[0] > 1e30f) : 0);
    }
}

template <int VG, int LDSB, int BLOCK>
void run(const char *name, int grid, int spin_us, bool dump = false) {
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
    std::vector<unsigned long long> h(2 * grid);
    Big big{};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> spread, total, evt;
    for (int it = 0; it < 30; ++it) {
        hipEventRecord(e0, 0);
        probe<VG, LDSB, BLOCK><<<grid, BLOCK, 0, 0>>>(d, spin_us * 100, big);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0, end = 0;
        for (int b = 0; b < grid; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b]); end = std::max(end, h[2 * b + 1]); }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 10) { spread.push_back((hi - lo) * 0.01); total.push_back((end - lo) * 0.01); evt.push_back(ms * 1e3); }
        if (dump && it == 29) {
            printf("  entry time by block (us):");
            for (int b = 0; b < grid; ++b) { if (b % 16 == 0) printf("\n   %4d:", b); printf(" %5.2f", (h[2 * b] - lo) * 0.01); }
            printf("\n");
        }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-34s grid %5d x %4d spin %3d us: entry spread %6.2f us, first-entry->last-end %6.2f us, events %6.2f us\n",
           name, grid, BLOCK, spin_us, med(spread), med(total), med(evt));
    hipFree(d);
}

int main() {
    run<0, 0, 256>("small", 256, 0, true);
    run<0, 0, 256>("small", 256, 40);
    run<0, 1536, 256>("lds1.5K", 256, 40);
    run<128, 0, 256>("vgpr128", 256, 40);
    run<200, 0, 256>("vgpr200", 256, 40);
    run<256, 0, 256>("vgpr256", 256, 40);
    run<200, 1536, 256>("vgpr200+lds", 256, 40, true);
    run<200, 1536, 64>("vgpr200+lds block64", 1024, 40);
    run<200, 1536, 128>("vgpr200+lds block128", 512, 40);
    run<200, 1536, 512>("vgpr200+lds block512", 128, 40);
    run<0, 0, 64>("small block64", 1024, 40);
    run<0, 0, 256>("small 4x grid", 1024, 40);
    run<200, 1536, 256>("vgpr200+lds 16 blocks", 16, 40, true);
    return 0;
}
