// Probe (GPU box): operand layout and rounding of v_mfma_f32_4x4x1_16b_f32 -- what step_etdlin_kernel's
// matrix-vector products rely on:  D[r] (VGPR r, lane l) = A(lane 4*(l/4) + r) * B(lane l) + C[r](lane l),
// computed as ONE fused multiply-add per element (bit-identical to fmaf).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float *a, const float *b, const float *c, float *d) {
    const int l = threadIdx.x;
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    d[4 * l] = acc.x; d[4 * l + 1] = acc.y; d[4 * l + 2] = acc.z; d[4 * l + 3] = acc.w;
}
// cbsz = 4, abid = 0: all 16 blocks take block 0's A operand -- only lanes 0..3 of the A register matter
__global__ void probe_bcast(const float *a, const float *b, const float *c, float *d) {
    const int l = threadIdx.x;
    float av = 1e30f;                       // poison: lanes >= 4 must not be read
    if (l < 4) av = a[l];
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b[l], acc, 4, 0, 0);
    d[4 * l] = acc.x; d[4 * l + 1] = acc.y; d[4 * l + 2] = acc.z; d[4 * l + 3] = acc.w;
}
int main() {
    float ha[64], hb[64], hc[256], hd[256], *a, *b, *c, *d;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (int i = 0; i < 64; ++i) { ha[i] = rnd(); hb[i] = rnd(); }
    for (int i = 0; i < 256; ++i) hc[i] = rnd() * 3.f;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, c, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad_layout = 0, bad_bits = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float want = fmaf(ha[4 * (l / 4) + r], hb[l], hc[4 * l + r]);
            const float got = hd[4 * l + r];
            if (fabsf(got - want) > 1e-5f * (1.f + fabsf(want))) ++bad_layout;
            else if (got != want) ++bad_bits;
        }
    printf("mfma_f32_4x4x1f32: layout mismatches %d, fused-rounding mismatches %d of 256\n", bad_layout, bad_bits);
    probe_bcast<<<1, 64>>>(a, b, c, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad_bc = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad_bc += hd[4 * l + r] != fmaf(ha[r], hb[l], hc[4 * l + r]);
    printf("cbsz=4 abid=0 (A of block 0 broadcast to all blocks): mismatches %d of 256\n", bad_bc);
    return (bad_layout || bad_bc) ? 1 : 0;
}
