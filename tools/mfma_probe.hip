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
// cbsz = 4, abid = B: all 16 blocks take block B's A operand -- ONE register holds sixteen A operands
// (lane 4B + r = row r of product B); what pvder_etdlin.hpp's packed coefficient registers rely on
template <int B>
__global__ void probe_abid(const float *a, const float *b, const float *c, float *d) {
    const int l = threadIdx.x;
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, B, 0);
    d[4 * l] = acc.x; d[4 * l + 1] = acc.y; d[4 * l + 2] = acc.z; d[4 * l + 3] = acc.w;
}
template <int B>
static int check_abid(const float *ha, const float *hb, const float *hc, float *hd, float *a, float *b, float *c, float *d) {
    probe_abid<B><<<1, 64>>>(a, b, c, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad += hd[4 * l + r] != fmaf(ha[4 * B + r], hb[l], hc[4 * l + r]);
    return bad;
}
int main() {
    float ha[64], hb[64], hc[256], hd[256], *a, *b, *c, *d;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (int i = 0; i < 64; ++i) { ha[i] = rnd(); hb[i] = rnd(); }
    for (int i = 0; i < 256; ++i) hc[i] = rnd() * 3.f;
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&c, 1024); (void)hipMalloc(&d, 1024);
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, c, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
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
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad_bc = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad_bc += hd[4 * l + r] != fmaf(ha[r], hb[l], hc[4 * l + r]);
    printf("cbsz=4 abid=0 (A of block 0 broadcast to all blocks): mismatches %d of 256\n", bad_bc);
    const int bad_abid = check_abid<0>(ha, hb, hc, hd, a, b, c, d) + check_abid<1>(ha, hb, hc, hd, a, b, c, d) +
                         check_abid<5>(ha, hb, hc, hd, a, b, c, d) + check_abid<10>(ha, hb, hc, hd, a, b, c, d) +
                         check_abid<15>(ha, hb, hc, hd, a, b, c, d);
    printf("cbsz=4 abid=B (A of block B broadcast; B = 0, 1, 5, 10, 15): mismatches %d of 1280\n", bad_abid);
    return (bad_layout || bad_bc || bad_abid) ? 1 : 0;
}
