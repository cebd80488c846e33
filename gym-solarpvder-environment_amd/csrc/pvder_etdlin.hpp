// pvder_etdlin.hpp -- device pieces of the linearised exponential integrator ("etdlin") for gfx950.
//
// dy/dt = f(y) = L y + N(y) with L the 11x11 Jacobian of the balanced model at its nominal operating
// point (host: gym_PVDER_amd/etd_linear.py).  Cox & Matthews' ETD3RK is evaluated in "f-form":
//     a  = y + M21 f(y)
//     b  = y + M31 f(y) + M32 f(a)
//     y+ = y + M41 f(y) + M42 f(a) + M43 f(b)
// -- an explicit Runge-Kutta step whose coefficients are DENSE 11x11 matrices (functions of hL,
// folded on the host in fp64).  The exponential carries the whole linear response (stiff current
// loop, PLL pole, DC link, outer PI loops), so TWO steps cover an env step: 1/(60*micro) s behind the
// input jump (new action / event), then one over the remainder (DESIGN.md "etdlin").
//
// This replaces, on the device, the LSODA + analytic-Jacobian integration the reference delegates to
// pvder/SciPy (gym_PVDER/envs/PVDER_env.py:166, :384-386) for envs in normal operation; envs near a
// limit (anti-windup) or off balance take the 23-state path (etdrk4_full_tail).
#pragma once
#include "pvder_model_pk.hpp"

namespace pvder {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kLinN = 11;                 // states of the balanced reduction
constexpr int kLinRows = 12;              // matrix padded to 12 x 12 (three row groups of four; 16-byte rows)
constexpr int kLinMatFloats = kLinRows * kLinRows;       // 144: row-major, row i = 12 floats (columns 0..10 + pad)
constexpr int kLinMats = 6;               // M21 | M31 M32 | M41 M42 M43
constexpr int kLinSetFloats = kLinMats * kLinMatFloats;  // 864 floats = 3456 B per step size
constexpr int kLinMaxSets = 8;
enum { LM21 = 0, LM31, LM32, LM41, LM42, LM43 };

// Full right-hand side of the balanced 11-state model as six pairs (the state layout of the lin
// path): F[0] = di, F[1] = dx, F[2] = du, F[3] = (dVdc, dxDC), F[4] = (dxQ, dxPLL), F[5] = (ddelta, 0).
// Same equations as pvder_model.hpp::rhs for balanced phases, without the anti-windup branches
// (the caller defers envs whose |m| / |iref| come near a limit: m2 = |m_a|^2, ir2 = |iref_a|^2).
template <int PH>
__device__ __forceinline__ void bal_f(const ModelK<float> &P, const PkConst &K, const SubPk &S, const v2 *__restrict__ y,
                                      float qref, float vdcref, v2 kpo, v2 kio, v2 *__restrict__ F, float &m2,
                                      float &ir2) {
    const v2 ia = y[0], xa = y[1], ua = y[2];
    const float Vdc = y[3].x, xPLL = y[4].y, delta = y[5].x;
    const v2 ma = fma2(bc2(P.KpG), ua, xa);
    const float Pa = __builtin_fmaf(ma.y, ia.y, ma.x * ia.x);             // Re(m conj i) of one of PH identical phases
    const v2 tq = fma2(bc2((float)PH * K.qk), ia, bc2(0.5f * PH) * S.Asw[0]);
    const float Q = __builtin_fmaf(tq.y, ia.y, tq.x * ia.x);             // Im S_PCC
    // outer loops, packed: e = (Vdc_ref - Vdc, Q_ref - Q); iref_a = (xDC, xQ) + (Kp_DC, -Kp_Q) e
    const v2 e = mk2(vdcref - Vdc, qref - Q);
    const v2 iar = fma2(kpo, e, mk2(y[3].y, y[4].x));
    const v2 dpi = kio * e;                                               // (dxDC, dxQ)
    ir2 = __builtin_fmaf(iar.y, iar.y, iar.x * iar.x);
    float s1, c1;
    Math<float>::sincos_turns(delta * (float)(1.0 / kTwoPi), s1, c1);
    v2 V1 = fma2(bc2(P.kviR), ia, S.A0);
    V1 = fma2(mk2(-P.kviI, P.kviI), swp(ia), V1);                         // v_a (balanced: V2 = 0)
    const float vd = __builtin_fmaf(V1.y, s1, V1.x * c1);
    const float dw = __builtin_fmaf(P.KpP, vd, xPLL);                      // we - wbase
    // di/dt = cm m - A/Lf - Re(Z) i + (Im(Z) + winv/wbase) (iI, -iR),  Z = (kvi + Rf)/Lf
    const float zw = K.zi0 + 1.f + dw * P.iwb;
    v2 n = fma2(bc2(K.hiLf * Vdc), ma, S.D[0]);
    n = fma2(bc2(-K.ZR), ia, n);
    F[0] = fma2(mk2(zw, -zw), swp(ia), n);
    F[1] = bc2(P.KiG) * ua;
    F[2] = bc2(P.wp) * (iar - ia - ua);
    // dVdc = (Ppv - Re S)/(Vdc C), Ppv = max(0, Vdc g(Vdc)), Re S = Vdc PH Pa/4  ->  no division
    const float g = __builtin_fmaf(-P.pvB, Math<float>::exp_pv(P.pvK * Vdc), S.pv1);
    const float dVdc = __builtin_fmaf(-0.25f * PH * P.iC, Pa, P.iC * fmaxf(g, 0.f));
    F[3] = mk2(dVdc, dpi.x);
    F[4] = mk2(dpi.y, P.KiP * vd);
    F[5] = mk2(dw, 0.f);
    m2 = __builtin_fmaf(ma.y, ma.y, ma.x * ma.x);
}

// ---- matrix-vector products on the matrix cores ------------------------------------------------------
// out = M f  with a DENSE 12x11 matrix M shared by all envs and f = one vector per env (lane) is a GEMM
// with a shared left operand: [12 x 11] x [11 x 64 lanes].  v_mfma_f32_4x4x1_16b_f32 computes, for each of
// its 16 blocks of 4 lanes, the outer product  D[r][j] += A[r] * B[j]  (r = row = accumulator register,
// j = lane within the block).  With B = f_k exactly as it sits in its VGPR (one env per lane) and
// A = M[4g + r][k] in lane r of block 0, BROADCAST to all 16 blocks by the instruction itself
// (cbsz = 4, abid = 0), the four accumulator registers of row group g receive rows 4g..4g+3 of  M f
// for the lane's own env -- the layout the rest of the kernel uses, no shuffles.  One MFMA does the
// work of two v_pk_fma_f32 + their operand read, on the otherwise idle matrix pipe, and each element
// is ONE fused multiply-add in the same k order a VALU chain would use (tools/mfma_probe.hip:
// bit-identical to fmaf; broadcast semantics checked there too).  fp32 in, fp32 out: no precision is
// traded.  The A operands of a coefficient set (6 matrices x 3 row groups x 11 columns = 198
// registers, only lanes 0..3 of which carry data) are read once -- three 16-byte loads per matrix row,
// by four lanes -- and stay in the register file (512 VGPRs + AGPRs at one wave per SIMD).
typedef float LinSetRegs[kLinMats][3][kLinRows];
typedef float LinMatRegs[3][kLinRows];

// lanes 0..3: row 4g + lane of the matrix (row-major, 12 floats per row); other lanes: not read by the MFMA
__device__ __forceinline__ void lin_load_mat(const float *__restrict__ m, int lane, LinMatRegs &A) {
    if (lane < 4) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const v4f *p = (const v4f *)(m + (4 * g + lane) * kLinRows);
            const v4f a = p[0], b = p[1], c = p[2];
            A[g][0] = a.x; A[g][1] = a.y; A[g][2] = a.z; A[g][3] = a.w;
            A[g][4] = b.x; A[g][5] = b.y; A[g][6] = b.z; A[g][7] = b.w;
            A[g][8] = c.x; A[g][9] = c.y; A[g][10] = c.z; A[g][11] = c.w;
            // pin the operands to the accumulation-register half of the file (the MFMA reads its A operand
            // from there just as well): left to itself hipcc keeps them in the 256 architectural VGPRs and
            // starves the VALU code of the step loop of registers
#pragma unroll
            for (int k = 0; k < kLinN; ++k) asm volatile("" : "+a"(A[g][k]));
        }
    }
}
__device__ __forceinline__ void lin_load_set(const float *__restrict__ set, int lane, LinSetRegs &A) {
#pragma unroll
    for (int m = 0; m < kLinMats; ++m) lin_load_mat(set + m * kLinMatFloats, lane, A[m]);
}

// acc[j] += M_mat[j] f  for NM matrices of the set that share the input vector f (six pairs)
template <int NM>
__device__ __forceinline__ void lin_mfma(const LinSetRegs &A, const int (&mat)[NM], const v2 *__restrict__ f,
                                         v4f (&acc)[NM][3]) {
#pragma unroll
    for (int k = 0; k < kLinN; ++k) {
        const float fk = (k & 1) ? f[k >> 1].y : f[k >> 1].x;
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[j][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[mat[j]][g][k], fk, acc[j][g], 4, 0, 0);
    }
}
__device__ __forceinline__ void lin_mfma1(const LinMatRegs &A, const v2 *__restrict__ f, v4f (&acc)[3]) {
#pragma unroll
    for (int k = 0; k < kLinN; ++k) {
        const float fk = (k & 1) ? f[k >> 1].y : f[k >> 1].x;
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[g][k], fk, acc[g], 4, 0, 0);
    }
}
// state in accumulator layout (rows 0-3, 4-7, 8-11) -> the six pairs bal_f reads
__device__ __forceinline__ void lin_pairs(const v4f (&Y)[3], v2 *__restrict__ y) {
#pragma unroll
    for (int g = 0; g < 3; ++g) { y[2 * g] = mk2(Y[g].x, Y[g].y); y[2 * g + 1] = mk2(Y[g].z, Y[g].w); }
}

}  // namespace pvder
