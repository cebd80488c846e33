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
constexpr int kLinMaxSets = 12;
enum { LM21 = 0, LM31, LM32, LM41, LM42, LM43 };

// Full right-hand side of the balanced 11-state model as six pairs (the state layout of the lin
// path): F[0] = di, F[1] = dx, F[2] = du, F[3] = (dVdc, dxDC), F[4] = (dxQ, dxPLL), F[5] = (ddelta, 0).
// Same equations as pvder_model.hpp::rhs for balanced phases, without the duty-cycle anti-windup (component-wise
// per phase, not rotation invariant: the caller hands envs whose |m| comes near its limit to the 23-state path;
// m2 = |m_a|^2, ir2 = |iref_a|^2).
// AW = true (the "careful" pass for envs at the CURRENT limit, lin_integrate_careful): the conditional integration
// of the outer PI loops -- rotation invariant, iref_a is common to the phases -- in the form the hybrid steps need:
// an outer integrator that moves outward (derivative and state of one sign: the component the reference model
// freezes while |iref| > iref_limit) advances at `alpha` times its rate; alpha = 1 free, 0 frozen, in between the
// fraction of the step spent free.  alpha < 0: decided here, frozen iff |iref| > iref_limit at this state.
template <int PH, bool AW = false>
__device__ __forceinline__ void bal_f(const ModelK<float> &P, const PkConst &K, const SubPk &S, const v2 *__restrict__ y,
                                      float qref, float vdcref, v2 kpo, v2 kio, v2 *__restrict__ F, float &m2,
                                      float &ir2, float alpha = 1.f) {
    const v2 ia = y[0], xa = y[1], ua = y[2];
    const float Vdc = y[3].x, xPLL = y[4].y, delta = y[5].x;
    const v2 ma = fma2(bc2(P.KpG), ua, xa);
    const float Pa = __builtin_fmaf(ma.y, ia.y, ma.x * ia.x);             // Re(m conj i) of one of PH identical phases
    const v2 tq = fma2(bc2((float)PH * K.qk), ia, bc2(0.5f * PH) * S.Asw[0]);
    const float Q = __builtin_fmaf(tq.y, ia.y, tq.x * ia.x);             // Im S_PCC
    // outer loops, packed: e = (Vdc_ref - Vdc, Q_ref - Q); iref_a = (xDC, xQ) + (Kp_DC, -Kp_Q) e
    const v2 e = mk2(vdcref - Vdc, qref - Q);
    const v2 iar = fma2(kpo, e, mk2(y[3].y, y[4].x));
    v2 dpi = kio * e;                                                     // (dxDC, dxQ)
    ir2 = __builtin_fmaf(iar.y, iar.y, iar.x * iar.x);
    if constexpr (AW) {
        if (alpha < 0.f) alpha = ir2 > P.ilim2 ? 0.f : 1.f;
        dpi.x = dpi.x * y[3].y > 0.f ? alpha * dpi.x : dpi.x;
        dpi.y = dpi.y * y[4].x > 0.f ? alpha * dpi.y : dpi.y;
    }
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
    float g = __builtin_fmaf(-P.pvB, Math<float>::exp_pv(P.pvK * Vdc), S.pv1);
    if (P.pvK == 0.f) g = -S.pv1 * Math<float>::rcp(Vdc);      // (model option U5, wave-uniform: pv1 = -Ppv(Vdcmpp0), see ModelK)
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
// A = M[4g + r][k] BROADCAST to all 16 blocks by the instruction itself (cbsz = 4: every block takes the
// A operand of block `abid`), the four accumulator registers of row group g receive rows 4g..4g+3 of
// M f  for the lane's own env -- the layout the rest of the kernel uses, no shuffles.  One MFMA does the
// work of two v_pk_fma_f32 + their operand read, on the otherwise idle matrix pipe, and each element
// is ONE fused multiply-add in the same k order a VALU chain would use (tools/mfma_probe.hip:
// bit-identical to fmaf; broadcast semantics checked there too).  fp32 in, fp32 out: no precision is
// traded.
// Operand packing: `abid` selects WHICH block's four lanes are broadcast, so ONE register holds the A
// operands of sixteen products -- lane 4k + r of the register of row group g carries M[4g + r][k], and the
// product with column k names abid = k.  A matrix is 3 registers, a coefficient set (6 matrices) 18, the
// tick matrix 3: 21 registers for everything the step loop multiplies with (the first version kept one
// register per (matrix, row group, column), data in lanes 0..3 only: 231 registers, the whole AGPR file).
typedef float LinMatRegs[3];
typedef float LinSetRegs[kLinMats][3];
// Krogstad's ETDRK4 in f-form (round 4: the env step behind a mere action, lin_k4): U2 = y + M21 F1, U3 = y + M31 F1 + M32 F2,
// U4 = y + M41 F1 + M42 F2 + M43 F3, y+ = y + B1 F1 + B2 F2 + B3 F3 + B4 F4 -- ten dense matrices (a zero of the N-form tableau
// is not one here), 30 registers
constexpr int kLinK4Mats = 10;
constexpr int kLinK4Floats = kLinK4Mats * kLinMatFloats;
enum { KM21 = 0, KM31, KM32, KM41, KM42, KM43, KB1, KB2, KB3, KB4 };
typedef float LinK4Regs[kLinK4Mats][3];

// lane 4k + r <- M[4g + r][k]  (row-major, 12 floats per row; blocks 11..15 are never selected)
__device__ __forceinline__ void lin_load_mat(const float *__restrict__ m, int lane, LinMatRegs &A) {
    const int k = lane >> 2, r = lane & 3;
    const float *p = m + r * kLinRows + (k < kLinRows ? k : 0);
#pragma unroll
    for (int g = 0; g < 3; ++g) A[g] = p[4 * g * kLinRows];
}
__device__ __forceinline__ void lin_load_set(const float *__restrict__ set, int lane, LinSetRegs &A) {
#pragma unroll
    for (int m = 0; m < kLinMats; ++m) lin_load_mat(set + m * kLinMatFloats, lane, A[m]);
}
__device__ __forceinline__ void lin_load_k4(const float *__restrict__ blk, int lane, LinK4Regs &A) {
#pragma unroll
    for (int m = 0; m < kLinK4Mats; ++m) lin_load_mat(blk + m * kLinMatFloats, lane, A[m]);
}

// acc[j] += M_mat[j] f  for NM matrices of the set that share the input vector f (six pairs); column K and
// the ones behind it (abid is an immediate of the instruction: compile-time recursion over the columns)
template <int NM, int K, int NSET>
__device__ __forceinline__ void lin_mfma_cols(const float (&A)[NSET][3], const int (&mat)[NM], const v2 *__restrict__ f,
                                              v4f (&acc)[NM][3]) {
    if constexpr (K < kLinN) {
        const float fk = (K & 1) ? f[K >> 1].y : f[K >> 1].x;
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[j][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[mat[j]][g], fk, acc[j][g], 4, K, 0);
        lin_mfma_cols<NM, K + 1, NSET>(A, mat, f, acc);
    }
}
template <int NM, int NSET>
__device__ __forceinline__ void lin_mfma(const float (&A)[NSET][3], const int (&mat)[NM], const v2 *__restrict__ f,
                                         v4f (&acc)[NM][3]) {
    lin_mfma_cols<NM, 0, NSET>(A, mat, f, acc);
}
// acc += M f for ONE matrix: three row groups give three dependent chains of eleven MFMAs, and a dependent
// 4x4x1 issues every ~16 clocks where an independent one issues every 8 -- so the columns are summed as two
// half-chains (0..5 into acc, 6..10 from zero), six chains in flight, joined by three packed adds.
template <int K, int KEND>
__device__ __forceinline__ void lin_mfma1_cols(const LinMatRegs &A, const v2 *__restrict__ f, v4f (&acc)[3]) {
    if constexpr (K < KEND) {
        const float fk = (K & 1) ? f[K >> 1].y : f[K >> 1].x;
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[g], fk, acc[g], 4, K, 0);
        lin_mfma1_cols<K + 1, KEND>(A, f, acc);
    }
}
template <int K>
__device__ __forceinline__ void lin_mfma1_pair(const LinMatRegs &A, const v2 *__restrict__ f, v4f (&lo)[3], v4f (&hi)[3]) {
    constexpr int H = 6;
    if constexpr (K < H) {
        const float fl = (K & 1) ? f[K >> 1].y : f[K >> 1].x;
#pragma unroll
        for (int g = 0; g < 3; ++g) lo[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[g], fl, lo[g], 4, K, 0);
        if constexpr (K + H < kLinN) {
            const float fh = ((K + H) & 1) ? f[(K + H) >> 1].y : f[(K + H) >> 1].x;
#pragma unroll
            for (int g = 0; g < 3; ++g) hi[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[g], fh, hi[g], 4, K + H, 0);
        }
        lin_mfma1_pair<K + 1>(A, f, lo, hi);
    }
}
__device__ __forceinline__ void lin_mfma1(const LinMatRegs &A, const v2 *__restrict__ f, v4f (&acc)[3]) {
    v4f hi[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    lin_mfma1_pair<0>(A, f, acc, hi);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] += hi[g];
}
// |iref_a|^2 at a state in accumulator layout (the limit function of the careful pass, s = |iref|^2 - iref_limit^2,
// at the END of a step; at its start and middle bal_f reports it)
template <int PH>
__device__ __forceinline__ float lin_iref2(const PkConst &K, const SubPk &S, const v4f (&Y)[3], float qref, float vdcref, v2 kpo) {
    const v2 ia = mk2(Y[0].x, Y[0].y);
    const v2 tq = fma2(bc2((float)PH * K.qk), ia, bc2(0.5f * PH) * S.Asw[0]);
    const float Q = __builtin_fmaf(tq.y, ia.y, tq.x * ia.x);
    const v2 iar = fma2(kpo, mk2(vdcref - Y[1].z, qref - Q), mk2(Y[1].w, Y[2].x));
    return __builtin_fmaf(iar.y, iar.y, iar.x * iar.x);
}
// state in accumulator layout (rows 0-3, 4-7, 8-11) -> the six pairs bal_f reads
__device__ __forceinline__ void lin_pairs(const v4f (&Y)[3], v2 *__restrict__ y) {
#pragma unroll
    for (int g = 0; g < 3; ++g) { y[2 * g] = mk2(Y[g].x, Y[g].y); y[2 * g + 1] = mk2(Y[g].z, Y[g].w); }
}

}  // namespace pvder
