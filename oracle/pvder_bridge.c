/* CPU ORACLE (test infrastructure only) -- C restatement of the PV-DER model + the integrators
 * the HIP kernels implement (classic RK4 micro-stepping, Dormand-Prince RK45), fp64 and fp32.
 *
 * Follows oracle/pvder_model.py (which cites the reference call sites, PVDER_env.py:166 etc.).
 * PARITY UNPINNED: see the header of oracle/pvder_model.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/_build/libpvder_bridge.so)
 */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

enum {
    PAR_PHASES = 0, PAR_RF, PAR_LF, PAR_C, PAR_KVG_RE, PAR_KVG_IM, PAR_KVI_RE, PAR_KVI_IM,
    PAR_KP_GCC, PAR_KI_GCC, PAR_KP_DC, PAR_KI_DC, PAR_KP_Q, PAR_KI_Q, PAR_WP, PAR_KP_PLL,
    PAR_KI_PLL, PAR_M_LIMIT, PAR_IREF_LIMIT, PAR_WBASE, PAR_VG_SCALE, PAR_VDCBASE, PAR_SBASE,
    PAR_NP, PAR_ISCR, PAR_IRS, PAR_PV_VT, PAR_VDC_REF0, PAR_PPV_AT_MPP, PAR_COUNT
};
#define NSMAX 23

/* ---- fp64 instance ---- */
#define REAL double
#define SUF f64
#define conj__f64_dummy 0
static inline double _Complex conj__f64(double _Complex z) { return conj(z); }
static inline double creal__f64(double _Complex z) { return creal(z); }
static inline double cimag__f64(double _Complex z) { return cimag(z); }
static inline double exp__f64(double x) { return exp(x); }
static inline double cos__f64(double x) { return cos(x); }
static inline double sin__f64(double x) { return sin(x); }
static inline double sqrt__f64(double x) { return sqrt(x); }
#include "pvder_bridge_impl.h"
#include "pvder_bridge_filippov.h"      /* event-locating / sliding-mode integrator (fp64 only) */
#undef REAL
#undef SUF

/* ---- fp32 instance ---- */
#define REAL float
#define SUF f32
static inline float _Complex conj__f32(float _Complex z) { return conjf(z); }
static inline float creal__f32(float _Complex z) { return crealf(z); }
static inline float cimag__f32(float _Complex z) { return cimagf(z); }
static inline float exp__f32(float x) { return expf(x); }
static inline float cos__f32(float x) { return cosf(x); }
static inline float sin__f32(float x) { return sinf(x); }
static inline float sqrt__f32(float x) { return sqrtf(x); }
#include "pvder_bridge_impl.h"
#undef REAL
#undef SUF

int pvder_bridge_par_count(void) { return PAR_COUNT; }

/* Batch integrate: env e advances from t0[e] by nsteps RK4 steps of size h with frozen inputs.
 * comp: [N][ns] Kahan compensation (in/out) or NULL.
 * Y is [N][ns] row-major (AoS on the host; the device uses SoA).  OpenMP over envs. */
void pvder_bridge_rk4_batch_f64(const double *par, int N, double *Y, double *comp, const double *t0, double h,
                                long nsteps, const double *vg, const double *si, const double *qref,
                                const double *vdcref) {
    const int ns = 6 * (int)par[PAR_PHASES] + 5;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < N; ++e)
        rk4_f64(par, Y + (size_t)e * ns, comp ? comp + (size_t)e * ns : NULL, t0[e], h, nsteps, vg[e], si[e], qref[e], vdcref[e]);
}

void pvder_bridge_rk4_batch_f32(const double *par, int N, float *Y, float *comp, const double *t0, double h,
                                long nsteps, const float *vg, const float *si, const float *qref,
                                const float *vdcref) {
    const int ns = 6 * (int)par[PAR_PHASES] + 5;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < N; ++e)
        rk4_f32(par, Y + (size_t)e * ns, comp ? comp + (size_t)e * ns : NULL, t0[e], h, nsteps, vg[e], si[e], qref[e], vdcref[e]);
}

/* hstep[e] in/out: natural step size carried between calls; nacc/nrej out (may be NULL) */
void pvder_bridge_rk45_batch_f64(const double *par, int N, double *Y, const double *t0, double T,
                                 double rtol, double atol, double *hstep, const double *vg,
                                 const double *si, const double *qref, const double *vdcref,
                                 long *nacc, long *nrej) {
    const int ns = 6 * (int)par[PAR_PHASES] + 5;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < N; ++e) {
        long rj = 0;
        double hl = hstep[e];
        long na = rk45_f64(par, Y + (size_t)e * ns, t0[e], T, rtol, atol, hstep[e], vg[e], si[e],
                           qref[e], vdcref[e], &rj, &hl);
        hstep[e] = hl;
        if (nacc) nacc[e] = na;
        if (nrej) nrej[e] = rj;
    }
}

/* The event-locating integrator of pvder_bridge_filippov.h.  region: [N][3] int (in/out, 0 at reset); nev out. */
void pvder_bridge_filippov_batch_f64(const double *par, int N, double *Y, const double *t0, double T,
                                     double rtol, double atol, double *hstep, const double *vg,
                                     const double *si, const double *qref, const double *vdcref,
                                     int *region, long *nacc, long *nrej, long *nev) {
    const int ns = 6 * (int)par[PAR_PHASES] + 5;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < N; ++e) {
        long rj = 0, ev = 0;
        double hl = hstep[e];
        long na = fil_integrate(par, Y + (size_t)e * ns, t0[e], T, rtol, atol, &hl, vg[e], si[e], qref[e], vdcref[e],
                                region + 3 * (size_t)e, &ev, &rj);
        hstep[e] = hl;
        if (nacc) nacc[e] = na;
        if (nrej) nrej[e] = rj;
        if (nev) nev[e] = ev;
    }
}

/* the event-locating integrator's right-hand side at one state with the given regions (0 IN, 1 OUT, 2 SLIDE per phase):
 * dy and aux[3p + 0..2] = s_p, N_p, D_p */
void pvder_bridge_filippov_eval_f64(const double *par, const double *y, double t, double vg, double si, double qref,
                                    double vdcref, const int *region, double *dy, double *aux) {
    fil_ctx c;
    c.par = par;
    memcpy(c.par_free, par, sizeof c.par_free);
    c.par_free[PAR_M_LIMIT] = 1e150;
    c.vgm = vg; c.sinsol = si; c.qref = qref; c.vdcref = vdcref;
    c.P = (int)par[PAR_PHASES];
    c.n_events = 0; c.n_evals = 0;
    for (int p = 0; p < 3; ++p) { c.region[p] = region[p]; for (int k = 0; k < 4; ++k) c.sgn0[p][k] = 0; }
    fil_rhs(&c, y, th2_of_f64(par, t), dy, aux);
}

/* dy and the 16 algebraic outputs at one state */
void pvder_bridge_eval_f64(const double *par, const double *y, double t, double vg, double si,
                           double qref, double vdcref, double *dy, double *outs) {
    rhs_f64(par, y, th2_of_f64(par, t), vg, si, qref, vdcref, dy, outs);
}
void pvder_bridge_eval_f32(const double *par, const float *y, double t, float vg, float si,
                           float qref, float vdcref, float *dy, float *outs) {
    rhs_f32(par, y, th2_of_f32(par, t), vg, si, qref, vdcref, dy, outs);
}
