/* CPU ORACLE (test infrastructure only) -- included once by pvder_bridge.c behind the fp64 instance.
 *
 * EVENT-LOCATING integration of the model with its conditional-integration anti-windup at the duty-cycle limit
 * (oracle/pvder_model.py::rhs, the `abs(m_p) > m_limit` branch; the reference reaches it through
 * PVDER_env.py:166 with Vdc_ref lowered by action 4, :220-221, :229).  There the right-hand side is DISCONTINUOUS
 * across s_p = |m_p|^2 - m_limit^2 = 0, and wherever the free field points outward and the frozen field inward the
 * solution SLIDES along s_p = 0: an adaptive Runge-Kutta scheme on the raw right-hand side only chatters around
 * that surface and its error falls like tol^0.17 (profiles/r04_saturation_convergence.txt).  This integrator
 * follows the Filippov solution of the same differential inclusion instead:
 *
 *   region IN    (s_p < 0): the free field  f
 *   region OUT   (s_p > 0): the frozen field f - G.f,  G_c = 1 for a component c of (x_p, u_p) whose derivative has
 *                the sign of c (the raw branch), and for a component that sits AT zero (it arrived there against its
 *                sign and the raw field holds it there: sgn(c) flips, the freeze engages)
 *   region SLIDE (s_p = 0): f - alpha G.f  with  alpha = N / D,  N = grad s . f,  D = grad s . (G.f):  ds/dt = 0;
 *                admissible while 0 <= N <= D  (free field outward, frozen field inward)
 *
 * with every region change LOCATED inside the Dormand-Prince step that crosses it (bisection on the step length to
 * 2^-46 of the step) and the sliding state projected back onto s_p = 0 after every step.  Phases carry their region
 * independently.  The current limit's freeze (|iref| > iref_limit, outer loops) stays in its raw form: there the
 * chatter is in slow integrators (Ki = 10) and RK45 converges (rtol 1e-8 against 1e-10: 2e-8).
 *
 * Components of (x_p, u_p) crossing zero change G across a surface the flow crosses transversally (SLIDE) or sticks
 * to (OUT): located as events too, the component is put ON zero.
 */

enum { FIL_IN = 0, FIL_OUT = 1, FIL_SLIDE = 2 };

typedef struct {
    const double *par;       /* the model's parameters */
    double par_free[PAR_COUNT]; /* ... with the duty-cycle limit out of reach: rhs_f64 then returns the free field */
    double vgm, sinsol, qref, vdcref;
    int P;
    int region[3];
    int sgn0[3][4];          /* sign a component of (xR, xI, uR, uI) must keep until its zero crossing is handled; 0 = none */
    long n_events, n_evals;
} fil_ctx;

static inline int fil_sgn(double x) { return (x > 0) - (x < 0); }

/* aux[3p + 0..2] = s_p, N_p, D_p */
static void fil_rhs(fil_ctx *c, const double *y, double th2, double *dy, double *aux) {
    const double KpG = c->par[PAR_KP_GCC], mlim = c->par[PAR_M_LIMIT];
    rhs_f64(c->par_free, y, th2, c->vgm, c->sinsol, c->qref, c->vdcref, dy, NULL);
    ++c->n_evals;
    for (int p = 0; p < c->P; ++p) {
        const double *cc = y + 6 * p + 2;          /* xR xI uR uI */
        double *f = dy + 6 * p + 2;
        const double mr = KpG * cc[2] + cc[0], mi = KpG * cc[3] + cc[1];
        const double g[4] = {2 * mr, 2 * mi, 2 * KpG * mr, 2 * KpG * mi};
        int G[4];
        double N = 0, D = 0;
        for (int k = 0; k < 4; ++k) {
            G[k] = (cc[k] == 0.0) || (fil_sgn(f[k]) == fil_sgn(cc[k]));
            N += g[k] * f[k];
            if (G[k]) D += g[k] * f[k];
        }
        aux[3 * p] = mr * mr + mi * mi - mlim * mlim;
        aux[3 * p + 1] = N;
        aux[3 * p + 2] = D;
        if (c->region[p] == FIL_OUT) {
            for (int k = 0; k < 4; ++k) if (G[k]) f[k] = 0;
        } else if (c->region[p] == FIL_SLIDE) {
            double al = D > 0 ? N / D : 0.0;
            al = al < 0 ? 0 : (al > 1 ? 1 : al);
            for (int k = 0; k < 4; ++k) if (G[k]) f[k] *= (1.0 - al);
        }
    }
}

/* region of phase p from the state alone (aux of fil_rhs at that state) */
static int fil_decide(const double *aux, int p) {
    const double s = aux[3 * p], N = aux[3 * p + 1], D = aux[3 * p + 2];
    const double eps = 1e-11;
    if (s > eps) return FIL_OUT;
    if (s < -eps) return FIL_IN;
    if (N <= 0) return FIL_IN;
    if (N - D >= 0) return FIL_OUT;
    return FIL_SLIDE;
}

/* 1 if the state violates what its regions assume */
static int fil_violated(const fil_ctx *c, const double *y, const double *aux) {
    for (int p = 0; p < c->P; ++p) {
        const double s = aux[3 * p], N = aux[3 * p + 1], D = aux[3 * p + 2];
        /* (|m|^2 resolves 2e-16: a state parked just outside the limit must not read as inside) */
        if (c->region[p] == FIL_IN) { if (s > 1e-13) return 1; continue; }
        if (c->region[p] == FIL_OUT) { if (s < -1e-13) return 1; }
        else if (N < 0 || D - N < 0) return 1;
        for (int k = 0; k < 4; ++k)
            if (c->sgn0[p][k] != 0 && y[6 * p + 2 + k] * c->sgn0[p][k] < 0) return 1;
    }
    return 0;
}

/* phase p back onto s_p = 0 along grad s (two Newton steps: s is quadratic) */
static void fil_project_phase(const fil_ctx *c, double *y, int p) {
    const double KpG = c->par[PAR_KP_GCC], mlim = c->par[PAR_M_LIMIT];
    double *cc = y + 6 * p + 2;
    for (int it = 0; it < 2; ++it) {
        const double mr = KpG * cc[2] + cc[0], mi = KpG * cc[3] + cc[1];
        const double s = mr * mr + mi * mi - mlim * mlim;
        const double g[4] = {2 * mr, 2 * mi, 2 * KpG * mr, 2 * KpG * mi};
        double gg = 0;
        for (int k = 0; k < 4; ++k) if (cc[k] != 0.0) gg += g[k] * g[k];
        if (gg <= 0) break;
        for (int k = 0; k < 4; ++k) if (cc[k] != 0.0) cc[k] -= s * g[k] / gg;
    }
}

static void fil_project(const fil_ctx *c, double *y) {
    for (int p = 0; p < c->P; ++p)
        if (c->region[p] == FIL_SLIDE) fil_project_phase(c, y, p);
}

/* settle the discrete state at a switching point: crossed components onto zero, regions from the fields, signs */
static void fil_settle(fil_ctx *c, double *y, double th2) {
    double dy[NSMAX], aux[9];
    for (int p = 0; p < c->P; ++p)
        for (int k = 0; k < 4; ++k)
            if (c->sgn0[p][k] != 0 && y[6 * p + 2 + k] * c->sgn0[p][k] < 0) y[6 * p + 2 + k] = 0.0;
    fil_rhs(c, y, th2, dy, aux);                     /* (N, D, s do not depend on the regions) */
    /* a phase within the decision band of its limit goes ONTO it (a move of <= 1e-11 in |m|^2): whatever region the
     * fields then give it, it starts there admissibly (balanced phases reach the limit together to 1e-12) */
    int moved = 0;
    for (int p = 0; p < c->P; ++p)
        if (fabs(aux[3 * p]) <= 1e-11) { fil_project_phase(c, y, p); moved = 1; }
    if (moved) fil_rhs(c, y, th2, dy, aux);
    for (int p = 0; p < c->P; ++p) c->region[p] = fil_decide(aux, p);
    for (int p = 0; p < c->P; ++p)
        for (int k = 0; k < 4; ++k)
            c->sgn0[p][k] = c->region[p] == FIL_IN ? 0 : fil_sgn(y[6 * p + 2 + k]);
}

/* one Dormand-Prince step of size h from (t, y) with k1 = f(y); yn, k7 = f(yn), aux7 out; returns the error norm */
static double fil_dp_step(fil_ctx *c, int n, double t0t, const double *y, const double *k1, double h, double rtol,
                          double atol, double *yn, double *k7, double *aux7) {
    static const double c2 = 1. / 5, c3 = 3. / 10, c4 = 4. / 5, c5 = 8. / 9;
    static const double a21 = 1. / 5, a31 = 3. / 40, a32 = 9. / 40, a41 = 44. / 45, a42 = -56. / 15,
                        a43 = 32. / 9, a51 = 19372. / 6561, a52 = -25360. / 2187, a53 = 64448. / 6561,
                        a54 = -212. / 729, a61 = 9017. / 3168, a62 = -355. / 33, a63 = 46732. / 5247,
                        a64 = 49. / 176, a65 = -5103. / 18656, b1 = 35. / 384, b3 = 500. / 1113,
                        b4 = 125. / 192, b5 = -2187. / 6784, b6 = 11. / 84;
    static const double e1 = 71. / 57600, e3 = -71. / 16695, e4 = 71. / 1920, e5 = -17253. / 339200,
                        e6 = 22. / 525, e7 = -1. / 40;
    double k2[NSMAX], k3[NSMAX], k4[NSMAX], k5[NSMAX], k6[NSMAX], yt[NSMAX] = {0}, aux[9];
    for (int k = 0; k < n; ++k) yt[k] = y[k] + h * a21 * k1[k];
    fil_rhs(c, yt, th2_of_f64(c->par, t0t + c2 * h), k2, aux);
    for (int k = 0; k < n; ++k) yt[k] = y[k] + h * (a31 * k1[k] + a32 * k2[k]);
    fil_rhs(c, yt, th2_of_f64(c->par, t0t + c3 * h), k3, aux);
    for (int k = 0; k < n; ++k) yt[k] = y[k] + h * (a41 * k1[k] + a42 * k2[k] + a43 * k3[k]);
    fil_rhs(c, yt, th2_of_f64(c->par, t0t + c4 * h), k4, aux);
    for (int k = 0; k < n; ++k) yt[k] = y[k] + h * (a51 * k1[k] + a52 * k2[k] + a53 * k3[k] + a54 * k4[k]);
    fil_rhs(c, yt, th2_of_f64(c->par, t0t + c5 * h), k5, aux);
    for (int k = 0; k < n; ++k) yt[k] = y[k] + h * (a61 * k1[k] + a62 * k2[k] + a63 * k3[k] + a64 * k4[k] + a65 * k5[k]);
    fil_rhs(c, yt, th2_of_f64(c->par, t0t + h), k6, aux);
    for (int k = 0; k < n; ++k) yn[k] = y[k] + h * (b1 * k1[k] + b3 * k3[k] + b4 * k4[k] + b5 * k5[k] + b6 * k6[k]);
    fil_rhs(c, yn, th2_of_f64(c->par, t0t + h), k7, aux7);
    double err = 0.0;
    for (int k = 0; k < n; ++k) {
        const double ek = h * (e1 * k1[k] + e3 * k3[k] + e4 * k4[k] + e5 * k5[k] + e6 * k6[k] + e7 * k7[k]);
        const double ay = fabs(y[k]) > fabs(yn[k]) ? fabs(y[k]) : fabs(yn[k]);
        const double sc = atol + rtol * ay;
        err += (ek / sc) * (ek / sc);
    }
    return sqrt(err / n);
}

/* integrates exactly [t0, t0 + T]; region[] in/out (carried between calls); returns #accepted steps */
static long fil_integrate(const double *par, double *y, double t0, double T, double rtol, double atol, double *hstep,
                          double vgm, double sinsol, double qref, double vdcref, int *region, long *nev, long *nrej) {
    fil_ctx c;
    const int fil_debug = getenv("FIL_DEBUG") != NULL;
    c.par = par;
    memcpy(c.par_free, par, sizeof c.par_free);
    c.par_free[PAR_M_LIMIT] = 1e150;
    c.vgm = vgm; c.sinsol = sinsol; c.qref = qref; c.vdcref = vdcref;
    c.P = (int)par[PAR_PHASES];
    c.n_events = 0; c.n_evals = 0;
    const int n = 6 * c.P + 5;
    for (int p = 0; p < 3; ++p) { c.region[p] = p < c.P ? region[p] : FIL_IN; for (int k = 0; k < 4; ++k) c.sgn0[p][k] = 0; }
    double k1[NSMAX], k7[NSMAX], yn[NSMAX], aux[9], aux7[9];
    double t = 0.0, hn = *hstep;
    long nacc = 0;
    *nrej = 0;
    fil_settle(&c, y, th2_of_f64(par, t0));          /* inputs may have jumped since the last call */
    fil_rhs(&c, y, th2_of_f64(par, t0), k1, aux);
    while (t < T * (1.0 - 1e-14)) {
        int last = 0;
        double h = hn;
        if (t + h >= T) { h = T - t; last = 1; }
        double err = fil_dp_step(&c, n, t0 + t, y, k1, h, rtol, atol, yn, k7, aux7);
        if (!(err <= 1.0) && h > 1e-13) {
            ++*nrej;
            double fac = 0.9 * pow(err, -0.2);
            if (!(fac > 0.2)) fac = 0.2;
            hn = h * fac;
            continue;
        }
        if (fil_violated(&c, yn, aux7)) {
            /* a region boundary inside the step: the longest sub-step that stays admissible, then one just beyond it */
            double lo = 0.0, hi = 1.0;
            for (int it = 0; it < 46; ++it) {
                const double mid = 0.5 * (lo + hi);
                fil_dp_step(&c, n, t0 + t, y, k1, h * mid, rtol, atol, yn, k7, aux7);
                if (fil_violated(&c, yn, aux7)) hi = mid; else lo = mid;
            }
            h *= hi;
            fil_dp_step(&c, n, t0 + t, y, k1, h, rtol, atol, yn, k7, aux7);
            t += h;
            for (int k = 0; k < n; ++k) y[k] = yn[k];
            fil_settle(&c, y, th2_of_f64(par, t0 + t));
            fil_rhs(&c, y, th2_of_f64(par, t0 + t), k1, aux);
            ++c.n_events; ++nacc;
            if (fil_debug) fprintf(stderr, "event t %.12e  regions %d%d%d  s %.2e %.2e %.2e  N %.3e %.3e %.3e  D %.3e %.3e %.3e\n", t0 + t,
                                   c.region[0], c.region[1], c.region[2], aux[0], aux[3], aux[6], aux[1], aux[4], aux[7], aux[2], aux[5], aux[8]);
            if (hn > 1e-6) hn = 1e-6;
            continue;
        }
        t += h;
        for (int k = 0; k < n; ++k) y[k] = yn[k];
        int slid = 0;
        for (int p = 0; p < c.P; ++p) slid |= c.region[p] == FIL_SLIDE;
        if (slid) {
            fil_project(&c, y);
            fil_rhs(&c, y, th2_of_f64(par, t0 + t), k1, aux);
        } else {
            for (int k = 0; k < n; ++k) k1[k] = k7[k];
        }
        ++nacc;
        if (last) break;
        double fac = 0.9 * pow(err > 1e-30 ? err : 1e-30, -0.2);
        if (fac > 5.0) fac = 5.0;
        if (fac < 0.2) fac = 0.2;
        hn = h * fac;
    }
    for (int p = 0; p < c.P; ++p) region[p] = c.region[p];
    *hstep = hn;
    *nev = c.n_events;
    return nacc;
}
