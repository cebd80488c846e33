/* CPU ORACLE (test infrastructure only) -- included twice by pvder_bridge.c with
 * REAL = double / float and SUF = f64 / f32.
 *
 * Independent C99-complex restatement of oracle/pvder_model.py::rhs (itself a restatement of the
 * published pvder model the reference calls at PVDER_env.py:166), in the "relative PLL angle"
 * form the fp32 kernels need:  state[last] = delta = wte - wbase*t  (SURVEY.md 7.2 hard part 3).
 *
 *   vd = Re(V1 e^{-j delta}) + Re(conj(V2) e^{-j(2 theta + delta)}),   theta = wbase t,
 *   V1 = (va + a vb + a^2 vc)/3,  V2 = (va + a^2 vb + a vc)/3,  a = e^{j 2pi/3}
 *
 * which is algebraically identical to pvder's phasor -> time-domain -> Park chain.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef REAL _Complex FN(cplx);
#define CPLX FN(cplx)

static inline int FN(sgn)(REAL x) { return (x > 0) - (x < 0); }

/* algebraic + derivative; th2 = 2*wbase*t (already reduced mod 2pi) */
static void FN(rhs)(const double *par, const REAL *y, REAL th2, REAL vgm, REAL sinsol, REAL qref,
                    REAL vdcref, REAL *dy, REAL *outs /* may be NULL: 16 algebraic outputs */) {
    const int P = (int)par[PAR_PHASES];
    const REAL Rf = (REAL)par[PAR_RF], iLf = (REAL)(1.0 / par[PAR_LF]), C = (REAL)par[PAR_C];
    const CPLX kvg = (REAL)par[PAR_KVG_RE] + I * (REAL)par[PAR_KVG_IM];
    const CPLX kvi = (REAL)par[PAR_KVI_RE] + I * (REAL)par[PAR_KVI_IM];
    const REAL KpG = (REAL)par[PAR_KP_GCC], KiG = (REAL)par[PAR_KI_GCC];
    const REAL KpDC = (REAL)par[PAR_KP_DC], KiDC = (REAL)par[PAR_KI_DC];
    const REAL KpQ = (REAL)par[PAR_KP_Q], KiQ = (REAL)par[PAR_KI_Q], wp = (REAL)par[PAR_WP];
    const REAL KpP = (REAL)par[PAR_KP_PLL], KiP = (REAL)par[PAR_KI_PLL];
    const REAL mlim = (REAL)par[PAR_M_LIMIT], ilim = (REAL)par[PAR_IREF_LIMIT];
    const REAL wbase = (REAL)par[PAR_WBASE];
    const CPLX a120 = (REAL)-0.5 + I * (REAL)0.86602540378443864676;
    const CPLX a240 = (REAL)-0.5 - I * (REAL)0.86602540378443864676;
    const CPLX rot[3] = {1, a240, a120};
    const int b = 6 * P;
    const REAL Vdc = y[b], xDC = y[b + 1], xQ = y[b + 2], xPLL = y[b + 3], delta = y[b + 4];
    CPLX i[3], x[3], u[3], m[3], vt[3], v[3];
    CPLX S = 0, Sp = 0;
    REAL v2sum = 0;
    const CPLX vga = vgm * (REAL)par[PAR_VG_SCALE];
    for (int p = 0; p < P; ++p) {
        i[p] = y[6 * p] + I * y[6 * p + 1];
        x[p] = y[6 * p + 2] + I * y[6 * p + 3];
        u[p] = y[6 * p + 4] + I * y[6 * p + 5];
        m[p] = KpG * u[p] + x[p];
        vt[p] = m[p] * (Vdc / 2);
        v[p] = kvg * (vga * rot[p]) + kvi * i[p];
        S += vt[p] * FN(conj_)(i[p]);
        Sp += v[p] * FN(conj_)(i[p]);
        v2sum += FN(creal_)(v[p]) * FN(creal_)(v[p]) + FN(cimag_)(v[p]) * FN(cimag_)(v[p]);
    }
    S *= (REAL)0.5;
    Sp *= (REAL)0.5;
    /* PV array */
    /* model option U5 (oracle/pvder_model.py::ModelOptions.ppv_uses_vdc = False): Ppv at the MPP voltage */
    const REAL Vpv = par[PAR_PPV_AT_MPP] != 0.0 ? (REAL)par[PAR_VDC_REF0] : Vdc;
    const REAL Vact = Vpv * (REAL)par[PAR_VDCBASE];
    const REAL Iph = (REAL)par[PAR_ISCR] * (sinsol / 100);
    REAL ppv = (REAL)par[PAR_NP] * Iph * Vact -
               (REAL)par[PAR_NP] * (REAL)par[PAR_IRS] * Vact * (FN(exp_)((REAL)par[PAR_PV_VT] * Vpv) - 1);
    ppv = (ppv > 0 ? ppv : 0) / (REAL)par[PAR_SBASE];
    /* current references */
    const REAL Qpcc = FN(cimag_)(Sp);
    const CPLX iaref = (xDC + KpDC * (vdcref - Vdc)) + I * (xQ - KpQ * (qref - Qpcc));
    /* PLL */
    REAL vd;
    if (P == 3) {
        const CPLX V1 = (v[0] + a120 * v[1] + a240 * v[2]) / 3;
        const CPLX V2 = (v[0] + a240 * v[1] + a120 * v[2]) / 3;
        const CPLX e1 = FN(cos_)(delta) - I * FN(sin_)(delta);
        const CPLX e2 = FN(cos_)(th2 + delta) - I * FN(sin_)(th2 + delta);
        vd = FN(creal_)(V1 * e1) + FN(creal_)(FN(conj_)(V2) * e2);
    } else {
        const CPLX e1 = FN(cos_)(delta) - I * FN(sin_)(delta);
        vd = FN(creal_)(v[0] * e1);
    }
    const REAL dw = KpP * vd + xPLL; /* we - wbase */
    const REAL wr = (wbase + dw) / wbase;
    for (int p = 0; p < P; ++p) {
        const CPLX irefp = iaref * rot[p];
        CPLX di = iLf * (-Rf * i[p] - v[p] + vt[p]) - I * wr * i[p];
        CPLX dx = KiG * u[p];
        CPLX du = wp * (-u[p] + irefp - i[p]);
        const REAL mm = FN(creal_)(m[p]) * FN(creal_)(m[p]) + FN(cimag_)(m[p]) * FN(cimag_)(m[p]);
        if (mm > mlim * mlim) {
            REAL dxr = FN(creal_)(dx), dxi = FN(cimag_)(dx), dur = FN(creal_)(du), dui = FN(cimag_)(du);
            if (FN(sgn)(dxr) == FN(sgn)(FN(creal_)(x[p]))) dxr = 0;
            if (FN(sgn)(dxi) == FN(sgn)(FN(cimag_)(x[p]))) dxi = 0;
            if (FN(sgn)(dur) == FN(sgn)(FN(creal_)(u[p]))) dur = 0;
            if (FN(sgn)(dui) == FN(sgn)(FN(cimag_)(u[p]))) dui = 0;
            dx = dxr + I * dxi;
            du = dur + I * dui;
        }
        dy[6 * p] = FN(creal_)(di);
        dy[6 * p + 1] = FN(cimag_)(di);
        dy[6 * p + 2] = FN(creal_)(dx);
        dy[6 * p + 3] = FN(cimag_)(dx);
        dy[6 * p + 4] = FN(creal_)(du);
        dy[6 * p + 5] = FN(cimag_)(du);
    }
    dy[b] = (ppv - FN(creal_)(S)) / (Vdc * C);
    REAL dxDC = KiDC * (vdcref - Vdc);
    REAL dxQ = -KiQ * (qref - Qpcc);
    const REAL ir2 = FN(creal_)(iaref) * FN(creal_)(iaref) + FN(cimag_)(iaref) * FN(cimag_)(iaref);
    if (ir2 > ilim * ilim) {
        if (FN(sgn)(dxDC) == FN(sgn)(xDC)) dxDC = 0;
        if (FN(sgn)(dxQ) == FN(sgn)(xQ)) dxQ = 0;
    }
    dy[b + 1] = dxDC;
    dy[b + 2] = dxQ;
    dy[b + 3] = KiP * vd;
    dy[b + 4] = dw;
    if (outs) {
        outs[0] = FN(creal_)(i[0]); outs[1] = FN(cimag_)(i[0]);
        outs[2] = FN(creal_)(v[0]); outs[3] = FN(cimag_)(v[0]);
        outs[4] = FN(creal_)(Sp);   outs[5] = FN(cimag_)(Sp);
        outs[6] = Vdc;              outs[7] = ppv;
        outs[8] = FN(sqrt_)(v2sum / P) / (REAL)1.41421356237309504880;  /* Vrms */
        outs[9] = FN(creal_)(S);    outs[10] = FN(cimag_)(S);
        outs[11] = FN(creal_)(m[0]); outs[12] = FN(cimag_)(m[0]);
        outs[13] = vd;              outs[14] = wbase + dw;
        outs[15] = FN(sqrt_)(ir2);
    }
}

static inline REAL FN(th2_of)(const double *par, double t) {
    /* 2*wbase*t mod 2pi, reduced in double (exact enough for fp32 callers) */
    double c = 2.0 * par[PAR_WBASE] * t / 6.283185307179586476925;
    c -= floor(c);
    return (REAL)(c * 6.283185307179586476925);
}

/* classic RK4, nsteps of size h from time t0 (inputs frozen) */
/* comp: Kahan compensation per state (in/out, may be NULL = plain summation).  With h ~ 4 us
 * the fp32 increments h*f fall below ulp(y)/2 and plain summation stagnates. */
static void FN(rk4)(const double *par, REAL *y, REAL *comp, double t0, double h, long nsteps, REAL vgm,
                    REAL sinsol, REAL qref, REAL vdcref) {
    const int n = 6 * (int)par[PAR_PHASES] + 5;
    REAL k1[NSMAX], k2[NSMAX], k3[NSMAX], k4[NSMAX], yt[NSMAX];
    const REAL hh = (REAL)h;
    for (long s = 0; s < nsteps; ++s) {
        const double t = t0 + (double)s * h;
        FN(rhs)(par, y, FN(th2_of)(par, t), vgm, sinsol, qref, vdcref, k1, NULL);
        for (int k = 0; k < n; ++k) yt[k] = y[k] + (hh / 2) * k1[k];
        FN(rhs)(par, yt, FN(th2_of)(par, t + h / 2), vgm, sinsol, qref, vdcref, k2, NULL);
        for (int k = 0; k < n; ++k) yt[k] = y[k] + (hh / 2) * k2[k];
        FN(rhs)(par, yt, FN(th2_of)(par, t + h / 2), vgm, sinsol, qref, vdcref, k3, NULL);
        for (int k = 0; k < n; ++k) yt[k] = y[k] + hh * k3[k];
        FN(rhs)(par, yt, FN(th2_of)(par, t + h), vgm, sinsol, qref, vdcref, k4, NULL);
        if (!comp) {
            for (int k = 0; k < n; ++k) y[k] += (hh / 6) * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]);
        } else {
            for (int k = 0; k < n; ++k) {
                const REAL inc = (hh / 6) * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) - comp[k];
                volatile REAL tsum = y[k] + inc;          /* volatile: no re-association */
                volatile REAL dif = tsum - y[k];
                comp[k] = dif - inc;
                y[k] = tsum;
            }
        }
    }
}

/* Dormand-Prince 5(4), adaptive, integrates exactly [t0, t0+T]; returns #accepted, *nrej */
static long FN(rk45)(const double *par, REAL *y, double t0, double T, double rtol, double atol,
                     double h0, REAL vgm, REAL sinsol, REAL qref, REAL vdcref, long *nrej,
                     double *hlast) {
    const int n = 6 * (int)par[PAR_PHASES] + 5;
    static const double c2 = 1. / 5, c3 = 3. / 10, c4 = 4. / 5, c5 = 8. / 9;
    static const double a21 = 1. / 5, a31 = 3. / 40, a32 = 9. / 40, a41 = 44. / 45, a42 = -56. / 15,
                        a43 = 32. / 9, a51 = 19372. / 6561, a52 = -25360. / 2187, a53 = 64448. / 6561,
                        a54 = -212. / 729, a61 = 9017. / 3168, a62 = -355. / 33, a63 = 46732. / 5247,
                        a64 = 49. / 176, a65 = -5103. / 18656, b1 = 35. / 384, b3 = 500. / 1113,
                        b4 = 125. / 192, b5 = -2187. / 6784, b6 = 11. / 84;
    static const double e1 = 71. / 57600, e3 = -71. / 16695, e4 = 71. / 1920, e5 = -17253. / 339200,
                        e6 = 22. / 525, e7 = -1. / 40;
    REAL k1[NSMAX], k2[NSMAX], k3[NSMAX], k4[NSMAX], k5[NSMAX], k6[NSMAX], k7[NSMAX], yt[NSMAX],
        yn[NSMAX];
    double t = 0.0, h, hn = h0;     /* hn = controller's natural step, h = step actually taken */
    long nacc = 0;
    *nrej = 0;
    FN(rhs)(par, y, FN(th2_of)(par, t0), vgm, sinsol, qref, vdcref, k1, NULL);
    while (t < T * (1.0 - 1e-14)) {
        int last = 0;
        h = hn;
        if (t + h >= T) { h = T - t; last = 1; }
        const REAL hh = (REAL)h;
        for (int k = 0; k < n; ++k) yt[k] = y[k] + hh * (REAL)a21 * k1[k];
        FN(rhs)(par, yt, FN(th2_of)(par, t0 + t + c2 * h), vgm, sinsol, qref, vdcref, k2, NULL);
        for (int k = 0; k < n; ++k) yt[k] = y[k] + hh * ((REAL)a31 * k1[k] + (REAL)a32 * k2[k]);
        FN(rhs)(par, yt, FN(th2_of)(par, t0 + t + c3 * h), vgm, sinsol, qref, vdcref, k3, NULL);
        for (int k = 0; k < n; ++k)
            yt[k] = y[k] + hh * ((REAL)a41 * k1[k] + (REAL)a42 * k2[k] + (REAL)a43 * k3[k]);
        FN(rhs)(par, yt, FN(th2_of)(par, t0 + t + c4 * h), vgm, sinsol, qref, vdcref, k4, NULL);
        for (int k = 0; k < n; ++k)
            yt[k] = y[k] + hh * ((REAL)a51 * k1[k] + (REAL)a52 * k2[k] + (REAL)a53 * k3[k] + (REAL)a54 * k4[k]);
        FN(rhs)(par, yt, FN(th2_of)(par, t0 + t + c5 * h), vgm, sinsol, qref, vdcref, k5, NULL);
        for (int k = 0; k < n; ++k)
            yt[k] = y[k] + hh * ((REAL)a61 * k1[k] + (REAL)a62 * k2[k] + (REAL)a63 * k3[k] +
                                 (REAL)a64 * k4[k] + (REAL)a65 * k5[k]);
        FN(rhs)(par, yt, FN(th2_of)(par, t0 + t + h), vgm, sinsol, qref, vdcref, k6, NULL);
        for (int k = 0; k < n; ++k)
            yn[k] = y[k] + hh * ((REAL)b1 * k1[k] + (REAL)b3 * k3[k] + (REAL)b4 * k4[k] +
                                 (REAL)b5 * k5[k] + (REAL)b6 * k6[k]);
        FN(rhs)(par, yn, FN(th2_of)(par, t0 + t + h), vgm, sinsol, qref, vdcref, k7, NULL);
        double err = 0.0;
        for (int k = 0; k < n; ++k) {
            const double ek = h * (e1 * k1[k] + e3 * k3[k] + e4 * k4[k] + e5 * k5[k] + e6 * k6[k] + e7 * k7[k]);
            const double ay = fabs((double)y[k]) > fabs((double)yn[k]) ? fabs((double)y[k]) : fabs((double)yn[k]);
            const double sc = atol + rtol * ay;
            err += (ek / sc) * (ek / sc);
        }
        err = sqrt(err / n);
        if (err <= 1.0 || h < 1e-12) {
            t += h;
            for (int k = 0; k < n; ++k) { y[k] = yn[k]; k1[k] = k7[k]; }   /* FSAL */
            ++nacc;
            if (last) break;
        } else {
            ++*nrej;
        }
        double fac = 0.9 * pow(err > 1e-30 ? err : 1e-30, -0.2);
        if (fac > 5.0) fac = 5.0;
        if (fac < 0.2) fac = 0.2;
        if (err > 1.0) hn = h * fac;                 /* rejected: shrink from the step tried */
        else if (!last || h * fac < hn) hn = h * fac; /* accepted: a clipped last step never grows hn */
    }
    if (hlast) *hlast = hn;
    return nacc;
}

#undef CPLX
