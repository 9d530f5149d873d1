/*
 * lsf_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement of the hot path of musheen/LevelSetFortran (WENO5 Hamilton-Jacobi
 * reinitialisation + min/max-flow smoothing), written from the behaviour of the reference and
 * citing the lines it follows.  It exists so that the HIP kernels in levelsetfortran_amd/csrc can
 * be checked on machines where the reference sources are absent.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it (as the checker, never as
 * the thing shipped or measured as the product).
 *
 * PINNING: this file is compared bit-for-bit against the reference itself, compiled here from
 * /root/reference by `make -C oracle ref` (amdflang -O3 -fdefault-real-8) -- see
 * tests/golden/make_golden.py, which ran both and committed the reference's outputs as fixtures,
 * and tests/test_oracle_golden.py, which re-checks the oracle against those fixtures everywhere.
 *
 * The reference is compiled with -fdefault-real-8 (Makefile:4): every REAL and every real literal
 * is an IEEE double, INTEGER is 32-bit.  Build this file with -ffp-contract=off: the reference
 * object code contains no FMA and evaluates each expression left to right as written.
 */
#include "lsf_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, k) ((size_t)(i) + sx * ((size_t)(j) + sy * (size_t)(k)))

/* Fortran MAX/MIN as flang lowers them for two REAL arguments (compare + select). */
static inline double fmax2(double a, double b) { return (a > b) ? a : b; }
static inline double fmin2(double a, double b) { return (a < b) ? a : b; }

/*
 * One axis of the WENO branch, subs.f90:509-552 (x), :555-598 (y), :601-644 (z).
 * q[0..6] = phi at offsets -3..+3 along the axis.  yquirk reproduces subs.f90:576, where the y
 * direction computes p5 from phi(i,j+3,k)-phi(i,j+3,k).
 * Outputs: *dm = D^- (a/c/e), *dp = D^+ (b/d/f).
 */
static inline void weno_axis(const double q[7], double dx, int yquirk, double *dm, double *dp)
{
    const double m3 = q[0], m2 = q[1], m1 = q[2], c0 = q[3], p1_ = q[4], p2_ = q[5], p3_ = q[6];
    double ap, am, bp, bm, cp, cm, dpp, dmm;
    double IS0p, IS0m, IS1p, IS1m, IS2p, IS2m;
    double p0, p1, p2, p3, p4, p5, epsp, epsm;
    double a0p, a0m, a1p, a1m, a2p, a2m, w0p, w0m, w2p, w2m, PWp, PWm;

    ap = (p3_ - 2. * p2_ + p1_) / dx;  /* :509 */
    am = (m3 - 2. * m2 + m1) / dx;     /* :510 */
    bp = (p2_ - 2. * p1_ + c0) / dx;   /* :511 */
    bm = (m2 - 2. * m1 + c0) / dx;     /* :512 */
    cp = (p1_ - 2. * c0 + m1) / dx;    /* :513 */
    cm = cp;                           /* :514 */
    dpp = bm;                          /* :515 */
    dmm = bp;                          /* :516 */

    IS0p = 13. * (ap - bp) * (ap - bp) + 3. * (ap - 3. * bp) * (ap - 3. * bp);     /* :518 */
    IS0m = 13. * (am - bm) * (am - bm) + 3. * (am - 3. * bm) * (am - 3. * bm);     /* :519 */
    IS1p = 13. * (bp - cp) * (bp - cp) + 3. * (bp + cp) * (bp + cp);               /* :520 */
    IS1m = 13. * (bm - cm) * (bm - cm) + 3. * (bm + cm) * (bm + cm);               /* :521 */
    IS2p = 13. * (cp - dpp) * (cp - dpp) + 3. * (3. * cp - dpp) * (3. * cp - dpp); /* :522 */
    IS2m = 13. * (cm - dmm) * (cm - dmm) + 3. * (3. * cm - dmm) * (3. * cm - dmm); /* :523 */

    p0 = (m2 - m3) / dx;   /* :525 */
    p1 = (m1 - m2) / dx;   /* :526 */
    p2 = (c0 - m1) / dx;   /* :527 */
    p3 = (p1_ - c0) / dx;  /* :528 */
    p4 = (p2_ - p1_) / dx; /* :529 */
    if (yquirk)
        p5 = (p3_ - p3_) / dx; /* :576 */
    else
        p5 = (p3_ - p2_) / dx; /* :530 */

    /* :533-534 */
    epsp = (1.E-6) * fmax2(p1 * p1, fmax2(p2 * p2, fmax2(p3 * p3, fmax2(p4 * p4, p5 * p5)))) + 1.E-99;
    epsm = (1.E-6) * fmax2(p0 * p0, fmax2(p1 * p1, fmax2(p2 * p2, fmax2(p3 * p3, p4 * p4)))) + 1.E-99;

    a0p = 1. / ((epsp + IS0p) * (epsp + IS0p)); /* :536 */
    a0m = 1. / ((epsm + IS0m) * (epsm + IS0m));
    a1p = 6. / ((epsp + IS1p) * (epsp + IS1p));
    a1m = 6. / ((epsm + IS1m) * (epsm + IS1m));
    a2p = 3. / ((epsp + IS2p) * (epsp + IS2p));
    a2m = 3. / ((epsm + IS2m) * (epsm + IS2m)); /* :541 */

    w0p = a0p / (a0p + a1p + a2p); /* :543 */
    w0m = a0m / (a0m + a1m + a2m);
    w2p = a2p / (a0p + a1p + a2p);
    w2m = a2m / (a0m + a1m + a2m); /* :546 */

    PWp = 1. / 3. * w0p * (ap - 2. * bp + cp) + 1. / 6. * (w2p - 0.5) * (bp - 2. * cp + dpp); /* :548 */
    PWm = 1. / 3. * w0m * (am - 2. * bm + cm) + 1. / 6. * (w2m - 0.5) * (bm - 2. * cm + dmm); /* :549 */

    *dm = 1. / 12. * (-p1 + 7. * p2 + 7. * p3 - p4) - PWm; /* :551 */
    *dp = 1. / 12. * (-p1 + 7. * p2 + 7. * p3 - p4) + PWp; /* :552 */
}

/* Godunov switch and magnitude, subs.f90:667-702, from the six one-sided derivatives. */
static inline double godunov(double phic, double a, double b, double c, double d, double e, double f)
{
    double pa = fmax2(a, 0.), pb = fmax2(b, 0.), pc = fmax2(c, 0.);
    double pd = fmax2(d, 0.), pe = fmax2(e, 0.), pf = fmax2(f, 0.);
    double na = fmin2(a, 0.), nb = fmin2(b, 0.), nc = fmin2(c, 0.);
    double nd = fmin2(d, 0.), ne = fmin2(e, 0.), nf = fmin2(f, 0.);
    double gradX, gradY, gradZ;
    if (phic > 0.) { /* :684 */
        gradX = fmax2(pa * pa, nb * nb);
        gradY = fmax2(pc * pc, nd * nd);
        gradZ = fmax2(pe * pe, nf * nf);
    } else {
        gradX = fmax2(pb * pb, na * na);
        gradY = fmax2(pd * pd, nc * nc);
        gradZ = fmax2(pf * pf, ne * ne);
    }
    return sqrt(gradX + gradY + gradZ); /* :702 */
}

/* weno (subs.f90:489-711) for the cell with GLOBAL index (i,j,k) stored at p[0] in an array with
 * strides (1, s1, s2); returns gM.  The branch test :506 is in global indices. */
static double weno_at(const double *p, size_t s1, size_t s2, int i, int j, int k, int nx, int ny, int nz,
                      double dx)
{
    const ptrdiff_t t1 = (ptrdiff_t)s1, t2 = (ptrdiff_t)s2;
    double a, b, c, d, e, f;
    /* :506 */
    if ((i > 3) && (i < nx - 4) && (j > 3) && (j < ny - 4) && (k > 3) && (k < nz - 4)) {
        double q[7];
        int m;
        for (m = -3; m <= 3; ++m) q[m + 3] = p[m];
        weno_axis(q, dx, 0, &a, &b);
        for (m = -3; m <= 3; ++m) q[m + 3] = p[m * t1];
        weno_axis(q, dx, 1, &c, &d);
        for (m = -3; m <= 3; ++m) q[m + 3] = p[m * t2];
        weno_axis(q, dx, 0, &e, &f);
    } else {
        /* :657-662 */
        a = (p[0] - p[-1]) / dx;
        b = (p[1] - p[0]) / dx;
        c = (p[0] - p[-t1]) / dx;
        d = (p[t1] - p[0]) / dx;
        e = (p[0] - p[-t2]) / dx;
        f = (p[t2] - p[0]) / dx;
    }
    return godunov(p[0], a, b, c, d, e, f);
}

static double weno_cell(int i, int j, int k, int nx, int ny, int nz, double dx, const double *phi)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    return weno_at(&phi[IDX(i, j, k)], sx, sx * sy, i, j, k, nx, ny, nz, dx);
}

double lsf_oracle_weno(int i, int j, int k, int nx, int ny, int nz, double dx, const double *phi)
{
    return weno_cell(i, j, k, nx, ny, nz, dx, phi);
}

double lsf_oracle_phisign(double pS, double dxx, double gM)
{
    return pS / sqrt(pS * pS + dxx * dxx * gM); /* subs.f90:169 */
}

/* subs.f90:747-750: the cell update.  src is the array weno reads, dst the array written. */
static inline void update_cell(int i, int j, int k, int nx, int ny, int nz, double dx, double h,
                               const double *src, double *dst, const double *phiS)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    double gM = weno_cell(i, j, k, nx, ny, nz, dx, src);
    double sgn = lsf_oracle_phisign(phiS[IDX(i, j, k)], dx, gM);
    double k1 = sgn * (1. - gM);
    dst[IDX(i, j, k)] = src[IDX(i, j, k)] + h * k1;
}

/* Direction signs per raster 1..8, subs.f90:744-746, :758-760, :772-774, :786-788, :800-802,
 * :814-816, :828-830, :842-844. */
static const int RASTER_SIGN[8][3] = {{+1, +1, +1}, {+1, +1, -1}, {+1, -1, -1}, {-1, -1, -1},
                                      {-1, +1, -1}, {-1, -1, +1}, {-1, +1, +1}, {+1, -1, +1}};

static void sweep_lex(double *phi, const double *phiS, int nx, int ny, int nz, double dx, double h,
                      const int s[3])
{
    int ii, jj, kk;
    for (ii = 1; ii <= nx - 1; ++ii) {
        int i = s[0] > 0 ? ii : nx - ii;
        for (jj = 1; jj <= ny - 1; ++jj) {
            int j = s[1] > 0 ? jj : ny - jj;
            for (kk = 1; kk <= nz - 1; ++kk) {
                int k = s[2] > 0 ? kk : nz - kk;
                update_cell(i, j, k, nx, ny, nz, dx, h, phi, phi, phiS);
            }
        }
    }
}

/* Hyperplane order: a+b+c ascending in the sweep frame. */
static void sweep_hyper(double *phi, const double *phiS, int nx, int ny, int nz, double dx, double h,
                        const int s[3])
{
    int p, aa, bb;
    const int na = nx - 1, nb = ny - 1, nc = nz - 1;
    for (p = 3; p <= na + nb + nc; ++p) {
        /* the cells of one hyperplane do not read one another: with -fopenmp (liblsf_oracle_omp.so, used only to
         * generate long fixtures) they are shared among threads; every cell still executes the same operations on the
         * same operands, so the result is the one of the serial loop bit for bit (tests/test_oracle_golden.py) */
#ifdef _OPENMP
#pragma omp parallel for private(bb) schedule(static)
#endif
        for (aa = 1; aa <= na; ++aa) {
            for (bb = 1; bb <= nb; ++bb) {
                int cc = p - aa - bb;
                if (cc < 1 || cc > nc) continue;
                update_cell(s[0] > 0 ? aa : nx - aa, s[1] > 0 ? bb : ny - bb, s[2] > 0 ? cc : nz - cc,
                            nx, ny, nz, dx, h, phi, phi, phiS);
            }
        }
    }
}

static void sweep_jacobi(double *phi, double *scratch, const double *phiS, int nx, int ny, int nz,
                         double dx, double h)
{
    const size_t n = ((size_t)nx + 1) * ((size_t)ny + 1) * ((size_t)nz + 1);
    int i, j, k;
    memcpy(scratch, phi, n * sizeof(double));
    for (k = 1; k <= nz - 1; ++k)
        for (j = 1; j <= ny - 1; ++j)
            for (i = 1; i <= nx - 1; ++i) update_cell(i, j, k, nx, ny, nz, dx, h, scratch, phi, phiS);
}

static void bc_literal(double *phi, int nx, int ny, int nz, double dx)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    int i, j, k;
#define P(a, b, c) phi[IDX(a, b, c)]
    for (i = 0; i <= nx; ++i)
        for (j = 0; j <= ny; ++j)
            for (k = 0; k <= nz; ++k) {
                /* corners, subs.f90:864-871 */
                P(0, 0, 0) = P(1, 1, 1) + dx;
                P(nx, 0, 0) = P(nx - 1, 1, 1) + dx;
                P(0, ny, 0) = P(1, ny - 1, 1) + dx;
                P(0, 0, nz) = P(1, 1, nz - 1) + dx;
                P(nx, ny, 0) = P(nx - 1, ny - 1, 1) + dx;
                P(0, ny, nz) = P(1, ny - 1, nz - 1) + dx;
                P(nx, 0, nz) = P(nx - 1, 1, nz - 1) + dx;
                P(nx, ny, nz) = P(nx - 1, ny - 1, nz - 1) + dx;
                /* edges, subs.f90:874-885 */
                P(i, 0, 0) = P(i, 1, 1) + dx;
                P(0, j, 0) = P(1, j, 1) + dx;
                P(0, 0, k) = P(1, 1, k) + dx;
                P(i, ny, nz) = P(i, ny - 1, nz - 1) + dx;
                P(nx, j, nz) = P(nx - 1, j, nz - 1) + dx;
                P(nx, ny, k) = P(nx - 1, ny - 1, k) + dx;
                P(i, 0, nz) = P(i, 1, nz - 1) + dx;
                P(nx, j, 0) = P(nx - 1, j, 1) + dx;
                P(nx, 0, k) = P(nx - 1, 1, k) + dx;
                P(i, ny, 0) = P(i, ny - 1, 1) + dx;
                P(0, j, nz) = P(1, j, nz - 1) + dx;
                P(0, ny, k) = P(1, ny - 1, k) + dx;
                /* faces, subs.f90:888-893 */
                P(0, j, k) = P(1, j, k) + dx;
                P(i, 0, k) = P(i, 1, k) + dx;
                P(i, j, 0) = P(i, j, 1) + dx;
                P(nx, j, k) = P(nx - 1, j, k) + dx;
                P(i, ny, k) = P(i, ny - 1, k) + dx;
                P(i, j, nz) = P(i, j, nz - 1) + dx;
            }
#undef P
}

/* Closed form of the same loop: every wall point takes the value of
 * the interior point obtained by clamping each coordinate to [1, n-1] and adds dx
 * m = min(nb, 1 + nh) times in sequence, nb = number of coordinates on a wall, nh = number on a
 * high wall. */
static void bc_closed(double *phi, int nx, int ny, int nz, double dx)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    int i, j, k;
    for (k = 0; k <= nz; ++k)
        for (j = 0; j <= ny; ++j)
            for (i = 0; i <= nx; ++i) {
                int nb = (i == 0 || i == nx) + (j == 0 || j == ny) + (k == 0 || k == nz);
                int nh, m, t, ci, cj, ck;
                double v;
                if (!nb) continue;
                nh = (i == nx) + (j == ny) + (k == nz);
                m = nb < 1 + nh ? nb : 1 + nh;
                ci = i < 1 ? 1 : (i > nx - 1 ? nx - 1 : i);
                cj = j < 1 ? 1 : (j > ny - 1 ? ny - 1 : j);
                ck = k < 1 ? 1 : (k > nz - 1 ? nz - 1 : k);
                v = phi[IDX(ci, cj, ck)];
                for (t = 0; t < m; ++t) v = v + dx;
                phi[IDX(i, j, k)] = v;
            }
}

void lsf_oracle_bc(double *phi, int nx, int ny, int nz, double dx, int bc_kind)
{
    if (bc_kind == LSF_ORACLE_BC_LITERAL)
        bc_literal(phi, nx, ny, nz, dx);
    else
        bc_closed(phi, nx, ny, nz, dx);
}

/* subs.f90:902-914 and set3d.f90:435-447: sequential sum, i outermost, k innermost, over all
 * points; divisor is the INTEGER*4 product nx*ny*nz (wraps like the reference for huge grids). */
static double rms_change(const double *phi, const double *phiN, int nx, int ny, int nz)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    double err = 0.;
    int i, j, k;
    int32_t den = (int32_t)((uint32_t)nx * (uint32_t)ny * (uint32_t)nz);
    for (i = 0; i <= nx; ++i)
        for (j = 0; j <= ny; ++j)
            for (k = 0; k <= nz; ++k) {
                double d = phi[IDX(i, j, k)] - phiN[IDX(i, j, k)];
                err = err + d * d;
            }
    return sqrt(err / den);
}

int lsf_oracle_reinit(double *phi, int nx, int ny, int nz, int iter, double dx, double h, double tol,
                      int order, int bc_kind, int first_raster, int *sweeps_done, double *rms_trace,
                      int trace_cap)
{
    const size_t n = ((size_t)nx + 1) * ((size_t)ny + 1) * ((size_t)nz + 1);
    double *phiS = (double *)malloc(n * sizeof(double));
    double *phiN = (double *)malloc(n * sizeof(double));
    double *scratch = order == LSF_ORACLE_JACOBI ? (double *)malloc(n * sizeof(double)) : NULL;
    int raster = first_raster, nsweep, done = 0, rc = 0;
    memcpy(phiS, phi, n * sizeof(double)); /* subs.f90:731 */
    memcpy(phiN, phi, n * sizeof(double)); /* subs.f90:732 */
    for (nsweep = 0; nsweep <= iter; ++nsweep) { /* subs.f90:735 */
        double err;
        raster = raster + 1; /* :740 */
        if (order == LSF_ORACLE_GS_LEX)
            sweep_lex(phi, phiS, nx, ny, nz, dx, h, RASTER_SIGN[raster - 1]);
        else if (order == LSF_ORACLE_GS_HYPER)
            sweep_hyper(phi, phiS, nx, ny, nz, dx, h, RASTER_SIGN[raster - 1]);
        else
            sweep_jacobi(phi, scratch, phiS, nx, ny, nz, dx, h);
        if (raster == 8) raster = 0; /* :855 */
        lsf_oracle_bc(phi, nx, ny, nz, dx, bc_kind);
        err = rms_change(phi, phiN, nx, ny, nz);
        ++done;
        if (rms_trace && done <= trace_cap) rms_trace[done - 1] = err;
        if (err < tol) break;                  /* :915-918 */
        memcpy(phiN, phi, n * sizeof(double)); /* :921 */
        if (isnan(err)) {                      /* :926 */
            rc = 1;
            break;
        }
    }
    if (sweeps_done) *sweeps_done = done;
    free(phiS);
    free(phiN);
    free(scratch);
    return rc;
}

void lsf_oracle_narrowband(int nx, int ny, int nz, double dx, const double *phi, int32_t *phiNB,
                           int32_t *phiSB)
{
    const size_t n = ((size_t)nx + 1) * ((size_t)ny + 1) * ((size_t)nz + 1);
    size_t t;
    for (t = 0; t < n; ++t) {
        phiNB[t] = fabs(phi[t]) < 4.1 * dx ? 1 : 0; /* subs.f90:194 */
        phiSB[t] = fabs(phi[t]) < 8.1 * dx ? 1 : 0; /* subs.f90:199 */
    }
}

/* secondDeriv order 2 (subs.f90:384-389) + minMax (subs.f90:453-481) + the host's update
 * (set3d.f90:426) for one narrow-band cell.  lap3[0..2] are the frozen second derivatives of this
 * cell computed by pass A; phi is the in-place array. */
static inline double minmax_F(const double *phi, size_t c, size_t sx, size_t sxy, const double *lap3)
{
    double curv = lap3[0] + lap3[1] + lap3[2]; /* subs.f90:461 */
    /* subs.f90:473-474, h = 1 */
    double pAve = phi[c] + phi[c - 1] + phi[c + 1] + phi[c + sx] + phi[c - sx] + phi[c + sxy] + phi[c - sxy];
    pAve = pAve / 7.;
    if (pAve < 0.) return fmin2(curv, 0.0); /* :477-481 */
    return fmax2(curv, 0.0);
}

/* A band cell ON a wall would make the reference read outside phi (subs.f90:387-389 / :473 with
 * i-1 = -1 or i+1 = nx+1): undefined there, and impossible with the host's 10-cell padding
 * (set3d.f90:148).  Oracle and product both leave wall points untouched. */
#define MM_INTERIOR(i, j, k) ((i) >= 1 && (i) <= nx - 1 && (j) >= 1 && (j) <= ny - 1 && (k) >= 1 && (k) <= nz - 1)

int lsf_oracle_minmax(double *phi, int32_t *phiNB, int32_t *phiSB, int nx, int ny, int nz, int iter,
                      double dx, double h1, double tol, int order, int *iters_done, double *rms_trace,
                      int trace_cap)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1, sxy = sx * sy;
    const size_t n = sxy * ((size_t)nz + 1);
    double *phiN = (double *)malloc(n * sizeof(double));
    double *lap = (double *)calloc(3 * n, sizeof(double)); /* grad2Phi, set3d.f90:373 */
    double *scratch = order == LSF_ORACLE_JACOBI ? (double *)malloc(n * sizeof(double)) : NULL;
    int it, done = 0, rc = 0, i, j, k;
    memcpy(phiN, phi, n * sizeof(double)); /* set3d.f90:377 */
    for (it = 1; it <= iter; ++it) {       /* set3d.f90:394 */
        const double dxx = 1. / (dx * dx); /* subs.f90:384 */
        double err;
        /* pass A, set3d.f90:399-414 */
        for (i = 0; i <= nx; ++i)
            for (j = 0; j <= ny; ++j)
                for (k = 0; k <= nz; ++k) {
                    size_t c = IDX(i, j, k);
                    if (phiNB[c] == 1 && MM_INTERIOR(i, j, k)) {
                        lap[3 * c + 0] = (-2. * phi[c] + phi[c + 1] + phi[c - 1]) * dxx;
                        lap[3 * c + 1] = (-2. * phi[c] + phi[c + sx] + phi[c - sx]) * dxx;
                        lap[3 * c + 2] = (-2. * phi[c] + phi[c + sxy] + phi[c - sxy]) * dxx;
                    }
                }
        /* pass B, set3d.f90:417-431 */
        if (order == LSF_ORACLE_GS_LEX) {
            for (i = 0; i <= nx; ++i)
                for (j = 0; j <= ny; ++j)
                    for (k = 0; k <= nz; ++k) {
                        size_t c = IDX(i, j, k);
                        if (phiNB[c] == 1 && MM_INTERIOR(i, j, k))
                            phi[c] = phi[c] + h1 * minmax_F(phi, c, sx, sxy, &lap[3 * c]);
                    }
        } else if (order == LSF_ORACLE_GS_HYPER) {
            int p;
            for (p = 0; p <= nx + ny + nz; ++p)
                for (i = 0; i <= nx; ++i)
                    for (j = 0; j <= ny; ++j) {
                        size_t c;
                        k = p - i - j;
                        if (k < 0 || k > nz) continue;
                        c = IDX(i, j, k);
                        if (phiNB[c] == 1 && MM_INTERIOR(i, j, k))
                            phi[c] = phi[c] + h1 * minmax_F(phi, c, sx, sxy, &lap[3 * c]);
                    }
        } else {
            memcpy(scratch, phi, n * sizeof(double));
            for (k = 0; k <= nz; ++k)
                for (j = 0; j <= ny; ++j)
                    for (i = 0; i <= nx; ++i) {
                        size_t c = IDX(i, j, k);
                        if (phiNB[c] == 1 && MM_INTERIOR(i, j, k))
                            phi[c] = scratch[c] + h1 * minmax_F(scratch, c, sx, sxy, &lap[3 * c]);
                    }
        }
        err = rms_change(phi, phiN, nx, ny, nz); /* set3d.f90:435-447 */
        ++done;
        if (rms_trace && done <= trace_cap) rms_trace[done - 1] = err;
        if (err < tol) break;                  /* set3d.f90:448-451: EXIT before narrowBand */
        memcpy(phiN, phi, n * sizeof(double)); /* :454 */
        if (isnan(err)) {                      /* :458 */
            rc = 1;
            break;
        }
        lsf_oracle_narrowband(nx, ny, nz, dx, phi, phiNB, phiSB); /* :460 */
    }
    if (iters_done) *iters_done = done;
    free(phiN);
    free(lap);
    free(scratch);
    return rc;
}

/* ---- block-decomposed pieces (tests of levelsetfortran_amd/distributed.py) --------------------
 * box[9] = {lx,ly,lz, gx0,gy0,gz0, nx,ny,nz}: a local box of the global field (include/lsf.h
 * lsf_box).  Jacobi update (subs.f90:747-750, all reads from in) of the local cells [lo,hi). */
void lsf_oracle_jacobi_box(const double *in, double *out, const double *phiS, const int box[9],
                           const int lo[3], const int hi[3], double dx, double h, double *sumsq)
{
    const size_t s1 = (size_t)box[0], s2 = (size_t)box[0] * (size_t)box[1];
    int i, j, k;
    for (k = lo[2]; k < hi[2]; ++k)
        for (j = lo[1]; j < hi[1]; ++j)
            for (i = lo[0]; i < hi[0]; ++i) {
                const size_t c = (size_t)i + s1 * (size_t)j + s2 * (size_t)k;
                double gM = weno_at(&in[c], s1, s2, i + box[3], j + box[4], k + box[5], box[6], box[7], box[8], dx);
                double sgn = lsf_oracle_phisign(phiS[c], dx, gM);
                double k1 = sgn * (1. - gM);
                double d;
                out[c] = in[c] + h * k1;
                d = out[c] - in[c];
                *sumsq += d * d;
            }
}

/* closed-form BC (subs.f90:859-897) on the global wall points inside the local range [lo,hi) */
void lsf_oracle_bc_box(const double *in, double *out, const int box[9], const int lo[3], const int hi[3],
                       double dx, double *sumsq)
{
    const size_t s1 = (size_t)box[0], s2 = (size_t)box[0] * (size_t)box[1];
    const int nx = box[6], ny = box[7], nz = box[8];
    int i, j, k, t;
    for (k = lo[2]; k < hi[2]; ++k)
        for (j = lo[1]; j < hi[1]; ++j)
            for (i = lo[0]; i < hi[0]; ++i) {
                const int gi = i + box[3], gj = j + box[4], gk = k + box[5];
                const int nb = (gi == 0 || gi == nx) + (gj == 0 || gj == ny) + (gk == 0 || gk == nz);
                int nh, m, ci, cj, ck;
                double v, d;
                size_t c;
                if (!nb) continue;
                nh = (gi == nx) + (gj == ny) + (gk == nz);
                m = nb < 1 + nh ? nb : 1 + nh;
                ci = (gi < 1 ? 1 : (gi > nx - 1 ? nx - 1 : gi)) - box[3];
                cj = (gj < 1 ? 1 : (gj > ny - 1 ? ny - 1 : gj)) - box[4];
                ck = (gk < 1 ? 1 : (gk > nz - 1 ? nz - 1 : gk)) - box[5];
                v = out[(size_t)ci + s1 * (size_t)cj + s2 * (size_t)ck];
                for (t = 0; t < m; ++t) v = v + dx;
                c = (size_t)i + s1 * (size_t)j + s2 * (size_t)k;
                d = v - in[c];
                out[c] = v;
                *sumsq += d * d;
            }
}

void lsf_oracle_phi0(double *phi, int nx, int ny, int nz, double dx, const double xLo[3],
                     const double minX[3], const double maxX[3], const double *surfX, int nSurfNode,
                     const int32_t *surfElem, int nSurfElem)
{
    const size_t sx = (size_t)nx + 1, sy = (size_t)ny + 1;
    /* set3d.f90:180-186 */
    const int im = (int)floor((minX[0] - xLo[0]) / dx) - 3, ip = (int)floor((maxX[0] - xLo[0]) / dx) + 3;
    const int jm = (int)floor((minX[1] - xLo[1]) / dx) - 3, jp = (int)floor((maxX[1] - xLo[1]) / dx) + 3;
    const int km = (int)floor((minX[2] - xLo[2]) / dx) - 3, kp = (int)floor((maxX[2] - xLo[2]) / dx) + 3;
    double *cen = (double *)malloc((size_t)nSurfElem * 3 * sizeof(double));
    int n, i, j, k;
    (void)nz;
#define SX(node, comp) surfX[(size_t)((node)-1) + (size_t)nSurfNode * (size_t)(comp)]
#define SE(el, v) surfElem[(size_t)(el) + (size_t)nSurfElem * (size_t)(v)]
    for (n = 0; n < nSurfElem; ++n) { /* set3d.f90:199-215 */
        int n1 = SE(n, 0), n2 = SE(n, 1), n3 = SE(n, 2), cdim;
        for (cdim = 0; cdim < 3; ++cdim)
            cen[3 * (size_t)n + cdim] = (SX(n1, cdim) + SX(n2, cdim) + SX(n3, cdim)) / 3.;
    }
    for (i = im; i <= ip; ++i)
        for (j = jm; j <= jp; ++j)
            for (k = km; k <= kp; ++k) {
                /* gridX, set3d.f90:168-170 */
                const double gX = xLo[0] + i * dx, gY = xLo[1] + j * dx, gZ = xLo[2] + k * dx;
                double minD = 100000., A1, A2, A3, B1, B2, B3, C1, C2, C3, pSx, pSy, pSz, pS;
                int fN = 0, n1, n2, n3;
                for (n = 0; n < nSurfElem; ++n) { /* set3d.f90:224-236 */
                    const double pX = cen[3 * (size_t)n], pY = cen[3 * (size_t)n + 1], pZ = cen[3 * (size_t)n + 2];
                    double dis = sqrt((pX - gX) * (pX - gX) + (pY - gY) * (pY - gY) + (pZ - gZ) * (pZ - gZ));
                    if (dis < minD) {
                        minD = dis;
                        fN = n;
                    }
                }
                n1 = SE(fN, 0), n2 = SE(fN, 1), n3 = SE(fN, 2);
                A1 = SX(n1, 0) - gX, A2 = SX(n1, 1) - gY, A3 = SX(n1, 2) - gZ; /* :242-250 */
                B1 = SX(n2, 0) - gX, B2 = SX(n2, 1) - gY, B3 = SX(n2, 2) - gZ;
                C1 = SX(n3, 0) - gX, C2 = SX(n3, 1) - gY, C3 = SX(n3, 2) - gZ;
                pSx = A2 * B3 - A3 * B2;          /* :253 */
                pSy = -(A1 * B3 - B1 * A3);       /* :254 */
                pSz = A1 * B2 - B1 * A2;          /* :255 */
                pS = -(pSx * C1 + pSy * C2 + pSz * C3); /* :258 */
                phi[IDX(i, j, k)] = lsf_oracle_phisign(pS, dx, 1.); /* :260-264 */
            }
#undef SX
#undef SE
    free(cen);
}

/* ---- post-smoothing gradients + surface-node advection, set3d.f90:464-501 ----
 * firstDeriv order 8 (subs.f90:309-347) on the stencil-band cells, setPhiSurf (subs.f90:1056-1170) and the
 * advection loop.  Reference quirks kept: the y derivative uses phi(i,j+1,k) twice (subs.f90:346); neighbours
 * are addressed linearly like Fortran does without bounds checking (a band cell fewer than 4 points from a wall
 * reads the neighbouring row/plane); a read before the first / after the last element of phi is undefined in
 * the reference and yields 0 here (such cells lie 8 cells from the surface and are never interpolated).
 * gradPhi: (0:nx,0:ny,0:nz,3) work array, zero outside the band (set3d.f90:372). */
static double phi_lin(const double *phi, ptrdiff_t p, ptrdiff_t n)
{
    return (p < 0 || p >= n) ? 0.0 : phi[p];
}

void lsf_oracle_firstderiv8(const double *phi, const int32_t *phiSB, int nx, int ny, int nz, double dx,
                            double *gradPhi)
{
    const ptrdiff_t sx = (ptrdiff_t)nx + 1, sxy = sx * ((ptrdiff_t)ny + 1), n = sxy * ((ptrdiff_t)nz + 1);
    const double aa1 = 1. / 280., aa2 = -4. / 105., aa3 = 1. / 5., aa4 = -4. / 5., aa6 = 4. / 5, aa7 = -1. / 5.,
                 aa8 = 4. / 105., aa9 = -1. / 280.;
    ptrdiff_t p;
    memset(gradPhi, 0, (size_t)n * 3 * sizeof(double));
    for (p = 0; p < n; ++p) {
        double phiX, phiY, phiZ;
        if (phiSB[p] != 1) continue;
#define L(off) phi_lin(phi, p + (off), n)
        phiX = (L(-4) * aa1 + L(-3) * aa2 + L(-2) * aa3 + L(-1) * aa4 + L(1) * aa6 + L(2) * aa7 + L(3) * aa8 + L(4) * aa9) / dx;
        phiY = (L(-4 * sx) * aa1 + L(-3 * sx) * aa2 + L(-2 * sx) * aa3 + L(-sx) * aa4 + L(sx) * aa6 + L(sx) * aa7 +
                L(3 * sx) * aa8 + L(4 * sx) * aa9) / dx; /* subs.f90:346: jp1 twice */
        phiZ = (L(-4 * sxy) * aa1 + L(-3 * sxy) * aa2 + L(-2 * sxy) * aa3 + L(-sxy) * aa4 + L(sxy) * aa6 + L(2 * sxy) * aa7 +
                L(3 * sxy) * aa8 + L(4 * sxy) * aa9) / dx;
#undef L
        gradPhi[p] = phiX;
        gradPhi[p + n] = phiY;
        gradPhi[p + 2 * n] = phiZ;
    }
}

/* setPhiSurf for one node position; returns phiSurf, fills g[3] = gradPhiSurf (subs.f90:1076-1166) */
static double interp_node(const double *phi, const double *gradPhi, int nx, int ny, int nz, double dx,
                          const double xLo[3], const double X[3], double g[3])
{
    const ptrdiff_t sx = (ptrdiff_t)nx + 1, sxy = sx * ((ptrdiff_t)ny + 1), n = sxy * ((ptrdiff_t)nz + 1);
    const double x = X[0], y = X[1], z = X[2];
    const int i0 = (int)floor((x - xLo[0]) / dx), j0 = (int)floor((y - xLo[1]) / dx), k0 = (int)floor((z - xLo[2]) / dx);
    const double x0 = i0 * dx + xLo[0], y0 = j0 * dx + xLo[1], z0 = k0 * dx + xLo[2];
    const int i1 = i0 + 1, j1 = j0 + 1, k1 = k0 + 1;
    const double x1 = i1 * dx + xLo[0], y1 = j1 * dx + xLo[1], z1 = k1 * dx + xLo[2];
    const double xd = (x - x0) / (x1 - x0), yd = (y - y0) / (y1 - y0), zd = (z - z0) / (z1 - z0);
    const ptrdiff_t p000 = i0 + sx * j0 + sxy * k0;
    double out[4], gradMag2;
    int f;
    (void)nz;
    for (f = 0; f < 4; ++f) {
        const double *a = f == 0 ? phi : gradPhi + (ptrdiff_t)(f - 1) * n;
        double c00 = a[p000] * (1. - xd) + a[p000 + 1] * xd;
        double c10 = a[p000 + sx] * (1. - xd) + a[p000 + sx + 1] * xd;
        double c01 = a[p000 + sxy] * (1. - xd) + a[p000 + sxy + 1] * xd;
        double c11 = a[p000 + sxy + sx] * (1. - xd) + a[p000 + sxy + sx + 1] * xd;
        double c0 = c00 * (1. - yd) + c10 * yd;
        double c1 = c01 * (1. - yd) + c11 * yd;
        out[f] = c0 * (1. - zd) + c1 * zd;
    }
    g[0] = -out[1];
    g[1] = -out[2];
    g[2] = -out[3];
    gradMag2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
    if (gradMag2 < 1.E-7) {
        g[0] = g[1] = g[2] = 0.;
    } else {
        double m = sqrt(gradMag2);
        g[0] = g[0] / m;
        g[1] = g[1] / m;
        g[2] = g[2] / m;
    }
    return out[0];
}

/* set3d.f90:485-501: surfXX (nSurfNode,3 Fortran-ordered) in = the surface nodes, out = advected nodes */
void lsf_oracle_advect(const double *phi, const double *gradPhi, int nx, int ny, int nz, double dx,
                       const double xLo[3], double *surfXX, int nSurfNode, int iters)
{
    int n, it;
    for (n = 0; n < nSurfNode; ++n) {
        double X[3] = {surfXX[n], surfXX[n + nSurfNode], surfXX[n + 2 * (size_t)nSurfNode]}, g[3];
        double ps = interp_node(phi, gradPhi, nx, ny, nz, dx, xLo, X, g);
        for (it = 0; it < iters; ++it) {
            if (!(ps > 1E-13)) break; /* nothing moves any more: every later pass repeats this test */
            X[0] = X[0] + ps * g[0];
            X[1] = X[1] + ps * g[1];
            X[2] = X[2] + ps * g[2];
            ps = interp_node(phi, gradPhi, nx, ny, nz, dx, xLo, X, g);
        }
        surfXX[n] = X[0];
        surfXX[n + nSurfNode] = X[1];
        surfXX[n + 2 * (size_t)nSurfNode] = X[2];
    }
}
