// lsf_cell.hpp -- per-cell arithmetic of the hot path, device side (gfx950).
//
// Two arithmetic flavours of the same update (include/lsf.h, LSF_ARITH_*):
//   STRICT: every operation exactly as subs.f90 writes it, left to right, no FMA contraction,
//           IEEE division and square root -> bit-identical to the reference Fortran.
//   FAST:   same mathematics, restructured for the fp64 VALU: differences are kept unscaled
//           (one multiply by 1/dx at the end instead of 17 divisions per axis), the three
//           non-linear weights of a WENO side share one reciprocal, FMA contraction is on.
//           Rounding-level differences only (measured <= 1e-13 RMS against STRICT).
//
// Reference lines: weno subs.f90:489-711, phiSign subs.f90:152-172, update subs.f90:747-750,
// secondDeriv subs.f90:384-389, minMax subs.f90:453-481.
#pragma once
#include <hip/hip_runtime.h>

namespace lsf {

// Fortran MAX/MIN as compiled by flang (compare + select); STRICT only.
__device__ __forceinline__ double fmax2(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ double fmin2(double a, double b) { return (a < b) ? a : b; }
// The same for the reinit sweep, as ONE instruction (v_max_f64 / v_min_f64) instead of a compare and two 32-bit selects:
// identical to the compare-and-select for every pair of numbers (the second operand is a literal 0., a square or another
// maximum at every call site, never -0, and every minimum is squared by its consumer, subs.f90:684-692, so the sign of a
// zero cannot matter).  Only a NaN operand is treated differently (the instruction returns the other operand), and a NaN
// reaches these only after the reference's own arithmetic has already produced one -- the sweep of its NaN STOP.
__device__ __forceinline__ double smax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double smin(double a, double b) { return __builtin_fmin(a, b); }

// ---------------------------------------------------------------------------------------------
// STRICT: one axis of the WENO branch (subs.f90:509-552 / :555-598 / :601-644).
// q[0..6] = phi at -3..+3 along the axis in ABSOLUTE orientation; yquirk = subs.f90:576.
// ---------------------------------------------------------------------------------------------
// n / d exactly as the IEEE division, by the sequence the compiler itself emits for an fp64 division (v_rcp_f64, two
// Newton steps on the reciprocal, one product, one residual correction) WITHOUT its v_div_scale / v_div_fmas / v_div_fixup
// frame: those only act when an operand or the quotient is near the ends of the exponent range (or is zero, infinite,
// NaN), which weno_axis_strict rules out before it comes here.  The refined reciprocal is a value of its own so that the
// two weights of a WENO side, which divide by the same sum (subs.f90:540-543), share it.
__device__ __forceinline__ double recip_refined(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double div_by(double n, double d, double r)
{
#pragma clang fp contract(off)
    const double q = n * r;
    const double e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}

// x / dx: 17 of the 27 divisions of an axis divide by dx (subs.f90:509-513, :525-530); with the refined reciprocal of dx
// hoisted out of the march each is three instructions instead of the ~11 of a general fp64 division.  phi, hence x, is
// far from the overflow and underflow thresholds (a zero numerator gives zero); NaN propagates (the reference's NaN is
// born in phiSign, subs.f90:169, not here: the twoCube10 stop sweep is tested).  An INFINITE numerator gives NaN where the
// IEEE division gives infinity: a field that already holds infinities is one sweep away from the reference's NaN STOP
// either way (tests/test_gpu_parity.py, ..._at_the_ends_of_the_exponent_range).
__device__ __forceinline__ double div_dx(double x, double dx, double rdx) { return div_by(x, dx, rdx); }

// The second and first differences of an axis as subs.f90:509-513 / :525-530 write them.  Along an axis they are ONE formula at
// several offsets: with X(j) = (phi(j+1) - 2 phi(j) + phi(j-1)) / dx and its mirror image Y(j) = (phi(j-1) - 2 phi(j) + phi(j+1)) / dx
// (the reference starts the sum at the far point on either side), P(j) = (phi(j+1) - phi(j)) / dx:
//     ap = X(i+2), bp = X(i+1), cp = cm = X(i), am = Y(i-2), bm = Y(i-1),   p0 .. p5 = P(i-3) .. P(i+2)
// so a kernel that marches along the axis over frozen data (the Jacobi ordering) computes X(i+2), Y(i-1), P(i+2) and P(i+2)^2 per
// cell and carries the rest -- the same doubles by construction (WenoDiffs; k_reinit_jacobi<true>).
struct WenoDiffs {
    double ap, am, bp, bm, cp;       // second differences
    double p[6];                     // first differences p0 .. p5
    double s[6];                     // their squares (the epsilons of subs.f90:533-534 take the largest of five each)
    double t1p;                      // 13 (bp - cp)(bp - cp), which is 13 (ap - bp)(ap - bp) of the cell before
};
__device__ __forceinline__ double weno_X(double far, double mid, double near, double dx, double rdx)
{
#pragma clang fp contract(off)
    return div_dx(far - 2. * mid + near, dx, rdx); // (phi(j+1) - 2 phi(j) + phi(j-1)) / dx, or its mirror image
}
__device__ __forceinline__ double weno_P(double hi, double lo, double dx, double rdx)
{
#pragma clang fp contract(off)
    return div_dx(hi - lo, dx, rdx);
}
__device__ __forceinline__ void weno_diffs_strict(const double q[7], double dx, double rdx, bool yquirk, WenoDiffs& d)
{
#pragma clang fp contract(off)
    const double m3 = q[0], m2 = q[1], m1 = q[2], c0 = q[3], r1 = q[4], r2 = q[5], r3 = q[6];
    d.ap = weno_X(r3, r2, r1, dx, rdx);
    d.am = weno_X(m3, m2, m1, dx, rdx);
    d.bp = weno_X(r2, r1, c0, dx, rdx);
    d.bm = weno_X(m2, m1, c0, dx, rdx);
    d.cp = weno_X(r1, c0, m1, dx, rdx);
    d.p[0] = weno_P(m2, m3, dx, rdx);
    d.p[1] = weno_P(m1, m2, dx, rdx);
    d.p[2] = weno_P(c0, m1, dx, rdx);
    d.p[3] = weno_P(r1, c0, dx, rdx);
    d.p[4] = weno_P(r2, r1, dx, rdx);
    d.p[5] = yquirk ? weno_P(r3, r3, dx, rdx) : weno_P(r3, r2, dx, rdx);
#pragma unroll
    for (int k = 0; k < 6; ++k) d.s[k] = d.p[k] * d.p[k];
    d.t1p = 13. * (d.bp - d.cp) * (d.bp - d.cp);
}

// Everything of an axis behind its differences (subs.f90:518-552).  t0p_out: 13 (ap - bp)(ap - bp), the t1p of the next cell along
// the axis.
__device__ __forceinline__ void weno_from_diffs_strict(const WenoDiffs& d, double& dm, double& dp, double& t0p_out)
{
#pragma clang fp contract(off)
    const double ap = d.ap, am = d.am, bp = d.bp, bm = d.bm, cp = d.cp;
    const double cm = cp, dpp = bm, dmm = bp;
    const double p1 = d.p[1], p2 = d.p[2], p3 = d.p[3], p4 = d.p[4];

    // 13.*(x)*(x) is ((13 x) x): the same double for x and -x, so the first term of IS2m (x = cm - dmm = cp - bp) is that
    // of IS1p (x = bp - cp), and the first term of IS2p (cp - dpp = cp - bm) that of IS1m (bm - cm = bm - cp): two of the
    // twelve quadratic terms of subs.f90:518-523 are evaluated once (the rest differ in value, not only in sign)
    const double T1p = d.t1p, T1m = 13. * (bm - cm) * (bm - cm);
    const double T0p = 13. * (ap - bp) * (ap - bp);
    t0p_out = T0p;
    const double IS0p = T0p + 3. * (ap - 3. * bp) * (ap - 3. * bp);
    const double IS0m = 13. * (am - bm) * (am - bm) + 3. * (am - 3. * bm) * (am - 3. * bm);
    const double IS1p = T1p + 3. * (bp + cp) * (bp + cp);
    const double IS1m = T1m + 3. * (bm + cm) * (bm + cm);
    const double IS2p = T1m + 3. * (3. * cp - dpp) * (3. * cp - dpp);
    const double IS2m = T1p + 3. * (3. * cm - dmm) * (3. * cm - dmm);

    // the largest of five squares on either side (subs.f90:533-534): four of them are common to the two sides, and the largest
    // element of a set does not depend on the order it is searched in -- five maxima instead of eight
    const double smid = smax(smax(d.s[1], d.s[2]), smax(d.s[3], d.s[4]));
    const double epsp = (1.E-6) * smax(smid, d.s[5]) + 1.E-99;
    const double epsm = (1.E-6) * smax(d.s[0], smid) + 1.E-99;

    const double x0p = (epsp + IS0p) * (epsp + IS0p), x0m = (epsm + IS0m) * (epsm + IS0m);
    const double x1p = (epsp + IS1p) * (epsp + IS1p), x1m = (epsm + IS1m) * (epsm + IS1m);
    const double x2p = (epsp + IS2p) * (epsp + IS2p), x2m = (epsm + IS2m) * (epsm + IS2m);
    double w0p, w0m, w2p, w2m;
    // The ten divisions of subs.f90:533-543.  eps >= 1e-99 and IS <= ~100 max(p^2) <= 1e8 eps bound every divisor from
    // below (x >= 1e-198) and the weights' quotients to [1e-17, 1]; with the six sums eps + IS below 1e120 every divisor,
    // numerator and quotient is also far from the exponent limits where the hardware division would rescale -- then
    // div_by is that division bit for bit.  Anything else (a diverging field on its way to NaN, NaN itself) takes `/`.
    const double big = smax(smax(smax(epsp + IS0p, epsp + IS1p), epsp + IS2p), smax(smax(epsm + IS0m, epsm + IS1m), epsm + IS2m));
    if (__builtin_expect(big < 1.0e120, 1)) {
        const double a0p = div_by(1., x0p, recip_refined(x0p)), a0m = div_by(1., x0m, recip_refined(x0m));
        const double a1p = div_by(6., x1p, recip_refined(x1p)), a1m = div_by(6., x1m, recip_refined(x1m));
        const double a2p = div_by(3., x2p, recip_refined(x2p)), a2m = div_by(3., x2m, recip_refined(x2m));
        const double sp = a0p + a1p + a2p, sm = a0m + a1m + a2m;
        const double rp = recip_refined(sp), rm = recip_refined(sm);
        w0p = div_by(a0p, sp, rp), w2p = div_by(a2p, sp, rp);
        w0m = div_by(a0m, sm, rm), w2m = div_by(a2m, sm, rm);
    } else {
        const double a0p = 1. / x0p, a0m = 1. / x0m, a1p = 6. / x1p, a1m = 6. / x1m, a2p = 3. / x2p, a2m = 3. / x2m;
        w0p = a0p / (a0p + a1p + a2p);
        w0m = a0m / (a0m + a1m + a2m);
        w2p = a2p / (a0p + a1p + a2p);
        w2m = a2m / (a0m + a1m + a2m);
    }

    const double PWp = 1. / 3. * w0p * (ap - 2. * bp + cp) + 1. / 6. * (w2p - 0.5) * (bp - 2. * cp + dpp);
    const double PWm = 1. / 3. * w0m * (am - 2. * bm + cm) + 1. / 6. * (w2m - 0.5) * (bm - 2. * cm + dmm);

    dm = 1. / 12. * (-p1 + 7. * p2 + 7. * p3 - p4) - PWm;
    dp = 1. / 12. * (-p1 + 7. * p2 + 7. * p3 - p4) + PWp;
}

__device__ __forceinline__ void weno_axis_strict(const double q[7], double dx, bool yquirk, double& dm,
                                                 double& dp)
{
#pragma clang fp contract(off)
    const double rdx = recip_refined(dx); // loop invariant: hoisted out of every march
    WenoDiffs d;
    weno_diffs_strict(q, dx, rdx, yquirk, d);
    double t0p_;
    weno_from_diffs_strict(d, dm, dp, t0p_);
}

// STRICT: Godunov switch + magnitude, subs.f90:667-702.
__device__ __forceinline__ double godunov_strict(double phic, double a, double b, double c, double d,
                                                 double e, double f)
{
#pragma clang fp contract(off)
    const double pa = smax(a, 0.), pb = smax(b, 0.), pc = smax(c, 0.);
    const double pd = smax(d, 0.), pe = smax(e, 0.), pf = smax(f, 0.);
    const double na = smin(a, 0.), nb = smin(b, 0.), nc = smin(c, 0.);
    const double nd = smin(d, 0.), ne = smin(e, 0.), nf = smin(f, 0.);
    double gX, gY, gZ;
    if (phic > 0.) {
        gX = smax(pa * pa, nb * nb);
        gY = smax(pc * pc, nd * nd);
        gZ = smax(pe * pe, nf * nf);
    } else {
        gX = smax(pb * pb, na * na);
        gY = smax(pd * pd, nc * nc);
        gZ = smax(pf * pf, ne * ne);
    }
    return __builtin_sqrt(gX + gY + gZ);
}

template <bool STRICT>
__device__ __forceinline__ void axis_pair(const double q[7], bool weno_ok, bool yquirk, double dx, double floor2,
                                          double& dm, double& dp);
template <bool STRICT>
__device__ __forceinline__ double axis_godunov(double phic, double dm, double dp);
template <bool STRICT>
__device__ __forceinline__ double finish_update(double phic, double gX, double gY, double gZ, double pS, double dx,
                                                double inv_dx, double h);

// STRICT cell update: returns phi_new.  qx/qy/qz: -3..+3 stencils (only [2..4] are read when
// !weno_ok, the first-order branch subs.f90:657-662).
__device__ __forceinline__ double cell_update_strict(const double qx[7], const double qy[7],
                                                     const double qz[7], bool weno_ok, double pS,
                                                     double dx, double h)
{
#pragma clang fp contract(off)
    double a, b, c, d, e, f;
    const double phic = qx[3];
    if (weno_ok) {
        weno_axis_strict(qx, dx, false, a, b);
        weno_axis_strict(qy, dx, true, c, d);
        weno_axis_strict(qz, dx, false, e, f);
    } else {
        const double rdx = recip_refined(dx);
        a = div_dx(phic - qx[2], dx, rdx);
        b = div_dx(qx[4] - phic, dx, rdx);
        c = div_dx(phic - qy[2], dx, rdx);
        d = div_dx(qy[4] - phic, dx, rdx);
        e = div_dx(phic - qz[2], dx, rdx);
        f = div_dx(qz[4] - phic, dx, rdx);
    }
    // Godunov switch, |grad|, smeared sign, Euler step (subs.f90:667-702, :169, :749-750): the per-axis pieces below, which
    // return the bits of the literal forms (godunov_strict above is kept as their written-out statement)
    return finish_update<true>(phic, axis_godunov<true>(phic, a, b), axis_godunov<true>(phic, c, d), axis_godunov<true>(phic, e, f),
                               pS, dx, 0.0, h);
}

// ---------------------------------------------------------------------------------------------
// FAST flavour
// ---------------------------------------------------------------------------------------------
// reciprocal from v_rcp_f64 (4.6e-8 relative on gfx950, measured) + one Newton step -> 2e-15 relative.
// It only normalises the three non-linear WENO weights, which multiply third differences: far below the
// 1e-12 RMS budget of the FAST arithmetic (no div_scale/div_fixup: the argument is a positive sum of squares).
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// One axis, unscaled: returns dm*dx and dp*dx (the caller multiplies by 1/dx once).
// floor2 = 1e-99 * dx^2 / 13 (the reference's epsilon floor in unscaled units; the function works with IS / 3 and
// eps / 3 and rescales the floor itself).
//
// Algebra used (same mathematics as subs.f90:509-552, fewer operations):
//  * the six smoothness indicators need only four distinct differences of second differences,
//    e.g. IS1m uses (bm-cm)^2 = (cp-bm)^2, which IS2p already has;
//  * with q_k = eps + IS_k the weights are w0 = n0/D, w2 = 3 m2/D, n0 = (q1 q2)^2, n1 = (q0 q2)^2,
//    m2 = (q0 q1)^2, D = n0 + 6 n1 + 3 m2 (one reciprocal per side), and the correction term is
//        PW = 1/3 w0 (a-2b+c) + 1/6 (w2 - 1/2) (b-2c+d) = r (n0/3 Sa + m2/2 S0) - S0/12,   r = 1/D,
//    so the weights themselves are never formed.
__device__ __forceinline__ void weno_axis_fast(const double q[7], double floor2, bool yquirk, double& dm,
                                               double& dp)
{
    // first differences d_k = q[k+1]-q[k]  (p_k * dx)
    const double d0 = q[1] - q[0], d1 = q[2] - q[1], d2 = q[3] - q[2];
    const double d3 = q[4] - q[3], d4 = q[5] - q[4], d5 = q[6] - q[5];
    // second differences (am,bm,cp,bp,ap)*dx
    const double am = d1 - d0, bm = d2 - d1, cp = d3 - d2, bp = d4 - d3, ap = d5 - d4;
    // the four distinct differences of second differences and the six linear forms of the indicators
    const double e_ab = ap - bp, e_bc = bp - cp, e_cm = cp - bm, e_mm = am - bm;
    const double t0p = __builtin_fma(-3.0, bp, ap), t1p = bp + cp, t2p = __builtin_fma(3.0, cp, -bm);
    const double t0m = __builtin_fma(-3.0, bm, am), t1m = bm + cp, t2m = __builtin_fma(3.0, cp, -bp);

    // eps = 1e-6 max(p^2) + 1e-99 (subs.f90:533-534), in unscaled units and times 1/3 like the IS below (floor2 is the
    // floor divided by 13: times 13/3 here); max of squares = square of the max magnitude (|x| is a free modifier)
    const double mid = __builtin_fmax(__builtin_fmax(__builtin_fabs(d1), __builtin_fabs(d2)),
                                      __builtin_fmax(__builtin_fabs(d3), __builtin_fabs(d4)));
    const double mp = yquirk ? mid : __builtin_fmax(mid, __builtin_fabs(d5));
    const double mm = __builtin_fmax(mid, __builtin_fabs(d0));
    constexpr double C = 13.0 / 3.0;
    const double fl = C * floor2;
    const double epsp = __builtin_fma((1.E-6 / 3.0) * mp, mp, fl);
    const double epsm = __builtin_fma((1.E-6 / 3.0) * mm, mm, fl);

    // q_k = (eps + IS_k) / 3 = t^2 + ((13/3) e^2 + eps/3): two FMAs per q on top of the four products (13/3) e
    // (the squares e_bc^2 and e_cm^2 serve both sides, each with its own eps)
    const double c_ab = C * e_ab, c_bc = C * e_bc, c_cm = C * e_cm, c_mm = C * e_mm;
    const double q0p = __builtin_fma(t0p, t0p, __builtin_fma(c_ab, e_ab, epsp));
    const double q1p = __builtin_fma(t1p, t1p, __builtin_fma(c_bc, e_bc, epsp));
    const double q2p = __builtin_fma(t2p, t2p, __builtin_fma(c_cm, e_cm, epsp));
    const double q0m = __builtin_fma(t0m, t0m, __builtin_fma(c_mm, e_mm, epsm));
    const double q1m = __builtin_fma(t1m, t1m, __builtin_fma(c_cm, e_cm, epsm));
    const double q2m = __builtin_fma(t2m, t2m, __builtin_fma(c_bc, e_bc, epsm));

    const double S0 = e_bc - e_cm;  // bp - 2cp + bm   (= b-2c+d on both sides)
    const double S12 = S0 * (1.0 / 12.0), S0h = S0 * 0.5;
    // both sides' products first, then ONE reciprocal for the two normalisations: 1/Dp = Dm * 1/(Dp Dm) (v_rcp_f64 is a
    // quarter-rate instruction).  Dp Dm underflows only where the stencil is flat to ~1e-30 relative, or exactly flat
    // (D = 0: eps is the floor); the clamp then turns both correction terms into -S0/12, the central candidate, which is
    // what every candidate equals there.
    const double t12p = q1p * q2p, t02p = q0p * q2p, t01p = q0p * q1p;
    const double t12m = q1m * q2m, t02m = q0m * q2m, t01m = q0m * q1m;
    const double n0p = t12p * t12p, n1p = t02p * t02p, m2p = t01p * t01p;
    const double n0m = t12m * t12m, n1m = t02m * t02m, m2m = t01m * t01m;
    const double Dp = __builtin_fma(3.0, m2p, __builtin_fma(6.0, n1p, n0p));
    const double Dm = __builtin_fma(3.0, m2m, __builtin_fma(6.0, n1m, n0m));
    const double R = rcp_nr(__builtin_fmax(Dp * Dm, 1e-300));
    const double rp = Dm * R, rm = Dp * R;
    const double Sap = e_ab - e_bc; // ap - 2bp + cp
    const double Sam = e_mm + e_cm; // am - 2bm + cp = (am-bm) - (bm-cp)
    const double PWp = __builtin_fma(rp, __builtin_fma(n0p * (1.0 / 3.0), Sap, m2p * S0h), -S12);
    const double PWm = __builtin_fma(rm, __builtin_fma(n0m * (1.0 / 3.0), Sam, m2m * S0h), -S12);
    const double cen12 = __builtin_fma(7.0, d2 + d3, -(d1 + d4));
    dm = __builtin_fma(1.0 / 12.0, cen12, -PWm);
    dp = __builtin_fma(1.0 / 12.0, cen12, PWp);
}

// ---------------------------------------------------------------------------------------------
// FAST, interface form (Jacobi ordering only).  In a double-buffered sweep D+ of cell i and D- of cell i+1 are
// built from the SAME six points phi(i-2..i+3): the three smoothness indicators (subs.f90:518-523; IS0p(i) =
// IS2m(i+1), IS1p(i) = IS1m(i+1), IS2p(i) = IS0m(i+1)), the epsilon (the five first differences of :533 for cell i
// are those of :534 for cell i+1) and the products of the weights are shared; only the two normalisations and
// corrections differ.  One interface costs 61 operations for two one-sided corrections where the per-cell form
// (weno_axis_fast) spends 90 on the two sides of one cell.  (Not valid in the reference's in-place ordering: there
// the centre value differs between the two uses.  Not bit-compatible with the reference either -- the second
// differences are associated differently on the two sides of a cell, subs.f90:509-513 -- hence FAST only.)
//
// v[0..5] = phi(i-2 .. i+3).  Unscaled like weno_axis_fast (the caller multiplies by 1/dx once):
//     D+_i * dx     = cen12_i / 12 + pwp          cen12 = -d1 + 7 d2 + 7 d3 - d4 of cell i
//     D-_{i+1} * dx = cen12_{i+1} / 12 - pwm
// The y axis (p5 = 0 in eps+ only, subs.f90:576) keeps the per-cell form.
__device__ __forceinline__ void weno_iface_fast(const double v[6], double floor2, double& pwp, double& pwm, double& cen12)
{
#pragma clang fp contract(off) // every fused operation is written out: the same bits in every kernel that calls this
    const double d1 = v[1] - v[0], d2 = v[2] - v[1], d3 = v[3] - v[2], d4 = v[4] - v[3], d5 = v[5] - v[4];
    const double um = d2 - d1, u0 = d3 - d2, u1 = d4 - d3, u2 = d5 - d4; // second differences at i-1 .. i+2
    const double e_ab = u2 - u1, e_bc = u1 - u0, e_cm = u0 - um;
    const double t0 = __builtin_fma(-3.0, u1, u2), t1 = u1 + u0, t2 = __builtin_fma(3.0, u0, -um);
    const double mx = __builtin_fmax(__builtin_fmax(__builtin_fmax(__builtin_fabs(d1), __builtin_fabs(d2)),
                                                    __builtin_fmax(__builtin_fabs(d3), __builtin_fabs(d4))),
                                     __builtin_fabs(d5));
    constexpr double C = 13.0 / 3.0;
    const double eps = __builtin_fma((1.E-6 / 3.0) * mx, mx, C * floor2);
    const double q0 = __builtin_fma(t0, t0, __builtin_fma(C * e_ab, e_ab, eps));
    const double q1 = __builtin_fma(t1, t1, __builtin_fma(C * e_bc, e_bc, eps));
    const double q2 = __builtin_fma(t2, t2, __builtin_fma(C * e_cm, e_cm, eps));
    const double t12 = q1 * q2, t02 = q0 * q2, t01 = q0 * q1;
    const double n0 = t12 * t12, n1 = t02 * t02, m2 = t01 * t01;
    const double Dp = __builtin_fma(3.0, m2, __builtin_fma(6.0, n1, n0)); // plus side of cell i:    1, 6, 3
    const double Dm = __builtin_fma(3.0, n0, __builtin_fma(6.0, n1, m2)); // minus side of cell i+1: mirrored
    const double R = rcp_nr(__builtin_fmax(Dp * Dm, 1e-300));
    const double rp = Dm * R, rm = Dp * R;
    const double Sa = e_ab - e_bc, S0 = e_bc - e_cm;
    const double A = n0 * Sa, B = m2 * S0; // the two products both corrections are made of
    pwp = __builtin_fma(rp, __builtin_fma(A, 1.0 / 3.0, B * 0.5), -(S0 * (1.0 / 12.0)));
    pwm = __builtin_fma(rm, __builtin_fma(B, 1.0 / 3.0, A * 0.5), -(Sa * (1.0 / 12.0)));
    cen12 = __builtin_fma(7.0, d2 + d3, -(d1 + d4));
}

// the two one-sided differences of ONE cell from its two interfaces (q[0..6] = -3..+3): what a kernel that cannot
// share interfaces between lanes (thin rims of a block-decomposed sweep) uses, so that every fp64 Jacobi kernel of the
// library returns the same bits for the same cell
__device__ __forceinline__ void weno_axis_from_ifaces(const double q[7], double floor2, double& dm, double& dp)
{
#pragma clang fp contract(off)
    double pwp_l, pwm_c, cen_l, pwp_c, pwm_r, cen_c;
    weno_iface_fast(q, floor2, pwp_l, pwm_c, cen_l);
    weno_iface_fast(q + 1, floor2, pwp_c, pwm_r, cen_c);
    dm = __builtin_fma(1.0 / 12.0, cen_c, -pwm_c);
    dp = __builtin_fma(1.0 / 12.0, cen_c, pwp_c);
}


__device__ __forceinline__ double cell_update_fast(const double qx[7], const double qy[7],
                                                   const double qz[7], bool weno_ok, double pS, double dx,
                                                   double inv_dx, double floor2, double h)
{
    double a, b, c, d, e, f; // unscaled one-sided differences (true value * dx)
    // x and z in the interface form (the arithmetic of k_reinit_jacobi_sh, which shares each interface between two
    // cells), y in the per-cell form
    if (weno_ok) {
        weno_axis_from_ifaces(qx, floor2, a, b);
        weno_axis_from_ifaces(qz, floor2, e, f);
    } else {
        a = qx[3] - qx[2], b = qx[4] - qx[3];
        e = qz[3] - qz[2], f = qz[4] - qz[3];
    }
    axis_pair<false>(qy, weno_ok, true, dx, floor2, c, d);
    const double phic = qx[3];
    return finish_update<false>(phic, axis_godunov<false>(phic, a, b), axis_godunov<false>(phic, c, d),
                                axis_godunov<false>(phic, e, f), pS, dx, inv_dx, h);
}

template <bool STRICT>
__device__ __forceinline__ double cell_update(const double qx[7], const double qy[7], const double qz[7],
                                              bool weno_ok, double pS, double dx, double inv_dx,
                                              double floor2, double h)
{
    if constexpr (STRICT)
        return cell_update_strict(qx, qy, qz, weno_ok, pS, dx, h);
    else
        return cell_update_fast(qx, qy, qz, weno_ok, pS, dx, inv_dx, floor2, h);
}

// ---------------------------------------------------------------------------------------------
// Per-axis pieces (used by the lane-per-axis exact-GS kernel, where the x, y and z derivatives of
// one cell are computed by three lanes of a quad).  Same expressions as cell_update_* above.
// ---------------------------------------------------------------------------------------------
// one-sided derivatives along one axis: STRICT -> true values (a,b); FAST -> values * dx
template <bool STRICT>
__device__ __forceinline__ void axis_pair(const double q[7], bool weno_ok, bool yquirk, double dx, double floor2,
                                          double& dm, double& dp)
{
    if constexpr (STRICT) {
#pragma clang fp contract(off)
        if (weno_ok) {
            weno_axis_strict(q, dx, yquirk, dm, dp);
        } else {
            const double rdx = recip_refined(dx);
            dm = div_dx(q[3] - q[2], dx, rdx); // subs.f90:657-662
            dp = div_dx(q[4] - q[3], dx, rdx);
        }
    } else {
        if (weno_ok) {
            weno_axis_fast(q, floor2, yquirk, dm, dp);
        } else {
            dm = q[3] - q[2];
            dp = q[4] - q[3];
        }
    }
}

// the Godunov term of one axis (gradX / gradY / gradZ of subs.f90:684-692)
template <bool STRICT>
__device__ __forceinline__ double axis_godunov(double phic, double dm, double dp)
{
    // phic > 0: max(max(dm, 0)^2, min(dp, 0)^2), else max(max(dp, 0)^2, min(dm, 0)^2) (subs.f90:684-692).  With
    // sg = +-1 for phic > 0 / <= 0 both are m^2, m = max(sg dm, -sg dp, 0): min(x, 0)^2 = max(-x, 0)^2, and the larger of
    // two squares of non-negative numbers is the square of the larger number -- the same double, so STRICT keeps its bits
    // (products by +-1 are exact).  Seven instructions where the literal form took twelve (two of them canonicalising
    // v_max x, x in front of the maxima of sign-flipped bit patterns).
#pragma clang fp contract(off)
    const double sg = phic > 0. ? 1.0 : -1.0;
    const double u = dm * sg, w = dp * -sg;
    const double m = STRICT ? smax(smax(u, w), 0.) : __builtin_fmax(__builtin_fmax(u, w), 0.);
    return m * m;
}

// IEEE sqrt(x) by the sequence the compiler emits for it (v_rsq_f64, one coupled Newton step on root and half reciprocal
// root, two residual corrections) WITHOUT its frame (scaling by 2^256 below 2^-767, pass-through of zeros and infinity):
// the same instructions in the same order, hence the same bits, for every x in [1e-230, DBL_MAX] (profiles/micro/divcheck.hip
// checks it by brute force).  STRICT only; callers test the range and take __builtin_sqrt otherwise.
__device__ __forceinline__ double sqrt_unframed(double x)
{
#pragma clang fp contract(off)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, hh = y * 0.5;
    const double r = __builtin_fma(-hh, g, 0.5);
    g = __builtin_fma(g, r, g), hh = __builtin_fma(hh, r, hh);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, hh, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, hh, g);
}

// 1/sqrt(t) from v_rsq_f64 (5.2e-8 relative, measured) + one Newton step -> 4e-15 relative (FAST only)
__device__ __forceinline__ double rsqrt_nr(double t)
{
    const double y = __builtin_amdgcn_rsq(t);
    const double e = __builtin_fma(-0.5 * t * y, y, 0.5);
    return __builtin_fma(y, e, y);
}

// gM, smeared sign and Euler step from the three axis terms (subs.f90:702, :169, :749-750)
template <bool STRICT>
__device__ __forceinline__ double finish_update(double phic, double gX, double gY, double gZ, double pS, double dx,
                                                double inv_dx, double h)
{
    if constexpr (STRICT) {
#pragma clang fp contract(off)
        // subs.f90:702, :169, :749-750.  Two square roots and one division per cell, by the hardware sequences without
        // their frames where the operands allow it (see sqrt_unframed, div_by): 27 instructions where the framed forms take
        // 47.  A flat neighbourhood (S = 0), a vanishing sign field, NaN or anything near the ends of the exponent range
        // takes the plain operators -- the 0 / 0 of subs.f90:169 included.
        const double S = gX + gY + gZ;
        const double pp = pS * pS, dd = dx * dx;
        if (__builtin_expect(S >= 1.0e-200 && S <= 1.0e200 && pp >= 1.0e-200 && pp <= 1.0e200 && dd >= 1.0e-200 && dd <= 1.0e200, 1)) {
            const double gM = sqrt_unframed(S);
            const double t = pp + dd * gM;                  // >= pp: in [1e-200, ~1e200]
            const double den = sqrt_unframed(t);            // >= |pS|: the quotient is at most 1 in magnitude, at least ~1e-100
            const double sgn = div_by(pS, den, recip_refined(den));
            const double k1 = sgn * (1. - gM);
            return phic + h * k1;
        }
        const double gM = __builtin_sqrt(S);
        const double sgn = pS / __builtin_sqrt(pS * pS + dx * dx * gM);
        const double k1 = sgn * (1. - gM);
        return phic + h * k1;
    } else {
        const double S = gX + gY + gZ; // unscaled: true value * dx^2
        // sqrt(S) = S y0 refined once with the residual (y0 = v_rsq_f64, 5e-8 relative -> ~1e-15): h (1 - gM) moves
        // phi by < 1e-18 of that
        // (S = 0, a flat neighbourhood: y0 stays finite, g0 = 0 and g = 0 -- no select needed)
        const double y0 = __builtin_amdgcn_rsq(__builtin_fmax(S, 1e-300));
        const double g0 = S * y0;
        const double g = __builtin_fma(__builtin_fma(-g0, g0, S), 0.5 * y0, g0);
        const double gM = g * inv_dx;
        const double sgn = pS * rsqrt_nr(__builtin_fma(pS, pS, dx * dx * gM));
        return __builtin_fma(h, sgn * (1. - gM), phic);
    }
}

// ---------------------------------------------------------------------------------------------
// min/max flow
// ---------------------------------------------------------------------------------------------
// secondDeriv order 2 (subs.f90:384-389) summed as minMax does (subs.f90:461): curv.
// c = centre, (xp,xm,yp,ym,zp,zm) frozen neighbours.  Written so that STRICT and FAST agree
// bit-for-bit (contraction is switched off here: the kernel is bandwidth-bound).
__device__ __forceinline__ double minmax_curv(double c, double xp, double xm, double yp, double ym,
                                              double zp, double zm, double dxx)
{
#pragma clang fp contract(off)
    const double pxx = (-2. * c + xp + xm) * dxx;
    const double pyy = (-2. * c + yp + ym) * dxx;
    const double pzz = (-2. * c + zp + zm) * dxx;
    return pxx + pyy + pzz;
}

// minMax switch (subs.f90:473-481) on the in-place neighbourhood, then the host update
// set3d.f90:426.  Order of the pAve sum: centre, i-1, i+1, j+1, j-1, k+1, k-1.
__device__ __forceinline__ double minmax_update(double c, double im, double ip, double jp, double jm,
                                                double kp, double km, double curv, double h1)
{
#pragma clang fp contract(off)
    double pAve = c + im + ip + jp + jm + kp + km;
    pAve = pAve / 7.;
    const double F = (pAve < 0.) ? fmin2(curv, 0.0) : fmax2(curv, 0.0);
    return c + h1 * F;
}

} // namespace lsf
