// lsf_f32.hpp -- single-precision Jacobi reinitialisation for gfx950 (BASELINE.json configuration 5:
// 1536^3 fp32 on 2x2x2 GPUs; SURVEY.md section 8d "C5").
//
// The reference is fp64 only (Makefile:4 -fdefault-real-8), so there is no fp32 field to be identical
// to: this path is checked against the fp64 oracle within a stated tolerance (tests/test_gpu_f32.py) and
// against the analytic signed distance.  Same update as subs.f90:743-852 / weno :489-711 / phiSign
// :152-172, in the operation-lean algebra of weno_axis_fast (lsf_cell.hpp), with two changes fp32 forces:
//   * the reference's epsilon floor 1e-99 (subs.f90:533-534) is below the fp32 range: the floor is
//     LSF_F32_FLOOR (in the unscaled units of weno_axis_fast, like `floor2` there);
//   * the products of four q_k = eps + IS_k underflow fp32 (q ~ 1e-11 ... 1e-22 on a 1536^3 grid), so the
//     three q_k of a side are divided by their sum first (the weights are ratios: nothing changes
//     mathematically, and a constant field still gives the linear weights 0.1 / 0.6 / 0.3).
//
// Packed arithmetic: a lane owns TWO cells, (i,j,k) and (i,j+1,k), held in the two halves of a float2, so
// that the adds, multiplies and FMAs of the WENO algebra issue as v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 (two results per lane per issue slot; fp32 scalar instructions issue at the fp64 rate on
// CDNA4, so without packing fp32 would buy nothing on this VALU-bound kernel).  max/min/rcp/rsq/sqrt have
// no packed form and are done per half.
#pragma once
#include <hip/hip_runtime.h>

#include "lsf_kernels.hpp"

namespace lsf {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr float LSF_F32_FLOOR = 1.0e-30f;

__device__ __forceinline__ f2 mk2(float a, float b)
{
    f2 r;
    r.x = a;
    r.y = b;
    return r;
}
__device__ __forceinline__ f2 splat(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 max2(f2 a, f2 b) { return mk2(__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)); }
__device__ __forceinline__ f2 abs2(f2 a) { return mk2(__builtin_fabsf(a.x), __builtin_fabsf(a.y)); }
__device__ __forceinline__ f2 rcp2(f2 a) { return mk2(__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)); }

// 2^-e for x = m * 2^e, 1 <= m < 2 (x positive, normal): two integer instructions per half instead of a
// quarter-rate reciprocal; only a scale, so its value need not be exact
__device__ __forceinline__ f2 pow2_inv(f2 x)
{
    const unsigned a = (254u << 23) - (__float_as_uint(x.x) & 0x7f800000u);
    const unsigned b = (254u << 23) - (__float_as_uint(x.y) & 0x7f800000u);
    return mk2(__uint_as_float(a), __uint_as_float(b));
}

// One WENO side from the normalised q's: returns r * (n0/3 * Sa + m2 * S0/2) - S0/12  (see weno_axis_fast).
__device__ __forceinline__ f2 weno_side_f32(f2 q0, f2 q1, f2 q2, f2 Sa, f2 S0h, f2 S12)
{
    const f2 s = pow2_inv(q0 + q1 + q2); // q_k * s < 2, the largest >= 1/3
    q0 *= s, q1 *= s, q2 *= s;
    const f2 t12 = q1 * q2, t02 = q0 * q2, t01 = q0 * q1;
    const f2 n0 = t12 * t12, n1 = t02 * t02, m2 = t01 * t01;
    const f2 D = fma2(splat(3.0f), m2, fma2(splat(6.0f), n1, n0));
    const f2 r = rcp2(D);
    return fma2(r, fma2(n0 * splat(1.0f / 3.0f), Sa, m2 * S0h), -S12);
}

// One axis for the two cells of a lane, unscaled like weno_axis_fast: returns dm*dx and dp*dx.
// q[0..6] = phi at -3..+3 along the axis.  yquirk = subs.f90:576 (p5 = 0 on the y axis).
__device__ __forceinline__ void weno_axis_f32(const f2 q[7], bool yquirk, f2& dm, f2& dp)
{
    const f2 d0 = q[1] - q[0], d1 = q[2] - q[1], d2 = q[3] - q[2];
    const f2 d3 = q[4] - q[3], d4 = q[5] - q[4], d5 = q[6] - q[5];
    const f2 am = d1 - d0, bm = d2 - d1, cp = d3 - d2, bp = d4 - d3, ap = d5 - d4;
    const f2 e_ab = ap - bp, e_bc = bp - cp, e_cm = cp - bm, e_mm = am - bm;
    const f2 C = splat(13.0f / 3.0f), three = splat(3.0f);
    const f2 t0p = fma2(-three, bp, ap), t1p = bp + cp, t2p = fma2(three, cp, -bm);
    const f2 t0m = fma2(-three, bm, am), t1m = bm + cp, t2m = fma2(three, cp, -bp);

    const f2 mid = max2(max2(abs2(d1), abs2(d2)), max2(abs2(d3), abs2(d4)));
    const f2 mp = yquirk ? mid : max2(mid, abs2(d5));
    const f2 mm = max2(mid, abs2(d0));
    const f2 E = splat(1.E-6f / 3.0f), fl = splat(LSF_F32_FLOOR);
    const f2 epsp = fma2(E * mp, mp, fl);
    const f2 epsm = fma2(E * mm, mm, fl);

    // q_k = (eps + IS_k) / 3 = t^2 + ((13/3) e^2 + eps/3), see weno_axis_fast
    const f2 c_ab = C * e_ab, c_bc = C * e_bc, c_cm = C * e_cm, c_mm = C * e_mm;
    const f2 q0p = fma2(t0p, t0p, fma2(c_ab, e_ab, epsp)), q1p = fma2(t1p, t1p, fma2(c_bc, e_bc, epsp)),
             q2p = fma2(t2p, t2p, fma2(c_cm, e_cm, epsp));
    const f2 q0m = fma2(t0m, t0m, fma2(c_mm, e_mm, epsm)), q1m = fma2(t1m, t1m, fma2(c_cm, e_cm, epsm)),
             q2m = fma2(t2m, t2m, fma2(c_bc, e_bc, epsm));

    const f2 S0 = e_bc - e_cm;
    const f2 S12 = S0 * splat(1.0f / 12.0f), S0h = S0 * splat(0.5f);
    const f2 PWp = weno_side_f32(q0p, q1p, q2p, e_ab - e_bc, S0h, S12);
    const f2 PWm = weno_side_f32(q0m, q1m, q2m, e_mm + e_cm, S0h, S12);
    const f2 cen12 = fma2(splat(7.0f), d2 + d3, -(d1 + d4));
    dm = fma2(splat(1.0f / 12.0f), cen12, -PWm);
    dp = fma2(splat(1.0f / 12.0f), cen12, PWp);
}

// Interface form (see weno_iface_fast in lsf_cell.hpp): v[0..5] = phi(i-2 .. i+3) for the two cells of a lane; returns the
// D+ correction of cell i, the D- correction of cell i+1 and cen12 of cell i (all unscaled).  The two normalisations keep
// their own reciprocal: the product of the two denominators can leave the fp32 range.
__device__ __forceinline__ void weno_iface_f32(const f2 v[6], f2& pwp, f2& pwm, f2& cen12)
{
    const f2 d1 = v[1] - v[0], d2 = v[2] - v[1], d3 = v[3] - v[2], d4 = v[4] - v[3], d5 = v[5] - v[4];
    const f2 um = d2 - d1, u0 = d3 - d2, u1 = d4 - d3, u2 = d5 - d4;
    const f2 e_ab = u2 - u1, e_bc = u1 - u0, e_cm = u0 - um;
    const f2 C = splat(13.0f / 3.0f), three = splat(3.0f);
    const f2 t0 = fma2(-three, u1, u2), t1 = u1 + u0, t2 = fma2(three, u0, -um);
    const f2 mx = max2(max2(max2(abs2(d1), abs2(d2)), max2(abs2(d3), abs2(d4))), abs2(d5));
    const f2 eps = fma2(splat(1.E-6f / 3.0f) * mx, mx, splat(LSF_F32_FLOOR));
    f2 q0 = fma2(t0, t0, fma2(C * e_ab, e_ab, eps)), q1 = fma2(t1, t1, fma2(C * e_bc, e_bc, eps)),
       q2 = fma2(t2, t2, fma2(C * e_cm, e_cm, eps));
    const f2 s = pow2_inv(q0 + q1 + q2);
    q0 *= s, q1 *= s, q2 *= s;
    const f2 t12 = q1 * q2, t02 = q0 * q2, t01 = q0 * q1;
    const f2 n0 = t12 * t12, n1 = t02 * t02, m2 = t01 * t01;
    const f2 rp = rcp2(fma2(three, m2, fma2(splat(6.0f), n1, n0))); // plus side of cell i:    1, 6, 3
    const f2 rm = rcp2(fma2(three, n0, fma2(splat(6.0f), n1, m2))); // minus side of cell i+1: mirrored
    const f2 Sa = e_ab - e_bc, S0 = e_bc - e_cm;
    const f2 A = n0 * Sa, B = m2 * S0;
    pwp = fma2(rp, fma2(A, splat(1.0f / 3.0f), B * splat(0.5f)), -(S0 * splat(1.0f / 12.0f)));
    pwm = fma2(rm, fma2(B, splat(1.0f / 3.0f), A * splat(0.5f)), -(Sa * splat(1.0f / 12.0f)));
    cen12 = fma2(splat(7.0f), d2 + d3, -(d1 + d4));
}

// the two one-sided differences of one cell pair from its two interfaces (q[0..6] = -3..+3): kernels that cannot share
// interfaces between lanes (thin rims) use this, so that every fp32 kernel returns the same bits for the same cell
__device__ __forceinline__ void weno_axis_from_ifaces_f32(const f2 q[7], f2& dm, f2& dp)
{
    f2 pwp_l, pwm_c, cen_l, pwp_c, pwm_r, cen_c;
    weno_iface_f32(q, pwp_l, pwm_c, cen_l);
    weno_iface_f32(q + 1, pwp_c, pwm_r, cen_c);
    dm = fma2(splat(1.0f / 12.0f), cen_c, -pwm_c);
    dp = fma2(splat(1.0f / 12.0f), cen_c, pwp_c);
}

// Godunov term of one axis for the pair (subs.f90:684-692), unscaled one-sided differences.  With
// sg = sign(phic) (+1 / -1): max(max(sg dm, 0)^2, min(sg dp, 0)^2) is the reference's switch in one expression.
__device__ __forceinline__ f2 godunov_f32(f2 sg, f2 dm, f2 dp)
{
    // = m^2 with m = max(sg dm, -sg dp, 0): the larger of two squares of non-negative numbers is the square of the larger
    // number (lsf_cell.hpp: axis_godunov)
    const f2 u = sg * dm, w = -sg * dp;
    const f2 m = mk2(__builtin_fmaxf(__builtin_fmaxf(u.x, w.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(u.y, w.y), 0.f));
    return m * m;
}

// gM, smeared sign and Euler step (subs.f90:702, :169, :749-750) for the pair; S = gX+gY+gZ unscaled
__device__ __forceinline__ f2 finish_f32(f2 phic, f2 S, f2 pS, float dx2, float inv_dx, float h)
{
    const f2 gM = mk2(__builtin_sqrtf(S.x), __builtin_sqrtf(S.y)) * splat(inv_dx);
    const f2 t = fma2(pS, pS, splat(dx2) * gM);
    // pS = 0 and gM = 0 -> NaN, like the reference
    const f2 sgn = pS * mk2(__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y));
    return fma2(splat(h), sgn * (splat(1.f) - gM), phic);
}

// =============================================================================================
// Jacobi sweep of a box region, fp32.  Block 64 x F32_BY threads; thread (tx,ty) owns the cell pair
// (i, j) / (i, j+1), j = lo1 + 2*(blockIdx.y*F32_BY + ty), and marches F32_KC cells in k with a 7-deep
// window of pairs; x / y neighbours come through the vector L1/L2.  Partial sums of (new-old)^2 go to
// `partials` in double (fixed order -> reproducible).
// =============================================================================================
constexpr int F32_BX = 64, F32_BY = 4, F32_KC = 32;

template <bool THINX>
__global__ __launch_bounds__(F32_BX* F32_BY) void k_reinit_jacobi_f32(const float* __restrict__ A,
                                                                      float* __restrict__ Bout,
                                                                      const float* __restrict__ phiS, Box bx, int lo0,
                                                                      int lo1, int lo2, int hi0, int hi1, int hi2,
                                                                      float dx, float h,
                                                                      double* __restrict__ partials,
                                                                      const int* __restrict__ done, int xwall, int kc)
{
    __shared__ double red[F32_BX * F32_BY / 64];
    static_assert(!THINX || (F32_BX == 64 && F32_BY == 4), "THINX lane map: a block is 64 x 4 threads");
    if (done && *done) return;
    // THINX (x rim of a decomposed sweep, a few cells wide): lanes along the pair index instead of x
    // (a wavefront: 4 cells in x by 16 pairs -- four x neighbours share a cache line; a block covers 4 x 64 pairs either way)
    const int li = THINX ? lo0 + (int)(blockIdx.y * F32_BY + (threadIdx.x & 3)) : lo0 + (int)(blockIdx.x * F32_BX + threadIdx.x);
    const int lj = THINX ? lo1 + 2 * (int)(blockIdx.x * F32_BX + threadIdx.y * 16 + (threadIdx.x >> 2))
                         : lo1 + 2 * (int)(blockIdx.y * F32_BY + threadIdx.y);
    const int k0 = lo2 + blockIdx.z * kc;
    const int k1 = min(k0 + kc, hi2);
    const int sx = bx.lx;
    const long sxy = (long)bx.lx * bx.ly;
    float acc = 0.f;
    if (li < hi0 && lj < hi1) {
        const bool two = lj + 1 < hi1; // the second cell of the pair exists
        const int gi = li + bx.gx0, gj = lj + bx.gy0;
        const bool i_weno = gi > 3 && gi < bx.nx - 4;
        const bool jA = i_weno && gj > 3 && gj < bx.ny - 4;
        const bool jB = i_weno && two && gj + 1 > 3 && gj + 1 < bx.ny - 4;
        const float inv_dx = 1.0f / dx, dx2 = dx * dx;
        const bool wave_on_xwall = xwall && __any((int)(gi == 1 || gi == bx.nx - 1)) != 0;
        // the eight rows j-3..j+4 of the pair, clamped to the box (rows beyond the reach of the branch in
        // use are never consumed); offsets relative to the plane
        // Addressing: one buffer descriptor per k-plane (rebuilt on the scalar unit every step; a whole 1536^3
        // field is beyond a descriptor's 4 GB) + the lane's 32-bit byte offset inside the plane, so the loop has no
        // 64-bit address arithmetic and the seven x neighbours of a row arrive as one dwordx4 + one dwordx3.
        // The descriptor of the stencil plane starts 3 floats early: x offsets -3..+3 become immediates 0..24
        // (nothing is ever read below the plane: offsets < 12 are only used by cells with i >= 4).
        unsigned row[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = lj + t - 3;
            row[t] = 4u * (unsigned)(li + sx * (r < 0 ? 0 : (r > bx.ly - 1 ? bx.ly - 1 : r)));
        }
        const unsigned colA = row[3], colB = row[4];
        const unsigned plane_bytes = 4u * (unsigned)sxy;
        auto desc = [&](const float* base, int shift) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) - shift, 0, (int)(plane_bytes + 24u), 0x00020000);
        };
        auto at = [](__amdgpu_buffer_rsrc_t r, unsigned boff) -> float {
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, boff, 0, 0));
        };
        auto ldz = [&](int k) -> f2 {
            const int kk = k < 0 ? 0 : (k > bx.lz - 1 ? bx.lz - 1 : k);
            const auto r = desc(A + sxy * kk, 0);
            return mk2(at(r, colA), at(r, colB));
        };
        f2 qz[7];
#pragma unroll
        for (int m = 0; m < 6; ++m) qz[m + 1] = ldz(k0 - 3 + m);
        for (int k = k0; k < k1; ++k) {
#pragma unroll
            for (int m = 0; m < 6; ++m) qz[m] = qz[m + 1];
            qz[6] = ldz(k + 3);
            const int gk = k + bx.gz0;
            const bool k_weno = gk > 3 && gk < bx.nz - 4;
            const bool wA = jA && k_weno, wB = jB && k_weno;
            const auto P = desc(A + sxy * k, 3); // offset of (col, x+m): col + 4*(m+3)
            const f2 c = qz[3];
            // first-order one-sided differences (subs.f90:657-662), always valid
            const f2 xm1 = mk2(at(P, colA + 8), at(P, colB + 8)), xp1 = mk2(at(P, colA + 16), at(P, colB + 16));
            const float r2 = at(P, row[2] + 12), r5 = at(P, row[5] + 12);
            f2 a = c - xm1, b = xp1 - c;
            f2 cc = c - mk2(r2, c.x), d = mk2(c.y, r5) - c;
            f2 e = c - qz[2], f = qz[4] - c;
            if (wA || wB) {
                f2 qx[7], qy[7];
                qx[0] = mk2(at(P, colA), at(P, colB));
                qx[1] = mk2(at(P, colA + 4), at(P, colB + 4));
                qx[2] = xm1, qx[3] = c, qx[4] = xp1;
                qx[5] = mk2(at(P, colA + 20), at(P, colB + 20));
                qx[6] = mk2(at(P, colA + 24), at(P, colB + 24));
                const float r0 = at(P, row[0] + 12), r1 = at(P, row[1] + 12), r6 = at(P, row[6] + 12),
                            r7 = at(P, row[7] + 12);
                qy[0] = mk2(r0, r1), qy[1] = mk2(r1, r2), qy[2] = mk2(r2, c.x), qy[3] = c;
                qy[4] = mk2(c.y, r5), qy[5] = mk2(r5, r6), qy[6] = mk2(r6, r7);
                f2 wa, wb, wc, wd, we, wf;
                weno_axis_from_ifaces_f32(qx, wa, wb); // x and z in the interface form (the arithmetic of
                weno_axis_f32(qy, true, wc, wd);       // k_reinit_jacobi_f32_sh), y per cell
                weno_axis_from_ifaces_f32(qz, we, wf);
                if (wA) a.x = wa.x, b.x = wb.x, cc.x = wc.x, d.x = wd.x, e.x = we.x, f.x = wf.x;
                if (wB) a.y = wa.y, b.y = wb.y, cc.y = wc.y, d.y = wd.y, e.y = we.y, f.y = wf.y;
            }
            const f2 sg = mk2(c.x > 0.f ? 1.f : -1.f, c.y > 0.f ? 1.f : -1.f);
            const f2 S = godunov_f32(sg, a, b) + godunov_f32(sg, cc, d) + godunov_f32(sg, e, f);
            const auto PS = desc(phiS + sxy * k, 0);
            const auto PB = desc(Bout + sxy * k, 0);
            const f2 nv = finish_f32(c, S, mk2(at(PS, colA), two ? at(PS, colB) : 1.f), dx2, inv_dx, h);
            const f2 dl = nv - c;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nv.x), PB, colA, 0, 0);
            acc = __builtin_fmaf(dl.x, dl.x, acc);
            if (two) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nv.y), PB, colB, 0, 0);
                acc = __builtin_fmaf(dl.y, dl.y, acc);
            }
            // xwall (single-domain sweeps): the pure x-face wall points beside this pair, see k_reinit_jacobi
            if (wave_on_xwall && (gi == 1 || gi == bx.nx - 1)) {
                const f2 wv = nv + splat(dx);
                for (int side = 0; side < 2; ++side) {
                    if (side == 0 ? gi != 1 : gi != bx.nx - 1) continue;
                    const f2 old = side == 0 ? xm1 : xp1;
                    const f2 wd = wv - old;
                    const unsigned sh = side == 0 ? (unsigned)-4 : 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wv.x), PB, colA + sh, 0, 0);
                    acc = __builtin_fmaf(wd.x, wd.x, acc);
                    if (two) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wv.y), PB, colB + sh, 0, 0);
                        acc = __builtin_fmaf(wd.y, wd.y, acc);
                    }
                }
            }
        }
    }
    const double tot = wave_sum((double)acc);
    const int tid = threadIdx.x + F32_BX * threadIdx.y;
    if ((tid & 63) == 0) red[tid >> 6] = tot;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < F32_BX * F32_BY / 64; ++w) t += red[w];
        partials[blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)] = t;
    }
}

// =============================================================================================
// The fp32 sweep with WENO interfaces shared along x (DPP + LDS between wavefronts) and z (register carry); same
// structure as k_reinit_jacobi_sh (lsf_kernels.hpp), a lane owning the pair (i, j) / (i, j+1).  Block = 64 WX lanes along
// x, first lane a helper; every load of a step is issued at its top without a branch in front of it.
// =============================================================================================
__device__ __forceinline__ f2 dpp_shr1_f2(f2 v)
{
    const int a = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v.x), 0x138, 0xf, 0xf, false);
    const int b = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v.y), 0x138, 0xf, 0xf, false);
    return mk2(__uint_as_float((unsigned)a), __uint_as_float((unsigned)b));
}

template <int WX>
__global__ __launch_bounds__(64 * WX) void k_reinit_jacobi_f32_sh(const float* __restrict__ A, float* __restrict__ Bout,
                                                                  const float* __restrict__ phiS, Box bx, int lo0, int lo1, int lo2,
                                                                  int hi0, int hi1, int hi2, float dx, float h,
                                                                  double* __restrict__ partials, const int* __restrict__ done,
                                                                  int xwall, int nbx, int nby, int nbz, int kc)
{
    __shared__ double red[WX];
    __shared__ f2 xch[2][WX];
    if (done && *done) return;
    const unsigned per = gridDim.x >> 3;
    const unsigned L = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); // XCD-aware numbering, see k_reinit_jacobi_sh
    const unsigned nblk = (unsigned)nbx * nby * nbz;
    const int tid = threadIdx.x, lane = tid & 63, wx = tid >> 6;
    float acc = 0.f;
    if (L < nblk) {
        const int bxi = (int)(L % (unsigned)nbx), byi = (int)((L / (unsigned)nbx) % (unsigned)nby), bzi = (int)(L / ((unsigned)nbx * nby));
        const int xl = 64 * wx + lane;
        const int li = lo0 - 1 + bxi * (64 * WX - 1) + xl;
        const int lj = lo1 + 2 * byi;
        const int k0 = lo2 + bzi * kc, k1 = min(k0 + kc, hi2);
        const int sx = bx.lx;
        const long sxy = (long)bx.lx * bx.ly;
        const bool cell = xl >= 1 && li < hi0;
        const bool two = lj + 1 < hi1;
        const int gi = li + bx.gx0, gj = lj + bx.gy0;
        const bool i_weno = gi > 3 && gi < bx.nx - 4;
        const bool jA = i_weno && gj > 3 && gj < bx.ny - 4;
        const bool jB = i_weno && two && gj + 1 > 3 && gj + 1 < bx.ny - 4;
        // every lane of this wavefront takes the WENO branch as far as i and j go (a lane without a second row: its unused half)
        const bool wave_weno = __builtin_amdgcn_ballot_w64(jA && (jB || !two)) == ~0ull;
        const float inv_dx = 1.0f / dx, dx2 = dx * dx;
        const bool on_xwall = xwall && cell && (gi == 1 || gi == bx.nx - 1);
        // byte offsets inside a k-plane, fixed along the march: the lane's two points, their x neighbours (colA/B - 8 +
        // immediates; range-checked by the descriptor where they leave the plane) and the six rows above and below
        const int lic = min(li, bx.lx - 1);
        unsigned row[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) row[t] = 4u * (unsigned)(lic + sx * min(max(lj + t - 3, 0), bx.ly - 1));
        const unsigned colA = row[3], colB = row[4];
        const unsigned plane_bytes = 4u * (unsigned)sxy;
        auto desc = [&](const float* base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)plane_bytes, 0x00020000);
        };
        auto at = [](__amdgpu_buffer_rsrc_t r, unsigned boff) -> float {
            return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, boff, 0, 0));
        };
        auto ldz = [&](int k) -> f2 {
            const auto r = desc(A + sxy * min(max(k, 0), bx.lz - 1));
            return mk2(at(r, colA), at(r, colB));
        };
        f2 qz[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) qz[m] = ldz(k0 - 4 + m);
        f2 pwm_z;
        {
            f2 t0_, t1_;
            weno_iface_f32(qz + 1, t0_, pwm_z, t1_);
        }
        int pb = 0;
        for (int k = k0; k < k1; ++k) {
            const auto P = desc(A + sxy * k);
            f2 vx[6];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                if (m != 2) vx[m] = mk2(at(P, colA - 8u + 4u * (unsigned)m), at(P, colB - 8u + 4u * (unsigned)m));
#pragma unroll
            for (int m = 0; m < 6; ++m) qz[m] = qz[m + 1];
            qz[6] = ldz(k + 3);
            const f2 c = qz[3];
            vx[2] = c;
            const int gk = k + bx.gz0;
            const bool k_weno = gk > 3 && gk < bx.nz - 4;
            const bool wA = jA && k_weno, wB = jB && k_weno;
            f2 pwp_z, pwm_z_next, cen_z, pwp_x, pwm_x, cen_x;
            weno_iface_f32(qz + 1, pwp_z, pwm_z_next, cen_z);
            const float r0 = at(P, row[0]), r1 = at(P, row[1]), r2 = at(P, row[2]), r5 = at(P, row[5]), r6 = at(P, row[6]),
                        r7 = at(P, row[7]);
            const auto PS = desc(phiS + sxy * k);
            const f2 pS = mk2(at(PS, colA), at(PS, colB));
            weno_iface_f32(vx, pwp_x, pwm_x, cen_x);
            f2 pwm_l = dpp_shr1_f2(pwm_x);
            if (WX > 1) {
                if (lane == 63) xch[pb][wx] = pwm_x;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (lane == 0 && wx > 0) pwm_l = xch[pb][wx - 1];
                pb ^= 1;
            }
            // WENO everywhere; first-order one-sided differences (subs.f90:657-662) where the branch test fails -- behind a
            // wavefront-uniform branch, so that a wavefront whose cells all take the WENO branch neither evaluates nor selects them
            const f2 twelfth = splat(1.0f / 12.0f);
            f2 a = fma2(twelfth, cen_x, -pwm_l), b = fma2(twelfth, cen_x, pwp_x);
            f2 e = fma2(twelfth, cen_z, -pwm_z), f = fma2(twelfth, cen_z, pwp_z);
            f2 cc, d;
            {
                f2 qy[7];
                qy[0] = mk2(r0, r1), qy[1] = mk2(r1, r2), qy[2] = mk2(r2, c.x), qy[3] = c;
                qy[4] = mk2(c.y, r5), qy[5] = mk2(r5, r6), qy[6] = mk2(r6, r7);
                weno_axis_f32(qy, true, cc, d);
            }
            if (!(wave_weno && k_weno)) {
                const f2 a1 = c - vx[1], b1 = vx[3] - c, c1 = c - mk2(r2, c.x), d1 = mk2(c.y, r5) - c, e1 = c - qz[2], f1 = qz[4] - c;
                if (!wA) a.x = a1.x, b.x = b1.x, cc.x = c1.x, d.x = d1.x, e.x = e1.x, f.x = f1.x;
                if (!wB) a.y = a1.y, b.y = b1.y, cc.y = c1.y, d.y = d1.y, e.y = e1.y, f.y = f1.y;
            }
            const f2 sg = mk2(c.x > 0.f ? 1.f : -1.f, c.y > 0.f ? 1.f : -1.f);
            const f2 S = godunov_f32(sg, a, b) + godunov_f32(sg, cc, d) + godunov_f32(sg, e, f);
            const f2 nv = finish_f32(c, S, mk2(pS.x, two ? pS.y : 1.f), dx2, inv_dx, h);
            if (cell) {
                const auto PB = desc(Bout + sxy * k);
                const f2 dl = nv - c;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nv.x), PB, colA, 0, 0);
                acc = __builtin_fmaf(dl.x, dl.x, acc);
                if (two) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nv.y), PB, colB, 0, 0);
                    acc = __builtin_fmaf(dl.y, dl.y, acc);
                }
                if (on_xwall) { // the pure x-face wall points beside this pair (single-domain sweeps)
                    const f2 wv = nv + splat(dx);
                    for (int side = 0; side < 2; ++side) {
                        if (side == 0 ? gi != 1 : gi != bx.nx - 1) continue;
                        const f2 old = side == 0 ? vx[1] : vx[3];
                        const f2 wdl = wv - old;
                        const unsigned sh = side == 0 ? (unsigned)-4 : 4u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wv.x), PB, colA + sh, 0, 0);
                        acc = __builtin_fmaf(wdl.x, wdl.x, acc);
                        if (two) {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wv.y), PB, colB + sh, 0, 0);
                            acc = __builtin_fmaf(wdl.y, wdl.y, acc);
                        }
                    }
                }
            }
            pwm_z = pwm_z_next;
        }
    }
    const double tot = wave_sum((double)acc);
    if (lane == 0) red[wx] = tot;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < WX; ++w) t += red[w];
        partials[L] = t;
    }
}

} // namespace lsf
