// lsf_kernels.hpp -- HIP kernels for gfx950 (MI355X).  No MFMA: the path is a radius-3 star
// stencil in fp64 (HBM / fp64-VALU bound).  See DESIGN.md for the layout and the rooflines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "lsf_cell.hpp"

namespace lsf {

// A rank-local box of the global field (single GPU: the whole field, offsets 0).
struct Box {
    int lx, ly, lz;    // local extents (points)
    int gx0, gy0, gz0; // global index of local point 0
    int nx, ny, nz;    // global field is (0:nx,0:ny,0:nz)
};

__device__ __forceinline__ double wave_sum(double v)
{
    // fixed butterfly -> deterministic
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same butterfly, the same pairs in the same order -- hence the same bits -- without the LDS crossbar: gfx950's lane swaps
// for the partners 32 and 16 lanes away, DPP row rotations and quad permutations for 8, 4, 2, 1.  Six ds_bpermute round trips
// (~700 cycles when nothing else runs on the SIMD) become ~30 vector instructions; used where the sum sits between a tile's
// last store and its flag (skew_tile).
__device__ __forceinline__ double wave_sum_x(double v)
{
    auto join = [](int lo, int hi) { return __hiloint2double(hi, lo); };
    {   // partner 32 lanes away: a = [v(0..31) | v(0..31)], b = [v(32..63) | v(32..63)]
        const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = join(l[0], h[0]) + join(l[1], h[1]);
    }
    {   // 16 lanes away: a = rows [0, 0, 2, 2], b = rows [1, 1, 3, 3]
        const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = join(l[0], h[0]) + join(l[1], h[1]);
    }
    // 8 lanes away: row_ror:8
    v += join(__builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xf, 0xf, false));
    {   // 4 lanes away: lanes 0..3 and 8..11 of a row (banks 0, 2) read lane + 4 = row_ror:12, the others lane - 4 = row_ror:4
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x12c, 0xf, 0x5, false);
        lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x124, 0xf, 0xa, false);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x12c, 0xf, 0x5, false);
        hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x124, 0xf, 0xa, false);
        v += join(lo, hi);
    }
    // 2 and 1 lanes away: quad_perm [2,3,0,1] and [1,0,3,2]
    v += join(__builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4e, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4e, 0xf, 0xf, false));
    v += join(__builtin_amdgcn_update_dpp(0, __double2loint(v), 0xb1, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xb1, 0xf, 0xf, false));
    return v;
}

// lane l receives the value of lane l - 1 of the wavefront (lane 0: undefined), DPP wave_shr:1
__device__ __forceinline__ double dpp_shr1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// =============================================================================================
// Reinit, Jacobi ordering on a box region.  Thread (i,j) marches KC cells in k with a 7-deep
// register window; x/y neighbours come through the vector L1/L2.
// =============================================================================================
#ifndef LSF_JAC_BY
#define LSF_JAC_BY 4
#endif
#ifndef LSF_JAC_KC
#define LSF_JAC_KC 32
#endif
constexpr int JAC_BX = 64, JAC_BY = LSF_JAC_BY, JAC_KC = LSF_JAC_KC;

#ifndef LSF_JAC_WAVES
#define LSF_JAC_WAVES 1
#endif
// THINX: for regions only a few cells wide in x (the x rim of a block-decomposed sweep, 3 cells) the lanes of a
// wavefront cover 4 cells in x by 16 in y (blockIdx.y picks the group of four x cells): 48 busy lanes instead of 3.
template <bool STRICT, bool THINX = false>
__global__ __launch_bounds__(JAC_BX* JAC_BY, LSF_JAC_WAVES) void k_reinit_jacobi(const double* __restrict__ A,
                                                                   double* __restrict__ Bout,
                                                                   const double* __restrict__ phiS, Box bx,
                                                                   int lo0, int lo1, int lo2, int hi0, int hi1,
                                                                   int hi2, double dx, double h,
                                                                   double* __restrict__ partials,
                                                                   const int* __restrict__ done, int kc)
{
    __shared__ double red[JAC_BX * JAC_BY / 64];
    // the THINX lane map below (4 cells in x by 16 in y per wavefront, 4 wavefronts in y per block) is written for 64 x 4 threads
    static_assert(!THINX || (JAC_BX == 64 && JAC_BY == 4), "THINX lane map: a block is 64 x 4 threads (LSF_JAC_BY must stay 4)");
    if (done && *done) return;
    // THINX: a wavefront is 4 cells in x by 16 in y (four x neighbours share a cache line: a quarter of the lines a column of 64
    // cells touches per load); a block covers 4 x 64 cells either way
    const int li = THINX ? lo0 + (int)(blockIdx.y * JAC_BY + (threadIdx.x & 3)) : lo0 + (int)(blockIdx.x * JAC_BX + threadIdx.x);
    const int lj = THINX ? lo1 + (int)(blockIdx.x * JAC_BX + threadIdx.y * 16 + (threadIdx.x >> 2)) : lo1 + (int)(blockIdx.y * JAC_BY + threadIdx.y);
    const int k0 = lo2 + blockIdx.z * kc; // kc: planes a block marches (JacPlan: JAC_KC, fewer for thin regions)
    const int k1 = min(k0 + kc, hi2);
    const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
    double acc = 0.0;
    if (li < hi0 && lj < hi1) {
        const int gi = li + bx.gx0, gj = lj + bx.gy0;
        const bool ij_weno = gi > 3 && gi < bx.nx - 4 && gj > 3 && gj < bx.ny - 4;
        const double inv_dx = 1.0 / dx, floor2 = 1.E-99 * dx * dx / 13.0;
        // Addressing: one buffer descriptor per k-plane (rebuilt on the scalar unit every step; a 1024^3 field is
        // beyond a descriptor's 4 GB) + the lane's 32-bit byte offset inside the plane: no 64-bit address
        // arithmetic in the loop, and the x neighbours of a cell arrive as a few wide loads.  The descriptor of the
        // stencil plane starts 3 doubles early, so x offsets -3..+3 are immediates 0..48 (offsets < 24 are only
        // used by cells with i >= 4: nothing is ever read below the plane).
        const unsigned plane_bytes = 8u * (unsigned)sxy, rowb = 8u * (unsigned)sx;
        const unsigned col = 8u * (unsigned)(li + sx * lj);
        auto desc = [&](const double* base, int shift) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base) - shift, 0, (int)(plane_bytes + 48u),
                                                     0x00020000);
        };
        auto at = [](__amdgpu_buffer_rsrc_t r, unsigned boff) -> double {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, boff, 0, 0);
            return __hiloint2double((int)v.y, (int)v.x);
        };
        double qz[7];
        // window holds k-3..k+3 of the current cell; clamp reads to the box (values outside the
        // +-1 / +-3 reach of the branch in use are never consumed)
        auto ldz = [&](int k) -> double {
            const int kk = k < 0 ? 0 : (k > bx.lz - 1 ? bx.lz - 1 : k);
            return at(desc(A + sxy * kk, 0), col);
        };
#pragma unroll
        for (int m = 0; m < 6; ++m) qz[m + 1] = ldz(k0 - 3 + m);
        // STRICT: the differences of the march axis carried from plane to plane (lsf_cell.hpp, WenoDiffs): per cell one new X, one
        // new Y, one new P and its square instead of five second differences, six first differences and six squares -- the same
        // doubles, the formulas being the same at every offset.  (The data are frozen in this ordering; the exact ordering
        // cannot do this: there the centre value differs between a cell's two uses.)
        [[maybe_unused]] WenoDiffs dz;
        [[maybe_unused]] double t0p_prev = 0.0;
        [[maybe_unused]] bool prev_weno = false;
        [[maybe_unused]] const double rdx = recip_refined(dx);
        for (int k = k0; k < k1; ++k) {
#pragma unroll
            for (int m = 0; m < 6; ++m) qz[m] = qz[m + 1];
            qz[6] = ldz(k + 3);
            const int gk = k + bx.gz0;
            const bool weno_ok = ij_weno && gk > 3 && gk < bx.nz - 4;
            if constexpr (STRICT) {
                if (k == k0) {
                    weno_diffs_strict(qz, dx, rdx, false, dz);
                } else {
                    dz.cp = dz.bp, dz.bp = dz.ap, dz.am = dz.bm;
                    dz.ap = weno_X(qz[6], qz[5], qz[4], dx, rdx);
                    dz.bm = weno_X(qz[1], qz[2], qz[3], dx, rdx);
#pragma unroll
                    for (int m = 0; m < 5; ++m) dz.p[m] = dz.p[m + 1], dz.s[m] = dz.s[m + 1];
                    dz.p[5] = weno_P(qz[6], qz[5], dx, rdx);
                    {
#pragma clang fp contract(off)
                        dz.s[5] = dz.p[5] * dz.p[5];
                    }
                }
            }
            const auto P = desc(A + sxy * k, 3); // (x+m, y+n) of this cell: col + 8*(m+3) + n*rowb
            double qx[7], qy[7];
            if (weno_ok) {
#pragma unroll
                for (int m = 0; m < 7; ++m) {
                    qx[m] = (m == 3) ? qz[3] : at(P, col + 8u * (unsigned)m);
                    qy[m] = (m == 3) ? qz[3] : at(P, col + 24u + (unsigned)(m - 3) * rowb);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 7; ++m) { qx[m] = 0.0; qy[m] = 0.0; }
                qx[2] = at(P, col + 16u); qx[3] = qz[3]; qx[4] = at(P, col + 32u);
                qy[2] = at(P, col + 24u - rowb); qy[3] = qz[3]; qy[4] = at(P, col + 24u + rowb);
            }
            const auto PS = desc(phiS + sxy * k, 0);
            const auto PB = desc(Bout + sxy * k, 0);
            double newv;
            if constexpr (STRICT) {
#pragma clang fp contract(off)
                const double phic = qz[3], pS = at(PS, col);
                double a, b, c, d, e, f;
                if (weno_ok) {
                    // 13 (bp - cp)^2 of this cell is 13 (ap - bp)^2 of the cell before, if that cell was evaluated
                    dz.t1p = prev_weno ? t0p_prev : 13. * (dz.bp - dz.cp) * (dz.bp - dz.cp);
                    weno_axis_strict(qx, dx, false, a, b);
                    weno_axis_strict(qy, dx, true, c, d);
                    weno_from_diffs_strict(dz, e, f, t0p_prev);
                } else {
                    a = div_dx(phic - qx[2], dx, rdx), b = div_dx(qx[4] - phic, dx, rdx);
                    c = div_dx(phic - qy[2], dx, rdx), d = div_dx(qy[4] - phic, dx, rdx);
                    e = div_dx(phic - qz[2], dx, rdx), f = div_dx(qz[4] - phic, dx, rdx);
                }
                prev_weno = weno_ok;
                newv = finish_update<true>(phic, axis_godunov<true>(phic, a, b), axis_godunov<true>(phic, c, d), axis_godunov<true>(phic, e, f),
                                           pS, dx, 0.0, h);
            } else {
                newv = cell_update<STRICT>(qx, qy, qz, weno_ok, at(PS, col), dx, inv_dx, floor2, h);
            }
            {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                u2 w;
                w.x = (unsigned)__double2loint(newv);
                w.y = (unsigned)__double2hiint(newv);
                __builtin_amdgcn_raw_buffer_store_b64(w, PB, col, 0, 0);
            }
            const double dlt = newv - qz[3];
            acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
        }
    }
    acc = wave_sum(acc);
    const int tid = threadIdx.x + JAC_BX * threadIdx.y;
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < JAC_BX * JAC_BY / 64; ++w) t += red[w];
        partials[blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)] = t;
    }
}

// =============================================================================================
// Reinit, Jacobi ordering, STRICT arithmetic, every difference of subs.f90:509-513 / :525-530 evaluated ONCE per point.
// Along an axis the five second and six first differences of a cell are one formula at several offsets (lsf_cell.hpp,
// WenoDiffs): X(j), its mirror image Y(j) and P(j).  In the Jacobi ordering all inputs are the previous sweep's, so the
// value a neighbour computes IS the double this cell would compute -- sharing keeps the bits.
//   z (march axis): carried in registers; the march loop is unrolled six times so that the windows (6 field values, 6 P, 6 P^2,
//                   3 X, 2 Y) rotate by renaming, without a move.  Per cell: one X, one Y, one P, one square.
//   x, y:           a wavefront owns a patch of 16 x 4 columns.  Every lane evaluates X, Y, P, P^2 of its own point along x and
//                   along y and puts them into the wavefront's part of LDS; the points the patch's cells reach beyond its edge
//                   (x: 9 per row, y: 8 per column -- 164 values) are evaluated by three "edge jobs", one value per lane and job
//                   (operands by their own loads).  Then every lane reads the 14 + 12 differences of its neighbours.  LDS
//                   is private to a wavefront here: its LDS operations complete in order, no barrier.
// (p5 = 0 of the y axis, subs.f90:576, is the constant it evaluates to.)
// 189 -> 78 instructions per cell for the differences; the rest of a cell (weno_from_diffs_strict, Godunov, Euler) is
// k_reinit_jacobi<true>'s, hence the same doubles.  The per-cell kernel stays for the 3-cell x rims (THINX).
// =============================================================================================
constexpr int JSS_PX = 16, JSS_PY = 4;            // a wavefront's patch of columns
constexpr int JSS_WX = 2, JSS_WY = 2;             // patches of a block: 32 x 8 columns, 256 threads
constexpr int JSS_XP = JSS_PX + 5;                // x arrays: points i0-3 .. i0+17 of a row
constexpr int JSS_XN = JSS_XP * JSS_PY;
constexpr int JSS_YR = JSS_PY + 5;                // y arrays: rows j0-3 .. j0+5
constexpr int JSS_YN = JSS_YR * JSS_PX;
constexpr int JSS_WAVE = 4 * JSS_XN + 4 * JSS_YN; // doubles of LDS per wavefront: P, P^2, X, Y along x, then along y
__global__ __launch_bounds__(64 * JSS_WX * JSS_WY) void k_reinit_jacobi_strict_sh(const double* __restrict__ A, double* __restrict__ Bout,
                                                                                 const double* __restrict__ phiS, Box bx, int lo0, int lo1,
                                                                                 int lo2, int hi0, int hi1, int hi2, double dx, double h,
                                                                                 double* __restrict__ partials, const int* __restrict__ done,
                                                                                 int kc, int nbx, int nby, int nbz)
{
#pragma clang fp contract(off)
    __shared__ double red[JSS_WX * JSS_WY];
    __shared__ double lds[JSS_WX * JSS_WY * JSS_WAVE];
    if (done && *done) return;
    // XCD-aware numbering (as k_reinit_jacobi_sh): the launch is one-dimensional and padded to a multiple of 8; the 8 XCDs take the
    // launch indices round robin, XCD x gets the logical blocks [x * per, (x + 1) * per) -- a contiguous range of patches (x fastest,
    // then y, then the chunks along z), so that the points a patch reaches beyond its edge (the edge jobs' operands, the +-1 loads)
    // are what another block of the SAME XCD loads too and come from its L2 instead of the fabric.  Round 6, VERDICT r5 item 7:
    // fabric traffic 1.96 x algorithmic with a three-dimensional grid dealt out block by block (neighbouring patches on eight
    // different XCDs).  The partial sums keep their order (index = logical block): the RMS bits do not change.
    const unsigned per = gridDim.x >> 3;
    const unsigned L = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (L >= (unsigned)nbx * (unsigned)nby * (unsigned)nbz) return; // padding
    const int bxi = (int)(L % (unsigned)nbx), byi = (int)((L / (unsigned)nbx) % (unsigned)nby), bzi = (int)(L / ((unsigned)nbx * (unsigned)nby));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & (JSS_PX - 1), ly = lane / JSS_PX;
    const int i0 = lo0 + (bxi * JSS_WX + (wave % JSS_WX)) * JSS_PX; // the patch's first column
    const int j0 = lo1 + (byi * JSS_WY + (wave / JSS_WX)) * JSS_PY;
    const int li = i0 + lx, lj = j0 + ly;
    const int k0 = lo2 + bzi * kc, k1 = min(k0 + kc, hi2);
    const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
    const bool cell = li < hi0 && lj < hi1; // lanes beyond the region still evaluate the differences their neighbours read (hi <= l - 1)
    const int gi = li + bx.gx0, gj = lj + bx.gy0;
    const bool ij_weno = gi > 3 && gi < bx.nx - 4 && gj > 3 && gj < bx.ny - 4;
    const double rdx = recip_refined(dx);
    // addressing as in k_reinit_jacobi: one buffer descriptor per k-plane, starting 3 doubles early, + 32-bit byte offsets; what
    // lies outside a plane reads as zero, and only differences no cell of the launch consumes are made of such values (a
    // cell reaches 3 points when it takes the WENO branch, 1 otherwise, and those exist)
    // Every offset handed to a load is INSIDE its plane by construction (the descriptor's range check is a second line only): a
    // lane or an edge job whose operands are not all points of the array reads around point (1, 1) instead.
    const unsigned plane_bytes = 8u * (unsigned)sxy, rowb = 8u * (unsigned)sx;
    const unsigned safe = 24u + 8u + rowb;
    auto inside = [&](int x, int y) { return x >= 1 && x <= bx.lx - 2 && y >= 1 && y <= bx.ly - 2; }; // the point and its four neighbours exist
    const unsigned col = inside(li, lj) ? 24u + 8u * (unsigned)(li + sx * lj) : safe;
    auto desc = [&](const double* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base) - 3, 0, (int)(plane_bytes + 48u), 0x00020000);
    };
    auto at = [](__amdgpu_buffer_rsrc_t r, unsigned boff) -> double {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, boff, 0, 0);
        return __hiloint2double((int)v.y, (int)v.x);
    };
    auto ldz = [&](int k) -> double {
        const int kk = k < 0 ? 0 : (k > bx.lz - 1 ? bx.lz - 1 : k);
        return at(desc(A + sxy * kk), col);
    };
    double* const w = lds + wave * JSS_WAVE;
    double* const wxo = w + ly * JSS_XP + lx + 3;              // the lane's own entry of an x array (arrays JSS_XN apart)
    double* const wyo = w + 4 * JSS_XN + (ly + 3) * JSS_PX + lx; // ... of a y array (JSS_YN apart)

    // edge jobs: kind 0 = P, 1 = X, 2 = Y; mid point offset, far = mid + s, near = mid + q; destination in LDS
    // (a P reads two points: its third operand is loaded from the middle one and dropped)
    auto job_mid = [&](int x, int y, long far, long near) {
        const long m = 24 + 8 * ((long)x + sx * (long)y), lo = m + (far < near ? far : near), hi = m + (far < near ? near : far);
        return lo >= 0 && hi <= (long)plane_bytes + 40 ? (unsigned)m : safe;
    };
    unsigned jm[3], js[3], jq[3];
    double jw[3];
    double* jd[3];
    bool jp[3], jon[3];
    {
        // x edges, 9 per row: P(-3) P(-2) P(-1) P(16) P(17) X(16) X(17) Y(-2) Y(-1)
        const int r = lane / 9, e = lane % 9;
        const int ex = e < 3 ? e - 3 : (e < 5 ? e + 13 : (e < 7 ? e + 11 : e - 9));
        const int kind = e < 5 ? 0 : (e < 7 ? 1 : 2);
        jon[0] = lane < 9 * JSS_PY;
        jp[0] = kind == 0;
        js[0] = kind == 2 ? 0u - 8u : 8u;
        jq[0] = kind == 0 ? 0u : 0u - js[0];
        jw[0] = kind == 0 ? 1. : 2.;
        jm[0] = job_mid(i0 + ex, j0 + r, (int)js[0], (int)jq[0]);
        jd[0] = w + (kind == 0 ? 0 : kind + 1) * JSS_XN + (jon[0] ? r : 0) * JSS_XP + ex + 3;
    }
#pragma unroll
    for (int t = 1; t < 3; ++t) {
        // y edges, 8 per column: P(-3) P(-2) P(-1) P(4) X(4) X(5) Y(-2) Y(-1)
        const int id = (t - 1) * 64 + lane, c = id & (JSS_PX - 1), e = id / JSS_PX;
        const int ey = e < 3 ? e - 3 : (e == 3 ? 4 : (e < 6 ? e : e - 8));
        const int kind = e < 4 ? 0 : (e < 6 ? 1 : 2);
        jon[t] = true;
        jp[t] = kind == 0;
        js[t] = kind == 2 ? 0u - rowb : rowb;
        jq[t] = kind == 0 ? 0u : 0u - js[t];
        jw[t] = kind == 0 ? 1. : 2.;
        jm[t] = job_mid(i0 + c, j0 + ey, (int)js[t], (int)jq[t]);
        jd[t] = w + 4 * JSS_XN + (kind == 0 ? 0 : kind + 1) * JSS_YN + (ey + 3) * JSS_PX + c;
    }

    // march-axis windows; slot of an entry = its plane index relative to k0, modulo the window length
    double W[6], Pz[6], Sz[6], Xz[3], Yz[2], Tz;
#pragma unroll
    for (int m = 0; m < 6; ++m) W[m] = ldz(k0 + (m < 3 ? m : m - 6)); // planes k0-3 .. k0+2; slot 3 = plane k0-3 until step 0 loads k0+3
#pragma unroll
    for (int m = 0; m < 5; ++m) { // P(k0-3) .. P(k0+1) -> slots 0 .. 4
        Pz[m] = weno_P(W[(m + 4) % 6], W[(m + 3) % 6], dx, rdx);
        Sz[m] = Pz[m] * Pz[m];
    }
    Pz[5] = Sz[5] = 0.0;
    Xz[0] = weno_X(W[1], W[0], W[5], dx, rdx); // X(k0), X(k0+1)
    Xz[1] = weno_X(W[2], W[1], W[0], dx, rdx);
    Xz[2] = 0.0;
    Yz[0] = weno_X(W[3], W[4], W[5], dx, rdx); // Y(k0-2); slot 1 = Y(k0-1) comes with step 0
    Yz[1] = 0.0;
    Tz = 13. * (Xz[1] - Xz[0]) * (Xz[1] - Xz[0]);

    double acc = 0.0;
    double qnext = ldz(k0 + 3); // the march-axis value one step ahead of its use
    auto step = [&](auto phc, int k) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr int PH = decltype(phc)::value;
        const auto PA = desc(A + sxy * k);
        // loads of this step; the z axis, which needs none of them, is evaluated while they are in flight
        const double qn = qnext;
        qnext = ldz(k + 4);
        const double xm = at(PA, col - 8u), xp = at(PA, col + 8u), ym = at(PA, col - rowb), yp = at(PA, col + rowb);
        double jf[3], jmid[3], jn[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            jf[t] = at(PA, jm[t] + js[t]);
            jmid[t] = at(PA, jm[t]);
            jn[t] = at(PA, jm[t] + jq[t]);
        }
        const double pS = at(desc(phiS + sxy * k), col);
        W[(PH + 3) % 6] = qn;
        const double c0 = W[PH % 6], zm1 = W[(PH + 5) % 6], zm2 = W[(PH + 4) % 6], zp1 = W[(PH + 1) % 6], zp2 = W[(PH + 2) % 6];
        const int gk = k + bx.gz0;
        const bool weno_ok = ij_weno && gk > 3 && gk < bx.nz - 4;
        double a, b, c, d, e, f;
        // z: P(k+2), X(k+2), Y(k-1) are new
        Pz[(PH + 5) % 6] = weno_P(qn, zp2, dx, rdx);
        Sz[(PH + 5) % 6] = Pz[(PH + 5) % 6] * Pz[(PH + 5) % 6];
        Xz[(PH + 2) % 3] = weno_X(qn, zp2, zp1, dx, rdx);
        Yz[(PH + 1) % 2] = weno_X(zm2, zm1, c0, dx, rdx);
        if (weno_ok) {
            WenoDiffs D;
#pragma unroll
            for (int m = 0; m < 6; ++m) D.p[m] = Pz[(PH + m) % 6], D.s[m] = Sz[(PH + m) % 6];
            D.cp = Xz[PH % 3], D.bp = Xz[(PH + 1) % 3], D.ap = Xz[(PH + 2) % 3];
            D.am = Yz[PH % 2], D.bm = Yz[(PH + 1) % 2];
            D.t1p = Tz;
            weno_from_diffs_strict(D, e, f, Tz); // 13 (ap - bp)^2 of this cell is 13 (bp - cp)^2 of the next one along z
        } else {
            e = Pz[(PH + 2) % 6], f = Pz[(PH + 3) % 6]; // subs.f90:661-662: P(k-1), P(k)
            Tz = 13. * (Xz[(PH + 2) % 3] - Xz[(PH + 1) % 3]) * (Xz[(PH + 2) % 3] - Xz[(PH + 1) % 3]);
        }
        const double gZ = axis_godunov<true>(c0, e, f);
        // x and y: the lane's own point
        {
            const double P = weno_P(xp, c0, dx, rdx);
            wxo[0] = P, wxo[JSS_XN] = P * P;
            wxo[2 * JSS_XN] = weno_X(xp, c0, xm, dx, rdx);
            wxo[3 * JSS_XN] = weno_X(xm, c0, xp, dx, rdx);
        }
        {
            const double P = weno_P(yp, c0, dx, rdx);
            wyo[0] = P, wyo[JSS_YN] = P * P;
            wyo[2 * JSS_YN] = weno_X(yp, c0, ym, dx, rdx);
            wyo[3 * JSS_YN] = weno_X(ym, c0, yp, dx, rdx);
        }
        // the points beyond the patch
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (jon[t]) {
                // one instruction stream for the three kinds: far - 1 mid is far - mid, (far - 2 mid) + near the second differences
                const double u = jf[t] - jw[t] * jmid[t];
                const double v = div_dx(jp[t] ? u : u + jn[t], dx, rdx);
                jd[t][0] = v;
                if (jp[t]) jd[t][t == 0 ? JSS_XN : JSS_YN] = v * v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (weno_ok) {
            WenoDiffs D;
            double t0;
            // x
#pragma unroll
            for (int m = 0; m < 6; ++m) D.p[m] = wxo[m - 3], D.s[m] = wxo[JSS_XN + m - 3];
            D.cp = wxo[2 * JSS_XN], D.bp = wxo[2 * JSS_XN + 1], D.ap = wxo[2 * JSS_XN + 2];
            D.bm = wxo[3 * JSS_XN - 1], D.am = wxo[3 * JSS_XN - 2];
            D.t1p = 13. * (D.bp - D.cp) * (D.bp - D.cp);
            weno_from_diffs_strict(D, a, b, t0);
            // y (p5 = 0, subs.f90:576)
#pragma unroll
            for (int m = 0; m < 5; ++m) D.p[m] = wyo[(m - 3) * JSS_PX], D.s[m] = wyo[JSS_YN + (m - 3) * JSS_PX];
            D.p[5] = 0.0, D.s[5] = 0.0;
            D.cp = wyo[2 * JSS_YN], D.bp = wyo[2 * JSS_YN + JSS_PX], D.ap = wyo[2 * JSS_YN + 2 * JSS_PX];
            D.bm = wyo[3 * JSS_YN - JSS_PX], D.am = wyo[3 * JSS_YN - 2 * JSS_PX];
            D.t1p = 13. * (D.bp - D.cp) * (D.bp - D.cp);
            weno_from_diffs_strict(D, c, d, t0);
        } else {
            a = wxo[-1], b = wxo[0], c = wyo[-JSS_PX], d = wyo[0]; // subs.f90:657-660: P(i-1), P(i)
        }
        __builtin_amdgcn_wave_barrier();
        const double newv = finish_update<true>(c0, axis_godunov<true>(c0, a, b), axis_godunov<true>(c0, c, d), gZ, pS, dx, 0.0, h);
        if (cell) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 wv;
            wv.x = (unsigned)__double2loint(newv);
            wv.y = (unsigned)__double2hiint(newv);
            __builtin_amdgcn_raw_buffer_store_b64(wv, desc(Bout + sxy * k), col, 0, 0);
            const double dlt = newv - c0;
            acc = acc + dlt * dlt;
        }
    };
    for (int k = k0; k < k1; k += 6) {
        step(std::integral_constant<int, 0>{}, k);
        if (k + 1 >= k1) break;
        step(std::integral_constant<int, 1>{}, k + 1);
        if (k + 2 >= k1) break;
        step(std::integral_constant<int, 2>{}, k + 2);
        if (k + 3 >= k1) break;
        step(std::integral_constant<int, 3>{}, k + 3);
        if (k + 4 >= k1) break;
        step(std::integral_constant<int, 4>{}, k + 4);
        if (k + 5 >= k1) break;
        step(std::integral_constant<int, 5>{}, k + 5);
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < JSS_WX * JSS_WY; ++q) t += red[q];
        partials[L] = t;
    }
}

// =============================================================================================
// Reinit, Jacobi ordering, FAST arithmetic, WENO interfaces SHARED between neighbouring cells along x and z
// (lsf_cell.hpp: weno_iface_fast).  SURVEY.md section 0 fact 5: the sweep is bound by the fp64 vector unit, and the
// lever is to compute every quadratic form once -- D+ of cell i and D- of cell i+1 are the two outputs of ONE
// interface evaluation.
//   z (march axis): the interface k+1/2 evaluated at step k leaves D- of cell k+1 in a register for step k+1.
//   x (lanes):      lane i evaluates the interface i+1/2 and hands the D- correction to lane i+1 with one DPP
//                   wave_shr:1; the first lane of a wavefront takes it from the last lane of the wavefront to its left
//                   through LDS (double-buffered, one workgroup barrier per step).  A block spans 64 WX lanes along x;
//                   its first lane is a helper (it only evaluates an interface), so a block covers 64 WX - 1 cells:
//                   510 = 2 x 255 at 512^3 with WX = 4, 254 = 2 x 127 at 256^3 with WX = 2, 1022 = 2 x 511 with WX = 8.
//   y:              per-cell form (the p5 = 0 quirk of subs.f90:576 makes the two sides of a y interface differ; the rows
//                   of a block are BY apart wavefronts).
// Blocks are numbered so that the blocks an XCD receives (every 8th of the launch order) cover a contiguous range of rows:
// the y neighbours a block loads are then in its own L2.
// 66 + 66 + 90 operations for the three axes instead of 3 x 90.  Bit-identical to k_reinit_jacobi<false> (whose FAST
// arithmetic evaluates the same interfaces twice per cell), which stays for the 3-cell x rims (THINX).
// =============================================================================================
template <int WX, int BY>
__global__ __launch_bounds__(64 * WX * BY) __attribute__((amdgpu_waves_per_eu(5))) void k_reinit_jacobi_sh(const double* __restrict__ A, double* __restrict__ Bout,
                                                                     const double* __restrict__ phiS, Box bx, int lo0, int lo1,
                                                                     int lo2, int hi0, int hi1, int hi2, double dx, double h,
                                                                     double* __restrict__ partials, const int* __restrict__ done,
                                                                     int nbx, int nby, int nbz, int kc)
{
    constexpr int NW = WX * BY;
    __shared__ double red[NW];
    __shared__ double xch[2][BY][WX];
    if (done && *done) return;
    // XCD-aware numbering: launch index id -> logical block L; the 8 XCDs take ids round robin, XCD x gets the logical
    // blocks [x * per, (x + 1) * per)
    const unsigned per = gridDim.x >> 3;
    const unsigned L = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const unsigned nblk = (unsigned)nbx * nby * nbz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wx = wave % WX, ry = wave / WX;
    double acc = 0.0;
    if (L < nblk) {
        const int bxi = (int)(L % (unsigned)nbx), byi = (int)((L / (unsigned)nbx) % (unsigned)nby), bzi = (int)(L / ((unsigned)nbx * nby));
        const int xl = 64 * wx + lane;                         // lane position inside the block, 0 = helper
        const int li = lo0 - 1 + bxi * (64 * WX - 1) + xl;     // local x index of the lane's point (>= 0: lo0 >= 1)
        const int lj = lo1 + byi * BY + ry;
        const int k0 = lo2 + bzi * kc, k1 = min(k0 + kc, hi2);
        const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
        const bool cell = xl >= 1 && li < hi0 && lj < hi1;     // the lane owns a cell of this launch
        const int gi = li + bx.gx0, gj = lj + bx.gy0;
        const bool ij_weno = gi > 3 && gi < bx.nx - 4 && gj > 3 && gj < bx.ny - 4;
        // every lane of this wavefront takes the WENO branch as far as i and j go (six wavefronts of eight at 512^3): its
        // first-order fix-ups are then skipped by a scalar branch instead of being evaluated and selected away per lane
        const bool wave_ij_weno = __builtin_amdgcn_ballot_w64(ij_weno) == ~0ull;
        const double inv_dx = 1.0 / dx, floor2 = 1.E-99 * dx * dx / 13.0;
        // Addressing: one buffer descriptor per k-plane (rebuilt on the scalar unit every step, base = the plane) + 32-bit
        // byte offsets inside the plane that do not change along the march.  No load of the loop sits behind a branch:
        //   x: the six points li-2 .. li+3 are `colx` + immediates; the offset may leave the row or the end of the plane near
        //      a wall -- the load then returns the value of a neighbouring row or, past the plane, the descriptor's 0 -- and
        //      neither is used: a WENO cell has all six points in its row, a first-order cell reads li +- 1 only, which always
        //      exist.  (It never wraps below zero: lj >= 1, so col >= one row.  A wrapped offset is NOT safe: k_reinit_jacobi_strict_sh);
        //   y: six offsets with the row clamped into the plane (same argument).
        const unsigned plane_bytes = 8u * (unsigned)sxy;
        const int lic = min(li, bx.lx - 1), ljc = min(lj, bx.ly - 1);
        const unsigned col = 8u * (unsigned)(lic + sx * ljc);
        const unsigned colx = col - 16u;
        unsigned oy[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) oy[m] = 8u * (unsigned)(lic + sx * min(max(ljc - 3 + m, 0), bx.ly - 1));
        auto desc = [&](const double* base) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)plane_bytes, 0x00020000);
        };
        auto at = [](__amdgpu_buffer_rsrc_t r, unsigned boff) -> double {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, boff, 0, 0);
            return __hiloint2double((int)v.y, (int)v.x);
        };
        auto ldz = [&](int k) -> double { return at(desc(A + sxy * min(max(k, 0), bx.lz - 1)), col); };
        double qz[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) qz[m] = ldz(k0 - 4 + m); // shifted by one: the loop shifts before it uses the window
        // D- correction of the first cell of the chunk: interface k0 - 1/2, points k0-3 .. k0+2 (consumed by WENO cells
        // only, whose window is real data)
        double pwm_z;
        {
            double t0_, t1_;
            weno_iface_fast(qz + 1, floor2, t0_, pwm_z, t1_);
        }
        int pb = 0;
        for (int k = k0; k < k1; ++k) {
            const auto P = desc(A + sxy * k);
            // ---- x and z loads, z interface k + 1/2, then the y loads (in flight during the x interface)
            double vx[6], qy[7];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                if (m != 2) vx[m] = at(P, colx + 8u * (unsigned)m);
#pragma unroll
            for (int m = 0; m < 6; ++m) qz[m] = qz[m + 1];
            qz[6] = ldz(k + 3);
            const double phic = qz[3];
            vx[2] = phic;
            const int gk = k + bx.gz0;
            const bool weno_ok = ij_weno && gk > 3 && gk < bx.nz - 4;
            double pwp_z, pwm_z_next, cen_z, pwp_x, pwm_x, cen_x;
            weno_iface_fast(qz + 1, floor2, pwp_z, pwm_z_next, cen_z);
#pragma unroll
            for (int m = 0; m < 7; ++m)
                if (m != 3) qy[m] = at(P, oy[m]);
            qy[3] = phic;
            const double pS = at(desc(phiS + sxy * k), col);
            weno_iface_fast(vx, floor2, pwp_x, pwm_x, cen_x);
            // hand the D- correction to the lane on the right
            double pwm_l = dpp_shr1(pwm_x);
            if (WX > 1) {
                if (lane == 63) xch[pb][ry][wx] = pwm_x;
                // LDS hand-off only: no global access has to be visible across the barrier, so do not drain the loads
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (lane == 0 && wx > 0) pwm_l = xch[pb][ry][wx - 1];
                pb ^= 1;
            }
            // ---- y (per-cell form), Godunov, sign, Euler step
            double a, b, c, d, e, f;
            a = __builtin_fma(1.0 / 12.0, cen_x, -pwm_l);
            b = __builtin_fma(1.0 / 12.0, cen_x, pwp_x);
            e = __builtin_fma(1.0 / 12.0, cen_z, -pwm_z);
            f = __builtin_fma(1.0 / 12.0, cen_z, pwp_z);
            weno_axis_fast(qy, floor2, true, c, d);
            if (!(wave_ij_weno && gk > 3 && gk < bx.nz - 4)) { // wavefront-uniform: some lane is within three cells of a wall
                if (!weno_ok) {                                 // first-order one-sided differences (subs.f90:657-662)
                    a = phic - vx[1], b = vx[3] - phic;
                    c = phic - qy[2], d = qy[4] - phic;
                    e = phic - qz[2], f = qz[4] - phic;
                }
            }
            const double newv = finish_update<false>(phic, axis_godunov<false>(phic, a, b), axis_godunov<false>(phic, c, d),
                                                     axis_godunov<false>(phic, e, f), pS, dx, inv_dx, h);
            if (cell) {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                u2 w;
                w.x = (unsigned)__double2loint(newv);
                w.y = (unsigned)__double2hiint(newv);
                __builtin_amdgcn_raw_buffer_store_b64(w, desc(Bout + sxy * k), col, 0, 0);
                const double dlt = newv - phic;
                acc = __builtin_fma(dlt, dlt, acc);
            }
            pwm_z = pwm_z_next;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[w];
        partials[L] = t;
    }
}

// =============================================================================================
// Extrapolation boundary condition, closed form of subs.f90:859-897:
// wall point <- interior point clamp(i,1,n-1) + dx added m = min(nb, 1+nh) times in sequence.
// grid = (ceil(maxext/64), maxext, faces present); each wall point is owned by exactly one face.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(64) void k_bc(const T* __restrict__ A, T* __restrict__ Bout, Box bx,
                                           int lo0, int lo1, int lo2, int hi0, int hi1, int hi2, T dx,
                                           double* __restrict__ partials, const int* __restrict__ done,
                                           int skip_xface, unsigned faces)
{
    if (done && *done) return;
    // `faces`: the wall faces this region touches, 3 bits each (gridDim.z of them): a block of a decomposed field has
    // walls only where it has no neighbour
    const int face = (int)((faces >> (3 * blockIdx.z)) & 7u); // 0:i=0 1:i=nx 2:j=0 3:j=ny 4:k=0 5:k=nz
    const int u = blockIdx.x * 64 + threadIdx.x, v = blockIdx.y;
    const int axis = face >> 1;
    const int nwall[3] = {bx.nx, bx.ny, bx.nz};
    const int g0[3] = {bx.gx0, bx.gy0, bx.gz0};
    const int lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const int wall_g = (face & 1) ? nwall[axis] : 0;
    const int wall_l = wall_g - g0[axis];
    double contrib = 0.0;
    const int a1 = axis == 0 ? 1 : 0, a2 = axis == 2 ? 1 : 2; // the two in-face axes (ascending)
    int l[3];
    l[axis] = wall_l;
    l[a1] = lo[a1] + u;
    l[a2] = lo[a2] + v;
    if (wall_l >= lo[axis] && wall_l < hi[axis] && l[a1] < hi[a1] && l[a2] < hi[a2]) {
        const int gi = l[0] + bx.gx0, gj = l[1] + bx.gy0, gk = l[2] + bx.gz0;
        const bool wi = gi == 0 || gi == bx.nx, wj = gj == 0 || gj == bx.ny, wk = gk == 0 || gk == bx.nz;
        // ownership: x faces own everything on them; y faces skip points on x walls; z faces skip
        // points on x or y walls
        // skip_xface: the sweep kernel has written the pure x-face points (k_reinit_jacobi_f32, xwall)
        const bool own = (axis == 0 && !(skip_xface && !wj && !wk)) || (axis == 1 && !wi) || (axis == 2 && !wi && !wj);
        if (own) {
            const int nb = (int)wi + (int)wj + (int)wk;
            const int nh = (int)(gi == bx.nx) + (int)(gj == bx.ny) + (int)(gk == bx.nz);
            const int m = min(nb, 1 + nh);
            const int ci = min(max(gi, 1), bx.nx - 1) - bx.gx0;
            const int cj = min(max(gj, 1), bx.ny - 1) - bx.gy0;
            const int ck = min(max(gk, 1), bx.nz - 1) - bx.gz0;
            const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
            T val = Bout[ci + sx * cj + sxy * ck];
            for (int t = 0; t < m; ++t) val = val + dx;
            const long p = l[0] + sx * l[1] + sxy * l[2];
            const T dlt = val - A[p];
            Bout[p] = val;
            contrib = (double)(dlt * dlt);
        }
    }
    contrib = wave_sum(contrib);
    if (threadIdx.x == 0) partials[blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)] = contrib;
}

// =============================================================================================
// Sweep epilogue: fixed-order sum of the partials, RMS, stop test -- all on the device so that the
// host never has to synchronise per sweep (subs.f90:902-926 / set3d.f90:435-458).
// ctl[0]=done flag, ctl[1]=sweeps completed, ctl[2]=status (0 ok, 1 NaN)
// =============================================================================================
// fixed-order sum of partials[0..nPart) by one block of RED_T threads: thread t adds elements t, t+RED_T, ... in
// eight independent chains (the loads of a round are all in flight together: a serial chain of ~1 us loads is
// what made the first version of this kernel cost 50 us), then a tree over the block.  Result valid in thread 0.
constexpr int RED_T = 1024;
__device__ __forceinline__ double block_sum(const double* __restrict__ partials, long nPart, double* red)
{
    double t[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    long p = threadIdx.x;
    for (; p + 7L * RED_T < nPart; p += 8L * RED_T) {
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += partials[p + (long)u * RED_T];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (p + (long)u * RED_T < nPart) t[u] += partials[p + (long)u * RED_T];
    red[threadIdx.x] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    __syncthreads();
    for (int s = RED_T / 2; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

static __global__ __launch_bounds__(RED_T) void k_finish(const double* __restrict__ partials, long nPart, double den,
                                                  double tol, double* __restrict__ trace, int trace_cap,
                                                  int* __restrict__ ctl)
{
    __shared__ double red[RED_T];
    if (ctl[0]) return;
    const double tot = block_sum(partials, nPart, red);
    if (threadIdx.x == 0) {
        const double rms = __builtin_sqrt(tot / den);
        const int n = ctl[1];
        if (n < trace_cap) trace[n] = rms;
        ctl[1] = n + 1;
        if (rms < tol) ctl[0] = 1;
        else if (rms != rms) { ctl[0] = 1; ctl[2] = 1; }
    }
}

// adds the fixed-order sum of the partials to *acc (building block for the decomposed path)
static __global__ __launch_bounds__(RED_T) void k_accumulate(const double* __restrict__ partials, long nPart,
                                                      double* __restrict__ accum)
{
    __shared__ double red[RED_T];
    const double tot = block_sum(partials, nPart, red);
    if (threadIdx.x == 0) *accum += tot;
}

// Measurement aid (lsf_copy_bandwidth, bench.py "roofline.peak_measured"): a streaming copy -- the denominator SURVEY.md section 8d
// asks for next to the vendor's 8 TB/s ("also against a measured device-copy bandwidth").  One 16-byte vector per lane, non-temporal
// load and store, a block per 4 KB: the fastest of the forms tried on this chip (profiles/micro/copy_bw.hip, 1 GiB: 6.5 TB/s read +
// written; the same with plain accesses 6.2, grid-stride loops with four accesses in flight per lane 4.1-5.0, eight vectors per lane
// 4.2-4.3, hipMemcpyAsync 5.4; read only 5.7, write only 6.9).  Nothing of the hot path calls it.
static __global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load((const v4u*)src + i), (v4u*)dst + i);
}

// =============================================================================================
// narrowBand, subs.f90:178-207
// =============================================================================================
static __global__ __launch_bounds__(256) void k_narrowband(const double* __restrict__ phi, int32_t* __restrict__ nb,
                                                    int32_t* __restrict__ sb, long n, double dx)
{
    const double tn = 4.1 * dx, ts = 8.1 * dx;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += 256L * gridDim.x) {
        const double a = __builtin_fabs(phi[p]);
        nb[p] = a < tn ? 1 : 0;
        sb[p] = a < ts ? 1 : 0;
    }
}

// =============================================================================================
// Min/max flow.  A = phi at iteration start (frozen: Laplacian, RMS reference), B = result.
// The band of iteration n is narrowBand(A) (set3d.f90:460 ran on exactly that field) except for the
// first iteration, which uses the caller's mask (nbmask != nullptr).  Wall points are never
// updated (a band cell on a wall would read outside the array in the reference).
// =============================================================================================
// (i, j, k) of the linear point index p of a field with extents (sx, sy, .): 32-bit divisions when the field has fewer
// than 2^31 points (a 64-bit division by a run-time value is ~150 instructions: two of them per point made the min/max
// streaming kernels vector-bound at half the HBM rate)
__device__ __forceinline__ void point_ijk(long p, int sx, int sy, long n, int& i, int& j, int& k)
{
    if (n <= 0x7fffffffL) {
        const unsigned pu = (unsigned)p, q = pu / (unsigned)sx;
        i = (int)(pu - q * (unsigned)sx);
        k = (int)(q / (unsigned)sy);
        j = (int)(q - (unsigned)k * (unsigned)sy);
    } else {
        const long q = p / sx;
        i = (int)(p - q * sx);
        k = (int)(q / sy);
        j = (int)(q - (long)k * sy);
    }
}
// ... and of p + d, given those of p and the decomposition (di, dj, dk) of the step d (di < sx, dj < sy)
__device__ __forceinline__ void step_ijk(int& i, int& j, int& k, int di, int dj, int dk, int sx, int sy)
{
    i += di;
    if (i >= sx) i -= sx, ++j;
    j += dj;
    if (j >= sy) j -= sy, ++k;
    k += dk;
}

__device__ __forceinline__ bool in_band(const int32_t* nbmask, long g, double a, double dx)
{
    return nbmask ? nbmask[g] == 1 : __builtin_fabs(a) < 4.1 * dx;
}

// Jacobi ordering: pAve from A as well.
static __global__ __launch_bounds__(256) void k_minmax_jacobi(const double* __restrict__ A, double* __restrict__ Bout,
                                                       const int32_t* __restrict__ nbmask, int nx, int ny,
                                                       int nz, double dx, double h1,
                                                       double* __restrict__ partials,
                                                       const int* __restrict__ done)
{
    __shared__ double red[4];
    if (*done) return;
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1), n = sxy * (nz + 1);
    const double dxx = 1. / (dx * dx);
    double acc = 0.0;
    int i, j, k, di, dj, dk;
    point_ijk(blockIdx.x * 256L + threadIdx.x, nx + 1, ny + 1, n, i, j, k);
    point_ijk(256L * gridDim.x, nx + 1, ny + 1, n, di, dj, dk); // the grid stride (below n whenever the loop repeats)
    for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += 256L * gridDim.x, step_ijk(i, j, k, di, dj, dk, nx + 1, ny + 1)) {
        const double c = A[p];
        double out = c;
        const bool interior = i >= 1 && i <= nx - 1 && j >= 1 && j <= ny - 1 && k >= 1 && k <= nz - 1;
        if (interior && in_band(nbmask, p, c, dx)) {
            const double xm = A[p - 1], xp = A[p + 1], ym = A[p - sx], yp = A[p + sx], zm = A[p - sxy],
                         zp = A[p + sxy];
            const double curv = minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx);
            out = minmax_update(c, xm, xp, yp, ym, zp, zm, curv, h1);
            const double d = out - c;
            acc += d * d;
        }
        Bout[p] = out;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------
// Exact Gauss-Seidel ordering by fixed-point iteration (the fast exact path).
// In pass B of the reference (set3d.f90:417-431) a band cell depends on its already-visited
// neighbours (i-1, j-1, k-1) ONLY through the sign test pAve < 0 (subs.f90:473-481): the curvature
// is frozen.  The in-place result is therefore the unique fixed point X of
//     X(c) = A(c) + h1 * F(curv_A(c), sign(pAve(A(c), X(i-1), A(i+1), A(j+1), X(j-1), A(k+1), X(k-1))))
// (the dependency graph is acyclic).  Start from the Jacobi result and re-evaluate every band cell
// with the current X of its three upstream neighbours until a whole pass changes nothing.  A cell is
// final one pass after its upstream cells are final, whatever mixture of old/new values a racing
// read observes, so the passes may update in place; "no cell changed" certifies the fixed point.
// Sign flips need |pAve| ~ h1*|curv|: a handful of cells hugging phi = 0 at first (2-3 passes), but the
// flow itself breeds such cells (1024^3 two-sphere case: 45 000 chunks change in pass 1 of iteration 12 and
// the count falls ~4x per pass), so passes after the first visit only the chunks downstream of a change.
// Fixed point-to-chunk map (MM_SUB consecutive points per chunk) -> per-chunk band flags, stamps and RMS
// partials (deterministic).  The scan handles MM_CH points (4 chunks) per block.
// ---------------------------------------------------------------------------------------------
constexpr int MM_CH = 2048, MM_SUB = 512;

// PASS 0: scan (copy + Jacobi update + band flags), MM_CH points per block (nchunks counts those blocks).
// PASS 1: fix pass `epoch` of this call.  The first fix pass of an iteration (first != 0) re-evaluates every
//         band chunk; later ones only the chunks a change of the previous pass can reach: a cell that changes
//         stamps the chunks of its three downstream neighbours with epoch+1, and pass epoch+1 visits the
//         chunks whose stamp is >= epoch+1 (stamps only grow inside a call, so nothing is ever cleared).
// PASS 2: RMS partials of the band chunks.
// Passes 1 and 2 find their chunks 64 at a time (one flag per lane of wave 0 + ballot) instead of walking
// the flag array serially.
template <int PASS>
__global__ __launch_bounds__(256) void k_minmax_fp(const double* __restrict__ A, double* __restrict__ B,
                                                   const int32_t* __restrict__ nbmask, int nx, int ny, int nz,
                                                   double dx, double h1, int* __restrict__ blockflag,
                                                   int* __restrict__ stamp, long nchunks, int epoch, int first,
                                                   const int* __restrict__ changed_prev,
                                                   int* __restrict__ changed_cur, double* __restrict__ partials,
                                                   int* __restrict__ ctl)
{
    __shared__ double red[4];
    __shared__ int flag, flag4[MM_CH / MM_SUB];
    __shared__ unsigned long long todo;
    constexpr int CH = PASS == 0 ? MM_CH : MM_SUB; // points one trip of the loop below covers
    if (ctl[0]) return;
    if (PASS == 1 && changed_prev && *changed_prev == 0) return;
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1), n = sxy * (nz + 1);
    const double dxx = 1. / (dx * dx);
    const long round = PASS == 0 ? (long)gridDim.x : 64L * gridDim.x;
    // decomposition of the step of 256 points between a thread's consecutive points (rows shorter than 256 points: the
    // step spans whole rows; planes smaller than 256 points are handled by step_ijk's carries as long as 256 < sx * sy,
    // smaller fields take the division every time)
    const bool tiny = sxy <= 256;
    const int d256j = (int)(256 / sx), d256i = (int)(256 - (long)d256j * sx);
    for (long base = 0; base < nchunks; base += round) {
        unsigned long long m = 1ull;
        if (PASS != 0) {
            if (threadIdx.x < 64) {
                const long ch = base + blockIdx.x + (long)threadIdx.x * gridDim.x;
                bool need = ch < nchunks && blockflag[ch] != 0;
                if (PASS == 1 && !first) need = need && stamp[ch] >= epoch;
                const unsigned long long b = __ballot(need);
                if (threadIdx.x == 0) todo = b;
            }
            __syncthreads();
            m = todo;
        }
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const long chunk = base + blockIdx.x + (long)l * gridDim.x;
            if (PASS == 0 && chunk >= nchunks) break;
            if (threadIdx.x == 0) flag = 0;
            if (PASS == 0 && threadIdx.x < MM_CH / MM_SUB) flag4[threadIdx.x] = 0;
            __syncthreads();
            double acc = 0.0;
            int mine = 0;
            int i, j, k;
            point_ijk(chunk * CH + threadIdx.x, nx + 1, ny + 1, n, i, j, k);
            // the scan issues all its loads of A before it looks at any of them (the loop below ends early at the end of
            // the field, which would otherwise keep the compiler from moving a load above the previous point's branch)
            double cpre[CH / 256];
            if (PASS == 0) {
#pragma unroll
                for (int t = 0; t < CH / 256; ++t) cpre[t] = A[min(chunk * CH + t * 256 + threadIdx.x, n - 1)];
            }
#pragma unroll
            for (int t = 0; t < CH / 256; ++t) {
                const long p = chunk * CH + t * 256 + threadIdx.x;
                if (p >= n) break;
                if (t > 0) {
                    if (tiny) point_ijk(p, nx + 1, ny + 1, n, i, j, k);
                    else step_ijk(i, j, k, d256i, d256j, 0, nx + 1, ny + 1);
                }
                const double c = PASS == 0 ? cpre[t] : A[p];
                const bool interior = i >= 1 && i <= nx - 1 && j >= 1 && j <= ny - 1 && k >= 1 && k <= nz - 1;
                const bool band = interior && in_band(nbmask, p, c, dx);
                if (PASS == 0) {
                    double out = c;
                    if (band) {
                        const double xm = A[p - 1], xp = A[p + 1], ym = A[p - sx], yp = A[p + sx], zm = A[p - sxy],
                                     zp = A[p + sxy];
                        out = minmax_update(c, xm, xp, yp, ym, zp, zm, minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx), h1);
                        mine |= 1 << (t * 256 / MM_SUB);
                    }
                    B[p] = out;
                } else if (band) {
                    if (PASS == 1) {
                        const double xm = A[p - 1], xp = A[p + 1], ym = A[p - sx], yp = A[p + sx], zm = A[p - sxy],
                                     zp = A[p + sxy];
                        const double curv = minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx);
                        // upstream neighbours from the evolving field (B holds A wherever nothing was updated)
                        const double nv = minmax_update(c, B[p - 1], xp, yp, B[p - sx], zp, B[p - sxy], curv, h1);
                        const double cur = B[p];
                        if (!(nv == cur)) {
                            B[p] = nv;
                            mine = 1;
                            // the three cells that read this one: interior, hence inside the array
                            stamp[(p + 1) / MM_SUB] = epoch + 1;
                            stamp[(p + sx) / MM_SUB] = epoch + 1;
                            stamp[(p + sxy) / MM_SUB] = epoch + 1;
                        }
                    } else {
                        const double d = B[p] - c;
                        acc += d * d;
                    }
                }
            }
            if (PASS == 2) {
                acc = wave_sum(acc);
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
                __syncthreads();
                if (threadIdx.x == 0) partials[chunk] = red[0] + red[1] + red[2] + red[3];
            } else if (PASS == 0) {
#pragma unroll
                for (int sb = 0; sb < MM_CH / MM_SUB; ++sb)
                    if (mine & (1 << sb)) flag4[sb] = 1;
                __syncthreads();
                const long sub = chunk * (MM_CH / MM_SUB) + threadIdx.x;
                if (threadIdx.x < MM_CH / MM_SUB && sub * MM_SUB < n) {
                    blockflag[sub] = flag4[threadIdx.x];
                    partials[sub] = 0.0; // pass 2 overwrites the band chunks
                }
            } else {
                if (mine) flag = 1;
                __syncthreads();
                if (threadIdx.x == 0 && flag) atomicAdd(changed_cur, 1); // number of chunks this pass still changed
            }
            __syncthreads();
        }
        __syncthreads();
    }
    if (PASS == 2 && blockIdx.x == 0 && threadIdx.x == 0 && changed_prev) {
        // `first` = number of fix passes that were enqueued, changed_prev = the counter of the last of them
        const int* c0 = changed_prev - (first - 1);
        int used = 0;
        for (int f = 0; f < first; ++f) used += c0[f] != 0;
        atomicMax(ctl + 4, used);
        // the last allowed fix pass still changed something: the fixed point is not certified
        if (*changed_prev != 0) ctl[3] = 1;
    }
}

// deterministic two-stage reduction of many partials: block b sums its contiguous slice
static __global__ __launch_bounds__(256) void k_reduce_slices(const double* __restrict__ in, long n, double* __restrict__ out)
{
    __shared__ double red[256];
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = min(lo + per, n);
    double t = 0.0;
    for (long p = lo + threadIdx.x; p < hi; p += 256) t += in[p];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// Exact Gauss-Seidel ordering of set3d.f90:417-431 (always the (+,+,+) raster): tile = TA x 8 x 8
// POINTS anchored at point 0, same lane/skew mapping as the reinit kernel, halo 1.
// LDS: box [10][10][TA+2] of in-place values + [8][8][TA] frozen curvature.
template <int TA>
__global__ __launch_bounds__(64) void k_minmax_gs_plane(const double* __restrict__ A, double* __restrict__ B,
                                                        const int32_t* __restrict__ nbmask, int nx, int ny,
                                                        int nz, const uint32_t* __restrict__ tiles, int nTi,
                                                        int nTj, int nTk, double dx, double h1,
                                                        double* __restrict__ partials,
                                                        const int* __restrict__ done)
{
    constexpr int RA = TA + 2;
    __shared__ double box[10 * 10 * RA];
    __shared__ double curvs[64 * TA];
    __shared__ int anyband;
    if (*done) return;
    const int lane = threadIdx.x;
    const uint32_t packed = tiles[blockIdx.x];
    const int ti = packed & 0x3ff, tj = (packed >> 10) & 0x3ff, tk = (packed >> 20) & 0x3ff;
    const int i_lo = ti * TA, j_lo = tj * 8, k_lo = tk * 8;
    const int ni = min(TA, nx + 1 - i_lo), nj = min(8, ny + 1 - j_lo), nk = min(8, nz + 1 - k_lo);
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1);
    const double dxx = 1. / (dx * dx);
    if (lane == 0) anyband = 0;

    // frozen values of the whole box from A (corners are never read)
    for (int idx = lane; idx < 100 * RA; idx += 64) {
        const int x = idx % RA - 1, r = idx / RA, y = r % 10 - 1, z = r / 10 - 1;
        const int gi = i_lo + x, gj = j_lo + y, gk = k_lo + z;
        const bool ok = gi >= 0 && gi <= nx && gj >= 0 && gj <= ny && gk >= 0 && gk <= nz;
        box[idx] = ok ? A[gi + sx * gj + sxy * gk] : 0.0;
    }
    __syncthreads();
    // frozen curvature of the band cells of the tile (pass A, set3d.f90:399-414); NaN marks "not band"
    bool mine = false;
    for (int idx = lane; idx < 64 * TA; idx += 64) {
        const int x = idx % TA, yz = idx / TA, y = yz & 7, z = yz >> 3;
        const int gi = i_lo + x, gj = j_lo + y, gk = k_lo + z;
        double cv = __builtin_nan("");
        if (x < ni && y < nj && z < nk && gi >= 1 && gi <= nx - 1 && gj >= 1 && gj <= ny - 1 && gk >= 1 &&
            gk <= nz - 1) {
            const int o = (x + 1) + RA * ((y + 1) + 10 * (z + 1));
            const double cc = box[o];
            if (in_band(nbmask, gi + sx * gj + sxy * gk, cc, dx)) {
                cv = minmax_curv(cc, box[o + 1], box[o - 1], box[o + RA], box[o - RA], box[o + 10 * RA],
                                 box[o - 10 * RA], dxx);
                mine = true;
            }
        }
        curvs[idx] = cv;
    }
    if (mine) anyband = 1;
    __syncthreads();
    double acc = 0.0;
    if (anyband) {
        // replace the upstream (low-side) halo faces by the already-updated values from B
        auto refresh = [&](int x, int y, int z) {
            const int gi = i_lo + x, gj = j_lo + y, gk = k_lo + z;
            if (gi >= 0 && gi <= nx && gj >= 0 && gj <= ny && gk >= 0 && gk <= nz)
                box[(x + 1) + RA * ((y + 1) + 10 * (z + 1))] = B[gi + sx * gj + sxy * gk];
        };
        refresh(-1, lane & 7, lane >> 3);
        for (int idx = lane; idx < 8 * TA; idx += 64) {
            refresh(idx % TA, -1, idx / TA);
            refresh(idx % TA, idx / TA, -1);
        }
        __syncthreads();
        const int b = lane & 7, c = lane >> 3;
        const int nsteps = ni + nj + nk - 2;
        for (int s = 0; s < nsteps; ++s) {
            const int a = s - b - c;
            const bool active = b < nj && c < nk && a >= 0 && a < ni;
            double newv = 0.0;
            bool upd = false;
            int o = 0;
            if (active) {
                const double cv = curvs[a + TA * (b + 8 * c)];
                if (cv == cv) {
                    o = (a + 1) + RA * ((b + 1) + 10 * (c + 1));
                    const double cc = box[o];
                    newv = minmax_update(cc, box[o - 1], box[o + 1], box[o + RA], box[o - RA], box[o + 10 * RA],
                                         box[o - 10 * RA], cv, h1);
                    const double d = newv - cc;
                    acc += d * d;
                    upd = true;
                }
            }
            __syncthreads();
            if (upd) box[o] = newv;
            __syncthreads();
        }
    }
    for (int idx = lane; idx < 64 * TA; idx += 64) {
        const int x = idx % TA, yz = idx / TA, y = yz & 7, z = yz >> 3;
        if (x < ni && y < nj && z < nk)
            B[(long)(i_lo + x) + sx * (j_lo + y) + sxy * (k_lo + z)] = box[(x + 1) + RA * ((y + 1) + 10 * (z + 1))];
    }
    acc = wave_sum(acc);
    if (lane == 0) partials[ti + nTi * (tj + (long)nTj * tk)] = acc;
}

// =============================================================================================
// phi0: inside/outside initialisation, set3d.f90:196-268 (the step just before the hot path;
// SURVEY.md section 8f rank 1).  One thread per grid point of the search box [im,ip]x[jm,jp]x[km,kp]:
// brute-force nearest triangle CENTROID (first minimum wins, `dis < minD`, set3d.f90:232), sign of the
// triple product of the three vertex vectors (set3d.f90:242-258), smeared with phiSign(pS,dx,gM=1)
// (set3d.f90:260-264).  Centroids are staged through LDS in chunks; every operation is written as in
// the reference and contraction is off, so the result is bit-identical.
// cen: [nElem][3] centroids computed on the host exactly as set3d.f90:212-214 does.
// =============================================================================================
constexpr int PHI0_CHUNK = 1024;

static __global__ __launch_bounds__(256) void k_phi0(double* __restrict__ phi, int nx, int ny, int im, int ip, int jm,
                                              int jp, int km, int kp, double dx, double xlo0, double xlo1,
                                              double xlo2, const double* __restrict__ cen,
                                              const double* __restrict__ vtx, int nElem)
{
#pragma clang fp contract(off)
    __shared__ double sc[PHI0_CHUNK * 3];
    const int ei = ip - im + 1, ej = jp - jm + 1, ek = kp - km + 1;
    const long npts = (long)ei * ej * ek;
    const long t = blockIdx.x * 256L + threadIdx.x;
    const bool live = t < npts;
    const long tt = live ? t : 0;
    const int i = im + (int)(tt % ei), j = jm + (int)((tt / ei) % ej), k = km + (int)(tt / ((long)ei * ej));
    const double gX = xlo0 + i * dx, gY = xlo1 + j * dx, gZ = xlo2 + k * dx; // gridX, set3d.f90:168-170
    double minD = 100000.;
    int fN = 0;
    // The reference compares ROUNDED distances, `dis < minD` with dis = sqrt(d2) (set3d.f90:231-235), and keeps the first minimum.  The
    // square root is monotone, so with m2 = the smallest d2 seen so far (minD = min(100000, sqrt(m2))) a triangle with d2 >= m2 cannot
    // pass that test; only a triangle with d2 < m2 needs its square root taken and the reference's comparison made -- which may still
    // say "not smaller" (two different d2 can round to the same distance: the first one stays, as in the reference).  A lane's running
    // minimum improves ~ln(nElem) times, not nElem times: the correctly rounded fp64 square root (~25 instructions) leaves the loop.
    double m2 = __builtin_inf();
    for (int base = 0; base < nElem; base += PHI0_CHUNK) {
        const int cnt = min(PHI0_CHUNK, nElem - base);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt * 3; q += 256) sc[q] = cen[(long)base * 3 + q];
        __syncthreads();
        for (int n = 0; n < cnt; ++n) {
            const double pX = sc[3 * n], pY = sc[3 * n + 1], pZ = sc[3 * n + 2];
            const double d2 = (pX - gX) * (pX - gX) + (pY - gY) * (pY - gY) + (pZ - gZ) * (pZ - gZ);
            // (a wavefront-uniform branch and an argument the compiler cannot see through: otherwise it computes the square root for
            // every triangle in front of the branch and selects.  Four triangles per branch: the same time.)
            if (__builtin_amdgcn_ballot_w64(d2 < m2) != 0ull) {
                double arg = d2;
                asm volatile("" : "+v"(arg));
                if (d2 < m2) {
                    m2 = d2;
                    const double dis = __builtin_sqrt(arg);
                    if (dis < minD) {
                        minD = dis;
                        fN = base + n;
                    }
                }
            }
        }
    }
    if (!live) return;
    const double* v = vtx + (long)fN * 9; // the three vertices of triangle fN: surfX(n1,:), surfX(n2,:), surfX(n3,:)
    const double A1 = v[0] - gX, A2 = v[1] - gY, A3 = v[2] - gZ;
    const double B1 = v[3] - gX, B2 = v[4] - gY, B3 = v[5] - gZ;
    const double C1 = v[6] - gX, C2 = v[7] - gY, C3 = v[8] - gZ;
    const double pSx = A2 * B3 - A3 * B2;
    const double pSy = -(A1 * B3 - B1 * A3);
    const double pSz = A1 * B2 - B1 * A2;
    const double pS = -(pSx * C1 + pSy * C2 + pSz * C3);
    const double gM = 1.;
    phi[i + (long)(nx + 1) * (j + (long)(ny + 1) * k)] = pS / __builtin_sqrt(pS * pS + dx * dx * gM); // subs.f90:169
}

static __global__ __launch_bounds__(256) void k_fill(double* __restrict__ p, long n, double v)
{
    for (long q = blockIdx.x * 256L + threadIdx.x; q < n; q += 256L * gridDim.x) p[q] = v;
}

// =============================================================================================
// Post-smoothing gradients and surface-node advection, set3d.f90:464-501.
// k_firstderiv8: firstDeriv order 8 (subs.f90:309-347) on the stencil-band cells, 0 elsewhere; the y
// derivative uses phi(i,j+1,k) twice (subs.f90:346) and neighbours are addressed linearly, like the
// reference; reads outside the allocation (undefined there) yield 0.
// k_advect_nodes: one thread per surface node; setPhiSurf (subs.f90:1076-1166) + the move of
// set3d.f90:493-496, repeated until phiSurf <= 1e-13 or `iters` passes.  The reference re-interpolates ALL
// nodes after every single move (O(iter n^2)); a node's value depends on its own position only.
// Contraction off, IEEE division / sqrt: bit-identical.
// =============================================================================================
static __global__ __launch_bounds__(256) void k_firstderiv8(const double* __restrict__ phi, const int32_t* __restrict__ sb,
                                                     double* __restrict__ grad, int nx, int ny, int nz, double dx)
{
#pragma clang fp contract(off)
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1), n = sxy * (nz + 1);
    const double aa1 = 1. / 280., aa2 = -4. / 105., aa3 = 1. / 5., aa4 = -4. / 5., aa6 = 4. / 5, aa7 = -1. / 5.,
                 aa8 = 4. / 105., aa9 = -1. / 280.;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += 256L * gridDim.x) {
        double gx = 0., gy = 0., gz = 0.;
        if (sb[p] == 1) {
            auto L = [&](long off) -> double {
                const long q = p + off;
                return (q < 0 || q >= n) ? 0.0 : phi[q];
            };
            gx = (L(-4) * aa1 + L(-3) * aa2 + L(-2) * aa3 + L(-1) * aa4 + L(1) * aa6 + L(2) * aa7 + L(3) * aa8 +
                  L(4) * aa9) / dx;
            gy = (L(-4 * sx) * aa1 + L(-3 * sx) * aa2 + L(-2 * sx) * aa3 + L(-sx) * aa4 + L(sx) * aa6 + L(sx) * aa7 +
                  L(3 * sx) * aa8 + L(4 * sx) * aa9) / dx;
            gz = (L(-4 * sxy) * aa1 + L(-3 * sxy) * aa2 + L(-2 * sxy) * aa3 + L(-sxy) * aa4 + L(sxy) * aa6 +
                  L(2 * sxy) * aa7 + L(3 * sxy) * aa8 + L(4 * sxy) * aa9) / dx;
        }
        grad[p] = gx;
        grad[p + n] = gy;
        grad[p + 2 * n] = gz;
    }
}

__device__ __forceinline__ double interp_node(const double* __restrict__ phi, const double* __restrict__ grad, long sx,
                                              long sxy, long n, double dx, double lo0, double lo1, double lo2, double x,
                                              double y, double z, double g[3])
{
#pragma clang fp contract(off)
    const int i0 = (int)__builtin_floor((x - lo0) / dx), j0 = (int)__builtin_floor((y - lo1) / dx),
              k0 = (int)__builtin_floor((z - lo2) / dx);
    const double x0 = i0 * dx + lo0, y0 = j0 * dx + lo1, z0 = k0 * dx + lo2;
    const double x1 = (i0 + 1) * dx + lo0, y1 = (j0 + 1) * dx + lo1, z1 = (k0 + 1) * dx + lo2;
    const double xd = (x - x0) / (x1 - x0), yd = (y - y0) / (y1 - y0), zd = (z - z0) / (z1 - z0);
    long p = i0 + sx * j0 + sxy * k0;
    p = p < 0 ? 0 : (p > n - sxy - sx - 2 ? n - sxy - sx - 2 : p); // stay inside the allocation (defensive)
    double out[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const double* a = f == 0 ? phi : grad + (long)(f - 1) * n;
        const double c00 = a[p] * (1. - xd) + a[p + 1] * xd;
        const double c10 = a[p + sx] * (1. - xd) + a[p + sx + 1] * xd;
        const double c01 = a[p + sxy] * (1. - xd) + a[p + sxy + 1] * xd;
        const double c11 = a[p + sxy + sx] * (1. - xd) + a[p + sxy + sx + 1] * xd;
        const double c0 = c00 * (1. - yd) + c10 * yd;
        const double c1 = c01 * (1. - yd) + c11 * yd;
        out[f] = c0 * (1. - zd) + c1 * zd;
    }
    g[0] = -out[1], g[1] = -out[2], g[2] = -out[3];
    const double m2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
    if (m2 < 1.E-7) {
        g[0] = g[1] = g[2] = 0.;
    } else {
        const double m = __builtin_sqrt(m2);
        g[0] = g[0] / m, g[1] = g[1] / m, g[2] = g[2] / m;
    }
    return out[0];
}

static __global__ __launch_bounds__(64) void k_advect_nodes(const double* __restrict__ phi, const double* __restrict__ grad,
                                                     int nx, int ny, int nz, double dx, double lo0, double lo1,
                                                     double lo2, double* __restrict__ nodes, int nnode, int iters)
{
#pragma clang fp contract(off)
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1), n = sxy * (nz + 1);
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nnode) return;
    double x = nodes[t], y = nodes[t + nnode], z = nodes[t + 2L * nnode], g[3];
    double ps = interp_node(phi, grad, sx, sxy, n, dx, lo0, lo1, lo2, x, y, z, g);
    for (int it = 0; it < iters; ++it) {
        if (!(ps > 1E-13)) break;
        x = x + ps * g[0];
        y = y + ps * g[1];
        z = z + ps * g[2];
        ps = interp_node(phi, grad, sx, sxy, n, dx, lo0, lo1, lo2, x, y, z, g);
    }
    nodes[t] = x;
    nodes[t + nnode] = y;
    nodes[t + 2L * nnode] = z;
}

// =============================================================================================
// pack / unpack of a sub-box (halo slabs)
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_pack(const T* __restrict__ f, T* __restrict__ buf, Box bx,
                                              int lo0, int lo1, int lo2, int e0, int e1, int e2, int unpack,
                                              T* __restrict__ fw)
{
    const long n = (long)e0 * e1 * e2;
    const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
    int x, y, z, dxs, dys, dzs; // point_ijk / step_ijk: no 64-bit division per point (a slab is far below 2^31 points)
    point_ijk(blockIdx.x * 256L + threadIdx.x, e0, e1, n, x, y, z);
    point_ijk(256L * gridDim.x, e0, e1, n, dxs, dys, dzs);
    for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += 256L * gridDim.x, step_ijk(x, y, z, dxs, dys, dzs, e0, e1)) {
        const long g = (lo0 + x) + sx * (lo1 + y) + sxy * (lo2 + z);
        if (unpack) fw[g] = buf[p];
        else buf[p] = f[g];
    }
}

// Up to six sub-boxes (the face slabs of a block) in ONE launch: a decomposed sweep packs three to six slabs and unpacks as
// many -- at 131^3 per rank each of those launches is shorter than the gap between two launches.  bytes = sizeof(T).
// (The per-lane slab index into the argument struct compiles to scalar selects, not to a scratch copy: 24 VGPRs, ScratchSize 0 for
// both instances, hipcc -Rpass-analysis=kernel-resource-usage, round 5 -- ADVICE r4.)
struct PackRegs {
    int n;
    int lo[6][3], e[6][3];
    long start[7]; // prefix sums of the slabs' point counts
    void* buf[6];
};
template <typename T>
__global__ __launch_bounds__(256) void k_pack_multi(const T* __restrict__ f, T* __restrict__ fw, Box bx, PackRegs r, int unpack)
{
    const long sx = bx.lx, sxy = (long)bx.lx * bx.ly;
    const long total = r.start[r.n];
    for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += 256L * gridDim.x) {
        int q = 0;
#pragma unroll
        for (int k = 1; k < 6; ++k) q += (k < r.n && p >= r.start[k]) ? 1 : 0;
        const int i = (int)(p - r.start[q]);
        const int e0 = r.e[q][0], e01 = e0 * r.e[q][1]; // a slab is far below 2^31 points
        const int z = i / e01, rem = i - z * e01, y = rem / e0, x = rem - y * e0;
        const long g = (r.lo[q][0] + x) + sx * (r.lo[q][1] + y) + sxy * (r.lo[q][2] + z);
        T* b = (T*)r.buf[q];
        if (unpack) fw[g] = b[i];
        else b[i] = f[g];
    }
}

} // namespace lsf
