// lsf_skew.hpp -- exact Gauss-Seidel reinit on SKEWED tiles (slot-synchronous schedule).
//
// The box tiles of lsf_flow.hpp (TA x NY x 4 cells, marched by in-tile hyperplanes a + b + c = step) keep only
// TA / (TA + NY + 2) = 70 % of their lanes busy: the first and last steps of every tile are a ramp.  Here the ramp
// is cut off: a tile is the set of cells of one (NY x 4) row bundle whose SKEW coordinate
//     s = Fx + Fy + Fz        (F = cell index counted along the sweep direction, "frame" coordinates)
// lies in [TA m, TA m + TA).  Row (b, c) of the bundle then covers Fx = X0 - b - c + t, t = 0..TA-1, so every
// lane works on every step, and a tile takes TA steps instead of TA + NY + 2.
//
// Dependencies stay a product order: a cell needs NEW values from cells with smaller Fx, Fy or Fz only, those have
// a smaller s, so they sit in a tile (m', B', C') <= (m, B, C) componentwise; tiles are launched by hyperplanes
// P = m + B + C (one launch per time slot, reinit_slot_core).  Consecutive sweeps are spaced by the host so that a
// tile of sweep g+1 runs after every sweep-g tile holding a cell within stencil reach of it.
//
// LDS image, FRAME orientation: 74 rows (20 bundle rows, 24 y-halo rows, 30 z-halo rows) of TA + 6 entries; entry k
// of row (b', c') is the cell Fx = X0 - b' - c' + k - 3, so the stencil of the cell a lane works on at step t sits at
// the same entry 3 + t + d of the row d steps up or down the bundle.  Entries below 3 of a bundle row and every entry
// of an upstream halo row were computed earlier in this sweep (read from `out`), everything else is old (`in`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsf_flow.hpp"

namespace lsf {

template <int TA, int NY>
struct SkTile {
    static constexpr int NZ = 4;
    static constexpr int RA = TA + 6;
    static constexpr int NCORE = NZ * NY;  // bundle rows, r = c * NY + b
    static constexpr int NYH = 6 * NZ;     // y-halo rows, q = c * 6 + hy  (b' = hy - 3 or nj + hy - 3)
    static constexpr int NZH = 6 * NY;     // z-halo rows, q = hz * NY + b (c' = hz - 3 or nk + hz - 3)
    static constexpr int YH0 = NCORE * RA, ZH0 = (NCORE + NYH) * RA, PS0 = (NCORE + NYH + NZH) * RA;
    static constexpr int TOTAL = PS0 + NCORE * TA;
};

template <int TA, int NY, bool STRICT>
__global__ __launch_bounds__(64) void k_reinit_gs_skew(FlowArgs a)
{
    static_assert(TA == 16, "row-per-16-lanes loader");
    using T = SkTile<TA, NY>;
    constexpr int RA = T::RA;
    __shared__ double lds[T::TOTAL];
    const int lane = threadIdx.x;
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1);
    const double dx = a.dx, h = a.h;
    const double inv_dx = 1.0 / dx, floor2 = 1.E-99 * dx * dx / 13.0;
    const int ncol = a.nTj * a.nTk;

    const int bx = (int)blockIdx.x;
    const int seg = (bx >= a.seg_end[0]) + (bx >= a.seg_end[1]) + (bx >= a.seg_end[2]);
    const uint32_t packed = a.seg_tiles[seg][bx - (seg ? a.seg_end[seg - 1] : 0)];
    const int g = a.seg_g[seg];
    const int si = a.seg_sign[seg][0], sj = a.seg_sign[seg][1], sk = a.seg_sign[seg][2];
    if (ld_flag(a.ctl + 0) != 0) return; // converged or failed in an earlier launch

    const int m = packed & 0x3ff, fB = (packed >> 10) & 0x3ff, fC = (packed >> 20) & 0x3ff;
    const int tj = sj > 0 ? fB : a.nTj - 1 - fB, tk = sk > 0 ? fC : a.nTk - 1 - fC;
    const int j_lo = 1 + tj * NY, k_lo = 1 + tk * 4;
    const int nj = min(NY, ny - j_lo), nk = min(4, nz - k_lo);
    const int nxi = nx - 1;                       // interior cells along x
    const int X0 = TA * m - NY * fB - 4 * fC;     // Fx of row (0,0) at step 0
    const int gb = g % a.nbuf;
    const double* in = a.buf[gb];
    double* out = a.buf[gb + 1 == a.nbuf ? 0 : gb + 1];
    const long dOI = out - in;

    // frame row (b', c') entry k -> global address; `fresh` = computed earlier in this sweep
    auto cell_ptr = [&](int bq, int cq, int k, bool fresh) -> const double* {
        const int fx = X0 - bq - cq + k - 3;
        const int gi_r = si > 0 ? 1 + fx : nx - 1 - fx;
        const int gj_r = j_lo + (sj > 0 ? bq : nj - 1 - bq);
        const int gk_r = k_lo + (sk > 0 ? cq : nk - 1 - cq);
        const bool interior = gi_r >= 1 && gi_r <= nx - 1 && gj_r >= 1 && gj_r <= ny - 1 && gk_r >= 1 && gk_r <= nz - 1;
        const int gi = min(max(gi_r, 0), nx), gj = min(max(gj_r, 0), ny), gk = min(max(gk_r, 0), nz);
        return in + (gi + sx * gj + sxy * gk) + ((fresh && interior) ? dOI : 0);
    };

    // ---- load: all global loads in flight before the first LDS write ------------------------------------
    {
        constexpr int U0 = T::NCORE / 4, U1 = T::NYH / 4, U2 = (T::NZH + 3) / 4; // 16 main entries of 4 rows per instruction
        constexpr int H0 = (6 * T::NCORE + 63) / 64, H1 = (6 * T::NYH + 63) / 64, H2 = (6 * T::NZH + 63) / 64;
        constexpr int NV = U0 + U1 + U2 + U0 + H0 + H1 + H2;
        static_assert(T::NCORE % 4 == 0 && T::NYH % 4 == 0, "row groups");
        double v[NV];
        int dst[NV];
        const int xx = lane & 15, rsub = lane >> 4;
        int n_ = 0;
#pragma unroll
        for (int u = 0; u < U0; ++u, ++n_) { // bundle rows, entries 3..18: old
            const int r = 4 * u + rsub, c = r / NY, b = r - NY * c;
            dst[n_] = r * RA + 3 + xx;
            v[n_] = *cell_ptr(b, c, 3 + xx, false);
        }
#pragma unroll
        for (int u = 0; u < U1; ++u, ++n_) { // y halo
            const int q = 4 * u + rsub, c = q / 6, hy = q - 6 * c;
            dst[n_] = T::YH0 + q * RA + 3 + xx;
            v[n_] = *cell_ptr(hy < 3 ? hy - 3 : nj + hy - 3, c, 3 + xx, hy < 3);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u, ++n_) { // z halo
            const int q = min(4 * u + rsub, T::NZH - 1), hz = q / NY, b = q - NY * hz;
            dst[n_] = T::ZH0 + q * RA + 3 + xx;
            v[n_] = *cell_ptr(b, hz < 3 ? hz - 3 : nk + hz - 3, 3 + xx, hz < 3);
        }
#pragma unroll
        for (int u = 0; u < U0; ++u, ++n_) { // phiS of the bundle cells
            const int r = 4 * u + rsub, c = r / NY, b = r - NY * c;
            dst[n_] = T::PS0 + r * TA + xx;
            v[n_] = *(a.phiS + (cell_ptr(b, c, 3 + xx, false) - in));
        }
        // the 6 window-halo entries of every row: k = 0..2 (earlier tile of the row: fresh for bundle rows) and 19..21
#pragma unroll
        for (int u = 0; u < H0; ++u, ++n_) {
            const int idx = min(lane + 64 * u, 6 * T::NCORE - 1), r = idx / 6, ee = idx - 6 * r;
            const int c = r / NY, b = r - NY * c, k = ee < 3 ? ee : TA + ee;
            dst[n_] = r * RA + k;
            v[n_] = *cell_ptr(b, c, k, ee < 3);
        }
#pragma unroll
        for (int u = 0; u < H1; ++u, ++n_) {
            const int idx = min(lane + 64 * u, 6 * T::NYH - 1), q = idx / 6, ee = idx - 6 * q;
            const int c = q / 6, hy = q - 6 * c, k = ee < 3 ? ee : TA + ee;
            dst[n_] = T::YH0 + q * RA + k;
            v[n_] = *cell_ptr(hy < 3 ? hy - 3 : nj + hy - 3, c, k, hy < 3);
        }
#pragma unroll
        for (int u = 0; u < H2; ++u, ++n_) {
            const int idx = min(lane + 64 * u, 6 * T::NZH - 1), q = idx / 6, ee = idx - 6 * q;
            const int hz = q / NY, b = q - NY * hz, k = ee < 3 ? ee : TA + ee;
            dst[n_] = T::ZH0 + q * RA + k;
            v[n_] = *cell_ptr(b, hz < 3 ? hz - 3 : nk + hz - 3, k, hz < 3);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) lds[dst[u]] = v[u]; // duplicates (clamped indices) rewrite the same value
    }
    __syncthreads();

    // ---- per-lane constants: 3 lanes per cell inside each 16-lane row (5 cells + 1 idle lane) -------------
    static_assert(NY == 5, "lane map: 5 cells x 3 axes per 16 lanes");
    const int t16 = lane & 15;
    const int b = t16 / 3, axis = t16 - 3 * b, c = lane >> 4; // t16 = 15: b = 5 >= nj, idle
    const bool row_ok = b < nj && c < nk;
    const int bc = row_ok ? b : 0, cc = row_ok ? c : 0;
    const int gj = j_lo + (sj > 0 ? bc : nj - 1 - bc), gk = k_lo + (sk > 0 ? cc : nk - 1 - cc);
    const bool yz_weno = gj > 3 && gj < ny - 4 && gk > 3 && gk < nz - 4;
    const bool yquirk = axis == 1;
    const int row_core = (cc * NY + bc) * RA + 3;
    int off[7];
#pragma unroll
    for (int mm = 0; mm < 7; ++mm) {
        const int dA = mm - 3; // absolute offset along the lane's axis
        const int dF = (axis == 0 ? si : (axis == 1 ? sj : sk)) > 0 ? dA : -dA;
        int base = row_core - 3;
        if (axis == 1) {
            const int bq = bc + dF;
            base = (bq >= 0 && bq < nj) ? (cc * NY + bq) * RA : T::YH0 + (cc * 6 + (bq < 0 ? bq + 3 : bq - nj + 3)) * RA;
        } else if (axis == 2) {
            const int cq = cc + dF;
            base = (cq >= 0 && cq < nk) ? (cq * NY + bc) * RA : T::ZH0 + ((cq < 0 ? cq + 3 : cq - nk + 3) * NY + bc) * RA;
        }
        off[mm] = base + 3 + dF;
    }
    const int ps_row = T::PS0 + (cc * NY + bc) * TA;
    const int fx0 = X0 - bc - cc;
    double acc = 0.0;

    // ---- march: TA steps, every lane busy ------------------------------------------------------------------
    for (int t = 0; t < TA; ++t) {
        const int fx = fx0 + t;
        const bool active = row_ok && fx >= 0 && fx < nxi;
        double q[7];
#pragma unroll
        for (int mm = 0; mm < 7; ++mm) q[mm] = lds[off[mm] + t];
        const double pS = lds[ps_row + t];
        const int gi = si > 0 ? 1 + fx : nx - 1 - fx;
        const bool weno_ok = yz_weno && gi > 3 && gi < nx - 4;
        double dm, dp;
        axis_pair<STRICT>(q, weno_ok, yquirk, dx, floor2, dm, dp);
        const double gg = axis_godunov<STRICT>(q[3], dm, dp);
        const double gX = gg, gY = dpp_mov<0x101>(gg), gZ = dpp_mov<0x102>(gg); // row_shl:1, row_shl:2
        const double newv = finish_update<STRICT>(q[3], gX, gY, gZ, pS, dx, inv_dx, h);
        if (active && axis == 0) {
            lds[row_core + t] = newv;
            const double dlt = newv - q[3];
            acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
        }
        __syncthreads();
    }

    // ---- write back, fused extrapolation BC (subs.f90:859-897, closed form) --------------------------------
    const int fx_min = X0 - (nj - 1) - (nk - 1), fx_max = X0 + TA - 1;
    const bool near_wall = j_lo == 1 || j_lo + nj == ny || k_lo == 1 || k_lo + nk == nz ||
                           (fx_min <= 0 && fx_max >= 0) || (fx_min <= nxi - 1 && fx_max >= nxi - 1);
#pragma unroll
    for (int u = 0; u < T::NCORE / 4; ++u) {
        const int r = 4 * u + (lane >> 4), cq = r / NY, bq = r - NY * cq, t = lane & 15;
        const int fx = X0 - bq - cq + t;
        const bool mine = bq < nj && cq < nk && fx >= 0 && fx < nxi;
        const int gi = si > 0 ? 1 + fx : nx - 1 - fx;
        const int gj2 = j_lo + (sj > 0 ? bq : nj - 1 - bq), gk2 = k_lo + (sk > 0 ? cq : nk - 1 - cq);
        const double val0 = lds[r * RA + 3 + t];
        if (mine) out[gi + sx * gj2 + sxy * gk2] = val0;
        if (near_wall && mine) {
            // wall points that clamp to this cell: move outward along any non-empty subset of its wall-adjacent axes
            const int ai = gi == 1 ? -1 : (gi == nx - 1 ? 1 : 0), aj = gj2 == 1 ? -1 : (gj2 == ny - 1 ? 1 : 0),
                      ak = gk2 == 1 ? -1 : (gk2 == nz - 1 ? 1 : 0);
            if (ai | aj | ak) {
                for (int sub = 1; sub < 8; ++sub) {
                    if (((sub & 1) && !ai) || ((sub & 2) && !aj) || ((sub & 4) && !ak)) continue;
                    const int wi = gi + ((sub & 1) ? ai : 0), wj = gj2 + ((sub & 2) ? aj : 0), wk = gk2 + ((sub & 4) ? ak : 0);
                    const int nb = __builtin_popcount(sub);
                    const int nh = (int)(wi == nx) + (int)(wj == ny) + (int)(wk == nz);
                    const int mrep = min(nb, 1 + nh);
                    double val = val0;
                    {
#pragma clang fp contract(off)
                        for (int rr = 0; rr < mrep; ++rr) val = val + dx;
                    }
                    const long p = wi + sx * wj + sxy * wk;
                    const double dlt = val - in[p];
                    out[p] = val;
                    acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
                }
            }
        }
    }
    acc = wave_sum(acc);

    // ---- RMS: per-bundle running sum along m (deterministic), epilogue by the last tile of the sweep --------
    if (lane == 0) {
        double* slot = a.colsum + (long)gb * ncol + (tj + (long)a.nTj * tk);
        const int m_lo = (NY * fB + 4 * fC) / TA;
        *slot = ((m == m_lo) ? 0.0 : *slot) + acc;
    }
    if (packed != a.last_packed) return;
    __syncthreads();
    const double* cs = a.colsum + (long)gb * ncol;
    double tsum = 0.0;
    for (int p = lane; p < ncol; p += 64) tsum += cs[p];
    tsum = wave_sum(tsum);
    if (lane == 0) {
        const double rms = __builtin_sqrt(tsum / a.den);
        if (g < a.trace_cap) a.trace[g] = rms;
        st_flag(a.ctl + 1, g + 1);
        if (rms < a.tol) st_flag(a.ctl + 0, 1);
        else if (rms != rms) { st_flag(a.ctl + 2, 1); st_flag(a.ctl + 0, 1); }
    }
}

} // namespace lsf
