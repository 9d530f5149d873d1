// lsf_boxtile.hpp -- exact Gauss-Seidel reinit on box tiles (TA x NY x 4 cells, one wavefront each) and the argument block
// shared with the skewed-tile kernels of lsf_skew.hpp (the product path: LSF_GS_SCHEDULE unset / dataflow / skew).
// The box tiles are the first exact executor of this library; they stay as an independent second implementation of the
// same sweep (LSF_GS_SCHEDULE=slots / planes), used by the parity tests and on grids with fewer than two interior cells
// on an axis.
//
// One block per task of one time slot: every predecessor of a tile ran in an earlier launch (reinit_slot_core), so
// there is nothing to wait for and plain loads / stores suffice (kernel boundaries order them).  Sweep g reads
// buf[g % nbuf] ("old" values, and the walls) and buf[(g + 1) % nbuf] (this sweep's values of upstream cells) and writes
// buf[(g + 1) % nbuf].
//
// The extrapolation BC (subs.f90:859-897) is fused: a wall point is written by the tile that owns the interior
// cell it clamps to (its value depends on that cell only), and its RMS contribution is added there.
// RMS partials are accumulated per (tj,tk) tile column along the dependency chain (deterministic); the far-corner tile,
// alone on the last hyperplane of its sweep, reduces the columns, writes the trace and applies the stop / NaN test
// (subs.f90:902-926).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsf_kernels.hpp"

namespace lsf {

// dataflow launch: sweeps per batch (the sweep index s of a task shares a word with its hyperplane P)
// sweeps per dataflow launch (a launch costs about one sweep time of fill and drain -- at 256^3, where a sweep is a chain of
// dependent tiles, 2.5 ms against 0.65 ms per sweep: 6 % at 64 sweeps per launch, 1.5 % at 256)
#ifndef LSF_DF_SWEEP_BITS
#define LSF_DF_SWEEP_BITS 8
#endif
constexpr int DF_SWEEP_BITS = LSF_DF_SWEEP_BITS, DF_BATCH = 1 << DF_SWEEP_BITS;

struct GsArgs {
    double* buf[4];     // sweep g reads buf[g % nbuf] and writes buf[(g + 1) % nbuf]
    int nbuf;           // 3 or 4 buffers in rotation: sweep g overwrites the result of sweep g - nbuf (reinit_slot_core)
    int quirk_axis;     // kernel axis that carries the reference's p5 = 0 quirk (subs.f90:576): 1 (y), or 0 when the
                        // library runs the sweep on the x <-> y transposed field (lsf_skew.hpp, "march axis")
    const double* phiS;
    int nx, ny, nz, nTi, nTj, nTk;
    double dx, h;
    const uint2* order; // dataflow launch: task list in slot order, {packed tile, s | P << DF_SWEEP_BITS}
    long total;
    int nsweeps;        // sweeps in this batch
    int g0;             // global index of the first sweep of the batch
    int* ticket;        // dataflow launch: task counter
    double* colsum;     // [nbuf][nTj*nTk]
    double* trace;
    int trace_cap;
    double den, tol;
    int* ctl;           // [0] done, [1] sweeps completed, [2] status (1 NaN, 2 timeout), [3] unused, [4] INT_MAX (dataflow launches)
    long nTiles;
    // slot launches (dependencies resolved by launch order): up to 4 tile-plane segments, one per sweep in
    // flight; seg_end[] = running block count (segment q holds the blocks seg_end[q-1] <= blockIdx.x < seg_end[q])
    const uint32_t* seg_tiles[4];
    int seg_end[4];
    int seg_g[4];       // global sweep index of each segment
    int seg_sign[4][3];
    uint32_t last_packed; // skewed tiles (lsf_skew.hpp): the tile that runs the sweep epilogue
    // dataflow launch on skewed tiles (k_reinit_gs_persist)
    int np;                 // hyperplanes per sweep
    int* tile_done;         // [nsweeps][nM * nTj * nTk] 1 once tile (m, B, C) of the sweep is done
    int nM;                 // row length of tile_done
    int* plane_cnt;         // [nsweeps][np] tiles finished
    int* planes_done;       // [nsweeps] leading hyperplanes complete; np + 1 once the epilogue has run
    const int* plane_size;  // [np] tiles per hyperplane
    const int* sweep_tab;   // [nsweeps][4] {sign i, sign j, sign k, spacing in hyperplanes behind sweep s - 1}
    unsigned long long timeout_ticks; // bound of every spin of the dataflow launch (100 MHz ticks)
    const uint32_t* tables;  // skewed tiles: lookup tables of the tile shape (SkTile: rel_tab | off_tab), or NULL
    unsigned long long* dbg; // optional phase timers (s_memrealtime ticks): wait, load, march, publish, tasks
#ifdef LSF_EXPERIMENTS       // probes of the experiment builds (profiles/micro); the product's argument block does not carry them
    int probe_us;            // k_reinit_gs_skew, LSF_PROBE_STAGGER: delay of the second block of a CU
    int probe_early;         // k_reinit_gs_persist, LSF_PROBE_EARLY_FLAG: marching step in front of which a tile raises its flag
#endif
    // exact ordering across z slabs (k_reinit_gs_slab, lsf_gs_slabs.hpp): this launch owns the tile columns tk_lo <= tk < tk_hi of
    // the same global tile graph; every slab holds field buffers with the global address map, of which its own planes and the
    // three planes beyond each cut are kept current (the neighbour stores them there)
    int slab, nslab, tk_lo, tk_hi;
    int* nb_tile_done[2];          // tile_done of the lower / upper neighbour slab (same indexing), NULL at the ends of the grid
    int* nb_pd[2];                 // the row of the neighbour's planes_done mirror that follows THIS slab
    const int* pd_of_nb[2];        // this slab's mirror of the neighbour's planes_done (unused where nb_pd is NULL)
    int* verdict;                  // [nsweeps] 0 = unknown, 1 = go on, 2 = stop: condition (c) across slabs
    const int* plane_size_neg;     // tiles of this slab per hyperplane in sweeps that run against z (plane_size: along z)
    const struct SlabPeers* peers; // the other slabs' addresses (device memory: 56 pointers in the kernel's arguments are 112
                                   // scalar registers the compiler keeps live across the ticket loop -- measured: 73 spilled)
};
struct SlabPeers {
    double* nb_buf[2][4];          // field buffers of the lower / upper neighbour slab, NULL at the ends of the grid
    int* all_ctl[8];               // ctl / verdict / trace / colsum of every slab, this one included (set by the sweep epilogue
    int* all_verdict[8];           // resp. by the tiles of the slabs that do not run it)
    double* all_trace[8];
    double* all_colsum[8];
};

__device__ __forceinline__ double ld_sc1(const double* p)
{
    return __longlong_as_double(__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double* p, double v)
{
    __hip_atomic_store((long long*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// system scope: the other side of the access may be another device (slabs of the exact ordering)
__device__ __forceinline__ double ld_sys(const double* p)
{
    return __longlong_as_double(__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_sys(double* p, double v)
{
    __hip_atomic_store((long long*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#ifndef LSF_SYS_SCOPE
#define LSF_SYS_SCOPE __HIP_MEMORY_SCOPE_SYSTEM // (experiment builds: agent)
#endif
__device__ __forceinline__ int ld_flag_sys(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, LSF_SYS_SCOPE);
}
__device__ __forceinline__ void st_flag_sys(int* p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, LSF_SYS_SCOPE);
}
#ifndef LSF_FLAG_LD_SCOPE
#define LSF_FLAG_LD_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#ifndef LSF_FLAG_ST_SCOPE
#define LSF_FLAG_ST_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ int ld_flag(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, LSF_FLAG_LD_SCOPE);
}
__device__ __forceinline__ void st_flag(int* p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, LSF_FLAG_ST_SCOPE);
}

// Tile geometry: TA cells along i, NY x 4 cells in the (j,k) cross-section.
//   NY = 4: 16 cells per wave, 4 lanes per cell (x, y, z, idle), Godunov terms exchanged by DPP quad_perm
//   NY = 5: 20 cells per wave, 3 lanes per cell inside each 16-lane row (5 cells + 1 idle lane),
//           Godunov terms gathered at the x lane by DPP row_shl:1 / row_shl:2
// LDS image (doubles), ABSOLUTE orientation, star shaped (faces only), x is the unit-stride index everywhere:
//   core [4][NY][TA+6], y-halo [4][6][TA], z-halo [6][NY][TA], phiS [4][NY][TA]
template <int TA, int NY>
struct GsTile {
    static constexpr int RA = TA + 6;
    static constexpr int NCORE = 4 * NY;          // rows
    static constexpr int CORE = NCORE * RA;
    static constexpr int YH = 4 * 6 * TA;
    static constexpr int ZH = 6 * NY * TA;
    static constexpr int PS = NCORE * TA;
    static constexpr int TOTAL = CORE + YH + ZH + PS;
    static constexpr int NROWS = NCORE + 24 + 6 * NY + NCORE; // rows of TA doubles to load
    static constexpr int R1 = NCORE, R2 = NCORE + 24, R3 = NCORE + 24 + 6 * NY;
};

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    // bound_ctrl: lanes the shift leaves without a source read 0 (they are never the x lane of a cell), so the
    // destination needs no copy of the source first
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr unsigned long long FLOW_TIMEOUT_TICKS = 400000000ull; // 4 s of the 100 MHz s_memrealtime clock


template <int TA, int NY, bool STRICT>
__global__ __launch_bounds__(64) void k_reinit_gs_box(GsArgs a)
{
    using T = GsTile<TA, NY>;
    __shared__ double lds[T::TOTAL];
    const int lane = threadIdx.x;
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1);
    const double dx = a.dx, h = a.h;
    const double inv_dx = 1.0 / dx, floor2 = 1.E-99 * dx * dx / 13.0;
    const int ncol = a.nTj * a.nTk;

    // ---- the task of this block ------------------------------------------------------------------------
    const int bx = (int)blockIdx.x;
    const int seg = (bx >= a.seg_end[0]) + (bx >= a.seg_end[1]) + (bx >= a.seg_end[2]);
    const uint32_t packed = a.seg_tiles[seg][bx - (seg ? a.seg_end[seg - 1] : 0)];
    const int g = a.seg_g[seg];
    const int si = a.seg_sign[seg][0], sj = a.seg_sign[seg][1], sk = a.seg_sign[seg][2];
    const int fA = packed & 0x3ff, fB = (packed >> 10) & 0x3ff, fC = (packed >> 20) & 0x3ff;
    const int ti = si > 0 ? fA : a.nTi - 1 - fA;
    const int tj = sj > 0 ? fB : a.nTj - 1 - fB;
    const int tk = sk > 0 ? fC : a.nTk - 1 - fC;
    if (ld_flag(a.ctl + 0) != 0) return; // converged or failed in an earlier launch

    const int gb = g % a.nbuf;
    const double* in = a.buf[gb];
    double* out = a.buf[gb + 1 == a.nbuf ? 0 : gb + 1];
    const long dOI = out - in; // element offset that turns an `in` address into an `out` address
    const int i_lo = 1 + ti * TA, j_lo = 1 + tj * NY, k_lo = 1 + tk * 4;
    const int ni = min(TA, nx - i_lo), nj = min(NY, ny - j_lo), nk = min(4, nz - k_lo);
    double* core = lds;

    // ---- load: upstream interior cells from `out`, everything else from `in` --------------------------------
    {
        // Four row segments (core, y halo, z halo, phiS), each enumerated in whole wave instructions
        // (64/TA rows each; a segment whose row count is not a multiple repeats its last row, harmless):
        // the segment of every load is known at compile time, addresses are clamped into the array
        // (clamped entries are never consumed), no branches, all loads in flight before the first LDS write.
        constexpr int RPI = 64 / TA;
        constexpr int U0 = (T::NCORE + RPI - 1) / RPI, U1 = (24 + RPI - 1) / RPI, U2 = (6 * NY + RPI - 1) / RPI;
        constexpr int NROW = U0 + U1 + U2 + U0;
        const int xx = lane & (TA - 1), rsub = lane / TA;
        const int gi = min(i_lo + xx, nx);
        const bool gi_int = gi <= nx - 1;
        double v[NROW];
        int dst[NROW];
        auto ld_row = [&](int u, int gj, int gk, bool up, bool from_phis) {
            const bool row_int = gj >= 1 && gj <= ny - 1 && gk >= 1 && gk <= nz - 1;
            gj = min(max(gj, 0), ny), gk = min(max(gk, 0), nz);
            const long off = gi + sx * gj + sxy * gk;
            v[u] = from_phis ? a.phiS[off] : in[off + ((up && row_int && gi_int) ? dOI : 0)];
        };
#pragma unroll
        for (int u = 0; u < U0; ++u) { // core rows, own cells: old values
            const int r = min(u * RPI + rsub, T::NCORE - 1), zz = r / NY, yy = r - NY * zz;
            dst[u] = r * T::RA + 3 + xx;
            ld_row(u, j_lo + yy, k_lo + zz, false, false);
        }
#pragma unroll
        for (int u = 0; u < U1; ++u) { // y halo
            const int q = min(u * RPI + rsub, 23), zz = q / 6, hy = q - 6 * zz;
            dst[U0 + u] = T::CORE + q * TA + xx;
            ld_row(U0 + u, j_lo + (hy < 3 ? hy - 3 : nj + hy - 3), k_lo + zz, (hy < 3) == (sj > 0), false);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) { // z halo
            const int q = min(u * RPI + rsub, 6 * NY - 1), hz = q / NY, yy = q - NY * hz;
            dst[U0 + U1 + u] = T::CORE + T::YH + q * TA + xx;
            ld_row(U0 + U1 + u, j_lo + yy, k_lo + (hz < 3 ? hz - 3 : nk + hz - 3), (hz < 3) == (sk > 0), false);
        }
#pragma unroll
        for (int u = 0; u < U0; ++u) { // phiS of the own cells
            const int q = min(u * RPI + rsub, T::NCORE - 1), zz = q / NY, yy = q - NY * zz;
            dst[U0 + U1 + U2 + u] = T::CORE + T::YH + T::ZH + q * TA + xx;
            ld_row(U0 + U1 + U2 + u, j_lo + yy, k_lo + zz, false, true);
        }
        // x halo of the core rows: entries x = -3..-1 and x = TA..TA+2 (6 per row)
        constexpr int NXH = 6 * T::NCORE, NH = (NXH + 63) / 64;
        double vh[NH];
        int dh[NH];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int idx = min(lane + 64 * u, NXH - 1), row = idx / 6, ee = idx - 6 * row;
            const int x = ee < 3 ? ee - 3 : TA + ee - 3;
            const int gih = min(max(i_lo + x, 0), nx);
            const int zz = row / NY, yy = row - NY * zz;
            const int gj = min(j_lo + yy, ny), gk = min(k_lo + zz, nz);
            const bool interior = gih >= 1 && gih <= nx - 1 && gj <= ny - 1 && gk <= nz - 1;
            const bool up = (ee < 3) ? (si > 0) : (si < 0 && ni == TA);
            vh[u] = in[gih + sx * gj + sxy * gk + ((up && interior) ? dOI : 0)];
            dh[u] = row * T::RA + 3 + x;
        }
#pragma unroll
        for (int u = 0; u < NROW; ++u) lds[dst[u]] = v[u];
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (lane + 64 * u < NXH) lds[dh[u]] = vh[u];
    }
    __syncthreads();

    // ---- per-lane constants: lane -> (cell (b,c) of the NY x 4 cross-section, axis); see GsTile ------------
    int axis, b, c;
    if constexpr (NY == 4) {
        axis = lane & 3;
        b = (lane >> 2) & 3, c = lane >> 4;
    } else {
        const int t = lane & 15;
        b = t / 3, axis = t - 3 * b, c = lane >> 4; // t = 15: b = 5 >= nj, idle
    }
    const bool row_ok = b < nj && c < nk;
    const int y = sj > 0 ? b : nj - 1 - b, z = sk > 0 ? c : nk - 1 - c;
    const int yc = row_ok ? y : 0, zc = row_ok ? z : 0;
    const int gj = j_lo + yc, gk = k_lo + zc;
    const bool yz_weno = gj > 3 && gj < ny - 4 && gk > 3 && gk < nz - 4;
    const bool yquirk = axis == 1;
    int off[7];
    const int row_core = (zc * NY + yc) * T::RA + 3;
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const int d = m - 3;
        const int yy = yc + (axis == 1 ? d : 0), zz = zc + (axis == 2 ? d : 0), dxm = (axis == 1 || axis == 2) ? 0 : d;
        const bool in_y = yy >= 0 && yy < nj, in_z = zz >= 0 && zz < nk;
        const int o_core = (zz * NY + yy) * T::RA + 3 + dxm;
        const int o_yh = T::CORE + (zc * 6 + (yy < 0 ? yy + 3 : yy - nj + 3)) * TA;
        const int o_zh = T::CORE + T::YH + ((zz < 0 ? zz + 3 : zz - nk + 3) * NY + yc) * TA;
        off[m] = !in_y ? o_yh : (!in_z ? o_zh : o_core);
    }
    const int ps_row = T::CORE + T::YH + T::ZH + (zc * NY + yc) * TA;
    double acc = 0.0;
    const int nsteps = ni + nj + nk - 2;

    // ---- march: step st updates the cells with a + b + c = st of the tile's frame -------------------------------
    for (int st = 0; st < nsteps; ++st) {
        const int aa = st - b - c;
        const bool active = row_ok && aa >= 0 && aa < ni;
        const int ac = active ? aa : 0;
        const int x = si > 0 ? ac : ni - 1 - ac;
        double q[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) q[m] = lds[off[m] + x];
        const double pS = lds[ps_row + x];
        const int gi = i_lo + x;
        const bool weno_ok = yz_weno && gi > 3 && gi < nx - 4;
        double dm, dp;
        axis_pair<STRICT>(q, weno_ok, yquirk, dx, floor2, dm, dp);
        const double gg = axis_godunov<STRICT>(q[3], dm, dp);
        double gX, gY, gZ; // valid on the axis-0 lane of every cell (on all lanes for NY = 4)
        if constexpr (NY == 4) {
            gX = dpp_mov<0x00>(gg), gY = dpp_mov<0x55>(gg), gZ = dpp_mov<0xAA>(gg); // quad_perm broadcasts
        } else {
            gX = gg, gY = dpp_mov<0x101>(gg), gZ = dpp_mov<0x102>(gg);               // row_shl:1, row_shl:2
        }
        const double newv = finish_update<STRICT>(q[3], gX, gY, gZ, pS, dx, inv_dx, h);
        if (active && axis == 0) {
            lds[row_core + x] = newv;
            const double dlt = newv - q[3];
            acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
        }
        __syncthreads();
    }

    // ---- write back ----------------------------------------------------------------------------------------
    {
        constexpr int RPI = 64 / TA;
        const int xx = lane & (TA - 1), rsub = lane / TA;
#pragma unroll
        for (int u = 0; u < (T::NCORE + RPI - 1) / RPI; ++u) {
            const int r = u * RPI + rsub, zz = r / NY, yy = r - NY * zz;
            if (r < T::NCORE && xx < ni && yy < nj && zz < nk)
                out[(long)(i_lo + xx) + sx * (j_lo + yy) + sxy * (k_lo + zz)] = core[r * T::RA + 3 + xx];
        }
    }
    // ---- fused extrapolation BC for the wall points this tile owns (closed form, subs.f90:859-897) ----------
    const bool touches_wall = i_lo == 1 || i_lo + ni == nx || j_lo == 1 || j_lo + nj == ny || k_lo == 1 || k_lo + nk == nz;
    if (touches_wall) {
        const int e0 = ni + 2, e1 = nj + 2, e2 = nk + 2;
        for (int idx = lane; idx < e0 * e1 * e2; idx += 64) {
            const int ex = idx % e0 - 1, ey = (idx / e0) % e1 - 1, ez = idx / (e0 * e1) - 1;
            const int gi = i_lo + ex, gj2 = j_lo + ey, gk2 = k_lo + ez;
            const bool wi = gi == 0 || gi == nx, wj = gj2 == 0 || gj2 == ny, wk = gk2 == 0 || gk2 == nz;
            const int nb = (int)wi + (int)wj + (int)wk;
            if (nb == 0) continue;
            if ((!wi && (ex < 0 || ex >= ni)) || (!wj && (ey < 0 || ey >= nj)) || (!wk && (ez < 0 || ez >= nk))) continue;
            const int nh = (int)(gi == nx) + (int)(gj2 == ny) + (int)(gk2 == nz);
            const int m = min(nb, 1 + nh);
            const int cx = min(max(gi, 1), nx - 1) - i_lo, cy = min(max(gj2, 1), ny - 1) - j_lo,
                      cz = min(max(gk2, 1), nz - 1) - k_lo;
            double val = core[(cz * NY + cy) * T::RA + 3 + cx];
            {
#pragma clang fp contract(off)
                for (int t = 0; t < m; ++t) val = val + dx;
            }
            const long p = gi + sx * gj2 + sxy * gk2;
            const double dlt = val - in[p];
            out[p] = val;
            acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
        }
    }
    acc = wave_sum(acc);
    // ---- RMS column accumulation: the column's previous tile ran in an earlier launch ----------------------------
    if (lane == 0) {
        double* slot = a.colsum + (long)gb * ncol + (tj + (long)a.nTj * tk);
        *slot = ((fA == 0) ? 0.0 : *slot) + acc;
    }
    __syncthreads();
    // the far-corner tile is alone on the last hyperplane of its sweep: every other tile of the sweep is done
    if (fA == a.nTi - 1 && fB == a.nTj - 1 && fC == a.nTk - 1) {
        // ---- sweep epilogue: RMS, trace, stop / NaN test (subs.f90:902-926) --------------------------------
        const double* cs = a.colsum + (long)gb * ncol;
        double t = 0.0;
        for (int p = lane; p < ncol; p += 64) t += cs[p];
        t = wave_sum(t);
        if (lane == 0) {
            const double rms = __builtin_sqrt(t / a.den);
            if (g < a.trace_cap) a.trace[g] = rms;
            st_flag(a.ctl + 1, g + 1);
            if (rms < a.tol) st_flag(a.ctl + 0, 1);
            else if (rms != rms) { st_flag(a.ctl + 2, 1); st_flag(a.ctl + 0, 1); }
        }
    }
}

} // namespace lsf
