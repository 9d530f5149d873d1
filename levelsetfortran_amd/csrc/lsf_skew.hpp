// lsf_skew.hpp -- exact Gauss-Seidel reinit on SKEWED tiles (slot-synchronous schedule).
//
// The box tiles of lsf_boxtile.hpp (TA x NY x 4 cells, marched by in-tile hyperplanes a + b + c = step) keep only
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

#include <type_traits>

#include "lsf_boxtile.hpp"

// 16-byte loads and stores for tiles whose whole LDS image lies inside the grid (see skew_tile, "wide path"): 0 = never,
// 1 = one lane per cell only (default), 2 = both lane maps.  Measured at 512^3 (profiles/r03_wide_ab.txt, one box): one lane
// per cell 2.89 against 2.97 ms per sweep, HBM traffic unchanged (2.76 x algorithmic); three lanes per cell 2.68 against
// 2.65 ms and MORE traffic (3.43 x against 3.10 x: unaligned 16-byte write-through stores cost 35 % more write traffic, the
// loads 7 % more) -- the over-fetch of this kernel is halo rows and 128-byte lines around 144..176-byte row segments that
// start anywhere, not the width of its requests.
#ifndef LSF_SKEW_WIDE
#define LSF_SKEW_WIDE 1
#endif
#ifndef LSF_CELL_UNROLL
#define LSF_CELL_UNROLL 16 // one lane per cell: marching steps per iteration of the march loop.  16 = the whole march, phiS of
                           // all its steps in registers, no refill (8 measured within 1 % on one box: 1024^3 19.04 against 18.84
                           // ms, STRICT 512^3 equal -- once its refill was taken out from behind a run-time branch, see the march)
#endif

#ifndef LSF_STRICT22_WAVES
#define LSF_STRICT22_WAVES 4
#endif
// work-term probes of a smaller LDS image of the one-lane-per-cell tile (EXPERIMENT BUILDS ONLY; the field comes out wrong): row
// pitches of the bundle / halo rows and the wavefronts per SIMD the kernels are compiled for (profiles/r05_ring_probe.txt)
#if !defined(LSF_EXPERIMENTS)
#undef LSF_PROBE_RA
#undef LSF_PROBE_RH
#endif
#ifndef LSF_PROBE_RA
#define LSF_PROBE_RA 0
#endif
#ifndef LSF_PROBE_RH
#define LSF_PROBE_RH 0
#endif
#ifndef LSF_WAVES16
#define LSF_WAVES16 2
#endif
#ifndef LSF_POLL_SLEEP
#define LSF_POLL_SLEEP 16 // 64-cycle units between two looks of a waiting tile at its flags
#endif

namespace lsf {

// A tile is WY x WZ adjacent 5 x 4 bundles (one wavefront each), marched in lock step: NYT = 5 WY by NZT = 4 WZ
// rows share one LDS image, the halo between the bundles of a tile is internal, and the sweep needs fewer, fatter time
// slots (a flip of the raster direction along y costs n/16 + (number of tiles in y) slots, the cycle flips y six
// times out of eight and z twice).
//
// LDS rows, grouped so that every 4-row load instruction is uniformly "old" or "new" (1 x 1: 76 rows):
//   [0, NCORE)    bundle rows, r = c * NYT + b                                   entries 0..21 (RA = TA + 6)
//   [YU0, ZU0)    upstream y halo,   r = YU0 + c * 3 + (b' + 3),    b' = -3..-1   entries 0..17 (the only ones a
//   [ZU0, YD0)    upstream z halo,   r = ZU0 + (c' + 3) * NYT + b,  c' = -3..-1   stencil d = 1..3 rows away reaches)
//   [YD0, ZD0)    downstream y halo, r = YD0 + c * 3 + (b' - nj),   b' = nj..nj+2 entries 4..21
//   [ZD0, NR)     downstream z halo, r = ZD0 + (c' - nk) * NYT + b, c' = nk..nk+2 entries 4..21
// (z groups padded to whole load instructions).  phiS of the bundle cells is not staged: the lanes of a cell read
// their 16 values straight into registers.
//
// Two lane maps (BY = rows of one wavefront in y; a wavefront always holds 4 rows in z):
//   BY = 5   three lanes per cell (x, y, z derivative), 5 cells + 1 idle lane per 16-lane row: 20 cells per wavefront and step
//   BY = 16  one lane per cell (all three derivatives): 64 cells per wavefront and step, a third of the LDS reads and none
//            of the lane-to-lane traffic per cell, but 64 rows + halo per wavefront in LDS (two 16 x 16 tiles per CU)
template <int TA, int WY, int WZ, int BY = 5>
struct SkTile {
    static constexpr int W = WY * WZ;         // wavefronts per tile
    static_assert(W >= 1 && W <= 16, "1 to 16 wavefronts per tile");
    static_assert(BY == 5 || BY == 16, "three lanes per cell (5 rows) or one lane per cell (16 rows)");
    static constexpr int BYW = BY;
    static constexpr int NYT = BY * WY, NZT = 4 * WZ; // rows of a tile in y and z
    static constexpr int RA = (BY == 16 && LSF_PROBE_RA) ? LSF_PROBE_RA : TA + 6;
    // one lane per cell: the 32 lanes of an LDS access group are 16 b x 2 c; b steps by RA = 22 doubles (the 16 b cover the
    // even bank pairs), so an odd plane pitch puts the second c on the odd ones: conflict-free ds_read_b64 / ds_write_b64
    static constexpr int PAD = BY == 16 ? 1 : 0;
    static constexpr int HB = WY * BY * NZT * RA + NZT * PAD; // first halo entry
    static constexpr int RH = (BY == 16 && LSF_PROBE_RH) ? LSF_PROBE_RH : TA + 2; // entries kept of a halo row
    static constexpr int NCORE = NZT * NYT;
    static constexpr int YH = 3 * NZT;
    static constexpr int ZP = (3 * NYT + 3) / 4 * 4;
    static constexpr int YU0 = NCORE, ZU0 = YU0 + YH, YD0 = ZU0 + ZP, ZD0 = YD0 + YH;
    static constexpr int NR = (ZD0 + ZP + 4 * W - 1) / (4 * W) * (4 * W);
    static_assert(NCORE % (4 * W) == 0, "whole store instructions");
    static constexpr int TOTAL = HB + (NR - NCORE) * RH + (LSF_PROBE_RA || LSF_PROBE_RH ? 32 : 0);
    // LDS index of entry 0 of bundle row r
    __host__ __device__ static constexpr int core_at(int r) { return r * RA + (PAD ? (r / NYT) * PAD : 0); }
    // LDS index of entry k of row r (halo rows store entry 0 resp. 4 first)
    __host__ __device__ static constexpr int at(int r, int k)
    {
        return r < NCORE ? core_at(r) + k : HB + (r - NCORE) * RH + (r < YD0 ? k : k - 4);
    }
    __host__ __device__ static constexpr int row_at(int r)
    {
        return r < NCORE ? core_at(r) : HB + (r - NCORE) * RH - (r < YD0 ? 0 : 4);
    }
    // frame coordinates (bq, cq) of LDS row r of a tile with nj x nk rows (rows beyond a partial tile alias a valid row),
    // and its class: up = upstream halo, core = bundle row
    __host__ __device__ static void row_coords(int r, int nj, int nk, int* bq_, int* cq_, int* up_, int* core_)
    {
        int bq, cq, up = 0, core = 0;
        if (r < YU0) {
            cq = r / NYT, bq = r - NYT * cq, core = 1;
            bq = bq < nj - 1 ? bq : nj - 1, cq = cq < nk - 1 ? cq : nk - 1;
        } else if (r < ZU0) {
            const int q = r - YU0;
            cq = q / 3, bq = q - 3 * cq - 3, up = 1;
            cq = cq < nk - 1 ? cq : nk - 1;
        } else if (r < YD0) {
            const int q0 = r - ZU0, q = q0 < 3 * NYT - 1 ? q0 : 3 * NYT - 1, hz = q / NYT;
            bq = q - NYT * hz, bq = bq < nj - 1 ? bq : nj - 1, cq = hz - 3, up = 1;
        } else if (r < ZD0) {
            const int q = r - YD0;
            cq = q / 3, bq = nj + q - 3 * cq;
            cq = cq < nk - 1 ? cq : nk - 1;
        } else {
            const int q0 = r - ZD0, q = q0 < 3 * NYT - 1 ? q0 : 3 * NYT - 1, hz = q / NYT;
            bq = q - NYT * hz, bq = bq < nj - 1 ? bq : nj - 1, cq = nk + hz;
        }
        *bq_ = bq, *cq_ = cq, *up_ = up, *core_ = core;
    }
    // LDS index of the stencil value mm - 3 (absolute offset along the lane's axis, mm = 0..6) of the cell a lane works
    // on at step 0; the value of step t sits t entries further.  pos = the sweep runs in the positive direction of that axis
    __host__ __device__ static int lane_off(int bc, int cc, int axis, bool pos, int nj, int nk, int mm)
    {
        const int dA = mm - 3, dF = pos ? dA : -dA;
        int base = core_at(cc * NYT + bc);
        if (axis == 1) {
            const int bq = bc + dF;
            base = row_at((bq >= 0 && bq < nj) ? cc * NYT + bq : (bq < 0 ? YU0 + cc * 3 + bq + 3 : YD0 + cc * 3 + bq - nj));
        } else if (axis == 2) {
            const int cq = cc + dF;
            base = row_at((cq >= 0 && cq < nk) ? cq * NYT + bc : (cq < 0 ? ZU0 + (cq + 3) * NYT + bc : ZD0 + (cq - nk) * NYT + bc));
        }
        return base + 3 + dF;
    }
    // Tables for FULL tiles deep inside the grid (every row of the LDS image is an interior row: no clamping, no wall
    // flags), which are > 90 % of the tiles of a BASELINE-size grid.  They replace ~250 of the ~3000 vector instructions
    // a wavefront spends on a tile -- the kernel is bound by vector issue -- by two loads:
    //   rel_tab[r][(sj > 0) * 2 + (sk > 0)] = rel_j | rel_k << 6 | (bq + cq + 3) << 12 | flags << 20   (rowtab entry of row r:
    //       offset = rel_j * sx + rel_k * sxy from the tile origin, flags as in rowtab)
    //   off_tab[tid][pos][8] (16 bit each) = lane_off(.., mm) of thread tid for a sweep running in the positive (pos = 1) or
    //       negative direction of the lane's axis
    // A thread's entries (all four / both directions: 48 bytes) do not depend on the tile, so they are requested when the
    // block starts -- before it knows its tile -- and have arrived long before they are used (SkPre).
    __host__ __device__ static uint32_t rel_entry(int r, bool sjpos, bool skpos)
    {
        int bq, cq, up, core;
        row_coords(r, NYT, NZT, &bq, &cq, &up, &core);
        const int rel_j = 3 + (sjpos ? bq : NYT - 1 - bq), rel_k = 3 + (skpos ? cq : NZT - 1 - cq);
        static_assert(NYT + 5 < 64 && NZT + 5 < 64 && NYT + NZT + 5 < 256, "fields of a rel_tab entry");
        return (uint32_t)rel_j | ((uint32_t)rel_k << 6) | ((uint32_t)(bq + cq + 3) << 12) | ((uint32_t)(up + 2 * (up | core)) << 20);
    }
    // 32-bit words of the two tables (one lane per cell: the offsets are computed, 19 per lane and tile)
    static constexpr int REL_WORDS = 4 * NR, OFF_WORDS = BY == 5 ? 2 * 64 * W * 4 : 0;
};

// host side: fills the two tables of a tile shape (see SkTile); layout [rel_tab | off_tab]
template <int TA, int WY, int WZ, int BY>
inline void sk_fill_tables(uint32_t* out)
{
    using T = SkTile<TA, WY, WZ, BY>;
    for (int r = 0; r < T::NR; ++r)
        for (int d = 0; d < 4; ++d) out[4 * r + d] = T::rel_entry(r, (d >> 1) != 0, (d & 1) != 0);
    if (BY != 5) return;
    uint16_t* off = (uint16_t*)(out + T::REL_WORDS);
    const int NT = 64 * T::W;
    for (int pos = 0; pos < 2; ++pos)
        for (int tid = 0; tid < NT; ++tid) {
            const int lane = tid & 63, wave = tid >> 6, t16 = lane & 15, bl = t16 / 3, axis = t16 - 3 * bl;
            const int b = 5 * (wave % WY) + bl, c = 4 * (wave / WY) + (lane >> 4);
            const bool row_ok = bl < 5;
            const int bc = row_ok ? b : 0, cc = row_ok ? c : 0;
            for (int mm = 0; mm < 8; ++mm)
                off[(size_t)(2 * tid + pos) * 8 + mm] = mm < 7 ? (uint16_t)T::lane_off(bc, cc, axis, pos != 0, T::NYT, T::NZT, mm) : 0;
        }
}

// a thread's table entries (see SkTile), requested at block start
struct SkPre {
    uint4 rel, off0, off1;
};
template <int TA, int WY, int WZ, int BY>
__device__ __forceinline__ SkPre sk_prefetch(const GsArgs& a, int tid = threadIdx.x)
{
    using T = SkTile<TA, WY, WZ, BY>;
    SkPre p;
    p.rel = p.off0 = p.off1 = make_uint4(0u, 0u, 0u, 0u);
    if (a.tables) {
        p.rel = ((const uint4*)a.tables)[tid < T::NR ? tid : 0];
        if constexpr (BY == 5) {
            const uint4* o = (const uint4*)(a.tables + T::REL_WORDS) + 2 * tid;
            p.off0 = o[0], p.off1 = o[1];
        }
    }
    return p;
}

// Wide path of skew_tile: the tile's image -- NZT + 6 planes and a few rows of the field, from the tile origin -- is addressed
// through ONE raw buffer descriptor of 0x7fffffff bytes with 32-bit byte offsets; out-of-range loads would return 0 and stores
// would be dropped without a trace.  (Planes of 3 500^2 ... 4 650^2 points passed the old test, 4.0e9 bytes, and lay outside.)
__host__ __device__ inline bool sk_wide_image_fits(long sxy, int nzt) { return (double)(nzt + 7) * (double)sxy * 8.0 <= (double)(0x7fffffff - 16); }

// LDS of one tile: declared by the kernel (a kernel that runs several tiles one after the other, or two tile functions, has one)
template <class T>
struct SkShared {
    double lds[T::TOTAL];
    int2 rowtab[T::NR];
    double wsum[T::W];
};

// The LDS offsets of a lane's stencil at step 0 (skew_tile).  Functions, not lambdas of skew_tile: captured by reference inside the
// loaders' lambdas the arrays went to scratch memory.
// Three lanes per cell: index of the stencil value mm - 3 along the lane's axis.
template <class T>
__device__ __forceinline__ void sk_lane_offsets3(int axis, int si, int sj, int sk, bool deep, const SkPre& pre, int bc, int cc, int nj, int nk,
                                                 int (&off)[7])
{
    const bool pos = (axis == 0 ? si : (axis == 1 ? sj : sk)) > 0;
    if (deep) { // one 16-byte load instead of ~25 vector instructions per offset
        const uint4 pk = pos ? pre.off1 : pre.off0;
        off[0] = (int)(pk.x & 0xffffu), off[1] = (int)(pk.x >> 16), off[2] = (int)(pk.y & 0xffffu), off[3] = (int)(pk.y >> 16);
        off[4] = (int)(pk.z & 0xffffu), off[5] = (int)(pk.z >> 16), off[6] = (int)(pk.w & 0xffffu);
    } else {
#pragma unroll
        for (int mm = 0; mm < 7; ++mm) off[mm] = T::lane_off(bc, cc, axis, pos, nj, nk, mm);
    }
#pragma unroll
    for (int mm = 0; mm < 7; ++mm) asm volatile("" : "+v"(off[mm])); // computed where the caller stands, not at the first use
}
// One lane per cell: index of the stencil value mm - 3 along x / y / z (ox[3] = the cell itself).
template <class T>
__device__ __forceinline__ void sk_lane_offsets1(int si, int sj, int sk, int bc, int cc, int nj, int nk, int (&ox)[7], int (&oy)[7], int (&oz)[7])
{
    constexpr int NYT = T::NYT, NZT = T::NZT;
    if (nj == NYT && nk == NZT) {
        // full tile: the 19 offsets in ~90 instructions instead of 19 calls of lane_off (~25 each): the row d places up or down
        // the bundle is d rows (planes) further in the same row group, or in the halo group of that side
        constexpr int RA_ = T::RA, RH_ = T::RH, PS_ = NYT * T::RA + T::PAD;
        const int base = T::core_at(cc * NYT + bc) + 3; // the cell itself at step 0
        const int yu = T::HB + (T::YU0 - T::NCORE + cc * 3 + 3 + bc) * RH_ + 3, yd = T::HB + (T::YD0 - T::NCORE + cc * 3 + bc - NYT) * RH_ - 1;
        const int zu = T::HB + (T::ZU0 - T::NCORE + (cc + 3) * NYT + bc) * RH_ + 3, zd = T::HB + (T::ZD0 - T::NCORE + (cc - NZT) * NYT + bc) * RH_ - 1;
        int fy[7], fz[7]; // by frame offset d + 3
#pragma unroll
        for (int d = -3; d <= 3; ++d) {
            const int bq = bc + d, cq = cc + d;
            fy[d + 3] = bq < 0 ? yu + d * (RH_ + 1) : (bq >= NYT ? yd + d * (RH_ + 1) : base + d * (RA_ + 1));
            fz[d + 3] = cq < 0 ? zu + d * (NYT * RH_ + 1) : (cq >= NZT ? zd + d * (NYT * RH_ + 1) : base + d * (PS_ + 1));
        }
#pragma unroll
        for (int mm = 0; mm < 7; ++mm) {
            ox[mm] = base + (si > 0 ? mm - 3 : 3 - mm);
            oy[mm] = sj > 0 ? fy[mm] : fy[6 - mm];
            oz[mm] = sk > 0 ? fz[mm] : fz[6 - mm];
        }
    } else {
#pragma unroll
        for (int mm = 0; mm < 7; ++mm) {
            ox[mm] = T::lane_off(bc, cc, 0, si > 0, nj, nk, mm);
            oy[mm] = T::lane_off(bc, cc, 1, sj > 0, nj, nk, mm);
            oz[mm] = T::lane_off(bc, cc, 2, sk > 0, nj, nk, mm);
        }
    }
#pragma unroll
    for (int mm = 0; mm < 7; ++mm) asm volatile("" : "+v"(ox[mm]), "+v"(oy[mm]), "+v"(oz[mm]));
}

// One tile.  SC1 = false: every value this tile reads was written by an earlier launch (slot schedule).
// SC1 = true: producers may have run in this launch on another XCD (persistent schedule): results are stored
// write-through and drained before the caller publishes the tile, phi is loaded past the non-coherent caches
// (cdna_hip_programming.md G16).  Measured on the slot schedule: no cost (4.75 vs 4.82 ms per 512^3 sweep), whereas
// an agent-scope acquire per tile (L2 invalidate) with plain loads made every tile 35 % slower.
//
// The load has two stages.  Stage 1 takes what the PREVIOUS sweep left (bundle rows, downstream halo, phiS): valid as soon
// as that sweep has passed this tile's neighbourhood.  Stage 2 takes what THIS sweep's upstream tiles wrote (upstream halo,
// entries 0..2 of the bundle rows).  wait_upstream() is called between the two with the stage-1 loads in flight: the
// dataflow launch waits there for the upstream tiles (k_reinit_gs_persist), so that most of a tile's bytes travel while the
// block would otherwise be idle; it returns false when the tile must be abandoned (stop flag, time-out).
// Returns true when the tile has been computed and stored.
//
// PUSH (slabs of the exact ordering, k_reinit_gs_slab): the tile belongs to one of several launches that share the tile graph,
// each with field buffers of its own.  Results within three planes of a cut are stored into the neighbour's buffer as well
// (same address map), the column's running RMS sum into the colsum of the slab that runs this sweep's epilogue, and the
// epilogue's verdict into every slab's control words; all of that and every load at system scope.
#ifdef LSF_EXPERIMENTS
// experiment builds: mid(t) is called in front of marching step t (the early-flag probe of profiles/micro hooks in here)
struct SkNoHook {
    __device__ __forceinline__ void operator()(int) const {}
};
#define LSF_SK_MID_TPARAM , class MidHook = SkNoHook
#define LSF_SK_MID_PARAM , MidHook mid = MidHook{}
#define LSF_SK_MID(t_) mid(t_)
#else
#define LSF_SK_MID_TPARAM
#define LSF_SK_MID_PARAM
#define LSF_SK_MID(t_)
#endif
template <int TA, int WY, int WZ, int BY, bool STRICT, bool SC1, bool PUSH = false, class WaitUp LSF_SK_MID_TPARAM>
__device__ __forceinline__ bool skew_tile(SkShared<SkTile<TA, WY, WZ, BY>>& sm, const GsArgs& a, uint32_t packed, int g, int si, int sj,
                                          int sk, const SkPre& pre, WaitUp&& wait_upstream LSF_SK_MID_PARAM)
{
    static_assert(TA == 16 || (TA == 32 && BY == 5 && !PUSH), "tiles of 32 marching steps: three lanes per cell, single launch only");
    constexpr int CPN = TA / 16; // the loaders and the write back handle a row 16 entries at a time
    using T = SkTile<TA, WY, WZ, BY>;
    constexpr int NYT = T::NYT, NZT = T::NZT, W = T::W, NT = 64 * W;
    double* const lds = sm.lds;
    int2* const rowtab = sm.rowtab;
    double* const wsum = sm.wsum;
    int tid_ = threadIdx.x;
    // called from a loop over tiles (k_reinit_gs_slab): what depends on the thread index only would be computed in front of the
    // loop and kept in vector registers across it (measured: 60 of them spilled to scratch) -- make it a value of this call
    if constexpr (PUSH) asm volatile("" : "+v"(tid_)); // (in the single launch it costs 0.4-1 %: measured)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sx = nx + 1, sxy = (long)(nx + 1) * (ny + 1);
    const double dx = a.dx, h = a.h;
    const double inv_dx = 1.0 / dx, floor2 = 1.E-99 * dx * dx / 13.0;
    const int ncol = a.nTj * a.nTk;

    constexpr bool SYS = PUSH; // slabs: the other side of an access may be another device -- system scope
    auto ldp = [](const double* p_) { return SYS ? ld_sys(p_) : (SC1 ? ld_sc1(p_) : *p_); };
    auto stp = [](double* p_, double v_) {
        if (SYS) st_sys(p_, v_);
        else if (SC1) st_sc1(p_, v_);
        else *p_ = v_;
    };

#ifdef LSF_EXPERIMENTS // phase times of a tile (thread 0; 100 MHz ticks) into dbg[3..]: row table, load, march, write back
    unsigned long long ph_[6];
#define LSF_PHASE(i_) ph_[i_] = __builtin_amdgcn_s_memrealtime()
#else
#define LSF_PHASE(i_)
#endif
    LSF_PHASE(0);
    const int m = packed & 0x3ff, fB = (packed >> 10) & 0x3ff, fC = (packed >> 20) & 0x3ff;
    const int tj = sj > 0 ? fB : a.nTj - 1 - fB, tk = sk > 0 ? fC : a.nTk - 1 - fC;
    const int j_lo = 1 + tj * NYT, k_lo = 1 + tk * NZT;
    const int nj = min(NYT, ny - j_lo), nk = min(NZT, nz - k_lo);
    const int nxi = nx - 1;                       // interior cells along x
    const int X0 = TA * m - NYT * fB - NZT * fC;  // Fx of row (0,0) at step 0
    const int gb = a.nbuf == 4 ? (g & 3) : g % 3; // nbuf is 3 or 4: no division by a run-time value
    const double* in = a.buf[gb];
    double* out = a.buf[gb + 1 == a.nbuf ? 0 : gb + 1];
    // slabs: the planes [k_own_lo, k_own_lo + 3) also live in the lower neighbour's buffer, [k_own_hi - 3, k_own_hi) in the upper
    // one's (their addresses come from device memory: asked for here, used by the write back)
    [[maybe_unused]] const int k_push_lo = PUSH ? 1 + a.tk_lo * NZT + 3 : 0, k_push_hi = PUSH ? min(1 + a.tk_hi * NZT, nz) - 3 : 0;
    [[maybe_unused]] double *nb_lo = nullptr, *nb_hi = nullptr;
    if constexpr (PUSH) {
        const int ob = gb + 1 == a.nbuf ? 0 : gb + 1;
        nb_lo = a.peers->nb_buf[0][ob], nb_hi = a.peers->nb_buf[1][ob];
    }

    // ---- row table ------------------------------------------------------------------------------------------
    // rowtab[r].x = (element offset of the row's i = 0 point from the tile origin) * 4 + flags
    //                 flag 1: entries 3..18 hold values of this sweep (upstream halo row inside the interior)
    //                 flag 2: entries 0..2 do (bundle row, or upstream halo row, inside the interior)
    // rowtab[r].y = global i of the row's entry 3 (may lie outside [0, nx]: clamped on use)
    // Offsets are tile-relative so that 32 bits suffice whatever the field size.
    const int org_j = max(j_lo - 3, 0), org_k = max(k_lo - 3, 0);
    const long org = sx * org_j + sxy * org_k;
    const double* in_t = in + org;
    double* out_t = out + org;
    const double* ps_t = a.phiS + org;
    const int gi0 = si > 0 ? 1 + X0 : nx - 1 - X0; // global i of frame position Fx = X0
    // deep tile: full, and every row of its LDS image (3 halo rows on each side included) is an interior row
    const bool deep = a.tables && nj == NYT && nk == NZT && j_lo >= 4 && j_lo + NYT + 2 <= ny - 1 && k_lo >= 4 && k_lo + NZT + 2 <= nz - 1;
    const int dir = (sj > 0 ? 2 : 0) + (sk > 0 ? 1 : 0);
    for (int r = tid; r < T::NR; r += NT) {
        if (deep) {
            const uint32_t w = r == tid ? (dir == 0 ? pre.rel.x : (dir == 1 ? pre.rel.y : (dir == 2 ? pre.rel.z : pre.rel.w)))
                                        : a.tables[4 * r + dir];
            const int o = (int)(w & 63u) * (int)sx + (int)((w >> 6) & 63u) * (int)sxy, bc_ = (int)((w >> 12) & 255u) - 3;
            rowtab[r] = make_int2(o * 4 + (int)(w >> 20), si > 0 ? gi0 - bc_ : gi0 + bc_);
            continue;
        }
        int bq, cq, up, core;
        T::row_coords(r, nj, nk, &bq, &cq, &up, &core);
        const int gj_r = j_lo + (sj > 0 ? bq : nj - 1 - bq), gk_r = k_lo + (sk > 0 ? cq : nk - 1 - cq);
        const int rin = (int)((unsigned)(gj_r - 1) <= (unsigned)(ny - 2)) & (int)((unsigned)(gk_r - 1) <= (unsigned)(nz - 2));
        const int gj = min(max(gj_r, 0), ny), gk = min(max(gk_r, 0), nz);
        const int o = (gj - org_j) * (int)sx + (gk - org_k) * (int)sxy; // a slab of NZT + 6 planes of any admissible field fits 29 bits
        rowtab[r] = make_int2(o * 4 + (rin & up) + 2 * (rin & (up | core)), si > 0 ? gi0 - (bq + cq) : gi0 + (bq + cq));
    }
    __syncthreads();
    LSF_PHASE(1);

    // ---- load: all global loads in flight before the first LDS write ------------------------------------
    // lane map of a wavefront: 3 lanes per cell inside each 16-lane row (5 cells + 1 idle lane), 4 rows = 4 c;
    // wavefront (wy, wz) of the tile owns rows b = 5 wy .. 5 wy + 4, c = 4 wz .. 4 wz + 3
    // (one lane per cell, BY = 16: the 16 lanes of a row of lanes are the 16 b of the wavefront, axis is unused)
    const int t16 = lane & 15;
    const int bl = BY == 5 ? t16 / 3 : t16, axis = BY == 5 ? t16 - 3 * bl : 0; // BY = 5, t16 = 15: bl = 5, idle
    const int b = BY * (wave % WY) + bl, c = 4 * (wave / WY) + (lane >> 4);
    const bool row_ok = bl < BY && b < nj && c < nk;
    const int bc = row_ok ? b : 0, cc = row_ok ? c : 0;
    // phiS of the 16 cells this lane's cell row holds (the x lane of a cell uses them): 8 registers, refilled for
    // step t + 8 as soon as step t has used its value
    // (one lane per cell: CU values, refilled CU steps ahead, CU = steps per iteration of the march loop)
    constexpr int CU = LSF_CELL_UNROLL;
    static_assert(CU == 4 || CU == 8 || CU == 16, "steps per iteration of the one-lane-per-cell march");
    constexpr int NPS = BY == 5 ? TA / 2 : CU;
    double ps[NPS];
    double colsum_prev = 0.0;
    const int2 e_ps = rowtab[cc * NYT + bc];
    auto ps_load = [&](int t) {
        return ps_t[(unsigned)(e_ps.x >> 2) + (unsigned)min(max(e_ps.y + (si > 0 ? t : -t), 0), nx)];
    };
    // ---- per-lane constants ---------------------------------------------------------------------------------
    const int gj = j_lo + (sj > 0 ? bc : nj - 1 - bc), gk = k_lo + (sk > 0 ? cc : nk - 1 - cc);
    const bool yz_weno = gj > 3 && gj < ny - 4 && gk > 3 && gk < nz - 4;
    const int fx0 = X0 - bc - cc;
    double acc = 0.0;
    // per-lane bit t: the cell of step t exists (0 <= fx0 + t < nxi) / takes the WENO branch (3 < gi < nx - 4)
    auto step_mask = [](int lo_t, int hi_t) { // bits lo_t .. hi_t - 1 of 16
        lo_t = min(max(lo_t, 0), TA), hi_t = min(max(hi_t, lo_t), TA);
        return (unsigned)(((1ull << hi_t) - 1ull) & ~((1ull << lo_t) - 1ull));
    };
    const unsigned act_bits = row_ok ? step_mask(-fx0, nxi - fx0) : 0u;
    // 3 < gi < nx - 4 with gi = 1 + fx (si > 0: 2 < fx < nx - 5) or gi = nx - 1 - fx (si < 0: 3 < fx < nx - 4)
    const unsigned weno_bits = yz_weno ? step_mask((si > 0 ? 3 : 4) - fx0, (si > 0 ? nx - 5 : nx - 4) - fx0) : 0u;

    // LDS offsets of a lane's stencil.  They used to be computed behind the barrier that ends the load phase: ~300 (three lanes per
    // cell) / ~700 (one lane per cell) instructions between the arrival of the upstream tiles' values and the first marching step,
    // on the critical path of every hand-off.  Now in front of the wait for the upstream tiles and pinned there -- one lane per cell:
    // behind the first load stage, the loads in flight; three lanes per cell: in front of the loads (96 registers per lane leave no
    // room for them beside the loads in flight: 832 bytes of scratch when tried).  Round 4, A/B/A/B on one box, ms per sweep:
    // 128^3 FAST 0.308 -> 0.279, STRICT 0.498 -> 0.474; 256^3 0.637 -> 0.607, 1.036 -> 1.008; 512^3 unchanged (not bound by a chain).
    [[maybe_unused]] bool yquirk = false;
    [[maybe_unused]] int row_core = 0;
    [[maybe_unused]] int off[7] = {0, 0, 0, 0, 0, 0, 0}; // three lanes per cell (sk_lane_offsets3)
    [[maybe_unused]] int ox[7] = {0, 0, 0, 0, 0, 0, 0}, oy[7] = {0, 0, 0, 0, 0, 0, 0}, oz[7] = {0, 0, 0, 0, 0, 0, 0}; // one lane per cell (sk_lane_offsets1)
    if constexpr (BY == 5) {
        yquirk = axis == a.quirk_axis;
        row_core = T::core_at(cc * NYT + bc) + 3;
        sk_lane_offsets3<T>(axis, si, sj, sk, deep, pre, bc, cc, nj, nk, off);
    }
    // Wide path: a deep tile (full, every row of its image an interior row) whose image also stays inside the grid along the
    // march axis -- entries 0 .. 22 of every row are interior cells -- needs no clamp, no interior test and no choice between
    // `in` and `out` per entry: its rows travel as 16-byte buffer loads (two entries per lane, 10 / 9 lanes per row; the pair is
    // swapped on its way into LDS when the sweep runs against the memory order) and its results leave as 16-byte stores.  One
    // request per 16 bytes instead of one per 8 (an 8-byte write-through store is a fabric write of its own), and the six
    // odd entries per bundle row no longer cost separate scattered requests.
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    constexpr int AUX_SC1 = SYS ? 17 : (SC1 ? 16 : 0); // cache policy of the buffer instructions: sc1 = bit 4, sc0 = bit 0 (system scope: both)
    const bool widex = TA == 16 && (LSF_SKEW_WIDE == 2 || (LSF_SKEW_WIDE == 1 && BY == 16)) && deep && X0 - (NYT + NZT + 4) >= 0 && X0 + 22 <= nxi - 1 &&
                       sk_wide_image_fits(sxy, NZT);
    bool loaded = true;
    auto load_wide = [&]() {
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(in_t), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out_t, 0, 0x7fffffff, 0x00020000);
        constexpr int RPI = 4 * W;
        constexpr int NB = T::NCORE / RPI, NU = (T::YD0 - T::YU0 + RPI - 1) / RPI, ND = (T::NR - T::YD0 + RPI - 1) / RPI;
        constexpr int XC = 3 * T::NCORE, NXC = (XC + NT - 1) / NT;
        u4_t w[NB + ND + NU];
        int dw[NB + ND + NU];
        double v3[NXC + 1];
        int d3[NXC + 1];
        const int xx = tid & 15, rsub = tid >> 4;
        int n_ = 0;
        // entries k, k + 1 of a row: one 16-byte load at the lower of their two addresses
        auto pair_at = [&](const __amdgpu_buffer_rsrc_t& rs, int2 e, int k) {
            const unsigned el = (unsigned)(e.x >> 2) + (unsigned)(e.y + (si > 0 ? k - 3 : 2 - k));
            return __builtin_amdgcn_raw_buffer_load_b128(rs, 8u * el, 0, AUX_SC1);
        };
        // ---- stage 1: what the previous sweep left (bundle rows entries 3 .. 21 (+ 22, dropped), downstream halo 4 .. 21)
#pragma unroll
        for (int u = 0; u < NB; ++u, ++n_) {
            const int r = RPI * u + rsub, k = 3 + 2 * min(xx, 9);
            dw[n_] = T::core_at(r) + k;
            w[n_] = pair_at(r_in, rowtab[r], k);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u, ++n_) {
            const int r = min(T::YD0 + RPI * u + rsub, T::NR - 1), k = 4 + 2 * min(xx, 8);
            dw[n_] = T::HB + (r - T::NCORE) * T::RH + k - 4;
            w[n_] = pair_at(r_in, rowtab[r], k);
        }
#pragma unroll
        for (int t = 0; t < NPS; ++t) ps[t] = ps_load(t);
        if constexpr (BY == 16) sk_lane_offsets1<T>(si, sj, sk, bc, cc, nj, nk, ox, oy, oz);
        if (!wait_upstream()) {
            loaded = false;
            return;
        }
        if (tid == 0 && m != (NYT * fB + NZT * fC) / TA) colsum_prev = ldp(a.colsum + (long)gb * ncol + (tj + (long)a.nTj * tk));
        // ---- stage 2: what the upstream tiles of this sweep wrote (upstream halo 0 .. 17, bundle rows entries 0 .. 2)
#pragma unroll
        for (int u = 0; u < NU; ++u, ++n_) {
            const int r = min(T::YU0 + RPI * u + rsub, T::YD0 - 1), k = 2 * min(xx, 8);
            dw[n_] = T::HB + (r - T::NCORE) * T::RH + k;
            w[n_] = pair_at(r_out, rowtab[r], k);
        }
#pragma unroll
        for (int u = 0; u < NXC; ++u) {
            const int idx = min(tid + NT * u, XC - 1);
            const int r = idx / 3, k = idx - 3 * r;
            const int2 e = rowtab[r];
            d3[u] = T::core_at(r) + k;
            v3[u] = ldp((const double*)out_t + ((unsigned)(e.x >> 2) + (unsigned)(e.y + (si > 0 ? k - 3 : 3 - k))));
        }
#pragma unroll
        for (int u = 0; u < NB + ND + NU; ++u) {
            const double lo = __hiloint2double((int)w[u].y, (int)w[u].x), hi = __hiloint2double((int)w[u].w, (int)w[u].z);
            lds[dw[u]] = si > 0 ? lo : hi;                                    // entry k
            if (u >= NB || xx < 9) lds[dw[u] + 1] = si > 0 ? hi : lo;         // entry k + 1 (a bundle row has no entry 22)
        }
#pragma unroll
        for (int u = 0; u < NXC; ++u) lds[d3[u]] = v3[u];
    };
    auto load_rows = [&]() {
        // 16 entries of one row per 16 lanes, 4 W rows per load instruction.  Rows are taken CLASS BY CLASS (bundle /
        // upstream halo / downstream halo) so that everything that depends on the class -- first entry, where the row
        // lives in the LDS image, whether a value of this sweep has to be fetched from `out` -- is a compile-time
        // property of the instruction (the class boundaries are not multiples of 4 W rows: the last instruction of a
        // class re-loads its last row in the lanes that would overshoot, which rewrites the same LDS value).
        constexpr int RPI = 4 * W;                                        // rows per load instruction
        // (TA = 32: every class twice, the second time 16 entries further -- u counts (row group, 16-entry piece) pairs)
        constexpr int NB = T::NCORE / RPI * CPN;                          // bundle rows: entries 3..18, old values
        constexpr int NU = (T::YD0 - T::YU0 + RPI - 1) / RPI * CPN;       // upstream halo: 2..17, this sweep's inside the interior
        constexpr int ND = (T::NR - T::YD0 + RPI - 1) / RPI * CPN;        // downstream halo: 4..19, old values
        // the remaining entries: 0..2 (this sweep's) and 19..21 (old) of the bundle rows, 0..1 of
        // the upstream and 20..21 of the downstream halo rows
        constexpr int XC = 3 * T::NCORE, XHU = 2 * (T::YD0 - T::NCORE), XHD = 2 * (T::NR - T::YD0);
        constexpr int NXC = (XC + NT - 1) / NT, NXHU = (XHU + NT - 1) / NT, NXHD = (XHD + NT - 1) / NT;
        constexpr int NV = NB + NU + ND + 2 * NXC + NXHU + NXHD;
        double v[NV];
        int dst[NV];
        const int xx = tid & 15, rsub = tid >> 4;
        int n_ = 0;
        auto gi_of = [&](int2 e, int k) { return e.y + (si > 0 ? k - 3 : 3 - k); };
        auto off_of = [&](int2 e, int gi_r) { return (unsigned)(e.x >> 2) + (unsigned)min(max(gi_r, 0), nx); };
        // ---- stage 1: what the previous sweep left
#pragma unroll
        for (int u = 0; u < NB; ++u, ++n_) {
            const int r = RPI * (u / CPN) + rsub, k = 3 + xx + 16 * (u % CPN);
            const int2 e = rowtab[r];
            dst[n_] = T::core_at(r) + k;
            v[n_] = ldp(in_t + off_of(e, gi_of(e, k)));
        }
#pragma unroll
        for (int u = 0; u < ND; ++u, ++n_) {
            const int r = min(T::YD0 + RPI * (u / CPN) + rsub, T::NR - 1), k = 4 + xx + 16 * (u % CPN);
            const int2 e = rowtab[r];
            dst[n_] = T::HB + (r - T::NCORE) * T::RH + k - 4;
            v[n_] = ldp(in_t + off_of(e, gi_of(e, k)));
        }
#pragma unroll
        for (int u = 0; u < NXC; ++u, ++n_) { // bundle rows: entries 19..21
            const int idx = min(tid + NT * u, XC - 1);
            const int r = idx / 3, k = TA + 3 + idx - 3 * r;
            const int2 e = rowtab[r];
            dst[n_] = T::core_at(r) + k;
            v[n_] = ldp(in_t + off_of(e, gi_of(e, k)));
        }
#pragma unroll
        for (int u = 0; u < NXHD; ++u, ++n_) { // downstream halo rows: entries 20, 21
            const int hh = min(tid + NT * u, XHD - 1);
            const int r = T::YD0 + (hh >> 1), k = TA + 4 + (hh & 1);
            const int2 e = rowtab[r];
            dst[n_] = T::at(r, k);
            v[n_] = ldp(in_t + off_of(e, gi_of(e, k)));
        }
#pragma unroll
        for (int t = 0; t < NPS; ++t) ps[t] = ps_load(t);
        if constexpr (BY == 16) sk_lane_offsets1<T>(si, sj, sk, bc, cc, nj, nk, ox, oy, oz);
        if (!wait_upstream()) {
            loaded = false;
            return;
        }
        // the running RMS sum of this tile column, left by the previous tile of the column (one of the upstream tiles):
        // requested now, used by thread 0 after the march (the dependent load used to sit between the tile's stores and its flag)
        if (tid == 0 && m != (NYT * fB + NZT * fC) / TA) colsum_prev = ldp(a.colsum + (long)gb * ncol + (tj + (long)a.nTj * tk));
        // ---- stage 2: what the upstream tiles of this sweep wrote (rows or entries outside the interior: the walls, from `in`)
#pragma unroll
        for (int u = 0; u < NU; ++u, ++n_) {
            const int r = min(T::YU0 + RPI * (u / CPN) + rsub, T::YD0 - 1), k = 2 + xx + 16 * (u % CPN);
            const int2 e = rowtab[r];
            const int gi_r = gi_of(e, k);
            const bool fresh = ((unsigned)(gi_r - 1) <= (unsigned)(nx - 2)) & (bool)(e.x & 1);
            dst[n_] = T::HB + (r - T::NCORE) * T::RH + k;
            v[n_] = ldp((fresh ? (const double*)out_t : in_t) + off_of(e, gi_r));
        }
#pragma unroll
        for (int u = 0; u < NXC; ++u, ++n_) { // bundle rows: entries 0..2 (the previous tile of the row)
            const int idx = min(tid + NT * u, XC - 1);
            const int r = idx / 3, k = idx - 3 * r;
            const int2 e = rowtab[r];
            const int gi_r = gi_of(e, k);
            const bool fresh = ((unsigned)(gi_r - 1) <= (unsigned)(nx - 2)) & (bool)(e.x & 2);
            dst[n_] = T::core_at(r) + k;
            v[n_] = ldp((fresh ? (const double*)out_t : in_t) + off_of(e, gi_r));
        }
#pragma unroll
        for (int u = 0; u < NXHU; ++u, ++n_) { // upstream halo rows: entries 0, 1
            const int hh = min(tid + NT * u, XHU - 1);
            const int r = T::NCORE + (hh >> 1), k = hh & 1;
            const int2 e = rowtab[r];
            const int gi_r = gi_of(e, k);
            const bool fresh = ((unsigned)(gi_r - 1) <= (unsigned)(nx - 2)) & (bool)(e.x & 2);
            dst[n_] = T::at(r, k);
            v[n_] = ldp((fresh ? (const double*)out_t : in_t) + off_of(e, gi_r));
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) lds[dst[u]] = v[u]; // duplicates (clamped indices) rewrite the same value
    };
    if (widex) load_wide();
    else load_rows();
    if (!loaded) return false;
    __syncthreads();
    LSF_PHASE(2);

    if constexpr (BY == 5) {
        // ---- march: TA steps, every lane busy; the wavefronts of a tile meet after every step -----------------
        // (one instance, fully unrolled: a second instance without step masks for tiles whose every cell takes the WENO branch,
        // -11 % vector instructions per step, was 5 % SLOWER -- the kernel grew from 44 to 52 KB --, the march as a loop of two
        // iterations of eight steps 20 % slower: profiles/r03_gs_march_ab.txt)
        {
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                LSF_SK_MID(t);
                const bool active = (bool)((act_bits >> t) & 1u) && axis == 0;
                double q[7];
#pragma unroll
                for (int mm = 0; mm < 7; ++mm) q[mm] = lds[off[mm] + t];
                const double pS = ps[t & (TA / 2 - 1)];
                if (t + TA / 2 < TA) ps[t & (TA / 2 - 1)] = ps_load(t + TA / 2);
                const bool weno_ok = (bool)((weno_bits >> t) & 1u);
                double dm, dp;
                axis_pair<STRICT>(q, weno_ok, yquirk, dx, floor2, dm, dp);
                const double gg = axis_godunov<STRICT>(q[3], dm, dp);
                const double gX = gg, gY = dpp_mov<0x101>(gg), gZ = dpp_mov<0x102>(gg); // row_shl:1, row_shl:2
                if (active) { // only the x lane of a cell needs the tail (|grad|, sign, Euler step)
                    const double newv = finish_update<STRICT>(q[3], gX, gY, gZ, pS, dx, inv_dx, h);
                    lds[row_core + t] = newv;
                    const double dlt = newv - q[3];
                    acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
                }
                __syncthreads();
            }
        }
    } else {
        // ---- one lane per cell: the lane evaluates the three axes of its cell; 19 LDS reads, one LDS write per cell ----
        const bool quirk_x = a.quirk_axis == 0, quirk_y = a.quirk_axis == 1;
#pragma unroll 1
        for (int t0 = 0; t0 < TA; t0 += CU) {
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int t = t0 + u;
                LSF_SK_MID(t);
                const bool active = (act_bits >> t) & 1u;
                const bool weno_ok = (weno_bits >> t) & 1u;
                double qx[7], qy[7], qz[7];
#pragma unroll
                for (int mm = 0; mm < 7; ++mm) qx[mm] = lds[ox[mm] + t];
#pragma unroll
                for (int mm = 0; mm < 7; ++mm) qy[mm] = mm == 3 ? qx[3] : lds[oy[mm] + t];
#pragma unroll
                for (int mm = 0; mm < 7; ++mm) qz[mm] = mm == 3 ? qx[3] : lds[oz[mm] + t];
                const double pS = ps[u];
                // refill for CU steps ahead.  STRICT: unconditionally (the index is clamped, the last iteration's loads are never
                // used): behind the run-time branch the compiler lost count of the outstanding loads there and waited for all of
                // them -- the one just issued included -- in the middle of every step (-3 % at 512^3).  FAST keeps the branch:
                // without it 1024^3 and 768^3 are 4-5 % slower (measured on one box; eight more loads per lane and tile).
                if (STRICT ? CU < TA : (CU < TA && t0 + CU < TA)) ps[u] = ps_load(min(t + CU, TA - 1));
                if (active) {
                    const double c0 = qx[3];
                    double xm, xp, ym, yp, zm, zp;
                    axis_pair<STRICT>(qx, weno_ok, quirk_x, dx, floor2, xm, xp);
                    axis_pair<STRICT>(qy, weno_ok, quirk_y, dx, floor2, ym, yp);
                    axis_pair<STRICT>(qz, weno_ok, false, dx, floor2, zm, zp);
                    const double newv = finish_update<STRICT>(c0, axis_godunov<STRICT>(c0, xm, xp), axis_godunov<STRICT>(c0, ym, yp),
                                                              axis_godunov<STRICT>(c0, zm, zp), pS, dx, inv_dx, h);
                    lds[ox[3] + t] = newv;
                    const double dlt = newv - c0;
                    acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
                }
                // LDS hand-off between the wavefronts of the tile: nothing global has to be visible, do not drain the phiS loads
                if (W > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }

    LSF_PHASE(3);
    // ---- write back, fused extrapolation BC (subs.f90:859-897, closed form) --------------------------------
    const int fx_min = X0 - (nj - 1) - (nk - 1), fx_max = X0 + TA - 1;
    const bool near_wall = j_lo == 1 || j_lo + nj == ny || k_lo == 1 || k_lo + nk == nz ||
                           (fx_min <= 0 && fx_max >= 0) || (fx_min <= nxi - 1 && fx_max >= nxi - 1);
    if (widex) { // every cell of the tile exists and none touches a wall: 16-byte stores, two results per lane
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out_t, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): no load is in flight (see the other branch)
#pragma unroll
        for (int u = 0; u < (T::NCORE + 8 * W - 1) / (8 * W); ++u) {
            const int r = 8 * W * u + (tid >> 3), t = 2 * (tid & 7);
            if (r < T::NCORE) {
                const int2 e = rowtab[r];
                const double v0 = lds[T::core_at(r) + 3 + t], v1 = lds[T::core_at(r) + 4 + t];
                const double lo = si > 0 ? v0 : v1, hi = si > 0 ? v1 : v0;
                u4_t pk;
                pk.x = (unsigned)__double2loint(lo), pk.y = (unsigned)__double2hiint(lo);
                pk.z = (unsigned)__double2loint(hi), pk.w = (unsigned)__double2hiint(hi);
                const unsigned el = (unsigned)(e.x >> 2) + (unsigned)(e.y + (si > 0 ? t : -t - 1));
                __builtin_amdgcn_raw_buffer_store_b128(pk, r_out, 8u * el, 0, AUX_SC1);
                if constexpr (PUSH) { // the copies for the neighbour slabs (the three planes next to a cut)
                    const int cq = r / NYT, gk2 = k_lo + (sk > 0 ? cq : nk - 1 - cq);
                    if (nb_lo && gk2 < k_push_lo)
                        __builtin_amdgcn_raw_buffer_store_b128(pk, __builtin_amdgcn_make_buffer_rsrc(nb_lo + org, 0, 0x7fffffff, 0x00020000), 8u * el, 0, AUX_SC1);
                    if (nb_hi && gk2 >= k_push_hi)
                        __builtin_amdgcn_raw_buffer_store_b128(pk, __builtin_amdgcn_make_buffer_rsrc(nb_hi + org, 0, 0x7fffffff, 0x00020000), 8u * el, 0, AUX_SC1);
                }
            }
        }
    } else {
    // Near the walls the old values of the wall points are needed (their change enters the RMS, subs.f90:902-914).  Taken one by
    // one inside the store loop, as this was written first, each cost a round trip past the caches, and -- the counter of
    // outstanding memory operations being one for loads and stores -- the wait for each also waited for the acknowledgement of
    // the write-through stores in front of it, at EVERY row of a tile near a wall whether the lane had wall points or not; the
    // tiles along the walls are the first of every hyperplane.  Now: wall points first (chunks of rows, all old values of a chunk
    // in flight together, wavefronts without wall points skip), then the cells.  At 256^3, where a sweep is bound by the latency
    // of its dependency chains: 0.725 -> 0.65 ms per sweep; the summation order of the RMS is unchanged.
    constexpr int NUW = T::NCORE / (4 * W) * CPN;  // (row group, 16-entry piece) pairs of a lane
#ifndef LSF_WB_CHUNK
#define LSF_WB_CHUNK 3
#endif
#ifndef LSF_WB_CHUNK16
#define LSF_WB_CHUNK16 3
#endif
    constexpr int GWC = BY == 16 ? LSF_WB_CHUNK16 : LSF_WB_CHUNK;
    constexpr int GW = NUW < GWC ? NUW : GWC; // ... per chunk (7 old values each: registers)
    struct WbRow {
        int cq, t, gi, gj2, gk2, ai, aj, ak;
        int2 e;
        bool mine;
    };
    auto wb_row_e = [&](int u, int2 e_) {
        WbRow w;
        const int r = 4 * W * (u / CPN) + (tid >> 4);
        w.cq = r / NYT;
        const int bq = r - NYT * w.cq;
        w.t = (tid & 15) + 16 * (u % CPN);
        w.e = e_;
        w.gi = w.e.y + (si > 0 ? w.t : -w.t);
        w.mine = bq < nj && w.cq < nk && (unsigned)(w.gi - 1) <= (unsigned)(nx - 2);
        w.gj2 = j_lo + (sj > 0 ? bq : nj - 1 - bq), w.gk2 = k_lo + (sk > 0 ? w.cq : nk - 1 - w.cq);
        // wall points that clamp to this cell: move outward along any non-empty subset of its wall-adjacent axes
        w.ai = w.gi == 1 ? -1 : (w.gi == nx - 1 ? 1 : 0), w.aj = w.gj2 == 1 ? -1 : (w.gj2 == ny - 1 ? 1 : 0);
        w.ak = w.gk2 == 1 ? -1 : (w.gk2 == nz - 1 ? 1 : 0);
        return w;
    };
    auto wb_row = [&](int u) { return wb_row_e(u, rowtab[4 * W * (u / CPN) + (tid >> 4)]); };
    auto sub_on = [](const WbRow& w, int sub) { return !(((sub & 1) && !w.ai) || ((sub & 2) && !w.aj) || ((sub & 4) && !w.ak)); };
    // 1. the wall points, chunk by chunk -- only loads are outstanding when a chunk waits for its old values, and a wavefront
    //    without wall points (most wavefronts of a tile at a wall) passes without waiting at all
    if (near_wall) {
#pragma unroll
        for (int u0 = 0; u0 < NUW; u0 += GW) {
            double oldw[GW][7];
            bool any = false;
#pragma unroll
            for (int g_ = 0; g_ < GW; ++g_) {
                if (u0 + g_ >= NUW) break;
                const WbRow w = wb_row(u0 + g_);
                any = any || (w.mine && (w.ai | w.aj | w.ak));
#pragma unroll
                for (int sub = 1; sub < 8; ++sub) {
                    oldw[g_][sub - 1] = 0.0;
                    if (!w.mine || !sub_on(w, sub)) continue;
                    const int wi = w.gi + ((sub & 1) ? w.ai : 0), wj = w.gj2 + ((sub & 2) ? w.aj : 0), wk = w.gk2 + ((sub & 4) ? w.ak : 0);
                    oldw[g_][sub - 1] = ldp(in + (wi + sx * wj + sxy * wk));
                }
            }
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;
#pragma unroll
            for (int g_ = 0; g_ < GW; ++g_) {
                const int u = u0 + g_;
                if (u >= NUW) break;
                const WbRow w = wb_row(u);
                if (!(w.mine && (w.ai | w.aj | w.ak))) continue;
                const double val0 = lds[T::core_at(4 * W * (u / CPN) + (tid >> 4)) + 3 + w.t];
#pragma unroll
                for (int sub = 1; sub < 8; ++sub) {
                    if (!sub_on(w, sub)) continue;
                    const int wi = w.gi + ((sub & 1) ? w.ai : 0), wj = w.gj2 + ((sub & 2) ? w.aj : 0), wk = w.gk2 + ((sub & 4) ? w.ak : 0);
                    const int nb = __builtin_popcount(sub);
                    const int nh = (int)(wi == nx) + (int)(wj == ny) + (int)(wk == nz);
                    const int mrep = min(nb, 1 + nh);
                    double val = val0;
                    {
#pragma clang fp contract(off)
                        for (int rr = 0; rr < mrep; ++rr) val = val + dx;
                    }
                    const long p = wi + sx * wj + sxy * wk;
                    const double dlt = val - oldw[g_][sub - 1];
                    stp(out + p, val);
                    if constexpr (PUSH) {
                        if (nb_lo && wk < k_push_lo) st_sys(nb_lo + p, val);
                        if (nb_hi && wk >= k_push_hi) st_sys(nb_hi + p, val);
                    }
                    acc = STRICT ? acc + dlt * dlt : __builtin_fma(dlt, dlt, acc);
                }
            }
        }
    }
    // 2. the cells: nothing waits between these stores -- and nothing behind them either: the compiler is told HERE (a wait it
    //    understands, free for a tile away from the walls: nothing is outstanding) that no load is in flight any more.  Without it
    //    the first reuse of a register behind the stores waited for ALL outstanding memory operations (a load of the wall-point
    //    branch counted as possibly pending): the result stores drained before the tile's RMS sum was formed, and the store of that
    //    sum drained again.  Worth 0.3-1 % on grids up to 256^3 (A/B/A/B, profiles/r04_hops.txt), nothing at 512^3.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    // (row-table entries and results of a lane's rows read in ONE batch: row by row, every store followed two dependent LDS reads of
    // its own -- ten LDS round trips in a row between the end of the march and the last store)
    int2 e_wb[NUW];
    double v_wb[NUW];
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        e_wb[u] = rowtab[4 * W * (u / CPN) + (tid >> 4)];
        v_wb[u] = lds[T::core_at(4 * W * (u / CPN) + (tid >> 4)) + 3 + (tid & 15) + 16 * (u % CPN)];
    }
#pragma unroll
    for (int u = 0; u < NUW; ++u) asm volatile("" : "+v"(e_wb[u].x), "+v"(e_wb[u].y), "+v"(v_wb[u]));
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        const WbRow w = wb_row_e(u, e_wb[u]);
        const double val0 = v_wb[u];
        if (w.mine) stp(out_t + ((unsigned)(w.e.x >> 2) + (unsigned)w.gi), val0);
        if constexpr (PUSH) {
            if (w.mine && nb_lo && w.gk2 < k_push_lo) st_sys(nb_lo + org + ((unsigned)(w.e.x >> 2) + (unsigned)w.gi), val0);
            if (w.mine && nb_hi && w.gk2 >= k_push_hi) st_sys(nb_hi + org + ((unsigned)(w.e.x >> 2) + (unsigned)w.gi), val0);
        }
    }
    }
    acc = wave_sum_x(acc);
    if (W > 1) { // fixed-order sum over the wavefronts of the tile
        if (lane == 0) wsum[wave] = acc;
        __syncthreads();
        acc = wsum[0];
        for (int w = 1; w < W; ++w) acc += wsum[w];
    }

    // ---- RMS: per-tile-column running sum along m (deterministic), epilogue by the last tile of the sweep ----
    if (tid == 0) {
        double* slot = a.colsum + (long)gb * ncol + (tj + (long)a.nTj * tk);
        stp(slot, colsum_prev + acc);
        if constexpr (PUSH) { // the sweep's last tile (frame C = nTk - 1) sums the columns of every slab
            const int epi = sk > 0 ? a.nslab - 1 : 0;
            if (epi != a.slab) st_sys(a.peers->all_colsum[epi] + (long)gb * ncol + (tj + (long)a.nTj * tk), colsum_prev + acc);
        }
    }
    if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // results are at the memory side before anyone is told
#ifdef LSF_EXPERIMENTS
    if (a.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSF_PHASE(4);
        if (tid == 0) {
            for (int i_ = 0; i_ < 4; ++i_) atomicAdd(a.dbg + 3 + i_, ph_[i_ + 1] - ph_[i_]);
            atomicAdd(a.dbg + 7, 1ull);
        }
    }
#endif
    if (packed != a.last_packed) return true;
    __syncthreads();
    if (wave != 0) return true;
    const double* cs = a.colsum + (long)gb * ncol;
    double tsum = 0.0;
    for (int p = lane; p < ncol; p += 64) tsum += ldp(cs + p);
    tsum = wave_sum(tsum);
    if (lane == 0) {
        const double rms = __builtin_sqrt(tsum / a.den);
        if constexpr (PUSH) {
            // the epilogues of consecutive sweeps are ordered (the last tile of a sweep waits for every hyperplane of the sweep
            // before), so a stop raised earlier -- here or on another slab, which stored it here before it finished -- is seen
            const bool stop = rms < a.tol || rms != rms || ld_flag_sys(a.ctl + 0) != 0;
            const SlabPeers* pe = a.peers;
            for (int q = 0; q < a.nslab; ++q) {
                if (g < a.trace_cap) st_sys(pe->all_trace[q] + g, rms);
                st_flag_sys(pe->all_ctl[q] + 1, g + 1);
                if (rms != rms) st_flag_sys(pe->all_ctl[q] + 2, 1);
                if (stop) st_flag_sys(pe->all_ctl[q] + 0, 1);
                st_flag_sys(pe->all_verdict[q] + (g - a.g0), stop ? 2 : 1); // the verdict travels in the word the waiters poll
            }
        } else {
            if (g < a.trace_cap) a.trace[g] = rms;
            st_flag(a.ctl + 1, g + 1);
            if (rms < a.tol) st_flag(a.ctl + 0, 1);
            else if (rms != rms) { st_flag(a.ctl + 2, 1); st_flag(a.ctl + 0, 1); }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Helpers of the dataflow launch
// ---------------------------------------------------------------------------------------------------------------------
// x <-> y transposition of a field: dst(j, i, k) = src(i, j, k); src has extents (ex, ey, ez), i unit stride; dst has
// extents (ey, ex, ez), j unit stride.  The dataflow launch marches its tiles along the axis the raster cycle of the
// reference flips most often (y: six of the eight transitions, subs.f90:740-855), because a flip of the MARCH axis costs
// only n / TA time slots of spacing between two sweeps whereas a flip of a cross-section axis costs n / TA + its number of
// tiles; the kernel marches along its unit-stride axis, so the library runs it on the transposed field (reinit_slot_core).
// dst2 (optional): a second copy of the result (phiS = phi on entry, subs.f90:731: one read of phi serves both).
static __global__ __launch_bounds__(256) void k_transpose_xy(const double* __restrict__ src, double* __restrict__ dst, int ex, int ey,
                                                      long planes, double* __restrict__ dst2)
{
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const long pl = (long)ex * ey;
    for (long k = blockIdx.z; k < planes; k += gridDim.z) {
        const double* s = src + k * pl;
        double* d = dst + k * pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + ty + 8 * r, i = i0 + tx;
            if (i < ex && j < ey) t[ty + 8 * r][tx] = s[i + (long)ex * j];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + ty + 8 * r, j = j0 + tx;
            if (i < ex && j < ey) {
                const double v = t[tx][ty + 8 * r];
                d[j + (long)ey * i] = v;
                if (dst2) dst2[k * pl + j + (long)ey * i] = v;
            }
        }
        __syncthreads();
    }
}

// Task list of a batch, built on the device (the list of a 64-sweep batch at 512^3 is 53 MB: building it on the host and
// copying it took longer than several sweeps and landed inside the first call that used a new sweep count).  One block
// per time slot; slot_base[slot] = index of the slot's first entry; the slot holds hyperplane P = slot - start[q] of
// every sweep q of the batch with 0 <= P < np, in increasing q (exactly what the slot schedule launches, launch after
// launch).
static __global__ __launch_bounds__(256) void k_build_order(uint2* __restrict__ order, const uint32_t* __restrict__ tiles,
                                                     const int* __restrict__ plane_off, const int* __restrict__ start,
                                                     const unsigned* __restrict__ slot_base, int ns, int np)
{
    const int slot = blockIdx.x;
    unsigned base = slot_base[slot];
    for (int q = 0; q < ns; ++q) {
        const int P = slot - start[q];
        if (P < 0) break; // start[] is increasing
        if (P >= np) continue;
        const int o = plane_off[P], cnt = plane_off[P + 1] - o;
        const unsigned tag = (unsigned)q | ((unsigned)P << DF_SWEEP_BITS);
        for (int i = threadIdx.x; i < cnt; i += 256) order[base + i] = make_uint2(tiles[o + i], tag);
        base += (unsigned)cnt;
    }
}

// Slot schedule: one launch per time slot, one block per tile of the slot (dependencies resolved by launch order).
template <int TA, int WY, int WZ, int BY, bool STRICT>
__global__ __launch_bounds__(64 * WY * WZ) __attribute__((amdgpu_waves_per_eu(BY == 16 ? (WY * WZ == 1 ? 1 : LSF_WAVES16) : (WY == 2 && WZ == 2 ? (TA == 32 ? 2 : 5) : 1)))) void k_reinit_gs_skew(GsArgs a)
{
    const int bx = (int)blockIdx.x;
    const int seg = (bx >= a.seg_end[0]) + (bx >= a.seg_end[1]) + (bx >= a.seg_end[2]);
    const uint32_t packed = a.seg_tiles[seg][bx - (seg ? a.seg_end[seg - 1] : 0)];
    if (ld_flag(a.ctl + 0) != 0) return; // converged or failed in an earlier launch
#ifdef LSF_EXPERIMENTS
    // work-term probe (profiles/r05_ring_probe.txt): the two tiles of a CU start every phase together when every tile of a sweep
    // is in one launch; the SECOND block to arrive on a CU sleeps probe_us microseconds so that one loads while the other marches
    if (a.ticket && a.probe_us > 0) {
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            const int ord = atomicAdd(a.ticket + ((xcc & 7u) * 256u + ((hw >> 8) & 255u)), 1);
            if (ord == 1) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * (unsigned)a.probe_us) __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
#endif
#if defined(LSF_EXPERIMENTS) && defined(LSF_PROBE_LDS_PAD)
    __shared__ int lds_pad[LSF_PROBE_LDS_PAD / 4]; // work-term probe: one tile per CU (its phases when it has the CU to itself)
    if (a.nx < 0) lds_pad[threadIdx.x] = a.ny, a.ctl[5] = lds_pad[threadIdx.x ^ 1];
#endif
    const SkPre pre = sk_prefetch<TA, WY, WZ, BY>(a);
    __shared__ SkShared<SkTile<TA, WY, WZ, BY>> sm;
    skew_tile<TA, WY, WZ, BY, STRICT, false>(sm, a, packed, a.seg_g[seg], a.seg_sign[seg][0], a.seg_sign[seg][1], a.seg_sign[seg][2], pre,
                                             [] { return true; }); // every predecessor ran in an earlier launch
}

// Dataflow schedule: ONE launch per batch of sweeps, one block per tile.  The tiles of the batch form a list in slot
// order (the order the slot schedule launches them in); a block takes the next entry from a ticket counter when it
// starts (blocks are not assumed to start in index order) and waits until
//   (a) its up to three upstream tiles of the same sweep are done                  tile_done[s][(m-1,B,C)], ...
//   (b) sweep s-1 has completed the hyperplanes within stencil reach of this one   planes_done[s-1] >= min(P + H[s], np)
//   (c) sweep s-nbuf, whose result this sweep overwrites, has its stop verdict     planes_done[s-nbuf] == np + 1 (np + 2: stop)
// Every entry a block waits for precedes it in the list and was taken by a block that is running or done, so there
// is no deadlock whatever the dispatch order or the number of resident blocks; every spin is bounded (ctl[2] = 2 on
// time-out).  A finished tile drains its write-through stores, raises its flag and counts itself into
// plane_cnt[s][P]; the last tile of a hyperplane publishes planes_done[s] = P + 1.
template <int TA, int WY, int WZ, int BY, bool STRICT>
__global__ __launch_bounds__(64 * WY * WZ) __attribute__((amdgpu_waves_per_eu(BY == 16 ? (WY * WZ == 1 ? 1 : LSF_WAVES16) : (WY == 2 && WZ == 2 ? (TA == 32 ? 2 : (STRICT ? LSF_STRICT22_WAVES : 5)) : 1)))) void k_reinit_gs_persist(GsArgs a)
{
    using T = SkTile<TA, WY, WZ, BY>;
    __shared__ SkShared<T> sm;
    __shared__ int sh_task[8]; // packed tile, s | P << DF_SWEEP_BITS, go flag, raster signs of the sweep, go flag of stage 2
    const int tid = threadIdx.x;
    const int np = a.np;
    const int nM = a.nM;                       // tile_done[s] is indexed m + nM * (B + nTj * C)
    const long per_sweep = (long)nM * a.nTj * a.nTk;
    {
        // (time stamps of LSF_TRACE_TILES only where asked for: s_memrealtime is a scalar memory read, and the first
        // s_waitcnt lgkmcnt(0) behind it -- the one in front of the tile's first LDS access -- waits for it)
        const unsigned long long tsA = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const SkPre pre = sk_prefetch<TA, WY, WZ, BY>(a); // in flight while the block takes its ticket and waits for its tile
        // thread 0 only: the flags of the up to three upstream tiles (condition (a), awaited between the two load stages)
        const int* const set_word = a.ctl + 4; // a word that is never 0 (host): stands in for an upstream tile that does not exist
        const int *w0 = set_word, *w1 = set_word, *w2 = set_word;
        unsigned long long t0 = 0;
        bool up_ready = false;
        auto give_up = [&]() { // converged, NaN or time-out elsewhere: skip, and let the blocks still to come leave at once
            __hip_atomic_fetch_max(a.ticket, (int)a.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (tid == 0) {
            const long t = (long)__hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int go = 0;
            uint2 e = make_uint2(0u, 0u);
            if (t < a.total) {
                e = a.order[t];
                const int s = (int)(e.y & (unsigned)(DF_BATCH - 1)), P = (int)(e.y >> DF_SWEEP_BITS);
                const int m = e.x & 0x3ff, B = (e.x >> 10) & 0x3ff, C = (e.x >> 20) & 0x3ff;
                auto m_lo = [&](int B_, int C_) { return (T::NYT * B_ + T::NZT * C_) / TA; };
                auto m_hi = [&](int B_, int C_) { return (T::NYT * B_ + T::NYT - 1 + T::NZT * C_ + T::NZT - 1 + a.nx - 2) / TA; };
                const int* td = a.tile_done + s * per_sweep;
                w0 = m - 1 >= m_lo(B, C) ? td + (m - 1) + (long)nM * (B + (long)a.nTj * C) : set_word;
                w1 = (B >= 1 && m >= m_lo(B - 1, C) && m <= m_hi(B - 1, C)) ? td + m + (long)nM * (B - 1 + (long)a.nTj * C) : set_word;
                w2 = (C >= 1 && m >= m_lo(B, C - 1) && m <= m_hi(B, C - 1)) ? td + m + (long)nM * (B + (long)a.nTj * (C - 1)) : set_word;
                const int* pd = a.planes_done;
                const int4 swp = *(const int4*)(a.sweep_tab + 4 * s); // signs and spacing of the sweep: one request
                const int need3 = s < a.nbuf ? 0 : np + 1;
                // absent conditions point at a word that always passes (the stop flag's neighbour ctl[1] >= 0)
                const int* always = a.ctl + 1;
                const int* p3 = s == 0 ? always : pd + s - 1;
                const int* p4 = s < a.nbuf ? always : pd + s - a.nbuf;
                t0 = __builtin_amdgcn_s_memrealtime();
                go = 1;
                // stage 1: conditions (b) and (c) -- what this tile reads of the previous sweep is final
                for (;;) {
                    // independent loads in flight at once, then one test
                    const int vstop = ld_flag(a.ctl + 0);
                    const int v3 = ld_flag(p3), v4 = ld_flag(p4);
                    // ... and a first look at the upstream tiles of THIS sweep (condition (a)) in the same round trip: when they
                    // are done already the second load stage follows the first at once, instead of one round trip for the
                    // look and one for the loads behind the first stage (round 4, one box, A/B/A/B: 2.585-2.593 against
                    // 2.600-2.602 ms per 512^3 sweep, STRICT 4.531-4.533 against 4.543-4.552: the upstream tiles are rarely done
                    // that early -- a sweep advances as a front -- but the look is free)
                    const int u0 = ld_flag(w0), u1 = ld_flag(w1), u2 = ld_flag(w2);
                    up_ready = (u0 != 0) & (u1 != 0) & (u2 != 0);
                    // (the spacing of the sweep is first used HERE, behind the flag loads: computed in front of the loop it made the
                    // table load a round trip of its own between the order entry and the first look)
                    int h = swp.w;
                    asm volatile("" : "+v"(h));
                    const int need1 = s == 0 ? 0 : min(P + h, np);
                    if (vstop != 0) {
                        go = 2;
                        give_up();
                        break;
                    }
                    if ((v3 >= need1) & (v4 >= need3)) {
                        // The verdict of the sweep whose result this one overwrites travels IN the word that releases
                        // it (np + 2 = that sweep, or an earlier one, raised the stop flag), so it cannot be missed by
                        // a stop-flag load that was issued before the epilogue's store and a planes_done load that was
                        // issued after it (the loads above are independent and relaxed).
                        if (s >= a.nbuf && v4 == np + 2) {
                            go = 2;
                            give_up();
                        }
                        break;
                    }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        st_flag(a.ctl + 2, 2);
                        st_flag(a.ctl + 0, 1);
                        go = 2;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(LSF_POLL_SLEEP); // ~0.5 us between looks; 8..64 measured within 2 % of each other
                }
                sh_task[3] = swp.x, sh_task[4] = swp.y, sh_task[5] = swp.z;
            }
            sh_task[0] = (int)e.x, sh_task[1] = (int)e.y, sh_task[2] = go, sh_task[7] = up_ready ? 1 : 0;
        }
        __syncthreads();
        // wave-uniform values: keep them in scalar registers (an LDS read alone would make them look divergent)
        auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        const uint32_t packed = (uint32_t)uni(sh_task[0]);
        const int sP = uni(sh_task[1]);
        int go = uni(sh_task[2]);
        const bool upstream_done = uni(sh_task[7]) != 0; // seen by thread 0 before the first load stage: nothing to wait for
        const int s = sP & (DF_BATCH - 1), P = (int)((unsigned)sP >> DF_SWEEP_BITS);
        if (go == 0) return;
        // the tile's own flag: its address is formed HERE and kept (scalar registers) -- formed behind the drain of the tile's stores
        // it cost two scalar loads of kernel arguments and their wait between the last barrier and the flag
        typedef __attribute__((address_space(1))) int* GlobalIntPtr; // (a pointer that has been through an asm statement is a flat one otherwise)
        GlobalIntPtr my_flag;
        {
            const int m_ = packed & 0x3ff, B_ = (packed >> 10) & 0x3ff, C_ = (packed >> 20) & 0x3ff;
            my_flag = (GlobalIntPtr)(a.tile_done + s * per_sweep + m_ + (long)nM * (B_ + (long)a.nTj * C_));
            asm volatile("" : "+s"(my_flag));
        }
        const unsigned long long tsB = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
        // stage 2, called by skew_tile with its stage-1 loads in flight: condition (a) -- the upstream tiles of this sweep
        auto wait_upstream = [&]() -> bool {
            if (upstream_done) return true; // (uniform over the block: no barrier needed either)
            if (tid == 0) {
                int go2 = 1;
                for (;;) {
                    const int vstop = ld_flag(a.ctl + 0);
                    // (three unconditional loads in flight together: `w ? load : 1` compiled to a branch, a load and a wait per
                    // flag -- three round trips past the caches one after the other between two looks)
                    const int v0 = ld_flag(w0), v1 = ld_flag(w1), v2 = ld_flag(w2);
                    if (vstop != 0) {
                        go2 = 2;
                        give_up();
                        break;
                    }
                    if ((v0 != 0) & (v1 != 0) & (v2 != 0)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        st_flag(a.ctl + 2, 2);
                        st_flag(a.ctl + 0, 1);
                        go2 = 2;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(LSF_POLL_SLEEP);
                }
                sh_task[6] = go2;
            }
            // LDS hand-off of the verdict only: the stage-1 loads stay in flight across the barrier
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            return uni(sh_task[6]) == 1;
        };
        if (go == 1) {
#ifdef LSF_EXPERIMENTS
            // LSF_PROBE_EARLY_FLAG = t: the tile raises its flag in front of marching step t, with nothing of it stored yet -- the
            // fields come out wrong; the launch's time is a LOWER bound of what a hand-off finer than a tile could reach
            // (profiles/r05_gs_probes.txt, item 7)
            auto early = [&](int t) {
                if (a.probe_early > 0 && t == a.probe_early && tid == 0) __hip_atomic_store(my_flag, 1, __ATOMIC_RELAXED, LSF_FLAG_ST_SCOPE);
            };
            if (!skew_tile<TA, WY, WZ, BY, STRICT, true>(sm, a, packed, a.g0 + s, uni(sh_task[3]), uni(sh_task[4]), uni(sh_task[5]), pre, wait_upstream, early))
                go = 2;
#else
            if (!skew_tile<TA, WY, WZ, BY, STRICT, true>(sm, a, packed, a.g0 + s, uni(sh_task[3]), uni(sh_task[4]), uni(sh_task[5]), pre, wait_upstream))
                go = 2;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads(); // every wave of the tile has drained its stores (and left the LDS image)
        if (tid == 0 && go == 1) {
            __hip_atomic_store(my_flag, 1, __ATOMIC_RELAXED, LSF_FLAG_ST_SCOPE);
            const int done = __hip_atomic_fetch_add(a.plane_cnt + s * np + P, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (done == a.plane_size[P]) {
                // Hyperplanes may complete out of order now; planes_done[s] counts the LEADING complete ones.  Whoever
                // completes a hyperplane pushes the counter as far as it goes: a hyperplane completed while the counter
                // was still behind it is picked up by the thread that moves the counter onto it (its tile count was
                // final before that thread looks), so nothing is lost whichever of the two runs first.
                for (;;) {
                    int lead = ld_flag(a.planes_done + s);
                    if (lead >= np || ld_flag(a.plane_cnt + s * np + lead) < a.plane_size[lead]) break;
                    // the tile that completes the last hyperplane has run the sweep epilogue (stop flag stored and
                    // drained) before counting itself, and this thread has seen that count: the flag read here is final
                    // for this sweep.  np + 1 = finished, go on; np + 2 = finished, stop (condition (c) of the waiters).
                    int next = lead + 1;
                    if (next == np) next = np + 1 + (ld_flag(a.ctl + 0) != 0 ? 1 : 0);
                    __hip_atomic_compare_exchange_strong(a.planes_done + s, &lead, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (a.dbg) {
                const unsigned long long tsC = __builtin_amdgcn_s_memrealtime();
                atomicAdd(a.dbg + 0, tsB - tsA);
                atomicAdd(a.dbg + 1, tsC - tsB);
                atomicAdd(a.dbg + 2, 1ull);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact ordering across z slabs: the dataflow schedule above, one launch PER SLAB (one device each; several on one device
// for the rehearsal), all working on the same tile graph.  A slab owns the tile columns tk_lo <= tk < tk_hi; its task list
// is the global list (slot order) restricted to them, so every tile a block waits for -- on this slab or on a neighbour --
// precedes it in the global order and has been taken by a resident block of its slab: no deadlock as long as every launch
// is resident at once, which is why this kernel is a loop over tickets with a grid the host sizes to the device's share
// (k_reinit_gs_persist is one block per tile).  Differences from the single launch:
//   (a)  the flag of an upstream tile across the cut is raised HERE by the neighbour's block, after it has stored the three
//        planes next to the cut into this slab's buffer and drained them (skew_tile<PUSH>);
//   (b)  "sweep s - 1 has completed the hyperplanes within reach" is asked of this slab's own hyperplane counter AND of the
//        mirrors of its two neighbours' counters (a tile reaches at most into the adjacent slab), pushed by atomic max;
//   (c)  the stop verdict of sweep s - nbuf is its own word per sweep, stored on every slab by the epilogue;
//   the stop flag, the NaN / time-out status, the sweep count and the RMS trace are stored on every slab as well.
// Every cross-slab word and value is stored and loaded at system scope (write-through to the fabric, loads past the caches);
// stores are drained (s_waitcnt vmcnt(0)) before the flag that announces them is raised.
// ---------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_build_order_slab(uint2* __restrict__ order, const uint32_t* __restrict__ tiles_pos,
                                                          const uint32_t* __restrict__ tiles_neg, const int* __restrict__ off_pos,
                                                          const int* __restrict__ off_neg, const int* __restrict__ sweep_tab,
                                                          const int* __restrict__ start, const unsigned* __restrict__ slot_base, int ns, int np)
{
    const int slot = blockIdx.x;
    unsigned base = slot_base[slot];
    for (int q = 0; q < ns; ++q) {
        const int P = slot - start[q];
        if (P < 0) break; // start[] is increasing
        if (P >= np) continue;
        const bool pos = sweep_tab[4 * q + 2] > 0;
        const int* po = pos ? off_pos : off_neg;
        const uint32_t* tl = pos ? tiles_pos : tiles_neg;
        const int o = po[P], cnt = po[P + 1] - o;
        const unsigned tag = (unsigned)q | ((unsigned)P << DF_SWEEP_BITS);
        for (int i = threadIdx.x; i < cnt; i += 256) order[base + i] = make_uint2(tl[o + i], tag);
        base += (unsigned)cnt;
    }
}

// LOOP = true: a loop over tickets, for slabs that SHARE a device (the rehearsal): the launches wait for each other, so each must
// be resident as a whole.  2 x 2 wavefronts, three lanes per cell: four tiles per CU (128 registers), not the five of
// k_reinit_gs_persist -- inside the ticket loop the compiler needs ~106 registers, at five per CU (96) it spills ten to scratch:
// 3.09 against 2.95 ms per 512^3 sweep on one box (k_reinit_gs_persist: 2.75).
// LOOP = false: one block per tile like k_reinit_gs_persist, for a slab that has its device to itself (what a node runs): the
// blocks of a launch take their tickets in list order as they become resident, so the earliest unfinished tile of a device is
// always held by a resident block, and what it waits for on OTHER devices is held by resident blocks there -- no deadlock
// without the loop, no loop-carried registers, five tiles per CU.
#ifndef LSF_SLAB_WAVES
#define LSF_SLAB_WAVES 4
#endif
template <int TA, int WY, int WZ, int BY, bool STRICT, bool LOOP>
__global__ __launch_bounds__(64 * WY * WZ) __attribute__((amdgpu_waves_per_eu(BY == 16 ? (WY * WZ == 1 ? 1 : LSF_WAVES16) : (WY == 2 && WZ == 2 ? (LOOP ? LSF_SLAB_WAVES : 5) : 1)))) void k_reinit_gs_slab(GsArgs args_)
{
    using T = SkTile<TA, WY, WZ, BY>;
    __shared__ SkShared<T> sm;
    __shared__ int sh_task[8]; // as in k_reinit_gs_persist
    __shared__ unsigned long long sh_wait[4]; // thread 0's upstream flags and start time: parked here while the tile loads (registers)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid)); // (and nothing that depends on the thread index only either: see skew_tile)
        // The arguments are read afresh in every iteration, through a pointer the compiler cannot see through: left to itself it
        // loads all of them in front of the loop and keeps them in registers across it -- 73 scalar registers spilled to vector
        // lanes, those to 272 bytes of scratch per lane, 38 % more time per tile than k_reinit_gs_persist.  (GsArgs is the
        // kernel's only argument: it starts the kernarg segment.)
        typedef const __attribute__((address_space(4))) GsArgs* KArg;
        KArg kp = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const GsArgs& a = *(const GsArgs*)kp;
        const int np = a.np;
        const int nM = a.nM;
        const long per_sweep = (long)nM * a.nTj * a.nTk;
        const SkPre pre = sk_prefetch<TA, WY, WZ, BY>(a, tid); // in flight while the block takes its ticket and waits for its tile
        auto give_up = [&]() { __hip_atomic_fetch_max(a.ticket, (int)a.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        // tell every slab, so that none of them spins on for a tile that will not come; the first tile of this slab to give up leaves
        // what it was waiting for in ctl[8..15] (the host puts it into the error message)
        auto time_out = [&](unsigned packed_, int sP_, int stage, int x0, int x1, int x2, int x3, int x4) {
            int expect = 0;
            if (__hip_atomic_compare_exchange_strong(a.ctl + 8, &expect, stage, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st_flag(a.ctl + 9, (int)packed_), st_flag(a.ctl + 10, sP_), st_flag(a.ctl + 11, x0), st_flag(a.ctl + 12, x1);
                st_flag(a.ctl + 13, x2), st_flag(a.ctl + 14, x3), st_flag(a.ctl + 15, x4);
            }
            for (int q = 0; q < a.nslab; ++q) st_flag_sys(a.peers->all_ctl[q] + 2, 2), st_flag_sys(a.peers->all_ctl[q] + 0, 1);
        };
        if (tid == 0) {
            const long t = (long)__hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int go = 0;
            uint2 e = make_uint2(0u, 0u);
            if (t < a.total) {
                e = a.order[t];
                const int s = (int)(e.y & (unsigned)(DF_BATCH - 1)), P = (int)(e.y >> DF_SWEEP_BITS);
                const int m = e.x & 0x3ff, B = (e.x >> 10) & 0x3ff, C = (e.x >> 20) & 0x3ff;
                auto m_lo = [&](int B_, int C_) { return (T::NYT * B_ + T::NZT * C_) / TA; };
                auto m_hi = [&](int B_, int C_) { return (T::NYT * B_ + T::NYT - 1 + T::NZT * C_ + T::NZT - 1 + a.nx - 2) / TA; };
                const int* td = a.tile_done + s * per_sweep; // the neighbour raises the flags of its tiles next to the cut here
                const int* const set_word = a.ctl + 4; // INT_MAX (host): stands in for an upstream tile that does not exist
                const int* w0 = m - 1 >= m_lo(B, C) ? td + (m - 1) + (long)nM * (B + (long)a.nTj * C) : set_word;
                const int* w1 = (B >= 1 && m >= m_lo(B - 1, C) && m <= m_hi(B - 1, C)) ? td + m + (long)nM * (B - 1 + (long)a.nTj * C) : set_word;
                const int* w2 = (C >= 1 && m >= m_lo(B, C - 1) && m <= m_hi(B, C - 1)) ? td + m + (long)nM * (B + (long)a.nTj * (C - 1)) : set_word;
                const int4 swp = *(const int4*)(a.sweep_tab + 4 * s); // signs and spacing of the sweep: one request
                const int* always = a.ctl + 4; // INT_MAX (host)
                const int* p3 = s == 0 ? always : a.planes_done + s - 1;
                const int* p3l = (s == 0 || !a.nb_pd[0]) ? always : a.pd_of_nb[0] + s - 1;
                const int* p3h = (s == 0 || !a.nb_pd[1]) ? always : a.pd_of_nb[1] + s - 1;
                const int* p4 = s < a.nbuf ? always : a.verdict + s - a.nbuf;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                sh_wait[0] = (unsigned long long)w0, sh_wait[1] = (unsigned long long)w1, sh_wait[2] = (unsigned long long)w2, sh_wait[3] = t0;
                go = 1;
                for (;;) { // stage 1: conditions (b) and (c)
                    const int vstop = ld_flag_sys(a.ctl + 0);
                    const int v3 = ld_flag(p3), v3l = ld_flag_sys(p3l), v3h = ld_flag_sys(p3h), v4 = ld_flag_sys(p4);
                    int h = swp.w; // first used behind the flag loads: the table load shares their round trip (see k_reinit_gs_persist)
                    asm volatile("" : "+v"(h));
                    const int need1 = s == 0 ? 0 : min(P + h, np);
                    if (vstop != 0) {
                        go = 2;
                        give_up();
                        break;
                    }
                    if ((v3 >= need1) & (v3l >= need1) & (v3h >= need1) & (v4 != 0)) {
                        if (s >= a.nbuf && v4 == 2) { // the sweep whose result this one would overwrite was the last one
                            go = 2;
                            give_up();
                        }
                        break;
                    }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        time_out(e.x, (int)e.y, 1, need1, v3, v3l, v3h, v4);
                        go = 2;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(LSF_POLL_SLEEP);
                }
                sh_task[3] = swp.x, sh_task[4] = swp.y, sh_task[5] = swp.z;
            }
            sh_task[0] = (int)e.x, sh_task[1] = (int)e.y, sh_task[2] = go;
        }
        __syncthreads();
        const uint32_t packed = (uint32_t)uni(sh_task[0]);
        const int sP = uni(sh_task[1]);
        int go = uni(sh_task[2]);
        const int s = sP & (DF_BATCH - 1), P = (int)((unsigned)sP >> DF_SWEEP_BITS);
        if (go == 0) return; // the list is exhausted (or was cut short: stop, time-out)
        const int sk = uni(sh_task[5]);
        auto wait_upstream = [&]() -> bool { // stage 2: condition (a)
            if (tid == 0) {
                int go2 = 1;
                const int *w0 = (const int*)sh_wait[0], *w1 = (const int*)sh_wait[1], *w2 = (const int*)sh_wait[2];
                const unsigned long long t0 = sh_wait[3];
                for (;;) {
                    const int vstop = ld_flag_sys(a.ctl + 0);
                    const int v0 = ld_flag(w0), v1 = ld_flag(w1), v2 = ld_flag_sys(w2); // (unconditional: see k_reinit_gs_persist)
                    if (vstop != 0) {
                        go2 = 2;
                        give_up();
                        break;
                    }
                    if ((v0 != 0) & (v1 != 0) & (v2 != 0)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                        time_out(packed, sP, 2, v0, v1, v2, 0, 0);
                        go2 = 2;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(LSF_POLL_SLEEP);
                }
                sh_task[6] = go2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            return uni(sh_task[6]) == 1;
        };
        if (go == 1) {
            if (!skew_tile<TA, WY, WZ, BY, STRICT, true, true>(sm, a, packed, a.g0 + s, uni(sh_task[3]), uni(sh_task[4]), sk, pre, wait_upstream))
                go = 2;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads(); // every wave of the tile has drained its stores, the neighbours' copies included
        if (tid == 0 && go == 1) {
            const int m = packed & 0x3ff, B = (packed >> 10) & 0x3ff, C = (packed >> 20) & 0x3ff;
            const long fl = s * per_sweep + m + (long)nM * (B + (long)a.nTj * C);
            st_flag(a.tile_done + fl, 1);
            const int tk = sk > 0 ? C : a.nTk - 1 - C;
            if (tk == a.tk_lo && a.nb_tile_done[0]) st_flag_sys(a.nb_tile_done[0] + fl, 1);
            if (tk == a.tk_hi - 1 && a.nb_tile_done[1]) st_flag_sys(a.nb_tile_done[1] + fl, 1);
            const int* psz = sk > 0 ? a.plane_size : a.plane_size_neg;
            const int done = __hip_atomic_fetch_add(a.plane_cnt + s * np + P, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (done == psz[P]) {
                // leading complete hyperplanes of THIS slab (hyperplanes in which it has no tile count as complete)
                for (;;) {
                    int lead = ld_flag(a.planes_done + s);
                    if (lead >= np || ld_flag(a.plane_cnt + s * np + lead) < psz[lead]) break;
                    const int next = lead + 1;
                    if (__hip_atomic_compare_exchange_strong(a.planes_done + s, &lead, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        for (int q = 0; q < 2; ++q)
                            if (a.nb_pd[q]) __hip_atomic_fetch_max(a.nb_pd[q] + s, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
        // thread 0 rewrites sh_task only after the barrier above, which every wave reaches after its last read of it
        if constexpr (!LOOP) return;
    }
}

} // namespace lsf
