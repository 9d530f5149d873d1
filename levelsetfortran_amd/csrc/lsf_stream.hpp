// lsf_stream.hpp -- the dataflow launch of the exact ordering with column continuation: k_reinit_gs_stream and the function that
// runs one tile of it (stream_tile).  Included by lsf_stream.hip only; the tile itself is skew_tile of lsf_skew.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsf_skew.hpp"

namespace lsf {

// ---------------------------------------------------------------------------------------------------------------------
// Dataflow schedule with COLUMN CONTINUATION (opt-in: LSF_GS_STREAM=1; the default launch is k_reinit_gs_persist -- this one is
// bit-identical and 9-20 % slower, for the reasons DESIGN.md section 4.1 "Round 4" and profiles/r04_stream_ab.txt give).
//
// A tile (m, B, C) of k_reinit_gs_persist pays, besides its march: a ticket, the row table, the load of 22 (18) entries per
// row, the drain of its write-through stores, its flag -- and the block that takes tile (m + 1, B, C) pays the look at that flag
// and loads six entries per bundle row that were in this block's LDS a moment ago.  Here a block that has finished tile m of a
// column GOES ON with tile m + 1 of the same column when it can: the LDS image moves along by one tile length (skew_tile,
// `cont`), the table of rows advances, 16 entries per row are loaded, and no flag round trip separates the two marches.
//
// Who runs what.  tile_done[s][tile] has three states: 0 free, 1 claimed, 2 done (claims are compare-and-swaps 0 -> 1; waiters
// want 2).  The launch is a fixed number of resident blocks, each a loop:
//   * ACQUIRE: claim the next free tile of the batch's task list (slot order, as before).  A ticket (a.ticket) is a position of
//     the list nobody else gets; the 64 lanes of a wavefront look at the 64 entries from it on, and the counter jumps over the
//     run of entries that continuing blocks have claimed already (claims are permanent: no free entry is ever skipped).
//   * run the tile (waits as in k_reinit_gs_persist: conditions (b), (c), then (a) between the two load stages), publish it;
//   * CONTINUE with (m + 1, B, C) if it exists, sweep s - 1 has already passed its neighbourhood (condition (b), checked, not
//     waited for: by the hyperplane counter or, exactly, by stream_prev_sweep_past), its two cross upstream tiles (m + 1, B - 1, C),
//     (m + 1, B, C - 1) are DONE (LSF_GS_CONT=2, default) or at least CLAIMED (LSF_GS_CONT=1: the block then waits for them
//     between the load stages, as usual), and the claim of the tile itself succeeds.  Otherwise ACQUIRE.
// No deadlock: a tile claimed by ACQUIRE has every predecessor claimed or handed out as a ticket to a live block that is about
// to claim it (they all precede it in the list, and every position in front of the counter is one or the other); a tile claimed
// by CONTINUE has its predecessors claimed by the rule above (its own column's by this block).  So every tile anybody waits for is held by a live block, and the unfinished claimed tile that comes first in the
// list waits for nothing: its holder finishes it (a block holds at most the tile it runs).  Every spin is bounded as before.
// The RMS sums, the hyperplane counters, the stop verdict and the epilogue are those of k_reinit_gs_persist: same bits.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef LSF_STREAM_FINE
#define LSF_STREAM_FINE 1 // a block that runs down a column asks the tiles of the previous sweep themselves (stream_prev_sweep_past)
#endif
#ifndef LSF_STREAM_WAVES
#define LSF_STREAM_WAVES 5 // three lanes per cell, 2 x 2 wavefronts: tiles per CU (94 registers + 104 bytes of scratch; 4: 120 registers)
#endif
// LDS of the launch, at namespace scope: the tile runs in a function of its own (below), which shares it with the kernel
struct StreamCtl {
    int task[12];               // packed tile, s | P << DF_SWEEP_BITS, go flag, raster signs (3), go flag of stage 2, continue flag, spacing
    unsigned long long wait[6]; // thread 0's upstream flags and start time, time stamps of LSF_TRACE_TILES
};
template <class T>
__shared__ SkShared<T> g_stream_sm;
template <class T>
__shared__ StreamCtl g_stream_ctl;

// Has sweep s - 1 left the neighbourhood of tile (m, fB, fC) of sweep s for good?  The launch's condition (b) asks the count of
// leading complete hyperplanes of that sweep -- right for the order of the list, but far more than a tile needs once a block
// runs ahead of the list down a column: after a flip of the march axis the previous sweep passed the column's far tiles long
// before its hyperplane count says so.  Exact form: every tile of sweep s - 1 that holds a cell of this tile's LDS image must be
// done.  The image is five boxes in the frame (bundle rows, four halo groups), each inside one tile column; the skew coordinate
// of sweep s - 1 is linear in the frame coordinates (b, c, entry k), so its maximum over a box is at a corner, and in a column
// the tiles of a sweep finish in the order of their m: one flag per box.  Only for tiles whose image holds interior cells only
// (deep, wide path).  Thread 0 only.
template <class T, int TA>
__device__ __forceinline__ bool stream_prev_sweep_past(const GsArgs& a, int s, int m, int fB, int fC, long per_sweep)
{
    if (s == 0) return true; // the sweeps of earlier launches are complete
    const int4 cur = *(const int4*)(a.sweep_tab + 4 * s), prv = *(const int4*)(a.sweep_tab + 4 * (s - 1));
    const int X0 = TA * m - T::NYT * fB - T::NZT * fC;
    // frame coordinates of a cell of the image: Fx = X0 - b - c + k - 3, Fy = NYT fB + b, Fz = NZT fC + c;  in the frame of the
    // previous sweep an axis whose direction flipped counts from the other end: Fx' = (nx - 2) - Fx along the march axis, and
    // Fy' = (NYT nTj - 1) - Fy, Fz' = (NZT nTk - 1) - Fz across it -- the tiles are anchored at the low wall, so against an axis
    // the frame starts at the far end of the LAST tile, partial or not (for the rows of a partial last column, which a tile's
    // halo may reach, that over-estimates the coordinate: a later tile of the previous sweep is asked, never an earlier one)
    const int ex = cur.x == prv.x ? 1 : -1, ey = cur.y == prv.y ? 1 : -1, ez = cur.z == prv.z ? 1 : -1;
    const int c0 = (ex > 0 ? X0 - 3 : a.nx - 2 - X0 + 3) + (ey > 0 ? T::NYT * fB : T::NYT * a.nTj - 1 - T::NYT * fB) +
                   (ez > 0 ? T::NZT * fC : T::NZT * a.nTk - 1 - T::NZT * fC);
    const int cb = -ex + ey, cc = -ex + ez, ck = ex; // d S' / d b, d c, d k
    const int tj = cur.y > 0 ? fB : a.nTj - 1 - fB, tk = cur.z > 0 ? fC : a.nTk - 1 - fC;
    const int dj = cur.y > 0 ? 1 : -1, dk = cur.z > 0 ? 1 : -1; // absolute column step of one frame column
    auto box_done = [&](int b0, int b1, int c0_, int c1, int k0, int k1, int tj2, int tk2) {
        const int smax = c0 + (cb > 0 ? cb * b1 : cb * b0) + (cc > 0 ? cc * c1 : cc * c0_) + (ck > 0 ? ck * k1 : ck * k0);
        const int mp = min(smax / TA, a.nM - 1); // smax >= 0: the cells are interior cells (beyond the column's last tile: a flag nobody raises)
        const int fBp = prv.y > 0 ? tj2 : a.nTj - 1 - tj2, fCp = prv.z > 0 ? tk2 : a.nTk - 1 - tk2;
        return ld_flag(a.tile_done + (s - 1) * per_sweep + mp + (long)a.nM * (fBp + (long)a.nTj * fCp));
    };
    const int v0 = box_done(0, T::NYT - 1, 0, T::NZT - 1, 0, T::RA - 1, tj, tk);
    const int v1 = box_done(-3, -1, 0, T::NZT - 1, 0, T::RH - 1, tj - dj, tk);
    const int v2 = box_done(T::NYT, T::NYT + 2, 0, T::NZT - 1, 4, T::RA - 1, tj + dj, tk);
    const int v3 = box_done(0, T::NYT - 1, -3, -1, 0, T::RH - 1, tj, tk - dk);
    const int v4 = box_done(0, T::NYT - 1, T::NZT, T::NZT + 2, 4, T::RA - 1, tj, tk + dk);
    return (v0 >= 2) & (v1 >= 2) & (v2 >= 2) & (v3 >= 2) & (v4 >= 2);
}

// One tile of the launch: waits, the tile itself (skew_tile), its flag, and the decision about the next tile of the column.
// NOT inlined: as the body of the kernel's loop over tiles the compiler moved everything that depends only on the kernel's
// arguments or the thread index in front of the loop and kept it in registers across the tile (230 vector registers, scalar
// registers spilled to vector lanes in the wall code); a function that runs a whole chain of tiles in a loop of its own does
// the same (256 registers, 832 bytes of scratch per lane).  One call per tile is compiled like the single tile of
// k_reinit_gs_persist.  Returns 0 (stop, NaN, time-out: leave), 1 (done: acquire the next tile from the list) or 2 (done, and
// task[0], task[1] hold the next tile of this column, claimed: call again with cont = 1).
template <int TA, int WY, int WZ, int BY, bool STRICT>
__device__ __noinline__ int stream_tile(int cont_, unsigned karg_lo, unsigned karg_hi)
{
    using T = SkTile<TA, WY, WZ, BY>;
    // the kernel's argument block (the intrinsic that returns its address is valid in a kernel only: the kernel passes it on), through
    // the constant address space so that its fields are scalar loads
    typedef const __attribute__((address_space(4))) GsArgs* KArg;
    const unsigned long long kaddr = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)karg_hi) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)karg_lo);
    const GsArgs& a = *(const GsArgs*)(KArg)kaddr;
    SkShared<T>& sm = g_stream_sm<T>;
    int* const sh_task = g_stream_ctl<T>.task;
    unsigned long long* const sh_wait = g_stream_ctl<T>.wait;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int cont = uni(cont_);
    const int tid = threadIdx.x;
    const int np = a.np, nM = a.nM;
    const long per_sweep = (long)nM * a.nTj * a.nTk;
    const SkPre pre = sk_prefetch<TA, WY, WZ, BY>(a, tid); // in flight while the block waits for its tile
    const int* const set_word = a.ctl + 4; // INT_MAX (host): stands in for an upstream tile that does not exist
    auto m_lo = [&](int B_, int C_) { return (T::NYT * B_ + T::NZT * C_) / TA; };
    auto m_hi = [&](int B_, int C_) { return (T::NYT * B_ + T::NYT - 1 + T::NZT * C_ + T::NZT - 1 + a.nx - 2) / TA; };
    auto tile_word = [&](int s, int m, int B, int C) { return a.tile_done + s * per_sweep + m + (long)nM * (B + (long)a.nTj * C); };
    auto give_up = [&]() { __hip_atomic_fetch_max(a.ticket, (int)a.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // the tile's LDS image holds interior cells only (full tile, three rows and 22 entries of interior around every bundle row)?
    auto interior_image = [&](int m, int B, int C, int sj, int sk) {
        const int tj = sj > 0 ? B : a.nTj - 1 - B, tk = sk > 0 ? C : a.nTk - 1 - C;
        const int j_lo = 1 + tj * T::NYT, k_lo = 1 + tk * T::NZT, X0 = TA * m - T::NYT * B - T::NZT * C;
        return j_lo >= 4 && j_lo + T::NYT + 2 <= a.ny - 1 && k_lo >= 4 && k_lo + T::NZT + 2 <= a.nz - 1 && X0 - (T::NYT + T::NZT + 4) >= 0 &&
               X0 + 22 <= a.nx - 2;
    };
    const uint32_t packed = (uint32_t)uni(sh_task[0]);
    const int sP = uni(sh_task[1]);
    const int s = sP & (DF_BATCH - 1), P = (int)((unsigned)sP >> DF_SWEEP_BITS);
    if (LSF_STREAM_PRIO) __builtin_amdgcn_s_setprio(0);
    if (tid == 0) {
        const int m = packed & 0x3ff, B = (packed >> 10) & 0x3ff, C = (packed >> 20) & 0x3ff;
        const int4 swp = *(const int4*)(a.sweep_tab + 4 * s); // signs and spacing of the sweep: one request
        int go = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const int* w0 = set_word; // continued: the previous tile of the column is this block's
        const int* w1 = (B >= 1 && m >= m_lo(B - 1, C) && m <= m_hi(B - 1, C)) ? tile_word(s, m, B - 1, C) : set_word;
        const int* w2 = (C >= 1 && m >= m_lo(B, C - 1) && m <= m_hi(B, C - 1)) ? tile_word(s, m, B, C - 1) : set_word;
        if (!cont) {
            w0 = m - 1 >= m_lo(B, C) ? tile_word(s, m - 1, B, C) : set_word;
            const int need1 = s == 0 ? 0 : min(P + swp.w, np);
            const int need3 = s < a.nbuf ? 0 : np + 1;
            const int* always = a.ctl + 1; // the stop flag's neighbour ctl[1] >= 0: a condition that does not exist
            const int* p3 = s == 0 ? always : a.planes_done + s - 1;
            const int* p4 = s < a.nbuf ? always : a.planes_done + s - a.nbuf;
            for (;;) { // stage 1: conditions (b) and (c) -- what this tile reads of the previous sweep is final
                const int vstop = ld_flag(a.ctl + 0);
                const int v3 = ld_flag(p3), v4 = ld_flag(p4);
                if (vstop != 0) {
                    go = 2;
                    give_up();
                    break;
                }
                if ((v3 >= need1) & (v4 >= need3)) {
                    // the verdict travels in the word that releases the waiter (np + 2: that sweep, or one before it, stopped)
                    if ((s >= a.nbuf && v4 == np + 2) || (s >= 1 && v3 == np + 2)) {
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
                __builtin_amdgcn_s_sleep(LSF_POLL_SLEEP);
            }
        }
        sh_task[2] = go, sh_task[3] = swp.x, sh_task[4] = swp.y, sh_task[5] = swp.z, sh_task[8] = swp.w;
        // (parked as offsets from tile_done, not as pointers: a pointer that has been through LDS is a flat pointer to the compiler)
        sh_wait[0] = (unsigned long long)(w0 - a.tile_done), sh_wait[1] = (unsigned long long)(w1 - a.tile_done);
        sh_wait[2] = (unsigned long long)(w2 - a.tile_done), sh_wait[3] = t0;
        if (a.dbg) sh_wait[5] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    int go = uni(sh_task[2]);
    auto wait_upstream = [&]() -> bool { // stage 2: condition (a), the upstream tiles DONE
        if (tid == 0) {
            int go2 = 1;
            const int *w0 = a.tile_done + (long)sh_wait[0], *w1 = a.tile_done + (long)sh_wait[1], *w2 = a.tile_done + (long)sh_wait[2];
            const unsigned long long t0 = sh_wait[3];
            for (;;) {
                const int vstop = ld_flag(a.ctl + 0);
                const int v0 = ld_flag(w0), v1 = ld_flag(w1), v2 = ld_flag(w2);
                if (vstop != 0) {
                    go2 = 2;
                    give_up();
                    break;
                }
                if ((v0 >= 2) & (v1 >= 2) & (v2 >= 2)) break;
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
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return uni(sh_task[6]) == 1;
    };
    if (go == 1) {
        if (!skew_tile<TA, WY, WZ, BY, STRICT, true, false, true>(sm, a, packed, a.g0 + s, uni(sh_task[3]), uni(sh_task[4]), uni(sh_task[5]), pre,
                                                                  wait_upstream, cont))
            go = 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads(); // every wave of the tile has drained its stores (and is done with the LDS image)
    if (tid == 0) {
        int next = 0;
        if (go == 1) {
            const int m = packed & 0x3ff, B = (packed >> 10) & 0x3ff, C = (packed >> 20) & 0x3ff;
            st_flag(tile_word(s, m, B, C), 2);
            const int done = __hip_atomic_fetch_add(a.plane_cnt + s * np + P, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (done == a.plane_size[P]) {
                for (;;) { // push the count of leading complete hyperplanes as far as it goes (see k_reinit_gs_persist)
                    int lead = ld_flag(a.planes_done + s);
                    if (lead >= np || ld_flag(a.plane_cnt + s * np + lead) < a.plane_size[lead]) break;
                    int nxt = lead + 1;
                    if (nxt == np) nxt = np + 1 + (ld_flag(a.ctl + 0) != 0 ? 1 : 0);
                    __hip_atomic_compare_exchange_strong(a.planes_done + s, &lead, nxt, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // ---- CONTINUE with (m + 1, B, C)? ----
            int why = 9; // (LSF_TRACE_TILES) 9 end of column, 10 previous sweep not past, 11 cross tiles not ready, 12 claim lost
            if (m + 1 <= m_hi(B, C) && a.cont_on) {
                const int* c1 = (B >= 1 && m + 1 >= m_lo(B - 1, C) && m + 1 <= m_hi(B - 1, C)) ? tile_word(s, m + 1, B - 1, C) : set_word;
                const int* c2 = (C >= 1 && m + 1 >= m_lo(B, C - 1) && m + 1 <= m_hi(B, C - 1)) ? tile_word(s, m + 1, B, C - 1) : set_word;
                const int* always = a.ctl + 4;
                const int vstop = ld_flag(a.ctl + 0);
                const int v3 = ld_flag(s == 0 ? always : a.planes_done + s - 1), v1 = ld_flag(c1), v2 = ld_flag(c2);
                const int need1 = s == 0 ? 0 : min(P + 1 + sh_task[8], np);
                bool past = s == 0 || (v3 >= need1 && v3 <= np + 1);
                if (!past && v3 <= np + 1 && LSF_STREAM_FINE && interior_image(m + 1, B, C, sh_task[4], sh_task[5]))
                    past = stream_prev_sweep_past<T, TA>(a, s, m + 1, B, C, per_sweep);
                why = !past ? 10 : 11;
                if (vstop == 0 && past && v1 >= a.cont_on && v2 >= a.cont_on) {
                    int expect = 0;
                    why = 12;
                    if (__hip_atomic_compare_exchange_strong(tile_word(s, m + 1, B, C), &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        next = 1;
                        why = 8;
                        sh_task[0] = (int)(packed + 1u); // m + 1 (m < 1023: get_skew_tiles)
                        sh_task[1] = s | ((P + 1) << DF_SWEEP_BITS);
                    }
                }
            }
            if (a.dbg) {
                const unsigned long long tsC = __builtin_amdgcn_s_memrealtime();
                atomicAdd(a.dbg + 0, sh_wait[5] - sh_wait[4]);
                atomicAdd(a.dbg + 1, tsC - sh_wait[5]);
                atomicAdd(a.dbg + 2, 1ull);
                atomicAdd(a.dbg + why, 1ull);
                sh_wait[4] = tsC;
            }
        }
        sh_task[7] = next;
    }
    __syncthreads();
    if (go != 1) return 0;
    return 1 + uni(sh_task[7]);
}

template <int TA, int WY, int WZ, int BY, bool STRICT>
__global__ __launch_bounds__(64 * WY * WZ) __attribute__((amdgpu_waves_per_eu(BY == 16 ? 2 : (WY == 2 && WZ == 2 ? LSF_STREAM_WAVES : 1)))) void k_reinit_gs_stream(GsArgs a)
{
    using T = SkTile<TA, WY, WZ, BY>;
    int* const sh_task = g_stream_ctl<T>.task;
    const int tid = threadIdx.x, lane = tid & 63;
    const long per_sweep = (long)a.nM * a.nTj * a.nTk;
    auto tile_word = [&](int s, int m, int B, int C) { return a.tile_done + s * per_sweep + m + (long)a.nM * (B + (long)a.nTj * C); };
    for (;;) {
        if (a.dbg && tid == 0) g_stream_ctl<T>.wait[4] = __builtin_amdgcn_s_memrealtime(); // LSF_TRACE_TILES: take + wait | work + publish
        // ---- ACQUIRE: the first free tile of the list ------------------------------------------------------------------
        if (tid < 64) {
            // a ticket is a position of the list that nobody else will get; tiles that a continuing block has claimed already
            // are skipped a run at a time: the 64 lanes look at the 64 entries from the ticket on, and the counter jumps over the
            // claimed entries in front of the first free one (claims are permanent: no free entry is ever skipped)
            uint2 e = make_uint2(0u, 0u);
            int go = 0;
            for (;;) {
                long t = 0;
                if (lane == 0) t = (long)__hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t = (long)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= a.total) break;
                bool took = false, mine = true; // position t is this block's ticket (the first window only)
                for (;;) { // windows of 64 entries from position t on
                    const long idx = t + lane;
                    const bool valid = idx < a.total;
                    const uint2 ei = valid ? a.order[idx] : make_uint2(0u, 0u);
                    const int s_ = (int)(ei.y & (unsigned)(DF_BATCH - 1));
                    int* wd = tile_word(s_, ei.x & 0x3ff, (ei.x >> 10) & 0x3ff, (ei.x >> 20) & 0x3ff);
                    int st = valid ? ld_flag(wd) : 0; // (an entry beyond the list counts as free: the run ends there)
                    if (mine && lane == 0 && st == 0) { // the entry of the ticket itself: this block's, unless a continuing block is faster
                        int expect = 0;
                        st = __hip_atomic_compare_exchange_strong(wd, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? -1 : 1;
                    }
                    if (__builtin_amdgcn_readfirstlane(st) == -1) {
                        e.x = (unsigned)__builtin_amdgcn_readfirstlane((int)ei.x), e.y = (unsigned)__builtin_amdgcn_readfirstlane((int)ei.y);
                        took = true;
                        break;
                    }
                    const unsigned long long free_ = __builtin_amdgcn_ballot_w64(st == 0);
                    const int run = free_ ? __builtin_ctzll(free_) : 64; // claimed entries from t on
                    if (run < 64 || t + 64 >= a.total) {
                        if (lane == 0) __hip_atomic_fetch_max(a.ticket, (int)min(t + run, a.total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break; // the next ticket is the first free entry, or the one after it if somebody else gets that
                    }
                    t += 64;
                    mine = false;
                }
                if (took) {
                    go = 1;
                    break;
                }
                if (ld_flag(a.ctl + 0) != 0) break;
            }
            if (lane == 0) sh_task[0] = (int)e.x, sh_task[1] = (int)e.y, sh_task[2] = go;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(sh_task[2]) == 0) return; // the list is exhausted (or cut short: stop, time-out)
        // ---- the acquired tile, then the tiles of its column as long as they can be continued ---------------------------
        const unsigned long long kaddr = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        int r = stream_tile<TA, WY, WZ, BY, STRICT>(0, (unsigned)kaddr, (unsigned)(kaddr >> 32));
        while (r == 2) r = stream_tile<TA, WY, WZ, BY, STRICT>(1, (unsigned)kaddr, (unsigned)(kaddr >> 32));
        if (r == 0) return;
    }
}

} // namespace lsf
