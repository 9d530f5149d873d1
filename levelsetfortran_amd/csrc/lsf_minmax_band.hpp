// lsf_minmax_band.hpp -- the min/max flow (set3d.f90:394-462) on the narrow band only.
//
// The reference visits every grid point in every iteration and updates the points with phiNB == 1 (set3d.f90:399, :420);
// phiNB is refreshed from phi after every iteration (subs.f90:178-207: |phi| < 4.1 dx).  A point outside the band is never
// written, so its phi -- hence its band membership -- never changes: every later band is a subset of
//     LIST = { interior points with mask == 1 on entry, or |phi| < 4.1 dx on entry }
// (the caller's mask decides the first iteration, narrowBand of the evolving field the later ones; a point in neither set keeps
// |phi| >= 4.1 dx for the whole call).  The dense executor (k_minmax_fp, lsf_kernels.hpp) copies and re-reads all N^3 points
// per iteration -- 2.1 GB at 512^3 to update a band of ~1 % of them.  Here the list is built once per call (one pass over phi
// and the mask) and the whole flow runs on COMPACT arrays indexed by list position: two copies of the list cells' values in
// rotation (frozen at iteration start / evolving), the list indices of the six neighbours, and what a fix pass needs of the frozen
// field (curvature and the three downstream values).  The field itself is read for neighbours outside the list only (they never
// change) and written once, at the end of the call.  Every pass of an iteration -- Jacobi start, fix passes to the in-place
// (Gauss-Seidel) fixed point, RMS partials -- costs what the band costs, not what the grid costs.
//
// Same arithmetic and the same fixed-point argument as k_minmax_fp (see there): curvature and centre value are frozen, the switch
// pAve < 0 (subs.f90:473-481) reads the evolving values of the three upstream neighbours.  The RMS (set3d.f90:437-447) is a
// fixed-order sum over list chunks: cells outside the list contribute exactly 0.  Field, masks, iteration count and stop
// iteration equal the dense executor's and the oracle's; the RMS trace agrees to rounding (another summation order).
//
// Fix passes: the first of an iteration is a wide launch over every band chunk; all further ones run inside ONE small resident
// launch that loops with a grid barrier until a pass changes nothing (k_minmax_band_tail): the number of passes never has to be
// guessed, and a pass costs a barrier instead of a launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/rocprim.hpp>

namespace lsf {

constexpr int MB_SCAN = 2048; // points per block of the list build
constexpr int MB_CH = 256;    // list entries per chunk (= threads per block of the iteration kernels)

// List build, pass 1: block b scans points [2048 b, 2048 b + 2048) and leaves its list cells, in increasing p, in
// staging[2048 b ...] and their number in counts[b].
static __global__ __launch_bounds__(256) void k_mb_collect(const double* __restrict__ phi, const int32_t* __restrict__ mask, int nx, int ny,
                                                    int nz, double dx, int* __restrict__ staging, int* __restrict__ counts)
{
    __shared__ int wcnt[MB_SCAN / 256][4];
    const long sxy = (long)(nx + 1) * (ny + 1), n = sxy * (nz + 1);
    const long base = (long)blockIdx.x * MB_SCAN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double tn = 4.1 * dx;
    const int sx = nx + 1;
    const bool tiny = sxy <= 256;
    const int d256j = 256 / sx, d256i = 256 - d256j * sx;
    int i, j, k;
    point_ijk(base + threadIdx.x, nx + 1, ny + 1, n, i, j, k);
    // all loads of a thread's eight points before the first test (phi and the mask: 12 bytes per point, the whole cost of the pass)
    double v[MB_SCAN / 256];
    int32_t mk[MB_SCAN / 256];
#pragma unroll
    for (int t = 0; t < MB_SCAN / 256; ++t) {
        const long p = min(base + t * 256 + threadIdx.x, n - 1);
        v[t] = phi[p], mk[t] = mask[p];
    }
    unsigned mine = 0u;
    int before[MB_SCAN / 256];
#pragma unroll
    for (int t = 0; t < MB_SCAN / 256; ++t) {
        const long p = base + t * 256 + threadIdx.x;
        bool in = false;
        if (p < n) {
            if (t > 0) {
                if (tiny) point_ijk(p, nx + 1, ny + 1, n, i, j, k);
                else step_ijk(i, j, k, d256i, d256j, 0, nx + 1, ny + 1);
            }
            const bool interior = i >= 1 && i <= nx - 1 && j >= 1 && j <= ny - 1 && k >= 1 && k <= nz - 1;
            in = interior && (mk[t] == 1 || __builtin_fabs(v[t]) < tn);
        }
        const unsigned long long b = __ballot(in);
        before[t] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[t][wave] = __popcll(b);
        if (in) mine |= 1u << t;
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int t = 0; t < MB_SCAN / 256; ++t) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w == wave && ((mine >> t) & 1u)) staging[base + run + before[t]] = (int)(base + t * 256 + threadIdx.x);
            run += wcnt[t][w];
        }
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = run;
}

// List build, pass 2: exclusive prefix sum of the per-block counts (one block; offsets[nblk] = length of the list).  A thread owns a
// run of `per` consecutive counts (a multiple of four) and moves them as 16-byte vectors, all loads of a run in flight together
// (one by one, 64 dependent round trips per thread made this 100 us at 512^3).
static __global__ __launch_bounds__(1024) void k_mb_offsets(const int* __restrict__ counts, long nblk, int* __restrict__ offsets)
{
    __shared__ int part[1024];
    const long per = (((nblk + 1023) / 1024) + 3) & ~3L, lo = (long)threadIdx.x * per, hi = min(lo + per, nblk);
    const bool vec = (((uintptr_t)counts | (uintptr_t)offsets) & 15) == 0;
    int s = 0;
    if (vec) {
        for (long q = lo; q < hi; q += 16) {
            int4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long qq = q + 4 * u;
                v[u] = qq + 3 < hi ? *(const int4*)(counts + qq) : make_int4(qq < hi ? counts[qq] : 0, qq + 1 < hi ? counts[qq + 1] : 0, qq + 2 < hi ? counts[qq + 2] : 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    } else {
        for (long q = lo; q < hi; ++q) s += counts[q];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive scan
        const int v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    if (vec) {
        for (long q = lo; q < hi; q += 4) {
            if (q + 3 < hi) {
                const int4 c = *(const int4*)(counts + q);
                *(int4*)(offsets + q) = make_int4(run, run + c.x, run + c.x + c.y, run + c.x + c.y + c.z);
                run += c.x + c.y + c.z + c.w;
            } else {
                for (long r = q; r < hi; ++r) {
                    offsets[r] = run;
                    run += counts[r];
                }
            }
        }
    } else {
        for (long q = lo; q < hi; ++q) {
            offsets[q] = run;
            run += counts[q];
        }
    }
    if (threadIdx.x == 1023) offsets[nblk] = part[1023];
}

// List build, pass 3: the blocks' segments back to back (the list in increasing p), and every cell's BRICK KEY: the list is then
// sorted by it (rocprim::radix_sort_pairs, once per call), so that a chunk of 256 consecutive list cells is a compact piece of
// space -- cells of a few neighbouring bricks of 8 x 8 x 4 points -- instead of a run along x: the chains of dependent sign flips
// that a fix pass follows (they run along the zero level set, in +x, +y and +z) then stay inside a chunk for several links and are
// resolved by ONE visit (mb_visit_chunk repeats while it changes cells).  Measured at 512^3, two spheres, iteration 50 of a call:
// 16 passes with the list in memory order, see DESIGN.md.
// One wavefront per block of the scan (most blocks hold no list cell).
__device__ __forceinline__ unsigned mb_brick_key(int p, int sx, int sy, int nbx, int nby)
{
    const unsigned pu = (unsigned)p, q = pu / (unsigned)sx;
    const unsigned i = pu - q * (unsigned)sx, k = q / (unsigned)sy, j = q - k * (unsigned)sy;
    return (((i >> 3) + (unsigned)nbx * ((j >> 3) + (unsigned)nby * (k >> 2))) << 8) | ((i & 7u) + 8u * ((j & 7u) + 8u * (k & 3u)));
}
static __global__ __launch_bounds__(256) void k_mb_gather(const int* __restrict__ staging, const int* __restrict__ counts,
                                                   const int* __restrict__ offsets, long nblk, int sx, int sy, int nbx, int nby,
                                                   int* __restrict__ L, unsigned* __restrict__ key)
{
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const int cnt = counts[b], o = offsets[b];
    for (int q = threadIdx.x & 63; q < cnt; q += 64) {
        const int p = staging[b * MB_SCAN + q];
        L[o + q] = p;
        key[o + q] = mb_brick_key(p, sx, sy, nbx, nby);
    }
}

// List build, pass 4 (on the list sorted by brick key): list index of the six neighbours of every list cell, nb[d][e] with d: 0 i-1,
// 1 i+1, 2 j-1, 3 j+1, 4 k-1, 5 k+1; -1 = not in the list (binary search for the neighbour's key); and the cells' values on entry
// (the frozen copy of iteration 1).
static __global__ __launch_bounds__(256) void k_mb_links(const int* __restrict__ L, const unsigned* __restrict__ key, int nL, int sx, int sy,
                                                  int nbx, int nby, const double* __restrict__ phi, int* __restrict__ nb,
                                                  double* __restrict__ v0)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nL) return;
    const int p = L[e];
    v0[e] = phi[p];
    auto find = [&](int q) {
        const unsigned kq = mb_brick_key(q, sx, sy, nbx, nby);
        int lo = 0, hi = nL;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key[mid] < kq) lo = mid + 1;
            else hi = mid;
        }
        return (lo < nL && key[lo] == kq) ? lo : -1;
    };
    const int sxy = sx * sy;
    nb[e] = find(p - 1);
    nb[(long)nL + e] = find(p + 1);
    nb[2L * nL + e] = find(p - sx);
    nb[3L * nL + e] = find(p + sx);
    nb[4L * nL + e] = find(p - sxy);
    nb[5L * nL + e] = find(p + sxy);
}

// What the iteration kernels share.  A = frozen values of the list cells (iteration start), X = evolving values (X becomes the next
// iteration's A: two compact arrays in rotation, so the frozen copy of the iteration that meets the stop test survives -- the host's
// masks are those of the field BEFORE it, set3d.f90:448-460); F = the field, read for neighbours outside the list only.
struct MbArgs {
    const double* F;
    const double* A;
    double* X;
    const int32_t* nbmask; // the caller's mask (first iteration) or NULL (|A| < 4.1 dx)
    const int* L;
    const int* nb;         // [6][nL]
    unsigned char* isband;
    double* curv;          // [nL] frozen curvature of the band cells
    double* down;          // [3][nL] frozen i+1, j+1, k+1 values of the band cells
    int nL, sx;
    long sxy;
    double dx, h1;
    int* chunkflag;        // [nchunks] the chunk holds band cells
    int* stamp;            // [nchunks]
    int nchunks;
    double* partials;      // [nchunks]
    int* ctl;              // [0] stop, [1] iterations done, [2] NaN, [3] an iteration was not certified, [4] most fix passes that changed cells
    int* chg;              // [64] chunks changed per fix pass of the iteration | [64] barrier word of the looping launch
};
constexpr int MB_EPOCHS = 64; // fix passes an iteration can take (epochs of iteration it: it * MB_EPOCHS + 1 + pass)

// PASS 0: Jacobi start, band test, per-chunk band flags, the frozen data of the fix passes (one chunk per block).
// PASS 2: RMS partials per chunk (one chunk per block).
template <int PASS>
__global__ __launch_bounds__(256) void k_minmax_band(MbArgs a)
{
    __shared__ double red[4];
    if (a.ctl[0] | a.ctl[3]) return;
    // (pass 0 also zeroes the iteration's change counts and barrier word: no memset command between the launches.  The block that
    // finishes pass 2 last could add the partials and apply the stop test as well -- built and measured: 5 800 blocks counting
    // themselves on ONE word cost 60 us per iteration on this chip, three times what the separate launch of k_mb_finish costs)
    if (PASS == 0 && blockIdx.x == 0 && threadIdx.x <= MB_EPOCHS) a.chg[threadIdx.x] = 0;
    const int chunk = blockIdx.x, e = chunk * MB_CH + threadIdx.x;
    const int nL = a.nL;
    double acc = 0.0;
    bool band = false;
    if (e < nL) {
        const double c = a.A[e];
        if (PASS == 0) {
            const long p = a.L[e];
            band = a.nbmask ? a.nbmask[p] == 1 : __builtin_fabs(c) < 4.1 * a.dx;
            a.isband[e] = band ? 1 : 0;
            double out = c;
            if (band) {
                const double dxx = 1. / (a.dx * a.dx);
                int id[6];
#pragma unroll
                for (int d = 0; d < 6; ++d) id[d] = a.nb[(long)d * nL + e];
                // frozen value of a neighbour: the compact copy where it is a list cell, else the field's (it never changes)
                const double xm = id[0] >= 0 ? a.A[id[0]] : a.F[p - 1], xp = id[1] >= 0 ? a.A[id[1]] : a.F[p + 1];
                const double ym = id[2] >= 0 ? a.A[id[2]] : a.F[p - a.sx], yp = id[3] >= 0 ? a.A[id[3]] : a.F[p + a.sx];
                const double zm = id[4] >= 0 ? a.A[id[4]] : a.F[p - a.sxy], zp = id[5] >= 0 ? a.A[id[5]] : a.F[p + a.sxy];
                const double cv = minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx);
                out = minmax_update(c, xm, xp, yp, ym, zp, zm, cv, a.h1);
                a.curv[e] = cv;
                a.down[e] = xp, a.down[(long)nL + e] = yp, a.down[2L * nL + e] = zp;
            }
            a.X[e] = out;
        } else if (a.isband[e]) {
            const double d = a.X[e] - c;
            acc = d * d;
        }
    }
    if (PASS == 0) {
        const int any = __syncthreads_or(band ? 1 : 0);
        if (threadIdx.x == 0) a.chunkflag[chunk] = any;
    } else {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) a.partials[chunk] = red[0] + red[1] + red[2] + red[3];
    }
}

// One visit of a fix pass to one chunk (one block): re-evaluate its band cells with the evolving upstream values; a cell that
// changes stamps the chunks of the three cells that read it with epoch + 1.  The visit REPEATS while it changes cells (at most
// MB_REPS times): a chain of dependent sign flips that stays inside the chunk -- consecutive list cells are neighbours along x, the
// rows of a plane follow each other -- is then resolved by one visit instead of one pass per link.  Evolving values are read and
// written past the vector L1 (sc1: the block re-reads what its own threads and other blocks have just stored).  Returns (to every
// thread) whether the visit changed a cell.
#ifndef LSF_MB_REPS
#define LSF_MB_REPS 8
#endif
constexpr int MB_REPS = LSF_MB_REPS;
__device__ __forceinline__ bool mb_visit_chunk(const MbArgs& a, int chunk, int epoch)
{
    const int nL = a.nL;
    const int e = chunk * MB_CH + (int)threadIdx.x;
    const bool mine = e < nL && a.isband[e];
    int ixm = -1, iym = -1, izm = -1, ixp = -1, iyp = -1, izp = -1;
    double c = 0., cv = 0., xp = 0., yp = 0., zp = 0., fxm = 0., fym = 0., fzm = 0.;
    if (mine) {
        ixm = a.nb[e], iym = a.nb[2L * nL + e], izm = a.nb[4L * nL + e];
        ixp = a.nb[(long)nL + e], iyp = a.nb[3L * nL + e], izp = a.nb[5L * nL + e];
        c = a.A[e], cv = a.curv[e];
        xp = a.down[e], yp = a.down[(long)nL + e], zp = a.down[2L * nL + e];
        // an upstream neighbour outside the list never changes: the field's value
        const long p = a.L[e];
        if (ixm < 0) fxm = a.F[p - 1];
        if (iym < 0) fym = a.F[p - a.sx];
        if (izm < 0) fzm = a.F[p - a.sxy];
    }
    bool any = false;
    for (int rep = 0; rep < MB_REPS; ++rep) {
        int ch = 0;
        if (mine) {
            const double xm = ixm >= 0 ? ld_sc1(a.X + ixm) : fxm;
            const double ym = iym >= 0 ? ld_sc1(a.X + iym) : fym;
            const double zm = izm >= 0 ? ld_sc1(a.X + izm) : fzm;
            const double nv = minmax_update(c, xm, xp, yp, ym, zp, zm, cv, a.h1);
            if (!(nv == ld_sc1(a.X + e))) {
                st_sc1(a.X + e, nv);
                ch = 1;
                // the three cells that read this one; a reader in this chunk is re-evaluated by the next repetition (or, after the
                // last one, through the chunk's own stamp below)
                if (ixp >= 0 && ixp / MB_CH != chunk) st_flag(a.stamp + ixp / MB_CH, epoch + 1);
                if (iyp >= 0 && iyp / MB_CH != chunk) st_flag(a.stamp + iyp / MB_CH, epoch + 1);
                if (izp >= 0 && izp / MB_CH != chunk) st_flag(a.stamp + izp / MB_CH, epoch + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores are at the memory side before anybody is told
        const int changed = __syncthreads_or(ch);
        if (!changed) break;
        any = true;
        if (rep == MB_REPS - 1 && threadIdx.x == 0) st_flag(a.stamp + chunk, epoch + 1); // still changing: come back in the next pass
    }
    return any;
}

// First fix pass of an iteration: every band chunk, one chunk per block.
static __global__ __launch_bounds__(256) void k_minmax_band_fix(MbArgs a, int epoch, int* __restrict__ changed_cur)
{
    if (a.ctl[0] | a.ctl[3]) return;
    if (!a.chunkflag[blockIdx.x]) return;
    const bool any = mb_visit_chunk(a, (int)blockIdx.x, epoch);
    if (threadIdx.x == 0 && any) atomicAdd(changed_cur, 1); // number of chunks this pass changed
}

// Every further fix pass of an iteration in ONE launch: a grid small enough to be resident as a whole (128 blocks) loops over the
// passes with a barrier of its own between them -- a pass costs a barrier (a few microseconds) instead of a launch, and the loop ends
// with the pass that changes nothing: that pass IS the certificate of the fixed point, no count has to be guessed.  The chain of
// dependent sign flips a min/max iteration resolves grows with the flow (two-sphere field at 512^3: 2 passes in the first ten
// iterations, 17 by the fiftieth).  Stamps, change counts and evolving values travel past the non-coherent caches (sc1 stores
// drained before the barrier, sc1 loads behind it: cdna_hip_programming.md G16); every spin is bounded: a grid that is not
// resident as a whole (a device shared with other work) ends with ctl[3] and ctl[5] (= timed out, as opposed to "passes exhausted")
// and the host takes the dense executor -- for this call and, having said so once, for every later one on that device.  The host
// sizes the grid from the occupancy query: at most one block per CU, never more than the device admits at once.
// (blocks: one per CU measured best at 512^3 -- 64: 0.177, 128: 0.152, 256: 0.147, 512: 0.172 ms per iteration of a 50-iteration call)
constexpr int MB_TAIL_BLOCKS = 256;
static __global__ __launch_bounds__(256) void k_minmax_band_tail(MbArgs a, int epoch_first, int max_passes, const int* __restrict__ changed_first,
                                                          int* __restrict__ chg, int* __restrict__ bar, unsigned long long timeout_ticks)
{
    __shared__ unsigned long long todo;
    __shared__ int sh_go;
    if (a.ctl[0] | a.ctl[3]) return;
    if (*changed_first == 0) return; // (written by the launch before this one)
    const int nchunks = a.nchunks, G = gridDim.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int pass = 0, go = 1;
    for (; pass < max_passes; ++pass) {
        const int epoch = epoch_first + pass;
        int mine_changed = 0;
        for (long base = 0; base < nchunks; base += 64L * G) {
            if (threadIdx.x < 64) {
                const long ch = base + blockIdx.x + (long)threadIdx.x * G;
                const bool need = ch < nchunks && a.chunkflag[ch] != 0 && ld_flag(a.stamp + ch) >= epoch;
                const unsigned long long b = __ballot(need);
                if (threadIdx.x == 0) todo = b;
            }
            __syncthreads();
            unsigned long long m = todo;
            __syncthreads();
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                mine_changed += mb_visit_chunk(a, (int)(base + blockIdx.x + (long)l * G), epoch) ? 1 : 0;
            }
        }
        // barrier of the grid: every block's stores of this pass are drained (mb_visit_chunk), its count is in, then everybody looks
        if (threadIdx.x == 0) {
            if (mine_changed) __hip_atomic_fetch_add(chg + pass, mine_changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int target = (pass + 1) * G;
            go = 1;
            while (ld_flag(bar) < target) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                    go = -1;
                    st_flag(a.ctl + 5, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (go == 1 && ld_flag(chg + pass) == 0) go = 0; // the pass changed nothing: certified
            sh_go = go;
        }
        __syncthreads();
        go = sh_go;
        __syncthreads();
        if (go != 1) break;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicMax(a.ctl + 4, pass + 1);
        if (go != 0) a.ctl[3] = 1; // passes exhausted or a block timed out: not certified
    }
}

// k_finish for the band executor: an iteration that is not certified is not counted (the host resumes it)
static __global__ __launch_bounds__(RED_T) void k_mb_finish(const double* __restrict__ partials, long nPart, double den, double tol,
                                                     double* __restrict__ trace, int trace_cap, int* __restrict__ ctl)
{
    __shared__ double red[RED_T];
    if (ctl[0] | ctl[3]) return;
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

// end of the call: the list cells' values into the field
static __global__ __launch_bounds__(256) void k_mb_scatter(const int* __restrict__ L, const double* __restrict__ v, int nL, double* __restrict__ F)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nL) F[L[e]] = v[e];
}

// masks of the field BEFORE the iteration that met the stop test (set3d.f90:448-460: EXIT comes before narrowBand): the list cells
// of that field are the frozen copy of the last iteration, everything else is the final field
static __global__ __launch_bounds__(256) void k_mb_patch_masks(const int* __restrict__ L, const double* __restrict__ aold, int nL, double dx,
                                                        int32_t* __restrict__ nb, int32_t* __restrict__ sb)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nL) return;
    const double v = __builtin_fabs(aold[e]);
    const int p = L[e];
    nb[p] = v < 4.1 * dx ? 1 : 0;
    sb[p] = v < 8.1 * dx ? 1 : 0;
}

} // namespace lsf
