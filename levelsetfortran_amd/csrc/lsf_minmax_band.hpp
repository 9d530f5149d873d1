// lsf_minmax_band.hpp -- the min/max flow (set3d.f90:394-462) on the narrow band only.
//
// The reference visits every grid point in every iteration and updates the points with phiNB == 1 (set3d.f90:399, :420);
// phiNB is refreshed from phi after every iteration (subs.f90:178-207: |phi| < 4.1 dx).  A point outside the band is never
// written, so its phi -- hence its band membership -- never changes: every later band is a subset of
//     LIST = { interior points with mask == 1 on entry, or |phi| < 4.1 dx on entry }
// (the caller's mask decides the first iteration, narrowBand of the evolving field the later ones; a point in neither set keeps
// |phi| >= 4.1 dx for the whole call).  The dense executor (k_minmax_fp, lsf_kernels.hpp) copies and re-reads all N^3 points
// per iteration -- 2.1 GB at 512^3 to update a band of ~1 % of them.  Here the list is built once per call (one pass over phi
// and the mask), phi stays IN PLACE, and every pass of an iteration -- Jacobi start, fix passes to the in-place (Gauss-Seidel)
// fixed point, RMS partials -- runs over the list: the cost follows the band, not the grid.
//
// Same arithmetic and the same fixed-point argument as k_minmax_fp (see there): the curvature and the centre value are frozen
// (compact copy `Aold` of the list cells' values at iteration start; a neighbour outside the list never changes, its frozen value
// is the field's), the switch pAve < 0 (subs.f90:473-481) reads the in-place values of the three upstream neighbours from the
// field itself.  The RMS (set3d.f90:437-447) is a fixed-order sum over list chunks: cells outside the list contribute exactly 0.
// The field, the masks, the iteration count and the stop iteration equal the dense executor's and the oracle's; the RMS trace
// agrees to rounding (another summation order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
            in = interior && (mask[p] == 1 || __builtin_fabs(phi[p]) < tn);
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

// List build, pass 2: exclusive prefix sum of the per-block counts (one block; offsets[nblk] = length of the list).
static __global__ __launch_bounds__(1024) void k_mb_offsets(const int* __restrict__ counts, long nblk, int* __restrict__ offsets)
{
    __shared__ int part[1024];
    const long per = (nblk + 1023) / 1024, lo = (long)threadIdx.x * per, hi = min(lo + per, nblk);
    int s = 0;
    for (long q = lo; q < hi; ++q) s += counts[q];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive scan
        const int v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (long q = lo; q < hi; ++q) {
        offsets[q] = run;
        run += counts[q];
    }
    if (threadIdx.x == 1023) offsets[nblk] = part[1023];
}

// List build, pass 3: the blocks' segments back to back; the cells' values on entry (frozen copy of iteration 1, and the copy an
// uncertified attempt is restored from).
static __global__ __launch_bounds__(256) void k_mb_gather(const int* __restrict__ staging, const int* __restrict__ counts,
                                                   const int* __restrict__ offsets, const double* __restrict__ phi, int* __restrict__ L,
                                                   double* __restrict__ aold, double* __restrict__ a0)
{
    const int cnt = counts[blockIdx.x], o = offsets[blockIdx.x];
    for (int q = threadIdx.x; q < cnt; q += 256) {
        const int p = staging[(long)blockIdx.x * MB_SCAN + q];
        L[o + q] = p;
        const double v = phi[p];
        aold[o + q] = v, a0[o + q] = v;
    }
}

// List build, pass 4: list index of the six neighbours of every list cell (order: i-1, i+1, j-1, j+1, k-1, k+1), -1 = not in the list.
// The list is sorted by p: the x neighbours are the adjacent entries, the others a binary search on their side of the entry.
static __global__ __launch_bounds__(256) void k_mb_links(const int* __restrict__ L, int nL, int sx, int sxy, int* __restrict__ nb6)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nL) return;
    const int p = L[e];
    auto find = [&](int lo, int hi, int q) { // first entry >= q in [lo, hi)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (L[mid] < q) lo = mid + 1;
            else hi = mid;
        }
        return (lo < nL && L[lo] == q) ? lo : -1;
    };
    int* o = nb6 + 6 * (long)e;
    o[0] = (e > 0 && L[e - 1] == p - 1) ? e - 1 : -1;
    o[1] = (e + 1 < nL && L[e + 1] == p + 1) ? e + 1 : -1;
    o[2] = find(0, e, p - sx);
    o[3] = find(e + 1, nL, p + sx);
    o[4] = find(0, e, p - sxy);
    o[5] = find(e + 1, nL, p + sxy);
}

// One iteration over the list.  F = phi, in place; aold = frozen values of the list cells (iteration start), anext = the same for
// the next iteration (written by pass 2; two compact buffers in rotation so that the frozen copy of the iteration that meets the
// stop test survives: the host's masks are those of the field BEFORE it, set3d.f90:448-460).
// PASS 0: Jacobi start, band test, per-chunk band flags (one chunk per block).
// PASS 1: fix pass `epoch` (see k_minmax_fp): re-evaluate band cells with the in-place upstream values until a whole pass changes
//         nothing; the first pass of an iteration visits every band chunk, later ones the chunks stamped by a change upstream.
// PASS 2: RMS partials per chunk, next frozen copy (one chunk per block).
template <int PASS>
__global__ __launch_bounds__(256) void k_minmax_band(double* __restrict__ F, const double* __restrict__ aold, double* __restrict__ anext,
                                                     const int32_t* __restrict__ nbmask, const int* __restrict__ L,
                                                     const int* __restrict__ nb6, unsigned char* __restrict__ isband, int nL, int sx,
                                                     long sxy, double dx, double h1, int* __restrict__ chunkflag,
                                                     int* __restrict__ stamp, int nchunks, int epoch, int first,
                                                     const int* __restrict__ changed_prev, int* __restrict__ changed_cur,
                                                     double* __restrict__ partials, int* __restrict__ ctl)
{
    __shared__ double red[4];
    __shared__ int flag;
    __shared__ unsigned long long todo;
    if (ctl[0]) return;
    if (PASS == 1 && changed_prev && *changed_prev == 0) return;
    const double dxx = 1. / (dx * dx);
    // frozen value of neighbour d of entry e (p + off): the compact copy where the neighbour is a list cell, else the field's
    auto frozen = [&](const int* o, int d, long q) {
        const int idx = o[d];
        return idx >= 0 ? aold[idx] : F[q];
    };
    if (PASS != 1) {
        const int chunk = blockIdx.x, e = chunk * MB_CH + threadIdx.x;
        double acc = 0.0;
        bool band = false;
        if (e < nL) {
            const long p = L[e];
            const double c = aold[e];
            if (PASS == 0) {
                band = nbmask ? nbmask[p] == 1 : __builtin_fabs(c) < 4.1 * dx;
                isband[e] = band ? 1 : 0;
                if (band) {
                    const int* o = nb6 + 6 * (long)e;
                    const double xm = frozen(o, 0, p - 1), xp = frozen(o, 1, p + 1), ym = frozen(o, 2, p - sx), yp = frozen(o, 3, p + sx),
                                 zm = frozen(o, 4, p - sxy), zp = frozen(o, 5, p + sxy);
                    F[p] = minmax_update(c, xm, xp, yp, ym, zp, zm, minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx), h1);
                }
            } else {
                const double v = F[p];
                if (isband[e]) {
                    const double d = v - c;
                    acc = d * d;
                }
                anext[e] = v;
            }
        }
        if (PASS == 0) {
            const int any = __syncthreads_or(band ? 1 : 0);
            if (threadIdx.x == 0) chunkflag[chunk] = any;
        } else {
            acc = wave_sum(acc);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
            __syncthreads();
            if (threadIdx.x == 0) partials[chunk] = red[0] + red[1] + red[2] + red[3];
            if (blockIdx.x == 0 && threadIdx.x == 0 && changed_prev) {
                // `first` = number of fix passes that were enqueued, changed_prev = the counter of the last of them
                const int* c0 = changed_prev - (first - 1);
                int used = 0;
                for (int f = 0; f < first; ++f) used += c0[f] != 0;
                atomicMax(ctl + 4, used);
                if (*changed_prev != 0) ctl[3] = 1; // the last allowed fix pass still changed something: not certified
            }
        }
        return;
    }
    const long round = 64L * gridDim.x;
    for (long base = 0; base < nchunks; base += round) {
        if (threadIdx.x < 64) {
            const long ch = base + blockIdx.x + (long)threadIdx.x * gridDim.x;
            bool need = ch < nchunks && chunkflag[ch] != 0;
            if (!first) need = need && stamp[ch] >= epoch;
            const unsigned long long b = __ballot(need);
            if (threadIdx.x == 0) todo = b;
        }
        __syncthreads();
        unsigned long long m = todo;
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const long chunk = base + blockIdx.x + (long)l * gridDim.x;
            if (threadIdx.x == 0) flag = 0;
            __syncthreads();
            const int e = (int)(chunk * MB_CH) + threadIdx.x;
            int mine = 0;
            if (e < nL && isband[e]) {
                const long p = L[e];
                const double c = aold[e];
                const int* o = nb6 + 6 * (long)e;
                const double xm = frozen(o, 0, p - 1), xp = frozen(o, 1, p + 1), ym = frozen(o, 2, p - sx), yp = frozen(o, 3, p + sx),
                             zm = frozen(o, 4, p - sxy), zp = frozen(o, 5, p + sxy);
                const double curv = minmax_curv(c, xp, xm, yp, ym, zp, zm, dxx);
                // upstream neighbours from the evolving field
                const double nv = minmax_update(c, F[p - 1], xp, yp, F[p - sx], zp, F[p - sxy], curv, h1);
                if (!(nv == F[p])) {
                    F[p] = nv;
                    mine = 1;
                    // the three cells that read this one (band cells are list cells)
                    if (o[1] >= 0) stamp[o[1] / MB_CH] = epoch + 1;
                    if (o[3] >= 0) stamp[o[3] / MB_CH] = epoch + 1;
                    if (o[5] >= 0) stamp[o[5] / MB_CH] = epoch + 1;
                }
            }
            if (mine) flag = 1;
            __syncthreads();
            if (threadIdx.x == 0 && flag) atomicAdd(changed_cur, 1); // number of chunks this pass still changed
            __syncthreads();
        }
        __syncthreads();
    }
}

// masks of the field BEFORE the iteration that met the stop test (set3d.f90:448-460: EXIT comes before narrowBand): the list cells
// of that field are the frozen copy of the last iteration, everything else is the final field
static __global__ __launch_bounds__(256) void k_mb_patch_masks(const int* __restrict__ L, const double* __restrict__ aold, int nL, double dx,
                                                        int32_t* __restrict__ nb, int32_t* __restrict__ sb)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nL) return;
    const double a = __builtin_fabs(aold[e]);
    const int p = L[e];
    nb[p] = a < 4.1 * dx ? 1 : 0;
    sb[p] = a < 8.1 * dx ? 1 : 0;
}

// the list cells' values back into the field (an uncertified attempt starts over)
static __global__ __launch_bounds__(256) void k_mb_restore(const int* __restrict__ L, const double* __restrict__ a0, int nL, double* __restrict__ F)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nL) F[L[e]] = a0[e];
}

} // namespace lsf
