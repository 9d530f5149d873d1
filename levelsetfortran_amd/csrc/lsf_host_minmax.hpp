// lsf_host_minmax.hpp -- host side of narrowBand (subs.f90:178-207) and of the min/max flow (set3d.f90:394-462): the exact fixed-point
// executor, the tile-wavefront fallback and the Jacobi ordering.  Included by lsf_api.hip inside its anonymous namespace.
#pragma once

int narrowband_core(const double* d_phi, int32_t* d_nb, int32_t* d_sb, size_t n, double dx, hipStream_t st)
{
    const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_narrowband, dim3(grid), dim3(256), 0, st, d_phi, d_nb, d_sb, (long)n, dx);
    HIPCHK(hipGetLastError());
    return LSF_OK;
}

constexpr int MM_MAX_FIX = 32;   // most fix passes ever enqueued per min/max iteration
constexpr int MM_FIX_START = 16; // adaptive mode: passes enqueued per iteration until the first host check
// how the exact ordering of the min/max flow is produced
enum MinmaxExact { MM_TILES = 0, MM_FP_ADAPTIVE = 1, MM_FP_FULL = 2 };

int minmax_core_impl(double* d_phi, int32_t* d_nb, int32_t* d_sb, int nx, int ny, int nz, int iter, double dx,
                     double h1, double tol, int mode, int* iters_done, double* rms_trace, int trace_cap,
                     hipStream_t st, int exact_mode, bool* inexact)
{
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    if (iter < 0) return fail(LSF_ERR_INVALID, "iter must be >= 0");
    const int order = mode & LSF_ORDER_MASK;
    if (order != LSF_ORDER_GS && order != LSF_ORDER_JACOBI) return fail(LSF_ERR_INVALID, "unknown ordering");
    if (!d_phi || !d_nb || !d_sb) return fail(LSF_ERR_INVALID, "NULL field");
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if ((rc = ws(c.slot[S_PONG], n * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_CTL], 64))) return rc;
    if ((rc = ws(c.slot[S_TRACE], (size_t)std::max(iter, 1) * sizeof(double)))) return rc;
    int* ctl = (int*)c.slot[S_CTL].p;
    double* d_trace = (double*)c.slot[S_TRACE].p;
    HIPCHK(hipMemsetAsync(ctl, 0, 64, st));

    TileList* tl = nullptr;
    int nTi = 0, nTj = 0, nTk = 0, jblocks = 0;
    long n_part;
    const bool fixed_point = order == LSF_ORDER_GS && exact_mode != MM_TILES;
    const long fp_blocks = (long)((n + MM_CH - 1) / MM_CH);  // blocks of the scan
    const long fp_chunks = (long)((n + MM_SUB - 1) / MM_SUB); // chunks: flags, stamps, RMS partials
    int *bflag = nullptr, *chg = nullptr, *stamp = nullptr;
    constexpr size_t CHG_BYTES = (MM_MAX_FIX + 1) * sizeof(int);
    double* part2 = nullptr;
    if (fixed_point) {
        n_part = fp_chunks;
        if ((rc = ws(c.slot[S_BFLAG], (size_t)fp_chunks * sizeof(int)))) return rc;
        if ((rc = ws(c.slot[S_CHG], CHG_BYTES))) return rc;
        if ((rc = ws(c.slot[S_STAMP], (size_t)fp_chunks * sizeof(int)))) return rc;
        if ((rc = ws(c.slot[S_PART2], 256 * sizeof(double)))) return rc;
        bflag = (int*)c.slot[S_BFLAG].p;
        chg = (int*)c.slot[S_CHG].p;
        stamp = (int*)c.slot[S_STAMP].p;
        HIPCHK(hipMemsetAsync(stamp, 0, (size_t)fp_chunks * sizeof(int), st));
        part2 = (double*)c.slot[S_PART2].p;
    } else if (order == LSF_ORDER_GS) {
        nTi = cdiv(nx + 1, MM_TA), nTj = cdiv(ny + 1, 8), nTk = cdiv(nz + 1, 8);
        if ((rc = get_tiles(nTi, nTj, nTk, &tl))) return rc;
        n_part = (long)nTi * nTj * nTk;
    } else {
        jblocks = (int)std::min<size_t>((n + 255) / 256, 8192);
        n_part = jblocks;
    }
    if ((rc = ws(c.slot[S_PART], (size_t)n_part * sizeof(double)))) return rc;
    double* part = (double*)c.slot[S_PART].p;
    const double den = rms_denominator(nx, ny, nz);

    double* bufs[2] = {d_phi, (double*)c.slot[S_PONG].p};
    int host_ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* tfp = getenv("LSF_TRACE_MINMAX");
    const bool trace_fp = tfp && atoi(tfp) != 0;
    // Fix passes enqueued per iteration.  A pass that finds the fixed point certified returns at once, but an empty
    // launch still costs ~6 us, so the count follows what the field needs (ctl[4] = most passes that changed cells,
    // read with the stop flag every CHECK_EVERY iterations): three times that plus four.  Too few -> ctl[3], the
    // caller repeats the call with MM_MAX_FIX passes.  Large grids skip the adaptation (minmax_core): their chains of
    // sign flips grow fast (1024^3 two spheres: 4, 7, 9, 16 passes in iterations 4..8 of a call) and 32 launches are
    // 4 % of an iteration there.
    int cap = exact_mode == MM_FP_FULL ? MM_MAX_FIX : MM_FIX_START;
    if (const char* e = getenv("LSF_MINMAX_FIX_START")) // test hook: start with too few passes to exercise the rerun
        if (exact_mode == MM_FP_ADAPTIVE) cap = std::min(MM_MAX_FIX, std::max(1, atoi(e)));
    for (int it = 0; it < iter; ++it) { // DO n = 1,iter (set3d.f90:394)
        const double* A = bufs[it & 1];
        double* B = bufs[(it + 1) & 1];
        const int32_t* mask = it == 0 ? d_nb : nullptr;
        if (fixed_point) {
            HIPCHK(hipMemsetAsync(chg, 0, CHG_BYTES, st));
            const dim3 g((unsigned)fp_blocks), b(256);
            const dim3 gwide((unsigned)std::min<long>(cdiv(fp_chunks, 64), 4096)), gthin((unsigned)std::min<long>(cdiv(fp_chunks, 64), 1024));
            const int epoch0 = it * (MM_MAX_FIX + 1) + 1; // stamps of this iteration: epoch0+1 .. epoch0+MM_MAX_FIX
            hipLaunchKernelGGL((k_minmax_fp<0>), g, b, 0, st, A, B, mask, nx, ny, nz, dx, h1, bflag, stamp, fp_blocks, 0, 0,
                               (const int*)nullptr, (int*)nullptr, part, ctl);
            for (int f = 0; f < cap; ++f)
                hipLaunchKernelGGL((k_minmax_fp<1>), f < 3 ? gwide : gthin, b, 0, st, A, B, mask, nx, ny, nz, dx, h1, bflag,
                                   stamp, fp_chunks, epoch0 + f, f == 0 ? 1 : 0,
                                   f == 0 ? (const int*)nullptr : (const int*)(chg + f - 1), chg + f, part, ctl);
            // pass 2 also records how many fix passes changed cells (first = cap) and flags an uncertified iteration
            hipLaunchKernelGGL((k_minmax_fp<2>), gwide, b, 0, st, A, B, mask, nx, ny, nz, dx, h1, bflag, stamp, fp_chunks, 0,
                               cap, (const int*)(chg + cap - 1), (int*)nullptr, part, ctl);
            hipLaunchKernelGGL(k_reduce_slices, dim3(256), dim3(256), 0, st, (const double*)part, fp_chunks, part2);
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(RED_T), 0, st, (const double*)part2, 256L, den, tol, d_trace,
                               std::max(iter, 1), ctl);
        } else if (order == LSF_ORDER_GS) {
            const int nplanes = (int)tl->off.size() - 1;
            for (int P = 0; P < nplanes; ++P) {
                const int cnt = tl->off[P + 1] - tl->off[P];
                if (cnt <= 0) continue;
                hipLaunchKernelGGL((k_minmax_gs_plane<MM_TA>), dim3(cnt), dim3(64), 0, st, A, B, mask, nx, ny, nz,
                                   tl->d + tl->off[P], nTi, nTj, nTk, dx, h1, part, ctl);
            }
        } else {
            hipLaunchKernelGGL(k_minmax_jacobi, dim3(jblocks), dim3(256), 0, st, A, B, mask, nx, ny, nz, dx, h1, part,
                               ctl);
        }
        if (!fixed_point)
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(RED_T), 0, st, part, n_part, den, tol, d_trace, std::max(iter, 1),
                               ctl);
        if (fixed_point && trace_fp) {
            int hc[MM_MAX_FIX + 1] = {0};
            HIPCHK(hipMemcpyAsync(hc, chg, sizeof hc, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            fprintf(stderr, "[lsf] min/max iteration %d: chunks changed per fix pass:", it + 1);
            for (int f = 0; f < MM_MAX_FIX; ++f) fprintf(stderr, " %d", hc[f]);
            fprintf(stderr, "\n");
        }
        // the adaptive pass count looks at the device early (after iterations 1, 2 and 4), then with the stop flag
        const bool early = fixed_point && exact_mode == MM_FP_ADAPTIVE && (it == 0 || it == 1 || it == 3);
        if (((it + 1) % CHECK_EVERY == 0 || early) && it + 1 < iter) {
            HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (host_ctl[0] || host_ctl[3]) break;
            if (exact_mode == MM_FP_ADAPTIVE) cap = std::min(MM_MAX_FIX, std::max(8, 3 * host_ctl[4] + 4));
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (fixed_point && getenv("LSF_TRACE")) {
        int hc[MM_MAX_FIX + 1] = {0};
        HIPCHK(hipMemcpy(hc, chg, sizeof hc, hipMemcpyDeviceToHost));
        int used = 0;
        for (int f = 0; f < MM_MAX_FIX; ++f) used += hc[f] != 0;
        fprintf(stderr, "[lsf] min/max fixed point: last iteration needed %d fix passes that changed cells (%d enqueued)%s\n",
                used, cap, host_ctl[3] ? "; NOT certified -> rerun" : "");
    }
    if (inexact) *inexact = host_ctl[3] != 0;
    if (host_ctl[3]) return LSF_OK; // caller restores the input and reruns with the tile wavefront
    const int nit = host_ctl[1];
    const bool stopped_early = host_ctl[0] != 0; // converged or NaN: EXIT/STOP before narrowBand
    // masks the host would hold now (set3d.f90:448-460)
    if (nit >= 1) {
        const double* src = nullptr;
        if (!stopped_early) src = bufs[nit & 1];                  // band refreshed after the last iteration
        else if (nit >= 2) src = bufs[(nit - 1) & 1];             // refreshed after iteration nit-1
        if (src && (rc = narrowband_core(src, d_nb, d_sb, n, dx, st))) return rc;
    }
    if (bufs[nit & 1] != d_phi)
        HIPCHK(hipMemcpyAsync(d_phi, bufs[nit & 1], n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (rms_trace && trace_cap > 0 && nit > 0)
        HIPCHK(hipMemcpyAsync(rms_trace, d_trace, sizeof(double) * (size_t)std::min(nit, trace_cap),
                              hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (iters_done) *iters_done = nit;
    if (host_ctl[2]) return fail(LSF_ERR_NAN, "RMS became NaN (the reference STOPs here, set3d.f90:458)");
    return LSF_OK;
}

// The min/max flow on the narrow band only (lsf_minmax_band.hpp): the list of cells that can ever be in the band is built once
// per call and the flow runs on compact arrays; the field is written once, at the end.  Both orderings (the Jacobi ordering is
// the start pass and the RMS pass alone).  *dense = true: not run (a list beyond 3.5 M cells + 30 % of the grid, no band cell at all, a
// field beyond 32-bit point indices or brick keys, or an iteration that 62 fix passes did not certify -- never observed) and phi, the masks
// untouched: the caller takes the dense executors.
int minmax_band_impl(double* d_phi, int32_t* d_nb, int32_t* d_sb, int nx, int ny, int nz, int iter, double dx, double h1, double tol,
                     int mode, int* iters_done, double* rms_trace, int trace_cap, hipStream_t st, bool* dense)
{
    *dense = true;
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if (n > (size_t)0x7fffffff) return LSF_OK;
    const bool gs = (mode & LSF_ORDER_MASK) == LSF_ORDER_GS;
    Ctx& c = ctx();
    if (gs && c.mm_band_off) return LSF_OK; // the looping launch timed out on this device before (said once, below)
    int rc;
    const long nblk = (long)((n + MB_SCAN - 1) / MB_SCAN);
    // staging of the list build (the band executor has no second field).  Asked for at the size the dense executors want their second
    // field at (n doubles; the staging uses n ints of it): a call that turns out to need them -- a band above a quarter of the grid,
    // cube40 as shipped -- does not free and allocate the buffer a second time
    if ((rc = ws(c.slot[S_PONG], n * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_MB_CNT], (size_t)(2 * nblk + 8) * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_CTL], 64))) return rc;
    int* staging = (int*)c.slot[S_PONG].p;
    int* counts = (int*)c.slot[S_MB_CNT].p;
    int* offsets = counts + ((nblk + 3) & ~3L); // 16-byte aligned like counts (k_mb_offsets moves vectors)
    const bool trace = getenv("LSF_TRACE") != nullptr;
    const double t_build0 = trace ? now_s() : 0.0;
    hipLaunchKernelGGL(k_mb_collect, dim3((unsigned)nblk), dim3(256), 0, st, (const double*)d_phi, (const int32_t*)d_nb, nx, ny, nz, dx,
                       staging, counts);
    hipLaunchKernelGGL(k_mb_offsets, dim3(1), dim3(1024), 0, st, (const int*)counts, nblk, offsets);
    int nL = 0;
    HIPCHK(hipMemcpyAsync(&nL, offsets + nblk, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // Which executor is faster is a question of list cells against grid points (round 6, profiles/r06_minmax_small.txt: two-sphere
    // fields of 48^3 ... 384^3 with 2 ... 57 % of the grid in the band, ms per exact iteration): this executor costs ~0.02 ms + 0.035 ms
    // per million LIST cells, the dense one ~0.15 ms (its launches) + 0.0105 ms per million GRID points -- so the band executor takes
    // every list up to 3.5 M cells + 30 % of the grid.  (Rounds 5: "a quarter of the grid", which sent the reference's default run --
    // 62^3, band 35 % -- to the dense executor: 0.17 ms per iteration against 0.022.)  LSF_MINMAX_BAND_MAX (per cent of the grid)
    // replaces the rule: test hook.
    bool take = nL > 0 && (double)nL <= 3.5e6 + 0.30 * (double)n;
    if (const char* e = getenv("LSF_MINMAX_BAND_MAX")) take = nL > 0 && (double)nL * 100.0 <= std::min(100.0, std::max(0.0, atof(e))) * (double)n;
    if (!take) return LSF_OK;
    const int nchunks = (nL + MB_CH - 1) / MB_CH;
    if ((rc = ws(c.slot[S_MB_L], (size_t)nL * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_MB_NB6], (size_t)nL * 6 * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_MB_AOLD], (size_t)nL * 2 * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_MB_A0], (size_t)nL * 4 * sizeof(double)))) return rc; // curvature | three downstream values
    if ((rc = ws(c.slot[S_MB_BAND], (size_t)nL))) return rc;
    if ((rc = ws(c.slot[S_BFLAG], (size_t)nchunks * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_STAMP], (size_t)nchunks * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_PART], (size_t)nchunks * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_PART2], 256 * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_CHG], 1024))) return rc;
    if ((rc = ws(c.slot[S_TRACE], (size_t)std::max(iter, 1) * sizeof(double)))) return rc;
    int* L = (int*)c.slot[S_MB_L].p;
    double* V[2] = {(double*)c.slot[S_MB_AOLD].p, (double*)c.slot[S_MB_AOLD].p + nL};
    double* part = (double*)c.slot[S_PART].p;
    double* part2 = (double*)c.slot[S_PART2].p;
    int* chg = (int*)c.slot[S_CHG].p;
    int* ctl = (int*)c.slot[S_CTL].p;
    double* d_trace = (double*)c.slot[S_TRACE].p;
    MbArgs a;
    a.F = d_phi, a.L = L, a.nb = (const int*)c.slot[S_MB_NB6].p, a.isband = (unsigned char*)c.slot[S_MB_BAND].p;
    a.curv = (double*)c.slot[S_MB_A0].p, a.down = a.curv + nL;
    a.nL = nL, a.sx = nx + 1, a.sxy = (long)(nx + 1) * (ny + 1), a.dx = dx, a.h1 = h1;
    a.chunkflag = (int*)c.slot[S_BFLAG].p, a.stamp = (int*)c.slot[S_STAMP].p, a.nchunks = nchunks, a.partials = part, a.ctl = ctl;
    a.chg = chg;
    const dim3 b256(256), gl((unsigned)nchunks), ge((unsigned)cdiv(nL, 256));
    {
        // the list in memory order and its brick keys -> sorted by key (both pairs of arrays live in the staging buffer's tail and in
        // the slots of the list: the scan's segments are no longer needed once they have been gathered)
        const int nbx = cdiv(nx + 1, 8), nby = cdiv(ny + 1, 8);
        if ((double)nbx * nby * cdiv(nz + 1, 4) * 256.0 > 4.0e9) return LSF_OK; // keys beyond 32 bits: dense executor
        if ((rc = ws(c.slot[S_MB_KEY], (size_t)nL * 3 * sizeof(int)))) return rc;
        unsigned* key_in = (unsigned*)c.slot[S_MB_KEY].p;
        unsigned* key = key_in + nL;
        int* L_in = (int*)(key + nL);
        hipLaunchKernelGGL(k_mb_gather, dim3((unsigned)((nblk + 3) / 4)), b256, 0, st, (const int*)staging, (const int*)counts,
                           (const int*)offsets, nblk, nx + 1, ny + 1, nbx, nby, L_in, key_in);
        size_t tmp_bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, key, L_in, L, (size_t)nL, 0, 32, st));
        void* tmp = staging; // (free again: its segments have been gathered -- in stream order)
        if (tmp_bytes > n * sizeof(int)) { // tiny grids
            if ((rc = ws(c.slot[S_MB_TMP], tmp_bytes))) return rc;
            tmp = c.slot[S_MB_TMP].p;
        }
        HIPCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, key_in, key, L_in, L, (size_t)nL, 0, 32, st));
        hipLaunchKernelGGL(k_mb_links, ge, b256, 0, st, (const int*)L, (const unsigned*)key, nL, nx + 1, ny + 1, nbx, nby, (const double*)d_phi,
                           (int*)c.slot[S_MB_NB6].p, V[0]);
    }
    HIPCHK(hipMemsetAsync(ctl, 0, 64, st));
    HIPCHK(hipMemsetAsync(a.stamp, 0, (size_t)nchunks * sizeof(int), st));
    if (trace) {
        HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[lsf] min/max on the band: %d list cells (%.2f %% of the grid), %d chunks; list built in %.3f ms\n", nL,
                100.0 * nL / (double)n, nchunks, (now_s() - t_build0) * 1e3);
    }
    const double den = rms_denominator(nx, ny, nz);
    const char* tfp = getenv("LSF_TRACE_MINMAX");
    const bool trace_fp = tfp && atoi(tfp) != 0;
    auto args_of = [&](int it) {
        MbArgs q = a;
        q.A = V[it & 1], q.X = V[(it + 1) & 1], q.nbmask = it == 0 ? d_nb : nullptr;
        return q;
    };
    // Exact ordering: the first fix pass as a wide launch over every band chunk, every further pass inside one small resident
    // launch that loops until a pass changes nothing (k_minmax_band_tail).  Epochs of iteration it: it * MB_EPOCHS + 1 + pass.
    int tail_max = MB_EPOCHS - 2;
    if (const char* e = getenv("LSF_MINMAX_TAIL_MAX")) tail_max = std::min(tail_max, std::max(0, atoi(e))); // test hook: too few passes (0: no certifying pass at all)
    int* bar = chg + MB_EPOCHS; // barrier word of the tail launch, behind its per-pass change counts
    // blocks of the looping launch: one per CU (measured best), never more than the device admits at once -- its grid barrier
    // needs every block resident (a partitioned device, fewer CUs): asked once per device
    if (gs && c.mm_tail_blocks == 0) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t pr;
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipGetDeviceProperties(&pr, dev));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_minmax_band_tail, 256, 0));
        c.mm_tail_blocks = std::max(1, std::min(MB_TAIL_BLOCKS, std::max(1, per_cu) * pr.multiProcessorCount));
    }
    int tail_blocks = gs ? c.mm_tail_blocks : MB_TAIL_BLOCKS;
    if (const char* e = getenv("LSF_MINMAX_TAIL_BLOCKS")) tail_blocks = std::min(512, std::max(8, atoi(e))); // measurement aid (all resident: <= 2 per CU)
    unsigned long long tail_timeout = 200000000ull; // 2 s of the 100 MHz clock
    if (const char* e = getenv("LSF_MINMAX_TAIL_TIMEOUT_TICKS")) tail_timeout = strtoull(e, nullptr, 10); // test hook
    int host_ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iter; ++it) { // DO n = 1,iter (set3d.f90:394)
        const MbArgs q = args_of(it);
        hipLaunchKernelGGL((k_minmax_band<0>), gl, b256, 0, st, q);
        if (gs) {
            const int epoch0 = it * MB_EPOCHS + 1;
            hipLaunchKernelGGL(k_minmax_band_fix, gl, b256, 0, st, q, epoch0, chg);
            hipLaunchKernelGGL(k_minmax_band_tail, dim3(tail_blocks), b256, 0, st, q, epoch0 + 1, tail_max, (const int*)chg, chg + 1, bar,
                               tail_timeout);
        }
        hipLaunchKernelGGL((k_minmax_band<2>), gl, b256, 0, st, q);
        if (nchunks > 16384) {
            hipLaunchKernelGGL(k_reduce_slices, dim3(256), dim3(256), 0, st, (const double*)part, (long)nchunks, part2);
            hipLaunchKernelGGL(k_mb_finish, dim3(1), dim3(RED_T), 0, st, (const double*)part2, 256L, den, tol, d_trace, std::max(iter, 1), ctl);
        } else {
            hipLaunchKernelGGL(k_mb_finish, dim3(1), dim3(RED_T), 0, st, (const double*)part, (long)nchunks, den, tol, d_trace, std::max(iter, 1),
                               ctl);
        }
        if (gs && trace_fp) {
            int hc[MB_EPOCHS] = {0};
            HIPCHK(hipMemcpyAsync(hc, chg, sizeof hc, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            fprintf(stderr, "[lsf] min/max iteration %d (band): chunks changed per fix pass:", it + 1);
            for (int f = 0; f < MB_EPOCHS && (f == 0 || hc[f - 1]); ++f) fprintf(stderr, " %d", hc[f]);
            fprintf(stderr, "\n");
        }
        if ((it + 1) % CHECK_EVERY == 0 && it + 1 < iter) {
            HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (host_ctl[0] || host_ctl[3]) break;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (host_ctl[3]) { // an iteration the tail launch did not certify (never observed): phi and the masks have not been written
        if (host_ctl[5]) {
            // a block waited for the others longer than the bound: the grid was not resident as a whole (a device shared with other
            // work).  Said once, whatever LSF_TRACE holds, and remembered: later calls go to the dense executors at once instead of
            // paying the bound and the list build again.
            c.mm_band_off = true;
            fprintf(stderr, "[lsf] min/max on the band: the looping launch (%d blocks) timed out waiting for its own blocks -- is the device shared? "
                            "This and every later exact min/max call on this device use the dense executors.\n", tail_blocks);
        } else if (trace)
            fprintf(stderr, "[lsf] min/max on the band: iteration %d NOT certified -> dense executor\n", host_ctl[1] + 1);
        return LSF_OK;
    }
    if (gs && trace) fprintf(stderr, "[lsf] min/max on the band: at most %d fix passes per iteration\n", host_ctl[4] + 1);
    *dense = false;
    const int nit = host_ctl[1];
    const bool stopped_early = host_ctl[0] != 0; // converged or NaN: EXIT/STOP before narrowBand
    if (nit >= 1) hipLaunchKernelGGL(k_mb_scatter, ge, b256, 0, st, (const int*)L, (const double*)V[nit & 1], nL, d_phi);
    // masks the host would hold now (set3d.f90:448-460): narrowBand of the final field, or -- when the loop was left by EXIT / STOP
    // -- of the field before the last iteration: the final field with the list cells as that iteration froze them
    if (nit >= 1 && (!stopped_early || nit >= 2)) {
        if ((rc = narrowband_core(d_phi, d_nb, d_sb, n, dx, st))) return rc;
        if (stopped_early) hipLaunchKernelGGL(k_mb_patch_masks, ge, b256, 0, st, (const int*)L, (const double*)V[(nit - 1) & 1], nL, dx, d_nb, d_sb);
    }
    if (rms_trace && trace_cap > 0 && nit > 0)
        HIPCHK(hipMemcpyAsync(rms_trace, d_trace, sizeof(double) * (size_t)std::min(nit, trace_cap), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (iters_done) *iters_done = nit;
    if (host_ctl[2]) return fail(LSF_ERR_NAN, "RMS became NaN (the reference STOPs here, set3d.f90:458)");
    return LSF_OK;
}

// Exact ordering: fixed-point passes (fast), as many per iteration as the field has needed so far; if a fixed point is
// ever not certified, restore the input and redo the call with MM_MAX_FIX passes per iteration, and if that is still
// not enough (never observed) with the tile-hyperplane wavefront.
int minmax_core(double* d_phi, int32_t* d_nb, int32_t* d_sb, int nx, int ny, int nz, int iter, double dx,
                double h1, double tol, int mode, int* iters_done, double* rms_trace, int trace_cap,
                hipStream_t st)
{
    const char* e = getenv("LSF_MINMAX_TILES");
    const bool force_tiles = e && atoi(e) != 0;
    const int order = mode & LSF_ORDER_MASK;
    const bool args_ok = d_phi && d_nb && d_sb && iter > 0 && (order == LSF_ORDER_GS || order == LSF_ORDER_JACOBI) && !check_dims(nx, ny, nz);
    // Default: the executor whose cost follows the band (lsf_minmax_band.hpp).  LSF_MINMAX_DENSE=1, a band above a quarter of the
    // grid or a field beyond 32-bit point indices: the dense executors below.
    const char* ed = getenv("LSF_MINMAX_DENSE");
    if (args_ok && !force_tiles && !(ed && atoi(ed) != 0)) {
        bool dense = false;
        const int rcb = minmax_band_impl(d_phi, d_nb, d_sb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace, trace_cap, st, &dense);
        if (rcb != LSF_OK || !dense) return rcb;
    }
    if (order != LSF_ORDER_GS || force_tiles || !args_ok)
        return minmax_core_impl(d_phi, d_nb, d_sb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace,
                                trace_cap, st, MM_TILES, nullptr);
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    // an uncertified attempt returns before it touches the masks, so only phi needs a copy to start over from
    int rc = ws(c.slot[S_BACKUP], n * sizeof(double));
    if (rc) return rc;
    char* bk = (char*)c.slot[S_BACKUP].p;
    HIPCHK(hipMemcpyAsync(bk, d_phi, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    for (int exact_mode : {MM_FP_ADAPTIVE, MM_FP_FULL}) {
        if (exact_mode == MM_FP_ADAPTIVE && n >= (size_t)200000000) continue; // >= ~585^3: always the full count
        bool inexact = false;
        rc = minmax_core_impl(d_phi, d_nb, d_sb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace, trace_cap,
                              st, exact_mode, &inexact);
        if (rc != LSF_OK || !inexact) return rc;
        HIPCHK(hipMemcpyAsync(d_phi, bk, n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return minmax_core_impl(d_phi, d_nb, d_sb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace, trace_cap,
                            st, MM_TILES, nullptr);
}
