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
// per call, phi stays in place, every pass of an iteration runs over the list.  Both orderings (the Jacobi ordering is the start
// pass and the RMS pass alone).  *dense = true: not run (band above a quarter of the grid, no band cell at all, or a field beyond
// 32-bit point indices) -- the caller takes the dense executor; *inexact as in minmax_core_impl (the field is restored).
int minmax_band_impl(double* d_phi, int32_t* d_nb, int32_t* d_sb, int nx, int ny, int nz, int iter, double dx, double h1, double tol,
                     int mode, int* iters_done, double* rms_trace, int trace_cap, hipStream_t st, int exact_mode, bool* inexact,
                     bool* dense)
{
    *dense = true;
    if (inexact) *inexact = false;
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if (n > (size_t)0x7fffffff) return LSF_OK;
    const bool gs = (mode & LSF_ORDER_MASK) == LSF_ORDER_GS;
    Ctx& c = ctx();
    int rc;
    const long nblk = (long)((n + MB_SCAN - 1) / MB_SCAN);
    if ((rc = ws(c.slot[S_PONG], n * sizeof(int)))) return rc; // staging of the list build (the band executor has no second field)
    if ((rc = ws(c.slot[S_MB_CNT], (size_t)(2 * nblk + 2) * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_CTL], 64))) return rc;
    int* staging = (int*)c.slot[S_PONG].p;
    int* counts = (int*)c.slot[S_MB_CNT].p;
    int* offsets = counts + nblk;
    const double t_build0 = getenv("LSF_TRACE") ? now_s() : 0.0;
    hipLaunchKernelGGL(k_mb_collect, dim3((unsigned)nblk), dim3(256), 0, st, (const double*)d_phi, (const int32_t*)d_nb, nx, ny, nz, dx,
                       staging, counts);
    hipLaunchKernelGGL(k_mb_offsets, dim3(1), dim3(1024), 0, st, (const int*)counts, nblk, offsets);
    int nL = 0;
    HIPCHK(hipMemcpyAsync(&nL, offsets + nblk, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nL <= 0 || (size_t)nL * 4 > n) return LSF_OK; // dense executor
    *dense = false;
    const int nchunks = (nL + MB_CH - 1) / MB_CH;
    if ((rc = ws(c.slot[S_MB_L], (size_t)nL * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_MB_NB6], (size_t)nL * 6 * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_MB_AOLD], (size_t)nL * 2 * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_MB_A0], (size_t)nL * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_MB_BAND], (size_t)nL))) return rc;
    if ((rc = ws(c.slot[S_BFLAG], (size_t)nchunks * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_STAMP], (size_t)nchunks * sizeof(int)))) return rc;
    if ((rc = ws(c.slot[S_PART], (size_t)nchunks * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_PART2], 256 * sizeof(double)))) return rc;
    constexpr size_t CHG_BYTES = (MM_MAX_FIX + 1) * sizeof(int);
    if ((rc = ws(c.slot[S_CHG], CHG_BYTES))) return rc;
    if ((rc = ws(c.slot[S_TRACE], (size_t)std::max(iter, 1) * sizeof(double)))) return rc;
    int* L = (int*)c.slot[S_MB_L].p;
    int* nb6 = (int*)c.slot[S_MB_NB6].p;
    double* aold[2] = {(double*)c.slot[S_MB_AOLD].p, (double*)c.slot[S_MB_AOLD].p + nL};
    double* a0 = (double*)c.slot[S_MB_A0].p;
    unsigned char* isband = (unsigned char*)c.slot[S_MB_BAND].p;
    int* chunkflag = (int*)c.slot[S_BFLAG].p;
    int* stamp = (int*)c.slot[S_STAMP].p;
    double* part = (double*)c.slot[S_PART].p;
    double* part2 = (double*)c.slot[S_PART2].p;
    int* chg = (int*)c.slot[S_CHG].p;
    int* ctl = (int*)c.slot[S_CTL].p;
    double* d_trace = (double*)c.slot[S_TRACE].p;
    const int sx = nx + 1;
    const long sxy = (long)(nx + 1) * (ny + 1);
    const dim3 b256(256), gl((unsigned)nchunks);
    hipLaunchKernelGGL(k_mb_gather, dim3((unsigned)nblk), b256, 0, st, (const int*)staging, (const int*)counts, (const int*)offsets,
                       (const double*)d_phi, L, aold[0], a0);
    hipLaunchKernelGGL(k_mb_links, gl, b256, 0, st, (const int*)L, nL, sx, (int)sxy, nb6);
    HIPCHK(hipMemsetAsync(ctl, 0, 64, st));
    HIPCHK(hipMemsetAsync(stamp, 0, (size_t)nchunks * sizeof(int), st));
    if (getenv("LSF_TRACE")) {
        HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[lsf] min/max on the band: %d list cells (%.2f %% of the grid), %d chunks; list built in %.3f ms\n", nL,
                100.0 * nL / (double)n, nchunks, (now_s() - t_build0) * 1e3);
    }
    const double den = rms_denominator(nx, ny, nz);
    int host_ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cap = !gs ? 0 : (exact_mode == MM_FP_FULL ? MM_MAX_FIX : MM_FIX_START);
    if (const char* e = getenv("LSF_MINMAX_FIX_START")) // test hook: start with too few passes to exercise the rerun
        if (gs && exact_mode == MM_FP_ADAPTIVE) cap = std::min(MM_MAX_FIX, std::max(1, atoi(e)));
    const dim3 gwide((unsigned)std::min(cdiv(nchunks, 64), 4096)), gthin((unsigned)std::min(cdiv(nchunks, 64), 1024));
    const char* tfp = getenv("LSF_TRACE_MINMAX");
    const bool trace_fp = tfp && atoi(tfp) != 0;
    for (int it = 0; it < iter; ++it) { // DO n = 1,iter (set3d.f90:394)
        const double* A = aold[it & 1];
        double* An = aold[(it + 1) & 1];
        const int32_t* mask = it == 0 ? d_nb : nullptr;
        const int epoch0 = it * (MM_MAX_FIX + 1) + 1; // stamps of this iteration: epoch0+1 .. epoch0+MM_MAX_FIX
        if (cap > 0) HIPCHK(hipMemsetAsync(chg, 0, CHG_BYTES, st));
        hipLaunchKernelGGL((k_minmax_band<0>), gl, b256, 0, st, d_phi, A, An, mask, (const int*)L, (const int*)nb6, isband, nL, sx, sxy, dx, h1,
                           chunkflag, stamp, nchunks, 0, 0, (const int*)nullptr, (int*)nullptr, part, ctl);
        for (int f = 0; f < cap; ++f)
            hipLaunchKernelGGL((k_minmax_band<1>), f < 3 ? gwide : gthin, b256, 0, st, d_phi, A, An, mask, (const int*)L, (const int*)nb6, isband,
                               nL, sx, sxy, dx, h1, chunkflag, stamp, nchunks, epoch0 + f, f == 0 ? 1 : 0,
                               f == 0 ? (const int*)nullptr : (const int*)(chg + f - 1), chg + f, part, ctl);
        hipLaunchKernelGGL((k_minmax_band<2>), gl, b256, 0, st, d_phi, A, An, mask, (const int*)L, (const int*)nb6, isband, nL, sx, sxy, dx, h1,
                           chunkflag, stamp, nchunks, 0, cap, cap > 0 ? (const int*)(chg + cap - 1) : (const int*)nullptr, (int*)nullptr, part,
                           ctl);
        if (nchunks > 16384) {
            hipLaunchKernelGGL(k_reduce_slices, dim3(256), dim3(256), 0, st, (const double*)part, (long)nchunks, part2);
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(RED_T), 0, st, (const double*)part2, 256L, den, tol, d_trace, std::max(iter, 1), ctl);
        } else {
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(RED_T), 0, st, (const double*)part, (long)nchunks, den, tol, d_trace, std::max(iter, 1),
                               ctl);
        }
        if (cap > 0 && trace_fp) {
            int hc[MM_MAX_FIX + 1] = {0};
            HIPCHK(hipMemcpyAsync(hc, chg, sizeof hc, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            fprintf(stderr, "[lsf] min/max iteration %d (band): chunks changed per fix pass:", it + 1);
            for (int f = 0; f < MM_MAX_FIX; ++f) fprintf(stderr, " %d", hc[f]);
            fprintf(stderr, "\n");
        }
        const bool early = gs && exact_mode == MM_FP_ADAPTIVE && (it == 0 || it == 1 || it == 3);
        if (((it + 1) % CHECK_EVERY == 0 || early) && it + 1 < iter) {
            HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (host_ctl[0] || host_ctl[3]) break;
            if (gs && exact_mode == MM_FP_ADAPTIVE) cap = std::min(MM_MAX_FIX, std::max(8, 3 * host_ctl[4] + 4));
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (gs && getenv("LSF_TRACE"))
        fprintf(stderr, "[lsf] min/max on the band: at most %d fix passes changed cells (%d enqueued at the end)%s\n", host_ctl[4], cap,
                host_ctl[3] ? "; NOT certified -> rerun" : "");
    const dim3 ge((unsigned)cdiv(nL, 256));
    if (host_ctl[3]) { // the caller repeats the call: phi as it was on entry (the masks have not been touched)
        hipLaunchKernelGGL(k_mb_restore, ge, b256, 0, st, (const int*)L, (const double*)a0, nL, d_phi);
        HIPCHK(hipStreamSynchronize(st));
        if (inexact) *inexact = true;
        return LSF_OK;
    }
    const int nit = host_ctl[1];
    const bool stopped_early = host_ctl[0] != 0; // converged or NaN: EXIT/STOP before narrowBand
    // masks the host would hold now (set3d.f90:448-460): narrowBand of the final field, or -- when the loop was left by EXIT / STOP
    // -- of the field before the last iteration: the final field with the list cells as that iteration froze them
    if (nit >= 1 && (!stopped_early || nit >= 2)) {
        if ((rc = narrowband_core(d_phi, d_nb, d_sb, n, dx, st))) return rc;
        if (stopped_early) hipLaunchKernelGGL(k_mb_patch_masks, ge, b256, 0, st, (const int*)L, (const double*)aold[(nit - 1) & 1], nL, dx, d_nb, d_sb);
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
        for (int exact_mode : {MM_FP_ADAPTIVE, MM_FP_FULL}) {
            bool inexact = false, dense = false;
            const int rcb = minmax_band_impl(d_phi, d_nb, d_sb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace, trace_cap, st,
                                             exact_mode, &inexact, &dense);
            if (dense) break;
            if (rcb != LSF_OK || !inexact) return rcb;
        }
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
