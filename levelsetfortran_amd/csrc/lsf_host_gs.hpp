// lsf_host_gs.hpp -- host side of the exact Gauss-Seidel reinit on one device: lookup tables of the tile shapes, the schedule switch, and
// reinit_slot_core (dataflow launch / slot launches).  Included by lsf_api.hip inside its anonymous namespace (uses Ctx, ws, fail, HIPCHK,
// get_tiles, get_skew_tiles, skew_spacing, the gs_* switches).
#pragma once

// lookup tables of a skewed tile shape (lsf_skew.hpp: sk_fill_tables), built once per shape and device
int get_sk_tables(int ta, int wy, int wz, int by, const uint32_t** out)
{
    Ctx& c = ctx();
    const int key = by * 256 + wy * 16 + wz + (ta == 32 ? 1 << 16 : 0);
    auto it = c.sk_tables.find(key);
    if (it == c.sk_tables.end()) {
        std::vector<uint32_t> h;
#define LSF_SK_TAB(WY_, WZ_, BY_)                                                            \
    do {                                                                                     \
        using T_ = SkTile<16, WY_, WZ_, BY_>;                                                \
        h.assign((size_t)T_::REL_WORDS + T_::OFF_WORDS, 0u);                                 \
        sk_fill_tables<16, WY_, WZ_, BY_>(h.data());                                         \
    } while (0)
#ifdef LSF_EXPERIMENTS
        if (ta == 32) {
            using T_ = SkTile<32, 2, 2, 5>;
            h.assign((size_t)T_::REL_WORDS + T_::OFF_WORDS, 0u);
            sk_fill_tables<32, 2, 2, 5>(h.data());
        } else
#endif
        {
            LSF_SK_SHAPES(LSF_SK_TAB, wy, wz, by);
        }
#undef LSF_SK_TAB
        uint32_t* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, h.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = c.sk_tables.emplace(key, d).first;
    }
    *out = it->second;
    return LSF_OK;
}

// LSF_GS_SCHEDULE selects how the exact Gauss-Seidel tile graph is executed (all are bit-identical):
//   "dataflow" (default) skewed tiles, one launch per batch of sweeps, dependencies resolved in the kernel
//   "skew"               skewed tiles, one launch per time slot (also the fallback of a dataflow launch that timed out)
//   "slots"              box tiles, overlapped sweeps, one launch per time slot
//   "planes"             box tiles, one launch per tile hyperplane, one sweep at a time
thread_local int g_schedule_override = -2; // set while a call is repeated on the slot schedule (see below)
int gs_schedule()
{
    if (g_schedule_override != -2) return g_schedule_override;
    const char* e = getenv("LSF_GS_SCHEDULE");
    if (e && std::strcmp(e, "planes") == 0) return 0;
    if (e && std::strcmp(e, "skew") == 0) return 3;
    if (e && std::strcmp(e, "slots") == 0) return 1;
    if (e && std::strcmp(e, "dataflow") == 0) return 5;
    return -1; // unset: dataflow on skewed tiles
}

// ---------------------------------------------------------------------------------------------
// Exact-GS reinit: the dataflow launch (default) and the slot-synchronous schedules with overlapped sweeps.
// Slot schedules: one launch per time slot; a slot holds the tile hyperplane P = slot - start[g] of every sweep g in
// flight (at most three).  start[] obeys two spacing rules -- start[g] >= start[g-1] + H(raster flip) so that a tile's
// neighbours finished the sweep before, and start[g] >= start[g-3] + nPlanes + 1 so that the stop verdict of the sweep
// whose buffer is overwritten is known -- hence every predecessor of a task ran in an earlier launch.  The BC, the
// wall mirror and the RMS epilogue are fused into the tile kernel.
// ---------------------------------------------------------------------------------------------
int reinit_slot_core(double* d_phi, const double* d_phiS_in, int nx, int ny, int nz, int iter, double dx, double h,
                     double tol, int mode, int first_raster, int* sweeps_done, double* rms_trace, int trace_cap,
                     hipStream_t st)
{
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    const bool strict = (mode & LSF_ARITH_STRICT) != 0;
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    const int max_sweeps = iter + 1;
    int ta = gs_ta();
    int nyc = gs_ny();
    int sched = gs_schedule();
    // Default: skewed tiles (lsf_skew.hpp), 2 x 2 wavefronts each, dependencies resolved in the kernel (`dataflow`,
    // one launch per batch of sweeps).  Measured per sweep: dataflow / slot launches on skewed tiles (`skew`) / slot
    // launches on the box tiles of lsf_boxtile.hpp (`slots`): 256^3 0.97 / 1.24 / 1.75 ms, 512^3 3.25 / 4.61 / 6.83 ms,
    // 1024^3 24.7 / 25.2 / 38.0 ms.
    if (sched < 0) sched = 5;
    bool persist = sched == 5; // k_reinit_gs_persist
    if (persist) sched = 3;
    // skewed tiles need TA = 16, NY = 5 and at least two interior cells per axis
    const bool skew = sched == 3 && ta == 16 && nyc == 5 && nx >= 3 && ny >= 3 && nz >= 3;
    persist = persist && skew;
    int wy = 1, wz = 1, by = 5, nzc = 4;
    if (skew) {
        gs_skew_w(std::min(nx, ny), nz, strict, &wy, &wz, &by); // (the dataflow launch may swap x and y)
        nyc = by * wy, nzc = 4 * wz; // rows of a tile in y and z
        ta = gs_skew_ta(std::min(std::min(nx, ny), nz) - 1, wy, wz, by);
    }
    // Dataflow launch: the kernel marches along ITS x axis; run it on the x <-> y transposed field so that the march axis
    // is the reference's y, the axis the raster cycle flips in six of its eight transitions (a flip of the march axis
    // spaces two sweeps by n / 16 time slots, a flip of a cross-section axis by n / 16 + its number of tiles: 70 instead of
    // 95 slots per sweep at 512^3).  Costs nbuf + 1 work fields (none of them the caller's) and three transpositions per
    // call; skipped when that does not fit.
    int nbuf = persist ? gs_nbuf() : 3;
    bool tr = persist && gs_march() == 1;
    if (tr) {
        size_t need = 0, fr = 0, tot = 0;
        for (Slot q : {S_PONG, S_PONG2, S_PONG3, S_PONG4, S_PHIS})
            if (q != S_PONG4 || nbuf == 4) need += c.slot[q].bytes >= n * sizeof(double) ? 0 : n * sizeof(double);
        HIPCHK(hipMemGetInfo(&fr, &tot));
        if (need + (2ull << 30) > fr) tr = false, nbuf = 3;
    }
    const int knx = tr ? ny : nx, kny = tr ? nx : ny; // the kernel's view of the grid
    auto ksign = [&](int raster, int* out) {           // raster signs in the kernel's axis order
        const int* r = RASTER_SIGN[raster & 7];
        out[0] = tr ? r[1] : r[0], out[1] = tr ? r[0] : r[1], out[2] = r[2];
    };
    const Slot pong[4] = {S_PONG, S_PONG2, S_PONG3, S_PONG4};
    for (int q = 0; q < (tr ? nbuf : nbuf - 1); ++q)
        if ((rc = ws(c.slot[pong[q]], n * sizeof(double)))) return rc;
    const double* d_phiS = d_phiS_in;
    if (tr) {
        if ((rc = ws(c.slot[S_PHIS], n * sizeof(double)))) return rc;
        const dim3 tg(cdiv(nx + 1, 32), cdiv(ny + 1, 32), (unsigned)std::min(nz + 1, 1024));
        // phiS = phi on entry (subs.f90:731): one pass over phi writes both transposed copies unless the caller has its own
        hipLaunchKernelGGL(k_transpose_xy, tg, dim3(256), 0, st, (const double*)d_phi, (double*)c.slot[S_PONG].p, nx + 1, ny + 1,
                           (long)(nz + 1), d_phiS_in ? (double*)nullptr : (double*)c.slot[S_PHIS].p);
        if (d_phiS_in)
            hipLaunchKernelGGL(k_transpose_xy, tg, dim3(256), 0, st, d_phiS_in, (double*)c.slot[S_PHIS].p, nx + 1, ny + 1,
                               (long)(nz + 1), (double*)nullptr);
        d_phiS = (const double*)c.slot[S_PHIS].p;
    } else if (!d_phiS) {
        if ((rc = ws(c.slot[S_PHIS], n * sizeof(double)))) return rc;
        HIPCHK(hipMemcpyAsync(c.slot[S_PHIS].p, d_phi, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        d_phiS = (const double*)c.slot[S_PHIS].p;
    }
    const bool overlap = sched != 0; // "planes": one sweep at a time (start[g+1] = start[g] + nPlanes)
    const int nTi = cdiv(knx - 1, ta), nTj = cdiv(kny - 1, nyc), nTk = cdiv(nz - 1, nzc);
    const int nT[3] = {nTi, nTj, nTk};
    TileList* tl = nullptr;
    if (skew) rc = get_skew_tiles(knx - 1, nTj, nTk, ta, nyc, nzc, &tl);
    else rc = get_tiles(nTi, nTj, nTk, &tl);
    if (rc) return rc;
    const int np = (int)tl->off.size() - 1;
    if ((rc = ws(c.slot[S_CTL], 64))) return rc;
    if ((rc = ws(c.slot[S_TRACE], (size_t)max_sweeps * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_COLSUM], (size_t)4 * nTj * nTk * sizeof(double)))) return rc;
    int* ctl = (int*)c.slot[S_CTL].p;
    HIPCHK(hipMemsetAsync(ctl, 0, 64, st));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(ctl + 4), 0x7fffffff, 1, st)); // ctl[4]: never 0 (k_reinit_gs_persist: the flag of an absent upstream tile)

    // nbuf field buffers in rotation: sweep g overwrites the result of sweep g - nbuf, so it has to wait for the
    // stop verdict of that sweep only, and consecutive sweeps are spaced by the raster-flip rule alone.
    GsArgs fa;
    std::memset(&fa, 0, sizeof fa);
    if (tr) {
        for (int q = 0; q < nbuf; ++q) fa.buf[q] = (double*)c.slot[pong[q]].p;
    } else {
        fa.buf[0] = d_phi;
        for (int q = 1; q < nbuf; ++q) fa.buf[q] = (double*)c.slot[pong[q - 1]].p;
    }
    fa.nbuf = nbuf;
    fa.quirk_axis = tr ? 0 : 1; // subs.f90:576 concerns the reference's y axis
    fa.phiS = d_phiS;
    fa.nx = knx, fa.ny = kny, fa.nz = nz, fa.nTi = nTi, fa.nTj = nTj, fa.nTk = nTk;
    fa.dx = dx, fa.h = h;
    fa.colsum = (double*)c.slot[S_COLSUM].p;
    fa.trace = (double*)c.slot[S_TRACE].p;
    fa.trace_cap = max_sweeps;
    fa.den = rms_denominator(nx, ny, nz);
    fa.tol = tol;
    fa.ctl = ctl;
    fa.nTiles = (long)nTi * nTj * nTk;
    fa.last_packed = tl->last;
    if (skew && !getenv("LSF_GS_NO_TABLES") && (rc = get_sk_tables(ta, wy, wz, by, &fa.tables))) return rc;

    // start slot of sweep g, generated on demand (slot schedules; never transposed)
    std::vector<long> start{0};
    auto start_of = [&](int g) -> long {
        while ((int)start.size() <= g) {
            const int q = (int)start.size();
            const int* da = RASTER_SIGN[(first_raster + q - 1) & 7];
            const int* db = RASTER_SIGN[(first_raster + q) & 7];
            long H = 2;
            for (int ax = 0; ax < 3; ++ax)
                if (da[ax] != db[ax]) H += nT[ax] - 1;
            if (skew) H = skew_spacing(da, db, nx, ny, nz, ta, nyc, nzc);
            long s0 = start[q - 1] + H;
            if (q >= 3) s0 = std::max(s0, start[q - 3] + np + 1);
            if (!overlap) s0 = start[q - 1] + np;
            start.push_back(s0);
        }
        return start[g];
    };
    int host_ctl[4] = {0, 0, 0, 0};
    prof_begin();
    long launches = 0;
    bool marked = false;
    if (persist) {
        // Dataflow schedule (k_reinit_gs_persist): one launch per batch of up to DF_BATCH = 256 sweeps, one block per tile,
        // dependencies resolved in the kernel.  What depends on the grid, the raster phase of the batch's first sweep and the
        // number of sweeps (start slots, entries per slot, spacing table) is small and cached on the device; the task list
        // itself is rebuilt by k_build_order in front of every launch.
        const long ntiles = tl->off[np];
        // sweeps per launch: up to 256 (a batch costs about one sweep time of fill and drain, measured 26.6 / 50.0 / 73.9 /
        // 97.4 ms for 8 / 16 / 24 / 32 sweeps at 512^3; 256^3 to convergence: 0.645 -> 0.628 ms per sweep against 64 per
        // launch), fewer on very large grids so that the task list stays below 512 MB; a multiple of 8 keeps the raster
        // phase, hence the cached plan, the same
        // (calls of up to 64 sweeps keep the 64-sweep layout of their control arrays: what bench.py times)
        int BATCH = (int)std::max<long>(8, std::min<long>(max_sweeps <= 64 ? 64 : DF_BATCH, (512L << 20) / (ntiles * 8) / 8 * 8));
        if (const char* e = getenv("LSF_DF_BATCH")) BATCH = std::max(8, std::min(BATCH, atoi(e) / 8 * 8)); // test hook: batch boundaries
        const int nM = (knx - 2 + nyc * nTj - 1 + nzc * nTk - 1) / ta + 1; // m_max + 1 (get_skew_tiles)
        const size_t tile_flags = (size_t)BATCH * nM * nTj * nTk;
        // hyperplane counters | 4 KB | leading-hyperplane counters of the sweeps | 4 KB | ticket: the three are polled / updated at
        // very different rates (see DF_PAD)
        constexpr size_t DF_PAD = LSF_DF_PAD;
        if ((rc = ws(c.slot[S_PLANECNT], ((size_t)BATCH * np + 2 * DF_PAD + BATCH + 16) * sizeof(int)))) return rc;
        if ((rc = ws(c.slot[S_BFLAG], tile_flags * sizeof(int)))) return rc;
        if ((rc = ws(c.slot[S_ORDER], (size_t)std::min(BATCH, max_sweeps) * ntiles * sizeof(uint2)))) return rc;
        int* d_cnt = (int*)c.slot[S_PLANECNT].p;
        int* d_done = d_cnt + (size_t)BATCH * np + DF_PAD;
        int* d_ticket = d_done + BATCH + DF_PAD;
        unsigned long long* d_dbg = nullptr;
        // per-tile wait / work times of the dataflow launch: three contended atomics per tile (+70 % run time), so its own
        // switch and not part of LSF_TRACE, whose per-call times are meant to be read as measurements
        if (getenv("LSF_TRACE_TILES")) {
            if ((rc = ws(c.slot[S_DBG], 128))) return rc;
            d_dbg = (unsigned long long*)c.slot[S_DBG].p;
        }
        if (c.plans.size() > 64) { // bounded: every earlier call has synchronised its stream before returning
            for (auto& kv : c.plans)
                if (kv.second.d_meta) HIPCHK(hipFree(kv.second.d_meta));
            c.plans.clear();
        }
        for (int g0 = 0; g0 < max_sweeps; g0 += BATCH) {
            const int ns = std::min(BATCH, max_sweeps - g0), phase = (first_raster + g0) & 7;
            const std::array<int, 6> key{knx, kny, nz, phase, ns, wy * 16 + wz + 256 * (int)tr + 1024 * nbuf + 8192 * by + (ta == 32 ? 1 << 20 : 0)};
            auto it = c.plans.find(key);
            if (it == c.plans.end()) {
                BatchPlan bp;
                std::vector<int> st0(ns, 0), tab(4 * DF_BATCH, 0);
                for (int q = 0; q < ns; ++q) {
                    int da[3], db[3];
                    ksign(phase + q, db);
                    for (int ax = 0; ax < 3; ++ax) tab[4 * q + ax] = db[ax];
                    if (q == 0) continue;
                    ksign(phase + q - 1, da);
                    const long H = skew_spacing(da, db, knx, kny, nz, ta, nyc, nzc);
                    tab[4 * q + 3] = (int)H;
                    long s0 = st0[q - 1] + H;
                    if (q >= nbuf) s0 = std::max<long>(s0, st0[q - nbuf] + np + 1); // list order respects condition (c)
                    st0[q] = (int)s0;
                }
                bp.nslots = st0[ns - 1] + np;
                // entries per slot: hyperplane slot - st0[q] of every sweep q in flight
                std::vector<unsigned> base((size_t)bp.nslots + 1, 0u);
                int lo_s = 0;
                for (int slot = 0; slot < bp.nslots; ++slot) {
                    unsigned cnt = 0;
                    for (int q = lo_s; q < ns && st0[q] <= slot; ++q) {
                        const int P = slot - st0[q];
                        if (P < np) cnt += (unsigned)(tl->off[P + 1] - tl->off[P]);
                    }
                    base[slot + 1] = base[slot] + cnt;
                    while (lo_s < ns && st0[lo_s] + np <= slot + 1) ++lo_s;
                }
                bp.total = (long)base[bp.nslots];
                if (bp.total != (long)ns * ntiles) return fail(LSF_ERR_HIP, "internal: batch plan does not cover every tile");
                std::vector<int> meta;
                meta.insert(meta.end(), st0.begin(), st0.end());
                for (unsigned v : base) meta.push_back((int)v);
                while (meta.size() % 4) meta.push_back(0); // the kernels read a sweep's four table entries as one 16-byte load
                meta.insert(meta.end(), tab.begin(), tab.end());
                for (int P = 0; P < np; ++P) meta.push_back(tl->off[P + 1] - tl->off[P]);
                meta.insert(meta.end(), tl->off.begin(), tl->off.end());
                HIPCHK(hipMalloc((void**)&bp.d_meta, meta.size() * sizeof(int)));
                HIPCHK(hipMemcpy(bp.d_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
                it = c.plans.emplace(key, bp).first;
            }
            const BatchPlan& bp = it->second;
            const int* m_start = bp.d_meta;
            const unsigned* m_base = (const unsigned*)(bp.d_meta + ns);
            const int* m_tab = bp.d_meta + (ns + bp.nslots + 1 + 3) / 4 * 4; // 16-byte aligned (hipMalloc aligns the block)
            const int* m_psize = m_tab + 4 * DF_BATCH;
            const int* m_poff = m_psize + np;
            if (!marked) prof_mark(st), marked = true; // the timed region starts once the first plan exists
            HIPCHK(hipMemsetAsync(d_cnt, 0, ((size_t)BATCH * np + 2 * DF_PAD + BATCH + 16) * sizeof(int), st)); // counters, ticket
            HIPCHK(hipMemsetAsync(c.slot[S_BFLAG].p, 0, (size_t)ns * nM * nTj * nTk * sizeof(int), st));
            if (d_dbg) HIPCHK(hipMemsetAsync(d_dbg, 0, 128, st));
            hipLaunchKernelGGL(k_build_order, dim3(bp.nslots), dim3(256), 0, st, (uint2*)c.slot[S_ORDER].p, (const uint32_t*)tl->d,
                               m_poff, m_start, m_base, ns, np);
            fa.sweep_tab = m_tab, fa.plane_size = m_psize;
            fa.order = (const uint2*)c.slot[S_ORDER].p, fa.total = bp.total;
            fa.nsweeps = ns, fa.g0 = g0, fa.np = np;
            fa.plane_cnt = d_cnt, fa.planes_done = d_done, fa.ticket = d_ticket;
            fa.tile_done = (int*)c.slot[S_BFLAG].p, fa.nM = nM;
            fa.dbg = d_dbg;
            fa.timeout_ticks = FLOW_TIMEOUT_TICKS;
            if (const char* e = getenv("LSF_GS_TIMEOUT_TICKS")) fa.timeout_ticks = strtoull(e, nullptr, 10); // test hook
#ifdef LSF_EXPERIMENTS
            if (const char* e = getenv("LSF_PROBE_EARLY_FLAG")) fa.probe_early = atoi(e); // work-term style probe: wrong fields on purpose
#endif
            // one block per tile; a block takes its tile from the ticket counter, so the grid only has to be large enough
            // (2-D: gridDim.x * blockDim.x must stay below 2^32)
            const dim3 grid((unsigned)std::min<long>(fa.total, 65536), (unsigned)((fa.total + 65535) / 65536));
#define LSF_LAUNCH_DF(WY_, WZ_, BY_)                                                                             \
    do {                                                                                                         \
        if (strict) hipLaunchKernelGGL((k_reinit_gs_persist<16, WY_, WZ_, BY_, true>), grid, dim3(64 * WY_ * WZ_), 0, st, fa); \
        else hipLaunchKernelGGL((k_reinit_gs_persist<16, WY_, WZ_, BY_, false>), grid, dim3(64 * WY_ * WZ_), 0, st, fa);       \
    } while (0)
            // one block per tile (the launch with column continuation of round 4, k_reinit_gs_stream, lives on the branch
            // r04-column-continuation: bit-identical and 9-20 % slower, profiles/r04_stream_ab.txt)
#ifdef LSF_EXPERIMENTS
            if (ta == 32) {
                if (strict) hipLaunchKernelGGL((k_reinit_gs_persist<32, 2, 2, 5, true>), grid, dim3(256), 0, st, fa);
                else hipLaunchKernelGGL((k_reinit_gs_persist<32, 2, 2, 5, false>), grid, dim3(256), 0, st, fa);
            } else
#endif
            {
                LSF_SK_SHAPES(LSF_LAUNCH_DF, wy, wz, by);
            }
#undef LSF_LAUNCH_DF
            ++launches;
            if (g0 + BATCH < max_sweeps || d_dbg) { // stop flag between batches (later batches would exit at once anyway)
                HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (d_dbg) {
                    unsigned long long hd[16];
                    HIPCHK(hipMemcpy(hd, d_dbg, sizeof hd, hipMemcpyDeviceToHost));
#ifdef LSF_EXPERIMENTS
                    if (hd[7])
                        fprintf(stderr, "[lsf] tile phases (us per tile): row table %.2f, load %.2f, march %.2f, write back %.2f\n",
                                hd[3] / 100.0 / hd[7], hd[4] / 100.0 / hd[7], hd[5] / 100.0 / hd[7], hd[6] / 100.0 / hd[7]);
#endif
                    fprintf(stderr, "[lsf] dataflow batch of %d sweeps: %llu tiles, per tile: take+wait %.2f us, work+publish %.2f us\n", ns,
                            hd[2], hd[2] ? hd[0] / 100.0 / hd[2] : 0.0, hd[2] ? hd[1] / 100.0 / hd[2] : 0.0);
                }
                if (host_ctl[0]) break;
            }
        }
    }
    if (!marked) prof_mark(st);
    const bool slots_loop = !persist;
#ifdef LSF_EXPERIMENTS
    if (slots_loop && skew && getenv("LSF_TRACE_TILES")) {
        if ((rc = ws(c.slot[S_DBG], 64))) return rc;
        fa.dbg = (unsigned long long*)c.slot[S_DBG].p;
        HIPCHK(hipMemsetAsync(fa.dbg, 0, 64, st));
    }
#endif
#ifdef LSF_EXPERIMENTS
    int* stagger_tab = nullptr;
    if (slots_loop && skew && getenv("LSF_PROBE_STAGGER")) { // k_reinit_gs_skew: the second block of every CU starts late (work-term probe)
        if ((rc = ws(c.slot[S_PLANECNT], 2048 * sizeof(int)))) return rc;
        stagger_tab = (int*)c.slot[S_PLANECNT].p;
        fa.ticket = stagger_tab, fa.probe_us = atoi(getenv("LSF_PROBE_STAGGER"));
    }
#endif
    auto launch_tiles = [&](int grid, hipStream_t s_) {
#ifdef LSF_EXPERIMENTS
        if (stagger_tab) (void)hipMemsetAsync(stagger_tab, 0, 2048 * sizeof(int), s_);
#endif
        if (skew) {
#define LSF_LAUNCH_SKEW(WY_, WZ_, BY_)                                                                                     \
    do {                                                                                                                   \
        if (strict)                                                                                                        \
            hipLaunchKernelGGL((k_reinit_gs_skew<16, WY_, WZ_, BY_, true>), dim3(grid), dim3(64 * WY_ * WZ_), 0, s_, fa);  \
        else                                                                                                               \
            hipLaunchKernelGGL((k_reinit_gs_skew<16, WY_, WZ_, BY_, false>), dim3(grid), dim3(64 * WY_ * WZ_), 0, s_, fa); \
    } while (0)
#ifdef LSF_EXPERIMENTS
            if (ta == 32) {
                if (strict) hipLaunchKernelGGL((k_reinit_gs_skew<32, 2, 2, 5, true>), dim3(grid), dim3(256), 0, s_, fa);
                else hipLaunchKernelGGL((k_reinit_gs_skew<32, 2, 2, 5, false>), dim3(grid), dim3(256), 0, s_, fa);
            } else
#endif
            {
                LSF_SK_SHAPES(LSF_LAUNCH_SKEW, wy, wz, by);
            }
#undef LSF_LAUNCH_SKEW
            return;
        }
#define LSF_LAUNCH_SLOT(TA_, NY_, ST_) \
    hipLaunchKernelGGL((k_reinit_gs_box<TA_, NY_, ST_>), dim3(grid), dim3(64), 0, s_, fa)
#define LSF_LAUNCH_SLOT_NY(TA_, ST_)           \
    do {                                       \
        if (nyc == 5) LSF_LAUNCH_SLOT(TA_, 5, ST_); \
        else LSF_LAUNCH_SLOT(TA_, 4, ST_);     \
    } while (0)
        if (strict) {
            if (ta == 16) LSF_LAUNCH_SLOT_NY(16, true);
            else LSF_LAUNCH_SLOT_NY(32, true);
        } else {
            if (ta == 16) LSF_LAUNCH_SLOT_NY(16, false);
            else LSF_LAUNCH_SLOT_NY(32, false);
        }
#undef LSF_LAUNCH_SLOT_NY
#undef LSF_LAUNCH_SLOT
    };
    int lo = 0;            // first sweep that still has hyperplanes to launch
    int epilogues = 0;     // sweeps whose last hyperplane has been launched
    bool stop = false;
    for (long slot = 0; slots_loop && !stop && lo < max_sweeps; ++slot) {
        int nseg = 0, grid = 0;
#ifdef LSF_EXPERIMENTS // never in the product library: profiles/micro builds its own copy with -DLSF_EXPERIMENTS
        // timing experiment only (results are wrong): every tile of a sweep in ONE launch = the pure work term
        static const bool nodeps = getenv("LSF_GS_NODEPS_EXPERIMENT") != nullptr;
#else
        constexpr bool nodeps = false;
#endif
        for (int g = lo; g < max_sweeps && start_of(g) <= slot; ++g) {
            const long P = slot - start_of(g);
            if (P >= np) continue;
            int cnt = tl->off[P + 1] - tl->off[P];
            if (nodeps) cnt = P == 0 ? tl->off[np] : 0;
            if (cnt <= 0) continue;
            if (nseg == 4) return fail(LSF_ERR_HIP, "internal: more than four sweeps in flight");
            fa.seg_tiles[nseg] = tl->d + tl->off[P];
            grid += cnt;
            fa.seg_end[nseg] = grid;
            fa.seg_g[nseg] = g;
            for (int ax = 0; ax < 3; ++ax) fa.seg_sign[nseg][ax] = RASTER_SIGN[(first_raster + g) & 7][ax];
            ++nseg;
            if (P == np - 1) ++epilogues;
        }
        while (lo < max_sweeps && start_of(lo) + np <= slot + 1) ++lo;
        for (int q = nseg; q < 4; ++q) fa.seg_end[q] = grid;
        if (grid > 0) {
            launch_tiles(grid, st);
            ++launches;
        }
        if (epilogues >= CHECK_EVERY && lo < max_sweeps) {
            epilogues = 0;
            HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (host_ctl[0]) stop = true;
        }
    }
    prof_mark(st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
#ifdef LSF_EXPERIMENTS
    if (stagger_tab) { // did the probe tell the CUs apart?  (blocks of the LAST launch per CU)
        std::vector<int> ht(2048);
        HIPCHK(hipMemcpy(ht.data(), stagger_tab, 2048 * sizeof(int), hipMemcpyDeviceToHost));
        int used = 0, mx = 0;
        long tot = 0;
        for (int v : ht) used += v != 0, mx = std::max(mx, v), tot += v;
        fprintf(stderr, "[lsf] stagger probe: %d CU entries used, %ld blocks, most per CU %d\n", used, tot, mx);
    }
    if (fa.dbg) {
        unsigned long long hd[8];
        HIPCHK(hipMemcpy(hd, fa.dbg, sizeof hd, hipMemcpyDeviceToHost));
        if (hd[7])
            fprintf(stderr, "[lsf] tile phases (us per tile, %llu tiles): row table %.2f, load %.2f, march %.2f, write back %.2f\n", hd[7],
                    hd[3] / 100.0 / hd[7], hd[4] / 100.0 / hd[7], hd[5] / 100.0 / hd[7], hd[6] / 100.0 / hd[7]);
    }
#endif
    const int nsw = host_ctl[1];
    if (g_prof.on && g_prof.ev.size() >= 2) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, g_prof.ev[0], g_prof.ev[1]);
        g_prof.sweep_ms = ms;
        g_prof.bc_ms = g_prof.finish_ms = 0;
        g_prof.sweeps = nsw;
        g_prof.sweep_launches = launches;
        if (skew) snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "%s<%d,%d,%d,%d,%s>", !slots_loop ? "k_reinit_gs_persist" : "k_reinit_gs_skew",
                           ta, wy, wz, by, strict ? "true" : "false");
        else snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "k_reinit_gs_box<%d,%d,%s>", ta, nyc, strict ? "true" : "false");
        g_prof.kernel = g_prof.kernel_buf;
    }
    if (host_ctl[2] == 2) {
        // A block of the dataflow launch waited 4 s for a predecessor: never observed, but the launch relies on
        // nothing else going wrong with the device.  When the call's input is still around (the transposed launch never
        // touches the caller's field; otherwise phiS was copied from it on entry) repeat the call on the slot schedule,
        // whose dependencies are launch boundaries.
        if ((tr || !d_phiS_in) && g_schedule_override == -2) {
            fprintf(stderr, "[lsf] dataflow launch timed out; repeating the call with slot launches\n");
            if (!tr) HIPCHK(hipMemcpyAsync(d_phi, c.slot[S_PHIS].p, n * sizeof(double), hipMemcpyDeviceToDevice, st));
            g_schedule_override = 3;
            rc = reinit_slot_core(d_phi, d_phiS_in, nx, ny, nz, iter, dx, h, tol, mode, first_raster, sweeps_done, rms_trace,
                                  trace_cap, st);
            g_schedule_override = -2;
            return rc;
        }
        return fail(LSF_ERR_HIP, "exact-GS dataflow schedule timed out waiting for a tile");
    }
    if (tr) { // back to the caller's layout: the kernel's field has extents (ny + 1, nx + 1, nz + 1)
        const dim3 tg(cdiv(ny + 1, 32), cdiv(nx + 1, 32), (unsigned)std::min(nz + 1, 1024));
        hipLaunchKernelGGL(k_transpose_xy, tg, dim3(256), 0, st, (const double*)fa.buf[nsw % nbuf], d_phi, ny + 1, nx + 1,
                           (long)(nz + 1), (double*)nullptr);
    } else if (fa.buf[nsw % nbuf] != d_phi)
        HIPCHK(hipMemcpyAsync(d_phi, fa.buf[nsw % nbuf], n * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (rms_trace && trace_cap > 0 && nsw > 0)
        HIPCHK(hipMemcpyAsync(rms_trace, c.slot[S_TRACE].p, sizeof(double) * (size_t)std::min(nsw, trace_cap),
                              hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sweeps_done) *sweeps_done = nsw;
    if (host_ctl[2] == 1) return fail(LSF_ERR_NAN, "RMS became NaN (the reference STOPs here, subs.f90:926)");
    return LSF_OK;
}
