// lsf_gs_slabs.hpp -- host side of the EXACT ordering across z slabs (kernel: k_reinit_gs_slab, lsf_skew.hpp).
// Included by lsf_api.hip (uses its helpers: fail, HIPCHK, gs_skew_w, get_sk_tables, skew_spacing, RASTER_SIGN ...).
//
// The reference's in-place Gauss-Seidel sweep (subs.f90:743-852) is one dependency graph over the whole grid; the dataflow
// launch of reinit_slot_core runs it tile by tile on one device.  Here the same graph -- same tiles, same hyperplanes, same
// spacing of consecutive sweeps, same RMS summation order -- is executed by one launch per slab of tile columns in z, each on
// its own device (or several on one device: the rehearsal of a one-GPU machine), so the field is the single-device field,
// hence the reference's, bit for bit.  A hyperplane P = m + B + C of the tile graph holds tiles of every slab, so the slabs
// work side by side from the first hyperplanes on; what crosses a cut is three planes of results per sweep and side, stored
// by the producing tile straight into the neighbour's field buffer (xGMI peer stores on a real node), a flag per tile next
// to the cut, a hyperplane counter per sweep, and the sweep epilogue's verdict.
//
// Memory: a slab allocates its own planes plus three beyond each cut, addressed through the GLOBAL address map (the kernel
// gets base pointers shifted by the slab's first plane), so field memory scales with 1 / slabs; the tile flags and the per-
// column RMS sums keep their global size (small).
#pragma once

namespace lsfs {

struct Slab {
    int device = 0, tk_lo = 0, tk_hi = 0;
    int ka = 0, kb = 0;              // planes held: ka .. kb inclusive
    hipStream_t st = nullptr;
    double* fld[4] = {nullptr, nullptr, nullptr, nullptr}; // allocations (plane ka first)
    double* phiS = nullptr;
    double* stage = nullptr;         // the slab's planes in the caller's layout (upload / download, x <-> y transposition)
    int* ctlblk = nullptr;           // ctl[16] | verdict[DF_BATCH] | planes_done: own, mirror of lower, mirror of upper [3][DF_BATCH]
    int* cnt = nullptr;              // plane_cnt[DF_BATCH * np] | ticket[16]
    int* tile_done = nullptr;
    double* colsum = nullptr;
    double* trace = nullptr;
    uint32_t* tiles[2] = {nullptr, nullptr}; // this slab's tiles, hyperplane by hyperplane: sweeps along z / against z
    std::vector<int> off[2];         // offsets of the hyperplanes in tiles[]
    uint2* order = nullptr;
    const uint32_t* tables = nullptr;
    std::map<std::pair<int, int>, std::pair<int*, long>> meta; // (phase, ns) -> device meta block, entries of the task list
    std::vector<void*> owned;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct Slabs {
    std::vector<Slab> s;
    ~Slabs()
    {
        for (Slab& b : s) {
            if (hipSetDevice(b.device) != hipSuccess) continue;
            if (b.st) (void)hipStreamSynchronize(b.st);
            for (void* p : b.owned) (void)hipFree(p);
            if (b.ev0) (void)hipEventDestroy(b.ev0);
            if (b.ev1) (void)hipEventDestroy(b.ev1);
            if (b.st) (void)hipStreamDestroy(b.st);
        }
        (void)hipGetLastError();
    }
};

// memory another device stores into while a kernel of this one polls it: fine-grained where the two are different devices
inline int slab_alloc(Slab& b, void** p, size_t bytes, bool fine)
{
    *p = nullptr;
    if (fine) HIPCHK(hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained));
    else HIPCHK(hipMalloc(p, bytes));
    b.owned.push_back(*p);
    return LSF_OK;
}

// last run, for lsf_profile-style reporting (bench.py)
struct SlabReport {
    int slabs = 0, sweeps = 0, grid = 0, fine = 0;
    double kernel_ms = 0; // longest of the slabs' launches, batches summed
};
thread_local SlabReport g_slab_report;

} // namespace lsfs

#ifdef LSF_EXPERIMENTS
// LSF_SLAB_MODEL="d/D" (experiment builds only; the field comes out WRONG): ONE slab of a D-slab run, alone on its device, with
// everything it would wait for from its neighbours granted in advance -- the flags of every tile outside its columns and the
// verdict of every sweep.  Its launch then takes the time a device of a D-device run needs for its share of the tile graph
// when communication is free: the throughput term of the scaling model (profiles/micro/slab_model.sh).
static __global__ __launch_bounds__(256) void k_slab_model_preset(int* tile_done, int* verdict, const int* sweep_tab, int ns, int nM, int nTj, int nTk,
                                                                  int tk_lo, int tk_hi)
{
    const long per_sweep = (long)nM * nTj * nTk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)ns * per_sweep; i += (long)gridDim.x * 256) {
        const int s = (int)(i / per_sweep);
        const int C = (int)((i - (long)s * per_sweep) / ((long)nM * nTj));
        const int tk = sweep_tab[4 * s + 2] > 0 ? C : nTk - 1 - C;
        if (tk < tk_lo || tk >= tk_hi) tile_done[i] = 1;
    }
    if (blockIdx.x == 0)
        for (int q = threadIdx.x; q < ns; q += 256) verdict[q] = 1;
}
#endif

// phi: HOST array in the caller's layout.  devices[0 .. ndev): one slab each, bottom (k = 0) to top.
int reinit_gs_slabs(double* phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode, const int* devices, int ndev,
                    int* sweeps_done, double* rms_trace, int trace_cap)
{
    using namespace lsfs;
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    if (ndev < 1 || ndev > 8) return fail(LSF_ERR_INVALID, "exact ordering across slabs: 1 to 8 devices");
    if (nx < 3 || ny < 3 || nz < 3) return fail(LSF_ERR_INVALID, "exact ordering across slabs: at least two interior cells per axis");
    const bool strict = (mode & LSF_ARITH_STRICT) != 0;
    const int max_sweeps = iter + 1, ta = 16;
    const bool tr = gs_march() == 1;
    const int nbuf = 4;
    int wy = 2, wz = 2, by = 5;
    gs_skew_w(std::min(nx, ny), nz, strict, &wy, &wz, &by);
    const int nyc = by * wy, nzc = 4 * wz;
    const int knx = tr ? ny : nx, kny = tr ? nx : ny;
    const int nTi = cdiv(knx - 1, ta), nTj = cdiv(kny - 1, nyc), nTk = cdiv(nz - 1, nzc);
    if (nTk < ndev) return fail(LSF_ERR_INVALID, "exact ordering across slabs: fewer tile layers in z than devices");
    const int m_max = (knx - 2 + nyc * nTj - 1 + nzc * nTk - 1) / ta;
    if (m_max > 1023 || nTj > 1023 || nTk > 1023) return fail(LSF_ERR_INVALID, "grid too large for tile index packing");
    const int nM = m_max + 1, np = m_max + nTj + nTk - 1;
    const long per_sweep = (long)nM * nTj * nTk;
    const int ncol = nTj * nTk;
    const size_t pl = (size_t)(nx + 1) * (ny + 1);
    auto ksign = [&](int raster, int* out) {
        const int* r = RASTER_SIGN[raster & 7];
        out[0] = tr ? r[1] : r[0], out[1] = tr ? r[0] : r[1], out[2] = r[2];
    };
    auto m_lo = [&](int B, int C) { return (nyc * B + nzc * C) / ta; };
    auto m_hi = [&](int B, int C) { return (nyc * B + nyc - 1 + nzc * C + nzc - 1 + knx - 2) / ta; };
    const uint32_t last_packed = (uint32_t)m_hi(nTj - 1, nTk - 1) | ((uint32_t)(nTj - 1) << 10) | ((uint32_t)(nTk - 1) << 20);

    int mod_d = -1, mod_D = 0; // LSF_SLAB_MODEL (experiment builds): this call's one slab is slab mod_d of mod_D
#ifdef LSF_EXPERIMENTS
    if (const char* e = getenv("LSF_SLAB_MODEL")) {
        if (ndev != 1 || sscanf(e, "%d/%d", &mod_d, &mod_D) != 2 || mod_d < 0 || mod_d >= mod_D || mod_D > nTk)
            return fail(LSF_ERR_INVALID, "LSF_SLAB_MODEL=d/D needs one device and 0 <= d < D <= tile layers in z");
    }
#endif
    const bool model = mod_D > 0;
    DeviceRestore restore_;
    Slabs S;
    S.s.resize(ndev);
    // fine-grained memory where a neighbour is another device (LSF_SLAB_FINEGRAINED = 0 / 1 forces it off / on)
    bool distinct = false;
    for (int d = 1; d < ndev; ++d) distinct = distinct || devices[d] != devices[0];
    bool fine = distinct;
    if (const char* e = getenv("LSF_SLAB_FINEGRAINED")) fine = atoi(e) != 0;
    // every pair: the cut planes go to the neighbours only, but the sweep epilogue stores its verdict on every slab
    for (int a = 0; a < ndev; ++a)
        for (int b = 0; b < ndev; ++b) {
            const int from = devices[a], to = devices[b];
            if (from == to) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can)
                return fail(LSF_ERR_INVALID, "exact ordering across slabs: the devices cannot access each other's memory");
            HIPCHK(hipSetDevice(from));
            const hipError_t pe = hipDeviceEnablePeerAccess(to, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(LSF_ERR_HIP, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }

    // First contact: the hand-offs below rely on four properties of peer memory that a one-GPU machine cannot show (DESIGN.md
    // section 6.1).  Every pair of distinct devices runs the litmus test once per process before the first launch that needs it;
    // a violated assumption ends the call with an error that names it -- not with a 4 s time-out or a field that differs.
    {
        static std::mutex mu;
        static std::set<std::pair<int, int>> passed;
        std::lock_guard<std::mutex> lk(mu);
        for (int a = 0; a + 1 < ndev; ++a) {
            const int da = std::min(devices[a], devices[a + 1]), db = std::max(devices[a], devices[a + 1]);
            if (da == db || passed.count({da, db}) || getenv("LSF_SLAB_NO_SELFTEST")) continue;
            int violated = 0;
            if ((rc = lsf_peer_selftest(da, db, &violated))) return rc;
            passed.insert({da, db});
        }
    }

    // ---- geometry, buffers, tile lists of every slab -----------------------------------------------------------------
    std::map<int, int> share; // slabs per device
    for (int d = 0; d < ndev; ++d) ++share[devices[d]];
    // Launches that wait for each other must run at the same time.  Streams of one device share its hardware queues (four
    // unless GPU_MAX_HW_QUEUES says otherwise) and two launches on one queue run one after the other, so a device takes at
    // most three slabs (the fourth queue is left to whatever else the process runs) -- a limit of the one-GPU rehearsal only.
    {
        int queues = 4;
        if (const char* e = getenv("GPU_MAX_HW_QUEUES")) queues = std::max(1, atoi(e));
        for (auto& kv : share)
            if (kv.second > 1 && kv.second > queues - 1)
                return fail(LSF_ERR_INVALID, "exact ordering across slabs: " + std::to_string(kv.second) + " slabs on device " +
                                                 std::to_string(kv.first) + " need GPU_MAX_HW_QUEUES >= " + std::to_string(kv.second + 1) +
                                                 " in the environment (their launches must run concurrently)");
    }
    for (int d = 0; d < ndev; ++d) {
        Slab& b = S.s[d];
        b.device = devices[d];
        const int gd_ = model ? mod_d : d, gD_ = model ? mod_D : ndev; // this slab's place among the slabs of the run
        b.tk_lo = (int)((long)nTk * gd_ / gD_), b.tk_hi = (int)((long)nTk * (gd_ + 1) / gD_);
        const int k_own_lo = 1 + b.tk_lo * nzc, k_own_hi = std::min(1 + b.tk_hi * nzc, nz); // interior planes [lo, hi)
        b.ka = gd_ == 0 ? 0 : k_own_lo - 3, b.kb = gd_ == gD_ - 1 ? nz : std::min(k_own_hi + 2, nz);
        g_device = b.device;
        if ((rc = ensure_device())) return rc;
        HIPCHK(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&b.ev0));
        HIPCHK(hipEventCreate(&b.ev1));
        const size_t planes = (size_t)(b.kb - b.ka + 1), bytes = planes * pl * sizeof(double);
        for (int q = 0; q < nbuf; ++q)
            if ((rc = slab_alloc(b, (void**)&b.fld[q], bytes, fine))) return rc;
        if ((rc = slab_alloc(b, (void**)&b.phiS, bytes, false))) return rc;
        if ((rc = slab_alloc(b, (void**)&b.stage, bytes, false))) return rc;
        if ((rc = slab_alloc(b, (void**)&b.ctlblk, (16 + 4 * DF_BATCH) * sizeof(int), fine))) return rc;
        if ((rc = slab_alloc(b, (void**)&b.cnt, ((size_t)DF_BATCH * np + 16) * sizeof(int), false))) return rc;
        const int BATCHcap = std::min(DF_BATCH, max_sweeps);
        if ((rc = slab_alloc(b, (void**)&b.tile_done, (size_t)BATCHcap * per_sweep * sizeof(int), fine))) return rc;
        if ((rc = slab_alloc(b, (void**)&b.colsum, (size_t)4 * ncol * sizeof(double), fine))) return rc;
        HIPCHK(hipMemset(b.colsum, 0, (size_t)4 * ncol * sizeof(double)));
        if ((rc = slab_alloc(b, (void**)&b.trace, (size_t)max_sweeps * sizeof(double), fine))) return rc;
        if ((rc = get_sk_tables(16, wy, wz, by, &b.tables))) return rc;
        // this slab's tiles per hyperplane: frame C runs along z in sweeps with sk > 0 (C = tk), against it otherwise
        long ntiles = 0;
        for (int v = 0; v < 2; ++v) {
            const int c_lo = v == 0 ? b.tk_lo : nTk - b.tk_hi, c_hi = v == 0 ? b.tk_hi : nTk - b.tk_lo;
            std::vector<int> cntp(np + 1, 0);
            for (int C = c_lo; C < c_hi; ++C)
                for (int B = 0; B < nTj; ++B)
                    for (int m = m_lo(B, C); m <= m_hi(B, C); ++m) ++cntp[m + B + C];
            b.off[v].assign(np + 1, 0);
            for (int P = 0; P < np; ++P) b.off[v][P + 1] = b.off[v][P] + cntp[P];
            std::vector<uint32_t> hbuf((size_t)b.off[v][np]);
            std::vector<int> fill(b.off[v].begin(), b.off[v].end() - 1);
            for (int C = c_lo; C < c_hi; ++C)
                for (int B = 0; B < nTj; ++B)
                    for (int m = m_lo(B, C); m <= m_hi(B, C); ++m)
                        hbuf[(size_t)fill[m + B + C]++] = (uint32_t)m | ((uint32_t)B << 10) | ((uint32_t)C << 20);
            if ((rc = slab_alloc(b, (void**)&b.tiles[v], std::max<size_t>(hbuf.size(), 1) * sizeof(uint32_t), false))) return rc;
            HIPCHK(hipMemcpy(b.tiles[v], hbuf.data(), hbuf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            ntiles = std::max<long>(ntiles, (long)hbuf.size());
        }
        if ((rc = slab_alloc(b, (void**)&b.order, (size_t)BATCHcap * ntiles * sizeof(uint2), false))) return rc;
        // upload the slab's planes; x <-> y transposition into buffer 0 and phiS (phiS = phi on entry, subs.f90:731)
        HIPCHK(hipMemcpyAsync(tr ? b.stage : b.fld[0], phi + (size_t)b.ka * pl, bytes, hipMemcpyHostToDevice, b.st));
        if (tr) {
            const dim3 tg(cdiv(nx + 1, 32), cdiv(ny + 1, 32), (unsigned)std::min<size_t>(planes, 1024));
            hipLaunchKernelGGL(k_transpose_xy, tg, dim3(256), 0, b.st, (const double*)b.stage, b.fld[0], nx + 1, ny + 1, (long)planes, b.phiS);
        } else {
            HIPCHK(hipMemcpyAsync(b.phiS, b.fld[0], bytes, hipMemcpyDeviceToDevice, b.st));
        }
        int init[16] = {0, 0, 0, 0, INT_MAX, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(b.ctlblk, init, sizeof init, hipMemcpyHostToDevice, b.st));
        HIPCHK(hipStreamSynchronize(b.st)); // `init` is a stack array
    }

    // ---- kernel arguments ---------------------------------------------------------------------------------------------
    std::vector<GsArgs> fa(ndev);
    for (int d = 0; d < ndev; ++d) {
        Slab& b = S.s[d];
        GsArgs& a = fa[d];
        std::memset(&a, 0, sizeof a);
        SlabPeers pe;
        std::memset(&pe, 0, sizeof pe);
        auto base = [&](const Slab& x, double* p) { return p - (size_t)x.ka * pl; }; // global address map: plane 0 of the grid
        for (int q = 0; q < nbuf; ++q) a.buf[q] = base(b, b.fld[q]);
        a.nbuf = nbuf, a.quirk_axis = tr ? 0 : 1;
        a.phiS = base(b, b.phiS);
        a.nx = knx, a.ny = kny, a.nz = nz, a.nTi = nTi, a.nTj = nTj, a.nTk = nTk;
        a.dx = dx, a.h = h;
        a.colsum = b.colsum, a.trace = b.trace, a.trace_cap = max_sweeps;
        a.den = rms_denominator(nx, ny, nz), a.tol = tol;
        a.ctl = b.ctlblk;
        a.nTiles = (long)nTi * nTj * nTk;
        a.last_packed = last_packed;
        a.tables = b.tables;
        a.np = np, a.nM = nM;
        a.tile_done = b.tile_done;
        a.plane_cnt = b.cnt, a.ticket = b.cnt + (size_t)DF_BATCH * np;
        a.verdict = b.ctlblk + 16;
        a.planes_done = b.ctlblk + 16 + DF_BATCH;
        a.slab = d, a.nslab = ndev, a.tk_lo = b.tk_lo, a.tk_hi = b.tk_hi;
        for (int side = 0; side < 2; ++side) {
            const int nbr = side == 0 ? d - 1 : d + 1;
            a.pd_of_nb[side] = b.ctlblk + 4; // INT_MAX: no neighbour, always passes
            if (nbr < 0 || nbr >= ndev) continue;
            Slab& n = S.s[nbr];
            for (int q = 0; q < nbuf; ++q) pe.nb_buf[side][q] = base(n, n.fld[q]);
            a.nb_tile_done[side] = n.tile_done;
            // the neighbour's mirror rows: [1] follows ITS lower neighbour, [2] its upper one
            a.nb_pd[side] = n.ctlblk + 16 + DF_BATCH + (side == 0 ? 2 : 1) * DF_BATCH;
            a.pd_of_nb[side] = b.ctlblk + 16 + DF_BATCH + (side == 0 ? 1 : 2) * DF_BATCH;
        }
        for (int q = 0; q < ndev; ++q) {
            pe.all_ctl[q] = S.s[q].ctlblk, pe.all_verdict[q] = S.s[q].ctlblk + 16;
            pe.all_trace[q] = S.s[q].trace, pe.all_colsum[q] = S.s[q].colsum;
        }
        HIPCHK(hipSetDevice(b.device));
        SlabPeers* d_pe = nullptr;
        if ((rc = slab_alloc(b, (void**)&d_pe, sizeof pe, false))) return rc;
        HIPCHK(hipMemcpy(d_pe, &pe, sizeof pe, hipMemcpyHostToDevice));
        a.peers = d_pe;
        a.timeout_ticks = FLOW_TIMEOUT_TICKS;
        if (const char* e = getenv("LSF_GS_TIMEOUT_TICKS")) a.timeout_ticks = strtoull(e, nullptr, 10);
    }

    // ---- batches of up to DF_BATCH sweeps: one launch per slab and batch -------------------------------------------------
    long max_tiles = 0;
    for (Slab& b : S.s) max_tiles = std::max<long>(max_tiles, std::max(b.off[0][np], b.off[1][np]));
    int BATCH = (int)std::max<long>(8, std::min<long>(max_sweeps <= 64 ? 64 : DF_BATCH, (512L << 20) / (std::max<long>(max_tiles, 1) * 8) / 8 * 8));
    if (const char* e = getenv("LSF_DF_BATCH")) BATCH = std::max(8, std::min(BATCH, atoi(e) / 8 * 8)); // test hook: batch boundaries
    int host_ctl[4] = {0, 0, 0, 0};
    std::vector<double> kernel_ms(ndev, 0.0);
    int grid_used = 0;
    for (int g0 = 0; g0 < max_sweeps; g0 += BATCH) {
        const int ns = std::min(BATCH, max_sweeps - g0), phase = g0 & 7;
        // start slots of the batch (global: they only fix the order of the task lists), per-sweep table
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
            if (q >= nbuf) s0 = std::max<long>(s0, st0[q - nbuf] + np + 1);
            st0[q] = (int)s0;
        }
        const int nslots = st0[ns - 1] + np;
        // A slab's hyperplane counter counts its LEADING complete hyperplanes, and those in front of its first tile are complete
        // from the start: the counter (and its mirrors on the neighbours) starts there.  Left at 0 it only moved when the slab's
        // first tile was done, and until then the neighbour's tiles of the NEXT sweep waited for hyperplanes that hold nothing --
        // a needless stall between devices, and with slabs that share a device (a ticket loop with a fixed number of blocks) a
        // deadlock as soon as more such tiles precede the neighbour's first tile in the list than there are blocks (found by the
        // soak: 154 x 186 x 239 points, single-wavefront tiles, two slabs).
        std::vector<std::vector<int>> pd_init((size_t)ndev, std::vector<int>((size_t)3 * DF_BATCH, 0));
        {
            auto first_plane = [&](const Slab& x, int v) {
                int P = 0;
                while (P < np && x.off[v][P + 1] == x.off[v][P]) ++P;
                return P;
            };
            for (int d = 0; d < ndev; ++d)
                for (int q = 0; q < ns; ++q) {
                    const int v = tab[4 * q + 2] > 0 ? 0 : 1;
                    pd_init[d][q] = first_plane(S.s[d], v);
                    if (!model && d > 0) pd_init[d][DF_BATCH + q] = first_plane(S.s[d - 1], v);
                    if (!model && d + 1 < ndev) pd_init[d][2 * DF_BATCH + q] = first_plane(S.s[d + 1], v);
                }
        }
        for (int d = 0; d < ndev; ++d) {
            Slab& b = S.s[d];
            HIPCHK(hipSetDevice(b.device));
            auto it = b.meta.find({phase, ns});
            if (it == b.meta.end()) {
                std::vector<unsigned> basev((size_t)nslots + 1, 0u);
                int lo_s = 0;
                for (int slot = 0; slot < nslots; ++slot) {
                    unsigned c_ = 0;
                    for (int q = lo_s; q < ns && st0[q] <= slot; ++q) {
                        const int P = slot - st0[q], v = tab[4 * q + 2] > 0 ? 0 : 1;
                        if (P < np) c_ += (unsigned)(b.off[v][P + 1] - b.off[v][P]);
                    }
                    basev[slot + 1] = basev[slot] + c_;
                    while (lo_s < ns && st0[lo_s] + np <= slot + 1) ++lo_s;
                }
                std::vector<int> meta;
                meta.insert(meta.end(), st0.begin(), st0.end());
                for (unsigned v : basev) meta.push_back((int)v);
                while (meta.size() % 4) meta.push_back(0); // the kernels read a sweep's four table entries as one 16-byte load
                meta.insert(meta.end(), tab.begin(), tab.end());
                for (int v = 0; v < 2; ++v)
                    for (int P = 0; P < np; ++P) meta.push_back(b.off[v][P + 1] - b.off[v][P]);
                for (int v = 0; v < 2; ++v) meta.insert(meta.end(), b.off[v].begin(), b.off[v].end());
                int* dm = nullptr;
                if ((rc = slab_alloc(b, (void**)&dm, meta.size() * sizeof(int), false))) return rc;
                HIPCHK(hipMemcpy(dm, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice));
                it = b.meta.emplace(std::make_pair(phase, ns), std::make_pair(dm, (long)basev[nslots])).first;
            }
            const int* dm = it->second.first;
            const int* m_start = dm;
            const unsigned* m_base = (const unsigned*)(dm + ns);
            const int* m_tab = dm + (ns + nslots + 1 + 3) / 4 * 4; // 16-byte aligned (hipMalloc aligns the block)
            const int* m_psize = m_tab + 4 * DF_BATCH;
            const int* m_poff = m_psize + 2 * np;
            GsArgs& a = fa[d];
            a.sweep_tab = m_tab, a.plane_size = m_psize, a.plane_size_neg = m_psize + np;
            a.order = b.order, a.total = it->second.second;
            a.nsweeps = ns, a.g0 = g0;
            // every slab's control words of the batch are clear before any slab starts (the neighbours store into them)
            HIPCHK(hipMemsetAsync(b.ctlblk + 16, 0, (size_t)4 * DF_BATCH * sizeof(int), b.st));
            HIPCHK(hipMemcpyAsync(b.ctlblk + 16 + DF_BATCH, pd_init[d].data(), (size_t)3 * DF_BATCH * sizeof(int), hipMemcpyHostToDevice, b.st));
            HIPCHK(hipMemsetAsync(b.cnt, 0, ((size_t)DF_BATCH * np + 16) * sizeof(int), b.st));
            HIPCHK(hipMemsetAsync(b.tile_done, 0, (size_t)ns * per_sweep * sizeof(int), b.st));
            hipLaunchKernelGGL(k_build_order_slab, dim3(nslots), dim3(256), 0, b.st, b.order, (const uint32_t*)b.tiles[0], (const uint32_t*)b.tiles[1],
                               m_poff, m_poff + np + 1, m_tab, m_start, m_base, ns, np);
#ifdef LSF_EXPERIMENTS
            if (model) hipLaunchKernelGGL(k_slab_model_preset, dim3(1024), dim3(256), 0, b.st, b.tile_done, b.ctlblk + 16, m_tab, ns, nM, nTj, nTk, b.tk_lo, b.tk_hi);
#endif
        }
        for (Slab& b : S.s) {
            HIPCHK(hipSetDevice(b.device));
            HIPCHK(hipStreamSynchronize(b.st));
        }
        for (int d = 0; d < ndev; ++d) {
            Slab& b = S.s[d];
            HIPCHK(hipSetDevice(b.device));
            GsArgs& a = fa[d];
            int grid = 0;
            // slabs that share a device: every launch must be resident as a whole (its blocks wait for tiles of the other
            // launches): a loop over tickets, the device's capacity for the kernel divided by the slabs on it.  A slab that has
            // its device to itself: one block per tile (k_reinit_gs_slab).  LSF_SLAB_LOOP = 0 / 1 overrides (0 only where no
            // device is shared).
            bool loop = share[b.device] > 1;
            if (const char* e = getenv("LSF_SLAB_LOOP")) loop = loop || atoi(e) != 0;
#define LSF_LAUNCH_SLAB_AS(WY_, WZ_, BY_, ST_, LP_)                                                                            \
    do {                                                                                                                      \
        dim3 gd(1);                                                                                                            \
        if (LP_) {                                                                                                             \
            int per_cu = 0, cus = 0;                                                                                           \
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_reinit_gs_slab<16, WY_, WZ_, BY_, ST_, LP_>, 64 * WY_ * WZ_, 0)); \
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b.device));                              \
            grid = share[b.device] > 1 ? std::max(1, per_cu * cus * 9 / 10 / share[b.device]) : std::max(1, per_cu * cus);     \
            if (const char* e = getenv("LSF_SLAB_GRID")) grid = std::max(1, atoi(e));                                          \
            grid = (int)std::min<long>(grid, std::max<long>(a.total, 1));                                                      \
            gd = dim3((unsigned)grid);                                                                                         \
        } else { /* one block per tile (2-D: gridDim.x * blockDim.x must stay below 2^32) */                                   \
            grid = (int)std::min<long>(a.total, INT_MAX);                                                                      \
            gd = dim3((unsigned)std::min<long>(std::max<long>(a.total, 1), 65536), (unsigned)((std::max<long>(a.total, 1) + 65535) / 65536)); \
        }                                                                                                                      \
        HIPCHK(hipEventRecord(b.ev0, b.st));                                                                                   \
        hipLaunchKernelGGL((k_reinit_gs_slab<16, WY_, WZ_, BY_, ST_, LP_>), gd, dim3(64 * WY_ * WZ_), 0, b.st, a);             \
        HIPCHK(hipEventRecord(b.ev1, b.st));                                                                                   \
    } while (0)
#define LSF_LAUNCH_SLAB(WY_, WZ_, BY_)                                                       \
    do {                                                                                     \
        if (strict && loop) LSF_LAUNCH_SLAB_AS(WY_, WZ_, BY_, true, true);                   \
        else if (strict) LSF_LAUNCH_SLAB_AS(WY_, WZ_, BY_, true, false);                     \
        else if (loop) LSF_LAUNCH_SLAB_AS(WY_, WZ_, BY_, false, true);                       \
        else LSF_LAUNCH_SLAB_AS(WY_, WZ_, BY_, false, false);                                \
    } while (0)
#ifdef LSF_EXPERIMENTS
            if (ndev == 1 && getenv("LSF_TRACE_TILES")) { // per-tile phase times (skew_tile, LSF_PHASE)
                unsigned long long* dd = nullptr;
                if ((rc = slab_alloc(b, (void**)&dd, 64, false))) return rc;
                HIPCHK(hipMemsetAsync(dd, 0, 64, b.st));
                a.dbg = dd;
            }
#endif
            LSF_SK_SHAPES(LSF_LAUNCH_SLAB, wy, wz, by);
#undef LSF_LAUNCH_SLAB
#undef LSF_LAUNCH_SLAB_AS
            grid_used = grid;
            HIPCHK(hipGetLastError());
        }
        for (int d = 0; d < ndev; ++d) {
            Slab& b = S.s[d];
            HIPCHK(hipSetDevice(b.device));
            HIPCHK(hipStreamSynchronize(b.st));
            float ms = 0;
            if (hipEventElapsedTime(&ms, b.ev0, b.ev1) == hipSuccess) kernel_ms[d] += ms;
        }
#ifdef LSF_EXPERIMENTS
        if (fa[0].dbg) {
            unsigned long long hd[8];
            HIPCHK(hipSetDevice(S.s[0].device));
            HIPCHK(hipMemcpy(hd, fa[0].dbg, sizeof hd, hipMemcpyDeviceToHost));
            if (hd[7])
                fprintf(stderr, "[lsf] tile phases (us per tile, %llu tiles): row table %.2f, load %.2f, march %.2f, write back %.2f\n", hd[7],
                        hd[3] / 100.0 / hd[7], hd[4] / 100.0 / hd[7], hd[5] / 100.0 / hd[7], hd[6] / 100.0 / hd[7]);
        }
#endif
        // slab 0's copy of the control words (every slab holds the same: the epilogues store them everywhere)
        HIPCHK(hipSetDevice(S.s[0].device));
        HIPCHK(hipMemcpy(host_ctl, S.s[0].ctlblk, sizeof host_ctl, hipMemcpyDeviceToHost));
        if (host_ctl[0]) break;
    }
    const int nsw = host_ctl[1];
    g_slab_report.slabs = ndev, g_slab_report.sweeps = nsw, g_slab_report.grid = grid_used, g_slab_report.fine = (int)fine;
    g_slab_report.kernel_ms = *std::max_element(kernel_ms.begin(), kernel_ms.end());
    if (host_ctl[2] == 2) {
        // what the first tile to give up on each slab was waiting for (k_reinit_gs_slab, time_out)
        std::string what;
        for (int d = 0; d < ndev; ++d) {
            int w[16] = {0};
            if (hipSetDevice(S.s[d].device) != hipSuccess || hipMemcpy(w, S.s[d].ctlblk, sizeof w, hipMemcpyDeviceToHost) != hipSuccess || !w[8]) continue;
            char buf[320];
            const unsigned pk = (unsigned)w[9];
            const int s_ = w[10] & (DF_BATCH - 1), P_ = (int)((unsigned)w[10] >> DF_SWEEP_BITS);
            if (w[8] == 1)
                snprintf(buf, sizeof buf, "; slab %d: tile (m %u, B %u, C %u) of sweep %d, hyperplane %d, waited for sweep %d to pass hyperplane %d: own counter %d, "
                         "lower neighbour's %d, upper neighbour's %d, verdict word %d", d, pk & 0x3ff, (pk >> 10) & 0x3ff, (pk >> 20) & 0x3ff, s_, P_, s_ - 1, w[11],
                         w[12], w[13], w[14], w[15]);
            else
                snprintf(buf, sizeof buf, "; slab %d: tile (m %u, B %u, C %u) of sweep %d, hyperplane %d, waited for its upstream tiles: flags along x %d, y %d, z %d",
                         d, pk & 0x3ff, (pk >> 10) & 0x3ff, (pk >> 20) & 0x3ff, s_, P_, w[11], w[12], w[13]);
            what += buf;
        }
        (void)hipGetLastError();
        return fail(LSF_ERR_HIP, "exact ordering across slabs: a tile waited longer than the time-out for a predecessor" + what);
    }

    // ---- result: every slab returns the planes it owns (walls k = 0 and k = nz with the first / last slab) ----------------
    for (int d = 0; d < ndev; ++d) {
        Slab& b = S.s[d];
        HIPCHK(hipSetDevice(b.device));
        const int k_own_lo = 1 + b.tk_lo * nzc, k_own_hi = std::min(1 + b.tk_hi * nzc, nz);
        const int gd_ = model ? mod_d : d, gD_ = model ? mod_D : ndev;
        const int g_lo = gd_ == 0 ? 0 : k_own_lo, g_hi = gd_ == gD_ - 1 ? nz : k_own_hi - 1; // inclusive
        const size_t planes = (size_t)(g_hi - g_lo + 1);
        const double* res = b.fld[nsw % nbuf] + (size_t)(g_lo - b.ka) * pl;
        if (tr) {
            const dim3 tg(cdiv(ny + 1, 32), cdiv(nx + 1, 32), (unsigned)std::min<size_t>(planes, 1024));
            hipLaunchKernelGGL(k_transpose_xy, tg, dim3(256), 0, b.st, res, b.stage, ny + 1, nx + 1, (long)planes, (double*)nullptr);
            res = b.stage;
        }
        HIPCHK(hipMemcpyAsync(phi + (size_t)g_lo * pl, res, planes * pl * sizeof(double), hipMemcpyDeviceToHost, b.st));
    }
    for (Slab& b : S.s) {
        HIPCHK(hipSetDevice(b.device));
        HIPCHK(hipStreamSynchronize(b.st));
    }
    if (rms_trace && trace_cap > 0 && nsw > 0) {
        HIPCHK(hipSetDevice(S.s[0].device));
        HIPCHK(hipMemcpy(rms_trace, S.s[0].trace, sizeof(double) * (size_t)std::min(nsw, trace_cap), hipMemcpyDeviceToHost));
    }
    if (sweeps_done) *sweeps_done = nsw;
    if (host_ctl[2] == 1) return fail(LSF_ERR_NAN, "RMS became NaN (the reference STOPs here, subs.f90:926)");
    return LSF_OK;
}
