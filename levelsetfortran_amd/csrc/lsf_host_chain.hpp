// lsf_host_chain.hpp -- the device-resident chain behind the host seams (include/lsf.h): lsf_mirror*, lsf_snapshot, lsf_sumsq_diff and
// lsf_write_vti (set3d.f90:311, :505-516, :319-351, :538-569).  Included by lsf_api.hip inside extern "C".
#pragma once

// ---- device-resident chain (include/lsf.h) -------------------------------------------------------
int lsf_mirror(int flags)
{
    if (flags & ~(LSF_MIRROR_TRUST | LSF_MIRROR_LAZY)) return fail(LSF_ERR_INVALID, "unknown mirror flag");
    int rc = ensure_device();
    if (rc) return rc;
    Ctx& c = ctx();
    if ((c.mirror & LSF_MIRROR_LAZY) && !(flags & LSF_MIRROR_LAZY)) {
        // leaving the lazy mode: bring every stale host array up to date
        struct { Twin* t; Slot s; } all[] = {{&c.twin_phi, S_HPHI}, {&c.twin_nb, S_HNB}, {&c.twin_sb, S_HSB}, {&c.twin_snap, S_SNAP}};
        for (auto& e : all)
            if (e.t->host_stale && e.t->host) {
                HIPCHK(hipMemcpy(const_cast<void*>(e.t->host), c.slot[e.s].p, e.t->bytes, hipMemcpyDeviceToHost));
                e.t->host_stale = false;
            }
    }
    c.mirror = flags;
    return LSF_OK;
}

int lsf_mirror_sync(void* host)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!host) return fail(LSF_ERR_INVALID, "NULL pointer");
    Ctx& c = ctx();
    Twin* t = nullptr;
    Slot s = S_HPHI;
    if (!twin_of(c, host, 0, &t, &s)) return LSF_OK; // no twin: the host copy is the only one
    if (t->host_stale) {
        HIPCHK(hipMemcpy(host, c.slot[s].p, t->bytes, hipMemcpyDeviceToHost));
        t->host_stale = false;
    }
    return LSF_OK;
}

int lsf_mirror_forget(const void* host)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!host) return fail(LSF_ERR_INVALID, "NULL pointer");
    Ctx& c = ctx();
    for (Twin* t : {&c.twin_phi, &c.twin_nb, &c.twin_sb, &c.twin_snap})
        if (t->host == host) twin_drop(*t);
    return LSF_OK;
}

int lsf_snapshot(const double* phi, double* phiO, int nx, int ny, int nz)
{
    Trace trace_("lsf_snapshot");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi || !phiO) return fail(LSF_ERR_INVALID, "NULL field");
    Ctx& c = ctx();
    const size_t bytes = (size_t)(nx + 1) * (ny + 1) * (nz + 1) * sizeof(double);
    const void* d = (c.mirror & (LSF_MIRROR_TRUST | LSF_MIRROR_LAZY)) ? twin_of(c, phi, bytes) : nullptr;
    if (!d) { // no usable twin: the plain host copy of set3d.f90:311
        std::memcpy(phiO, phi, bytes);
        if (c.twin_snap.host == phiO) twin_drop(c.twin_snap); // the host copy just written is the newer one
        return LSF_OK;
    }
    if ((rc = twin_claim(c, c.twin_snap, S_SNAP, phiO, bytes))) return rc;
    HIPCHK(hipMemcpy(c.slot[S_SNAP].p, d, bytes, hipMemcpyDeviceToDevice));
    return twin_out(c, c.twin_snap, S_SNAP, phiO, bytes);
}

int lsf_sumsq_diff(const double* phi, const double* phiO, int nx, int ny, int nz, double* sum)
{
    Trace trace_("lsf_sumsq_diff");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi || !phiO || !sum) return fail(LSF_ERR_INVALID, "NULL pointer");
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if ((rc = twin_in(c, c.twin_phi, S_HPHI, phi, n * sizeof(double)))) return rc;
    if ((rc = twin_in(c, c.twin_snap, S_SNAP, phiO, n * sizeof(double)))) return rc;
    const int grid = 2048;
    if ((rc = ws(c.slot[S_PART2], (grid + 1) * sizeof(double)))) return rc;
    double* part = (double*)c.slot[S_PART2].p;
    hipLaunchKernelGGL(k_sumsq_diff, dim3(grid), dim3(256), 0, nullptr, (const double*)c.slot[S_HPHI].p,
                       (const double*)c.slot[S_SNAP].p, (long)n, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(RED_T), 0, nullptr, (const double*)part, (long)grid, part + grid);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(sum, part + grid, sizeof(double), hipMemcpyDeviceToHost));
    return LSF_OK;
}

int lsf_write_vti(const char* path, const double* phi, int nx, int ny, int nz, double dx, const double xLo[3])
{
    Trace trace_("lsf_write_vti");
    if (!path || !phi || !xLo) return fail(LSF_ERR_INVALID, "NULL pointer");
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    const size_t npts = (size_t)(nx + 1) * (ny + 1) * (nz + 1), bytes = npts * sizeof(double);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(LSF_ERR_INVALID, std::string("cannot open ") + path);
    // header: the reference's text (set3d.f90:324-345): extent '(3(A3,I6))', origin and spacing '(3(F20.8,A1))' TRIMmed
    char extent[96], origin[96], spacing[96];
    snprintf(extent, sizeof extent, " 0 %6d 0 %6d 0 %6d", nx, ny, nz);
    snprintf(origin, sizeof origin, "%20.8f %20.8f %20.8f", xLo[0], xLo[1], xLo[2]);
    snprintf(spacing, sizeof spacing, "%20.8f %20.8f %20.8f", dx, dx, dx);
    // LSF_VTI_WIDE=1 writes the 64-bit count for any size (include/lsf.h): the wide header can be exercised without a 4 GB field
    const char* wide_env = getenv("LSF_VTI_WIDE");
    const bool wide = bytes > 0xffffffffull || (wide_env && atoi(wide_env) != 0);
    fprintf(f, "<?xml version=\"1.0\"?>\n");
    fprintf(f, "<VTKFile type=\"ImageData\" version=\"0.1\" byte_order=\"LittleEndian\"%s>\n", wide ? " header_type=\"UInt64\"" : "");
    fprintf(f, "<ImageData WholeExtent=\"%s\" Origin=\"%s\" Spacing=\"%s\">\n", extent, origin, spacing);
    fprintf(f, "<Piece Extent=\"%s\">\n<PointData Scalars=\"phi\">\n", extent);
    fprintf(f, "<DataArray type=\"Float64\" Name=\"phi\" format=\"appended\" offset=\"%16d\"/>\n", 0);
    fprintf(f, "</PointData>\n</Piece>\n</ImageData>\n<AppendedData encoding=\"raw\">\n_");
    if (wide) {
        const uint64_t cnt = bytes;
        fwrite(&cnt, sizeof cnt, 1, f);
    } else {
        const uint32_t cnt = (uint32_t)bytes;
        fwrite(&cnt, sizeof cnt, 1, f);
    }
    bool ok = true;
    const void* d = nullptr;
    if (hipGetDeviceCount(&rc) == hipSuccess && rc > 0 && ensure_device() == LSF_OK) d = twin_of(ctx(), phi, bytes);
    (void)hipGetLastError();
    if (d) {
        // stream from the device twin: chunk n + 1 is copied into one pinned buffer while chunk n is written from the other
        // (buffers no larger than the payload, the second one only when there is a second chunk: pinning 2 x 64 MB for the 1.9 MB of
        // the reference's default grid was most of this call's 40-55 ms there)
        const size_t CH = std::min<size_t>(64u << 20, (bytes + 4095) & ~(size_t)4095);
        const size_t nch = (bytes + CH - 1) / CH;
        void* pin[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        if (hipHostMalloc(&pin[0], CH, hipHostMallocDefault) != hipSuccess ||
            (nch > 1 && hipHostMalloc(&pin[1], CH, hipHostMallocDefault) != hipSuccess) ||
            hipStreamCreate(&st) != hipSuccess || hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
            ok = false;
        } else {
            auto issue = [&](size_t q) {
                const size_t off = q * CH, len = std::min(CH, bytes - off);
                return hipMemcpyAsync(pin[q & 1], (const char*)d + off, len, hipMemcpyDeviceToHost, st) == hipSuccess &&
                       hipEventRecord(ev[q & 1], st) == hipSuccess;
            };
            ok = issue(0);
            for (size_t q = 0; q < nch && ok; ++q) {
                if (q + 1 < nch) ok = issue(q + 1);
                ok = ok && hipEventSynchronize(ev[q & 1]) == hipSuccess;
                const size_t len = std::min(CH, bytes - q * CH);
                ok = ok && fwrite(pin[q & 1], 1, len, f) == len;
            }
        }
        if (st) (void)hipStreamSynchronize(st);
        for (int q = 0; q < 2; ++q) {
            if (ev[q]) (void)hipEventDestroy(ev[q]);
            if (pin[q]) (void)hipHostFree(pin[q]);
        }
        if (st) (void)hipStreamDestroy(st);
    } else {
        ok = fwrite(phi, 1, bytes, f) == bytes;
    }
    fprintf(f, "\n</AppendedData>\n</VTKFile>\n");
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(LSF_ERR_HIP, std::string("writing ") + path + " failed");
    return LSF_OK;
}
