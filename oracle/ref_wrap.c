/*
 * ref_wrap.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Link-time interposer used only when building oracle/_ref/set3d_ref.exec from the
 * reference sources where they lie under /root/reference (see oracle/Makefile).
 * The reference's main program (set3d.f90) calls the module procedures
 * `reinit` (set3d.f90:308, :582) and `narrowBand` (set3d.f90:360, :460); with
 * `-Wl,--wrap=<flang symbol>` those calls land here, we dump the arrays that cross
 * the seam to raw little-endian files, and forward to the untouched reference
 * routine.  Nothing in the reference is modified; this file contains none of it.
 *
 * Files written into $LSF_REF_DUMP_DIR (nothing is written if it is unset):
 *   reinit<c>_in.f64 / reinit<c>_out.f64   phi before/after the c-th reinit call
 *   reinit<c>.meta                          "nx ny nz iter dx h" (%d %d %d %d %.17g %.17g)
 *   nb<c>_phi.f64, nb<c>_NB.i32, nb<c>_SB.i32   phi seen by / masks made by the c-th
 *                                           narrowBand call (c = 0 is set3d.f90:360,
 *                                           c = n is the call ending min/max iteration n)
 *                                           only for c listed in $LSF_REF_NB_DUMPS ("0,1,10")
 *   nb.count                                number of narrowBand calls so far
 *   advect_surfXX.f64, advect.meta          the advected surface nodes surfXX(nSurfNode,3) after the loop
 *                                           set3d.f90:491-501, and "nSurfNode  setPhiSurf-call-count"
 * $LSF_REF_STOP_AT_REINIT2=1 ends the program (exit 0) when reinit is entered the
 * second time, after dumping its input (= phi after min/max flow); the reference
 * never writes the result of that second call anywhere.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void __real__QMset_subsPreinit(double *phi, double *gradPhi, double *gradPhiMag, int *nx, int *ny,
                               int *nz, int *iter, double *dx, double *h);
void __real__QMset_subsPnarrowband(int *nx, int *ny, int *nz, double *dx, double *phi, int *phiNB,
                                   int *phiSB);
void __real__QMset_subsPsetphisurf(double *xLo, int *nx, int *ny, int *nz, double *dx, double *phiSurf,
                                   double *phi, int *nSurfNode, double *surfX, double *gradPhiSurf,
                                   double *gradPhi);

/* set3d.f90:487-501 advects a copy of the surface nodes (surfXX) by calling setPhiSurf once per move;
 * remember where that array lives so that its final state can be dumped when the main program reaches
 * its second reinit call (set3d.f90:582), i.e. right after the advection loop. */
static double *g_nodes = 0;
static int g_nnodes = 0;
static long g_setphisurf_calls = 0;

static void dump(const char *name, const void *p, size_t bytes)
{
    const char *dir = getenv("LSF_REF_DUMP_DIR");
    char path[4096];
    FILE *f;
    if (!dir) return;
    snprintf(path, sizeof path, "%s/%s", dir, name);
    f = fopen(path, "wb");
    if (!f) { perror(path); exit(3); }
    if (fwrite(p, 1, bytes, f) != bytes) { perror(path); exit(3); }
    fclose(f);
}

static int listed(const char *env, int c)
{
    const char *s = getenv(env);
    if (!s) return 0;
    while (*s) {
        char *e;
        long v = strtol(s, &e, 10);
        if (e == s) break;
        if (v == c) return 1;
        s = (*e == ',') ? e + 1 : e;
    }
    return 0;
}

void __wrap__QMset_subsPreinit(double *phi, double *gradPhi, double *gradPhiMag, int *nx, int *ny,
                               int *nz, int *iter, double *dx, double *h)
{
    static int call = 0;
    size_t n = (size_t)(*nx + 1) * (size_t)(*ny + 1) * (size_t)(*nz + 1);
    char name[64], meta[256];
    ++call;
    snprintf(name, sizeof name, "reinit%d_in.f64", call);
    dump(name, phi, n * sizeof(double));
    snprintf(meta, sizeof meta, "%d %d %d %d %.17g %.17g\n", *nx, *ny, *nz, *iter, *dx, *h);
    snprintf(name, sizeof name, "reinit%d.meta", call);
    dump(name, meta, strlen(meta));
    if (call == 2 && g_nodes) {
        char cnt[64];
        dump("advect_surfXX.f64", g_nodes, (size_t)g_nnodes * 3 * sizeof(double));
        snprintf(cnt, sizeof cnt, "%d %ld\n", g_nnodes, g_setphisurf_calls);
        dump("advect.meta", cnt, strlen(cnt));
    }
    if (call == 2 && getenv("LSF_REF_STOP_AT_REINIT2")) {
        fflush(stdout);
        exit(0);
    }
    __real__QMset_subsPreinit(phi, gradPhi, gradPhiMag, nx, ny, nz, iter, dx, h);
    snprintf(name, sizeof name, "reinit%d_out.f64", call);
    dump(name, phi, n * sizeof(double));
}

void __wrap__QMset_subsPnarrowband(int *nx, int *ny, int *nz, double *dx, double *phi, int *phiNB,
                                   int *phiSB)
{
    static int call = 0; /* 0 = set3d.f90:360, n = end of min/max iteration n */
    size_t n = (size_t)(*nx + 1) * (size_t)(*ny + 1) * (size_t)(*nz + 1);
    char name[64], cnt[32];
    __real__QMset_subsPnarrowband(nx, ny, nz, dx, phi, phiNB, phiSB);
    if (listed("LSF_REF_NB_DUMPS", call)) {
        snprintf(name, sizeof name, "nb%d_phi.f64", call);
        dump(name, phi, n * sizeof(double));
        snprintf(name, sizeof name, "nb%d_NB.i32", call);
        dump(name, phiNB, n * sizeof(int));
        snprintf(name, sizeof name, "nb%d_SB.i32", call);
        dump(name, phiSB, n * sizeof(int));
    }
    snprintf(cnt, sizeof cnt, "%d\n", call);
    dump("nb.count", cnt, strlen(cnt));
    ++call;
}

void __wrap__QMset_subsPsetphisurf(double *xLo, int *nx, int *ny, int *nz, double *dx, double *phiSurf,
                                   double *phi, int *nSurfNode, double *surfX, double *gradPhiSurf,
                                   double *gradPhi)
{
    g_nodes = surfX;
    g_nnodes = *nSurfNode;
    ++g_setphisurf_calls;
    __real__QMset_subsPsetphisurf(xLo, nx, ny, nz, dx, phiSurf, phi, nSurfNode, surfX, gradPhiSurf, gradPhi);
}
