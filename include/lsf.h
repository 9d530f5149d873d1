/*
 * lsf.h -- C ABI of liblsf_hip.so: the MI355X (gfx950) implementation of the hot path of
 * musheen/LevelSetFortran -- WENO5 Hamilton-Jacobi reinitialisation and min/max-flow smoothing
 * on a uniform 3-D grid.
 *
 * The reference has no FFI: its seam is the Fortran module procedure `reinit`
 * (subs.f90:717-725, called at set3d.f90:308 and :582) and the min/max loop written inline in the
 * main program (set3d.f90:394-462).  Each entry point below names the reference interface it
 * replaces.  levelsetfortran_amd/fortran/lsf_hip.f90 is the iso_c_binding shim that gives
 * the reference host those procedures back under their original names.
 *
 * Conventions
 *  - Plain C types only.  `int` is the reference's INTEGER*4, `double` its REAL under
 *    -fdefault-real-8 (Makefile:4).
 *  - Every field is Fortran-ordered exactly like `REAL phi(0:nx,0:ny,0:nz)` (subs.f90:721):
 *    extents (nx+1, ny+1, nz+1), `i` unit stride, element (i,j,k) at i + (nx+1)*(j + (ny+1)*k).
 *  - Functions without a `_device` suffix take HOST pointers owned by the caller; the library
 *    copies in on entry and out on exit and is quiescent on return.  `_device` functions take
 *    DEVICE pointers (HBM-resident fields, e.g. torch tensors) and a hipStream_t passed as void*
 *    (NULL = the null stream); they enqueue work and, unless stated otherwise, return after the
 *    stream has been synchronised.
 *  - All functions return LSF_OK (0) or an LSF_ERR_* code; lsf_last_error() describes the last
 *    failure on the calling thread.  There is no CPU fallback: without a usable gfx950 device the
 *    compute entry points return LSF_ERR_NO_DEVICE.
 */
#ifndef LSF_H
#define LSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSF_VERSION 106 /* 0.1.6: lsf_copy_bandwidth (measurement aid); the argument block of the exact-ordering kernels no longer carries experiment fields */

/* ---- return codes ---------------------------------------------------------------------- */
#define LSF_OK 0
#define LSF_ERR_NAN 1       /* RMS became NaN: the reference STOPs (subs.f90:926, set3d.f90:458) */
#define LSF_ERR_INVALID 2   /* bad argument                                                      */
#define LSF_ERR_HIP 3       /* HIP runtime failure                                               */
#define LSF_ERR_NO_DEVICE 4 /* no HIP device / not gfx950                                        */

/* ---- mode word: ordering | arithmetic ---------------------------------------------------- */
/* ordering (low byte) */
#define LSF_ORDER_GS 0     /* the reference's in-place Gauss-Seidel raster sweeps, reproduced      \
                              exactly by a tiled hyperplane wavefront     */
#define LSF_ORDER_JACOBI 1 /* double-buffered sweep: shards across GPUs, NOT reference-equal       */
#define LSF_ORDER_MASK 0xff
/* arithmetic (bit 8) */
#define LSF_ARITH_FAST 0x000   /* restructured fp64 arithmetic (FMA, shared terms, one reciprocal   \
                                  per WENO side): ~1e-16 per sweep from LSF_ARITH_STRICT, within   \
                                  1e-12 RMS of it for a thousand sweeps; over thousands of sweeps  \
                                  the scheme itself amplifies any rounding difference at kinks of  \
                                  the surface (isolated cells up to 1e-5 apart)  */
#define LSF_ARITH_STRICT 0x100 /* every operation as written in subs.f90, no contraction:           \
                                  bit-identical to the reference for LSF_ORDER_GS (the default of  \
                                  the Fortran shim); 1.8 x the time of LSF_ARITH_FAST              */
/* Operand range of the bit-identity promise of LSF_ARITH_STRICT.  The divisions by dx (subs.f90:509-530) and the weight
 * divisions (:536-546) are carried out by the hardware's division sequence WITHOUT its rescaling frame, which is the IEEE
 * quotient as long as numerator, divisor and quotient are normal numbers away from the ends of the exponent range: pinned
 * for fields and grid spacings scaled from 1e-140 to 1e140 (tests/test_gpu_parity.py, ..._at_the_ends_of_the_exponent_range)
 * and by 1.7e10 random quotients (profiles/micro/divcheck.hip); the weight divisions fall back to the framed division when
 * eps + IS >= 1e120.  Outside that range -- a difference of phi values divided by dx whose quotient is subnormal or
 * overflows, an infinite phi -- the last bit, or inf against NaN, may differ from the reference; such a field is already
 * outside anything a distance function holds (|phi| <= a few domain lengths, dx >= 1e-100 of it). */

/* ---- library / device ------------------------------------------------------------------- */
int lsf_version(void);
const char *lsf_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int lsf_device_count(void);
/* select the device used by the calling thread's subsequent calls (default 0) */
int lsf_set_device(int device);
/* release every cached device buffer of the current device */
int lsf_release_workspace(void);
/* Measurement aid (bench.py): with profiling enabled the next lsf_reinit*_ call brackets, per sweep,
 * the sweep kernel(s), the boundary-condition kernel and the RMS/stop kernel with HIP events on the
 * stream they are launched on.  lsf_profile_get returns the sums over the sweeps of that call (ms),
 * the number of sweep-kernel launches and the number of sweeps timed. */
/* diagnostic: 1 when the 16-byte loads / stores of the exact-ordering tiles (one raw buffer descriptor of 2^31 - 1 bytes per
 * tile image of rows_z + 6 planes) can address every point of a tile on an (nx, ny) grid; the kernels test the same expression. */
int lsf_skew_wide_fits(int nx, int ny, int rows_z);
int lsf_profile(int enable);
/* Measurement aid (bench.py "roofline.peak_measured", SURVEY.md section 8d: the roofline "also against a measured device-copy
 * bandwidth"): copies `bytes` bytes (>= 1 MiB, a multiple of 16) between two fresh device buffers reps + 1 times with a
 * streaming kernel (16 bytes per lane and access) and returns the rate of the fastest timed pass in GB/s, counting the bytes
 * read AND the bytes written.  No reference counterpart. */
int lsf_copy_bandwidth(size_t bytes, int reps, double *gbps);
int lsf_profile_get(double *sweep_kernel_ms, double *bc_ms, double *finish_ms,
                    long long *sweep_kernel_launches, int *sweeps);
/* name of the sweep kernel the last profiled call launched (the exact ordering picks its tile shape by
 * grid size); "" before the first profiled call.  The string is static. */
const char *lsf_profile_kernel(void);

/* ---- seam 1: reinit ----------------------------------------------------------------------
 * Replaces SUBROUTINE reinit(phi,gradPhi,gradPhiMag,nx,ny,nz,iter,dx,h), subs.f90:717-931.
 * Runs at most iter+1 sweeps (the reference loop is DO n=0,iter, subs.f90:735) of
 * {one raster sweep of the WENO5/Godunov update (subs.f90:743-852, weno :489-711, phiSign
 * :152-172); extrapolation boundary condition (:859-897); RMS change over all points (:902-914)}
 * and stops after the first sweep whose RMS is < tol (the reference uses 1e-5, :915).
 * gradPhi / gradPhiMag are dead outputs of the reference and are not part
 * of this interface.
 *   sweeps_done  (out, may be NULL) number of sweeps executed
 *   rms_trace    (out, may be NULL) RMS of sweep s (0-based) in rms_trace[s], s < trace_cap; the
 *                Fortran shim prints the reference's " Iteration: n  RMS Error: e" lines from it
 * Returns LSF_ERR_NAN if an RMS is NaN (phi then holds the state after that sweep).
 */
int lsf_reinit(double *phi, int nx, int ny, int nz, int iter, double dx, double h, double tol,
               int mode, int *sweeps_done, double *rms_trace, int trace_cap);

/* Same on an HBM-resident field.  d_phi is updated in place.  first_raster (0..7) is the number
 * of raster directions already consumed (0 = start with scan 1, like the reference); d_phiS may be
 * NULL (then phiS = phi on entry, subs.f90:731) or a device copy of the sign field to use. */
int lsf_reinit_device(double *d_phi, const double *d_phiS, int nx, int ny, int nz, int iter, double dx,
                      double h, double tol, int mode, int first_raster, int *sweeps_done,
                      double *rms_trace, int trace_cap, void *stream);

/* ---- seam 2: min/max flow ---------------------------------------------------------------
 * Replaces the loop DO n = 1,iter ... END DO at set3d.f90:394-462 (secondDeriv subs.f90:370-407,
 * minMax subs.f90:413-483, narrowBand subs.f90:178-207), hoisted into one call.
 * phiNB / phiSB: in = the masks made at set3d.f90:360; out = the masks the host holds after the
 * loop (the band is refreshed only on the non-exit path, set3d.f90:448-460).
 * Stops after the first iteration whose RMS is < tol (reference 1e-7, set3d.f90:448).
 */
int lsf_minmax(double *phi, int32_t *phiNB, int32_t *phiSB, int nx, int ny, int nz, int iter,
               double dx, double h1, double tol, int mode, int *iters_done, double *rms_trace,
               int trace_cap);

int lsf_minmax_device(double *d_phi, int32_t *d_phiNB, int32_t *d_phiSB, int nx, int ny, int nz,
                      int iter, double dx, double h1, double tol, int mode, int *iters_done,
                      double *rms_trace, int trace_cap, void *stream);

/* ---- narrowBand --------------------------------------------------------------------------
 * Replaces SUBROUTINE narrowBand(nx,ny,nz,dx,phi,phiNB,phiSB), subs.f90:178-207. */
int lsf_narrowband(const double *phi, int32_t *phiNB, int32_t *phiSB, int nx, int ny, int nz,
                   double dx);
int lsf_narrowband_device(const double *d_phi, int32_t *d_phiNB, int32_t *d_phiSB, int nx, int ny,
                          int nz, double dx, void *stream);

/* ---- phi0: inside/outside initialisation (the step before the hot path) -------------------
 * Replaces the centroid table + search loop of the main program, set3d.f90:196-268 (SURVEY.md section 8f
 * rank 1): for every grid point within 3 cells of the surface bounding box, the nearest triangle centroid
 * (first minimum), the sign of the triple product of its vertex vectors, smeared by phiSign(pS,dx,1);
 * 1.0 elsewhere (set3d.f90:161).  surfX is the host's REAL surfX(nSurfNode,3), surfElem its
 * INTEGER*4 surfElem(nSurfElem,3) (1-based), both Fortran-ordered HOST arrays; xLo/minX/maxX as computed at
 * set3d.f90:94-157.  Bit-identical to the reference. */
int lsf_phi0(double *phi, int nx, int ny, int nz, double dx, const double xLo[3], const double minX[3],
             const double maxX[3], const double *surfX, int nSurfNode, const int32_t *surfElem,
             int nSurfElem);
int lsf_phi0_device(double *d_phi, int nx, int ny, int nz, double dx, const double xLo[3],
                    const double minX[3], const double maxX[3], const double *surfX, int nSurfNode,
                    const int32_t *surfElem, int nSurfElem, void *stream);

/* ---- post-smoothing gradients + surface-node advection (the step after the hot path) ----------
 * Replaces set3d.f90:470-501: firstDeriv order 8 (subs.f90:309-347, with its
 * quirks) on the cells of phiSB, then every surface node is moved by x += phiSurf * gradPhiSurf with
 * setPhiSurf's trilinear interpolation (subs.f90:1056-1170) until phiSurf <= 1e-13 or `iters` (host: 1000)
 * passes.  surfXX is the host's REAL surfXX(nSurfNode,3) (Fortran-ordered HOST array), in = the nodes,
 * out = the advected nodes.  Bit-identical to the reference. */
int lsf_advect_nodes(const double *phi, const int32_t *phiSB, int nx, int ny, int nz, double dx,
                     const double xLo[3], double *surfXX, int nSurfNode, int iters);
int lsf_advect_nodes_device(const double *d_phi, const int32_t *d_phiSB, int nx, int ny, int nz, double dx,
                            const double xLo[3], double *surfXX, int nSurfNode, int iters, void *stream);

/* ---- device-resident chain behind the host seams -----------------
 * The reference's main program hands the same arrays from seam to seam (set3d.f90:196-582: inside/outside search ->
 * reinit #1 -> narrowBand -> min/max flow -> node advection -> reinit #2) and never changes them in between.  The
 * host-pointer entry points above keep the device copies of phi / phiNB / phiSB they worked on ("twins", tagged with the
 * host address and size).  lsf_mirror() lets a host that knows its own data flow use them:
 *   LSF_MIRROR_TRUST  a seam call whose host pointer and size match a current twin skips the host-to-device copy
 *                     (promise: the host has not written the array since the last seam call);
 *   LSF_MIRROR_LAZY   seam calls do not copy results back (promise: the host does not READ phi / phiNB / phiSB until it
 *                     has called lsf_mirror_sync on them); implies LSF_MIRROR_TRUST.
 * With both set the whole chain crosses PCIe once per array at most.  Default 0: every call copies in and out.
 * What the reference host does with phi between the seams has twins too:
 *   lsf_snapshot     phiO = phi                                   (set3d.f90:311)
 *   lsf_sumsq_diff   sum((phi - phiO)^2) over all points          (set3d.f90:505-516, the "Asymptotic Error")
 *   lsf_write_vti    the VTK ImageData writers                    (set3d.f90:319-351, :538-569)
 * Each works on the twins when they are current and on the host arrays otherwise.  */
#define LSF_MIRROR_TRUST 1
#define LSF_MIRROR_LAZY 2
int lsf_mirror(int flags);
/* copies the twin of `host` (phi, phiNB or phiSB of an earlier seam call) back if the host copy is stale */
int lsf_mirror_sync(void *host);
/* Drops the twin of `host` without copying anything.  A twin keeps the HOST ADDRESS of its array: under LSF_MIRROR_LAZY
 * an un-synced result is written through that address by lsf_release_workspace, by lsf_mirror() when LAZY is switched
 * off and by a later seam call that needs the slot for another array -- so an array with a twin must be synced
 * (lsf_mirror_sync) or forgotten (lsf_mirror_forget) BEFORE it is freed, and a new array that happens to get the same
 * address must not be taken for the old one (forget, or keep LSF_MIRROR_TRUST off).  A seam call that FAILS drops the
 * twins it touched: nothing of a failed call is ever copied to the host, and an un-synced earlier result in those slots is
 * lost with it (the error is the caller's signal). */
int lsf_mirror_forget(const void *host);
int lsf_snapshot(const double *phi, double *phiO, int nx, int ny, int nz);
int lsf_sumsq_diff(const double *phi, const double *phiO, int nx, int ny, int nz, double *sum);
/* Writes `phi` as VTK ImageData (raw appended Float64, i fastest) with the reference's header text.  The reference
 * writes an INTEGER*4 byte count that is 3 x too large and overflows at >= 448^3 points (set3d.f90:330); this writer
 * stores the true count, as UInt32 when it fits and with header_type="UInt64" otherwise (or whenever the environment
 * holds LSF_VTI_WIDE=1).  The payload streams from the
 * device twin through two pinned staging buffers (copy of chunk n + 1 overlaps the write of chunk n). */
int lsf_write_vti(const char *path, const double *phi, int nx, int ny, int nz, double dx, const double xLo[3]);

/* ---- block-decomposed building blocks (multi-GPU Jacobi; one process per GPU) -------------
 * A rank holds a box of the global field: local extents (lx,ly,lz), whose element (0,0,0) is the
 * global point (gx0,gy0,gz0); global extents are (nx+1,ny+1,nz+1).  The box includes ghost layers
 * (3 points towards each neighbouring rank) and the physical wall points it owns.  All calls are
 * asynchronous on `stream`.  The sweep and BC calls keep their partial sums in a per-stream buffer that grows on demand
 * (an allocation, and inside a lsf_sumsq bracket a flush of what has been summed so far); lsf_box_reserve sizes it up
 * front, after which the calls neither allocate nor synchronise.
 */
int lsf_box_reserve(void *stream, size_t max_partials);
typedef struct lsf_box {
    int lx, ly, lz;    /* local allocation extents (points)                          */
    int gx0, gy0, gz0; /* global index of local point (0,0,0)                        */
    int nx, ny, nz;    /* global: field is (0:nx,0:ny,0:nz)                          */
} lsf_box;

/* One Jacobi update (subs.f90:747-750 per cell, all reads from d_in) of the local cells
 * [lo[0],hi[0]) x [lo[1],hi[1]) x [lo[2],hi[2]) (local indices; must be interior cells of the
 * global grid, 1..n-1).  Adds sum((out-in)^2) over those cells to *d_sumsq (a device double;
 * contributions are combined in a fixed order, so results are reproducible).  */
int lsf_jacobi_sweep_box(const double *d_in, double *d_out, const double *d_phiS, const lsf_box *box,
                         const int lo[3], const int hi[3], double dx, double h, int mode,
                         double *d_sumsq, void *stream);

/* Extrapolation boundary condition (subs.f90:859-897, closed form) on the wall points of the
 * global grid that lie inside the local index range [lo,hi); reads interior values from d_out,
 * writes the wall points of d_out, adds sum((out-in)^2) over those wall points to *d_sumsq. */
int lsf_bc_box(const double *d_in, double *d_out, const lsf_box *box, const int lo[3], const int hi[3],
               double dx, double *d_sumsq, void *stream);

/* Optional bracket around the box calls of one sweep on one stream: between lsf_sumsq_begin and lsf_sumsq_end the
 * calls above keep their partial sums and lsf_sumsq_end adds them to *d_sumsq in ONE fixed-order reduction (one
 * launch instead of one per call; all bracketed calls must name the same d_sumsq, a call naming another one is reduced
 * at once).  The value of *d_sumsq is complete after lsf_sumsq_end. */
int lsf_sumsq_begin(void *stream);
int lsf_sumsq_end(void *stream);

/* Pack / unpack the local sub-box [lo,hi) to / from a contiguous buffer (i fastest). */
/* Up to six sub-boxes in ONE launch (the face slabs of a block: a decomposed sweep packs and unpacks three to six of them, and
 * below ~200^3 points per block each of those launches is shorter than the gap between two launches).  lo / hi: nreg rows of
 * three; d_bufs[q]: the buffer of sub-box q.  Same element order as lsf_pack_box; empty sub-boxes are skipped. */
int lsf_pack_boxes(const double *d_field, const lsf_box *box, int nreg, const int (*lo)[3], const int (*hi)[3],
                   double *const *d_bufs, void *stream);
int lsf_unpack_boxes(double *d_field, const lsf_box *box, int nreg, const int (*lo)[3], const int (*hi)[3],
                     double *const *d_bufs, void *stream);
int lsf_pack_boxes_f32(const float *d_field, const lsf_box *box, int nreg, const int (*lo)[3], const int (*hi)[3],
                       float *const *d_bufs, void *stream);
int lsf_unpack_boxes_f32(float *d_field, const lsf_box *box, int nreg, const int (*lo)[3], const int (*hi)[3],
                         float *const *d_bufs, void *stream);
int lsf_pack_box(const double *d_field, const lsf_box *box, const int lo[3], const int hi[3],
                 double *d_buf, void *stream);
int lsf_unpack_box(double *d_field, const lsf_box *box, const int lo[3], const int hi[3],
                   const double *d_buf, void *stream);

/* ---- the surface format in front of the path: binary STL -----------------------------
 * stlRead (subs.f90:17-121) reads the triangles and merges repeated vertices with a linear search per vertex:
 * O(triangles x nodes), minutes for a million triangles.  lsf_stl_read does the same job with a hash and returns EXACTLY
 * the reference's node numbering and connectivity: a vertex is merged with the FIRST earlier node whose three REAL*4
 * coordinates differ by less than 1e-13 each (subs.f90:73-75), nodes created by the triangle being read are not yet
 * searchable (the search bound nSurfNode is updated once per triangle, subs.f90:91; it starts at 3), numbering follows
 * first occurrence.  Host-only (no device needed).
 *   lsf_stl_read   parses `path`, keeps the result for the calling thread, returns the counts
 *   lsf_stl_get    copies it out -- surfX(nSurfNode,3) REAL(8), surfElem(nSurfElem,3) INTEGER*4 1-based, both in
 *                  Fortran order -- and releases it */
int lsf_stl_read(const char *path, int *nSurfElem, int *nSurfNode);
int lsf_stl_get(double *surfX, int32_t *surfElem);

/* ---- one process, every GPU of the node (replaces the call site set3d.f90:308 for a host that wants them all) --------
 * lsf_reinit_multi has lsf_reinit's arguments plus a device list.  The field is split into dims[0] x dims[1] x dims[2]
 * blocks (dims NULL: 1x1x2, 1x2x2, 2x2x2 for 2, 4, 8 devices -- BASELINE configurations 4 and 5; x, the unit-stride
 * axis, is cut last -- otherwise the prime factors dealt to z, y, x in turn), one block per entry of `devices`; the
 * library runs the Jacobi sweep (LSF_ORDER_JACOBI; for LSF_ORDER_GS see below) with 3-cell face halos copied peer to peer
 * over xGMI on a communication stream per device while the interior cells are updated on the compute stream, and one
 * host thread per device that only enqueues; the RMS of sweep s is judged while sweep s + 1 runs.  The result is bit-
 * identical to lsf_reinit with the same mode on one device.  `devices` may name a device more than once (several blocks
 * share it): that is how the path is tested on a one-GPU machine.  `phi` is a HOST array; a device twin of it left by
 * an earlier seam call (lsf_mirror) is brought home first and dropped afterwards.  The lsf_multi_* calls are the same
 * thing in pieces, for callers that keep the blocks resident (bench.py).
 * The RMS is judged a window of sweeps late; a run that can stop (tol > 0) keeps the field at the start of the last two
 * windows (two more field copies per block) and goes back to the stop sweep, a NaN sweep included.  With tol <= 0 nothing is
 * kept: when such a run returns LSF_ERR_NAN the sweep count and the trace are exact, the FIELD is that of the last sweep
 * enqueued (up to two windows later) -- undefined for the caller, as after the reference's STOP (subs.f90:926).
 *
 * LSF_ORDER_GS (fp64; dims NULL or {1, 1, ndev}): the reference's in-place ordering (subs.f90:743-852) itself, over ndev
 * slabs of tile layers in z, devices[0] at k = 0.  Every slab runs the dataflow launch of lsf_reinit on its own tile columns
 * of the SAME tile graph; the three planes next to a cut, a flag per tile next to it, a hyperplane counter per sweep and the
 * sweep's stop verdict are stored by the producing kernel straight into the neighbour's memory (peer stores over xGMI,
 * system scope, drained before the flag that announces them).  Field, sweep count and RMS trace are those of lsf_reinit
 * with the same mode, bit for bit -- with LSF_ARITH_STRICT the reference's.  Field memory per device is its slab plus three
 * planes per cut.  Needs peer access between the devices; slabs that share a device (the one-GPU rehearsal) need
 * their launches resident together: at most three per device unless GPU_MAX_HW_QUEUES is raised.  A tile that waits longer
 * than 4 s for a predecessor ends the call with LSF_ERR_HIP on every slab (bounded spins, no hang).  lsf_slabs_info: the
 * last such call of this thread -- slabs, resident blocks per slab, whether the shared buffers were fine-grained
 * allocations (LSF_SLAB_FINEGRAINED = 0 / 1 overrides: on when the devices differ), sweeps, and the time the longest of the
 * slabs' launches took (device events; uploads, transpositions and downloads excluded). */
typedef struct lsf_multi lsf_multi;
int lsf_slabs_info(int *slabs, int *blocks_per_slab, int *finegrained, int *sweeps, double *kernel_s);
/* First-contact self-test of what those slab launches assume of memory shared between two devices (nothing in the reference
 * corresponds: it is serial, README.md:17): two kernels, one per device, that wait for each other -- 200 rounds of a
 * message-passing litmus in both directions at once (8 KB stored into the peer's memory at system scope, s_waitcnt vmcnt(0),
 * a flag; the peer polls the flag and checks every word) and an atomic-max contest on one word.  Returns LSF_OK, or
 * LSF_ERR_HIP with *violated = the assumption of DESIGN.md section 6.1 that failed: 1 / 2 a payload announced by its flag
 * was not there (the drain does not cover stores to the peer, or the peer's loads were stale), 3 the atomic lost an update,
 * 4 the kernels did not run at the same time (bounded spins: no hang).  devA == devB runs both kernels on one device.
 * lsf_reinit_multi(LSF_ORDER_GS) runs it once per process for every pair of distinct neighbouring devices. */
int lsf_peer_selftest(int devA, int devB, int *violated);
int lsf_reinit_multi(double *phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
                     const int *devices, int ndev, const int dims[3], int *sweeps_done, double *rms_trace, int trace_cap);
int lsf_reinit_multi_f32(float *phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
                         const int *devices, int ndev, const int dims[3], int *sweeps_done, double *rms_trace,
                         int trace_cap);
/* f32 != 0: float fields.  Allocates two field buffers, the sign field and the halo buffers of every block. */
int lsf_multi_create(int nx, int ny, int nz, const int *devices, int ndev, const int dims[3], int f32, lsf_multi **out);
int lsf_multi_destroy(lsf_multi *m);
/* geometry of block r: global index of its local point 0, local extents (ghost layers included), owned global range */
int lsf_multi_block(const lsf_multi *m, int r, int g0[3], int ext[3], int own_lo[3], int own_hi[3], int *device);
int lsf_multi_scatter(lsf_multi *m, const void *host_phi);              /* host field -> blocks (ghosts included) */
int lsf_multi_upload_block(lsf_multi *m, int r, const void *d_block);   /* device pointer on block r's device     */
int lsf_multi_run(lsf_multi *m, int iter, double dx, double h, double tol, int mode, int *sweeps_done, double *rms_trace,
                  int trace_cap);                                       /* continues from the current block fields */
int lsf_multi_gather(lsf_multi *m, void *host_phi);                     /* owned points of every block -> host     */
/* How the blocks talk and how often the host looks at the RMS.
 *   check_every  1..64 sweeps per judging window (default 8): the host threads only enqueue inside a window and read
 *                the window's block sums one window late; a run with tol > 0 keeps the field at the start of the last
 *                two windows and, when a window holds the stop sweep, repeats the sweeps from its start to that sweep:
 *                field, sweep count and RMS trace are those of a driver that looks after every sweep (subs.f90:915).
 *   transport    LSF_TRANSPORT_PEER  one peer copy per neighbour and sweep (hipMemcpyPeerAsync, xGMI between the devices
 *                                    of a node), events + a host hand-shake of the two neighbours' threads; works with
 *                                    several blocks on one device (default)
 *                LSF_TRANSPORT_RCCL  ncclGroupStart / ncclSend + ncclRecv per neighbour / ncclGroupEnd on the block's
 *                                    communication stream, one communicator per block (ncclCommInitAll over the device
 *                                    list; librccl.so is loaded on first use); needs a distinct device per block
 *                LSF_TRANSPORT_MOCK  test aid: the RCCL schedule with a stand-in for the library (pull copies) that runs
 *                                    with several blocks on one device
 * lsf_multi_defaults sets what lsf_multi_create / lsf_reinit_multi of the calling thread start from (the environment variables
 * LSF_MULTI_CHECK_EVERY and LSF_MULTI_TRANSPORT = peer | rccl | mock override it).  lsf_multi_info: NULL for what is not
 * wanted; rccl_ranks = communicators held (0 unless the RCCL transport is on); host_enqueue_s = the longest any block's
 * thread spent enqueuing during the last lsf_multi_run (waits for its neighbours' threads and, when blocks share a device,
 * for that device's enqueue lock included), host_calls_s = the longest any thread spent inside the runtime / library calls
 * themselves, wall_s = that run's wall clock. */
#define LSF_TRANSPORT_PEER 0
#define LSF_TRANSPORT_RCCL 1
#define LSF_TRANSPORT_MOCK 2
int lsf_multi_defaults(int check_every, int transport);   /* per calling thread */
int lsf_multi_defaults_get(int *check_every, int *transport);
int lsf_multi_configure(lsf_multi *m, int check_every, int transport);
int lsf_multi_info(const lsf_multi *m, int *check_every, int *transport, int *rccl_ranks, int *rccl_version,
                   double *host_enqueue_s, double *host_calls_s, double *wall_s, int *sweeps_enqueued);

/* ---- single precision (BASELINE.json configuration 5: 1536^3 fp32 on 2x2x2 GPUs) -------------------
 * The reference is fp64 only (Makefile:4, -fdefault-real-8): there is no fp32 field to be identical to,
 * so these entry points exist for the Jacobi ordering and the FAST arithmetic only (mode must be
 * LSF_ORDER_JACOBI | LSF_ARITH_FAST, anything else is LSF_ERR_INVALID).  Same update as subs.f90:743-852 with
 * weno :489-711 and phiSign :152-172; the epsilon floor 1e-99 of subs.f90:533-534, which fp32 cannot hold,
 * becomes 1e-30 (in the unscaled units of the FAST algebra) and the three non-linear weights of a WENO side
 * are formed from q_k / (q_0+q_1+q_2) so that their products stay inside the fp32 range.  Fields are
 * `float` with the layout of the fp64 entry points; dx, h, tol and the RMS trace stay double (the RMS is
 * accumulated in double from fp32 differences, and divided by the true product nx*ny*nz: the reference's
 * INTEGER*4 product, subs.f90:914, which the fp64 entry points reproduce, is negative for the 1536^3 grid of
 * configuration 5).  Checked against the fp64 path within the tolerance
 * stated in tests/test_gpu_f32.py. */
int lsf_reinit_f32(float *phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
                   int *sweeps_done, double *rms_trace, int trace_cap);
int lsf_reinit_f32_device(float *d_phi, const float *d_phiS, int nx, int ny, int nz, int iter, double dx,
                          double h, double tol, int mode, int *sweeps_done, double *rms_trace, int trace_cap,
                          void *stream);
/* fp32 twins of the block-decomposed building blocks above (same contracts) */
int lsf_jacobi_sweep_box_f32(const float *d_in, float *d_out, const float *d_phiS, const lsf_box *box,
                             const int lo[3], const int hi[3], double dx, double h, int mode,
                             double *d_sumsq, void *stream);
int lsf_bc_box_f32(const float *d_in, float *d_out, const lsf_box *box, const int lo[3], const int hi[3],
                   double dx, double *d_sumsq, void *stream);
int lsf_pack_box_f32(const float *d_field, const lsf_box *box, const int lo[3], const int hi[3],
                     float *d_buf, void *stream);
int lsf_unpack_box_f32(float *d_field, const lsf_box *box, const int lo[3], const int hi[3],
                       const float *d_buf, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LSF_H */
