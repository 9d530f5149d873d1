// lsf_api.hip -- C ABI (include/lsf.h) over the gfx950 kernels.  Host orchestration only: the
// arithmetic lives in lsf_cell.hpp / lsf_kernels.hpp.  There is deliberately no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <chrono>
#include <map>
#include <set>
#include <unordered_map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/lsf.h"
#include "lsf_kernels.hpp"
#include "lsf_boxtile.hpp"
#include "lsf_skew.hpp"
#include "lsf_f32.hpp"

using namespace lsf;

namespace {

thread_local std::string g_err;
thread_local int g_device = 0;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

// LSF_TRACE=1: one stderr line per ABI call (the Fortran host buffers its stdout, so a crash loses it)
struct Trace {
    const char* name;
    bool on;
    std::chrono::steady_clock::time_point t0;
    explicit Trace(const char* n) : name(n), on(getenv("LSF_TRACE") != nullptr)
    {
        if (on) {
            fprintf(stderr, "[lsf] -> %s\n", name);
            t0 = std::chrono::steady_clock::now();
        }
    }
    ~Trace()
    {
        if (on)
            fprintf(stderr, "[lsf] <- %s (%.1f ms)\n", name,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(LSF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

// Exact-GS tile geometry: TA cells along i (LSF_GS_TA=16|32), NY x 4 cells in the cross-section
// (LSF_GS_NY=5: three lanes per cell, default; 4: four lanes per cell with one idle)
int gs_ta()
{
    const char* e = getenv("LSF_GS_TA");
    return (e && atoi(e) == 32) ? 32 : 16;
}
// wavefronts per skewed tile (lsf_skew.hpp): WY x WZ adjacent bundles marched in lock step, "WYxWZ" or "WY"; a leading
// 'c' selects the one-lane-per-cell map (bundles of 16 x 4 rows: "c1x4" = 16 x 16 rows), otherwise three lanes per cell
// (bundles of 5 x 4 rows).  *by = rows of a bundle in y (5 or 16).
// Unset: three lanes per cell, 2 x 2 wavefronts; one lane per cell (16 x 16 rows) on grids large enough for a sweep to offer
// the independent 16 x 16 tiles that keep two per CU busy.  Measured, ms per sweep, 2x2 / c1x4 (profiles/r03_tile_shapes.txt):
//   FAST    512^3 2.69 / 2.89   640^3 5.25 / 5.5   768^3 8.96 / 8.92   1024^3 22.1 / 21.2     -> from 700 cells across
//   STRICT  384^3 2.14 / 2.63   512^3 4.61 / 4.54  640^3 8.87 / 8.06                          -> from 500 cells across
// (the longer march of the reference's own arithmetic hides more of a tile's memory phases)
void gs_skew_w(int ny, int nz, bool strict, int* wy, int* wz, int* by)
{
    const char* e = getenv("LSF_GS_SKEW_W");
    if (!e && std::min(ny, nz) >= (strict ? 500 : 700)) e = "c1x4";
    const bool cell = e && (e[0] == 'c' || e[0] == 'C');
    if (cell) ++e;
    int y = 2, z = 2; // measured best at 256^3, 512^3 and 1024^3
    if (e && sscanf(e, "%dx%d", &y, &z) < 1) y = z = 2;
    if (e && !std::strchr(e, 'x')) z = 1;
    static const int ok[][2] = {{1, 1}, {2, 1}, {4, 1}, {1, 2}, {2, 2}, {4, 2}, {2, 4}};
    static const int okc[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}};
    if (cell) {
        *by = 16, *wy = 1, *wz = 4;
        for (auto& s : okc)
            if (s[0] == y && s[1] == z) *wy = y, *wz = z;
        return;
    }
    *by = 5, *wy = *wz = 2;
    for (auto& s : ok)
        if (s[0] == y && s[1] == z) *wy = y, *wz = z;
}
// dispatch on the tile shape: CALL(WY, WZ, BY) with compile-time arguments
#define LSF_SK_SHAPES(CALL, wy_, wz_, by_)               \
    do {                                                 \
        const int shape_ = (by_)*256 + (wy_)*16 + (wz_); \
        if (shape_ == 5 * 256 + 0x11) CALL(1, 1, 5);     \
        else if (shape_ == 5 * 256 + 0x21) CALL(2, 1, 5); \
        else if (shape_ == 5 * 256 + 0x41) CALL(4, 1, 5); \
        else if (shape_ == 5 * 256 + 0x12) CALL(1, 2, 5); \
        else if (shape_ == 5 * 256 + 0x42) CALL(4, 2, 5); \
        else if (shape_ == 5 * 256 + 0x24) CALL(2, 4, 5); \
        else if (shape_ == 16 * 256 + 0x11) CALL(1, 1, 16); \
        else if (shape_ == 16 * 256 + 0x12) CALL(1, 2, 16); \
        else if (shape_ == 16 * 256 + 0x13) CALL(1, 3, 16); \
        else if (shape_ == 16 * 256 + 0x14) CALL(1, 4, 16); \
        else CALL(2, 2, 5);                              \
    } while (0)
int gs_ny()
{
    const char* e = getenv("LSF_GS_NY");
    return (e && atoi(e) == 4) ? 4 : 5;
}
// dataflow launch: axis the tiles march along -- "y" (default; the kernel then runs on the x <-> y transposed field, see
// k_transpose_xy) or "x" (the field as it is) -- and the number of field buffers in rotation (3 or 4)
int gs_march()
{
    const char* e = getenv("LSF_GS_MARCH");
    return (e && (e[0] == 'x' || e[0] == 'X')) ? 0 : 1;
}
// dataflow launch: one block per tile (k_reinit_gs_persist, default) or resident blocks that carry on down a tile column
// (k_reinit_gs_stream, LSF_GS_STREAM=1); LSF_GS_CONT selects when such a block continues
int gs_stream()
{
    const char* e = getenv("LSF_GS_STREAM");
    return e && atoi(e) != 0; // opt-in: measured slower than the one-block-per-tile launch
}
// 0 = never; 1 = when the two cross upstream tiles of the next tile are claimed (the block then waits for them); 2 (default) = only
// when they are done already (the block never waits while it holds a column)
int gs_cont()
{
    const char* e = getenv("LSF_GS_CONT");
    return e ? std::max(0, std::min(2, atoi(e))) : 2;
}
int gs_nbuf()
{
    const char* e = getenv("LSF_GS_NBUF");
    return (e && atoi(e) == 3) ? 3 : 4;
}
#ifndef LSF_DF_PAD
#define LSF_DF_PAD 1024 // ints between the control words of the dataflow launch (experiment: 0 = adjacent, as before)
#endif
constexpr int MM_TA = 32;      // tile length along i of the exact-GS min/max kernel
constexpr int CHECK_EVERY = 8; // sweeps between host reads of the device stop flag
// fp32 sweep: the pure x-face wall points are written by the sweep kernel instead of k_bc (kernel argument `xwall`):
// k_bc 41 -> 17 us for +8 us in the sweep kernel at 512^3.  Not done for the fp64 kernel: k_bc 48 -> 18 us there, but
// the sweep kernel itself lost 30-50 us (same bench command, 1.70 -> 1.73..1.75 ms): no net gain.
constexpr int F32_XWALL = 1;

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct TileList {
    uint32_t* d = nullptr;
    std::vector<int> off; // plane offsets, size nplanes+1
    uint32_t last = 0;    // skewed lists: the only tile of the last plane
};

// dataflow schedule of the exact ordering: tiles of a batch of sweeps in slot order, per-sweep table, hyperplane sizes
// (small: the task list itself is rebuilt on the device for every launch, k_build_order)
struct BatchPlan {
    int* d_meta = nullptr; // [ns] start slot per sweep | [nslots + 1] first entry per slot | [4 * DF_BATCH] {sign i, j, k,
                           // spacing} per sweep | [np] tiles per hyperplane | [np + 1] offsets of the hyperplanes in the tile list
    long total = 0;        // entries of the task list
    int nslots = 0;
};

enum Slot { S_PONG, S_PHIS, S_PART, S_CTL, S_TRACE, S_HPHI, S_HNB, S_HSB, S_CEN, S_VTX, S_BFLAG, S_CHG, S_BACKUP, S_PART2, S_PLANECNT, S_DBG, S_COLSUM, S_ORDER, S_GRAD, S_NODES, S_STAMP, S_PONG2, S_PONG3, S_PONG4, S_SNAP, S_NSLOTS };

// partial sums of the box calls issued on one stream; `deferred`: between lsf_sumsq_begin and lsf_sumsq_end the calls
// append their partials instead of reducing them one by one
struct StreamPart {
    Buf buf;
    bool deferred = false;
    size_t used = 0;          // doubles appended so far
    double* target = nullptr; // the d_sumsq of the deferred calls
};

// device twin of a host array handed through the host seams (lsf_mirror)
struct Twin {
    const void* host = nullptr;
    size_t bytes = 0;
    bool current = false;    // the device copy holds the latest content
    bool host_stale = false; // ... and the host copy does not (LSF_MIRROR_LAZY)
};

struct Ctx {
    Buf slot[S_NSLOTS];
    Twin twin_phi, twin_nb, twin_sb, twin_snap; // S_HPHI, S_HNB, S_HSB, S_SNAP
    int mirror = 0;
    std::map<hipStream_t, StreamPart> part_by_stream;
    std::map<uint64_t, TileList> tiles;
    std::map<uint64_t, TileList> skew_tiles;
    std::map<int, uint32_t*> sk_tables;            // skewed tiles: lookup tables per tile shape (SkTile::rel_entry / lane_off)
    std::map<std::array<int, 6>, BatchPlan> plans; // dataflow schedule: batch plans per grid / raster phase / sweep count
    bool checked = false;
};

std::mutex g_mu;
std::map<int, Ctx> g_ctx;

// optional event timing of the sweep kernels (lsf_profile): accumulated over the last core call
struct Profile {
    bool on = false;
    double sweep_ms = 0, bc_ms = 0, finish_ms = 0;
    long sweep_launches = 0;
    int sweeps = 0;
    const char* kernel = ""; // sweep kernel of the last profiled call, with its template arguments (the instance rocprofv3 lists)
    char kernel_buf[96] = "";
    std::vector<hipEvent_t> ev; // 4 per timed sweep
};
thread_local Profile g_prof;

void prof_begin()
{
    g_prof.sweep_ms = g_prof.bc_ms = g_prof.finish_ms = 0;
    g_prof.sweep_launches = 0;
    g_prof.sweeps = 0;
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
}
void prof_mark(hipStream_t st)
{
    if (!g_prof.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    g_prof.ev.push_back(e);
}
void prof_end(int sweeps_done)
{
    if (!g_prof.on) return;
    const int timed = std::min<int>(sweeps_done, (int)g_prof.ev.size() / 4);
    for (int s = 0; s < timed; ++s) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, g_prof.ev[4 * s], g_prof.ev[4 * s + 1]);
        (void)hipEventElapsedTime(&b, g_prof.ev[4 * s + 1], g_prof.ev[4 * s + 2]);
        (void)hipEventElapsedTime(&c, g_prof.ev[4 * s + 2], g_prof.ev[4 * s + 3]);
        g_prof.sweep_ms += a;
        g_prof.bc_ms += b;
        g_prof.finish_ms += c;
    }
    g_prof.sweeps = timed;
}

int ensure_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(LSF_ERR_NO_DEVICE, "no HIP device visible: liblsf_hip has no CPU fallback");
    }
    if (g_device >= n) return fail(LSF_ERR_NO_DEVICE, "selected device index out of range");
    HIPCHK(hipSetDevice(g_device));
    std::lock_guard<std::mutex> lk(g_mu);
    Ctx& c = g_ctx[g_device];
    if (!c.checked) {
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, g_device));
        if (std::strncmp(pr.gcnArchName, "gfx950", 6) != 0)
            return fail(LSF_ERR_NO_DEVICE, std::string("device is ") + pr.gcnArchName +
                                               ", this library carries gfx950 (MI355X) code only");
        c.checked = true;
    }
    return LSF_OK;
}

Ctx& ctx()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_ctx[g_device];
}

int ws(Buf& b, size_t bytes)
{
    if (b.bytes >= bytes && b.p) return LSF_OK;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    HIPCHK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return LSF_OK;
}

// tiles of an (nA,nB,nC) tile grid sorted by hyperplane A+B+C (sweep frame)
int get_tiles(int nA, int nB, int nC, TileList** out)
{
    if (nA > 1023 || nB > 1023 || nC > 1023) return fail(LSF_ERR_INVALID, "grid too large for tile index packing");
    const uint64_t key = ((uint64_t)nA << 40) | ((uint64_t)nB << 20) | (uint64_t)nC;
    Ctx& c = ctx();
    auto it = c.tiles.find(key);
    if (it == c.tiles.end()) {
        TileList tl;
        std::vector<uint32_t> h;
        h.reserve((size_t)nA * nB * nC);
        const int nplanes = nA + nB + nC - 2;
        tl.off.assign(nplanes + 1, 0);
        for (int P = 0; P < nplanes; ++P) {
            tl.off[P] = (int)h.size();
            // a-fastest inside a plane: neighbouring blocks share halo rows
            for (int C = 0; C < nC; ++C)
                for (int B = 0; B < nB; ++B) {
                    const int A = P - B - C;
                    if (A < 0 || A >= nA) continue;
                    h.push_back((uint32_t)A | ((uint32_t)B << 10) | ((uint32_t)C << 20));
                }
        }
        tl.off[nplanes] = (int)h.size();
        HIPCHK(hipMalloc((void**)&tl.d, h.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(tl.d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = c.tiles.emplace(key, std::move(tl)).first;
    }
    *out = &it->second;
    return LSF_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Skewed tiles (lsf_skew.hpp): (m, fB, fC) with TA m <= Fx + Fy + Fz < TA m + TA for some cell of row bundle
// (fB, fC); the m range assumes full bundles (NY x 4 rows), a superset for the partial bundles at the far walls
// (such a tile simply finds no cell to work on).  Sorted by hyperplane m + fB + fC.
int get_skew_tiles(int nxi, int nTj, int nTk, int ta, int nyc, int nzc, TileList** out)
{
    const int m_max = (nxi - 1 + nyc * nTj - 1 + nzc * nTk - 1) / ta;
    if (m_max > 1023 || nTj > 1023 || nTk > 1023) return fail(LSF_ERR_INVALID, "grid too large for tile index packing");
    const uint64_t key = ((uint64_t)nxi << 44) | ((uint64_t)nTj << 28) | ((uint64_t)nTk << 12) | (uint64_t)(nyc << 6 | nzc);
    Ctx& c = ctx();
    auto it = c.skew_tiles.find(key);
    if (it == c.skew_tiles.end()) {
        const int nplanes = m_max + nTj + nTk - 1;
        std::vector<int> cnt(nplanes + 1, 0);
        auto m_lo = [&](int B, int C) { return (nyc * B + nzc * C) / ta; };
        auto m_hi = [&](int B, int C) { return (nyc * B + nyc - 1 + nzc * C + nzc - 1 + nxi - 1) / ta; };
        for (int C = 0; C < nTk; ++C)
            for (int B = 0; B < nTj; ++B)
                for (int m = m_lo(B, C); m <= m_hi(B, C); ++m) ++cnt[m + B + C];
        TileList tl;
        tl.off.assign(nplanes + 1, 0);
        for (int P = 0; P < nplanes; ++P) tl.off[P + 1] = tl.off[P] + cnt[P];
        std::vector<uint32_t> h((size_t)tl.off[nplanes]);
        std::vector<int> fill(tl.off.begin(), tl.off.end() - 1);
        // bundles of one plane in (C, B) order: neighbouring blocks share halo rows
        for (int C = 0; C < nTk; ++C)
            for (int B = 0; B < nTj; ++B)
                for (int m = m_lo(B, C); m <= m_hi(B, C); ++m)
                    h[(size_t)fill[m + B + C]++] = (uint32_t)m | ((uint32_t)B << 10) | ((uint32_t)C << 20);
        tl.last = (uint32_t)m_hi(nTj - 1, nTk - 1) | ((uint32_t)(nTj - 1) << 10) | ((uint32_t)(nTk - 1) << 20);
        if (tl.off[nplanes] - tl.off[nplanes - 1] != 1 || h.back() != tl.last)
            return fail(LSF_ERR_HIP, "internal: skewed tile list does not end in a single tile");
        HIPCHK(hipMalloc((void**)&tl.d, h.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(tl.d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = c.skew_tiles.emplace(key, std::move(tl)).first;
    }
    *out = &it->second;
    return LSF_OK;
}

// Spacing of two consecutive sweeps on skewed tiles: sweep g+1 (signs db) may run hyperplane P' of its tiles in
// slot start[g+1] + P' only after every sweep-g tile (signs da) that holds a cell within stencil reach (the cell
// itself or up to 3 cells along one axis) has run:  H > P_g(v) - P_{g+1}(u)  for all such pairs.  With
// P = floor((Fx + Fy + Fz) / TA) + fB + fC and floor(p) - floor(q) <= floor(p - q) + 1 the maximum separates into
// one 1-D scan per axis (exact up to one slot).
long skew_spacing(const int* da, const int* db, int nx, int ny, int nz, int ta, int nyc, int nzc)
{
    const int nq[3] = {nx, ny, nz}, ts[3] = {0, nyc, nzc};
    long m0[3], md[3];
    for (int ax = 0; ax < 3; ++ax) {
        const int n = nq[ax], TS = ts[ax], nT = TS ? cdiv(n - 1, TS) : 0;
        // frame index + TA * bundle index of interior coordinate g (1..n-1) for direction sgn
        auto w = [&](int g, int sgn) -> long {
            if (!TS) return sgn > 0 ? g - 1 : n - 1 - g;
            const int t = (g - 1) / TS, y = (g - 1) - t * TS, cntt = std::min(TS, n - 1 - t * TS);
            const int fT = sgn > 0 ? t : nT - 1 - t, b = sgn > 0 ? y : cntt - 1 - y;
            return (long)TS * fT + b + (long)ta * fT;
        };
        m0[ax] = md[ax] = -(1L << 40);
        for (int g = 1; g <= n - 1; ++g) {
            const long wu = w(g, db[ax]);
            for (int d = -3; d <= 3; ++d) {
                if (g + d < 1 || g + d > n - 1) continue;
                const long v = w(g + d, da[ax]) - wu;
                if (d == 0) m0[ax] = std::max(m0[ax], v);
                else md[ax] = std::max(md[ax], v);
            }
        }
        if (md[ax] < m0[ax]) md[ax] = m0[ax]; // an axis with a single interior cell has no neighbour
    }
    long tot = m0[0] + m0[1] + m0[2];
    for (int ax = 0; ax < 3; ++ax) tot = std::max(tot, m0[0] + m0[1] + m0[2] - m0[ax] + md[ax]);
    const long q = tot >= 0 ? tot / ta : -((-tot + ta - 1) / ta);
    return q + 2;
}

double rms_denominator(int nx, int ny, int nz)
{
    // INTEGER*4 product nx*ny*nz, subs.f90:914 / set3d.f90:447 (wraps like the reference)
    return (double)(int32_t)((uint32_t)nx * (uint32_t)ny * (uint32_t)nz);
}

const int RASTER_SIGN[8][3] = {{+1, +1, +1}, {+1, +1, -1}, {+1, -1, -1}, {-1, -1, -1},
                               {-1, +1, -1}, {-1, -1, +1}, {-1, +1, +1}, {+1, -1, +1}};

// launch geometry of k_bc for the region [lo, hi) of a box: one z-slice of the grid per wall face the region touches
dim3 bc_grid(const Box& bx, const int lo[3], const int hi[3], unsigned* faces)
{
    const int m = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    const int nwall[3] = {bx.nx, bx.ny, bx.nz}, g0[3] = {bx.gx0, bx.gy0, bx.gz0};
    unsigned list = 0, cnt = 0;
    for (int f = 0; f < 6; ++f) {
        const int a = f >> 1, wl = ((f & 1) ? nwall[a] : 0) - g0[a];
        if (wl >= lo[a] && wl < hi[a]) list |= (unsigned)f << (3 * cnt++);
    }
    *faces = list;
    return dim3(cdiv(m, 64), m, cnt);
}

int check_dims(int nx, int ny, int nz)
{
    if (nx < 2 || ny < 2 || nz < 2) return fail(LSF_ERR_INVALID, "nx, ny, nz must be >= 2");
    if ((double)(nx + 1) * (ny + 1) * (nz + 1) > 9.0e9) return fail(LSF_ERR_INVALID, "field too large");
    // the Jacobi kernels address one k-plane through a buffer descriptor with 32-bit byte offsets
    if ((double)(nx + 1) * (ny + 1) * 8.0 > 2.0e9) return fail(LSF_ERR_INVALID, "a k-plane of the field exceeds 2 GB");
    return LSF_OK;
}

// ---------------------------------------------------------------------------------------------
// One fp64 Jacobi sweep over the cells [lo, hi) of a box: picks the kernel and its launch geometry.
//   STRICT            k_reinit_jacobi<true>  (per-cell arithmetic of the reference)
//   FAST              k_reinit_jacobi_sh<WX, BY> (WENO interfaces shared along x and z); WX = wavefronts a block
//                     spans along x, chosen so that the 64 WX - 1 cells of a block tile the row with the fewest wavefronts
//   3-cell x rims     k_reinit_jacobi<., true> (lanes along y)
// LSF_JAC_SH = 0 forces the per-cell kernel, "WXxBY" a block shape (measurement aids; all FAST choices are bit-identical).
struct JacPlan {
    int kind = 0; // 0 per-cell kernel, 1 per-cell THINX, 2 shared-interface kernel
    dim3 grid;
    int wx = 1, by = 4, nbx = 0, nby = 0, nbz = 0;
    int kc = JAC_KC; // planes a block marches: JAC_KC, halved while a thin region (a rim of a decomposed sweep) would leave CUs idle
    long nparts = 0; // partial sums the launch writes
};
// planes per block: JAC_KC (= F32_KC) amortises the six-plane window a block loads before its first cell; a region with few
// block columns (the 3-cell rims of a decomposed sweep) gets shorter marches until the launch has ~8 blocks per CU
static int jacobi_kc(long columns, int cz)
{
    int kc = JAC_KC;
    while (kc > 4 && columns * cdiv(cz, kc) < 2048) kc >>= 1;
    return kc;
}
JacPlan jacobi_plan(const int lo[3], const int hi[3], bool strict)
{
    JacPlan p;
    const int cx = hi[0] - lo[0], cy = hi[1] - lo[1], cz = hi[2] - lo[2];
    const bool thinx = cx <= 8 && cy >= 32;
    const char* env = getenv("LSF_JAC_SH");
    int fwx = 0, fby = 0;
    const bool off = env && env[0] == '0' && env[1] == 0;
    if (env && !off && sscanf(env, "%dx%d", &fwx, &fby) != 2) fwx = fby = 0;
    if (strict || thinx || off) {
        p.kind = thinx ? 1 : 0;
        const long cols = thinx ? (long)cdiv(cy, JAC_BX) * cdiv(cx, JAC_BY) : (long)cdiv(cx, JAC_BX) * cdiv(cy, JAC_BY);
        p.kc = jacobi_kc(cols, cz);
        p.grid = thinx ? dim3(cdiv(cy, JAC_BX), cdiv(cx, JAC_BY), cdiv(cz, p.kc)) : dim3(cdiv(cx, JAC_BX), cdiv(cy, JAC_BY), cdiv(cz, p.kc));
        p.nparts = (long)p.grid.x * p.grid.y * p.grid.z;
        return p;
    }
    p.kind = 2;
    int best = 1 << 30;
    for (int wx : {1, 2, 4, 8}) {
        const int waves = wx * cdiv(cx, 64 * wx - 1);
        if (waves < best) best = waves, p.wx = wx;
    }
    p.by = p.wx == 1 ? 4 : (p.wx == 2 ? 2 : 1);
    static const int shapes[][2] = {{1, 4}, {2, 2}, {4, 1}, {4, 2}, {8, 1}};
    for (auto& sh : shapes)
        if (sh[0] == fwx && sh[1] == fby) p.wx = fwx, p.by = fby;
    p.nbx = cdiv(cx, 64 * p.wx - 1), p.nby = cdiv(cy, p.by);
    p.kc = jacobi_kc((long)p.nbx * p.nby, cz), p.nbz = cdiv(cz, p.kc);
    const long nblk = (long)p.nbx * p.nby * p.nbz;
    p.nparts = (nblk + 7) / 8 * 8; // the launch is padded to a multiple of the 8 XCDs (k_reinit_jacobi_sh)
    p.grid = dim3((unsigned)p.nparts);
    return p;
}
void jacobi_launch(const JacPlan& p, bool strict, const double* A, double* B, const double* phiS, const Box& bx, const int lo[3],
                   const int hi[3], double dx, double h, double* part, const int* done, hipStream_t st)
{
#define LSF_JAC_OLD(ST_, TX_)                                                                                               \
    hipLaunchKernelGGL((k_reinit_jacobi<ST_, TX_>), p.grid, dim3(JAC_BX, JAC_BY), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], \
                       hi[0], hi[1], hi[2], dx, h, part, done, p.kc)
#define LSF_JAC_SH(WX_, BY_)                                                                                                   \
    hipLaunchKernelGGL((k_reinit_jacobi_sh<WX_, BY_>), p.grid, dim3(64 * WX_ * BY_), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], \
                       hi[0], hi[1], hi[2], dx, h, part, done, p.nbx, p.nby, p.nbz, p.kc)
    if (p.kind == 2) {
        const int sh = p.wx * 16 + p.by;
        if (sh == 0x14) LSF_JAC_SH(1, 4);
        else if (sh == 0x22) LSF_JAC_SH(2, 2);
        else if (sh == 0x41) LSF_JAC_SH(4, 1);
        else if (sh == 0x42) LSF_JAC_SH(4, 2);
        else LSF_JAC_SH(8, 1);
    } else if (strict) {
        if (p.kind == 1) LSF_JAC_OLD(true, true);
        else LSF_JAC_OLD(true, false);
    } else {
        if (p.kind == 1) LSF_JAC_OLD(false, true);
        else LSF_JAC_OLD(false, false);
    }
#undef LSF_JAC_OLD
#undef LSF_JAC_SH
}

// the same for float fields (k_reinit_jacobi_f32_sh<WX>: a lane owns the pair (i, j) / (i, j+1); one pair row per block)
JacPlan jacobi_plan_f32(const int lo[3], const int hi[3])
{
    JacPlan p;
    const int cx = hi[0] - lo[0], cy = hi[1] - lo[1], cz = hi[2] - lo[2], npair = cdiv(cy, 2);
    const bool thinx = cx <= 8 && cy >= 64;
    const char* env = getenv("LSF_JAC_SH");
    const bool off = env && env[0] == '0' && env[1] == 0;
    if (thinx || off) {
        p.kind = thinx ? 1 : 0;
        const long cols = thinx ? (long)cdiv(npair, F32_BX) * cdiv(cx, F32_BY) : (long)cdiv(cx, F32_BX) * cdiv(npair, F32_BY);
        p.kc = jacobi_kc(cols, cz);
        p.grid = thinx ? dim3(cdiv(npair, F32_BX), cdiv(cx, F32_BY), cdiv(cz, p.kc)) : dim3(cdiv(cx, F32_BX), cdiv(npair, F32_BY), cdiv(cz, p.kc));
        p.nparts = (long)p.grid.x * p.grid.y * p.grid.z;
        return p;
    }
    p.kind = 2;
    int best = 1 << 30;
    for (int wx : {1, 2, 4, 8}) {
        const int waves = wx * cdiv(cx, 64 * wx - 1);
        if (waves < best) best = waves, p.wx = wx;
    }
    int fwx = 0, fby = 0;
    if (env && sscanf(env, "%dx%d", &fwx, &fby) >= 1 && (fwx == 1 || fwx == 2 || fwx == 4 || fwx == 8)) p.wx = fwx;
    p.by = 1;
    p.nbx = cdiv(cx, 64 * p.wx - 1), p.nby = npair;
    p.kc = jacobi_kc((long)p.nbx * p.nby, cz), p.nbz = cdiv(cz, p.kc);
    p.nparts = ((long)p.nbx * p.nby * p.nbz + 7) / 8 * 8;
    p.grid = dim3((unsigned)p.nparts);
    return p;
}
void jacobi_launch_f32(const JacPlan& p, const float* A, float* B, const float* phiS, const Box& bx, const int lo[3], const int hi[3],
                       double dx, double h, double* part, const int* done, int xwall, hipStream_t st)
{
#define LSF_F32_SH(WX_)                                                                                                   \
    hipLaunchKernelGGL((k_reinit_jacobi_f32_sh<WX_>), p.grid, dim3(64 * WX_), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], hi[0], \
                       hi[1], hi[2], (float)dx, (float)h, part, done, xwall, p.nbx, p.nby, p.nbz, p.kc)
    if (p.kind == 2) {
        if (p.wx == 1) LSF_F32_SH(1);
        else if (p.wx == 2) LSF_F32_SH(2);
        else if (p.wx == 4) LSF_F32_SH(4);
        else LSF_F32_SH(8);
    } else if (p.kind == 1)
        hipLaunchKernelGGL((k_reinit_jacobi_f32<true>), p.grid, dim3(F32_BX, F32_BY), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], hi[0],
                           hi[1], hi[2], (float)dx, (float)h, part, done, xwall, p.kc);
    else
        hipLaunchKernelGGL((k_reinit_jacobi_f32<false>), p.grid, dim3(F32_BX, F32_BY), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], hi[0],
                           hi[1], hi[2], (float)dx, (float)h, part, done, xwall, p.kc);
#undef LSF_F32_SH
}

int gs_schedule();
int reinit_slot_core(double* d_phi, const double* d_phiS_in, int nx, int ny, int nz, int iter, double dx, double h,
                     double tol, int mode, int first_raster, int* sweeps_done, double* rms_trace, int trace_cap,
                     hipStream_t st);

// ---------------------------------------------------------------------------------------------
// The Jacobi reinit loop of one device (fp64 and fp32): per sweep the sweep kernel, the extrapolation BC and the
// fixed-order RMS reduction with the stop / NaN test on the device; the host looks at the stop flag every CHECK_EVERY
// sweeps (sweeps enqueued past the verdict return at once).
template <typename T>
int jacobi_loop(T* d_phi, const T* d_phiS_in, int nx, int ny, int nz, int iter, double dx, double h, double tol, bool strict,
                int* sweeps_done, double* rms_trace, int trace_cap, hipStream_t st)
{
    constexpr bool F32 = sizeof(T) == 4;
    int rc = LSF_OK;
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    const int max_sweeps = iter + 1; // DO n=0,iter (subs.f90:735)
    if ((rc = ws(c.slot[S_PONG], n * sizeof(T)))) return rc;
    const T* d_phiS = d_phiS_in;
    if (!d_phiS) {
        if ((rc = ws(c.slot[S_PHIS], n * sizeof(T)))) return rc;
        HIPCHK(hipMemcpyAsync(c.slot[S_PHIS].p, d_phi, n * sizeof(T), hipMemcpyDeviceToDevice, st)); // subs.f90:731
        d_phiS = (const T*)c.slot[S_PHIS].p;
    }
    if ((rc = ws(c.slot[S_CTL], 64))) return rc;
    if ((rc = ws(c.slot[S_TRACE], (size_t)max_sweeps * sizeof(double)))) return rc;
    int* ctl = (int*)c.slot[S_CTL].p;
    double* d_trace = (double*)c.slot[S_TRACE].p;
    HIPCHK(hipMemsetAsync(ctl, 0, 64, st));

    const int jlo[3] = {1, 1, 1}, jhi[3] = {nx, ny, nz};
    JacPlan jp;
    long n_sweep_part;
    if constexpr (F32) jp = jacobi_plan_f32(jlo, jhi);
    else jp = jacobi_plan(jlo, jhi, strict);
    n_sweep_part = jp.nparts;
    const Box bx{nx + 1, ny + 1, nz + 1, 0, 0, 0, nx, ny, nz};
    const int blo[3] = {0, 0, 0}, bhi[3] = {nx + 1, ny + 1, nz + 1};
    unsigned bfaces = 0;
    const dim3 bgrid = bc_grid(bx, blo, bhi, &bfaces); // the whole grid: all six faces
    const long n_bc_part = (long)bgrid.x * bgrid.y * bgrid.z;
    const long n_part = n_sweep_part + n_bc_part;
    if ((rc = ws(c.slot[S_PART], (size_t)n_part * sizeof(double)))) return rc;
    double* part = (double*)c.slot[S_PART].p;
    // fp64: the reference's INTEGER*4 product nx*ny*nz (subs.f90:914), wrapping like the reference; it is negative for the
    // 1536^3 grid of configuration 5 (RMS = NaN, STOP) and there is no fp32 reference behaviour to mirror: fp32 fields
    // divide by the true product
    const double den = F32 ? (double)nx * (double)ny * (double)nz : rms_denominator(nx, ny, nz);
    const int xwall = F32 ? F32_XWALL : 0;

    T* bufs[2] = {d_phi, (T*)c.slot[S_PONG].p};
    int host_ctl[3] = {0, 0, 0};
    prof_begin();
    for (int s = 0; s < max_sweeps; ++s) {
        const T* A = bufs[s & 1];
        T* B = bufs[(s + 1) & 1];
        prof_mark(st);
        if constexpr (F32) jacobi_launch_f32(jp, A, B, d_phiS, bx, jlo, jhi, dx, h, part, ctl, xwall, st);
        else jacobi_launch(jp, strict, A, B, d_phiS, bx, jlo, jhi, dx, h, part, ctl, st);
        prof_mark(st);
        hipLaunchKernelGGL(k_bc<T>, bgrid, dim3(64), 0, st, A, B, bx, 0, 0, 0, nx + 1, ny + 1, nz + 1, (T)dx, part + n_sweep_part, ctl,
                           xwall, bfaces);
        prof_mark(st);
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(RED_T), 0, st, part, n_part, den, tol, d_trace, max_sweeps, ctl);
        prof_mark(st);
        if ((s + 1) % CHECK_EVERY == 0 && s + 1 < max_sweeps) {
            HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (host_ctl[0]) break;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int nsw = host_ctl[1];
    prof_end(nsw);
    g_prof.sweep_launches = g_prof.sweeps;
    if (jp.kind == 2 && F32) snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "k_reinit_jacobi_f32_sh<%d>", jp.wx);
    else if (jp.kind == 2) snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "k_reinit_jacobi_sh<%d,%d>", jp.wx, jp.by);
    else snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, F32 ? "k_reinit_jacobi_f32<%s>" : (strict ? "k_reinit_jacobi<true,%s>" : "k_reinit_jacobi<false,%s>"), jp.kind == 1 ? "true" : "false");
    g_prof.kernel = g_prof.kernel_buf;
    if (bufs[nsw & 1] != d_phi)
        HIPCHK(hipMemcpyAsync(d_phi, bufs[nsw & 1], n * sizeof(T), hipMemcpyDeviceToDevice, st));
    if (rms_trace && trace_cap > 0 && nsw > 0)
        HIPCHK(hipMemcpyAsync(rms_trace, d_trace, sizeof(double) * (size_t)std::min(nsw, trace_cap), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sweeps_done) *sweeps_done = nsw;
    if (host_ctl[2]) return fail(LSF_ERR_NAN, "RMS became NaN (the reference STOPs here, subs.f90:926)");
    return LSF_OK;
}

// ---------------------------------------------------------------------------------------------
int reinit_core(double* d_phi, const double* d_phiS_in, int nx, int ny, int nz, int iter, double dx,
                double h, double tol, int mode, int first_raster, int* sweeps_done, double* rms_trace,
                int trace_cap, hipStream_t st)
{
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    if (iter < 0) return fail(LSF_ERR_INVALID, "iter must be >= 0");
    if (first_raster < 0 || first_raster > 7) return fail(LSF_ERR_INVALID, "first_raster must be 0..7");
    const int order = mode & LSF_ORDER_MASK;
    const bool strict = (mode & LSF_ARITH_STRICT) != 0;
    if (order != LSF_ORDER_GS && order != LSF_ORDER_JACOBI) return fail(LSF_ERR_INVALID, "unknown ordering");
    if (!d_phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    if (order == LSF_ORDER_GS)
        return reinit_slot_core(d_phi, d_phiS_in, nx, ny, nz, iter, dx, h, tol, mode, first_raster, sweeps_done,
                                rms_trace, trace_cap, st);
    return jacobi_loop<double>(d_phi, d_phiS_in, nx, ny, nz, iter, dx, h, tol, strict, sweeps_done, rms_trace, trace_cap, st);
}

// fp32 Jacobi reinit (BASELINE configuration 5): the same loop on float fields; the RMS is accumulated in double from
// fp32 differences
int reinit_f32_core(float* d_phi, const float* d_phiS_in, int nx, int ny, int nz, int iter, double dx, double h,
                    double tol, int* sweeps_done, double* rms_trace, int trace_cap, hipStream_t st)
{
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    if (iter < 0) return fail(LSF_ERR_INVALID, "iter must be >= 0");
    if (!d_phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    return jacobi_loop<float>(d_phi, d_phiS_in, nx, ny, nz, iter, dx, h, tol, false, sweeps_done, rms_trace, trace_cap, st);
}

// lookup tables of a skewed tile shape (lsf_skew.hpp: sk_fill_tables), built once per shape and device
int get_sk_tables(int wy, int wz, int by, const uint32_t** out)
{
    Ctx& c = ctx();
    const int key = by * 256 + wy * 16 + wz;
    auto it = c.sk_tables.find(key);
    if (it == c.sk_tables.end()) {
        std::vector<uint32_t> h;
#define LSF_SK_TAB(WY_, WZ_, BY_)                                                            \
    do {                                                                                     \
        using T_ = SkTile<16, WY_, WZ_, BY_>;                                                \
        h.assign((size_t)T_::REL_WORDS + T_::OFF_WORDS, 0u);                                 \
        sk_fill_tables<16, WY_, WZ_, BY_>(h.data());                                         \
    } while (0)
        LSF_SK_SHAPES(LSF_SK_TAB, wy, wz, by);
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
    const int ta = gs_ta();
    int nyc = gs_ny();
    int sched = gs_schedule();
    // Default: skewed tiles (lsf_skew.hpp), 2 x 2 wavefronts each, dependencies resolved in the kernel (`dataflow`,
    // one launch per batch of sweeps).  Measured per sweep: dataflow / slot launches on skewed tiles (`skew`) / slot
    // launches on the box tiles of lsf_boxtile.hpp (`slots`): 256^3 0.97 / 1.24 / 1.75 ms, 512^3 3.25 / 4.61 / 6.83 ms,
    // 1024^3 24.7 / 25.2 / 38.0 ms.
    if (sched < 0) sched = 5;
    bool persist = sched == 5; // k_reinit_gs_stream / k_reinit_gs_persist
    const bool stream = gs_stream();
    if (persist) sched = 3;
    // skewed tiles need TA = 16, NY = 5 and at least two interior cells per axis
    const bool skew = sched == 3 && ta == 16 && nyc == 5 && nx >= 3 && ny >= 3 && nz >= 3;
    persist = persist && skew;
    int wy = 1, wz = 1, by = 5, nzc = 4;
    if (skew) {
        gs_skew_w(std::min(nx, ny), nz, strict, &wy, &wz, &by); // (the dataflow launch may swap x and y)
        nyc = by * wy, nzc = 4 * wz; // rows of a tile in y and z
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
    if (skew && !getenv("LSF_GS_NO_TABLES") && (rc = get_sk_tables(wy, wz, by, &fa.tables))) return rc;

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
            const std::array<int, 6> key{knx, kny, nz, phase, ns, wy * 16 + wz + 256 * (int)tr + 1024 * nbuf + 8192 * by};
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
            // one block per tile; a block takes its tile from the ticket counter, so the grid only has to be large enough
            // (2-D: gridDim.x * blockDim.x must stay below 2^32)
            const dim3 grid((unsigned)std::min<long>(fa.total, 65536), (unsigned)((fa.total + 65535) / 65536));
#define LSF_LAUNCH_DF(WY_, WZ_, BY_)                                                                             \
    do {                                                                                                         \
        if (strict) hipLaunchKernelGGL((k_reinit_gs_persist<16, WY_, WZ_, BY_, true>), grid, dim3(64 * WY_ * WZ_), 0, st, fa); \
        else hipLaunchKernelGGL((k_reinit_gs_persist<16, WY_, WZ_, BY_, false>), grid, dim3(64 * WY_ * WZ_), 0, st, fa);       \
    } while (0)
            // Default: one block per tile (k_reinit_gs_persist).  LSF_GS_STREAM=1: the launch with column continuation
            // (k_reinit_gs_stream: resident blocks that loop over tiles and carry on down a tile column; lsf_stream.hip);
            // LSF_GS_CONT=0: that loop without continuation (every tile acquired from the list).
            if (stream) {
                fa.cont_on = gs_cont();
                int cus = 0;
                HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g_device));
                const hipError_t le = (hipError_t)launch_gs_stream(wy, wz, by, strict, st, fa, cus, nullptr);
                if (le != hipSuccess) return fail(LSF_ERR_HIP, std::string("k_reinit_gs_stream: ") + hipGetErrorString(le));
            } else {
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
                    if (stream)
                        fprintf(stderr, "[lsf] column continuation: %llu of %llu tiles continued; not continued: end of column %llu, previous sweep not past %llu, "
                                "cross tiles unclaimed %llu, claim lost %llu\n", hd[8], hd[2], hd[9], hd[10], hd[11], hd[12]);
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
    auto launch_tiles = [&](int grid, hipStream_t s_) {
        if (skew) {
#define LSF_LAUNCH_SKEW(WY_, WZ_, BY_)                                                                                     \
    do {                                                                                                                   \
        if (strict)                                                                                                        \
            hipLaunchKernelGGL((k_reinit_gs_skew<16, WY_, WZ_, BY_, true>), dim3(grid), dim3(64 * WY_ * WZ_), 0, s_, fa);  \
        else                                                                                                               \
            hipLaunchKernelGGL((k_reinit_gs_skew<16, WY_, WZ_, BY_, false>), dim3(grid), dim3(64 * WY_ * WZ_), 0, s_, fa); \
    } while (0)
            LSF_SK_SHAPES(LSF_LAUNCH_SKEW, wy, wz, by);
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
        if (skew) snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "%s<16,%d,%d,%d,%s>", !slots_loop ? (stream ? "k_reinit_gs_stream" : "k_reinit_gs_persist") : "k_reinit_gs_skew",
                           wy, wz, by, strict ? "true" : "false");
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

// Exact ordering: fixed-point passes (fast), as many per iteration as the field has needed so far; if a fixed point is
// ever not certified, restore the input and redo the call with MM_MAX_FIX passes per iteration, and if that is still
// not enough (never observed) with the tile-hyperplane wavefront.
int minmax_core(double* d_phi, int32_t* d_nb, int32_t* d_sb, int nx, int ny, int nz, int iter, double dx,
                double h1, double tol, int mode, int* iters_done, double* rms_trace, int trace_cap,
                hipStream_t st)
{
    const char* e = getenv("LSF_MINMAX_TILES");
    const bool force_tiles = e && atoi(e) != 0;
    if ((mode & LSF_ORDER_MASK) != LSF_ORDER_GS || force_tiles || !d_phi || !d_nb || !d_sb || iter <= 0 ||
        check_dims(nx, ny, nz))
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

int box_ok(const lsf_box* b, const int lo[3], const int hi[3])
{
    if (!b || !lo || !hi) return fail(LSF_ERR_INVALID, "NULL box/range");
    if (b->lx < 1 || b->ly < 1 || b->lz < 1) return fail(LSF_ERR_INVALID, "empty box");
    if ((double)b->lx * b->ly * 8.0 > 2.0e9) return fail(LSF_ERR_INVALID, "a k-plane of the box exceeds 2 GB");
    const int ext[3] = {b->lx, b->ly, b->lz};
    for (int a = 0; a < 3; ++a)
        if (lo[a] < 0 || hi[a] > ext[a]) return fail(LSF_ERR_INVALID, "range outside the local box");
    return LSF_OK;
}

// the region of a box sweep must consist of interior cells of the global grid with their stencil inside the box
int sweep_region_ok(const lsf_box* box, const int lo[3], const int hi[3])
{
    const int g0[3] = {box->gx0, box->gy0, box->gz0}, nn[3] = {box->nx, box->ny, box->nz};
    const int ext[3] = {box->lx, box->ly, box->lz};
    for (int a = 0; a < 3; ++a) {
        if (lo[a] + g0[a] < 1 || hi[a] - 1 + g0[a] > nn[a] - 1)
            return fail(LSF_ERR_INVALID, "sweep region must lie in the global interior 1..n-1");
        for (int e = 0; e < 2; ++e) {
            const int l = e ? hi[a] - 1 : lo[a], g = l + g0[a];
            // reach: 3 only if the cell can take the WENO branch along this axis; 1 is always needed
            const int reach = (g > 3 && g < nn[a] - 4) ? 3 : 1;
            if (l - reach < 0 || l + reach > ext[a] - 1)
                return fail(LSF_ERR_INVALID, "stencil of the sweep region leaves the local box (ghost layers missing)");
        }
    }
    return LSF_OK;
}

int flush_partials(hipStream_t st, StreamPart& sp)
{
    if (sp.used > 0 && sp.target)
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(RED_T), 0, st, (const double*)sp.buf.p, (long)sp.used, sp.target);
    sp.used = 0;
    HIPCHK(hipGetLastError());
    return LSF_OK;
}

int stream_partials(hipStream_t st, size_t count, double** out)
{
    Ctx& c = ctx();
    StreamPart& sp = c.part_by_stream[st];
    const size_t at = sp.deferred ? sp.used : 0;
    if ((at + count) * sizeof(double) > sp.buf.bytes) {
        // growing replaces the buffer: reduce what the kernels in flight have written to it first
        int rc = flush_partials(st, sp);
        if (rc) return rc;
        if (sp.deferred) HIPCHK(hipStreamSynchronize(st));
        if ((rc = ws(sp.buf, std::max(count, (size_t)1 << 16) * 2 * sizeof(double)))) return rc;
        *out = (double*)sp.buf.p;
        return LSF_OK;
    }
    *out = (double*)sp.buf.p + at;
    return LSF_OK;
}

// after the kernel of a box call has been launched: reduce its partials now, or leave them for lsf_sumsq_end
int finish_partials(hipStream_t st, double* part, long np, double* d_sumsq)
{
    StreamPart& sp = ctx().part_by_stream[st];
    if (sp.deferred && (sp.target == nullptr || sp.target == d_sumsq)) {
        sp.target = d_sumsq;
        sp.used = (size_t)(part - (double*)sp.buf.p) + (size_t)np;
    } else {
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(RED_T), 0, st, (const double*)part, np, d_sumsq);
    }
    HIPCHK(hipGetLastError());
    return LSF_OK;
}

template <typename T>
int bc_box_impl(const T* d_in, T* d_out, const lsf_box* box, const int lo[3], const int hi[3], double dx,
                       double* d_sumsq, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = box_ok(box, lo, hi))) return rc;
    if (!d_in || !d_out || !d_sumsq) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) return LSF_OK;
    hipStream_t st = (hipStream_t)stream;
    const Box bx{box->lx, box->ly, box->lz, box->gx0, box->gy0, box->gz0, box->nx, box->ny, box->nz};
    unsigned faces = 0;
    const dim3 grid = bc_grid(bx, lo, hi, &faces);
    if (grid.z == 0) return LSF_OK; // no wall of the global grid inside this region
    const long np = (long)grid.x * grid.y * grid.z;
    double* part = nullptr;
    if ((rc = stream_partials(st, (size_t)np, &part))) return rc;
    hipLaunchKernelGGL(k_bc<T>, grid, dim3(64), 0, st, d_in, d_out, bx, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], (T)dx,
                       part, (const int*)nullptr, 0, faces);
    return finish_partials(st, part, np, d_sumsq);
}

template <typename T>
int pack_impl(const T* d_field, T* d_field_w, const lsf_box* box, const int lo[3], const int hi[3], T* d_buf,
                     int unpack, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = box_ok(box, lo, hi))) return rc;
    if (!d_buf || (!d_field && !d_field_w)) return fail(LSF_ERR_INVALID, "NULL pointer");
    const int e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
    if (e0 <= 0 || e1 <= 0 || e2 <= 0) return LSF_OK;
    const long n = (long)e0 * e1 * e2;
    const int grid = (int)std::min<long>((n + 255) / 256, 4096);
    const Box bx{box->lx, box->ly, box->lz, box->gx0, box->gy0, box->gz0, box->nx, box->ny, box->nz};
    hipLaunchKernelGGL(k_pack<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_field, d_buf, bx, lo[0], lo[1],
                       lo[2], e0, e1, e2, unpack, d_field_w);
    HIPCHK(hipGetLastError());
    return LSF_OK;
}

int f32_mode_ok(int mode)
{
    if ((mode & LSF_ORDER_MASK) != LSF_ORDER_JACOBI || (mode & LSF_ARITH_STRICT))
        return fail(LSF_ERR_INVALID,
                    "fp32 fields: only LSF_ORDER_JACOBI | LSF_ARITH_FAST exists (the reference is fp64; there is no "
                    "fp32 field to be identical to)");
    return LSF_OK;
}

} // namespace

// =============================================================================================

// ---- twins of the host seams (include/lsf.h: lsf_mirror) ---------------------------------------
// bring the host array into its device slot unless the twin is current and may be trusted
// Every writer of a twinned slot (S_HPHI, S_HNB, S_HSB, S_SNAP) comes through here FIRST: a result that lives only in the
// slot (LSF_MIRROR_LAZY, host_stale) and belongs to another host array -- or to the same address with another size -- is
// written home before the slot is resized (ws may free it) or overwritten; then the slot is sized for the new owner and the
// twin forgotten (the caller tags it again through twin_in / twin_out when it has put something there).
int twin_claim(Ctx& c, Twin& t, Slot slot, const void* host, size_t bytes)
{
    if (t.host_stale && t.host && c.slot[slot].p && !(t.host == host && t.bytes == bytes))
        HIPCHK(hipMemcpy(const_cast<void*>(t.host), c.slot[slot].p, t.bytes, hipMemcpyDeviceToHost));
    if (!(t.host == host && t.bytes == bytes)) t = Twin{};
    return ws(c.slot[slot], bytes);
}
// a seam call has failed: what it left in the slot is undefined and nothing of it may ever reach the host
void twin_drop(Twin& t) { t = Twin{}; }

int twin_in(Ctx& c, Twin& t, Slot slot, const void* host, size_t bytes)
{
    int rc = twin_claim(c, t, slot, host, bytes);
    if (rc) return rc;
    const bool hit = (c.mirror & (LSF_MIRROR_TRUST | LSF_MIRROR_LAZY)) && t.current && t.host == host && t.bytes == bytes;
    if (!hit) {
        HIPCHK(hipMemcpy(c.slot[slot].p, host, bytes, hipMemcpyHostToDevice));
        t.host_stale = false;
    }
    t.host = host, t.bytes = bytes, t.current = true;
    return LSF_OK;
}
// the device slot now holds a result for `host`: copy it back unless the host asked for lazy twins
int twin_out(Ctx& c, Twin& t, Slot slot, void* host, size_t bytes)
{
    t.host = host, t.bytes = bytes, t.current = true;
    if (c.mirror & LSF_MIRROR_LAZY) {
        t.host_stale = true;
        return LSF_OK;
    }
    HIPCHK(hipMemcpy(host, c.slot[slot].p, bytes, hipMemcpyDeviceToHost));
    t.host_stale = false;
    return LSF_OK;
}
// device pointer of the current twin of `host`, or nullptr
const void* twin_of(Ctx& c, const void* host, size_t bytes, Twin** which = nullptr, Slot* slot = nullptr)
{
    struct { Twin* t; Slot s; } all[] = {{&c.twin_phi, S_HPHI}, {&c.twin_nb, S_HNB}, {&c.twin_sb, S_HSB}, {&c.twin_snap, S_SNAP}};
    for (auto& e : all)
        if (e.t->current && e.t->host == host && (bytes == 0 || e.t->bytes == bytes) && c.slot[e.s].p) {
            if (which) *which = e.t;
            if (slot) *slot = e.s;
            return c.slot[e.s].p;
        }
    return nullptr;
}

__global__ __launch_bounds__(256) void k_sumsq_diff(const double* __restrict__ a, const double* __restrict__ b, long n,
                                                    double* __restrict__ partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
        const double d = a[p] - b[p];
        acc = __builtin_fma(d, d, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(RED_T) void k_sum_partials(const double* __restrict__ partials, long nPart, double* __restrict__ out)
{
    __shared__ double red[RED_T];
    const double tot = block_sum(partials, nPart, red);
    if (threadIdx.x == 0) *out = tot;
}

extern "C" {

int lsf_version(void) { return LSF_VERSION; }

const char* lsf_last_error(void) { return g_err.c_str(); }

int lsf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int lsf_set_device(int device)
{
    if (device < 0) return fail(LSF_ERR_INVALID, "negative device index");
    g_device = device;
    return ensure_device();
}

// diagnostic: does the 16-byte path of the exact-ordering tiles apply to tiles of rows_z rows in z on an (nx, ny) grid?  (the guard
// of skew_tile's buffer descriptor, evaluated on the host: tests/test_host_logic.py)
int lsf_skew_wide_fits(int nx, int ny, int rows_z) { return sk_wide_image_fits((long)(nx + 1) * (ny + 1), rows_z) ? 1 : 0; }

int lsf_profile(int enable)
{
    g_prof.on = enable != 0;
    prof_begin();
    return LSF_OK;
}

int lsf_profile_get(double* sweep_kernel_ms, double* bc_ms, double* finish_ms, long long* sweep_kernel_launches,
                    int* sweeps)
{
    if (sweep_kernel_ms) *sweep_kernel_ms = g_prof.sweep_ms;
    if (bc_ms) *bc_ms = g_prof.bc_ms;
    if (finish_ms) *finish_ms = g_prof.finish_ms;
    if (sweep_kernel_launches) *sweep_kernel_launches = g_prof.sweep_launches;
    if (sweeps) *sweeps = g_prof.sweeps;
    return LSF_OK;
}

const char* lsf_profile_kernel(void) { return g_prof.kernel; }

int lsf_release_workspace(void)
{
    int rc = ensure_device();
    if (rc) return rc;
    Ctx& c = ctx();
    HIPCHK(hipDeviceSynchronize());
    {   // lazy twins: the host copies they stand for are brought up to date before the device copies go
        struct { Twin* t; Slot s; } all[] = {{&c.twin_phi, S_HPHI}, {&c.twin_nb, S_HNB}, {&c.twin_sb, S_HSB}, {&c.twin_snap, S_SNAP}};
        for (auto& e : all) {
            if (e.t->host_stale && e.t->host && c.slot[e.s].p)
                HIPCHK(hipMemcpy(const_cast<void*>(e.t->host), c.slot[e.s].p, e.t->bytes, hipMemcpyDeviceToHost));
            *e.t = Twin{};
        }
    }
    for (auto& b : c.slot) {
        if (b.p) HIPCHK(hipFree(b.p));
        b = Buf{};
    }
    for (auto& kv : c.part_by_stream)
        if (kv.second.buf.p) HIPCHK(hipFree(kv.second.buf.p));
    c.part_by_stream.clear();
    for (auto& kv : c.plans) {
        if (kv.second.d_meta) HIPCHK(hipFree(kv.second.d_meta));
    }
    c.plans.clear();
    for (auto& kv : c.sk_tables)
        if (kv.second) HIPCHK(hipFree(kv.second));
    c.sk_tables.clear();
    for (auto* lists : {&c.tiles, &c.skew_tiles}) {
        for (auto& kv : *lists)
            if (kv.second.d) HIPCHK(hipFree(kv.second.d));
        lists->clear();
    }
    return LSF_OK;
}

int lsf_reinit_device(double* d_phi, const double* d_phiS, int nx, int ny, int nz, int iter, double dx, double h,
                      double tol, int mode, int first_raster, int* sweeps_done, double* rms_trace,
                      int trace_cap, void* stream)
{
    Trace trace_("lsf_reinit_device");
    int rc = ensure_device();
    if (rc) return rc;
    return reinit_core(d_phi, d_phiS, nx, ny, nz, iter, dx, h, tol, mode, first_raster, sweeps_done, rms_trace,
                       trace_cap, (hipStream_t)stream);
}

int lsf_reinit(double* phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
               int* sweeps_done, double* rms_trace, int trace_cap)
{
    Trace trace_("lsf_reinit");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    Ctx& c = ctx();
    const size_t bytes = (size_t)(nx + 1) * (ny + 1) * (nz + 1) * sizeof(double);
    if ((rc = twin_in(c, c.twin_phi, S_HPHI, phi, bytes))) return rc;
    double* d = (double*)c.slot[S_HPHI].p;
    rc = reinit_core(d, nullptr, nx, ny, nz, iter, dx, h, tol, mode, 0, sweeps_done, rms_trace, trace_cap, nullptr);
    if (rc == LSF_OK || rc == LSF_ERR_NAN) {
        const std::string keep = g_err;
        const int rc2 = twin_out(c, c.twin_phi, S_HPHI, phi, bytes);
        if (rc2) return rc2;
        g_err = keep;
    } else
        twin_drop(c.twin_phi);
    return rc;
}

int lsf_minmax_device(double* d_phi, int32_t* d_phiNB, int32_t* d_phiSB, int nx, int ny, int nz, int iter,
                      double dx, double h1, double tol, int mode, int* iters_done, double* rms_trace,
                      int trace_cap, void* stream)
{
    Trace trace_("lsf_minmax_device");
    int rc = ensure_device();
    if (rc) return rc;
    return minmax_core(d_phi, d_phiNB, d_phiSB, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace,
                       trace_cap, (hipStream_t)stream);
}

int lsf_minmax(double* phi, int32_t* phiNB, int32_t* phiSB, int nx, int ny, int nz, int iter, double dx,
               double h1, double tol, int mode, int* iters_done, double* rms_trace, int trace_cap)
{
    Trace trace_("lsf_minmax");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi || !phiNB || !phiSB) return fail(LSF_ERR_INVALID, "NULL field");
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if ((rc = twin_in(c, c.twin_phi, S_HPHI, phi, n * sizeof(double)))) return rc;
    if ((rc = twin_in(c, c.twin_nb, S_HNB, phiNB, n * sizeof(int32_t)))) return rc;
    if ((rc = twin_in(c, c.twin_sb, S_HSB, phiSB, n * sizeof(int32_t)))) return rc;
    double* d = (double*)c.slot[S_HPHI].p;
    int32_t* dnb = (int32_t*)c.slot[S_HNB].p;
    int32_t* dsb = (int32_t*)c.slot[S_HSB].p;
    rc = minmax_core(d, dnb, dsb, nx, ny, nz, iter, dx, h1, tol, mode, iters_done, rms_trace, trace_cap, nullptr);
    if (rc == LSF_OK || rc == LSF_ERR_NAN) {
        const std::string keep = g_err;
        int rc2 = twin_out(c, c.twin_phi, S_HPHI, phi, n * sizeof(double));
        if (!rc2) rc2 = twin_out(c, c.twin_nb, S_HNB, phiNB, n * sizeof(int32_t));
        if (!rc2) rc2 = twin_out(c, c.twin_sb, S_HSB, phiSB, n * sizeof(int32_t));
        if (rc2) return rc2;
        g_err = keep;
    } else
        twin_drop(c.twin_phi), twin_drop(c.twin_nb), twin_drop(c.twin_sb);
    return rc;
}

int lsf_narrowband_device(const double* d_phi, int32_t* d_phiNB, int32_t* d_phiSB, int nx, int ny, int nz,
                          double dx, void* stream)
{
    Trace trace_("lsf_narrowband_device");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!d_phi || !d_phiNB || !d_phiSB) return fail(LSF_ERR_INVALID, "NULL field");
    rc = narrowband_core(d_phi, d_phiNB, d_phiSB, (size_t)(nx + 1) * (ny + 1) * (nz + 1), dx, (hipStream_t)stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return LSF_OK;
}

int lsf_narrowband(const double* phi, int32_t* phiNB, int32_t* phiSB, int nx, int ny, int nz, double dx)
{
    Trace trace_("lsf_narrowband");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi || !phiNB || !phiSB) return fail(LSF_ERR_INVALID, "NULL field");
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if ((rc = twin_in(c, c.twin_phi, S_HPHI, phi, n * sizeof(double)))) return rc;
    // the masks are outputs here: un-synced masks of OTHER host arrays go home first, then the slots are ours
    if ((rc = twin_claim(c, c.twin_nb, S_HNB, phiNB, n * sizeof(int32_t)))) return rc;
    if ((rc = twin_claim(c, c.twin_sb, S_HSB, phiSB, n * sizeof(int32_t)))) return rc;
    twin_drop(c.twin_nb), twin_drop(c.twin_sb);
    if ((rc = narrowband_core((const double*)c.slot[S_HPHI].p, (int32_t*)c.slot[S_HNB].p, (int32_t*)c.slot[S_HSB].p, n,
                              dx, nullptr)))
        return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    if ((rc = twin_out(c, c.twin_nb, S_HNB, phiNB, n * sizeof(int32_t)))) return rc;
    return twin_out(c, c.twin_sb, S_HSB, phiSB, n * sizeof(int32_t));
}

int lsf_phi0_device(double* d_phi, int nx, int ny, int nz, double dx, const double xLo[3], const double minX[3],
                    const double maxX[3], const double* surfX, int nSurfNode, const int32_t* surfElem, int nSurfElem,
                    void* stream)
{
    Trace trace_("lsf_phi0_device");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!d_phi || !xLo || !minX || !maxX || !surfX || !surfElem) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (nSurfNode < 1 || nSurfElem < 1) return fail(LSF_ERR_INVALID, "empty surface");
    hipStream_t st = (hipStream_t)stream;
    // search box, set3d.f90:180-186 (same expressions, host side)
    const int im = (int)std::floor((minX[0] - xLo[0]) / dx) - 3, ip = (int)std::floor((maxX[0] - xLo[0]) / dx) + 3;
    const int jm = (int)std::floor((minX[1] - xLo[1]) / dx) - 3, jp = (int)std::floor((maxX[1] - xLo[1]) / dx) + 3;
    const int km = (int)std::floor((minX[2] - xLo[2]) / dx) - 3, kp = (int)std::floor((maxX[2] - xLo[2]) / dx) + 3;
    if (im < 0 || jm < 0 || km < 0 || ip > nx || jp > ny || kp > nz)
        return fail(LSF_ERR_INVALID, "search box leaves the grid (the reference would write outside phi)");
    // centroids (set3d.f90:199-215) and per-triangle vertex coordinates; surfX is (nSurfNode,3) and surfElem
    // (nSurfElem,3), both Fortran-ordered, connectivity 1-based
    std::vector<double> cen((size_t)nSurfElem * 3), vtx((size_t)nSurfElem * 9);
    for (int n = 0; n < nSurfElem; ++n) {
        int id[3];
        for (int v = 0; v < 3; ++v) {
            id[v] = surfElem[(size_t)n + (size_t)nSurfElem * v];
            if (id[v] < 1 || id[v] > nSurfNode) return fail(LSF_ERR_INVALID, "surfElem index out of range");
        }
        for (int c = 0; c < 3; ++c) {
            const double p1 = surfX[(size_t)(id[0] - 1) + (size_t)nSurfNode * c];
            const double p2 = surfX[(size_t)(id[1] - 1) + (size_t)nSurfNode * c];
            const double p3 = surfX[(size_t)(id[2] - 1) + (size_t)nSurfNode * c];
            cen[(size_t)n * 3 + c] = (p1 + p2 + p3) / 3.;
            vtx[(size_t)n * 9 + c] = p1;
            vtx[(size_t)n * 9 + 3 + c] = p2;
            vtx[(size_t)n * 9 + 6 + c] = p3;
        }
    }
    Ctx& c = ctx();
    if ((rc = ws(c.slot[S_CEN], cen.size() * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_VTX], vtx.size() * sizeof(double)))) return rc;
    HIPCHK(hipMemcpyAsync(c.slot[S_CEN].p, cen.data(), cen.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c.slot[S_VTX].p, vtx.data(), vtx.size() * sizeof(double), hipMemcpyHostToDevice, st));
    const long n = (long)(nx + 1) * (ny + 1) * (nz + 1);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, st, d_phi, n,
                       1.0); // phi = 1., set3d.f90:161
    const long npts = (long)(ip - im + 1) * (jp - jm + 1) * (kp - km + 1);
    hipLaunchKernelGGL(k_phi0, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, d_phi, nx, ny, im, ip, jm, jp,
                       km, kp, dx, xLo[0], xLo[1], xLo[2], (const double*)c.slot[S_CEN].p,
                       (const double*)c.slot[S_VTX].p, nSurfElem);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // cen/vtx are host temporaries
    return LSF_OK;
}

int lsf_phi0(double* phi, int nx, int ny, int nz, double dx, const double xLo[3], const double minX[3],
             const double maxX[3], const double* surfX, int nSurfNode, const int32_t* surfElem, int nSurfElem)
{
    Trace trace_("lsf_phi0");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    Ctx& c = ctx();
    const size_t bytes = (size_t)(nx + 1) * (ny + 1) * (nz + 1) * sizeof(double);
    if ((rc = twin_claim(c, c.twin_phi, S_HPHI, phi, bytes))) return rc; // (an un-synced result of another array goes home first)
    twin_drop(c.twin_phi);                                                // phi is an output here
    rc = lsf_phi0_device((double*)c.slot[S_HPHI].p, nx, ny, nz, dx, xLo, minX, maxX, surfX, nSurfNode, surfElem,
                         nSurfElem, nullptr);
    if (rc) return rc;
    return twin_out(c, c.twin_phi, S_HPHI, phi, bytes);
}

int lsf_advect_nodes_device(const double* d_phi, const int32_t* d_phiSB, int nx, int ny, int nz, double dx,
                            const double xLo[3], double* surfXX, int nSurfNode, int iters, void* stream)
{
    Trace trace_("lsf_advect_nodes_device");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!d_phi || !d_phiSB || !xLo || !surfXX) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (nSurfNode < 1 || iters < 0) return fail(LSF_ERR_INVALID, "bad node count / iteration count");
    hipStream_t st = (hipStream_t)stream;
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    // every node must sit in a cell whose 8 corners exist (the reference would read outside phi otherwise)
    for (int q = 0; q < nSurfNode; ++q)
        for (int ax = 0; ax < 3; ++ax) {
            const double v = surfXX[(size_t)q + (size_t)nSurfNode * ax];
            const int nn = ax == 0 ? nx : (ax == 1 ? ny : nz);
            if (!(v >= xLo[ax] && v < xLo[ax] + dx * (nn - 1)))
                return fail(LSF_ERR_INVALID, "surface node outside the grid");
        }
    if ((rc = ws(c.slot[S_GRAD], 3 * n * sizeof(double)))) return rc;
    if ((rc = ws(c.slot[S_NODES], (size_t)nSurfNode * 3 * sizeof(double)))) return rc;
    double* grad = (double*)c.slot[S_GRAD].p;
    double* nodes = (double*)c.slot[S_NODES].p;
    HIPCHK(hipMemcpyAsync(nodes, surfXX, (size_t)nSurfNode * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_firstderiv8, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, st, d_phi,
                       d_phiSB, grad, nx, ny, nz, dx);
    hipLaunchKernelGGL(k_advect_nodes, dim3((unsigned)cdiv(nSurfNode, 64)), dim3(64), 0, st, d_phi, (const double*)grad,
                       nx, ny, nz, dx, xLo[0], xLo[1], xLo[2], nodes, nSurfNode, iters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(surfXX, nodes, (size_t)nSurfNode * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return LSF_OK;
}

int lsf_advect_nodes(const double* phi, const int32_t* phiSB, int nx, int ny, int nz, double dx, const double xLo[3],
                     double* surfXX, int nSurfNode, int iters)
{
    Trace trace_("lsf_advect_nodes");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi || !phiSB) return fail(LSF_ERR_INVALID, "NULL field");
    Ctx& c = ctx();
    const size_t n = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    if ((rc = twin_in(c, c.twin_phi, S_HPHI, phi, n * sizeof(double)))) return rc;
    if ((rc = twin_in(c, c.twin_sb, S_HSB, phiSB, n * sizeof(int32_t)))) return rc;
    return lsf_advect_nodes_device((const double*)c.slot[S_HPHI].p, (const int32_t*)c.slot[S_HSB].p, nx, ny, nz, dx, xLo,
                                   surfXX, nSurfNode, iters, nullptr);
}

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
        const size_t CH = 64u << 20;
        void* pin[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        if (hipHostMalloc(&pin[0], CH, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&pin[1], CH, hipHostMallocDefault) != hipSuccess ||
            hipStreamCreate(&st) != hipSuccess || hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
            ok = false;
        } else {
            const size_t nch = (bytes + CH - 1) / CH;
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

int lsf_jacobi_sweep_box(const double* d_in, double* d_out, const double* d_phiS, const lsf_box* box,
                         const int lo[3], const int hi[3], double dx, double h, int mode, double* d_sumsq,
                         void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = box_ok(box, lo, hi))) return rc;
    if (!d_in || !d_out || !d_phiS || !d_sumsq) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) return LSF_OK; // empty region
    if ((rc = sweep_region_ok(box, lo, hi))) return rc;
    hipStream_t st = (hipStream_t)stream;
    // regions a few cells wide in x (the x rim of a decomposed sweep) run with the lanes along y (jacobi_plan)
    const bool strict = (mode & LSF_ARITH_STRICT) != 0;
    const JacPlan jp = jacobi_plan(lo, hi, strict);
    const long np = jp.nparts;
    double* part = nullptr;
    if ((rc = stream_partials(st, (size_t)np, &part))) return rc;
    const Box bx{box->lx, box->ly, box->lz, box->gx0, box->gy0, box->gz0, box->nx, box->ny, box->nz};
    jacobi_launch(jp, strict, d_in, d_out, d_phiS, bx, lo, hi, dx, h, part, nullptr, st);
    return finish_partials(st, part, np, d_sumsq);
}

int lsf_sumsq_begin(void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    StreamPart& sp = ctx().part_by_stream[(hipStream_t)stream];
    if (sp.deferred) return fail(LSF_ERR_INVALID, "lsf_sumsq_begin: already open on this stream");
    sp.deferred = true, sp.used = 0, sp.target = nullptr;
    return LSF_OK;
}

int lsf_sumsq_end(void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    StreamPart& sp = ctx().part_by_stream[(hipStream_t)stream];
    if (!sp.deferred) return fail(LSF_ERR_INVALID, "lsf_sumsq_end without lsf_sumsq_begin on this stream");
    sp.deferred = false;
    rc = flush_partials((hipStream_t)stream, sp);
    sp.target = nullptr;
    return rc;
}

int lsf_bc_box(const double* d_in, double* d_out, const lsf_box* box, const int lo[3], const int hi[3],
               double dx, double* d_sumsq, void* stream)
{
    return bc_box_impl<double>(d_in, d_out, box, lo, hi, dx, d_sumsq, stream);
}

int lsf_bc_box_f32(const float* d_in, float* d_out, const lsf_box* box, const int lo[3], const int hi[3],
                   double dx, double* d_sumsq, void* stream)
{
    return bc_box_impl<float>(d_in, d_out, box, lo, hi, dx, d_sumsq, stream);
}

int lsf_pack_box(const double* d_field, const lsf_box* box, const int lo[3], const int hi[3], double* d_buf,
                 void* stream)
{
    return pack_impl<double>(d_field, nullptr, box, lo, hi, d_buf, 0, stream);
}

int lsf_unpack_box(double* d_field, const lsf_box* box, const int lo[3], const int hi[3], const double* d_buf,
                   void* stream)
{
    return pack_impl<double>(nullptr, d_field, box, lo, hi, const_cast<double*>(d_buf), 1, stream);
}

int lsf_pack_box_f32(const float* d_field, const lsf_box* box, const int lo[3], const int hi[3], float* d_buf,
                     void* stream)
{
    return pack_impl<float>(d_field, nullptr, box, lo, hi, d_buf, 0, stream);
}

int lsf_unpack_box_f32(float* d_field, const lsf_box* box, const int lo[3], const int hi[3], const float* d_buf,
                       void* stream)
{
    return pack_impl<float>(nullptr, d_field, box, lo, hi, const_cast<float*>(d_buf), 1, stream);
}

// ---- fp32 Jacobi path (BASELINE configuration 5) -------------------------------------------------
int lsf_jacobi_sweep_box_f32(const float* d_in, float* d_out, const float* d_phiS, const lsf_box* box,
                             const int lo[3], const int hi[3], double dx, double h, int mode, double* d_sumsq,
                             void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = f32_mode_ok(mode))) return rc;
    if ((rc = box_ok(box, lo, hi))) return rc;
    if (!d_in || !d_out || !d_phiS || !d_sumsq) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (hi[0] <= lo[0] || hi[1] <= lo[1] || hi[2] <= lo[2]) return LSF_OK;
    if ((rc = sweep_region_ok(box, lo, hi))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const JacPlan jp = jacobi_plan_f32(lo, hi);
    const long np = jp.nparts;
    double* part = nullptr;
    if ((rc = stream_partials(st, (size_t)np, &part))) return rc;
    const Box bx{box->lx, box->ly, box->lz, box->gx0, box->gy0, box->gz0, box->nx, box->ny, box->nz};
    jacobi_launch_f32(jp, d_in, d_out, d_phiS, bx, lo, hi, dx, h, part, nullptr, 0, st);
    return finish_partials(st, part, np, d_sumsq);
}

int lsf_reinit_f32_device(float* d_phi, const float* d_phiS, int nx, int ny, int nz, int iter, double dx, double h,
                          double tol, int mode, int* sweeps_done, double* rms_trace, int trace_cap, void* stream)
{
    Trace trace_("lsf_reinit_f32_device");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = f32_mode_ok(mode))) return rc;
    return reinit_f32_core(d_phi, d_phiS, nx, ny, nz, iter, dx, h, tol, sweeps_done, rms_trace, trace_cap,
                           (hipStream_t)stream);
}

int lsf_reinit_f32(float* phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
                   int* sweeps_done, double* rms_trace, int trace_cap)
{
    Trace trace_("lsf_reinit_f32");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = f32_mode_ok(mode))) return rc;
    if ((rc = check_dims(nx, ny, nz))) return rc;
    if (!phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    Ctx& c = ctx();
    const size_t bytes = (size_t)(nx + 1) * (ny + 1) * (nz + 1) * sizeof(float);
    // the fp32 seam borrows the slot of the fp64 field twin: whatever twin lives there is written home if need be and
    // forgotten (float data must never be taken for the doubles of an earlier array at this address)
    if ((rc = twin_claim(c, c.twin_phi, S_HPHI, nullptr, bytes))) return rc;
    twin_drop(c.twin_phi);
    float* d = (float*)c.slot[S_HPHI].p;
    HIPCHK(hipMemcpy(d, phi, bytes, hipMemcpyHostToDevice));
    rc = reinit_f32_core(d, nullptr, nx, ny, nz, iter, dx, h, tol, sweeps_done, rms_trace, trace_cap, nullptr);
    if (rc == LSF_OK || rc == LSF_ERR_NAN) {
        const std::string keep = g_err;
        HIPCHK(hipMemcpy(phi, d, bytes, hipMemcpyDeviceToHost));
        g_err = keep;
    }
    return rc;
}

// ---- binary STL with the reference's vertex merge (subs.f90:17-121) --------------------------------------------
namespace {
struct StlResult {
    std::vector<float> nodes;   // 3 per node
    std::vector<int32_t> elem;  // 3 per triangle, 1-based
};
thread_local StlResult g_stl;
} // namespace

int lsf_stl_read(const char* path, int* nSurfElem, int* nSurfNode)
{
    Trace trace_("lsf_stl_read");
    if (!path || !nSurfElem || !nSurfNode) return fail(LSF_ERR_INVALID, "NULL pointer");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LSF_ERR_INVALID, std::string("cannot open ") + path);
    unsigned char head[84];
    if (fread(head, 1, 84, f) != 84) {
        fclose(f);
        return fail(LSF_ERR_INVALID, "STL file shorter than its header");
    }
    int32_t ntri = 0;
    std::memcpy(&ntri, head + 80, 4); // subs.f90:38-39
    if (ntri < 1) {
        fclose(f);
        return fail(LSF_ERR_INVALID, "STL file holds no triangle");
    }
    std::vector<unsigned char> rec((size_t)ntri * 50); // normal, 3 vertices (REAL*4), INTEGER*2 padding: subs.f90:47-53
    const size_t got = fread(rec.data(), 1, rec.size(), f);
    fclose(f);
    if (got != rec.size()) return fail(LSF_ERR_INVALID, "STL file shorter than its triangle count");
    StlResult& R = g_stl;
    R.nodes.clear();
    R.elem.assign((size_t)ntri * 3, 0);
    // Merge (subs.f90:64-93).  Two REAL*4 values can differ by less than 1e-13 without being equal only below 2^-19
    // (above it neighbouring floats are >= 1.1e-13 apart, also across that threshold), so a coordinate is keyed by its
    // bits when it is large and by one shared key when it is small; candidates of a key are kept in node order and tested
    // with the reference's own predicate, the first one inside the search bound wins.
    struct Key {
        uint32_t a, b, c;
        bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const { return ((size_t)k.a * 0x9E3779B1u) ^ ((size_t)k.b * 0x85EBCA77u << 1) ^ ((size_t)k.c * 0xC2B2AE3Du << 2); }
    };
    auto key1 = [](float v) -> uint32_t {
        if (std::fabs(v) < 1.9073486328125e-06f) return 0xFFFFFFFFu; // 2^-19
        uint32_t u;
        std::memcpy(&u, &v, 4);
        return u == 0x80000000u ? 0u : u;
    };
    std::unordered_map<Key, std::vector<int32_t>, KeyHash> map;
    map.reserve((size_t)ntri);
    int32_t bound = 3, k = 0; // nSurfNode (search bound) and the number of nodes so far
    for (int32_t n = 0; n < ntri; ++n) {
        for (int p = 0; p < 3; ++p) {
            float v[3];
            std::memcpy(v, rec.data() + (size_t)n * 50 + 12 + 12 * p, 12);
            const Key key{key1(v[0]), key1(v[1]), key1(v[2])};
            int32_t share = 0;
            auto it = map.find(key);
            if (it != map.end())
                for (int32_t cand : it->second) { // ascending node numbers
                    if (cand > bound) break;
                    const float* q = &R.nodes[(size_t)(cand - 1) * 3];
                    if ((double)std::fabs(q[0] - v[0]) < 1.e-13 && (double)std::fabs(q[1] - v[1]) < 1.e-13 &&
                        (double)std::fabs(q[2] - v[2]) < 1.e-13) {
                        share = cand;
                        break;
                    }
                }
            if (share > 0) {
                R.elem[(size_t)n * 3 + p] = share;
            } else {
                ++k;
                R.nodes.insert(R.nodes.end(), v, v + 3);
                R.elem[(size_t)n * 3 + p] = k;
                map[key].push_back(k);
            }
        }
        bound = k; // subs.f90:91
    }
    *nSurfElem = ntri;
    *nSurfNode = k;
    return LSF_OK;
}

int lsf_stl_get(double* surfX, int32_t* surfElem)
{
    if (!surfX || !surfElem) return fail(LSF_ERR_INVALID, "NULL pointer");
    StlResult& R = g_stl;
    if (R.elem.empty()) return fail(LSF_ERR_INVALID, "lsf_stl_get without lsf_stl_read");
    const size_t nn = R.nodes.size() / 3, nt = R.elem.size() / 3;
    for (size_t q = 0; q < nn; ++q)
        for (int c = 0; c < 3; ++c) surfX[q + nn * c] = (double)R.nodes[q * 3 + c]; // REAL*4 -> REAL(8), subs.f90:99-103
    for (size_t t = 0; t < nt; ++t)
        for (int p = 0; p < 3; ++p) surfElem[t + nt * p] = R.elem[t * 3 + p];
    R = StlResult{};
    return LSF_OK;
}

// ---- lsf_box_reserve --------------------------------------------------------------------------------
int lsf_box_reserve(void* stream, size_t max_partials)
{
    int rc = ensure_device();
    if (rc) return rc;
    StreamPart& sp = ctx().part_by_stream[(hipStream_t)stream];
    if (sp.deferred) return fail(LSF_ERR_INVALID, "lsf_box_reserve inside a lsf_sumsq bracket");
    if (max_partials * sizeof(double) > sp.buf.bytes) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        if ((rc = ws(sp.buf, max_partials * sizeof(double)))) return rc;
    }
    return LSF_OK;
}

} // extern "C"

// ---- one process, every GPU: lsf_multi_* / lsf_reinit_multi (lsf_multi.hpp) ------------------------------
#include "lsf_multi.hpp"

namespace lsfm {
bool Rccl::load(std::string* err)
{
    if (lib) return true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) {
        *err = std::string("RCCL transport requested but librccl.so cannot be loaded: ") + dlerror();
        return false;
    }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    GetVersion = (int (*)(int*))sym("ncclGetVersion");
    GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
    CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    GroupStart = (int (*)())sym("ncclGroupStart");
    GroupEnd = (int (*)())sym("ncclGroupEnd");
    Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
    Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
    if (!GetErrorString || !CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv) {
        *err = "librccl.so lacks a symbol of the point-to-point API";
        dlclose(lib);
        lib = nullptr;
        return false;
    }
    if (GetVersion) (void)GetVersion(&version);
    return true;
}
Rccl::~Rccl()
{
    for (void* c : comms)
        if (c && CommDestroy) (void)CommDestroy(c);
    comms.clear();
    // the library stays loaded: RCCL keeps threads and device state of its own
}
} // namespace lsfm

// process-wide defaults of lsf_multi_create (lsf_multi_defaults; LSF_MULTI_TRANSPORT = peer | rccl | mock and
// LSF_MULTI_CHECK_EVERY in the environment override them: the Fortran host has no other way in)
// (per thread: a caller that sets them around a call of its own does not race with other threads' calls)
static thread_local int g_multi_check_every = 8, g_multi_transport = LSF_TRANSPORT_PEER;

// the lsf_multi_* calls visit other devices: the calling thread gets its own device back (HIP's and the library's)
struct DeviceRestore {
    int hip_dev = -1, lsf_dev = 0;
    DeviceRestore() : lsf_dev(g_device) { if (hipGetDevice(&hip_dev) != hipSuccess) hip_dev = -1; (void)hipGetLastError(); }
    ~DeviceRestore()
    {
        g_device = lsf_dev;
        if (hip_dev >= 0) (void)hipSetDevice(hip_dev);
    }
};

// ---- first-contact self-test of the device-to-device hand-offs the slab launches rely on (lsf_peer_selftest) ------------
#include "lsf_peer.hpp"

// ---- the exact ordering across z slabs, one launch per device (lsf_reinit_multi with LSF_ORDER_GS) --------------------
#include "lsf_gs_slabs.hpp"

extern "C" {

int lsf_multi_defaults(int check_every, int transport)
{
    if (check_every < 1 || check_every > lsfm::MAX_CHECK) return fail(LSF_ERR_INVALID, "check_every must be 1..64");
    if (transport != LSF_TRANSPORT_PEER && transport != LSF_TRANSPORT_RCCL && transport != LSF_TRANSPORT_MOCK)
        return fail(LSF_ERR_INVALID, "unknown transport");
    g_multi_check_every = check_every, g_multi_transport = transport;
    return LSF_OK;
}

int lsf_multi_defaults_get(int* check_every, int* transport)
{
    if (check_every) *check_every = g_multi_check_every;
    if (transport) *transport = g_multi_transport;
    return LSF_OK;
}

int lsf_multi_configure(lsf_multi* M, int check_every, int transport)
{
    if (!M) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (check_every < 1 || check_every > lsfm::MAX_CHECK) return fail(LSF_ERR_INVALID, "check_every must be 1..64");
    if (transport != LSF_TRANSPORT_PEER && transport != LSF_TRANSPORT_RCCL && transport != LSF_TRANSPORT_MOCK)
        return fail(LSF_ERR_INVALID, "unknown transport");
    if (transport == LSF_TRANSPORT_RCCL && M->rccl.comms.empty()) {
        // one communicator per block; RCCL wants every rank of a process on a device of its own
        for (int a = 0; a < M->ndev; ++a)
            for (int b = a + 1; b < M->ndev; ++b)
                if (M->devs[a] == M->devs[b])
                    return fail(LSF_ERR_INVALID, "the RCCL transport needs a distinct device per block (use the peer transport to share a device)");
        std::string err;
        if (!M->rccl.load(&err)) return fail(LSF_ERR_HIP, err);
        int cur = -1;
        (void)hipGetDevice(&cur);
        M->rccl.comms.assign((size_t)M->ndev, nullptr);
        const int rc = M->rccl.CommInitAll(M->rccl.comms.data(), M->ndev, M->devs.data());
        if (cur >= 0) (void)hipSetDevice(cur);
        if (rc != 0) {
            M->rccl.comms.clear();
            return fail(LSF_ERR_HIP, std::string("ncclCommInitAll: ") + M->rccl.GetErrorString(rc));
        }
    }
    M->check_every = check_every, M->transport = transport;
    return LSF_OK;
}

int lsf_multi_info(const lsf_multi* M, int* check_every, int* transport, int* rccl_ranks, int* rccl_version, double* host_enqueue_s,
                   double* host_calls_s, double* wall_s, int* sweeps_enqueued)
{
    if (!M) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (check_every) *check_every = M->check_every;
    if (transport) *transport = M->transport;
    if (rccl_ranks) *rccl_ranks = (int)M->rccl.comms.size();
    if (rccl_version) *rccl_version = M->rccl.version;
    double he = 0.0;
    for (auto& R : M->r64) he = std::max(he, R.host_enqueue_s);
    for (auto& R : M->r32) he = std::max(he, R.host_enqueue_s);
    double hc = 0.0;
    for (auto& R : M->r64) hc = std::max(hc, R.host_calls_s);
    for (auto& R : M->r32) hc = std::max(hc, R.host_calls_s);
    if (host_calls_s) *host_calls_s = hc;
    if (host_enqueue_s) *host_enqueue_s = he;
    if (wall_s) *wall_s = M->last_wall_s;
    if (sweeps_enqueued) *sweeps_enqueued = M->last_sweeps_enqueued;
    return LSF_OK;
}

int lsf_multi_create(int nx, int ny, int nz, const int* devices, int ndev, const int dims_in[3], int f32, lsf_multi** out)
{
    DeviceRestore restore_;
    Trace trace_("lsf_multi_create");
    if (!out) return fail(LSF_ERR_INVALID, "NULL pointer");
    *out = nullptr;
    int rc = check_dims(nx, ny, nz);
    if (rc) return rc;
    if (ndev < 1 || ndev > 64 || !devices) return fail(LSF_ERR_INVALID, "device list must hold 1..64 entries");
    int ndevs = 0;
    if (hipGetDeviceCount(&ndevs) != hipSuccess || ndevs <= 0) {
        (void)hipGetLastError();
        return fail(LSF_ERR_NO_DEVICE, "no HIP device visible: liblsf_hip has no CPU fallback");
    }
    for (int r = 0; r < ndev; ++r)
        if (devices[r] < 0 || devices[r] >= ndevs) return fail(LSF_ERR_NO_DEVICE, "device index out of range");
    int dims[3];
    if (dims_in) {
        for (int a = 0; a < 3; ++a) dims[a] = dims_in[a];
        if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || dims[0] * dims[1] * dims[2] != ndev)
            return fail(LSF_ERR_INVALID, "dims must multiply to the number of devices");
    } else
        lsfm::default_dims(ndev, dims);
    lsf_multi* M = new lsf_multi(ndev);
    M->nx = nx, M->ny = ny, M->nz = nz, M->f32 = f32 ? 1 : 0;
    for (int a = 0; a < 3; ++a) M->dims[a] = dims[a];
    M->devs.assign(devices, devices + ndev);
    const int n[3] = {nx, ny, nz};
    std::string err;
    auto build = [&](auto& ranks) -> int {
        ranks.resize(ndev);
        for (int r = 0; r < ndev; ++r) {
            ranks[r].dev = devices[r];
            if (!lsfm::make_geom(r, dims, n, &ranks[r].g, &err)) return fail(LSF_ERR_INVALID, err);
            if ((double)ranks[r].g.ext[0] * ranks[r].g.ext[1] * 8.0 > 2.0e9) return fail(LSF_ERR_INVALID, "a k-plane of a block exceeds 2 GB");
        }
        // direct peer copies between distinct devices (already-enabled is not an error)
        for (int a = 0; a < ndev; ++a)
            for (int b = 0; b < ndev; ++b)
                if (devices[a] != devices[b]) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can && hipSetDevice(devices[a]) == hipSuccess)
                        (void)hipDeviceEnablePeerAccess(devices[b], 0);
                    (void)hipGetLastError();
                }
        for (int r = 0; r < ndev; ++r) {
            lsfm::alloc_rank(ranks[r]);
            if (ranks[r].rc) return fail(ranks[r].rc, ranks[r].err);
        }
        return LSF_OK;
    };
    rc = f32 ? build(M->r32) : build(M->r64);
    if (!rc) {
        int ce = g_multi_check_every, tp = g_multi_transport;
        if (const char* e = getenv("LSF_MULTI_CHECK_EVERY")) ce = std::min(std::max(atoi(e), 1), (int)lsfm::MAX_CHECK);
        if (const char* e = getenv("LSF_MULTI_TRANSPORT"))
            tp = !std::strcmp(e, "rccl") ? LSF_TRANSPORT_RCCL : (!std::strcmp(e, "mock") ? LSF_TRANSPORT_MOCK : LSF_TRANSPORT_PEER);
        if (const char* e = getenv("LSF_MULTI_GRAPHS")) M->graphs = atoi(e) != 0; // default off: measured slower, see lsf_multi.hpp
        rc = lsf_multi_configure(M, ce, tp);
    }
    if (rc) {
        const std::string keep = g_err;
        lsf_multi_destroy(M);
        g_err = keep;
        return rc;
    }
    *out = M;
    return LSF_OK;
}

int lsf_multi_destroy(lsf_multi* M)
{
    DeviceRestore restore_;
    if (!M) return LSF_OK;
    for (auto& R : M->r64) lsfm::free_rank(R);
    for (auto& R : M->r32) lsfm::free_rank(R);
    delete M;
    return LSF_OK;
}

int lsf_multi_block(const lsf_multi* M, int r, int g0[3], int ext[3], int own_lo[3], int own_hi[3], int* device)
{
    if (!M || r < 0 || r >= M->ndev) return fail(LSF_ERR_INVALID, "bad block index");
    const lsfm::Geom& g = M->f32 ? M->r32[r].g : M->r64[r].g;
    for (int a = 0; a < 3; ++a) {
        if (g0) g0[a] = g.g0[a];
        if (ext) ext[a] = g.ext[a];
        if (own_lo) own_lo[a] = g.own[a][0];
        if (own_hi) own_hi[a] = g.own[a][1];
    }
    if (device) *device = M->devs[r];
    return LSF_OK;
}

int lsf_multi_scatter(lsf_multi* M, const void* host_phi)
{
    DeviceRestore restore_;
    if (!M || !host_phi) return fail(LSF_ERR_INVALID, "NULL pointer");
    std::string err;
    const int rc = M->f32 ? lsfm::scatter(M->r32, (const float*)host_phi, M->nx, M->ny, &err)
                          : lsfm::scatter(M->r64, (const double*)host_phi, M->nx, M->ny, &err);
    M->result_parity = 0;
    return rc ? fail(rc, err) : LSF_OK;
}

int lsf_multi_upload_block(lsf_multi* M, int r, const void* d_block)
{
    DeviceRestore restore_;
    if (!M || !d_block || r < 0 || r >= M->ndev) return fail(LSF_ERR_INVALID, "bad block index / NULL pointer");
    HIPCHK(hipSetDevice(M->devs[r]));
    if (M->f32) HIPCHK(hipMemcpy(M->r32[r].buf[0], d_block, M->r32[r].g.npoints() * sizeof(float), hipMemcpyDeviceToDevice));
    else HIPCHK(hipMemcpy(M->r64[r].buf[0], d_block, M->r64[r].g.npoints() * sizeof(double), hipMemcpyDeviceToDevice));
    M->result_parity = 0;
    return LSF_OK;
}

int lsf_multi_run(lsf_multi* M, int iter, double dx, double h, double tol, int mode, int* sweeps_done, double* rms_trace, int trace_cap)
{
    DeviceRestore restore_;
    Trace trace_("lsf_multi_run");
    if (!M) return fail(LSF_ERR_INVALID, "NULL pointer");
    if (iter < 0) return fail(LSF_ERR_INVALID, "iter must be >= 0");
    if ((mode & LSF_ORDER_MASK) != LSF_ORDER_JACOBI)
        return fail(LSF_ERR_INVALID, "lsf_multi: LSF_ORDER_JACOBI only (the reference's in-place ordering does not shard)");
    if (M->f32) {
        const int rc = f32_mode_ok(mode);
        if (rc) return rc;
    }
    // the sweeps start from buf[0]: bring the latest field there if the previous run ended on the other buffer
    if (M->result_parity) {
        for (int r = 0; r < M->ndev; ++r) {
            HIPCHK(hipSetDevice(M->devs[r]));
            if (M->f32) std::swap(M->r32[r].buf[0], M->r32[r].buf[1]);
            else std::swap(M->r64[r].buf[0], M->r64[r].buf[1]);
        }
        M->result_parity = 0;
    }
    std::string err;
    const int rc = M->f32 ? lsfm::run(M, M->r32, iter, dx, h, tol, mode, sweeps_done, rms_trace, trace_cap, &err)
                          : lsfm::run(M, M->r64, iter, dx, h, tol, mode, sweeps_done, rms_trace, trace_cap, &err);
    return rc ? fail(rc, err) : LSF_OK;
}

int lsf_multi_gather(lsf_multi* M, void* host_phi)
{
    DeviceRestore restore_;
    if (!M || !host_phi) return fail(LSF_ERR_INVALID, "NULL pointer");
    std::string err;
    const int rc = M->f32 ? lsfm::gather(M->r32, M->result_parity, (float*)host_phi, M->nx, M->ny, &err)
                          : lsfm::gather(M->r64, M->result_parity, (double*)host_phi, M->nx, M->ny, &err);
    return rc ? fail(rc, err) : LSF_OK;
}

static int reinit_multi_any(void* phi, int f32, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode,
                            const int* devices, int ndev, const int dims[3], int* sweeps_done, double* rms_trace, int trace_cap)
{
    if (!phi) return fail(LSF_ERR_INVALID, "phi is NULL");
    if (!devices || ndev < 1) return fail(LSF_ERR_INVALID, "empty device list");
    const bool exact = (mode & LSF_ORDER_MASK) == LSF_ORDER_GS;
    if (exact && f32) return fail(LSF_ERR_INVALID, "single precision: LSF_ORDER_JACOBI | LSF_ARITH_FAST only");
    if (exact && dims && (dims[0] != 1 || dims[1] != 1 || dims[2] != ndev))
        return fail(LSF_ERR_INVALID, "LSF_ORDER_GS shards into z slabs: dims must be NULL or {1, 1, ndev}");
    lsf_multi* M = nullptr;
    int rc = LSF_OK;
    // The blocks are scattered from and gathered into the HOST array.  If an earlier seam call left the latest content of
    // this array in its device twin (lsf_mirror LAZY), bring it home first; afterwards the host copy is the newer one, so
    // the twin no longer counts (the next seam call uploads again).
    Twin* tw = nullptr;
    if (ensure_device() == LSF_OK) {
        Slot sl = S_HPHI;
        if (twin_of(ctx(), phi, 0, &tw, &sl) && tw->host_stale) {
            HIPCHK(hipMemcpy(phi, ctx().slot[sl].p, tw->bytes, hipMemcpyDeviceToHost));
            tw->host_stale = false;
        }
    }
    if (exact) { // the reference's ordering, slab by slab (lsf_gs_slabs.hpp): same field as lsf_reinit, bit for bit
        if (tw) tw->current = false;
        if (iter < 0) return fail(LSF_ERR_INVALID, "iter must be >= 0");
        return reinit_gs_slabs((double*)phi, nx, ny, nz, iter, dx, h, tol, mode, devices, ndev, sweeps_done, rms_trace, trace_cap);
    }
    rc = lsf_multi_create(nx, ny, nz, devices, ndev, dims, f32, &M);
    if (rc) return rc;
    if (tw) tw->current = false;
    rc = lsf_multi_scatter(M, phi);
    if (!rc) rc = lsf_multi_run(M, iter, dx, h, tol, mode, sweeps_done, rms_trace, trace_cap);
    if (rc == LSF_OK || rc == LSF_ERR_NAN) {
        const std::string keep = g_err;
        const int rc2 = lsf_multi_gather(M, phi);
        if (rc2) rc = rc2;
        else g_err = keep;
    }
    const std::string keep = g_err;
    lsf_multi_destroy(M);
    g_err = keep;
    return rc;
}

int lsf_reinit_multi(double* phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode, const int* devices,
                     int ndev, const int dims[3], int* sweeps_done, double* rms_trace, int trace_cap)
{
    Trace trace_("lsf_reinit_multi");
    return reinit_multi_any(phi, 0, nx, ny, nz, iter, dx, h, tol, mode, devices, ndev, dims, sweeps_done, rms_trace, trace_cap);
}

int lsf_slabs_info(int* slabs, int* blocks_per_slab, int* finegrained, int* sweeps, double* kernel_s)
{
    const lsfs::SlabReport& r = lsfs::g_slab_report;
    if (slabs) *slabs = r.slabs;
    if (blocks_per_slab) *blocks_per_slab = r.grid;
    if (finegrained) *finegrained = r.fine;
    if (sweeps) *sweeps = r.sweeps;
    if (kernel_s) *kernel_s = r.kernel_ms * 1e-3;
    return LSF_OK;
}

int lsf_reinit_multi_f32(float* phi, int nx, int ny, int nz, int iter, double dx, double h, double tol, int mode, const int* devices,
                         int ndev, const int dims[3], int* sweeps_done, double* rms_trace, int trace_cap)
{
    Trace trace_("lsf_reinit_multi_f32");
    return reinit_multi_any(phi, 1, nx, ny, nz, iter, dx, h, tol, mode, devices, ndev, dims, sweeps_done, rms_trace, trace_cap);
}

} // extern "C"
