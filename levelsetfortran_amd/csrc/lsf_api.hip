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
#include "lsf_minmax_band.hpp"

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
// marching steps of a skewed tile: 16.  Tiles of 32 steps (three lanes per cell, 2 x 2 wavefronts; skew_tile and the host code
// below are written for both) were built in round 5 for grids below 384 cells across, where a sweep costs its chain of dependent
// tiles: consecutive sweeps are spaced by n / TA time slots (+ the tiles across where a cross-section axis flips), so a tile twice
// as long was expected to trade 34.5 hand-offs per 256^3 sweep for 24.5 slightly longer ones.  Measured (profiles/r05_ta32_ab.txt,
// one box, bit-identical): SLOWER at every size from 64^3 to 512^3 -- 256^3 0.591 -> 0.673 ms per sweep FAST, 0.928 -> 1.176
// STRICT -- a hand-off carries the tile's whole load and write back, which double with the tile.  Experiment builds only
// (LSF_GS_SKEW_TA=32): the product library holds no such kernel.
int gs_skew_ta(int cells_across, int wy, int wz, int by)
{
#ifdef LSF_EXPERIMENTS
    if (by == 5 && wy == 2 && wz == 2)
        if (const char* e = getenv("LSF_GS_SKEW_TA")) return atoi(e) == 32 ? 32 : 16;
#endif
    (void)cells_across, (void)wy, (void)wz, (void)by;
    return 16;
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

enum Slot { S_PONG, S_PHIS, S_PART, S_CTL, S_TRACE, S_HPHI, S_HNB, S_HSB, S_CEN, S_VTX, S_BFLAG, S_CHG, S_BACKUP, S_PART2, S_PLANECNT, S_DBG, S_COLSUM, S_ORDER, S_GRAD, S_NODES, S_STAMP, S_PONG2, S_PONG3, S_PONG4, S_SNAP, S_MB_CNT, S_MB_L, S_MB_NB6, S_MB_AOLD, S_MB_A0, S_MB_BAND, S_MB_KEY, S_MB_TMP, S_NSLOTS };

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
    // min/max flow on the band (lsf_host_minmax.hpp): blocks of its looping launch (resident as a whole: occupancy x CUs, at most
    // one per CU; 0 = not asked yet) and whether that launch has ever timed out on this device (then the dense executors run)
    int mm_tail_blocks = 0;
    bool mm_band_off = false;
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
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Skewed tiles (lsf_skew.hpp): (m, fB, fC) with TA m <= Fx + Fy + Fz < TA m + TA for some cell of row bundle
// (fB, fC); the m range assumes full bundles (NY x 4 rows), a superset for the partial bundles at the far walls
// (such a tile simply finds no cell to work on).  Sorted by hyperplane m + fB + fC.
int get_skew_tiles(int nxi, int nTj, int nTk, int ta, int nyc, int nzc, TileList** out)
{
    const int m_max = (nxi - 1 + nyc * nTj - 1 + nzc * nTk - 1) / ta;
    if (m_max > 1023 || nTj > 1023 || nTk > 1023) return fail(LSF_ERR_INVALID, "grid too large for tile index packing");
    const uint64_t key = ((uint64_t)nxi << 44) | ((uint64_t)nTj << 28) | ((uint64_t)nTk << 12) | (uint64_t)(nyc << 6 | nzc) | (ta == 32 ? 1u << 11 : 0u);
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
//   STRICT            k_reinit_jacobi_strict_sh (the reference's arithmetic, every difference evaluated once per point);
//                     k_reinit_jacobi<true> (per cell) with LSF_JAC_SH = 0
//   FAST              k_reinit_jacobi_sh<WX, BY> (WENO interfaces shared along x and z); WX = wavefronts a block
//                     spans along x, chosen so that the 64 WX - 1 cells of a block tile the row with the fewest wavefronts
//   3-cell x rims     k_reinit_jacobi<., true> (lanes along y)
// LSF_JAC_SH = 0 forces the per-cell kernel, "WXxBY" a block shape (measurement aids; all FAST choices are bit-identical).
struct JacPlan {
    int kind = 0; // 0 per-cell kernel, 1 per-cell THINX, 2 shared-interface kernel (FAST), 3 shared-difference kernel (STRICT)
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
JacPlan jacobi_plan(const int lo[3], const int hi[3], bool strict, const int ext[3])
{
    JacPlan p;
    const int cx = hi[0] - lo[0], cy = hi[1] - lo[1], cz = hi[2] - lo[2];
    // k_reinit_jacobi_strict_sh forms its buffer descriptors three doubles in front of a plane and clamps its operand offsets one
    // point inside the region's surroundings: every cell of the region needs a point of the LOCAL array on each side (true for
    // interior cells of the global grid in a box that holds its wall points / ghost layers -- asserted here, not assumed)
    bool inside = true;
    for (int d = 0; d < 3; ++d) inside = inside && lo[d] >= 1 && hi[d] <= ext[d] - 1;
    const bool thinx = cx <= 8 && cy >= 32;
    const char* env = getenv("LSF_JAC_SH");
    int fwx = 0, fby = 0;
    const bool off = env && env[0] == '0' && env[1] == 0;
    if (env && !off && sscanf(env, "%dx%d", &fwx, &fby) != 2) fwx = fby = 0;
    if (strict && !thinx && !off && inside) {
        p.kind = 3;
        const int gx = cdiv(cx, JSS_PX * JSS_WX), gy = cdiv(cy, JSS_PY * JSS_WY);
        p.kc = jacobi_kc((long)gx * gy, cz);
        p.nbx = gx, p.nby = gy, p.nbz = cdiv(cz, p.kc);
        p.nparts = (long)gx * gy * p.nbz;                    // one partial sum per logical block, in block order
        p.grid = dim3((unsigned)((p.nparts + 7) / 8 * 8));   // one-dimensional, padded to the 8 XCDs (k_reinit_jacobi_strict_sh)
        return p;
    }
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
    for (int wx : {4, 2, 1, 8}) { // on a tie the first wins: four wavefronts across measured 5-7 % faster than 2 x 2 or eight (256^3, the 506-cell cores of a decomposed 512^3 block)
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
    if (p.kind == 3) {
        hipLaunchKernelGGL(k_reinit_jacobi_strict_sh, p.grid, dim3(64 * JSS_WX * JSS_WY), 0, st, A, B, phiS, bx, lo[0], lo[1], lo[2], hi[0],
                           hi[1], hi[2], dx, h, part, done, p.kc, p.nbx, p.nby, p.nbz);
    } else if (p.kind == 2) {
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
    for (int wx : {4, 2, 1, 8}) { // on a tie the first wins: four wavefronts across measured 5-7 % faster than 2 x 2 or eight (256^3, the 506-cell cores of a decomposed 512^3 block)
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
    else {
        const int jext[3] = {nx + 1, ny + 1, nz + 1};
        jp = jacobi_plan(jlo, jhi, strict, jext);
    }
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
    else if (jp.kind == 3) snprintf(g_prof.kernel_buf, sizeof g_prof.kernel_buf, "k_reinit_jacobi_strict_sh");
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

#include "lsf_host_gs.hpp"

#include "lsf_host_minmax.hpp"

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

template <typename T>
int pack_multi_impl(const T* d_field, T* d_field_w, const lsf_box* box, int nreg, const int (*lo)[3], const int (*hi)[3], T* const* d_bufs,
                    int unpack, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (nreg < 0 || nreg > 6) return fail(LSF_ERR_INVALID, "0 to 6 sub-boxes");
    if (!box || !lo || !hi || !d_bufs || (!d_field && !d_field_w)) return fail(LSF_ERR_INVALID, "NULL pointer");
    PackRegs r;
    std::memset(&r, 0, sizeof r);
    long tot = 0;
    for (int q = 0; q < nreg; ++q) {
        if ((rc = box_ok(box, lo[q], hi[q]))) return rc;
        const int e0 = hi[q][0] - lo[q][0], e1 = hi[q][1] - lo[q][1], e2 = hi[q][2] - lo[q][2];
        if (e0 <= 0 || e1 <= 0 || e2 <= 0) continue;
        if (!d_bufs[q]) return fail(LSF_ERR_INVALID, "NULL pointer");
        if ((double)e0 * e1 * e2 > 2.0e9) return fail(LSF_ERR_INVALID, "a slab of more than 2e9 points");
        for (int a = 0; a < 3; ++a) r.lo[r.n][a] = lo[q][a], r.e[r.n][a] = hi[q][a] - lo[q][a];
        r.buf[r.n] = (void*)d_bufs[q];
        r.start[r.n] = tot;
        tot += (long)e0 * e1 * e2;
        ++r.n;
    }
    for (int q = r.n; q <= 6; ++q) r.start[q] = tot;
    if (tot == 0) return LSF_OK;
    const int grid = (int)std::min<long>((tot + 255) / 256, 8192);
    const Box bx{box->lx, box->ly, box->lz, box->gx0, box->gy0, box->gz0, box->nx, box->ny, box->nz};
    hipLaunchKernelGGL(k_pack_multi<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_field, d_field_w, bx, r, unpack);
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

int lsf_copy_bandwidth(size_t bytes, int reps, double* gbps)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!gbps || reps < 1 || bytes < (size_t)(1 << 20) || bytes % 16) return fail(LSF_ERR_INVALID, "lsf_copy_bandwidth: bytes >= 1 MiB, a multiple of 16; reps >= 1");
    uint4 *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc((void**)&a, bytes));
    if (hipMalloc((void**)&b, bytes) != hipSuccess) {
        (void)hipFree(a);
        return fail(LSF_ERR_HIP, "lsf_copy_bandwidth: out of device memory");
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float best = 0.f;
    rc = LSF_OK;
    do {
        if (hipMemset(a, 0x3c, bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            rc = fail(LSF_ERR_HIP, "lsf_copy_bandwidth: set-up failed");
            break;
        }
        const long n16 = (long)(bytes / 16);
        if (n16 / 256 + 1 > 0x7fffffffL) {
            rc = fail(LSF_ERR_INVALID, "lsf_copy_bandwidth: at most 2^31 blocks of 4 KB");
            break;
        }
        const dim3 grid((unsigned)((n16 + 255) / 256)); // one 16-byte vector per lane
        for (int r = 0; r <= reps; ++r) { // the first pass warms up (page tables, clocks) and is not timed
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, 0, (const uint4*)a, b, n16);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) {
                rc = fail(LSF_ERR_HIP, "lsf_copy_bandwidth: the copy kernel failed");
                break;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms > 0.f && (best == 0.f || ms < best)) best = ms;
        }
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (rc) return rc;
    if (best <= 0.f) return fail(LSF_ERR_HIP, "lsf_copy_bandwidth: no pass was timed");
    *gbps = 2.0 * (double)bytes / (best * 1e-3) / 1e9; // bytes read + bytes written per second of the fastest pass
    return LSF_OK;
}

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

#include "lsf_host_chain.hpp"

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
    const int ext[3] = {box->lx, box->ly, box->lz};
    const JacPlan jp = jacobi_plan(lo, hi, strict, ext);
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

int lsf_pack_boxes(const double* d_field, const lsf_box* box, int nreg, const int (*lo)[3], const int (*hi)[3], double* const* d_bufs, void* stream)
{
    return pack_multi_impl<double>(d_field, nullptr, box, nreg, lo, hi, d_bufs, 0, stream);
}
int lsf_unpack_boxes(double* d_field, const lsf_box* box, int nreg, const int (*lo)[3], const int (*hi)[3], double* const* d_bufs, void* stream)
{
    return pack_multi_impl<double>(nullptr, d_field, box, nreg, lo, hi, d_bufs, 1, stream);
}
int lsf_pack_boxes_f32(const float* d_field, const lsf_box* box, int nreg, const int (*lo)[3], const int (*hi)[3], float* const* d_bufs, void* stream)
{
    return pack_multi_impl<float>(d_field, nullptr, box, nreg, lo, hi, d_bufs, 0, stream);
}
int lsf_unpack_boxes_f32(float* d_field, const lsf_box* box, int nreg, const int (*lo)[3], const int (*hi)[3], float* const* d_bufs, void* stream)
{
    return pack_multi_impl<float>(nullptr, d_field, box, nreg, lo, hi, d_bufs, 1, stream);
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

#include "lsf_host_stl.hpp"

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

#include "lsf_rccl.hpp"


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
    HIPCHK(hipDeviceSynchronize()); // (a device-to-device hipMemcpy returns early; the run's streams are non-blocking: see lsfm::run)
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
