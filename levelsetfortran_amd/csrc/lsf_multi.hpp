// lsf_multi.hpp -- the block-decomposed Jacobi sweep behind the C ABI: ONE process drives every device of the node.
//
// The reference is serial and its call site is `CALL reinit(...)` in a single-threaded Fortran program (set3d.f90:308);
// a drop-in that wants the eight GPUs of a node therefore has to fan out below the seam.  lsf_reinit_multi takes the
// same arguments as lsf_reinit plus a device list:
//
//   * 3-D block decomposition of the (0:nx, 0:ny, 0:nz) field, one block per entry of the device list (default_dims: 1x1x2
//     on two, 1x2x2 on four, 2x2x2 on eight entries -- BASELINE's 2x2x1 / 2x2x2 with the unit-stride axis x cut last -- or as
//     given), 3 ghost layers towards every neighbour;
//   * one host thread per block: it owns the block's device, a compute stream and a communication stream and only ever
//     ENQUEUES inside a window of `check_every` sweeps (no barrier of all threads, no wait for the device);
//   * per sweep, communication stream: pack the six 3-cell face slabs (star stencil: faces only) -> transport -> unpack
//     into the ghost layers; compute stream: the core (cells that need no ghost) at the same time, then the rims, the
//     extrapolation BC on the owned wall points, one fixed-order reduction of the block's sum of squares;
//   * transport (lsf_multi_configure): peer copies straight into the neighbour's receive buffer (hipMemcpyPeerAsync: xGMI
//     between the devices of a node) with an event per message and a host hand-shake of the two neighbours' threads; or RCCL
//     -- ncclGroupStart / ncclSend + ncclRecv per neighbour / ncclGroupEnd on the block's communicator (ncclCommInitAll over
//     the device list, librccl.so loaded on first use);
//   * RMS / stop test once per window, one window late: the host threads add the block sums in rank order (every thread
//     computes the same numbers, so all take the same decision without a collective); a window that turns out to hold the
//     stop sweep is repeated from its kept start up to that sweep (see `worker`).
//
// The peer transport is the default because the SAME code runs with several blocks on one device (the device list may
// repeat a device), which is how the decomposition, the halo schedule, the event protocol and the reduction are tested
// bit for bit against the single-domain sweep on a one-GPU box (tests/test_gpu_multi.py); RCCL wants a device per block,
// so its schedule is tested through a stand-in transport (LSF_TRANSPORT_MOCK) and the library itself with one block.
// levelsetfortran_amd/distributed.py remains the one-process-per-GPU variant of the same sweep over torch.distributed
// (RCCL), built on the same four box calls.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lsf.h"

namespace lsfm {

constexpr int HALO = 3; // WENO5 reaches +-3 (subs.f90:509-530)

struct Reg {
    int lo[3], hi[3];
    long vol() const
    {
        long v = 1;
        for (int a = 0; a < 3; ++a) v *= std::max(hi[a] - lo[a], 0);
        return v;
    }
};

// blocks below this many owned points along their longest axis run a sweep as ONE launch after the exchange (LSF_MULTI_SMALL in
// the environment: threshold, 0 = never)
inline bool small_block(const int own[3][2])
{
    const char* e = getenv("LSF_MULTI_SMALL");
    const int thr = e ? atoi(e) : 192;
    int longest = 0;
    for (int a = 0; a < 3; ++a) longest = std::max(longest, own[a][1] - own[a][0]);
    return longest < thr;
}

struct Geom {
    int dims[3], coords[3], n[3];
    int own[3][2]; // owned global point range [s, e) per axis
    int g0[3];     // global index of local point 0
    int ext[3];    // local extents, ghosts included
    int nb[6];     // neighbour rank per face 2 * axis + side, -1 = wall
    Reg core;
    std::vector<Reg> rims;
    Reg send[6], recv[6];
    lsf_box box() const { return lsf_box{ext[0], ext[1], ext[2], g0[0], g0[1], g0[2], n[0], n[1], n[2]}; }
    size_t npoints() const { return (size_t)ext[0] * ext[1] * ext[2]; }
};

inline void split_points(int npoints, int parts, int p, int* s, int* e)
{
    const int base = npoints / parts, extra = npoints % parts;
    *s = p * base + std::min(p, extra);
    *e = *s + base + (p < extra ? 1 : 0);
}

inline int rank_of(const int c[3], const int dims[3]) { return c[0] + dims[0] * (c[1] + dims[1] * c[2]); }

// same decomposition as levelsetfortran_amd/distributed.py (make_block, sweep_regions, halo_plan)
inline bool make_geom(int rank, const int dims[3], const int n[3], Geom* g, std::string* err)
{
    for (int a = 0; a < 3; ++a) g->dims[a] = dims[a], g->n[a] = n[a];
    g->coords[0] = rank % dims[0], g->coords[1] = (rank / dims[0]) % dims[1], g->coords[2] = rank / (dims[0] * dims[1]);
    for (int a = 0; a < 3; ++a) {
        int s, e;
        split_points(n[a] + 1, dims[a], g->coords[a], &s, &e);
        if (dims[a] > 1 && e - s < 2 * HALO) {
            *err = "fewer than 2 * 3 owned points per block along an axis";
            return false;
        }
        g->own[a][0] = s, g->own[a][1] = e;
        const int lo = g->coords[a] > 0 ? s - HALO : s, hi = g->coords[a] < dims[a] - 1 ? e + HALO : e;
        g->g0[a] = lo, g->ext[a] = hi - lo;
    }
    // owned cells that are interior cells of the global grid, local indices
    int cells[3][2];
    for (int a = 0; a < 3; ++a) {
        cells[a][0] = std::max(g->own[a][0], 1) - g->g0[a];
        cells[a][1] = std::min(g->own[a][1], n[a]) - g->g0[a];
    }
    for (int a = 0; a < 3; ++a) {
        int lo = cells[a][0], hi = cells[a][1];
        if (g->coords[a] > 0) lo += HALO;
        if (g->coords[a] < dims[a] - 1) hi -= HALO;
        g->core.lo[a] = lo, g->core.hi[a] = std::max(hi, lo);
    }
    g->rims.clear();
    int cur[3][2];
    std::memcpy(cur, cells, sizeof cur);
    // Small blocks (north_star's 256^3 on eight GPUs: 131^3 per rank): a sweep is 0.03 ms of arithmetic under seven launches
    // whose fixed costs are the time (profiles/r04_jacobi_rank_model.txt) -- one sweep launch over all owned cells after the
    // ghosts have arrived instead of core || exchange, then up to three rims.  Same cells, same kernels per cell, same field.
    if (small_block(g->own)) {
        for (int a = 0; a < 3; ++a) g->core.hi[a] = g->core.lo[a]; // empty: nothing runs beside the exchange
        Reg r;
        for (int b = 0; b < 3; ++b) r.lo[b] = cells[b][0], r.hi[b] = cells[b][1];
        g->rims.push_back(r);
        for (int a = 0; a < 3; ++a) cur[a][0] = cur[a][1] = 0, g->core.lo[a] = g->core.hi[a] = cells[a][0]; // (nothing left to peel)
    }
    for (int a = 0; a < 3 && !small_block(g->own); ++a) { // peel one axis at a time: the rims are disjoint
        if (g->core.lo[a] > cur[a][0]) {
            Reg r;
            for (int b = 0; b < 3; ++b) r.lo[b] = cur[b][0], r.hi[b] = cur[b][1];
            r.hi[a] = g->core.lo[a];
            g->rims.push_back(r);
            cur[a][0] = g->core.lo[a];
        }
        if (g->core.hi[a] < cur[a][1]) {
            Reg r;
            for (int b = 0; b < 3; ++b) r.lo[b] = cur[b][0], r.hi[b] = cur[b][1];
            r.lo[a] = g->core.hi[a];
            g->rims.push_back(r);
            cur[a][1] = g->core.hi[a];
        }
    }
    for (int a = 0; a < 3; ++a)
        for (int side = 0; side < 2; ++side) {
            const int f = 2 * a + side;
            int c[3] = {g->coords[0], g->coords[1], g->coords[2]};
            c[a] += side ? 1 : -1;
            g->nb[f] = (c[a] < 0 || c[a] >= dims[a]) ? -1 : rank_of(c, dims);
            Reg s, r;
            for (int b = 0; b < 3; ++b) s.lo[b] = r.lo[b] = g->own[b][0] - g->g0[b], s.hi[b] = r.hi[b] = g->own[b][1] - g->g0[b];
            if (side == 0) {
                s.hi[a] = s.lo[a] + HALO;
                r.hi[a] = r.lo[a], r.lo[a] -= HALO;
            } else {
                s.lo[a] = s.hi[a] - HALO;
                r.lo[a] = r.hi[a], r.hi[a] += HALO;
            }
            g->send[f] = s, g->recv[f] = r;
        }
    return true;
}

inline void default_dims(int world, int dims[3])
{
    // BASELINE.json: 4 GPUs -> a 2x2x1 decomposition, 8 GPUs -> 2x2x2.  The unit-stride axis x is cut last (2 devices: z;
    // 4: y and z): slabs of whole rows, rims the sweep kernel covers with full wavefronts.  Otherwise the prime factors
    // are dealt to z, y, x in turn (the same rule as distributed.py default_dims).
    dims[0] = dims[1] = dims[2] = 1;
    if (world == 2) { dims[2] = 2; return; }
    if (world == 4) { dims[1] = dims[2] = 2; return; }
    if (world == 8) { dims[0] = dims[1] = dims[2] = 2; return; }
    int nn = world, a = 0;
    for (int p = 2; p <= world; ++p)
        while (nn % p == 0) dims[2 - a % 3] *= p, nn /= p, ++a;
}

// reusable barrier of the worker threads (they only enqueue between two barriers, so a spin is fine)
class SpinBarrier {
    std::atomic<int> count_{0}, gen_{0};
    int n_;

  public:
    explicit SpinBarrier(int n) : n_(n) {}
    void wait()
    {
        const int g = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            gen_.store(g + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen_.load(std::memory_order_acquire) == g)
                if (++spins > 2000) std::this_thread::yield();
        }
    }
};

constexpr int MAX_CHECK = 64; // most sweeps between two looks at the RMS

// the enqueue lock of a device, timed: what a thread spends HOLDING it is the cost of its runtime / library calls
struct TimedLock {
    std::unique_lock<std::mutex> lk;
    double* acc;
    std::chrono::steady_clock::time_point t0;
    TimedLock(std::mutex& m, double* a) : lk(m), acc(a), t0(std::chrono::steady_clock::now()) {}
    void unlock()
    {
        if (lk.owns_lock()) *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), lk.unlock();
    }
    void lock() { lk.lock(), t0 = std::chrono::steady_clock::now(); }
    ~TimedLock() { unlock(); }
};

template <typename T>
struct RankState {
    int dev = 0;
    Geom g;
    hipStream_t compute = nullptr, comm = nullptr;
    T* buf[2] = {nullptr, nullptr};
    T* phiS = nullptr;
    T* snap[2] = {nullptr, nullptr}; // the field at the start of the last two judging windows (runs with a stop tolerance only)
    T* sendb[6] = {};
    T* recvb[2][6] = {};
    double* d_sums = nullptr;       // [MAX_CHECK] block sum of squares per sweep of the current window
    double* h_sums = nullptr;       // pinned, [2][MAX_CHECK]
    hipEvent_t sent[2][6] = {};     // slab of face f has landed in the neighbour's receive buffer (ring of two enqueues)
    hipEvent_t ready[2] = {};       // group transports: the slabs of an enqueue are packed
    hipEvent_t taken[2][6] = {};    // mock transport: the neighbour has copied the slab of face f
    hipEvent_t halo = nullptr;      // ghosts of the sweep's input are complete
    hipEvent_t done[2] = {};        // sweep finished
    hipEvent_t chk[2] = {};         // the sums of a window are on the host
    double host_enqueue_s = 0.0;    // time this block's thread spent enqueuing during the last run, the waits for its neighbours'
                                    // threads and (blocks sharing a device) for the device's enqueue lock included
    double host_calls_s = 0.0;      // ... of which inside the runtime / library calls themselves
    // The compute stream of a sweep as two HIP graphs per buffer parity (core | rims + BC + block sum; the wait for the ghosts
    // sits between them): captured from the third sweep of a run on, replayed while the run's parameters stay the same
    struct SweepGraph {
        hipGraphExec_t core = nullptr, rest = nullptr;
        const void *in = nullptr, *out = nullptr;
        double dx = 0, h = 0;
        int mode = -1;
        bool failed = false;
    } graph[2];
    double* d_sum1 = nullptr;       // target of the captured reductions (copied into d_sums[slot] behind the graph)
    std::string err;
    int rc = LSF_OK;
};

// ---- RCCL, loaded on demand (the library has no link-time dependency on it) -------------------------------------------
// Only what the halo exchange needs: one communicator per block (ncclCommInitAll over the device list), and per sweep one
// group of ncclSend / ncclRecv of bytes on the block's communication stream.
struct Rccl {
    void* lib = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    std::vector<void*> comms;
    int version = 0;
    static constexpr int kInt8 = 0; // ncclInt8 / ncclChar: the slabs travel as bytes
    bool load(std::string* err);
    ~Rccl();
};

} // namespace lsfm

struct lsf_multi {
    int nx, ny, nz, ndev, f32;
    int dims[3];
    std::vector<lsfm::RankState<double>> r64;
    std::vector<lsfm::RankState<float>> r32;
    std::vector<std::mutex> devlock; // enqueue sections of blocks that share a device (the one-GPU rehearsal)
    // host-side hand-shake of neighbouring blocks: number of exchanges a block's thread has enqueued so far (`posted`: its
    // sends / packed slabs are in its queue) and, mock transport, how many of them each neighbour has picked up
    std::vector<std::atomic<long>> posted, taken_seq;
    int result_parity = 0;           // buf[result_parity] holds the field after lsf_multi_run
    std::vector<int> devs;
    int check_every = 8;             // sweeps per judging window (lsf_multi_configure)
    // compute stream of a sweep replayed from two HIP graphs (LSF_MULTI_GRAPHS=1).  Off by default: with eight blocks of 128^3
    // on one MI355X the host threads spend 9 % less time enqueuing (633 against 693 us per sweep) but the device takes 23 % longer
    // (884 against 717 us per sweep): a graph launch costs the device more than the five plain launches it replaces
    int graphs = 0;
    int transport = LSF_TRANSPORT_PEER;
    lsfm::Rccl rccl;
    double last_wall_s = 0.0;
    int last_sweeps_enqueued = 0;
    lsf_multi(int nd) : devlock(64), posted(64), taken_seq(64 * 6) { ndev = nd; }
};

namespace lsfm {

#define LSFM_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            R.err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            R.rc = LSF_ERR_HIP;                                                             \
            return;                                                                         \
        }                                                                                   \
    } while (0)
#define LSFM_LSF(expr)                                                   \
    do {                                                                 \
        const int rc_ = (expr);                                          \
        if (rc_ != LSF_OK) {                                             \
            R.err = std::string(#expr) + ": " + lsf_last_error();        \
            R.rc = rc_;                                                  \
            return;                                                      \
        }                                                                \
    } while (0)
#define LSFM_NCCL(expr)                                                                                  \
    do {                                                                                                 \
        const int rc_ = (expr);                                                                          \
        if (rc_ != 0) {                                                                                  \
            R.err = std::string(#expr) + ": " + (M->rccl.GetErrorString ? M->rccl.GetErrorString(rc_) : "RCCL error"); \
            R.rc = LSF_ERR_HIP;                                                                          \
            return;                                                                                      \
        }                                                                                                \
    } while (0)

template <typename T> struct BoxCalls;
template <> struct BoxCalls<double> {
    static int sweep(const double* a, double* b, const double* s, const lsf_box* bx, const int* lo, const int* hi, double dx, double h,
                     int mode, double* sum, void* st) { return lsf_jacobi_sweep_box(a, b, s, bx, lo, hi, dx, h, mode, sum, st); }
    static int bc(const double* a, double* b, const lsf_box* bx, const int* lo, const int* hi, double dx, double* sum, void* st)
    { return lsf_bc_box(a, b, bx, lo, hi, dx, sum, st); }
    static int pack(const double* f, const lsf_box* bx, const int* lo, const int* hi, double* buf, void* st) { return lsf_pack_box(f, bx, lo, hi, buf, st); }
    static int unpack(double* f, const lsf_box* bx, const int* lo, const int* hi, const double* buf, void* st) { return lsf_unpack_box(f, bx, lo, hi, buf, st); }
    static int pack_all(const double* f, const lsf_box* bx, int n, const int (*lo)[3], const int (*hi)[3], double* const* bufs, void* st) { return lsf_pack_boxes(f, bx, n, lo, hi, bufs, st); }
    static int unpack_all(double* f, const lsf_box* bx, int n, const int (*lo)[3], const int (*hi)[3], double* const* bufs, void* st) { return lsf_unpack_boxes(f, bx, n, lo, hi, bufs, st); }
};
template <> struct BoxCalls<float> {
    static int sweep(const float* a, float* b, const float* s, const lsf_box* bx, const int* lo, const int* hi, double dx, double h,
                     int mode, double* sum, void* st) { return lsf_jacobi_sweep_box_f32(a, b, s, bx, lo, hi, dx, h, mode, sum, st); }
    static int bc(const float* a, float* b, const lsf_box* bx, const int* lo, const int* hi, double dx, double* sum, void* st)
    { return lsf_bc_box_f32(a, b, bx, lo, hi, dx, sum, st); }
    static int pack(const float* f, const lsf_box* bx, const int* lo, const int* hi, float* buf, void* st) { return lsf_pack_box_f32(f, bx, lo, hi, buf, st); }
    static int unpack(float* f, const lsf_box* bx, const int* lo, const int* hi, const float* buf, void* st) { return lsf_unpack_box_f32(f, bx, lo, hi, buf, st); }
    static int pack_all(const float* f, const lsf_box* bx, int n, const int (*lo)[3], const int (*hi)[3], float* const* bufs, void* st) { return lsf_pack_boxes_f32(f, bx, n, lo, hi, bufs, st); }
    static int unpack_all(float* f, const lsf_box* bx, int n, const int (*lo)[3], const int (*hi)[3], float* const* bufs, void* st) { return lsf_unpack_boxes_f32(f, bx, n, lo, hi, bufs, st); }
};

template <typename T>
void alloc_rank(RankState<T>& R)
{
    LSFM_LSF(lsf_set_device(R.dev));
    LSFM_HIP(hipSetDevice(R.dev));
    LSFM_HIP(hipStreamCreateWithFlags(&R.compute, hipStreamNonBlocking));
    LSFM_HIP(hipStreamCreateWithFlags(&R.comm, hipStreamNonBlocking));
    const size_t np = R.g.npoints();
    for (int q = 0; q < 2; ++q) LSFM_HIP(hipMalloc((void**)&R.buf[q], np * sizeof(T)));
    LSFM_HIP(hipMalloc((void**)&R.phiS, np * sizeof(T)));
    LSFM_HIP(hipMalloc((void**)&R.d_sums, MAX_CHECK * sizeof(double)));
    LSFM_HIP(hipMalloc((void**)&R.d_sum1, sizeof(double)));
    LSFM_HIP(hipHostMalloc((void**)&R.h_sums, 2 * MAX_CHECK * sizeof(double), hipHostMallocDefault));
    for (int f = 0; f < 6; ++f) {
        if (R.g.nb[f] < 0) continue;
        const size_t b = (size_t)R.g.send[f].vol() * sizeof(T);
        LSFM_HIP(hipMalloc((void**)&R.sendb[f], b));
        for (int q = 0; q < 2; ++q) {
            LSFM_HIP(hipMalloc((void**)&R.recvb[q][f], b));
            LSFM_HIP(hipEventCreateWithFlags(&R.sent[q][f], hipEventDisableTiming));
            LSFM_HIP(hipEventCreateWithFlags(&R.taken[q][f], hipEventDisableTiming));
        }
    }
    LSFM_HIP(hipEventCreateWithFlags(&R.halo, hipEventDisableTiming));
    for (int q = 0; q < 2; ++q) {
        LSFM_HIP(hipEventCreateWithFlags(&R.done[q], hipEventDisableTiming));
        LSFM_HIP(hipEventCreateWithFlags(&R.chk[q], hipEventDisableTiming));
        LSFM_HIP(hipEventCreateWithFlags(&R.ready[q], hipEventDisableTiming));
    }
}

template <typename T>
void free_rank(RankState<T>& R)
{
    (void)hipSetDevice(R.dev);
    if (R.compute) (void)hipStreamSynchronize(R.compute);
    if (R.comm) (void)hipStreamSynchronize(R.comm);
    for (int q = 0; q < 2; ++q) {
        if (R.buf[q]) (void)hipFree(R.buf[q]);
        if (R.snap[q]) (void)hipFree(R.snap[q]);
        if (R.done[q]) (void)hipEventDestroy(R.done[q]);
        if (R.chk[q]) (void)hipEventDestroy(R.chk[q]);
        if (R.ready[q]) (void)hipEventDestroy(R.ready[q]);
        for (int f = 0; f < 6; ++f) {
            if (R.recvb[q][f]) (void)hipFree(R.recvb[q][f]);
            if (R.sent[q][f]) (void)hipEventDestroy(R.sent[q][f]);
            if (R.taken[q][f]) (void)hipEventDestroy(R.taken[q][f]);
        }
    }
    for (int f = 0; f < 6; ++f)
        if (R.sendb[f]) (void)hipFree(R.sendb[f]);
    if (R.phiS) (void)hipFree(R.phiS);
    if (R.d_sums) (void)hipFree(R.d_sums);
    if (R.d_sum1) (void)hipFree(R.d_sum1);
    for (auto& g : R.graph) {
        if (g.core) (void)hipGraphExecDestroy(g.core);
        if (g.rest) (void)hipGraphExecDestroy(g.rest);
    }
    if (R.h_sums) (void)hipHostFree(R.h_sums);
    if (R.halo) (void)hipEventDestroy(R.halo);
    if (R.compute) (void)hipStreamDestroy(R.compute);
    if (R.comm) (void)hipStreamDestroy(R.comm);
    R = RankState<T>{};
}

struct RunShared {
    double dx, h, tol, den;
    int iter, mode;
    std::vector<double> vals;   // [block][sweep of the window] block sums of the window being judged
    std::vector<double> trace;  // global RMS per sweep
    std::atomic<int> failed{0};
    long seq0 = 0;              // exchanges enqueued by earlier runs on this object (the hand-shake counters never go back)
};

// ---------------------------------------------------------------------------------------------------------------------
// One worker thread = one block.  Between two looks at the RMS a thread only enqueues:
//
//   exchange e (e counts the exchanges this object has enqueued, in every run):
//     comm stream     wait for the previous sweep -> pack the (up to) six face slabs -> transport -> unpack into the ghosts
//                     -> `halo`
//     compute stream  core cells (need no ghost) -> wait `halo` -> rims -> BC on the owned wall points -> block sum of
//                     squares into d_sums[sweep mod window] -> `done`
//
//   transport PEER   one peer copy per neighbour straight into ITS receive buffer e mod 2, `sent` recorded behind it; the
//                    receiving thread waits (host, that one neighbour only) until the sender has ENQUEUED exchange e --
//                    posted[sender] > e -- and lets its stream wait for that event.  A neighbour can be at most one
//                    exchange ahead (its exchange e + 1 needs my `posted` > e + 1, which follows my enqueue of e), so two
//                    events and two receive buffers per face suffice, and there is no barrier of all threads per sweep.
//   transport RCCL   ncclGroupStart; ncclSend + ncclRecv per neighbour on the block's communicator; ncclGroupEnd -- the
//                    rendezvous is RCCL's, no host hand-shake at all.
//   transport MOCK   the RCCL schedule (pack everything, ONE group call, unpack everything; no `sent` events) with a
//                    stand-in for the library that runs with several blocks on one device: the receiver pulls the slab out
//                    of the sender's send buffer once `posted` says it is packed, the sender's stream waits until the
//                    receiver has taken it (what stream-ordered completion of ncclSend means).  Test aid.
//
// The RMS is judged once per window of `check_every` sweeps, one window late: while the GPU works on window w + 1 the host
// threads read the block sums of window w (one event wait, one copy of check_every doubles), add them in rank order -- every
// thread computes the same numbers, so all take the same decision without a collective -- and look for the first sweep
// below the tolerance.  If there is one, the sweeps enqueued beyond it have overwritten both field buffers: the window's
// start was kept (`snap`, one device copy per window, only for runs with a positive tolerance) and the sweeps up to the stop
// sweep are run again from it -- the field, the sweep count and the trace are those of a driver that looks every sweep.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
void worker(lsf_multi* M, std::vector<RankState<T>>* ranks, int r, SpinBarrier* bar, RunShared* S)
{
    using C = BoxCalls<T>;
    using clk = std::chrono::steady_clock;
    RankState<T>& R = (*ranks)[r];
    const int nr = (int)ranks->size();
    auto fail_all = [&]() { S->failed.store(1); };
    if (lsf_set_device(R.dev) != LSF_OK || hipSetDevice(R.dev) != hipSuccess) {
        R.rc = LSF_ERR_HIP, R.err = "cannot select the block's device";
        fail_all();
    }
    const lsf_box bx = R.g.box();
    int own_lo[3], own_hi[3];
    for (int a = 0; a < 3; ++a) own_lo[a] = R.g.own[a][0] - R.g.g0[a], own_hi[a] = R.g.own[a][1] - R.g.g0[a];
    std::mutex& dl = M->devlock[(size_t)R.dev % M->devlock.size()];
    const int K = std::min(std::max(M->check_every, 1), MAX_CHECK);
    const int transport = M->transport;
    const bool use_graphs = M->graphs != 0;
    R.host_enqueue_s = R.host_calls_s = 0.0;
    long seq = S->seq0; // exchanges enqueued so far

    // wait (host) until counter > want; gives up when another block has failed
    auto await = [&](std::atomic<long>& counter, long want) -> bool {
        int spins = 0;
        while (counter.load(std::memory_order_acquire) <= want) {
            if (S->failed.load(std::memory_order_relaxed)) return false;
            if (++spins > 2000) std::this_thread::yield();
        }
        return true;
    };

    // the face slabs of the block travel in ONE pack and ONE unpack launch per sweep (lsf_pack_boxes): the faces that have a neighbour
    int nf = 0, face_of[6], s_lo[6][3], s_hi[6][3], r_lo[6][3], r_hi[6][3];
    T* s_buf[6];
    for (int f = 0; f < 6; ++f) {
        if (R.g.nb[f] < 0) continue;
        for (int a = 0; a < 3; ++a) s_lo[nf][a] = R.g.send[f].lo[a], s_hi[nf][a] = R.g.send[f].hi[a], r_lo[nf][a] = R.g.recv[f].lo[a], r_hi[nf][a] = R.g.recv[f].hi[a];
        s_buf[nf] = R.sendb[f];
        face_of[nf++] = f;
    }
    auto unpack_faces = [&](T* a_in, int eq) -> int {
        T* r_buf[6];
        for (int k = 0; k < nf; ++k) r_buf[k] = R.recvb[eq][face_of[k]];
        return C::unpack_all(a_in, &bx, nf, r_lo, r_hi, r_buf, R.comm);
    };

    // ---- comm stream, first half: pack (and, PEER, push) the slabs of sweep s, exchange number e
    auto enqueue_sends = [&](int s, long e) {
        if (R.rc) return;
        TimedLock lk(dl, &R.host_calls_s);
        const int q = s & 1, eq = (int)(e & 1);
        const T* a_in = R.buf[q];
        // the input of this sweep is the output of the previous one (compute stream)
        if (e > S->seq0) LSFM_HIP(hipStreamWaitEvent(R.comm, R.done[(s - 1) & 1], 0));
        LSFM_LSF(C::pack_all(a_in, &bx, nf, s_lo, s_hi, s_buf, R.comm));
        for (int f = 0; f < 6; ++f) {
            const int p = R.g.nb[f];
            if (p < 0) continue;
            if (transport == LSF_TRANSPORT_PEER) {
                RankState<T>& P = (*ranks)[p];
                LSFM_HIP(hipMemcpyPeerAsync(P.recvb[eq][f ^ 1], P.dev, R.sendb[f], R.dev, (size_t)R.g.send[f].vol() * sizeof(T), R.comm));
                LSFM_HIP(hipEventRecord(R.sent[eq][f], R.comm));
            }
        }
        if (transport == LSF_TRANSPORT_MOCK) LSFM_HIP(hipEventRecord(R.ready[eq], R.comm));
    };
    // ---- comm stream, second half (receive + unpack) and the compute stream of sweep s
    auto enqueue_sweep = [&](int s, long e) {
        if (R.rc) return;
        const int q = s & 1, eq = (int)(e & 1);
        T* a_in = R.buf[q];
        T* a_out = R.buf[q ^ 1];
        if (transport == LSF_TRANSPORT_PEER) {
            for (int f = 0; f < 6; ++f) { // the neighbours have enqueued their sends of this exchange
                const int p = R.g.nb[f];
                if (p >= 0 && !await(M->posted[p], e)) { R.rc = LSF_ERR_HIP, R.err = "another block failed"; return; }
            }
        }
        TimedLock lk(dl, &R.host_calls_s);
        if (transport == LSF_TRANSPORT_PEER) {
            for (int f = 0; f < 6; ++f) {
                const int p = R.g.nb[f];
                if (p < 0) continue;
                LSFM_HIP(hipStreamWaitEvent(R.comm, (*ranks)[p].sent[eq][f ^ 1], 0));
            }
            LSFM_LSF(unpack_faces(a_in, eq));
        } else if (transport == LSF_TRANSPORT_RCCL) {
            LSFM_NCCL(M->rccl.GroupStart());
            for (int f = 0; f < 6; ++f) {
                const int p = R.g.nb[f];
                if (p < 0) continue;
                const size_t bytes = (size_t)R.g.send[f].vol() * sizeof(T);
                LSFM_NCCL(M->rccl.Send(R.sendb[f], bytes, Rccl::kInt8, p, M->rccl.comms[r], R.comm));
                LSFM_NCCL(M->rccl.Recv(R.recvb[eq][f], bytes, Rccl::kInt8, p, M->rccl.comms[r], R.comm));
            }
            LSFM_NCCL(M->rccl.GroupEnd());
            LSFM_LSF(unpack_faces(a_in, eq));
        } else { // MOCK: the group call of the RCCL schedule, carried out by pull copies
            lk.unlock();
            for (int f = 0; f < 6; ++f) { // "recv": the neighbour's slab is packed -> copy it out of its send buffer
                const int p = R.g.nb[f];
                if (p < 0) continue;
                if (!await(M->posted[p], e)) { R.rc = LSF_ERR_HIP, R.err = "another block failed"; return; }
                RankState<T>& P = (*ranks)[p];
                TimedLock lk2(dl, &R.host_calls_s);
                LSFM_HIP(hipStreamWaitEvent(R.comm, P.ready[eq], 0));
                LSFM_HIP(hipMemcpyPeerAsync(R.recvb[eq][f], R.dev, P.sendb[f ^ 1], P.dev, (size_t)R.g.send[f].vol() * sizeof(T), R.comm));
                LSFM_HIP(hipEventRecord(R.taken[eq][f], R.comm));
                M->taken_seq[(size_t)r * 6 + f].store(e + 1, std::memory_order_release);
            }
            for (int f = 0; f < 6; ++f) { // "send" completes on my stream once the neighbour has taken the slab
                const int p = R.g.nb[f];
                if (p < 0) continue;
                if (!await(M->taken_seq[(size_t)p * 6 + (f ^ 1)], e)) { R.rc = LSF_ERR_HIP, R.err = "another block failed"; return; }
                TimedLock lk2(dl, &R.host_calls_s);
                LSFM_HIP(hipStreamWaitEvent(R.comm, (*ranks)[p].taken[eq][f ^ 1], 0));
            }
            lk.lock();
            LSFM_LSF(unpack_faces(a_in, eq));
        }
        LSFM_HIP(hipEventRecord(R.halo, R.comm));
        // ---- compute stream
        double* d_slot = R.d_sums + (s % K);
        auto core_part = [&](double* d_sum) -> int {
            if (hipMemsetAsync(d_sum, 0, sizeof(double), R.compute) != hipSuccess) return LSF_ERR_HIP;
            int rc = lsf_sumsq_begin(R.compute);
            if (rc == LSF_OK && R.g.core.vol() > 0)
                rc = C::sweep(a_in, a_out, R.phiS, &bx, R.g.core.lo, R.g.core.hi, S->dx, S->h, S->mode, d_sum, R.compute); // overlaps the exchange
            return rc;
        };
        auto rest_part = [&](double* d_sum) -> int {
            int rc = LSF_OK;
            for (size_t k = 0; k < R.g.rims.size() && rc == LSF_OK; ++k)
                if (R.g.rims[k].vol() > 0)
                    rc = C::sweep(a_in, a_out, R.phiS, &bx, R.g.rims[k].lo, R.g.rims[k].hi, S->dx, S->h, S->mode, d_sum, R.compute);
            if (rc == LSF_OK) rc = C::bc(a_in, a_out, &bx, own_lo, own_hi, S->dx, d_sum, R.compute);
            const int rc2 = lsf_sumsq_end(R.compute); // always close the bracket
            return rc ? rc : rc2;
        };
        auto& G = R.graph[q];
        const bool same = G.core && G.rest && G.in == a_in && G.out == a_out && G.dx == S->dx && G.h == S->h && G.mode == S->mode;
        if (use_graphs && !same && !G.failed && e - S->seq0 >= 2) {
            // capture this sweep (the two sweeps before it have sized every workspace: nothing allocates any more)
            if (G.core) (void)hipGraphExecDestroy(G.core);
            if (G.rest) (void)hipGraphExecDestroy(G.rest);
            G.core = G.rest = nullptr;
            hipGraph_t g1 = nullptr, g2 = nullptr;
            bool ok = hipStreamBeginCapture(R.compute, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int rc = core_part(R.d_sum1);
                ok = hipStreamEndCapture(R.compute, &g1) == hipSuccess && rc == LSF_OK && g1;
            }
            if (ok) ok = hipStreamBeginCapture(R.compute, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                const int rc = rest_part(R.d_sum1);
                ok = hipStreamEndCapture(R.compute, &g2) == hipSuccess && rc == LSF_OK && g2;
            } else {
                (void)lsf_sumsq_end(R.compute);
            }
            if (ok) ok = hipGraphInstantiate(&G.core, g1, nullptr, nullptr, 0) == hipSuccess && hipGraphInstantiate(&G.rest, g2, nullptr, nullptr, 0) == hipSuccess;
            if (g1) (void)hipGraphDestroy(g1);
            if (g2) (void)hipGraphDestroy(g2);
            (void)hipGetLastError();
            if (ok) G.in = a_in, G.out = a_out, G.dx = S->dx, G.h = S->h, G.mode = S->mode;
            else {
                if (G.core) (void)hipGraphExecDestroy(G.core);
                if (G.rest) (void)hipGraphExecDestroy(G.rest);
                G.core = G.rest = nullptr, G.failed = true; // this object keeps to plain launches
            }
        }
        if (use_graphs && G.core && G.rest && G.in == a_in && G.out == a_out && G.dx == S->dx && G.h == S->h && G.mode == S->mode) {
            LSFM_HIP(hipGraphLaunch(G.core, R.compute));
            LSFM_HIP(hipStreamWaitEvent(R.compute, R.halo, 0));
            LSFM_HIP(hipGraphLaunch(G.rest, R.compute));
            LSFM_HIP(hipMemcpyAsync(d_slot, R.d_sum1, sizeof(double), hipMemcpyDeviceToDevice, R.compute));
        } else {
            int rc = core_part(d_slot);
            const hipError_t he = hipStreamWaitEvent(R.compute, R.halo, 0);
            const int rc2 = rest_part(d_slot);
            if (he != hipSuccess) { R.rc = LSF_ERR_HIP, R.err = hipGetErrorString(he); return; }
            LSFM_LSF(rc);
            LSFM_LSF(rc2);
        }
        LSFM_HIP(hipEventRecord(R.done[q], R.compute));
    };
    auto one_sweep = [&](int s) {
        const auto t0 = clk::now();
        enqueue_sends(s, seq);
        if (R.rc) fail_all();
        M->posted[r].store(seq + 1, std::memory_order_release); // also when this block has failed: nobody waits for ever
        enqueue_sweep(s, seq);
        if (R.rc) fail_all();
        ++seq;
        R.host_enqueue_s += std::chrono::duration<double>(clk::now() - t0).count();
    };
    // the sums of window w (sweeps c0 .. c0 + n - 1) travel home behind its last sweep
    auto close_window = [&](int w, int n) {
        if (R.rc) return;
        std::lock_guard<std::mutex> lk(dl);
        LSFM_HIP(hipMemcpyAsync(R.h_sums + (size_t)(w & 1) * MAX_CHECK, R.d_sums, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, R.compute));
        LSFM_HIP(hipEventRecord(R.chk[w & 1], R.compute));
    };
    // judge window w: returns the index (within the run) of the first sweep that ends the run, or -1
    auto judge = [&](int w, int c0, int n) -> int {
        if (!R.rc && hipEventSynchronize(R.chk[w & 1]) != hipSuccess) R.rc = LSF_ERR_HIP, R.err = "event synchronisation failed";
        for (int j = 0; j < n; ++j) S->vals[(size_t)r * MAX_CHECK + j] = R.rc ? std::nan("") : R.h_sums[(size_t)(w & 1) * MAX_CHECK + j];
        if (R.rc) fail_all();
        bar->wait();
        int stop = -1;
        for (int j = 0; j < n; ++j) {
            double tot = 0.0;
            for (int k = 0; k < nr; ++k) tot += S->vals[(size_t)k * MAX_CHECK + j]; // rank order: fixed
            const double quo = tot / S->den;
            const double rms = quo >= 0.0 ? std::sqrt(quo) : std::nan(""); // the wrapped INTEGER*4 product may be negative
            if (r == 0 && stop < 0) S->trace.push_back(rms);
            if (stop < 0 && (rms < S->tol || rms != rms)) stop = c0 + j; // a failed block reports NaN: every thread takes the same decision
        }
        bar->wait(); // vals may be overwritten from here on
        return stop;
    };

    const int max_sweeps = S->iter + 1; // DO n = 0, iter (subs.f90:735)
    const bool keep_start = S->tol > 0.0; // a run that can stop early must be able to go back to the stop sweep
    int s = 0, w = 0, stop_at = -1;
    int win_c0[2] = {0, 0}, win_n[2] = {0, 0};
    while (s < max_sweeps && stop_at < 0) {
        const int c0 = s, n = std::min(K, max_sweeps - s);
        if (keep_start && !R.rc) {
            std::lock_guard<std::mutex> lk(dl);
            if (hipMemcpyAsync(R.snap[w & 1], R.buf[c0 & 1], R.g.npoints() * sizeof(T), hipMemcpyDeviceToDevice, R.compute) != hipSuccess)
                R.rc = LSF_ERR_HIP, R.err = "keeping the start of a window failed";
        }
        for (int j = 0; j < n; ++j, ++s) one_sweep(s);
        close_window(w, n);
        if (R.rc) fail_all();
        win_c0[w & 1] = c0, win_n[w & 1] = n;
        if (w >= 1) stop_at = judge(w - 1, win_c0[(w - 1) & 1], win_n[(w - 1) & 1]); // one window late: window w is already in the queues
        ++w;
    }
    if (stop_at < 0 && w >= 1) stop_at = judge(w - 1, win_c0[(w - 1) & 1], win_n[(w - 1) & 1]);
    if (!R.rc) {
        (void)hipStreamSynchronize(R.comm);
        (void)hipStreamSynchronize(R.compute);
    }
    // the run ends at sweep stop_at, but sweeps beyond it have been enqueued: go back to the start of its window and repeat
    // the sweeps up to it (every thread takes this branch or none does)
    if (stop_at >= 0 && stop_at + 1 < s && keep_start && !S->failed.load()) {
        const int wv = stop_at / K, c0 = wv * K; // windows start at multiples of K
        {
            std::lock_guard<std::mutex> lk(dl);
            if (!R.rc && hipMemcpyAsync(R.buf[c0 & 1], R.snap[wv & 1], R.g.npoints() * sizeof(T), hipMemcpyDeviceToDevice, R.compute) != hipSuccess)
                R.rc = LSF_ERR_HIP, R.err = "restoring the start of a window failed";
            if (!R.rc && hipEventRecord(R.done[(c0 - 1) & 1], R.compute) != hipSuccess) R.rc = LSF_ERR_HIP, R.err = "event record failed";
        }
        if (R.rc) fail_all();
        for (int t = c0; t <= stop_at; ++t) one_sweep(t);
        if (!R.rc) {
            (void)hipStreamSynchronize(R.comm);
            (void)hipStreamSynchronize(R.compute);
        }
    }
    if (r == 0) M->last_sweeps_enqueued = (int)(seq - S->seq0);
}

template <typename T>
int run(lsf_multi* M, std::vector<RankState<T>>& ranks, int iter, double dx, double h, double tol, int mode, int* sweeps_done,
        double* rms_trace, int trace_cap, std::string* err)
{
    const int nr = (int)ranks.size();
    RunShared S;
    S.dx = dx, S.h = h, S.tol = tol, S.iter = iter, S.mode = mode;
    // the reference divides by the INTEGER*4 product nx*ny*nz (subs.f90:914), which wraps; fp32 fields have no reference
    // to mirror and use the true product (include/lsf.h)
    S.den = sizeof(T) == 4 ? (double)M->nx * M->ny * M->nz
                           : (double)(int32_t)((uint32_t)M->nx * (uint32_t)M->ny * (uint32_t)M->nz);
    S.vals.assign((size_t)nr * MAX_CHECK, 0.0);
    S.seq0 = M->posted[0].load();
    for (auto& R : ranks) { // phiS = phi on entry (subs.f90:731)
        // ON THE BLOCK'S COMPUTE STREAM: a device-to-device hipMemcpy returns before the copy has run, on the null stream, and the
        // block's streams are non-blocking -- nothing ordered the first sweep behind it.  Found in round 5 by the at-size
        // rehearsal of BASELINE configuration 5 (tests/test_gpu_configs45.py): with blocks of >= 1 GB the first workgroups of the
        // first sweep read a sign field that had not arrived (NaN); blocks of a few hundred MB had always won the race.
        if (hipSetDevice(R.dev) != hipSuccess ||
            hipMemcpyAsync(R.phiS, R.buf[0], R.g.npoints() * sizeof(T), hipMemcpyDeviceToDevice, R.compute) != hipSuccess) {
            *err = "copying the sign field failed";
            return LSF_ERR_HIP;
        }
        if (tol > 0.0)
            for (int q = 0; q < 2; ++q)
                if (!R.snap[q] && hipMalloc((void**)&R.snap[q], R.g.npoints() * sizeof(T)) != hipSuccess) {
                    *err = "no memory for the window-start copies of a block";
                    return LSF_ERR_HIP;
                }
    }
    SpinBarrier bar(nr);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nr; ++r) th.emplace_back(worker<T>, M, &ranks, r, &bar, &S);
    for (auto& t : th) t.join();
    M->last_wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // a failed run leaves the hand-shake counters uneven: level them (no thread is running)
    long top = 0;
    for (int r = 0; r < nr; ++r) top = std::max(top, M->posted[r].load());
    for (int r = 0; r < nr; ++r) M->posted[r].store(top);
    for (size_t k = 0; k < M->taken_seq.size(); ++k) M->taken_seq[k].store(top);
    for (auto& R : ranks)
        if (R.rc) {
            *err = R.err;
            const int rc = R.rc;
            for (auto& Q : ranks) Q.rc = LSF_OK, Q.err.clear();
            return rc;
        }
    // stop sweep: the first sweep whose RMS is below the tolerance or NaN (the trace ends there)
    int nsw = (int)S.trace.size();
    for (int k = 0; k < (int)S.trace.size(); ++k)
        if (S.trace[k] < tol || S.trace[k] != S.trace[k]) {
            nsw = k + 1;
            break;
        }
    M->result_parity = nsw & 1;
    if (sweeps_done) *sweeps_done = nsw;
    if (rms_trace)
        for (int k = 0; k < nsw && k < trace_cap; ++k) rms_trace[k] = S.trace[k];
    if (nsw > 0 && S.trace[nsw - 1] != S.trace[nsw - 1]) {
        *err = "RMS became NaN (the reference STOPs here, subs.f90:926)";
        return LSF_ERR_NAN;
    }
    return LSF_OK;
}

// host <-> blocks: every block receives its local box (ghost layers included) / returns its owned points
template <typename T>
int scatter(std::vector<RankState<T>>& ranks, const T* host, int nx, int ny, std::string* err)
{
    const size_t sx = (size_t)nx + 1, sxy = sx * ((size_t)ny + 1);
    std::vector<T> tmp;
    for (auto& R : ranks) {
        const Geom& g = R.g;
        tmp.resize(g.npoints());
        for (int k = 0; k < g.ext[2]; ++k)
            for (int j = 0; j < g.ext[1]; ++j)
                std::memcpy(&tmp[(size_t)g.ext[0] * (j + (size_t)g.ext[1] * k)],
                            host + g.g0[0] + sx * (size_t)(g.g0[1] + j) + sxy * (size_t)(g.g0[2] + k), (size_t)g.ext[0] * sizeof(T));
        if (hipSetDevice(R.dev) != hipSuccess ||
            hipMemcpy(R.buf[0], tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
            *err = "host-to-device copy of a block failed";
            return LSF_ERR_HIP;
        }
    }
    return LSF_OK;
}
template <typename T>
int gather(std::vector<RankState<T>>& ranks, int parity, T* host, int nx, int ny, std::string* err)
{
    const size_t sx = (size_t)nx + 1, sxy = sx * ((size_t)ny + 1);
    std::vector<T> tmp;
    for (auto& R : ranks) {
        const Geom& g = R.g;
        tmp.resize(g.npoints());
        if (hipSetDevice(R.dev) != hipSuccess ||
            hipMemcpy(tmp.data(), R.buf[parity], tmp.size() * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) {
            *err = "device-to-host copy of a block failed";
            return LSF_ERR_HIP;
        }
        const int o0 = g.own[0][0] - g.g0[0], w0 = g.own[0][1] - g.own[0][0];
        for (int k = g.own[2][0]; k < g.own[2][1]; ++k)
            for (int j = g.own[1][0]; j < g.own[1][1]; ++j)
                std::memcpy(host + g.own[0][0] + sx * (size_t)j + sxy * (size_t)k,
                            &tmp[o0 + (size_t)g.ext[0] * ((j - g.g0[1]) + (size_t)g.ext[1] * (k - g.g0[2]))], (size_t)w0 * sizeof(T));
    }
    return LSF_OK;
}

} // namespace lsfm
