// lsf_multi.hpp -- the block-decomposed Jacobi sweep behind the C ABI: ONE process drives every device of the node.
//
// The reference is serial and its call site is `CALL reinit(...)` in a single-threaded Fortran program (set3d.f90:308);
// a drop-in that wants the eight GPUs of a node therefore has to fan out below the seam.  lsf_reinit_multi takes the
// same arguments as lsf_reinit plus a device list:
//
//   * 3-D block decomposition of the (0:nx, 0:ny, 0:nz) field, one block per entry of the device list (2x2x1 on four,
//     2x2x2 on eight entries -- BASELINE configurations 4 and 5 -- or as given), 3 ghost layers towards every neighbour;
//   * one host thread per block: it owns the block's device, a compute stream and a communication stream and only ever
//     ENQUEUES (no host synchronisation inside a sweep);
//   * per sweep, communication stream: pack the six 3-cell face slabs (star stencil: faces only) -> one peer copy per
//     neighbour straight into the neighbour's receive buffer (hipMemcpyPeerAsync: xGMI between the devices of a node) ->
//     unpack into the ghost layers once the neighbour's event says its slab has landed; compute stream: the core (cells
//     that need no ghost) at the same time, then the rims, the extrapolation BC on the owned wall points, one fixed-order
//     reduction of the block's sum of squares;
//   * RMS / stop test one sweep late: while sweep s runs, the host threads add the block sums of sweep s - 1 in rank order
//     (every thread computes the same number, so all take the same decision without a collective); a sweep enqueued past
//     the stop sweep only writes the buffer the result is not in.
//
// The transport is the runtime's peer copy rather than an RCCL communicator: inside one process it is the same xGMI
// path without a rendezvous, and -- the reason it is the default -- the SAME code runs with several blocks on one device
// (the device list may repeat a device), which is how the decomposition, the halo schedule, the event protocol and the
// reduction are tested bit for bit against the single-domain sweep on a one-GPU box (tests/test_gpu_multi.py).
// levelsetfortran_amd/distributed.py remains the one-process-per-GPU variant of the same sweep over torch.distributed
// (RCCL), built on the same four box calls.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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
    for (int a = 0; a < 3; ++a) { // peel one axis at a time: the rims are disjoint
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

template <typename T>
struct RankState {
    int dev = 0;
    Geom g;
    hipStream_t compute = nullptr, comm = nullptr;
    T* buf[2] = {nullptr, nullptr};
    T* phiS = nullptr;
    T* sendb[6] = {};
    T* recvb[2][6] = {};
    double* d_sum = nullptr;
    double* h_sum = nullptr;        // pinned, [2]
    hipEvent_t sent[6] = {};        // slab of face f has landed in the neighbour's receive buffer
    hipEvent_t halo = nullptr;      // ghosts of the sweep's input are complete
    hipEvent_t done[2] = {};        // sweep finished, block sum copied to h_sum[parity]
    std::string err;
    int rc = LSF_OK;
};

} // namespace lsfm

struct lsf_multi {
    int nx, ny, nz, ndev, f32;
    int dims[3];
    std::vector<lsfm::RankState<double>> r64;
    std::vector<lsfm::RankState<float>> r32;
    std::vector<std::mutex> devlock; // enqueue sections of blocks that share a device (the one-GPU rehearsal)
    int result_parity = 0;           // buf[result_parity] holds the field after lsf_multi_run
    std::vector<int> devs;
    lsf_multi(int nd) : devlock(64) { ndev = nd; }
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

template <typename T> struct BoxCalls;
template <> struct BoxCalls<double> {
    static int sweep(const double* a, double* b, const double* s, const lsf_box* bx, const int* lo, const int* hi, double dx, double h,
                     int mode, double* sum, void* st) { return lsf_jacobi_sweep_box(a, b, s, bx, lo, hi, dx, h, mode, sum, st); }
    static int bc(const double* a, double* b, const lsf_box* bx, const int* lo, const int* hi, double dx, double* sum, void* st)
    { return lsf_bc_box(a, b, bx, lo, hi, dx, sum, st); }
    static int pack(const double* f, const lsf_box* bx, const int* lo, const int* hi, double* buf, void* st) { return lsf_pack_box(f, bx, lo, hi, buf, st); }
    static int unpack(double* f, const lsf_box* bx, const int* lo, const int* hi, const double* buf, void* st) { return lsf_unpack_box(f, bx, lo, hi, buf, st); }
};
template <> struct BoxCalls<float> {
    static int sweep(const float* a, float* b, const float* s, const lsf_box* bx, const int* lo, const int* hi, double dx, double h,
                     int mode, double* sum, void* st) { return lsf_jacobi_sweep_box_f32(a, b, s, bx, lo, hi, dx, h, mode, sum, st); }
    static int bc(const float* a, float* b, const lsf_box* bx, const int* lo, const int* hi, double dx, double* sum, void* st)
    { return lsf_bc_box_f32(a, b, bx, lo, hi, dx, sum, st); }
    static int pack(const float* f, const lsf_box* bx, const int* lo, const int* hi, float* buf, void* st) { return lsf_pack_box_f32(f, bx, lo, hi, buf, st); }
    static int unpack(float* f, const lsf_box* bx, const int* lo, const int* hi, const float* buf, void* st) { return lsf_unpack_box_f32(f, bx, lo, hi, buf, st); }
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
    LSFM_HIP(hipMalloc((void**)&R.d_sum, sizeof(double)));
    LSFM_HIP(hipHostMalloc((void**)&R.h_sum, 2 * sizeof(double), hipHostMallocDefault));
    for (int f = 0; f < 6; ++f) {
        if (R.g.nb[f] < 0) continue;
        const size_t b = (size_t)R.g.send[f].vol() * sizeof(T);
        LSFM_HIP(hipMalloc((void**)&R.sendb[f], b));
        for (int q = 0; q < 2; ++q) LSFM_HIP(hipMalloc((void**)&R.recvb[q][f], b));
        LSFM_HIP(hipEventCreateWithFlags(&R.sent[f], hipEventDisableTiming));
    }
    LSFM_HIP(hipEventCreateWithFlags(&R.halo, hipEventDisableTiming));
    for (int q = 0; q < 2; ++q) LSFM_HIP(hipEventCreateWithFlags(&R.done[q], hipEventDisableTiming));
}

template <typename T>
void free_rank(RankState<T>& R)
{
    (void)hipSetDevice(R.dev);
    if (R.compute) (void)hipStreamSynchronize(R.compute);
    if (R.comm) (void)hipStreamSynchronize(R.comm);
    for (int q = 0; q < 2; ++q) {
        if (R.buf[q]) (void)hipFree(R.buf[q]);
        if (R.done[q]) (void)hipEventDestroy(R.done[q]);
        for (int f = 0; f < 6; ++f)
            if (R.recvb[q][f]) (void)hipFree(R.recvb[q][f]);
    }
    for (int f = 0; f < 6; ++f) {
        if (R.sendb[f]) (void)hipFree(R.sendb[f]);
        if (R.sent[f]) (void)hipEventDestroy(R.sent[f]);
    }
    if (R.phiS) (void)hipFree(R.phiS);
    if (R.d_sum) (void)hipFree(R.d_sum);
    if (R.h_sum) (void)hipHostFree(R.h_sum);
    if (R.halo) (void)hipEventDestroy(R.halo);
    if (R.compute) (void)hipStreamDestroy(R.compute);
    if (R.comm) (void)hipStreamDestroy(R.comm);
    R = RankState<T>{};
}

struct RunShared {
    double dx, h, tol, den;
    int iter, mode;
    std::vector<double> vals;   // block sums of the sweep being judged
    std::vector<double> trace;  // global RMS per sweep
    std::atomic<int> failed{0};
    int sweeps = 0;
};

// one worker thread = one block
template <typename T>
void worker(lsf_multi* M, std::vector<RankState<T>>* ranks, int r, SpinBarrier* bar, RunShared* S)
{
    using C = BoxCalls<T>;
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

    auto enqueue_sends = [&](int s) {
        if (R.rc) return;
        std::lock_guard<std::mutex> lk(dl);
        const int q = s & 1;
        const T* a_in = R.buf[q];
        // the input of this sweep is the output of the previous one (compute stream)
        if (s > 0) LSFM_HIP(hipStreamWaitEvent(R.comm, R.done[(s - 1) & 1], 0));
        for (int f = 0; f < 6; ++f) {
            const int p = R.g.nb[f];
            if (p < 0) continue;
            LSFM_LSF(C::pack(a_in, &bx, R.g.send[f].lo, R.g.send[f].hi, R.sendb[f], R.comm));
            RankState<T>& P = (*ranks)[p];
            LSFM_HIP(hipMemcpyPeerAsync(P.recvb[q][f ^ 1], P.dev, R.sendb[f], R.dev, (size_t)R.g.send[f].vol() * sizeof(T), R.comm));
            LSFM_HIP(hipEventRecord(R.sent[f], R.comm));
        }
    };
    auto enqueue_sweep = [&](int s) {
        if (R.rc) return;
        std::lock_guard<std::mutex> lk(dl);
        const int q = s & 1;
        T* a_in = R.buf[q];
        T* a_out = R.buf[q ^ 1];
        for (int f = 0; f < 6; ++f) {
            const int p = R.g.nb[f];
            if (p < 0) continue;
            LSFM_HIP(hipStreamWaitEvent(R.comm, (*ranks)[p].sent[f ^ 1], 0)); // recorded before the barrier in front of us
            LSFM_LSF(C::unpack(a_in, &bx, R.g.recv[f].lo, R.g.recv[f].hi, R.recvb[q][f], R.comm));
        }
        LSFM_HIP(hipEventRecord(R.halo, R.comm));
        LSFM_HIP(hipMemsetAsync(R.d_sum, 0, sizeof(double), R.compute));
        LSFM_LSF(lsf_sumsq_begin(R.compute));
        int rc = LSF_OK;
        if (R.g.core.vol() > 0)
            rc = C::sweep(a_in, a_out, R.phiS, &bx, R.g.core.lo, R.g.core.hi, S->dx, S->h, S->mode, R.d_sum, R.compute); // overlaps the exchange
        hipError_t he = hipStreamWaitEvent(R.compute, R.halo, 0);
        for (size_t k = 0; k < R.g.rims.size() && rc == LSF_OK && he == hipSuccess; ++k)
            if (R.g.rims[k].vol() > 0)
                rc = C::sweep(a_in, a_out, R.phiS, &bx, R.g.rims[k].lo, R.g.rims[k].hi, S->dx, S->h, S->mode, R.d_sum, R.compute);
        if (rc == LSF_OK && he == hipSuccess) rc = C::bc(a_in, a_out, &bx, own_lo, own_hi, S->dx, R.d_sum, R.compute);
        const int rc2 = lsf_sumsq_end(R.compute); // always close the bracket
        if (he != hipSuccess) { R.rc = LSF_ERR_HIP, R.err = hipGetErrorString(he); return; }
        LSFM_LSF(rc);
        LSFM_LSF(rc2);
        LSFM_HIP(hipMemcpyAsync(R.h_sum + q, R.d_sum, sizeof(double), hipMemcpyDeviceToHost, R.compute));
        LSFM_HIP(hipEventRecord(R.done[q], R.compute));
    };
    // judge sweep s (all threads compute the same number): returns true to stop
    auto judge = [&](int s) -> bool {
        if (!R.rc) {
            if (hipEventSynchronize(R.done[s & 1]) != hipSuccess) R.rc = LSF_ERR_HIP, R.err = "event synchronisation failed";
        }
        S->vals[r] = R.rc ? std::nan("") : R.h_sum[s & 1];
        if (R.rc) fail_all();
        bar->wait();
        double tot = 0.0;
        for (int k = 0; k < nr; ++k) tot += S->vals[k]; // rank order: fixed
        const double quo = tot / S->den;
        const double rms = quo >= 0.0 ? std::sqrt(quo) : std::nan(""); // the wrapped INTEGER*4 product may be negative
        if (r == 0) S->trace.push_back(rms), S->sweeps = s + 1;
        bar->wait(); // vals may be overwritten from here on
        return rms < S->tol || rms != rms; // a failed block reports NaN: every thread takes the same decision
    };

    const int max_sweeps = S->iter + 1; // DO n = 0, iter (subs.f90:735)
    bool stop = false;
    int s = 0;
    for (; s < max_sweeps && !stop; ++s) {
        enqueue_sends(s);
        if (R.rc) fail_all();
        bar->wait(); // every block has recorded the `sent` events of this sweep
        enqueue_sweep(s);
        if (R.rc) fail_all();
        if (s >= 1) stop = judge(s - 1); // one sweep late: sweep s is already in the queues
    }
    if (!stop && s >= 1) (void)judge(s - 1);
    if (!R.rc) {
        (void)hipStreamSynchronize(R.comm);
        (void)hipStreamSynchronize(R.compute);
    }
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
    S.vals.assign(nr, 0.0);
    for (auto& R : ranks) { // phiS = phi on entry (subs.f90:731)
        if (hipSetDevice(R.dev) != hipSuccess ||
            hipMemcpy(R.phiS, R.buf[0], R.g.npoints() * sizeof(T), hipMemcpyDeviceToDevice) != hipSuccess) {
            *err = "copying the sign field failed";
            return LSF_ERR_HIP;
        }
    }
    SpinBarrier bar(nr);
    std::vector<std::thread> th;
    for (int r = 0; r < nr; ++r) th.emplace_back(worker<T>, M, &ranks, r, &bar, &S);
    for (auto& t : th) t.join();
    for (auto& R : ranks)
        if (R.rc) {
            *err = R.err;
            return R.rc;
        }
    // stop sweep: the first sweep whose RMS is below the tolerance or NaN; later entries of the trace (the sweep that
    // was already enqueued) are dropped
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
