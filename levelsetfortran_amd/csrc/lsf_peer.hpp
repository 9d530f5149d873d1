// lsf_peer.hpp -- first-contact self-test of the memory model the exact ordering across z slabs relies on between two devices
// (lsf_gs_slabs.hpp, k_reinit_gs_slab).  Included by lsf_api.hip (uses fail, HIPCHK, DeviceRestore).
//
// The slab launches hand data from device to device inside running kernels: results are stored into the neighbour's memory at
// system scope, drained, and announced by a flag the neighbour's kernel polls.  DESIGN.md section 6.1 lists what that assumes:
//   (1) a relaxed system-scope store to a peer's fine-grained allocation has reached that device's memory when the storing
//       wave's s_waitcnt vmcnt(0) returns;
//   (2) the peer's relaxed system-scope loads then see it (no stale copy in its caches);
//   (3) a system-scope atomic max on peer memory is atomic with respect to the owner's own atomics and loads;
//   (4) kernels of different devices that spin on each other's flags run at the same time when one host thread launches them
//       one after the other.
// Nothing in the reference corresponds to this (it is serial, README.md:17).  The test is the message-passing litmus of (1) +
// (2) -- a payload of several cache lines per round, many rounds, both directions at once -- and a two-sided atomic-max
// contest for (3); both are pairs of kernels that wait for each other, which is (4).  Every spin is bounded: a failure is an
// error code that names the assumption, never a hang.  With devA == devB it runs both kernels on one device (the only form a
// one-GPU box can execute; the slab rehearsal with several slabs on one device relies on exactly that).
#pragma once

namespace lsfp {

constexpr int PT_WORDS = 1024;  // payload per round: 8 KB = 64 cache lines, written by 256 lanes
constexpr int PT_ROUNDS = 200;
constexpr int PT_AMAX = 20000;  // atomic-max contest: values per side

struct PeerArgs {
    double* payload_there; // in the OTHER device's memory: this side's payload lands here
    int* flag_there;       //   "round r is complete" (r + 1), stored after the payload has been drained
    const double* payload_here; // in THIS device's memory: what the other side stores
    const int* flag_here;
    int* amax_word;        // one word in device A's memory, both sides raise it
    int* result;           // [0] status (0 ok, else the violated assumption), [1] round / value of the failure, [2] rounds done
    int side;              // 0 = A, 1 = B
    unsigned long long timeout_ticks;
};

// one block of 256 threads per side
static __global__ __launch_bounds__(256) void k_peer_litmus(PeerArgs a)
{
    __shared__ int sh_ok;
    const int tid = threadIdx.x;
    if (a.side < 0) return; // warm-up launch: the code object is loaded on this device before the paired launch (lsf_peer_selftest)
    int status = 0, where = 0;
    // ---- (1) + (2) + (4): message passing, both directions at once -------------------------------------------------
    for (int r = 0; r < PT_ROUNDS && status == 0; ++r) {
        // payload of this round: a function of side, round and word, so that a value left from another round is recognised
        for (int w = tid; w < PT_WORDS; w += 256) lsf::st_sys(a.payload_there + w, (double)(a.side * 1000003 + r * 1031 + w));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (1): the stores have reached the other device
        __syncthreads();
        if (tid == 0) {
            lsf::st_flag_sys(a.flag_there, r + 1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int ok = 1;
            while (lsf::ld_flag_sys(a.flag_here) < r + 1) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                    ok = 0; // the other side never announced this round: the two kernels do not run together (4)
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            sh_ok = ok;
        }
        __syncthreads();
        if (sh_ok == 0) {
            status = 4, where = r;
            break;
        }
        // (2): every word of the other side's payload of THIS round (it cannot have started the next one: it waits for this
        // side's flag of round r + 1 before it overwrites -- see the end of the loop body)
        int bad = 0;
        for (int w = tid; w < PT_WORDS; w += 256) {
            const double v = lsf::ld_sys(a.payload_here + w);
            if (v != (double)((1 - a.side) * 1000003 + r * 1031 + w)) bad = 1;
        }
        if (__syncthreads_or(bad)) {
            status = 1, where = r; // announced but not there: the drain did not cover the store (1) or the load was stale (2)
            break;
        }
        // both sides have read round r before either writes round r + 1: a second handshake on the same flag words (r + 1 -> -(r + 1))
        if (tid == 0) {
            lsf::st_flag_sys(a.flag_there + 16, r + 1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int ok = 1;
            while (lsf::ld_flag_sys(a.flag_here + 16) < r + 1) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            sh_ok = ok;
        }
        __syncthreads();
        if (sh_ok == 0) status = 4, where = r;
    }
    // ---- (3): both sides raise ONE word (in A's memory) by atomic max with interleaved values; a read never goes back ----
    if (status == 0 && tid == 0) {
        int seen = 0;
        for (int i = 0; i < PT_AMAX; ++i) {
            const int v = 2 * i + 1 + a.side; // A: odd, B: even
            const int old = __hip_atomic_fetch_max(a.amax_word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const int now = lsf::ld_flag_sys(a.amax_word);
            if (old < seen || now < v || now < old) { // a maximum that shrinks, or one that does not hold this side's value
                status = 3, where = i;
                break;
            }
            seen = now;
        }
    }
    if (tid == 0) a.result[0] = status, a.result[1] = where, a.result[2] = PT_ROUNDS;
}

} // namespace lsfp

extern "C" int lsf_peer_selftest(int devA, int devB, int* violated)
{
    using namespace lsfp;
    if (violated) *violated = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(LSF_ERR_NO_DEVICE, "no HIP device visible: liblsf_hip has no CPU fallback");
    }
    if (devA < 0 || devB < 0 || devA >= ndev || devB >= ndev) return fail(LSF_ERR_INVALID, "lsf_peer_selftest: device index out of range");
    DeviceRestore restore_;
    const bool distinct = devA != devB;
    if (distinct) {
        for (int q = 0; q < 2; ++q) {
            const int from = q ? devB : devA, to = q ? devA : devB;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can)
                return fail(LSF_ERR_INVALID, "lsf_peer_selftest: the devices cannot access each other's memory");
            HIPCHK(hipSetDevice(from));
            const hipError_t pe = hipDeviceEnablePeerAccess(to, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(LSF_ERR_HIP, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }
    }
    struct Side {
        int dev = 0;
        hipStream_t st = nullptr;
        double* payload = nullptr; // what the other side writes
        int* flags = nullptr;      // [0] round flag, [16] read flag (written by the other side) | [32] the atomic-max word (side A's) | [48..] result
    } S[2];
    S[0].dev = devA, S[1].dev = devB;
    int rc = LSF_OK;
    auto cleanup = [&]() {
        for (Side& s : S) {
            if (hipSetDevice(s.dev) != hipSuccess) continue;
            if (s.st) (void)hipStreamSynchronize(s.st), (void)hipStreamDestroy(s.st);
            if (s.payload) (void)hipFree(s.payload);
            if (s.flags) (void)hipFree(s.flags);
        }
        (void)hipGetLastError();
    };
    for (Side& s : S) {
        if (hipSetDevice(s.dev) != hipSuccess || hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) rc = LSF_ERR_HIP;
        // the allocations the slab launches use for what a neighbour stores into: fine-grained where the devices differ
        const unsigned fl = distinct ? hipDeviceMallocFinegrained : hipDeviceMallocDefault;
        if (rc == LSF_OK && hipExtMallocWithFlags((void**)&s.payload, PT_WORDS * sizeof(double), fl) != hipSuccess) rc = LSF_ERR_HIP;
        if (rc == LSF_OK && hipExtMallocWithFlags((void**)&s.flags, 64 * sizeof(int), fl) != hipSuccess) rc = LSF_ERR_HIP;
        if (rc == LSF_OK && (hipMemset(s.payload, 0xff, PT_WORDS * sizeof(double)) != hipSuccess || hipMemset(s.flags, 0, 64 * sizeof(int)) != hipSuccess))
            rc = LSF_ERR_HIP;
        if (rc != LSF_OK) {
            cleanup();
            return fail(LSF_ERR_HIP, std::string("lsf_peer_selftest: set-up failed: ") + hipGetErrorString(hipGetLastError()));
        }
    }
    // both devices load the kernel BEFORE the paired launch: the first launch on a device can take longer than the 2 s the other
    // side waits, which would read as assumption (4) on a healthy pair
    for (Side& s : S) {
        PeerArgs w{};
        w.side = -1;
        (void)hipSetDevice(s.dev);
        hipLaunchKernelGGL(k_peer_litmus, dim3(1), dim3(256), 0, s.st, w);
    }
    for (Side& s : S) (void)hipSetDevice(s.dev), (void)hipDeviceSynchronize();
    // two kernels that wait for each other, launched one after the other by this thread on two streams (two devices)
    for (int q = 0; q < 2; ++q) {
        PeerArgs a;
        a.payload_there = S[1 - q].payload, a.flag_there = S[1 - q].flags;
        a.payload_here = S[q].payload, a.flag_here = S[q].flags;
        a.amax_word = S[0].flags + 32;
        a.result = S[q].flags + 48;
        a.side = q;
        a.timeout_ticks = 200000000ull; // 2 s of the 100 MHz clock
        if (const char* e = getenv("LSF_PEER_TIMEOUT_TICKS")) a.timeout_ticks = strtoull(e, nullptr, 10); // test hook (its own:
                                                                     // LSF_GS_TIMEOUT_TICKS must reach the slab launches behind this test)
        (void)hipSetDevice(S[q].dev);
        hipLaunchKernelGGL(k_peer_litmus, dim3(1), dim3(256), 0, S[q].st, a);
    }
    int res[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int q = 0; q < 2; ++q) {
        (void)hipSetDevice(S[q].dev);
        if (hipStreamSynchronize(S[q].st) != hipSuccess || hipMemcpy(res[q], S[q].flags + 48, sizeof res[q], hipMemcpyDeviceToHost) != hipSuccess) rc = LSF_ERR_HIP;
    }
    int amax = 0;
    (void)hipSetDevice(S[0].dev);
    if (rc == LSF_OK && hipMemcpy(&amax, S[0].flags + 32, sizeof amax, hipMemcpyDeviceToHost) != hipSuccess) rc = LSF_ERR_HIP;
    cleanup();
    if (rc != LSF_OK) return fail(LSF_ERR_HIP, std::string("lsf_peer_selftest: ") + hipGetErrorString(hipGetLastError()));
    // a side that finds a violated assumption leaves, and the other side then times out: report the finding, not the time-out
    int bad = 0, bad_at = 0;
    for (int q = 0; q < 2; ++q)
        if (res[q][0] && (!bad || (bad == 4 && res[q][0] != 4))) bad = res[q][0], bad_at = res[q][1];
    if (!bad && amax != 2 * (PT_AMAX - 1) + 2) bad = 3; // the final maximum is B's last value
    if (violated) *violated = bad;
    if (bad) {
        static const char* what[5] = {"", "(1)/(2): a payload announced by its flag was not (all) there -- the drain does not cover system-scope stores to the peer, or the peer's loads are stale",
                                      "(2)", "(3): a system-scope atomic max on peer memory lost an update or went back",
                                      "(4): the two kernels did not run at the same time (one never saw the other's flag)"};
        char buf[512];
        snprintf(buf, sizeof buf, "peer self-test between devices %d and %d: assumption %s (DESIGN.md section 6.1); at round / value %d", devA, devB, what[bad], bad_at);
        return fail(LSF_ERR_HIP, buf);
    }
    return LSF_OK;
}
