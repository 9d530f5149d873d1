// Which streaming copy does this chip do fastest?  (lsf_copy_bandwidth's kernel is the winner; bench.py "roofline.peak_measured").
// hipcc --offload-arch=gfx950 -O3 -o build/exp/copy_bw profiles/micro/copy_bw.hip && build/exp/copy_bw   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_ilp4(const uint4* __restrict__ s, uint4* __restrict__ d, long n)
{
    const long st = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n; i += 4 * st) {
        const uint4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
        d[i] = a, d[i + st] = b, d[i + 2 * st] = c, d[i + 3 * st] = e;
    }
    for (; i < n; i += st) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_one(const uint4* __restrict__ s, uint4* __restrict__ d, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_one_nt(const uint4* __restrict__ s, uint4* __restrict__ d, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const v4u v = __builtin_nontemporal_load((const v4u*)s + i);
        __builtin_nontemporal_store(v, (v4u*)d + i);
    }
}
// a block owns a contiguous chunk of U * 256 vectors, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const uint4* __restrict__ s, uint4* __restrict__ d, long n)
{
    const long base = (long)blockIdx.x * (U * 256) + threadIdx.x;
    v4u v[U];
    const v4u* sv = (const v4u*)s;
    v4u* dv = (v4u*)d;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(sv + i) : sv[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            if (NT) __builtin_nontemporal_store(v[u], dv + i);
            else dv[i] = v[u];
        }
    }
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ s, unsigned* __restrict__ out, long n)
{
    const long st = (long)gridDim.x * 256;
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += st) {
        const uint4 v = s[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ d, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = make_uint4(1u, 2u, 3u, (unsigned)i);
}

int main()
{
    const size_t bytes = 1ull << 30;
    const long n = (long)(bytes / 16);
    uint4 *a, *b;
    unsigned* o;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 0x3c, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double moved, auto&& launch) {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-28s %8.3f ms  %8.1f GB/s\n", name, best, moved / (best * 1e-3) / 1e9);
    };
    const unsigned gb = (unsigned)((n + 255) / 256);
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, sizeof nm, "ilp4 grid %d", g);
        run(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_ilp4, dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    run("one vector per lane", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_one, dim3(gb), dim3(256), 0, 0, a, b, n); });
    run("one vector per lane, nt", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_one_nt, dim3(gb), dim3(256), 0, 0, a, b, n); });
    run("chunk 2", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_chunk<2, false>), dim3((gb + 1) / 2), dim3(256), 0, 0, a, b, n); });
    run("chunk 4", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_chunk<4, false>), dim3((gb + 3) / 4), dim3(256), 0, 0, a, b, n); });
    run("chunk 8", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_chunk<8, false>), dim3((gb + 7) / 8), dim3(256), 0, 0, a, b, n); });
    run("chunk 4 nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_chunk<4, true>), dim3((gb + 3) / 4), dim3(256), 0, 0, a, b, n); });
    run("chunk 8 nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_chunk<8, true>), dim3((gb + 7) / 8), dim3(256), 0, 0, a, b, n); });
    run("hipMemcpyDtoD", 2.0 * bytes, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    run("read only (grid 8192)", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, a, o, n); });
    run("write only", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_write, dim3(gb), dim3(256), 0, 0, b, n); });
    return 0;
}
