// Micro-benchmark: cost of a wave-load in which every lane touches a different 128-B line (row-per-lane
// marching along x), L1/L2-resident, versus a coalesced load.  hipcc --offload-arch=gfx950 -O3 gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(64) void k(const double* __restrict__ p, double* out, long rowstride, int steps, int reps)
{
    const int lane = threadIdx.x;
    const double* base = p + (long)blockIdx.x * 64 * rowstride;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) { // row per lane, one double per step
            const double* q = base + (long)lane * rowstride;
            for (int s = 0; s < steps; ++s) acc += q[s];
        } else if (MODE == 1) { // row per lane, two doubles (dwordx4) per two steps
            const double2* q = (const double2*)(base + (long)lane * rowstride);
            for (int s = 0; s < steps / 2; ++s) { double2 v = q[s]; acc += v.x + v.y; }
        } else if (MODE == 2) { // skewed: lane l is at x = s - l (clamped) in row l: the y-marching pattern
            for (int s = 0; s < steps; ++s) { int x = s - (lane & 15); x = x < 0 ? 0 : x; acc += base[(long)lane * rowstride + x]; }
        } else { // coalesced: 64 consecutive doubles of row s
            for (int s = 0; s < steps; ++s) acc += base[(long)(s & 63) * rowstride + lane];
        }
    }
    if (acc == 123.456) out[0] = acc;
}
int main()
{
    const long rowstride = 512, rows = 64L * 4096;
    double* d; double* o;
    hipMalloc(&d, rows * rowstride * 8); hipMalloc(&o, 8);
    hipMemset(d, 0, rows * rowstride * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 256, reps = 20;
    for (int blocks : {256, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, o, rowstride, steps, reps);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, o, rowstride, steps, reps);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, o, rowstride, steps, reps);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, o, rowstride, steps, reps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it == 1) {
                    const double instr = (double)blocks * reps * (mode == 1 ? steps / 2 : steps);
                    // wave-load instructions per CU per microsecond and ns per instruction per CU
                    printf("blocks %5d mode %d: %.3f ms, %.1f ns per wave-load per CU (256 CUs), %.1f cycles at 2.1 GHz\n", blocks, mode, ms,
                           ms * 1e6 / (instr / 256.0), ms * 1e6 / (instr / 256.0) * 2.1);
                }
            }
        }
    }
    return 0;
}
