// Brute-force check of the division sequence of lsf_cell.hpp (STRICT arithmetic) against the IEEE division the compiler
// emits (A is what the library uses; B is a variant it does NOT use: Markstein's theorem covers it only for faithful
// first quotients):  hipcc --offload-arch=gfx950 -O3 -I levelsetfortran_amd/csrc -o /tmp/divcheck profiles/micro/divcheck.hip && /tmp/divcheck
//   A: div_by(n, d, recip_refined(d))            -- the hardware sequence without its scale / fixup frame
//   B: div_by(n, d, 1. / d)                      -- the same with the correctly rounded reciprocal (one correction)
// Operands: n, d random significands, exponents of n in [-40, 40], of d in [-660, 660] (the WENO divisors reach 1e-198).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "lsf_cell.hpp"
__device__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__global__ void k(unsigned long long* bad, int rounds, int dexp)
{
    const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long a = 0, b = 0, c = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint64_t u = mix(id * 0x10001ull + r), v = mix(u);
        const int en = (int)(mix(v) % 81) - 40, ed = (int)(mix(v + 1) % (2 * dexp + 1)) - dexp;
        double n = __longlong_as_double((long long)((u >> 12) | ((uint64_t)(1023 + en) << 52)));
        double d = __longlong_as_double((long long)((v >> 12) | ((uint64_t)(1023 + ed) << 52)));
        if (u & 1) n = -n;
        const double q = n / d;
        const double rd = 1. / d;
        a += __double_as_longlong(lsf::div_by(n, d, lsf::recip_refined(d))) != __double_as_longlong(q);
        b += __double_as_longlong(lsf::div_by(n, d, rd)) != __double_as_longlong(q);
        // the unframed square root of finish_update<STRICT> against the compiler's: |d| has exponents up to +-dexp
        const double x = __builtin_fabs(d);
        if (x >= 1.0e-200 && x <= 1.0e200) c += __double_as_longlong(lsf::sqrt_unframed(x)) != __double_as_longlong(__builtin_sqrt(x));
    }
    atomicAdd(bad + 0, a), atomicAdd(bad + 1, b), atomicAdd(bad + 2, c);
}
int main()
{
    unsigned long long* d_bad; (void)hipMalloc(&d_bad, 24);
    for (int dexp : {6, 660}) {
        (void)hipMemset(d_bad, 0, 24);
        const int rounds = 4096, blocks = 8192;
        k<<<blocks, 256>>>(d_bad, rounds, dexp);
        unsigned long long h[3]; (void)hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost);
        printf("divisor exponents +-%d: %.3g quotients; mismatches vs n / d:  A (refined reciprocal, 1 correction) %llu   B (1./d, 1 correction) %llu"
               "   unframed sqrt vs sqrt %llu\n",
               dexp, (double)rounds * blocks * 256, h[0], h[1], h[2]);
    }
    return 0;
}
