// lsf_stream.hip -- the instances of k_reinit_gs_stream (lsf_skew.hpp: the dataflow launch of the exact ordering with column
// continuation), in a translation unit of their own: they are the largest kernels of the library and compile beside the rest.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "lsf_stream.hpp"

namespace lsf {

// blocks: the resident blocks the device holds (every block is a loop over tiles).  Returns a hipError_t.
int launch_gs_stream(int wy, int wz, int by, bool strict, hipStream_t st, const GsArgs& fa, int cus, int* blocks_out)
{
    hipError_t err = hipSuccess;
    int blocks = 0;
#define LSF_STREAM_ONE(WY_, WZ_, BY_, ST_)                                                                                          \
    do {                                                                                                                            \
        int per_cu = 0;                                                                                                             \
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_reinit_gs_stream<16, WY_, WZ_, BY_, ST_>, 64 * WY_ * WZ_, 0); \
        if (err != hipSuccess) break;                                                                                               \
        blocks = (int)std::min<long>((long)std::max(per_cu, 1) * cus, fa.total);                                                    \
        if (const char* e_ = getenv("LSF_GS_BLOCKS")) blocks = std::max(1, std::min(blocks, atoi(e_))); /* experiments: fewer resident blocks */ \
        hipLaunchKernelGGL((k_reinit_gs_stream<16, WY_, WZ_, BY_, ST_>), dim3(blocks), dim3(64 * WY_ * WZ_), 0, st, fa);            \
    } while (0)
#define LSF_STREAM_SHAPE(WY_, WZ_, BY_)                  \
    do {                                                 \
        if (strict) LSF_STREAM_ONE(WY_, WZ_, BY_, true); \
        else LSF_STREAM_ONE(WY_, WZ_, BY_, false);       \
    } while (0)
    const int shape = by * 256 + wy * 16 + wz;
#ifdef LSF_DEV_SHAPES // development builds: the two default shapes only (compile time)
    if (shape == 16 * 256 + 0x14) LSF_STREAM_SHAPE(1, 4, 16);
    else LSF_STREAM_SHAPE(2, 2, 5);
#else
    if (shape == 5 * 256 + 0x11) LSF_STREAM_SHAPE(1, 1, 5);
    else if (shape == 5 * 256 + 0x21) LSF_STREAM_SHAPE(2, 1, 5);
    else if (shape == 5 * 256 + 0x41) LSF_STREAM_SHAPE(4, 1, 5);
    else if (shape == 5 * 256 + 0x12) LSF_STREAM_SHAPE(1, 2, 5);
    else if (shape == 5 * 256 + 0x42) LSF_STREAM_SHAPE(4, 2, 5);
    else if (shape == 5 * 256 + 0x24) LSF_STREAM_SHAPE(2, 4, 5);
    else if (shape == 16 * 256 + 0x11) LSF_STREAM_SHAPE(1, 1, 16);
    else if (shape == 16 * 256 + 0x12) LSF_STREAM_SHAPE(1, 2, 16);
    else if (shape == 16 * 256 + 0x13) LSF_STREAM_SHAPE(1, 3, 16);
    else if (shape == 16 * 256 + 0x14) LSF_STREAM_SHAPE(1, 4, 16);
    else LSF_STREAM_SHAPE(2, 2, 5);
#endif
#undef LSF_STREAM_SHAPE
#undef LSF_STREAM_ONE
    if (blocks_out) *blocks_out = blocks;
    return (int)err;
}

} // namespace lsf
