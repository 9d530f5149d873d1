#!/bin/bash
# Registers, scratch and spills of the exact-ordering kernels as hipcc reports them (no GPU needed), and the instruction mix of one
# marching step of k_reinit_gs_persist<16,2,2,5,false>.   bash profiles/micro/kernel_resources.sh > profiles/rNN_kernel_resources.txt
set -e
cd "$(dirname "$0")/../../levelsetfortran_amd/csrc"
T=$(mktemp -d)
cat > $T/k.hip <<'HIP'
#include <hip/hip_runtime.h>
#include "lsf_kernels.hpp"
#include "lsf_boxtile.hpp"
#include "lsf_skew.hpp"
template __global__ void lsf::k_reinit_gs_persist<16, 2, 2, 5, false>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_persist<16, 2, 2, 5, true>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_persist<16, 1, 4, 16, false>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_persist<16, 1, 4, 16, true>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_slab<16, 2, 2, 5, false, false>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_slab<16, 2, 2, 5, false, true>(lsf::GsArgs);
template __global__ void lsf::k_reinit_gs_slab<16, 1, 4, 16, true, false>(lsf::GsArgs);
template __global__ void lsf::k_reinit_jacobi_sh<4, 1>(const double*, double*, const double*, lsf::Box, int, int, int, int, int, int, double, double, double*, const int*, int, int, int, int);
HIP
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I. -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o $T/k.s $T/k.hip 2>&1 |
  grep -E "Function Name|SGPRs:|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//;s/ *\[-Rpass.*//' | paste - - - - - - - - |
  grep -E "k_reinit_gs_|k_reinit_jacobi_sh" | sed 's/Function Name: //'
echo
echo "one marching step of k_reinit_gs_persist<16,2,2,5,false> (between two s_barrier of the march): instructions by mnemonic"
awk '/^_ZN3lsf19k_reinit_gs_persistILi16ELi2ELi2ELi5ELb0EEEvNS_6GsArgsE:/,/s_endpgm/' $T/k.s > $T/p.s
B=($(grep -n "s_barrier" $T/p.s | cut -d: -f1))
sed -n "${B[10]},${B[11]}p" $T/p.s | grep -v "^\s*;" | grep -v "^\." | awk '{print $1}' | sort | uniq -c | sort -rn
rm -rf $T
