#!/bin/bash
# kernel resource usage (registers, scratch, LDS, occupancy) of one translation unit: resusage.sh lsf_api.hip [filter] [EXTRA flags]
cd "$(dirname "$0")/../../levelsetfortran_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $3 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /tmp/${1%.hip}.s $1 2>&1 |
  grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "${2:-.}" | c++filt | sed -E 's/Function Name: //'
