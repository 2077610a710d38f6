// One-kernel translation unit for the general-rows split kernel (rows_kernel.hip.inc):
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -c tools/rows_split_tu.hip -o /tmp/rs.o -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/rows_kernel.hip.inc"
template __global__ void dm_din_rows_split_kernel<128>(RowsSplitParams);
template __global__ void dm_din_rows_split_l_kernel<128, 10>(RowsSplitParams);
