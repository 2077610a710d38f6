// One-file translation unit for the Deep-Retrieval sliced pipeline's kernels (register / spill report):
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -c tools/dr_sliced_tu.hip -o /tmp/drs.o -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <type_traits>
#include <utility>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/dr_kernel.hip.inc"
#include "../dismember_amd/csrc/dr_sliced.hip.inc"
template __global__ void drs_stats_kernel<float, 1>(DrsParams<float>, int);
template __global__ void drs_stats_kernel<float, 2>(DrsParams<float>, int);
template __global__ void drs_stats_kernel<double, 1>(DrsParams<double>, int);
template __global__ void drs_stats_kernel<double, 2>(DrsParams<double>, int);
