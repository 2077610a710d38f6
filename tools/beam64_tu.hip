// One-kernel translation unit for the fused fp64 OTM beam kernel (beam_kernel_f64.hip.inc):
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -c tools/beam64_tu.hip -o /tmp/b64.o -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_w.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_f64.hip.inc"
template __global__ void dm_beam64_kernel<128, 3>(Beam64Params);
template __global__ void dm_beam64_kernel<128, 4>(Beam64Params);
template __global__ void dm_beam64_kernel<128, 4, 2>(Beam64Params);
