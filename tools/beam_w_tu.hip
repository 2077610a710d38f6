// One-kernel translation unit for the headline beam kernel (beam_kernel_w.hip.inc), for the compiler's resource report and the ISA:
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -S --cuda-device-only tools/beam_w_tu.hip -o /tmp/bw.s -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <type_traits>
#include <utility>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_w.hip.inc"
template __global__ void dm_beam_w_kernel<128, 3>(BeamParams);
