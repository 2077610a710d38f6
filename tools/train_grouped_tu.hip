// One-kernel-family translation unit for the user-grouped fp64 training kernels (train_grouped_f64.hip.inc): compiles in well under a
// minute with the compiler's resource report —
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -c tools/train_grouped_tu.hip -o /tmp/tg.o -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_w.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_f64.hip.inc"
#include "../dismember_amd/csrc/train_kernel.hip.inc"
#include "../dismember_amd/csrc/train_grouped_f64.hip.inc"
template __global__ void tg_setup_kernel<128>(TgSetupParams);
template __global__ void tg_rows_kernel<128, 3>(TgRowsParams);
template __global__ void tg_wgrad_kernel<128>(TgWgradParams);
template __global__ void tg_user_bwd_kernel<128>(TgUserBwdParams);
