// One-kernel translation unit for iterating on dm_train_rows_kernel: compiles in 15 s instead of 4 min and prints the register / scratch use
// (how the fp64 kernel's 1 556 spilled registers — weight-fragment addresses hoisted out of the tile loop — were found and removed):
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
//       tools/train_kernel_tu.hip -o /tmp/tk.o 2>&1 | grep -A9 "Function Name: _Z20dm_train_rows"
#include "../include/dismember_hip.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define DM_IF_ALL_E(...)
#include "../dismember_amd/csrc/beam_kernel.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_w.hip.inc"
#include "../dismember_amd/csrc/beam_kernel_f64.hip.inc"
#include "../dismember_amd/csrc/rows_kernel.hip.inc"
#include "../dismember_amd/csrc/train_kernel.hip.inc"
template __global__ void dm_train_rows_kernel<double, 128, 16>(TrainParamsT<double>);
template __global__ void dm_train_rows_kernel<double, 128, 2>(TrainParamsT<double>);
template __global__ void dm_train_rows_kernel<double, 128, 8>(TrainParamsT<double>);
template __global__ void dm_train_rows_kernel<float, 128, 16>(TrainParamsT<float>);
template __global__ void dm_train_rows_kernel<double, 128, DM_MAXL, true>(TrainParamsT<double>);
