// One-kernel translation unit for the Deep-Retrieval history GEMMs (dr_kernel.hip.inc): compiles in seconds with the compiler's
// resource report —
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -c tools/dr_gemm_tu.hip -o /tmp/drg.o -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dm_h2 __attribute__((ext_vector_type(2)));
#define DM_DR_GEMM_TU
__device__ __forceinline__ void dm_split8(const f32x4 &u, const f32x4 &v, float s, dm_h8 &hi, dm_h8 &lo) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const float x[8] = {u[0] * s, u[1] * s, u[2] * s, u[3] * s, v[0] * s, v[1] * s, v[2] * s, v[3] * s};
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const dm_h2 a = __builtin_convertvector((f32x2_){x[i], x[i + 1]}, dm_h2);
    hi[i] = a[0]; hi[i + 1] = a[1];
    const dm_h2 b = __builtin_convertvector((f32x2_){x[i] - (float)a[0], x[i + 1] - (float)a[1]}, dm_h2);
    lo[i] = b[0]; lo[i + 1] = b[1];
  }
}
#include "../dismember_amd/csrc/dr_kernel.hip.inc"
