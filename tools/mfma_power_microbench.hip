// Sustained matrix-pipe rate with quiet and with busy operands (two waves per SIMD, ~20 ms per run, no memory traffic at all):
// the same MFMA stream fed (a) one constant value per lane, (b) eight full-mantissa random values per lane rotating from one MFMA to the
// next.  What the difference shows is the clock the chip sustains under the switching activity of real data — the ceiling a GEMM on
// random operands can reach, as opposed to the nominal peak (2.4 GHz x issue rate).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_microbench.hip -o tools/_bin/mfma_power && tools/_bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <bool BUSY>
__global__ __launch_bounds__(512) void k64(double *out, int iters) {
  f64x4 acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (f64x4){0, 0, 0, 0};
  double a[8], b[8];
  for (int i = 0; i < 8; i++) {
    const uint64_t s = mix(threadIdx.x * 16 + i + 1), t = mix(s);
    // BUSY: random mantissas and signs, exponents near 0 (values in [1, 2) with random sign); quiet: the same value everywhere
    a[i] = BUSY ? __longlong_as_double((s & 0x800fffffffffffffull) | 0x3ff0000000000000ull) : 1.0;
    b[i] = BUSY ? __longlong_as_double((t & 0x800fffffffffffffull) | 0x3ff0000000000000ull) * 1e-3 : 1e-3;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool BUSY>
__global__ __launch_bounds__(512) void k16(float *out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (f32x4){0, 0, 0, 0};
  h8 a[8], b[8];
  for (int i = 0; i < 8; i++)
    for (int e = 0; e < 8; e++) {
      const uint64_t s = mix(threadIdx.x * 128 + i * 8 + e + 1);
      const unsigned short ua = BUSY ? (unsigned short)((s & 0x83ff) | 0x3c00) : 0x3c00, ub = BUSY ? (unsigned short)(((s >> 16) & 0x83ff) | 0x2000) : 0x2000;
      a[i][e] = __builtin_bit_cast(_Float16, ua); b[i][e] = __builtin_bit_cast(_Float16, ub);
    }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 7], b[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool BUSY>
__global__ __launch_bounds__(512) void k16w(float *out, int iters) {      // v_mfma_f32_32x32x16_f16: twice the flops per operand register read
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  h8 a[8], b[8];
  for (int i = 0; i < 8; i++)
    for (int e = 0; e < 8; e++) {
      const uint64_t s = mix(threadIdx.x * 128 + i * 8 + e + 1);
      const unsigned short ua = BUSY ? (unsigned short)((s & 0x83ff) | 0x3c00) : 0x3c00, ub = BUSY ? (unsigned short)(((s >> 16) & 0x83ff) | 0x2000) : 0x2000;
      a[i][e] = __builtin_bit_cast(_Float16, ua); b[i][e] = __builtin_bit_cast(_Float16, ub);
    }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 7], b[(i + (i >> 3)) & 7], acc[i & 3], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K, typename T> void run(const char *name, K kern, T *out, int iters, double flops_per_mfma) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, 512>>>(out, 100);
  hipEventRecord(e0); kern<<<256, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 256.0 * 8 * iters * 16 * flops_per_mfma;
  printf("%-44s %8.1f TFLOP/s (%.1f ms)\n", name, fl / ms / 1e9, ms);
}
int main() {
  void *out; hipMalloc(&out, 256 * 512 * 8);
  for (int rep = 0; rep < 2; rep++) {
    run("v_mfma_f64_16x16x4_f64, quiet operands", k64<false>, (double *)out, 20000, 2048.0);
    run("v_mfma_f64_16x16x4_f64, random operands", k64<true>, (double *)out, 20000, 2048.0);
    run("v_mfma_f32_16x16x32_f16, quiet operands", k16<false>, (float *)out, 40000, 16384.0);
    run("v_mfma_f32_16x16x32_f16, random operands", k16<true>, (float *)out, 40000, 16384.0);
    run("v_mfma_f32_32x32x16_f16, quiet operands", k16w<false>, (float *)out, 20000, 32768.0);
    run("v_mfma_f32_32x32x16_f16, random operands", k16w<true>, (float *)out, 20000, 32768.0);
  }
  return 0;
}
