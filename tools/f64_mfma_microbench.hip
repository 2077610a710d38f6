// fp64 matrix-pipe ceiling by occupancy: one wave per SIMD reaches 35 TFLOP/s, two reach 77 of the 78.6 peak (why dr_gemm_kernel<double> is built for two workgroups per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/f64_mfma_microbench.hip -o /tmp/f64mb && /tmp/f64mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int WPS>
__global__ __launch_bounds__(256 * WPS) void k(double *out, int iters) {
  f64x4 acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (f64x4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WPS> void run() {
  double *out; hipMalloc(&out, 256 * 512 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<WPS><<<256, 256 * WPS>>>(out, 100);
  hipEventRecord(e0); k<WPS><<<256, 256 * WPS>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 256.0 * 4 * WPS * iters * 16 * 2048.0;
  printf("%d wave(s) per SIMD: %.2f TFLOP/s f64 (%.1f ms)\n", WPS, fl / ms / 1e9, ms);
}
int main() { run<1>(); run<2>(); return 0; }
