// How many 16-row tiles must share one pass over the W1a fragments before v_mfma_f32_16x16x32_f16 stops being starved by the
// LDS?  Every wave streams the 64 KB of fp16 hi / lo planes (ds_read_b128, the beam kernel's fragment order) and issues, per
// (k-step, feature tile), 3 MFMAs for each of T row tiles.  T = 1 is the beam kernel's split tile.  WPS = waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_mfma_microbench.hip -o tools/_bin/lds_mfma_mb ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int T, int WPS, int MODE>
__global__ __launch_bounds__(256 * WPS, WPS) void k(float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  h8 *W = (h8 *)smem;                              // [2 planes][4 k-steps][8 feature tiles][64 lanes] x 16 B = 64 KB
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256 * WPS) { h8 v; for (int e = 0; e < 8; e++) v[e] = (_Float16)(0.001f * ((i + e) & 63)); W[i] = v; }
  __syncthreads();
  if (MODE == 2 && threadIdx.x >= 256) return;       // the second wave of every SIMD stays out
  f32x4 acc[T][8];
  h8 qh[T][4], ql[T][4];
#pragma unroll
  for (int t = 0; t < T; t++) {
#pragma unroll
    for (int n = 0; n < 8; n++) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int e = 0; e < 8; e++) { qh[t][s][e] = (_Float16)(0.01f * (lane + e + t)); ql[t][s][e] = (_Float16)(1e-5f * (lane + s)); }
  }
  int off = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(off));      // opaque: the fragment reads are NOT loop-invariant for the compiler
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
      for (int nb = 0; nb < 8; nb += 4) {
        h8 w0[4], w1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (MODE == 1) { w0[i] = qh[0][(s + i) & 3]; w1[i] = ql[0][(s + i) & 3]; }      // no LDS traffic at all
          else { w0[i] = W[(s * 8 + nb + i) * 64 + lane + off]; w1[i] = W[((4 + s) * 8 + nb + i) * 64 + lane + off]; }
        }
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
          for (int i = 0; i < 4; i++) acc[t][nb + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[i], qh[t][s], acc[t][nb + i], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
          for (int i = 0; i < 4; i++) acc[t][nb + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[i], ql[t][s], acc[t][nb + i], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
          for (int i = 0; i < 4; i++) acc[t][nb + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[i], qh[t][s], acc[t][nb + i], 0, 0, 0);
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int n = 0; n < 8; n++) r += acc[t][n][0] + acc[t][n][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int T, int WPS, int MODE = 0>
static void run(const char *name, float *d_out, int iters) {
  hipFuncSetAttribute((const void *)k<T, WPS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<T, WPS, MODE><<<256, 256 * WPS, 65536>>>(d_out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<T, WPS, MODE><<<256, 256 * WPS, 65536>>>(d_out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double mfma = 256.0 * 4 * (MODE == 2 ? 1 : WPS) * iters * 96.0 * T;                       // per launch
  const double tflops = mfma * 16384 / (ms * 1e-3) / 1e12;
  const double lds_gbps = 256.0 * 4 * WPS * iters * 65536.0 / (ms * 1e-3) / 1e9;
  printf("%-34s %8.3f ms  %7.1f TFLOP/s fp16 issued = %.2f of 2516.6   LDS %.0f GB/s = %.0f B/clk/CU at 2.4 GHz   hipError %d\n", name, ms, tflops,
         tflops / 2516.6, lds_gbps, lds_gbps / 256 / 2.4, (int)hipGetLastError());
}

int main() {
  float *d_out;
  hipMalloc(&d_out, 256 * 512 * 4);
  const int iters = 2000;
  run<1, 2>("1 tile / pass, 2 waves per SIMD", d_out, iters);
  run<2, 2>("2 tiles / pass, 2 waves per SIMD", d_out, iters);
  run<1, 1>("1 tile / pass, 1 wave per SIMD", d_out, iters);
  run<2, 1>("2 tiles / pass, 1 wave per SIMD", d_out, iters);
  run<4, 1>("4 tiles / pass, 1 wave per SIMD", d_out, iters);
  run<1, 2, 1>("1 tile, 2 waves/SIMD, NO LDS reads", d_out, iters);
  run<1, 1, 1>("1 tile, 1 wave/SIMD, NO LDS reads", d_out, iters);
  run<1, 2, 2>("1 tile, 2 waves resident, 1 active", d_out, iters);
  run<2, 2, 1>("2 tiles, 2 waves/SIMD, NO LDS reads", d_out, iters);
  return 0;
}
