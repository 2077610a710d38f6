// Ceiling probe for v_mfma_f32_16x16x4_f32 on gfx950: what fraction of the 157.3 TFLOP/s dense fp32 peak can a
// 2-waves-per-SIMD kernel sustain (a) with MFMAs only, (b) with the beam kernel's B-fragment traffic (one
// ds_read_b128 per 4 MFMAs, double-buffered), (c) with VALU work threaded between the MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_microbench.hip -o gpurun_out/mfma_mb ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4 *W = (f32x4 *)smem;                       // 64 KB of "fragments"
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 512) W[i] = (f32x4){1.f * i, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int n = 0; n < 8; n++) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 q[8];
#pragma unroll
  for (int j = 0; j < 8; j++) q[j] = (f32x4){1.f + lane, 2.f + j, 3.f, 4.f};
  float v = (float)lane;
  float vv[8];
#pragma unroll
  for (int n = 0; n < 8; n++) vv[n] = (float)(lane + n);
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int jc = 0; jc < 8; jc++)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int n = 0; n < 8; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[jc][t], q[n][t], acc[n], 0, 0, 0);
    } else {
      f32x4 wb[2][8];
#pragma unroll
      for (int n = 0; n < 8; n++) wb[0][n] = W[n * 64 + lane];
#pragma unroll
      for (int jc = 0; jc < 8; jc++) {
        if (jc + 1 < 8) {
#pragma unroll
          for (int n = 0; n < 8; n++) wb[(jc + 1) & 1][n] = W[((jc + 1) * 8 + n) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
          for (int n = 0; n < 8; n++) {
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[jc][t], wb[jc & 1][n][t], acc[n], 0, 0, 0);
            if (MODE == 2) { v = fmaf(v, 1.0001f, 0.5f); v = fmaf(v, 0.9999f, 0.25f); v = fmaf(v, 1.0002f, 0.125f); }
            if (MODE == 4) { vv[n] = fmaf(vv[n], 1.0001f, 0.5f); vv[(n + 3) & 7] = fmaf(vv[(n + 3) & 7], 0.9999f, 0.25f); vv[(n + 5) & 7] = fmaf(vv[(n + 5) & 7], 1.0002f, 0.125f); }
            if (MODE == 5) { vv[n] = fmaf(vv[n], 1.0001f, 0.5f); }
            if (MODE == 6 && (n & 3) == 0) { vv[n] = fmaf(vv[n], 1.0001f, 0.5f); }
          }
        }
      }
      if (MODE == 3) {      // a serial VALU block between bursts (like softmax + epilogue)
#pragma unroll
        for (int e = 0; e < 300; e++) v = fmaf(v, 1.0001f, 0.5f);
      }
    }
  }
  float s = v;
#pragma unroll
  for (int n = 0; n < 8; n++) s += vv[n];
#pragma unroll
  for (int n = 0; n < 8; n++) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, float *d, int iters) {
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flop = 256.0 * 8 * iters * 256 * 2048;
  printf("%-34s %8.3f ms  %7.2f TFLOP/s  %.3f of 157.3\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *d; hipMalloc(&d, 256 * 512 * 4);
  run<0>("mfma only", d, 2000);
  run<1>("mfma + ds_read_b128 per 4", d, 2000);
  run<2>("mfma + lds + 3 VALU per mfma", d, 2000);
  run<3>("mfma + lds + 300-VALU block", d, 2000);
  run<4>("mfma + lds + 3 indep VALU per mfma", d, 2000);
  run<5>("mfma + lds + 1 indep VALU per mfma", d, 2000);
  run<6>("mfma + lds + 1 VALU per 4 mfma", d, 2000);
  return 0;
}
