// Standalone timing harness for the split-fp16 history GEMMs (dr_kernel.hip.inc) at the config-5 shape (3 000 columns, K = 1 280,
// rows = argv[1], default 16 384; random rows of a 2 M-item table), with knock-outs of the 256 x 256 kernel:
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 [-DDR_KO_GLOAD|-DDR_KO_ALOAD|-DDR_KO_BLOAD|-DDR_KO_MFMA] -Itools tools/dr_gemm_probe.hip -o tools/_bin/dr_gemm_probe[_x]
// Prints the average launch time of both kernels.  A knocked-out build computes nothing meaningful; only its time is read.
#include "dr_gemm_tu.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 16384, items = 2000000;
  const int N = 3000, E = 128, L = 10, Kd = L * E;
  float *emb, *C, *bias, *zero; _Float16 *Bp; int32_t *gidx;
  CK(hipMalloc(&emb, items * E * 4)); CK(hipMalloc(&C, M * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&zero, 4096));
  CK(hipMalloc(&Bp, (size_t)2 * N * Kd * 2 + 4096)); CK(hipMalloc(&gidx, M * L * 4));
  CK(hipMemset(emb, 0x3c, items * E * 4)); CK(hipMemset(Bp, 0x3c, (size_t)2 * N * Kd * 2)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(zero, 0, 4096));
  std::vector<int32_t> g(M * L);
  unsigned long long x = 88172645463325252ull;
  for (auto &v : g) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (x % 100) < 15 ? -1 : (int32_t)(x % items); }
  CK(hipMemcpy(gidx, g.data(), g.size() * 4, hipMemcpyHostToDevice));
  DrGemmSplitParams q{};
  q.emb = emb; q.gidx = gidx; q.Lg = L; q.E = E; q.Bp = Bp; q.bias = bias; q.zero = zero; q.C = C; q.ldc = N; q.M = M; q.N = N; q.Kd = Kd;
  q.a_scale = 1.0f; q.c_unscale = 1.0f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  _Float16 *embp; CK(hipMalloc(&embp, items * E * 4));
  hipLaunchKernelGGL(dr_split_rows_kernel, dim3(4096), dim3(256), 0, 0, emb, items, E, 1.0f, embp);
  q.embp = embp; q.zeroh = (const _Float16 *)zero; q.Bt = Bp;
  for (int x = 0; x < 2; x++) {
    const int reps = 20;
    dim3 grid((N + DR_TN - 1) / DR_TN, (unsigned)((M + DR_TM - 1) / DR_TM));
    for (int it = 0; it < reps + 3; it++) {
      if (it == 3) CK(hipEventRecord(e0, 0));
      if (x) hipLaunchKernelGGL(dr_gemm_split_x_kernel, dim3(dr_gemm_x_grid(M, N)), dim3(512), 0, 0, q);
      else hipLaunchKernelGGL(dr_gemm_split_kernel, grid, dim3(256), 0, 0, q);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rows %lld, tile %s: %.4f ms per launch\n", (long long)M, x ? "256x256 (pre-split operands, direct-to-LDS)" : "128x128", ms / reps);
  }
  return 0;
}
