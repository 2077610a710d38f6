// Microbenchmark for the one-wave-per-SIMD design (DESIGN.md §3 "W1a in registers"): every wave keeps the 64 fp16 A fragments
// of W1a (hi + lo planes, 256 registers) in its AccVGPRs and feeds them to v_mfma_f32_16x16x32_f16 directly (srcA = AGPR,
// vdst / srcC / srcB = VGPR), so the W1a chain makes NO LDS reads.  Per 16-row "tile": 96 MFMAs + V VALU instructions
// (stand-in for the operand split / softmax / epilogue) + one 512-byte global row gather per lane group.
// Prints the fraction of the fp16 dense peak the chain reaches with 1 wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/agpr_mfma_microbench.hip -o /tmp/agpr_mb && /tmp/agpr_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_av(f32x4 &acc, const h8 &a_agpr, const h8 &b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a_agpr), "v"(b));
}

template <int VALU, bool GATHER>
__global__ __launch_bounds__(256, 1) void k(const h8 *W, const f32x4 *emb, int64_t nrows, float *out, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 w[64];
#pragma unroll
  for (int i = 0; i < 64; i++) { h8 t = W[i * 64 + lane]; asm volatile("" : "=a"(w[i]) : "0"(t)); }
  f32x4 tot = {0, 0, 0, 0};
  f32x4 q[8];
  uint64_t rs = (blockIdx.x * 4 + wave) * 7919u + lane;
#pragma unroll
  for (int j = 0; j < 8; j++) q[j] = (f32x4){1.f + j, 2.f, 3.f, 4.f};
  for (int t = 0; t < tiles; t++) {
    h8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {      // stand-in for the split: 16 VALU ops per k-step when VALU == 64
      f32x4 a = q[2 * s], b = q[2 * s + 1];
      if (VALU >= 64) { a = a * 1.0001f + b; b = b * 0.9999f + a; a = a * 1.0001f + b; b = b * 0.9999f + a; }
      qh[s] = __builtin_bit_cast(h8, a); ql[s] = __builtin_bit_cast(h8, b);
    }
    if (GATHER) {
      rs = rs * 6364136223846793005ull + 1442695040888963407ull;
      const f32x4 *src = emb + ((rs >> 20) % (uint64_t)nrows) * 32 + (lane >> 4);
#pragma unroll
      for (int j = 0; j < 8; j++) q[j] = src[4 * j];
    }
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; nt++) acc[nt] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
      for (int nt = 0; nt < 8; nt++) mfma_av(acc[nt], w[32 + s * 8 + nt], qh[s]);
#pragma unroll
      for (int nt = 0; nt < 8; nt++) mfma_av(acc[nt], w[s * 8 + nt], ql[s]);
#pragma unroll
      for (int nt = 0; nt < 8; nt++) mfma_av(acc[nt], w[s * 8 + nt], qh[s]);
    }
    asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
#pragma unroll
    for (int nt = 0; nt < 8; nt++) tot += acc[nt];
  }
  out[blockIdx.x * 256 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3] + q[0][0];
}

template <int VALU, bool GATHER>
static void run(const char *name, const h8 *W, const f32x4 *emb, int64_t nrows, float *out, int tiles) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<VALU, GATHER><<<256, 256>>>(W, emb, nrows, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<VALU, GATHER><<<256, 256>>>(W, emb, nrows, out, tiles);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 256.0 * 4 * tiles * 96 * 16384.0;
  printf("%-34s %8.3f ms  %7.1f TFLOP/s  %.3f of 2516.6 (fp16 dense)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2516.6);
}

int main() {
  h8 *W; f32x4 *emb; float *out;
  const int64_t nrows = 1 << 22;     // 2 GB of 512-byte rows
  hipMalloc(&W, 64 * 64 * 16); hipMalloc(&emb, nrows * 512); hipMalloc(&out, 256 * 256 * 4);
  hipMemset(W, 0, 64 * 64 * 16); hipMemset(emb, 0, nrows * 512);
  const int tiles = 20000;
  run<0, false>("mfma chain only", W, emb, nrows, out, tiles);
  run<64, false>("+ 64 VALU per tile", W, emb, nrows, out, tiles);
  run<64, true>("+ 64 VALU + random row gather", W, emb, nrows, out, tiles);
  return 0;
}
