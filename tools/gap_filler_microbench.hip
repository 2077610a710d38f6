// What does ONE instruction cost behind an MFMA when a SIMD runs a single wave?  (DESIGN.md §3 "one instruction per gap".)
// A stream of v_mfma_f32_16x16x32_f16 (A from the AccVGPRs, four independent accumulators, as the W1a chain of
// beam_kernel_w.hip.inc) with a filler pattern pinned behind the MFMAs; prints cycles per MFMA for each pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/gap_filler_microbench.hip -o /tmp/gap_mb && /tmp/gap_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ __launch_bounds__(256, 1) void k(const h8 *W, float *out, int iters, unsigned long long *cyc) {
  __shared__ f32x4 lds[1024];
  const int lane = threadIdx.x & 63;
  h8 w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { h8 t = W[i * 64 + lane]; asm volatile("" : "=a"(w[i]) : "0"(t)); }
  lds[threadIdx.x] = (f32x4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 b = W[lane];
  h8 bv[4] = {W[lane + 64], W[lane + 128], W[lane + 192], W[lane + 256]};
  float x[4] = {1.f, 2.f, 3.f, 4.f}, y = 1.0001f;
  f32x2 px[4] = {{1.f, 2.f}, {1.f, 2.f}, {1.f, 2.f}, {1.f, 2.f}}, py = {1.0001f, 0.9999f};
  int ix[4] = {1, -2, 3, -4};
  f32x4 ld = {0, 0, 0, 0}, gl = {0, 0, 0, 0};
  f32x4 lq[8] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const f32x4 *gsrc = (const f32x4 *)W + lane;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < 96; m++) {
      if (V >= 30 && V <= 33) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % (V - 29)]) : "a"(w[(m >> 2) & 15]), "v"(b));
      else if (V == 34) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(bv[m & 3]), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "a"(w[(m >> 2) & 15]), "v"(b));
      PIN();
      const int r = m & 3;
      if (V == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y));
      if (V == 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(r + 2) & 3]) : "v"(y)); }
      if (V == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(px[r]) : "v"(py));
      if (V == 4) { if (m & 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); else asm volatile("v_max_i32 %0, 0, %0" : "+v"(ix[r])); }
      if (V == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
      if (V == 6) { ld = lds[(lane + m) & 1023]; asm volatile("" :: "v"(ld)); }
      if (V == 7 && m % 3 == 0) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[0]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[1]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[2]) : "v"(y)); }
      if (V == 8 && (m & 1)) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(px[r]) : "v"(py));
      if (V == 9 && (m & 1)) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(r + 2) & 3]) : "v"(y)); }
      if (V == 10) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(r + 1) & 3]) : "v"(y)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(r + 2) & 3]) : "v"(y)); }
      if (V == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(px[r]) : "v"(py));
      if (V == 12) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[r]) : "v"(y));
      if (V == 13) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(x[r]) : "v"(y));
      if (V == 14) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[r]));
      if (V == 15) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); ld = lds[(lane + m) & 1023]; asm volatile("" :: "v"(ld)); }
      // loads are consumed eight loads later (a use straight behind the load would only measure its latency)
      if (V == 16 && (m & 3) == 0) { asm volatile("" :: "v"(lq[(m / 4) & 7])); lq[(m / 4) & 7] = lds[(lane + m) & 1023]; }
      if (V == 17 && (m & 7) == 0) { asm volatile("" :: "v"(lq[(m / 8) & 7])); lq[(m / 8) & 7] = lds[(lane + m) & 1023]; }
      if (V == 18 && (m & 3) == 0) { asm volatile("" :: "v"(lq[(m / 4) & 7][0])); lq[(m / 4) & 7][0] = ((const float *)lds)[(lane + m) & 1023]; }
      if (V == 19 && (m & 7) == 0) x[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x[r]), 0x1F | (16 << 10)));
      if (V == 20 && (m & 7) == 0) x[r] = __shfl_xor(x[r], 32);
      if (V == 21 && m % 12 == 0) { asm volatile("" :: "v"(lq[(m / 12) & 7])); lq[(m / 12) & 7] = gsrc[(m / 12) * 8]; }
      if (V == 22 && (m & 3) == 0) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(y)); asm volatile("" :: "v"(lq[(m / 4) & 7])); lq[(m / 4) & 7] = lds[(lane + m) & 1023]; }
      if (V == 23 && (m & 1) == 0) { asm volatile("" :: "v"(lq[(m / 2) & 7])); lq[(m / 2) & 7] = lds[(lane + m) & 1023]; }
      if (V == 24) { asm volatile("" :: "v"(lq[m & 7])); lq[m & 7] = lds[(lane + m) & 1023]; }
      PIN();
    }
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x[0] + x[1] + x[2] + x[3] + px[0][0] + px[1][1] + px[2][0] + px[3][1] + ld[0] + gl[0] + lq[0][0] + lq[1][0] + lq[2][0] + lq[3][0] + lq[4][0] + lq[5][0] + lq[6][0] + lq[7][0];
  s += (float)(ix[0] + ix[1] + ix[2] + ix[3]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[V] = t1 - t0;
}

template <int V>
static void run(const char *name, const h8 *W, float *out, unsigned long long *cyc) {
  const int iters = 2000;
  k<V><<<256, 256>>>(W, out, 50, cyc);
  k<V><<<256, 256>>>(W, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc + V, 8, hipMemcpyDeviceToHost);
  printf("%-52s %6.2f cycles per MFMA\n", name, (double)c / (96.0 * iters));
}

int main() {
  h8 *W; float *out; unsigned long long *cyc;
  hipMalloc(&W, 64 * 64 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 40 * 8);
  hipMemset(W, 0, 64 * 64 * 16);
  run<0>("MFMAs only", W, out, cyc);
  run<30>("MFMAs only, ONE accumulator (dependent chain)", W, out, cyc);
  run<31>("MFMAs only, two accumulators", W, out, cyc);
  run<32>("MFMAs only, three accumulators", W, out, cyc);
  run<33>("MFMAs only, four accumulators", W, out, cyc);
  run<34>("MFMAs only, A from VGPRs", W, out, cyc);
  run<1>("1 v_fma_f32 per gap", W, out, cyc);
  run<2>("2 v_fma_f32 per gap", W, out, cyc);
  run<10>("3 v_fma_f32 per gap", W, out, cyc);
  run<9>("2 v_fma_f32 in every other gap", W, out, cyc);
  run<7>("3 v_fma_f32 in every third gap", W, out, cyc);
  run<3>("1 v_pk_fma_f32 per gap", W, out, cyc);
  run<8>("1 v_pk_fma_f32 in every other gap", W, out, cyc);
  run<11>("1 v_pk_add_f32 per gap", W, out, cyc);
  run<4>("v_max_i32 / v_fma_f32 alternating, 1 per gap", W, out, cyc);
  run<5>("1 v_exp_f32 per gap", W, out, cyc);
  run<12>("1 v_cvt_pk_f16_f32 per gap", W, out, cyc);
  run<13>("1 v_fma_mixlo_f16 per gap", W, out, cyc);
  run<14>("1 DPP v_mov per gap", W, out, cyc);
  run<6>("1 ds_read_b128 per gap", W, out, cyc);
  run<15>("1 v_fma_f32 + 1 ds_read_b128 per gap", W, out, cyc);
  run<24>("1 ds_read_b128 per gap, consumed 8 later", W, out, cyc);
  run<23>("1 ds_read_b128 in every 2nd gap", W, out, cyc);
  run<16>("1 ds_read_b128 in every 4th gap", W, out, cyc);
  run<17>("1 ds_read_b128 in every 8th gap", W, out, cyc);
  run<22>("1 v_fma_f32 + 1 ds_read_b128 in every 4th gap", W, out, cyc);
  run<18>("1 ds_read_b32 in every 4th gap", W, out, cyc);
  run<19>("1 ds_swizzle in every 8th gap", W, out, cyc);
  run<20>("1 ds_bpermute in every 8th gap", W, out, cyc);
  run<21>("1 global_load_dwordx4 in every 12th gap", W, out, cyc);
  return 0;
}
