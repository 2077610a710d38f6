// Microbenchmark: cycles of the in-register bitonic sort (512 keys, 2 waves) in isolation.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../dismember_amd/csrc/beam_kernel.hip.inc"

__global__ __launch_bounds__(512, 2) void sort_bench(unsigned long long *data, unsigned long long *cycles, int reps, int active_waves) {
  __shared__ unsigned long long keys_s[4][512];
  __shared__ int cnt[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 8) cnt[tid] = 0;
  __syncthreads();
  if (wave >= active_waves) return;
  const int team = wave >> 1, tw = wave & 1;
  unsigned long long *keys = keys_s[team];
  int bar_target = 0;
  int *bar_cnt = cnt + team;
  auto team_barrier = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    bar_target += 2;
    if (lane == 0) {
      __hip_atomic_fetch_add(bar_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(bar_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - bar_target < 0) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  const int base = tw * 256 + 4 * lane;
  unsigned long long total = 0;
  unsigned long long key[4];
  for (int r = 0; r < reps; r++) {
    for (int e = 0; e < 4; e++) key[e] = data[((blockIdx.x * 4 + team) * 512 + base + e)] ^ (unsigned long long)r * 0x9E3779B97F4A7C15ull;
    team_barrier();
    unsigned long long t0 = clock64();
    const int Pn = 512;
    for (int k = 2; k <= Pn; k <<= 1) {
      for (int j = k >> 1; j >= 256; j >>= 1) {
        *(ulonglong2 *)(keys + base) = make_ulonglong2(key[0], key[1]);
        *(ulonglong2 *)(keys + base + 2) = make_ulonglong2(key[2], key[3]);
        team_barrier();
        const ulonglong2 o01 = *(const ulonglong2 *)(keys + (base ^ j));
        const ulonglong2 o23 = *(const ulonglong2 *)(keys + (base ^ j) + 2);
        const unsigned long long o[4] = {o01.x, o01.y, o23.x, o23.y};
        const bool keepmin = ((base & j) == 0) == ((base & k) == 0);
        for (int e = 0; e < 4; e++) key[e] = ((key[e] < o[e]) == keepmin) ? key[e] : o[e];
        team_barrier();
      }
      if (k >= 256) dm_sort_stage<128>(key, lane, base, k);
      if (k >= 128) dm_sort_stage<64>(key, lane, base, k);
      if (k >= 64) dm_sort_stage<32>(key, lane, base, k);
      if (k >= 32) dm_sort_stage<16>(key, lane, base, k);
      if (k >= 16) dm_sort_stage<8>(key, lane, base, k);
      if (k >= 8) dm_sort_stage<4>(key, lane, base, k);
      if (k >= 4) dm_sort_stage<2>(key, lane, base, k);
      dm_sort_stage<1>(key, lane, base, k);
    }
    total += clock64() - t0;
  }
  for (int e = 0; e < 4; e++) data[((blockIdx.x * 4 + team) * 512 + base + e)] = key[e];
  if (lane == 0) atomicAdd(cycles, total);
}

int main() {
  const int blocks = 256, reps = 200;
  std::vector<unsigned long long> h((size_t)blocks * 4 * 512);
  unsigned long long s = 12345;
  for (auto &x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = s; }
  unsigned long long *d, *c;
  hipMalloc(&d, h.size() * 8); hipMalloc(&c, 8);
  for (int aw : {2, 8}) {
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(c, 0, 8);
    hipLaunchKernelGGL(sort_bench, dim3(blocks), dim3(512), 0, 0, d, c, reps, aw);
    hipDeviceSynchronize();
    unsigned long long cyc; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> o(h.size()); hipMemcpy(o.data(), d, o.size() * 8, hipMemcpyDeviceToHost);
    bool ok = true; for (int i = 1; i < 512; i++) ok &= o[i - 1] <= o[i];
    printf("active waves/block %d: %.0f cycles per sort per wave, sorted=%d\n", aw, (double)cyc / (blocks * aw * reps), (int)ok);
  }
  return 0;
}
