// Row-gather ceiling of the beam kernels' access pattern, by row size and by how much of the tree is touched.
//
// A wave gathers 16 rows per "tile" exactly as the beam kernels do (lane (r, g) loads float4 at row r, columns 16jc + 4g: eight 16-byte
// loads per 512-byte row), one wave per SIMD (4 per CU), 256 persistent workgroups, tile t+1's loads issued before tile t's are consumed.
// Rows are drawn the way a beam search visits a complete binary tree: every level between `lo` and `hi` equally often, uniformly inside
// the level.  `bytes` = 512 (the embedding row) or 1024 (embedding row + a second 512-byte row from a second table, the "P = W1a emb" idea
// of DESIGN.md §7).  Prints the delivered TB/s — the ceiling a tile that did nothing but gather would see.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_microbench.hip -o /tmp/gmb && /tmp/gmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}

// rows [n_rows][128] floats; second table (or null); per tile the wave's 16 rows: level l uniform in [lo, hi], node uniform in the level
// PAT 0: the MFMA operand pattern (4 lanes per row: an instruction touches 16 rows x 64 B = 16 half lines);
// PAT 1: 8 lanes per row (an instruction touches 8 rows x 128 B = 8 whole lines; the data would need a cross-lane move to become an operand)
// PAT 2: 32 lanes per row (2 rows x 512 B per instruction)
template <bool TWO, int PAT>
__global__ __launch_bounds__(256) void gather_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, int lo, int hi, int tiles,
                                                     float *out) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float4 cur[TWO ? 16 : 8], acc = {0, 0, 0, 0};
  auto row_of = [&](int t) {
    const uint64_t h = mix(wave * 1000003ull + (uint64_t)t * 16 + r);
    const int lv = lo + (int)(h % (uint64_t)(hi - lo + 1));
    const uint64_t first = (1ull << lv) - 1;
    return (int64_t)(first + (mix(h) & ((1ull << lv) - 1)));
  };
  auto load = [&](int t) {
    if (PAT == 0) {
      const int64_t row = row_of(t);
#pragma unroll
      for (int jc = 0; jc < 8; jc++) {
        cur[jc] = a[row * 32 + jc * 4 + g];
        if (TWO) cur[8 + jc] = b[row * 32 + jc * 4 + g];
      }
    } else if (PAT == 1) {
      // instruction i = (half h, line q): rows 8h .. 8h+7, line q of the row; lane l -> row 8h + (l >> 3), 16 B at (l & 7) * 16
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int64_t row = __shfl(row_of(t), 8 * h + (lane >> 3));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          cur[h * 4 + q] = a[row * 32 + q * 8 + (lane & 7)];
          if (TWO) cur[8 + h * 4 + q] = b[row * 32 + q * 8 + (lane & 7)];
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 8; h++) {
        const int64_t row = __shfl(row_of(t), 2 * h + (lane >> 5));
        cur[h] = a[row * 32 + (lane & 31)];
        if (TWO) cur[8 + h] = b[row * 32 + (lane & 31)];
      }
    }
  };
  load(0);
  for (int t = 0; t < tiles; t++) {
    float4 use[TWO ? 16 : 8];
#pragma unroll
    for (int i = 0; i < (TWO ? 16 : 8); i++) use[i] = cur[i];
    if (t + 1 < tiles) load(t + 1);
#pragma unroll
    for (int i = 0; i < (TWO ? 16 : 8); i++) { acc.x += use[i].x; acc.y += use[i].y; acc.z += use[i].z; acc.w += use[i].w; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv) {
  const int depth = argc > 1 ? atoi(argv[1]) : 24;
  const int64_t n_rows = (1ll << (depth + 1)) - 1;
  float4 *a, *b; float *out;
  if (hipMalloc(&a, n_rows * 512) != hipSuccess || hipMalloc(&b, n_rows * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, n_rows * 512); hipMemset(b, 0, n_rows * 512);
  hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { int lo, hi, two; const char *what; int pat = 0; };
  const Case cases[] = {
      {8, depth, 0, "all levels, 512 B per row (today's tile)"},
      {8, depth, 1, "all levels, 1 KB per row (P table for every level)"},
      {8, 17, 0, "levels 8..17, 512 B per row"},
      {8, 17, 1, "levels 8..17, 1 KB per row (P table for the cache-resident levels)"},
      {8, 12, 1, "levels 8..12, 1 KB per row (L2-resident: 8 MB)"},
      {18, depth, 0, "levels 18..depth, 512 B per row"},
      {8, depth, 0, "all levels, 512 B per row, 8 lanes per row (whole 128-B lines)", 1},
      {8, 17, 1, "levels 8..17, 1 KB per row, 8 lanes per row", 1},
      {8, depth, 0, "all levels, 512 B per row, 32 lanes per row (a row per half wave)", 2},
      {8, 17, 1, "levels 8..17, 1 KB per row, 32 lanes per row", 2},
  };
  for (const Case &c : cases) {
    const int tiles = 4000;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (c.pat == 0) { if (c.two) gather_kernel<true, 0><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); else gather_kernel<false, 0><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); }
      if (c.pat == 1) { if (c.two) gather_kernel<true, 1><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); else gather_kernel<false, 1><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); }
      if (c.pat == 2) { if (c.two) gather_kernel<true, 2><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); else gather_kernel<false, 2><<<256, 256>>>(a, b, c.lo, c.hi, tiles, out); }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 1024.0 * tiles * 16 * (c.two ? 1024.0 : 512.0);
    const double cyc = ms * 1e-3 * 2.4e9 / tiles;
    printf("%-70s %6.2f TB/s  %7.0f cycles per 16-row tile at 2.4 GHz  (%.1f ms)\n", c.what, bytes / ms / 1e9, cyc, ms);
  }
  return 0;
}
