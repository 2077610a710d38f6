// Lane-exchange building blocks of the register sorts, checked lane by lane on the device: v_permlane32_swap / v_permlane16_swap
// (gfx950) as lane ^ 32 / lane ^ 16, DPP row_ror:8 as lane ^ 8, row_ror:4 / row_ror:12 as the two halves of lane ^ 4.
//   hipcc --offload-arch=gfx950 -O2 tools/lane_xor_probe.hip -o /tmp/lx && /tmp/lx      (prints "bad mask 16": row_ror:n moves lane i to i + n)
#include <hip/hip_runtime.h>
__global__ void k(unsigned *out) {
  unsigned x = threadIdx.x * 7u + 3u;
  // xor 32 via permlane32_swap
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  unsigned a = r[0], b = r[1];
  unsigned y32 = (threadIdx.x & 32) ? a : b;
  auto r2 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  unsigned y16 = (threadIdx.x & 16) ? r2[0] : r2[1];
  unsigned y8 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);   // row_ror:8
  unsigned y4a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xF, 0xF, true);  // row_ror:4
  unsigned y4b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x12C, 0xF, 0xF, true);  // row_ror:12
  out[threadIdx.x * 5 + 0] = y32; out[threadIdx.x * 5 + 1] = y16; out[threadIdx.x * 5 + 2] = y8; out[threadIdx.x * 5 + 3] = y4a; out[threadIdx.x * 5 + 4] = y4b;
}
int main() {
  unsigned *d; hipMalloc(&d, 64 * 5 * 4);
  k<<<1, 64>>>(d);
  unsigned h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    auto v = [](int t) { return (unsigned)t * 7u + 3u; };
    if (h[l * 5 + 0] != v(l ^ 32)) bad |= 1;
    if (h[l * 5 + 1] != v(l ^ 16)) bad |= 2;
    if (h[l * 5 + 2] != v(l ^ 8)) bad |= 4;
    unsigned y4 = (l & 4) ? h[l * 5 + 4] : h[l * 5 + 3];
    // row_ror:n: lane i gets lane (i + n) mod 16?  or (i - n)?  check both
    if (y4 != v(l ^ 4)) { unsigned z = (l & 4) ? h[l * 5 + 3] : h[l * 5 + 4]; if (z != v(l ^ 4)) bad |= 8; else bad |= 16; }
  }
  printf("bad mask %d (0 = all as assumed; 16 = ror direction is the other way)\n", bad);
  return 0;
}
