// Which lane does each exchange read?  ds_swizzle 0x101F (lane ^ 4) against the two DPP forms tried for the pair-mode
// kernels' cross-half exchange (profiles/r04/EXPERIMENTS.md section 12).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int a = __builtin_amdgcn_ds_swizzle(lane, 0x101F);
  const int m = __builtin_amdgcn_mov_dpp(lane, 0x141, 0xf, 0xf, false);
  const int b = __builtin_amdgcn_mov_dpp(m, 0x1B, 0xf, 0xf, false);
  int r = __builtin_amdgcn_update_dpp(lane, lane, 0x104, 0xf, 0x5, false);
  r = __builtin_amdgcn_update_dpp(r, lane, 0x114, 0xf, 0xA, false);
  out[lane] = a; out[64 + lane] = b; out[128 + lane] = r; out[192 + lane] = m;
}
int main() {
  int* d; hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"ds_swizzle 0x101F", "half_mirror + quad reverse", "row_shl:4 | row_shr:4", "row_half_mirror alone"};
  for (int v = 0; v < 4; ++v) { printf("%-28s", names[v]); for (int l = 0; l < 16; ++l) printf(" %2d", h[64 * v + l]); printf("\n"); }
  return 0;
}
