// Does a wavefront whose EXEC mask has only its first 16 (32, 48) lanes set issue fp64 instructions faster?  (A wave64
// VALU instruction runs as four 16-lane passes: if passes without an active lane were skipped, a lane kernel with 16
// problems per wavefront would run four times as fast as one with 64 - and the headline sweep's 16384 problems could use
// the one-lane-per-problem layout on all 1024 SIMDs.)  Four independent v_fma_f64 chains, one wavefront per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o exec_quarter exec_quarter.hip && ./exec_quarter
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256
__global__ void __launch_bounds__(64, 2) chain(double* out, long long* cycles, double seed, int active) {
  const int lane = threadIdx.x;
  double b = 1.0000001, c = 1e-9;
  double x0 = seed + lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = 0, t1 = 0;
  if (lane < active) {  // (EXEC = the first `active` lanes for the whole timed loop)
    __builtin_amdgcn_s_waitcnt(0);
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
#pragma unroll
      for (int k = 0; k < REP / 16; ++k)
        asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
    }
    t1 = __builtin_readcyclecounter();
  }
  out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 64 * 2048); (void)hipMalloc(&cyc, 8 * 2048);
  for (int blocks : {1, 1024})
    for (int active : {64, 48, 32, 16, 8, 1}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0, active);
        hipDeviceSynchronize();
      }
      long long h[2048];
      (void)hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      long long lo = h[0], hi = h[0];
      for (int i = 1; i < blocks; ++i) { lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; }
      std::printf("%2d active lanes, %4d wavefronts: %6.2f .. %6.2f cycles per 4 independent v_fma_f64\n", active, blocks, lo / (double)REP, hi / (double)REP);
    }
  return 0;
}
