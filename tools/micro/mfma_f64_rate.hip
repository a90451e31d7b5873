// Issue rate of v_mfma_f64_4x4x4_4b_f64 against v_fma_f64 on one wavefront per SIMD (MI355X): the costing behind
// "no MFMA for the 3x3 block updates" (VERDICT round 2, item 1b).  One instruction of either kind per loop slot, 8
// independent accumulators, 4096 iterations; cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_rate tools/micro/mfma_f64_rate.hip && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d1 __attribute__((ext_vector_type(1)));
__global__ void __launch_bounds__(64) rate(double* out, long long* cyc, int iters) {
  double a = out[threadIdx.x], b = out[64 + threadIdx.x];
  double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double d0 = 0, d1_ = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
  for (int i = 0; i < iters; ++i) {
#define FMA(d) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
    FMA(d0); FMA(d1_); FMA(d2); FMA(d3); FMA(d4); FMA(d5); FMA(d6); FMA(d7);
  }
  long long t2 = __builtin_readcyclecounter();
  // the same with 32 independent chains (is the 8-chain figure a latency?) and with 1 (the dependent-issue latency)
  double e[32];
  for (int k = 0; k < 32; ++k) e[k] = k;
  for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) FMA(e[k]);
  }
  long long t3 = __builtin_readcyclecounter();
  double g0 = 1.0;
  for (int i = 0; i < iters; ++i) { FMA(g0); FMA(g0); FMA(g0); FMA(g0); FMA(g0); FMA(g0); FMA(g0); FMA(g0); }
  long long t4 = __builtin_readcyclecounter();
  for (int k = 0; k < 32; ++k) d0 += e[k];
  d0 += g0;
  out[128 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + d0 + d1_ + d2 + d3 + d4 + d5 + d6 + d7;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 192 * 8); hipMalloc(&cyc, 32); hipMemset(out, 0, 192 * 8);
  const int iters = 4096;
  rate<<<1, 64>>>(out, cyc, iters); rate<<<1, 64>>>(out, cyc, iters);
  long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  const double n = 8.0 * iters;
  std::printf("v_mfma_f64_4x4x4_4b: %.2f clock ticks / instruction (512 flop each: 4 blocks x 4x4x4 x 2)\n", h[0] / n);
  std::printf("v_fma_f64          : %.2f clock ticks / instruction (128 flop each: 64 lanes x 2)\n", h[1] / n);
  std::printf("v_fma_f64, 32 independent chains: %.2f ticks / instruction; one dependent chain: %.2f ticks / instruction\n", h[2] / n, h[3] / n);
  std::printf("flop per tick: mfma %.1f, fma %.1f (s_memtime ticks; ratio is what matters)\n", 512.0 * n / h[0], 128.0 * n / h[1]);
  return 0;
}
