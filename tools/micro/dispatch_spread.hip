// When do the wavefronts of a 1024-wavefront launch start?  One wavefront per workgroup (the solve kernels' shape) against
// four per workgroup, with the solve kernels' resources (12 KB of LDS per wavefront, the whole register file): every
// wavefront stamps the device's 100 MHz real-time counter on entry, spins ~10 us, stamps again.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_spread dispatch_spread.hip && ./dispatch_spread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) spin(unsigned long long* stamps, int spin_cycles) {
  __shared__ double lds[1536 * WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::: "v255", "a255");  // the whole register file, as the solve kernels have it
  lds[wave * 1536 + lane] = (double)lane;
  const long long c0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - c0 < spin_cycles) __builtin_amdgcn_s_sleep(1);
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const int w = blockIdx.x * WAVES + wave;
    stamps[2 * w] = t0;
    stamps[2 * w + 1] = t1 + (lds[wave * 1536 + 7] > 1e9 ? 1 : 0);
  }
}

template <int WAVES>
void run(const char* label, unsigned long long* d, int n_waves) {
  std::vector<unsigned long long> h(2 * n_waves);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(spin<WAVES>, dim3(n_waves / WAVES), dim3(64 * WAVES), 0, 0, d, 24000);
    hipEventRecord(e0, 0);
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(spin<WAVES>, dim3(n_waves / WAVES), dim3(64 * WAVES), 0, 0, d, 24000);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * n_waves, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0, end_hi = 0;
    std::vector<unsigned long long> starts;
    for (int w = 0; w < n_waves; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w]); end_hi = std::max(end_hi, h[2 * w + 1]); starts.push_back(h[2 * w]); }
    std::sort(starts.begin(), starts.end());
    std::printf("%s: %d wavefronts, %.2f us per launch; starts spread over %.2f us (median %.2f), last end %.2f us after the first start\n", label, n_waves,
                1e3 * ms / 200, 0.01 * (hi - lo), 0.01 * (starts[n_waves / 2] - lo), 0.01 * (end_hi - lo));
  }
}

int main() {
  unsigned long long* d = nullptr;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * 4096);
  run<1>("1 wavefront per workgroup ", d, 1024);
  run<4>("4 wavefronts per workgroup", d, 1024);
  run<2>("2 wavefronts per workgroup", d, 1024);
  run<1>("1 wavefront per workgroup, 16 wavefronts", d, 16);
  return 0;
}
