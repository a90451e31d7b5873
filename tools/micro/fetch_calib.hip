// fetch_calib.hip - what rocprofv3's FETCH_SIZE reports for the READ SHAPES of this repository's streaming kernels on
// gfx950, against a known byte count (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on
// a known byte count in your own access pattern").  Buffer: N records of 360 bytes (a double-wishbone state record),
// 1.07 GB - four times the Infinity Cache - read ONCE per kernel:
//   rec8    one lane per record, 45 loads of 8 B at a 360-byte lane stride (the lane kernels' GIVEN-state reads:
//           okx_lane_evaluate_*, okx_corner_metrics_kernel)
//   tile16  one wavefront per tile of 64 records = 23 040 contiguous bytes, 16 B per lane (okx_corner_metrics_tiled,
//           okx_quad_expand's input rows, the cold bodies' staged tables)
//   tile8   the same tiles, 8 B per lane
//   stream16 / stream8   a plain grid-stride stream over the whole buffer, 16 B / 8 B per lane (the guide's reference shape)
// Every lane folds what it read into one double and stores it (8 B per lane: the stores are in WRITE_SIZE, not FETCH_SIZE).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o f -- ./fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr int kRecord = 45;  // doubles per record (360 B)

__global__ void __launch_bounds__(64) calib_rec8(const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
  if (r >= n) return;
  const double* p = in + r * kRecord;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < kRecord; ++k) s += p[k];
  out[r] = s;
}

__global__ void __launch_bounds__(64) calib_tile16(const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long tile = blockIdx.x;
  const double2* p = reinterpret_cast<const double2*>(in + tile * 64 * kRecord);
  const int lane = threadIdx.x;
  const long long left = n - tile * 64;
  const int n2 = (int)(left < 64 ? left : 64) * kRecord / 2;
  double s = 0.0;
  for (int i = lane; i < n2; i += 64) { const double2 v = p[i]; s += v.x + v.y; }
  out[tile * 64 + lane] = s;
}

__global__ void __launch_bounds__(64) calib_tile8(const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long tile = blockIdx.x;
  const double* p = in + tile * 64 * kRecord;
  const int lane = threadIdx.x;
  const long long left = n - tile * 64;
  const int n1 = (int)(left < 64 ? left : 64) * kRecord;
  double s = 0.0;
  for (int i = lane; i < n1; i += 64) s += p[i];
  out[tile * 64 + lane] = s;
}

__global__ void __launch_bounds__(256) calib_stream16(const double2* __restrict__ in, double* __restrict__ out, long long n2) {
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) { const double2 v = in[i]; s += v.x + v.y; }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) calib_stream8(const double* __restrict__ in, double* __restrict__ out, long long n1) {
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1; i += (long long)gridDim.x * 256) s += in[i];
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const long long n = 2981888;  // records: a multiple of 64; 1 073 479 680 bytes
  const long long doubles = n * kRecord;
  double *in = nullptr, *out = nullptr;
  CHECK(hipMalloc(&in, doubles * sizeof(double)));
  CHECK(hipMalloc(&out, n * sizeof(double)));
  CHECK(hipMemset(in, 0, doubles * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int tiles = (int)(n / 64);
  const int stream_blocks = 256 * 8;
  std::printf("{\"bytes_read_per_launch\": %lld, \"records\": %lld, \"kernels\": {", doubles * 8, n);
  for (int which = 0; which < 5; ++which) {
    const char* name = which == 0 ? "calib_rec8" : which == 1 ? "calib_tile16" : which == 2 ? "calib_tile8" : which == 3 ? "calib_stream16" : "calib_stream8";
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {  // three dispatches each: per-dispatch counter rows
      CHECK(hipEventRecord(e0));
      if (which == 0) calib_rec8<<<tiles, 64>>>(in, out, n);
      if (which == 1) calib_tile16<<<tiles, 64>>>(in, out, n);
      if (which == 2) calib_tile8<<<tiles, 64>>>(in, out, n);
      if (which == 3) calib_stream16<<<stream_blocks, 256>>>(reinterpret_cast<const double2*>(in), out, doubles / 2);
      if (which == 4) calib_stream8<<<stream_blocks, 256>>>(in, out, doubles);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    std::printf("%s\"%s\": {\"ms\": %.4f, \"tb_per_s\": %.3f}", which ? ", " : "", name, best, doubles * 8 / (best * 1e-3) / 1e12);
  }
  std::printf("}}\n");
  CHECK(hipFree(in));
  CHECK(hipFree(out));
  return 0;
}
