// One wavefront per SIMD (the solve kernels' occupancy): what does a DEPENDENT chain of fp64 instructions cost per link
// against independent ones?  Chains of v_fma_f64, of v_mov_b32_dpp + v_fma_f64 (the quad kernel's broadcast-then-use), of
// ds_read_b64 -> v_fma_f64 (LDS state), of v_accvgpr_read x2 -> v_fma_f64 (parked values), of v_rcp_f64 -> v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 -o fma_f64_chain fma_f64_chain.hip && ./fma_f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256
template <int MODE>
__global__ void __launch_bounds__(64, 2) chain(double* out, long long* cycles, double seed) {
  __shared__ double lds[64 * 8];
  const int lane = threadIdx.x;
  double a = seed + lane, b = 1.0000001, c = 1e-9;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  lds[lane] = a; lds[64 + lane] = b;
  __builtin_amdgcn_s_waitcnt(0);
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int k = 0; k < REP / 16; ++k) {
      if (MODE == 0) {  // dependent fma chain
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));
      } else if (MODE == 1) {  // four independent chains
        asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
      } else if (MODE == 2) {  // dpp broadcast of the result (2 movs), then a dependent fma
        int lo = __double2loint(x0), hi = __double2hiint(x0), l2, h2;
        asm volatile("s_nop 1\n v_mov_b32_dpp %0, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
                     : "=&v"(l2), "=&v"(h2) : "v"(lo), "v"(hi));
        double y = __hiloint2double(h2, l2);
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x0) : "v"(y), "v"(b), "v"(c));
      } else if (MODE == 3) {  // LDS write + read in the chain
        lds[128 + lane] = x0;
        double y = lds[128 + (lane ^ 1)];
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x0) : "v"(y), "v"(b), "v"(c));
      } else if (MODE == 4) {  // rcp in the chain
        double r;
        asm volatile("v_rcp_f64 %0, %1\n s_nop 0" : "=v"(r) : "v"(x0));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x0) : "v"(r), "v"(b), "v"(a));
      } else if (MODE == 5) {  // s_nop between: does an idle slot cost 4 cycles?
        asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(x0) : "v"(b), "v"(c));
      } else if (MODE == 6) {  // scalar ALU op between
        asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s90, 0" : "+v"(x0) : "v"(b), "v"(c) : "s90");
      } else if (MODE == 7) {  // two accvgpr reads then fma
        int lo = __double2loint(x0), hi = __double2hiint(x0), al, ah, l2, h2;
        asm volatile("v_accvgpr_write_b32 %0, %2\n v_accvgpr_write_b32 %1, %3" : "=a"(al), "=a"(ah) : "v"(lo), "v"(hi));
        asm volatile("v_accvgpr_read_b32 %0, %2\n v_accvgpr_read_b32 %1, %3" : "=v"(l2), "=v"(h2) : "a"(al), "a"(ah));
        double y = __hiloint2double(h2, l2);
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x0) : "v"(y), "v"(b), "v"(c));
      } else if (MODE == 8) {  // dependent v_add_f64 chain
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(c));
      } else if (MODE == 10) {  // eight independent chains
        double y0 = x0 + 10, y1 = x1 + 10, y2 = x2 + 10, y3 = x3 + 10;
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(b), "v"(c));
        x0 += 0; x1 = x1 + (y0 + y1 + y2 + y3) * 0.0;
      } else if (MODE == 11) {  // four independent chains, constants as inline operands (one VGPR pair read per instruction)
        asm volatile("v_fma_f64 %0, %0, 1.0, 0.5\n v_fma_f64 %1, %1, 1.0, 0.5\n v_fma_f64 %2, %2, 1.0, 0.5\n v_fma_f64 %3, %3, 1.0, 0.5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      } else if (MODE == 12) {  // four independent 32-bit movs
        int v0 = __double2loint(x0), v1 = __double2loint(x1), v2 = __double2loint(x2), v3 = __double2loint(x3);
        asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        x0 = __hiloint2double(__double2hiint(x0), v0); x1 = __hiloint2double(__double2hiint(x1), v1);
        x2 = __hiloint2double(__double2hiint(x2), v2); x3 = __hiloint2double(__double2hiint(x3), v3);
      } else if (MODE == 13) {  // four independent v_add_f64
        asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c));
      } else if (MODE == 14) {  // four independent v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
      } else if (MODE == 15) {  // four independent v_fma_f32
        float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3, fb = 1.0001f, fc = 1e-6f;
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));
        x0 = f0; x1 = f1; x2 = f2; x3 = f3;
      } else if (MODE == 9) {  // dependent 32-bit mov chain
        int v = __double2loint(x0);
        asm volatile("v_mov_b32 %0, %0" : "+v"(v));
        x0 = __hiloint2double(__double2hiint(x0), v);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* label, double* out, long long* cyc, int blocks, double links_per_rep) {
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
  hipDeviceSynchronize();
  long long h[2048];
  (void)hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  long long lo = h[0], hi = h[0];
  for (int i = 1; i < blocks; ++i) { lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; }
  std::printf("%-58s %4d wavefronts: %6.2f .. %6.2f cycles per repetition\n", label, blocks, lo / (double)REP, hi / (double)REP);
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 64 * 2048); (void)hipMalloc(&cyc, 8 * 2048);
  for (int blocks : {1, 1024, 2048}) {
    run<0>("dependent v_fma_f64", out, cyc, blocks, 1);
    run<1>("4 independent v_fma_f64", out, cyc, blocks, 4);
    run<8>("dependent v_add_f64", out, cyc, blocks, 1);
    run<9>("dependent v_mov_b32", out, cyc, blocks, 1);
    run<2>("s_nop 1 + 2 x v_mov_b32_dpp + dependent v_fma_f64", out, cyc, blocks, 1);
    run<3>("ds_write_b64 + ds_read_b64 + dependent v_fma_f64", out, cyc, blocks, 1);
    run<4>("v_rcp_f64 + s_nop 0 + dependent v_fma_f64", out, cyc, blocks, 1);
    run<5>("v_fma_f64 + s_nop 0 (dependent)", out, cyc, blocks, 1);
    run<6>("v_fma_f64 + s_mov_b32 (dependent)", out, cyc, blocks, 1);
    run<7>("2 x accvgpr_write + 2 x accvgpr_read + dependent v_fma_f64", out, cyc, blocks, 1);
    run<10>("8 independent v_fma_f64", out, cyc, blocks, 8);
    run<11>("4 independent v_fma_f64, inline constants", out, cyc, blocks, 4);
    run<12>("4 independent v_mov_b32", out, cyc, blocks, 4);
    run<13>("4 independent v_add_f64", out, cyc, blocks, 4);
    run<14>("4 independent v_mul_f64", out, cyc, blocks, 4);
    run<15>("4 independent v_fma_f32 (+ conversions)", out, cyc, blocks, 4);
  }
  return 0;
}
