// okx_api.hip — the C-ABI of include/okx.h on top of the gfx950 kernels.
// Thin by design: argument validation, program upload, launch geometry, error strings.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "okx_kernels.hip"
#include "okx_packed.hip"

struct okx_program {
  okx::DevProgram host;        // host copy (dimensions, launch sizing)
  okx::DevProgram* dev;        // device copy
  int device;
  int n_cu;
  size_t lds_bytes;        // eval / rebind / single-problem solve kernels
  size_t solve_lds_bytes;  // selected solve kernel
  int blocks_per_cu;
  int nreg;                // padded row length of the register-resident factorisation
  const void* solve_fn;    // okx_solve_kernel<NREG> (one problem per wavefront)
  int groups;              // problems per wavefront of the packed kernel (1 = not available)
  int group_width;         // lanes per problem in the packed kernel
  const void* packed_fn;   // okx_solve_packed_kernel<NREG, G> or null
  size_t packed_lds_bytes;
  int packed_blocks_per_cu;
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(OKX_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));         \
  } while (0)

typedef void (*solve_kernel_t)(const okx::DevProgram*, okx::SolveArgs);
typedef void (*packed_kernel_t)(const okx::DevProgram*, okx::SolveArgs, int);

// Every program gets the one-problem-per-wavefront kernel (register LDL^T, template on the
// padded row length) and, when a problem fits 32 lanes, the lane-group packed kernel too.
void select_solve_kernels(okx_program* p) {
  const int n = p->host.n, m = p->host.m;
  solve_kernel_t fn;
  if (n <= 15) {
    fn = okx::okx_solve_kernel<15, false>;
    p->nreg = 15;
  } else if (n <= 18) {
    fn = okx::okx_solve_kernel<18, false>;
    p->nreg = 18;
  } else if (n <= 21) {
    fn = okx::okx_solve_kernel<21, false>;
    p->nreg = 21;
  } else if (n <= 24) {
    fn = okx::okx_solve_kernel<24, false>;
    p->nreg = 24;
  } else if (n <= 36) {
    fn = okx::okx_solve_kernel<36, false>;
    p->nreg = 36;
  } else if (n <= 48) {
    fn = okx::okx_solve_kernel<48, false>;
    p->nreg = 48;
  } else {
    fn = okx::okx_solve_kernel<63, false>;
    p->nreg = 63;
  }
  p->solve_fn = (const void*)fn;

  p->groups = 1;
  p->group_width = 64;
  p->packed_fn = nullptr;
  if (n > 24) return;
  const int width = m > p->nreg + 1 ? m : p->nreg + 1;  // the rhs row lives at local lane nreg
  int groups = 64 / width;
  if (groups > 4) groups = 4;
  if (groups < 2) return;
  packed_kernel_t pk;
  if (p->nreg == 15) {
    pk = groups == 4 ? okx::okx_solve_packed_kernel<15, 4, false>
       : groups == 3 ? okx::okx_solve_packed_kernel<15, 3, false> : okx::okx_solve_packed_kernel<15, 2, false>;
  } else if (p->nreg == 18) {
    if (groups > 3) groups = 3;
    pk = groups == 3 ? okx::okx_solve_packed_kernel<18, 3, false> : okx::okx_solve_packed_kernel<18, 2, false>;
  } else if (p->nreg == 21) {
    groups = 2;
    pk = okx::okx_solve_packed_kernel<21, 2, false>;
  } else {
    groups = 2;
    pk = okx::okx_solve_packed_kernel<24, 2, false>;
  }
  p->groups = groups;
  p->group_width = width;
  p->packed_fn = (const void*)pk;
}

// Resident single-wave workgroups per CU.  The occupancy API assumes 64 KiB of LDS per CU on
// this stack, so the limit is derived here: 512 VGPRs per SIMD lane (8-register granules),
// 160 KiB LDS per CU, 8 waves per SIMD.
int resident_blocks_per_cu(const void* fn, size_t lds_bytes) {
  int occ = 32;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {
    const int alloc = (fa.numRegs + 7) / 8 * 8;
    int per_simd = 512 / alloc;
    if (per_simd > 8) per_simd = 8;
    if (per_simd < 1) per_simd = 1;
    occ = 4 * per_simd;
  }
  const int by_lds = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
  if (by_lds < occ) occ = by_lds;
  if (occ < 1) occ = 1;
  if (const char* cap = getenv("OKX_BLOCKS_PER_CU")) {  // tuning knob
    const int c = atoi(cap);
    if (c >= 1 && c < occ) occ = c;
  }
  return occ;
}

int grid_for(const okx_program* p, long long units) {
  long long cap = (long long)p->n_cu * p->blocks_per_cu;
  if (cap < 1) cap = 1;
  return (int)(units < cap ? (units < 1 ? 1 : units) : cap);
}

}  // namespace

extern "C" {

int32_t okx_abi_version(void) { return OKX_ABI_VERSION; }

const char* okx_last_error(void) { return g_err; }

void okx_default_opts(okx_solve_opts* o) {
  if (!o) return;
  o->max_iter = 100;
  o->chain = 0;
  o->steps_per_geometry = 0;
  o->chain_len = 0;
  o->step_tol = 1e-11;
  o->grad_tol = 0.0;
  o->ftol = 1e-10;
  o->lambda0 = 1e-6;
  o->residual_tolerance = 1e-3;
  o->kernel = 0;
  o->reserved = 0;
}

int32_t okx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t okx_program_create(const okx_program_desc* desc, okx_program** out) {
  if (!out) return fail(OKX_ERR_INVALID, "out is null");
  *out = nullptr;
  okx_program* p = new (std::nothrow) okx_program;
  if (!p) return fail(OKX_ERR_ALLOC, "out of host memory");
  std::memset(p, 0, sizeof(*p));
  int rc = okx::build_dev_program(desc, &p->host, g_err, (int)sizeof(g_err));
  if (rc != OKX_OK) {
    delete p;
    return rc;
  }
  p->host.lds_doubles = okx::lds_doubles(p->host);
  select_solve_kernels(p);
  p->lds_bytes = sizeof(double) * (size_t)p->host.lds_doubles;
  p->solve_lds_bytes = p->lds_bytes;
  p->packed_lds_bytes = p->packed_fn ? sizeof(double) * (size_t)okx::packed_lds_doubles(p->host, p->groups) : 0;
  if (p->packed_lds_bytes > 160 * 1024) p->packed_fn = nullptr;
  if (p->lds_bytes > 160 * 1024) {
    delete p;
    return fail(OKX_ERR_LIMIT, "problem needs %zu bytes of LDS (max 163840)", p->lds_bytes);
  }
  hipError_t e = hipGetDevice(&p->device);
  if (e != hipSuccess) {
    delete p;
    return fail(OKX_ERR_DEVICE, "hipGetDevice failed: %s (no GPU?)", hipGetErrorString(e));
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, p->device);
  if (e != hipSuccess) {
    delete p;
    return fail(OKX_ERR_DEVICE, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
  }
  p->n_cu = prop.multiProcessorCount;
  e = hipMalloc((void**)&p->dev, sizeof(okx::DevProgram));
  if (e != hipSuccess) {
    delete p;
    return fail(OKX_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  e = hipMemcpy(p->dev, &p->host, sizeof(okx::DevProgram), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p->dev);
    delete p;
    return fail(OKX_ERR_DEVICE, "hipMemcpy failed: %s", hipGetErrorString(e));
  }
  // >64 KiB of dynamic LDS needs the opt-in attribute
  (void)hipFuncSetAttribute(p->solve_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->solve_lds_bytes);
  (void)hipFuncSetAttribute((const void*)okx::okx_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)p->lds_bytes);
  (void)hipFuncSetAttribute((const void*)okx::okx_rebind_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)p->lds_bytes);
  p->blocks_per_cu = resident_blocks_per_cu(p->solve_fn, p->solve_lds_bytes);
  p->packed_blocks_per_cu = 0;
  if (p->packed_fn) {
    (void)hipFuncSetAttribute(p->packed_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->packed_lds_bytes);
    p->packed_blocks_per_cu = resident_blocks_per_cu(p->packed_fn, p->packed_lds_bytes);
  }
  *out = p;
  return OKX_OK;
}

void okx_program_destroy(okx_program* p) {
  if (!p) return;
  if (p->dev) (void)hipFree(p->dev);
  delete p;
}

int32_t okx_solve_batch(okx_program* p, const okx_solve_opts* opts, int64_t n_problems,
                        const double* d_targets, const double* d_geom_pos,
                        const double* d_geom_row_param, double* d_out_pos, okx_info* d_info,
                        void* stream) {
  if (!p || !opts) return fail(OKX_ERR_INVALID, "null program or options");
  if (n_problems < 0) return fail(OKX_ERR_INVALID, "negative problem count");
  if (n_problems == 0) return OKX_OK;
  if (!d_out_pos || !d_info) return fail(OKX_ERR_INVALID, "null output pointer");
  if (p->host.n_targets > 0 && !d_targets) return fail(OKX_ERR_INVALID, "null targets");
  if ((d_geom_pos == nullptr) != (d_geom_row_param == nullptr))
    return fail(OKX_ERR_INVALID, "geometry positions and row parameters must be given together");
  const long long spg = opts->steps_per_geometry;
  if (spg < 0 || (spg > 0 && n_problems % spg != 0))
    return fail(OKX_ERR_INVALID, "n_problems must be a multiple of steps_per_geometry");
  if (d_geom_pos && spg == 0 )
    return fail(OKX_ERR_INVALID, "a geometry table needs steps_per_geometry > 0");
  if (opts->max_iter < 1) return fail(OKX_ERR_INVALID, "max_iter must be >= 1");
  okx::SolveArgs a;
  a.targets = d_targets;
  a.geom_pos = d_geom_pos;
  a.geom_row_param = d_geom_row_param;
  a.out_pos = d_out_pos;
  a.info = d_info;
  a.n_problems = n_problems;
  a.steps_per_geometry = spg;
  a.max_iter = opts->max_iter;
  a.pad_ = 0;
  // Kernel choice (profiles/r01/config_sweep_v3.txt).  The packed kernel keeps more problems in
  // flight per CU (G lane groups x resident waves): measured 1.5x on saturating batches of
  // n <= 15 systems (MacPherson grid), no gain for n = 18 (DW corner), so auto = packed only
  // for n <= 15 and batches of at least 8 problems per resident slot.
  const long long single_slots = (long long)p->n_cu * p->blocks_per_cu;
  const long long packed_slots = (long long)p->n_cu * p->packed_blocks_per_cu * p->groups;
  bool use_packed = false;
  if (p->packed_fn) {
    if (opts->kernel == 2) use_packed = true;
    else if (opts->kernel == 0) use_packed = p->nreg <= 15 && n_problems >= 8 * single_slots;
    if (const char* env = getenv("OKX_PACKED")) use_packed = env[0] == '1';
  }
  {
    const long long span = spg > 0 ? spg : n_problems;
    long long len = opts->chain_len;
    if (len == 0) len = opts->chain ? span : 1;
    if (len < 0) {  // auto: about one chain per resident problem slot, balanced inside a geometry
      const long long slots = use_packed ? packed_slots : single_slots;
      const long long ideal = (n_problems + slots - 1) / slots;
      if (ideal >= span) {
        len = span;
      } else {
        const long long per_span = (span + ideal - 1) / ideal;
        len = (span + per_span - 1) / per_span;
      }
    }
    if (len < 1) len = 1;
    if (len > span) len = span;
    a.chain_len = len;
  }
  a.step_tol = opts->step_tol;
  a.grad_tol = opts->grad_tol;
  a.ftol = opts->ftol;
  a.lambda0 = opts->lambda0;
  a.residual_tolerance = opts->residual_tolerance;
  a.phase_cycles = nullptr;
  const long long span_ = spg > 0 ? spg : n_problems;
  const long long units = (n_problems / span_) * ((span_ + a.chain_len - 1) / a.chain_len);
  const okx::DevProgram* dev = p->dev;
  if (use_packed) {
    int width = p->group_width;
    long long cap = (long long)p->n_cu * p->packed_blocks_per_cu;
    const long long wave_units = (units + p->groups - 1) / p->groups;
    const int grid = (int)(wave_units < cap ? (wave_units < 1 ? 1 : wave_units) : cap);
    void* kargs[] = {(void*)&dev, (void*)&a, (void*)&width};
    HIP_TRY(hipLaunchKernel(p->packed_fn, dim3(grid), dim3(okx::kWave), kargs, p->packed_lds_bytes,
                            (hipStream_t)stream));
    return OKX_OK;
  }
  const int grid = grid_for(p, units);
  void* kargs[] = {(void*)&dev, (void*)&a};
  HIP_TRY(hipLaunchKernel(p->solve_fn, dim3(grid), dim3(okx::kWave), kargs, p->lds_bytes,
                          (hipStream_t)stream));
  return OKX_OK;
}

int32_t okx_eval_batch(okx_program* p, int64_t n_problems, const double* d_x,
                       const double* d_targets, double* d_r, double* d_jac, void* stream) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  if (n_problems <= 0) return n_problems == 0 ? OKX_OK : fail(OKX_ERR_INVALID, "negative count");
  if (!d_x || !d_r) return fail(OKX_ERR_INVALID, "null pointer");
  okx::EvalArgs a;
  a.x = d_x;
  a.targets = d_targets;
  a.r = d_r;
  a.jac = d_jac;
  a.ata = nullptr;
  a.atr = nullptr;
  a.n_problems = n_problems;
  hipLaunchKernelGGL(okx::okx_eval_kernel, dim3(grid_for(p, n_problems)), dim3(okx::kWave),
                     p->lds_bytes, (hipStream_t)stream, (const okx::DevProgram*)p->dev, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

/* Test hook (not part of the reference boundary): J^T J and J^T r as the solver forms them. */
int32_t okx_debug_normal_equations(okx_program* p, int64_t n_problems, const double* d_x,
                                   const double* d_targets, double* d_r, double* d_ata,
                                   double* d_atr, void* stream) {
  if (!p || !d_x || !d_r) return fail(OKX_ERR_INVALID, "null pointer");
  if (n_problems <= 0) return OKX_OK;
  okx::EvalArgs a;
  a.x = d_x;
  a.targets = d_targets;
  a.r = d_r;
  a.jac = nullptr;
  a.ata = d_ata;
  a.atr = d_atr;
  a.n_problems = n_problems;
  hipLaunchKernelGGL(okx::okx_eval_kernel, dim3(grid_for(p, n_problems)), dim3(okx::kWave),
                     p->lds_bytes, (hipStream_t)stream, (const okx::DevProgram*)p->dev, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

int32_t okx_rebind_design(okx_program* p, int64_t n_geometries, const double* d_hardpoints,
                          double* d_geom_pos, double* d_geom_row_param, void* stream) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  if (n_geometries <= 0) return n_geometries == 0 ? OKX_OK : fail(OKX_ERR_INVALID, "negative count");
  if (!d_hardpoints || !d_geom_pos || !d_geom_row_param) return fail(OKX_ERR_INVALID, "null pointer");
  okx::RebindArgs a;
  a.hardpoints = d_hardpoints;
  a.geom_pos = d_geom_pos;
  a.geom_row_param = d_geom_row_param;
  a.n_geometries = n_geometries;
  hipLaunchKernelGGL(okx::okx_rebind_kernel, dim3(grid_for(p, n_geometries)), dim3(okx::kWave),
                     p->lds_bytes, (hipStream_t)stream, (const okx::DevProgram*)p->dev, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

/* Diagnostic (not part of the reference boundary): same as okx_solve_batch for an n = 18
   program, but runs the stamped kernel instantiation and returns per-phase cycle sums of
   workgroup 0 in d_phase_cycles[12]: 0 staging, 1 problem setup, 2 x->pos + derived points,
   3 rows, 4 reductions + LM logic, 5 normal equations, 6 factorisation, 7 substitutions,
   8 output. */
int32_t okx_debug_phase_profile(okx_program* p, const okx_solve_opts* opts, int64_t n_problems,
                                const double* d_targets, double* d_out_pos, okx_info* d_info,
                                unsigned long long* d_phase_cycles, void* stream) {
  if (!p || !opts || p->nreg != 18) return fail(OKX_ERR_INVALID, "phase profile needs an n = 18 program");
  okx::SolveArgs a;
  a.targets = d_targets;
  a.geom_pos = nullptr;
  a.geom_row_param = nullptr;
  a.out_pos = d_out_pos;
  a.info = d_info;
  a.n_problems = n_problems;
  a.steps_per_geometry = 0;
  a.max_iter = opts->max_iter;
  a.pad_ = 0;
  a.chain_len = 1;
  a.step_tol = opts->step_tol;
  a.grad_tol = opts->grad_tol;
  a.ftol = opts->ftol;
  a.lambda0 = opts->lambda0;
  a.residual_tolerance = opts->residual_tolerance;
  a.phase_cycles = d_phase_cycles;
  const okx::DevProgram* dev = p->dev;
  if (p->groups == 3 && opts->kernel == 2) {
    int width = p->group_width;
    void* kargs[] = {(void*)&dev, (void*)&a, (void*)&width};
    packed_kernel_t fn = okx::okx_solve_packed_kernel<18, 3, true>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)p->packed_lds_bytes);
    HIP_TRY(hipLaunchKernel((const void*)fn, dim3(grid_for(p, (n_problems + 2) / 3)), dim3(okx::kWave),
                            kargs, p->packed_lds_bytes, (hipStream_t)stream));
    return OKX_OK;
  }
  void* kargs[] = {(void*)&dev, (void*)&a};
  solve_kernel_t fn = okx::okx_solve_kernel<18, true>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes);
  HIP_TRY(hipLaunchKernel((const void*)fn, dim3(grid_for(p, n_problems)), dim3(okx::kWave), kargs,
                          p->lds_bytes, (hipStream_t)stream));
  return OKX_OK;
}

/* Plan introspection for CPU-side tests: fills counts without touching a device. */
int32_t okx_plan_stats(const okx_program_desc* desc, int32_t* out8) {
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  if (rc == OKX_OK && out8) {
    out8[0] = tmp->n;
    out8[1] = tmp->m;
    out8[2] = tmp->n_pairs;
    out8[3] = tmp->pair_start[tmp->n_pairs];
    out8[4] = tmp->n_active;
    out8[5] = tmp->js_stride;
    out8[6] = tmp->lda;
    out8[7] = okx::lds_doubles(*tmp) * 8;
  }
  delete tmp;
  return rc;
}

}  // extern "C"
