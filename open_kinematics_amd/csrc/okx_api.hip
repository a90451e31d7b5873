// okx_api.hip — the C-ABI of include/okx.h on top of the gfx950 kernels.
// Thin by design: argument validation, program upload, launch geometry, error strings.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <cmath>
#include <functional>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "okx_kernels.hip"
#include "okx_packed.hip"
#include "okx_metrics.hip"
#include "okx_shim.hip"
#include "okx_quad.hpp"
#include "../../include/okx_debug.h"

struct okx_program {
  okx::DevProgram host;        // host copy (dimensions, launch sizing)
  okx::DevProgram* dev;        // device copy
  int device;
  int n_cu;
  size_t lds_bytes;        // eval / rebind / single-problem solve kernels
  size_t solve_lds_bytes;  // selected solve kernel
  int blocks_per_cu;
  int nreg;                // padded row length of the register-resident factorisation
  const void* solve_fn;    // okx_solve_kernel<NREG> (one problem per wavefront; two wavefronts for n > 63)
  const void* eval_fn;     // okx_eval_kernel<threads>
  int threads;             // threads per problem of the generic kernels: 64, or 128 for n > 63
  const void* tangent_fn;  // okx_tangent_kernel<NREG> (generic tangents)
  int groups;              // problems per wavefront of the packed kernel (1 = not available)
  int group_width;         // lanes per problem in the packed kernel
  const void* packed_fn;   // okx_solve_packed_kernel<NREG, G> or null
  size_t packed_lds_bytes;
  int packed_blocks_per_cu;
  // runtime-specialised quad kernel (okx_quadgen.cpp / okx_jit.cpp); null when not available
  hipModule_t quad_mod;
  hipFunction_t quad_fn_u;  // program's own geometry
  hipFunction_t quad_fn_g;  // per-geometry tables
  hipFunction_t quad_fn_eval;  // parity kernel
  hipFunction_t quad_fn_expand;  // positions from free coordinates (single mode)
  hipFunction_t quad_fn_tan_u, quad_fn_tan_g;  // tangents (null when a free point is not an output point)
  int quad_waves_per_cu;
  int quad_ppw;             // problems per wavefront: 16 (one quad each) or 8 (pair mode: one quad per half)
  char quad_note[256];      // why the quad kernel is not in use (empty when it is)
  double* predictor_dev;    // chain-head model fitted by okx_program_fit_predictor, or null
  long long predictor_len;  // doubles in it
  // shared first step of the chain heads (okx_quad_head_u/_g; null functions: not generated for this program)
  hipFunction_t quad_fn_head_u, quad_fn_head_g;
  hipFunction_t quad_fn_cold_u;  // independent solves from the own geometry's design state with its first-step table (null: none)
  int head_stride;          // doubles per geometry in the table (okx::quad_head_stride)
  // own geometry's tables, one per lambda0 ever asked for (never overwritten: launches on other streams may still be
  // reading an older one); the default lambda0's is filled synchronously at okx_program_create, any other on first use on
  // the caller's stream, with an event that launches on other streams wait for
  struct HeadTable { double lambda0; double* dev; hipEvent_t ready; hipStream_t filled_on; };
  std::vector<HeadTable>* head_tables;
  std::mutex* head_mutex;
  // Tiered start.  A program whose generated kernels are not in the kernel cache is served by the interpreter kernels
  // while a host thread runs the compiler (hiprtc: no device call on that thread); the first entry point that finds the
  // job finished loads the code objects and switches the program over, under `head_mutex`.  Null: nothing pending.
  // The job owns everything it touches (its own copy of the host program, the code objects it produced): the program may
  // be destroyed while the compiler runs, and the switch-over loads the job's results from memory - the kernel cache on disk
  // is only a cache (a read-only cache directory must not cost a second compile).
  struct JitJob {
    std::thread thread;
    std::atomic<int> finished{0};
    std::atomic<int> quad_ready{0};   // the quad module is compiled (the lane module may still be in the works)
    std::atomic<int> quad_attached{0};  // ... and already switched over to (written under the program's jit_mutex)
    okx::DevProgram host;
    bool want_quad = false, want_lane = false;   // what was not in the cache at create
    bool quad_ok = false, lane_ok = false;
    std::string quad_code, quad_why, lane_code, lane_why;
    std::vector<okx::LaneOverride> lane_overrides;
  };
  std::atomic<JitJob*> jit;
  std::mutex* jit_mutex;
  // Generated-kernel state (module handles, function pointers, notes) is read by every launching entry point under a
  // shared lock and rewritten by the switch-over / okx_program_enable_evaluation under the exclusive one.
  std::shared_mutex* kern_mutex;
  double* head_geom_dev;    // scratch table of the latest launch with geometry tables (grow-only)
  long long head_geom_cap;  // geometries it holds
  double* quad_trace;            // diagnostic hook, see okx_debug_quad_trace (null: off)
  long long quad_trace_problem;
  // lane kernel (okx_lanegen.cpp): one lane per problem, for batches of at least lane_min_problems; null when the
  // program does not fit one lane's registers (or the quad kernel, whose first-step tables it shares, is absent)
  hipModule_t lane_mod;
  std::vector<hipModule_t>* lane_extra_mods;  // modules single kernels are taken from (okx::LaneOverride), or null
  hipFunction_t lane_fn_u, lane_fn_g, lane_fn_eval;  // independent solves (chain_len 1), parity kernel
  hipFunction_t lane_chain_u, lane_chain_g;          // chains
  hipFunction_t lane_compact[4];                     // the same four with compact outputs (solve_u, solve_g, chain_u, chain_g)
  hipFunction_t lane_nest[4];                        // nested start mode: u, g, u compact, g compact (null: none)
  hipFunction_t lane_refine[8] = {};                 // coarse-to-fine start (developer switch lane_refine): coarse u, g, u compact, g compact; warm likewise
  int lane_nest_scratch;
  long long lane_min_problems;
  int lane_cold_scratch, lane_chain_scratch;  // private-segment bytes of the two bodies (code object metadata)
  bool lane_cold_ok, lane_chain_ok;           // bodies that auto selection may use
  char lane_note[256];
  // evaluated modules (okx_program_enable_evaluation): the solve bodies with the tangent / metric epilogue, specialised to
  // one set of metric role points; null until enabled
  hipModule_t ev_mod, ev_lane_mod;
  hipFunction_t ev_solve_u, ev_solve_g, ev_cold_u, ev_pos_u, ev_pos_g;  // quad form (single mode)
  hipFunction_t ev_lane_u, ev_lane_g;                                   // lane form: independent solves (null: none)
  hipFunction_t ev_lane_pos_u = nullptr, ev_lane_pos_g = nullptr;       // lane form of okx_evaluate_batch (null: none)
  int ev_lane_scratch;
  okx::EvalSpec ev_spec;     // the role points compiled into them
  okx::EvalScalars ev_cfg;   // the roles' numeric part, a kernel argument
  // a composed axle's evaluated module (okx_program_enable_axle_evaluation): the same kernel slots, specialised to both
  // corners' role points and the roles' points and kinds
  bool ev_axle = false;
  okx::AxleEvalSpec ev_axle_spec;
  okx::EvalScalars ev_cfg_r;        // the right corner's numbers
  okx::EvalRoleNum ev_roles[8];     // the roles' numbers
  char ev_note[256];         // why there are none / no lane form
};

namespace {

thread_local char g_err[512] = "";
constexpr int kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950

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
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<15>;
    p->nreg = 15;
  } else if (n <= 18) {
    fn = okx::okx_solve_kernel<18, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<18>;
    p->nreg = 18;
  } else if (n <= 21) {
    fn = okx::okx_solve_kernel<21, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<21>;
    p->nreg = 21;
  } else if (n <= 24) {
    fn = okx::okx_solve_kernel<24, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<24>;
    p->nreg = 24;
  } else if (n <= 36) {
    fn = okx::okx_solve_kernel<36, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<36>;
    p->nreg = 36;
  } else if (n <= 48) {
    fn = okx::okx_solve_kernel<48, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<48>;
    p->nreg = 48;
  } else if (n <= 63) {
    fn = okx::okx_solve_kernel<63, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<63>;
    p->nreg = 63;
  } else {  // 64 ... 126 variables: two wavefronts per problem, LDL^T rows in LDS (okx_kernels.hip ldlt_solve_wide)
    fn = okx::okx_solve_kernel<126, false>;
    p->tangent_fn = (const void*)okx::okx_tangent_kernel<126>;
    p->nreg = 126;
  }
  p->solve_fn = (const void*)fn;
  p->threads = okx::GroupWidth<126>::value * (n > 63) + okx::kWave * (n <= 63);
  p->eval_fn = n > 63 ? (const void*)okx::okx_eval_kernel<2 * okx::kWave> : (const void*)okx::okx_eval_kernel<okx::kWave>;

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
int resident_blocks_per_cu(const void* fn, size_t lds_bytes, int threads = okx::kWave) {
  int occ = 32;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {
    const int alloc = (fa.numRegs + 7) / 8 * 8;
    int per_simd = 512 / alloc;
    if (per_simd > 8) per_simd = 8;
    if (per_simd < 1) per_simd = 1;
    occ = 4 * per_simd;
  }
  occ /= threads > okx::kWave ? threads / okx::kWave : 1;  // workgroups of two wavefronts (n > 63)
  const int by_lds = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
  if (by_lds < occ) occ = by_lds;
  if (occ < 1) occ = 1;
  return occ;
}

int quad_waves_per_simd() {
  // (developer switch quad_two_waves: __launch_bounds__(64, 2), i.e. at most 256 registers per lane - what a second resident
  //  wavefront per SIMD would need; profiles/r05/EXPERIMENTS.md section 4 has what the compiler makes of it)
  return okx::dev_switch("quad_two_waves") ? 2 : 1;
}

// The first-step table of the program's own geometry for `lambda0`: found, or filled by one wavefront of okx_quad_head_u
// on `stream` (a new buffer per lambda0: an older table may still be read by launches in flight).  A launch on another
// stream than the one that filled the table waits for the fill's event.
constexpr size_t kMaxHeadTables = 16;  // distinct lambda0 values with a table of their own per program

// true while `stream` records into a HIP graph: nothing may be allocated, filled or waited for on its behalf then
bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &status) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return status != hipStreamCaptureStatusNone;
}

// *table = nullptr with OKX_OK: no table for this launch (the chain heads take their own first pass): a launch that is
// being captured into a graph and finds no table of its lambda0 yet, or a program whose table list is full.
int own_head_table(okx_program* p, double lambda0, hipStream_t stream, double** table) {
  *table = nullptr;
  std::lock_guard<std::mutex> lock(*p->head_mutex);
  const bool capturing = stream_is_capturing(stream);
  for (okx_program::HeadTable& t : *p->head_tables)
    if (t.lambda0 == lambda0) {
      // (always ordered behind the fill: an event that has completed costs nothing, and a stream handle can be reused.
      //  Under capture the wait would become a graph dependency on an event outside the graph: the table of a captured
      //  launch must have been filled before the capture began - okx_program_create fills the default's synchronously.)
      if (!capturing) HIP_TRY(hipStreamWaitEvent(stream, t.ready, 0));
      *table = t.dev;
      return OKX_OK;
    }
  if (capturing || p->head_tables->size() >= kMaxHeadTables) return OKX_OK;
  okx_program::HeadTable t;
  t.lambda0 = lambda0;
  t.filled_on = stream;
  HIP_TRY(hipMalloc((void**)&t.dev, sizeof(double) * ((size_t)p->head_stride + 2)));  // (+ pad: the cold body reads the table in 16-byte pieces)
  if (hipEventCreateWithFlags(&t.ready, hipEventDisableTiming) != hipSuccess) {
    (void)hipFree(t.dev);
    return fail(OKX_ERR_DEVICE, "hipEventCreate failed");
  }
  okx::QuadHeadArgs h;
  h.geom_pos = nullptr;
  h.geom_row_param = nullptr;
  h.head = t.dev;
  h.n_geometries = 1;
  h.lambda0 = lambda0;
  const char* base = reinterpret_cast<const char*>(p->dev);
  h.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
  h.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
  h.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
  void* hargs[] = {(void*)&h};
  hipError_t e = hipModuleLaunchKernel(p->quad_fn_head_u, 1, 1, 1, okx::kWave, 1, 1, 0, stream, hargs, nullptr);
  if (e == hipSuccess) e = hipEventRecord(t.ready, stream);
  if (e != hipSuccess) {
    (void)hipEventDestroy(t.ready);
    (void)hipFree(t.dev);
    return fail(OKX_ERR_DEVICE, "first-step table: %s", hipGetErrorString(e));
  }
  p->head_tables->push_back(t);
  *table = t.dev;
  return OKX_OK;
}

// Generate, compile (or fetch from the cache) and load the kernel specialised to this program.
// Failure is not an error of okx_program_create: the generic kernels stay in charge and
// okx_program_kernel_note() says why.
// `cache_only`: only what the kernel cache already holds (okx_program_create); a miss sets *pending and leaves the
// interpreter kernels in charge until the compile job has filled the cache.
// `job`: the compile job's results (switch-over): the code object comes from memory, the compiler is never run here.
void attach_quad_kernel(okx_program* p, bool cache_only = false, bool* pending = nullptr, const okx_program::JitJob* job = nullptr) {
  p->quad_mod = nullptr;
  p->quad_fn_u = p->quad_fn_g = nullptr;
  p->quad_fn_cold_u = nullptr;
  p->quad_fn_eval = nullptr;
  p->quad_fn_expand = nullptr;
  p->quad_fn_tan_u = p->quad_fn_tan_g = nullptr;
  p->quad_waves_per_cu = 0;
  p->quad_ppw = p->host.n_free > okx::kQuadMaxFree ? 8 : 16;
  p->quad_note[0] = 0;
  if (okx::dev_switch("no_quad")) {  // (tests: the interpreter kernels on a program that has generated ones)
    std::snprintf(p->quad_note, sizeof(p->quad_note), "disabled by OKX_DEV=no_quad");
    return;
  }
  std::string src, why, code;
  if (job) {
    if (!job->quad_ok) {
      std::snprintf(p->quad_note, sizeof(p->quad_note), "not generated: %.200s", job->quad_why.c_str());
      if (getenv("OKX_VERBOSE")) std::fprintf(stderr, "okx: quad kernel: %s\n", job->quad_why.c_str());
      return;
    }
    code = job->quad_code;
  } else if (!okx::quad_build(p->host, quad_waves_per_simd(), &src, &code, &why, false, cache_only)) {
    if (cache_only && why == okx::kNotCached) {
      if (pending) *pending = true;
      std::snprintf(p->quad_note, sizeof(p->quad_note), "being compiled (the interpreter kernels serve the program until then)");
      return;
    }
    std::snprintf(p->quad_note, sizeof(p->quad_note), "not generated: %.200s", why.c_str());
    if (getenv("OKX_VERBOSE")) std::fprintf(stderr, "okx: quad kernel: %s\n", why.c_str());
    return;
  }
  hipModule_t mod = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess && !job) {
    // a damaged cache entry (truncated file, other toolchain): rebuild it once
    (void)hipGetLastError();
    if (okx::quad_build(p->host, quad_waves_per_simd(), &src, &code, &why, true)) e = hipModuleLoadData(&mod, code.data());
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    std::snprintf(p->quad_note, sizeof(p->quad_note), "hipModuleLoadData: %s", hipGetErrorString(e));
    return;
  }
  hipFunction_t fu = nullptr, fg = nullptr;
  if (hipModuleGetFunction(&fu, mod, "okx_quad_solve_u") != hipSuccess ||
      hipModuleGetFunction(&fg, mod, "okx_quad_solve_g") != hipSuccess) {
    (void)hipModuleUnload(mod);
    std::snprintf(p->quad_note, sizeof(p->quad_note), "kernel symbols missing in the code object");
    return;
  }
  int regs = 0;
  int per_simd = quad_waves_per_simd();
  if (hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fg) == hipSuccess && regs > 0) {
    const int alloc = (regs + 7) / 8 * 8;
    per_simd = 512 / alloc;
    if (per_simd > 8) per_simd = 8;
    if (per_simd < 1) per_simd = 1;
  }
  p->quad_mod = mod;
  p->quad_fn_g = fg;
  if (hipModuleGetFunction(&p->quad_fn_eval, mod, "okx_quad_eval") != hipSuccess) p->quad_fn_eval = nullptr;
  if (hipModuleGetFunction(&p->quad_fn_expand, mod, "okx_quad_expand") != hipSuccess) p->quad_fn_expand = nullptr;
  if (hipModuleGetFunction(&p->quad_fn_tan_u, mod, "okx_quad_tangent_u") != hipSuccess ||
      hipModuleGetFunction(&p->quad_fn_tan_g, mod, "okx_quad_tangent_g") != hipSuccess)
    p->quad_fn_tan_u = p->quad_fn_tan_g = nullptr;
  p->quad_fn_head_u = p->quad_fn_head_g = nullptr;
  if (hipModuleGetFunction(&p->quad_fn_cold_u, mod, "okx_quad_cold_u") != hipSuccess) p->quad_fn_cold_u = nullptr;
  p->head_stride = okx::quad_head_stride(p->host);
  if (!(p->head_stride > 0 && hipModuleGetFunction(&p->quad_fn_head_u, mod, "okx_quad_head_u") == hipSuccess &&
        hipModuleGetFunction(&p->quad_fn_head_g, mod, "okx_quad_head_g") == hipSuccess))
    p->quad_fn_head_u = p->quad_fn_head_g = nullptr;
  (void)hipGetLastError();  // optional kernels absent from a module must not leave a sticky error behind
  p->quad_waves_per_cu = 4 * per_simd;
  // The first-step table of the program's own geometry for the default damping belongs to the program's set-up, like the
  // kernel itself: filled here (one wavefront, ~10 us) on a private non-blocking stream that is waited for before this
  // returns - no solve launch ever pays for it, a later stream capture finds it complete, and nothing is ordered against
  // the legacy stream (a synchronise there would serialise every blocking stream of the process).
  if (p->quad_fn_head_u) {
    okx_solve_opts o;
    okx_default_opts(&o);
    double* unused = nullptr;
    hipStream_t fill = nullptr;
    bool ok = hipStreamCreateWithFlags(&fill, hipStreamNonBlocking) == hipSuccess;
    ok = ok && own_head_table(p, o.lambda0, fill, &unused) == OKX_OK && hipStreamSynchronize(fill) == hipSuccess;
    if (fill) (void)hipStreamDestroy(fill);
    if (!ok) {
      (void)hipGetLastError();
      p->quad_fn_head_u = p->quad_fn_head_g = nullptr;
    }
  }
  // the gate every launch path tests, published last (a program may be switched over while it is in use)
  std::atomic_thread_fence(std::memory_order_release);
  p->quad_fn_u = fu;
}

// The lane kernel of a program that has a quad kernel (same policy: failure only means the quad kernel serves every
// batch size; okx_program_lane_note() says why).
// Largest scratch among the four kernels `prefix`_u / _u_c / _g / _g_c as they will be launched: a kernel that spills less in
// another emission variant's module is taken from there (okx::LaneOverride).
static int lane_worst_scratch(const std::string& code, const std::vector<okx::LaneOverride>& overrides, const char* prefix) {
  int worst = -1;
  for (const char* tail : {"_u", "_u_c", "_g", "_g_c"}) {
    const std::string kernel = std::string(prefix) + tail;
    int sc = okx::quad_code_kernel_scratch_bytes(code, kernel.c_str());
    for (const okx::LaneOverride& o : overrides)
      if (o.kernel == kernel && o.scratch >= 0) sc = o.scratch;
    if (sc > worst) worst = sc;
  }
  return worst;
}

void attach_lane_kernel(okx_program* p, bool cache_only = false, bool* pending = nullptr, const okx_program::JitJob* job = nullptr) {
  p->lane_mod = nullptr;
  p->lane_extra_mods = nullptr;
  p->lane_fn_u = p->lane_fn_g = p->lane_fn_eval = nullptr;
  p->lane_chain_u = p->lane_chain_g = nullptr;
  p->lane_nest[0] = p->lane_nest[1] = p->lane_nest[2] = p->lane_nest[3] = nullptr;
  p->lane_note[0] = 0;
  // The quad kernel runs 16 problems per wavefront, one wavefront per SIMD: up to n_cu * 4 * 16 problems (16384) are ONE
  // round of it (~21 us for the double wishbone).  One problem more is a second round (~38 us), while the lane kernel
  // takes 25 ... 29 us for anything up to n_cu * 4 * 64 problems (profiles/r03/EXPERIMENTS.md): auto selection switches there.
  p->lane_min_problems = (long long)p->n_cu * 4 * 16 + 1;
  if (!p->quad_fn_u || p->quad_ppw != 16) {
    std::snprintf(p->lane_note, sizeof(p->lane_note), "no single-mode quad kernel to share first-step tables with");
    return;
  }
  if (okx::dev_switch("no_lane")) {
    std::snprintf(p->lane_note, sizeof(p->lane_note), "disabled by OKX_DEV=no_lane");
    return;
  }
  std::string src, why, code;
  std::vector<okx::LaneOverride> overrides;
  if (job) {
    if (!job->lane_ok) {
      std::snprintf(p->lane_note, sizeof(p->lane_note), "not generated: %.200s", job->lane_why.c_str());
      return;
    }
    code = job->lane_code;
    overrides = job->lane_overrides;
  } else if (!okx::lane_build(p->host, &src, &code, &why, false, nullptr, 256, cache_only, &overrides)) {
    if (cache_only && why == okx::kNotCached) {
      if (pending) *pending = true;
      std::snprintf(p->lane_note, sizeof(p->lane_note), "being compiled");
      return;
    }
    std::snprintf(p->lane_note, sizeof(p->lane_note), "not generated: %.200s", why.c_str());
    return;
  }
  // A body that spills is only worth having while the spill is small.  Measured on MI355X: the double wishbone's
  // independent-solve body (104 - 192 B of scratch) is still 1.8x the quad kernel on 4096 geometries x 256 steps, its
  // looping chain body (668 B) was 8 % slower than the quad kernel's chains; MacPherson (0 B) wins both ways.
  p->lane_cold_scratch = lane_worst_scratch(code, overrides, "okx_lane_solve");
  p->lane_chain_scratch = lane_worst_scratch(code, overrides, "okx_lane_chain");
  p->lane_cold_ok = p->lane_cold_scratch >= 0 && p->lane_cold_scratch <= 256;
  // ... and a flat chain body (okx_quad.hpp lane_chain_is_flat: the double wishbone; 0 B of scratch) is correct but does not
  // pay: each chain step repeats the independent solve's prologue and its records leave lane by lane, so 4096 x 256 in
  // chains of 4 takes 0.62 ms against 0.51 ms of independent solves, and a 1048576-step sweep of one geometry 0.47
  // against 0.43 ms although its evaluations drop from 2.97 to 1.53 (tools/lane_chain_modes.py, lane_chain_own.py).
  // Auto selection keeps resolving chain_len = -1 to independent solves there; kernel = 4 with chains runs it.
  p->lane_chain_ok = p->lane_chain_scratch == 0 && !okx::lane_chain_is_flat(p->host.n);
  if (!p->lane_cold_ok && !p->lane_chain_ok) {
    std::snprintf(p->lane_note, sizeof(p->lane_note), "the lane kernel of this program spills (%d / %d B of scratch): not used",
                  p->lane_cold_scratch, p->lane_chain_scratch);
    return;
  }
  hipModule_t mod = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess && !cache_only && !job) {
    (void)hipGetLastError();
    if (okx::lane_build(p->host, &src, &code, &why, true, nullptr, 256)) e = hipModuleLoadData(&mod, code.data());
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    std::snprintf(p->lane_note, sizeof(p->lane_note), "hipModuleLoadData: %s", hipGetErrorString(e));
    return;
  }
  hipFunction_t lane_u = nullptr;
  if (hipModuleGetFunction(&lane_u, mod, "okx_lane_solve_u") != hipSuccess ||
      hipModuleGetFunction(&p->lane_fn_g, mod, "okx_lane_solve_g") != hipSuccess ||
      hipModuleGetFunction(&p->lane_chain_u, mod, "okx_lane_chain_u") != hipSuccess ||
      hipModuleGetFunction(&p->lane_chain_g, mod, "okx_lane_chain_g") != hipSuccess ||
      hipModuleGetFunction(&p->lane_compact[0], mod, "okx_lane_solve_u_c") != hipSuccess ||
      hipModuleGetFunction(&p->lane_compact[1], mod, "okx_lane_solve_g_c") != hipSuccess ||
      hipModuleGetFunction(&p->lane_compact[2], mod, "okx_lane_chain_u_c") != hipSuccess ||
      hipModuleGetFunction(&p->lane_compact[3], mod, "okx_lane_chain_g_c") != hipSuccess ||
      hipModuleGetFunction(&p->lane_fn_eval, mod, "okx_lane_eval") != hipSuccess) {
    (void)hipGetLastError();
    (void)hipModuleUnload(mod);
    p->lane_fn_u = p->lane_fn_g = p->lane_fn_eval = nullptr;
    p->lane_chain_u = p->lane_chain_g = nullptr;
    std::snprintf(p->lane_note, sizeof(p->lane_note), "kernel symbols missing in the code object");
    return;
  }
  for (const okx::LaneOverride& o : overrides) {
    hipFunction_t* slot = o.kernel == "okx_lane_solve_u" ? &lane_u : o.kernel == "okx_lane_solve_g" ? &p->lane_fn_g :
                          o.kernel == "okx_lane_chain_u" ? &p->lane_chain_u : o.kernel == "okx_lane_chain_g" ? &p->lane_chain_g :
                          o.kernel == "okx_lane_solve_u_c" ? &p->lane_compact[0] : o.kernel == "okx_lane_solve_g_c" ? &p->lane_compact[1] :
                          o.kernel == "okx_lane_chain_u_c" ? &p->lane_compact[2] : o.kernel == "okx_lane_chain_g_c" ? &p->lane_compact[3] : nullptr;
    hipModule_t extra = nullptr;
    hipFunction_t fn = nullptr;
    if (!slot || hipModuleLoadData(&extra, o.code.data()) != hipSuccess) {
      (void)hipGetLastError();
      continue;  // (the kept module's kernel stays)
    }
    if (hipModuleGetFunction(&fn, extra, o.kernel.c_str()) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipModuleUnload(extra);
      continue;
    }
    if (!p->lane_extra_mods) p->lane_extra_mods = new std::vector<hipModule_t>;
    p->lane_extra_mods->push_back(extra);
    *slot = fn;
  }
  // the nested start mode (chain_len = -1 on large sweeps): kept while it does not spill more than the independent-solve body may
  static const char* const kNest[4] = {"okx_lane_nest_u", "okx_lane_nest_g", "okx_lane_nest_u_c", "okx_lane_nest_g_c"};
  p->lane_nest_scratch = okx::quad_code_scratch_bytes(code, "okx_lane_nest");
  for (int k = 0; k < 4; ++k)
    if (hipModuleGetFunction(&p->lane_nest[k], mod, kNest[k]) != hipSuccess) {
      (void)hipGetLastError();
      p->lane_nest[k] = nullptr;
    }
  if (p->lane_nest_scratch < 0 || p->lane_nest_scratch > (okx::dev_switch("lane_timeline") ? 1 << 20 : 256) || !p->lane_nest[0] || !p->lane_nest[1] || !p->lane_nest[2] || !p->lane_nest[3])
    p->lane_nest[0] = p->lane_nest[1] = p->lane_nest[2] = p->lane_nest[3] = nullptr;
  {  // coarse-to-fine start: present only in modules generated with the developer switch
    static const char* const kRefine[8] = {"okx_lane_refc_u", "okx_lane_refc_g", "okx_lane_refc_u_c", "okx_lane_refc_g_c",
                                           "okx_lane_refw_u", "okx_lane_refw_g", "okx_lane_refw_u_c", "okx_lane_refw_g_c"};
    bool all = true;
    for (int k = 0; k < 8; ++k)
      if (hipModuleGetFunction(&p->lane_refine[k], mod, kRefine[k]) != hipSuccess) {
        (void)hipGetLastError();
        all = false;
      }
    const int refine_scratch = all ? okx::quad_code_scratch_bytes(code, "okx_lane_ref") : -1;
    if (!all || refine_scratch < 0 || refine_scratch > 512)
      for (int k = 0; k < 8; ++k) p->lane_refine[k] = nullptr;
  }
  p->lane_mod = mod;
  std::atomic_thread_fence(std::memory_order_release);
  p->lane_fn_u = lane_u;  // the gate of the lane kernel's launch path, published last
}

// The compile job of a program whose kernels were not in the cache: compiles what was missing (same calls, same policy as
// the attach functions), keeps the results in the job and - through quad_compile - in the cache.  Touches nothing but
// the job.  No device call on this thread.
void jit_job(okx_program::JitJob* job) {
  std::string src;
  if (job->want_quad) job->quad_ok = okx::quad_build(job->host, quad_waves_per_simd(), &src, &job->quad_code, &job->quad_why);
  job->quad_ready.store(1, std::memory_order_release);  // (a launch may switch over to the quad kernels now: two stages)
  if (job->want_lane && (job->quad_ok || !job->want_quad) && job->host.n_free <= okx::kQuadMaxFree && !okx::dev_switch("no_lane")) {
    std::string lsrc;
    job->lane_ok = okx::lane_build(job->host, &lsrc, &job->lane_code, &job->lane_why, false, nullptr, 256, false, &job->lane_overrides);
  } else if (job->want_lane) {
    job->lane_why = "no quad kernel to share first-step tables with";
  }
  job->finished.store(1, std::memory_order_release);
}

// Jobs of programs that were destroyed while their compiler was still running: joined (and freed) by a later destroy that
// finds them finished, or at process exit - a short script that drops its program early still leaves a filled cache behind,
// at the price of waiting for the compiler when it exits.
std::mutex g_orphan_mutex;
std::vector<okx_program::JitJob*> g_orphans;
void reap_orphans(bool wait) {
  std::lock_guard<std::mutex> lock(g_orphan_mutex);
  for (size_t k = 0; k < g_orphans.size();) {
    okx_program::JitJob* job = g_orphans[k];
    if (!wait && !job->finished.load(std::memory_order_acquire)) {
      ++k;
      continue;
    }
    job->thread.join();
    delete job;
    g_orphans.erase(g_orphans.begin() + (long)k);
  }
}
void reap_orphans_at_exit() { reap_orphans(true); }

// Switches a program over to its generated kernels once the compile job is done (`wait`: block until it is).  Called at
// the top of every entry point that launches; costs one atomic load when nothing is pending.  Never while the caller's
// stream records a graph (`stream` non-null: the launch's stream) - module loads and allocations are illegal inside a
// capture; the interpreter serves that launch and a later one switches over.
void attach_when_ready(okx_program* p, bool wait, const hipStream_t* stream = nullptr) {
  okx_program::JitJob* job = p->jit.load(std::memory_order_acquire);
  if (!job) return;
  const bool all_done = wait || job->finished.load(std::memory_order_acquire);
  // first stage: the quad module alone is ready (the lane module's variants take another minute) - switch over to it now,
  // it serves every batch size until the lane kernels arrive
  const bool quad_stage = !all_done && job->want_quad && job->want_lane && job->quad_ready.load(std::memory_order_acquire) && !job->quad_attached.load(std::memory_order_acquire);
  if (!all_done && !quad_stage) return;
  if (!wait && stream && stream_is_capturing(*stream)) return;
  std::lock_guard<std::mutex> lock(*p->jit_mutex);
  job = p->jit.load(std::memory_order_acquire);
  if (!job) return;  // another caller got here first
  if (!all_done) {
    if (job->quad_attached.load(std::memory_order_acquire)) return;  // (another caller did the first stage meanwhile)
    int current = p->device;
    (void)hipGetDevice(&current);
    if (current != p->device) (void)hipSetDevice(p->device);
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    {
      std::unique_lock<std::shared_mutex> kernels(*p->kern_mutex);
      attach_quad_kernel(p, false, nullptr, job);   // (reads job->quad_code / quad_why only: written before quad_ready)
      job->quad_attached.store(1, std::memory_order_release);
    }
    if (exchanged) (void)hipThreadExchangeStreamCaptureMode(&mode);
    (void)hipGetLastError();
    if (current != p->device) (void)hipSetDevice(current);
    return;
  }
  job->thread.join();
  int current = p->device;
  (void)hipGetDevice(&current);
  if (current != p->device) (void)hipSetDevice(p->device);
  // (another thread of the process may be capturing in global mode - torch's default: this thread's module loads and
  //  allocations must not invalidate that capture)
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  {
    std::unique_lock<std::shared_mutex> kernels(*p->kern_mutex);  // launches in progress finish first, later ones see the new set
    if (job->want_quad && !job->quad_attached.load(std::memory_order_acquire)) attach_quad_kernel(p, false, nullptr, job);
    if (job->want_lane) attach_lane_kernel(p, false, nullptr, job);
  }
  if (exchanged) (void)hipThreadExchangeStreamCaptureMode(&mode);
  (void)hipGetLastError();
  if (current != p->device) (void)hipSetDevice(current);
  p->jit.store(nullptr, std::memory_order_release);
  delete job;
}

// frees the first-step tables and the host-side containers (every exit path of okx_program_create / _destroy)
void release_host_side(okx_program* p) {
  if (p->head_tables) {
    for (okx_program::HeadTable& t : *p->head_tables) {
      (void)hipEventDestroy(t.ready);
      (void)hipFree(t.dev);
    }
    delete p->head_tables;
    p->head_tables = nullptr;
  }
  delete p->head_mutex;
  p->head_mutex = nullptr;
  delete p->jit_mutex;
  p->jit_mutex = nullptr;
  delete p->kern_mutex;
  p->kern_mutex = nullptr;
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
  o->confirm_full_pass = 0;
  o->predictor = 0;
  o->shared_first_step = 1;
  o->output = OKX_OUTPUT_RECORDS;
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
  p->head_tables = new (std::nothrow) std::vector<okx_program::HeadTable>();
  p->head_mutex = new (std::nothrow) std::mutex();
  p->jit_mutex = new (std::nothrow) std::mutex();
  p->kern_mutex = new (std::nothrow) std::shared_mutex();
  if (!p->head_tables || !p->head_mutex || !p->jit_mutex || !p->kern_mutex) {
    delete p->head_tables;
    delete p->head_mutex;
    delete p->jit_mutex;
    delete p->kern_mutex;
    delete p;
    return fail(OKX_ERR_ALLOC, "out of host memory");
  }
  int rc = okx::build_dev_program(desc, &p->host, g_err, (int)sizeof(g_err));
  if (rc != OKX_OK) {
    release_host_side(p);
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
    const size_t need = p->lds_bytes;
    release_host_side(p);
    delete p;
    return fail(OKX_ERR_LIMIT, "problem needs %zu bytes of LDS (max 163840)", need);
  }
  hipError_t e = hipGetDevice(&p->device);
  if (e != hipSuccess) {
    release_host_side(p);
    delete p;
    return fail(OKX_ERR_DEVICE, "hipGetDevice failed: %s (no GPU?)", hipGetErrorString(e));
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, p->device);
  if (e != hipSuccess) {
    release_host_side(p);
    delete p;
    return fail(OKX_ERR_DEVICE, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
  }
  p->n_cu = prop.multiProcessorCount;
  e = hipMalloc((void**)&p->dev, sizeof(okx::DevProgram));
  if (e != hipSuccess) {
    release_host_side(p);
    delete p;
    return fail(OKX_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  e = hipMemcpy(p->dev, &p->host, sizeof(okx::DevProgram), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p->dev);
    release_host_side(p);
    delete p;
    return fail(OKX_ERR_DEVICE, "hipMemcpy failed: %s", hipGetErrorString(e));
  }
  // >64 KiB of dynamic LDS needs the opt-in attribute.  The kernels are shared by every program of the
  // process, so the limit is raised to the hardware maximum once per kernel (a per-program size would let a
  // later, smaller program lower it under an earlier one's feet).
  for (const void* fn : {p->solve_fn, p->eval_fn, (const void*)okx::okx_rebind_kernel,
                         p->packed_fn, p->tangent_fn}) {
    if (!fn) continue;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
    if (e != hipSuccess) {
      (void)hipFree(p->dev);
      release_host_side(p);
      delete p;
      return fail(OKX_ERR_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    }
  }
  p->blocks_per_cu = resident_blocks_per_cu(p->solve_fn, p->solve_lds_bytes, p->threads);
  p->packed_blocks_per_cu = p->packed_fn ? resident_blocks_per_cu(p->packed_fn, p->packed_lds_bytes) : 0;
  // Generated kernels: loaded here when the kernel cache has them (the usual case: __graft_entry__.build() and
  // okx_precompile fill it).  Otherwise a host thread compiles them (10 ... 80 s per module) while this call returns at
  // once and the interpreter kernels solve; the first launch after the job is done switches the program over.
  bool pending_quad = false, pending_lane = false;
  attach_quad_kernel(p, true, &pending_quad);
  if (!pending_quad) attach_lane_kernel(p, true, &pending_lane);
  if (pending_quad || pending_lane) {
    okx_program::JitJob* job = new (std::nothrow) okx_program::JitJob;
    if (job) {
      job->host = p->host;
      job->want_quad = pending_quad;
      job->want_lane = true;  // (a pending quad kernel means the lane kernel, which shares its tables, was not looked at yet)
      try {
        job->thread = std::thread(jit_job, job);
        p->jit.store(job, std::memory_order_release);
      } catch (...) {  // no thread to be had: compile here, as before
        delete job;
        job = nullptr;
      }
    }
    if (!job) {
      if (pending_quad) attach_quad_kernel(p);
      attach_lane_kernel(p);
    }
  }
  *out = p;
  return OKX_OK;
}

int32_t okx_program_ready(okx_program* p, int32_t wait) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  attach_when_ready(p, wait != 0);
  return p->jit.load(std::memory_order_acquire) ? 0 : 1;
}

void okx_program_destroy(okx_program* p) {
  if (!p) return;
  if (okx_program::JitJob* job = p->jit.exchange(nullptr)) {
    // a compile job still pending: it owns what it works on, so the program goes now and the job is joined when it has
    // finished (by a later destroy, or at exit: it still fills the cache for the next program) - never a wait of minutes here
    if (job->finished.load(std::memory_order_acquire)) {
      job->thread.join();
      delete job;
    } else {
      std::lock_guard<std::mutex> lock(g_orphan_mutex);
      static bool registered = false;
      if (!registered) {
        registered = true;
        std::atexit(reap_orphans_at_exit);
      }
      g_orphans.push_back(job);
    }
  }
  reap_orphans(false);
  if (p->ev_mod) (void)hipModuleUnload(p->ev_mod);
  if (p->ev_lane_mod) (void)hipModuleUnload(p->ev_lane_mod);
  if (p->quad_mod) (void)hipModuleUnload(p->quad_mod);
  if (p->lane_mod) (void)hipModuleUnload(p->lane_mod);
  if (p->lane_extra_mods) {
    for (hipModule_t m : *p->lane_extra_mods) (void)hipModuleUnload(m);
    delete p->lane_extra_mods;
  }
  if (p->predictor_dev) (void)hipFree(p->predictor_dev);
  release_host_side(p);
  if (p->head_geom_dev) (void)hipFree(p->head_geom_dev);
  if (p->dev) (void)hipFree(p->dev);
  delete p;
}

const char* okx_program_kernel(const okx_program* p) {
  if (!p) return "";
  return p->quad_fn_u ? "quad" : "wave";
}

const char* okx_program_kernel_note(const okx_program* p) { return p ? p->quad_note : ""; }

/* 1 when chain heads of this program's own geometry take their first step from the shared first-step table
   (okx_solve_opts.shared_first_step with a generated head kernel): their okx_info.nfev then omits that evaluation. */
int32_t okx_program_shares_first_step(const okx_program* p) { return p && p->quad_fn_head_u ? 1 : 0; }
int32_t okx_program_has_cold_body(const okx_program* p) { return p && p->quad_fn_cold_u && p->quad_fn_head_u ? 1 : 0; }

/* Why the program has no lane kernel (empty string: it has one), and the batch size from which auto selection uses it. */
const char* okx_program_lane_note(const okx_program* p) { return p ? p->lane_note : ""; }
int64_t okx_program_lane_threshold(const okx_program* p) { return p && p->lane_fn_u ? p->lane_min_problems : -1; }
/* bit0: auto selection uses the lane kernel's independent-solve body, bit1: its chain body (0: neither / no lane kernel) */
int32_t okx_program_lane_bodies(const okx_program* p) {
  return p && p->lane_fn_u ? (p->lane_cold_ok ? 1 : 0) | (p->lane_chain_ok ? 2 : 0) : 0;
}

/* Generated source of the lane kernel for a program (no device needed); same contract as okx_quad_source. */
int64_t okx_lane_source(const okx_program_desc* desc, char* buf, int64_t buflen) {
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  std::string src, why;
  if (rc == OKX_OK && !okx::lane_generate(*tmp, &src, &why))
    rc = fail(OKX_ERR_LIMIT, "no lane kernel for this program: %s", why.c_str());
  delete tmp;
  if (rc != OKX_OK) return rc;
  if (buf && buflen > 0) {
    const size_t ncopy = src.size() < (size_t)buflen - 1 ? src.size() : (size_t)buflen - 1;
    std::memcpy(buf, src.data(), ncopy);
    buf[ncopy] = 0;
  }
  return (int64_t)src.size() + 1;
}

/* Generated source of the quad kernel for a program (no device needed).  Returns the number of
   bytes the full text needs (including the terminator) or a negative okx_status. */
int64_t okx_quad_source(const okx_program_desc* desc, char* buf, int64_t buflen) {
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  std::string src, why;
  if (rc == OKX_OK && !okx::quad_generate(*tmp, quad_waves_per_simd(), &src, &why))
    rc = fail(OKX_ERR_LIMIT, "no quad kernel for this program: %s", why.c_str());
  delete tmp;
  if (rc != OKX_OK) return rc;
  if (buf && buflen > 0) {
    const size_t ncopy = src.size() < (size_t)buflen - 1 ? src.size() : (size_t)buflen - 1;
    std::memcpy(buf, src.data(), ncopy);
    buf[ncopy] = 0;
  }
  return (int64_t)src.size() + 1;
}

/* Generates and compiles the quad kernel of a program into the on-disk cache (no device needed):
   what __graft_entry__.build() calls for the BASELINE topologies. */
int32_t okx_precompile(const okx_program_desc* desc) {
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  std::string src, why, code;
  if (rc == OKX_OK && !okx::quad_build(*tmp, quad_waves_per_simd(), &src, &code, &why))
    rc = fail(why.compare(0, 14, "compile failed") == 0 ? OKX_ERR_DEVICE : OKX_ERR_LIMIT, "no quad kernel for this program: %s", why.c_str());
  if (rc == OKX_OK && tmp->n_free <= okx::kQuadMaxFree) {
    // the lane kernel of the same program (programs it does not fit simply have none)
    std::string lsrc, lwhy, lcode;
    std::vector<okx::LaneOverride> overrides;  // (asked for so that the per-kernel choice is made and remembered)
    if (okx::lane_generate(*tmp, &lsrc, &lwhy, 0) && !okx::lane_build(*tmp, &lsrc, &lcode, &lwhy, false, nullptr, 0, false, &overrides))
      rc = fail(OKX_ERR_DEVICE, "lane kernel: %s", lwhy.c_str());
  }
  delete tmp;
  return rc;
}

/* Scratch (private segment) bytes of the solve kernels okx_precompile / okx_program_create would use for this program:
   0 means the register allocation holds everything (what every BASELINE program is expected to report). */
int32_t okx_debug_kernel_scratch(const okx_program_desc* desc, int32_t* scratch_bytes) {
  if (!scratch_bytes) return fail(OKX_ERR_INVALID, "null output pointer");
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  std::string src, why, code;
  if (rc == OKX_OK && !okx::quad_build(*tmp, quad_waves_per_simd(), &src, &code, &why))
    rc = fail(OKX_ERR_LIMIT, "no quad kernel for this program: %s", why.c_str());
  delete tmp;
  if (rc != OKX_OK) return rc;
  *scratch_bytes = okx::quad_code_scratch_bytes(code, "okx_quad_solve");
  return OKX_OK;
}

/* The lane kernel of a program as okx_precompile / okx_program_create would build it: scratch bytes of its independent-solve
   and chain bodies and the emission variant lane_build kept (out3; no device needed). */
int32_t okx_debug_lane_scratch(const okx_program_desc* desc, int32_t* out3) {
  if (!out3) return fail(OKX_ERR_INVALID, "null output pointer");
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  std::string src, why, code;
  int variant = -1;
  std::vector<okx::LaneOverride> overrides;
  if (rc == OKX_OK && !okx::lane_build(*tmp, &src, &code, &why, false, &variant, 0, false, &overrides))
    rc = fail(OKX_ERR_LIMIT, "no lane kernel for this program: %s", why.c_str());
  delete tmp;
  if (rc != OKX_OK) return rc;
  out3[0] = lane_worst_scratch(code, overrides, "okx_lane_solve");
  out3[1] = lane_worst_scratch(code, overrides, "okx_lane_chain");
  out3[2] = variant;
  return OKX_OK;
}

static int32_t check_corner_roles(const okx_corner_roles* roles, int32_t n_out, const char* who);

// okx_solve_batch and okx_solve_evaluated_batch (`evaluated`: the launch runs the program's evaluated kernels, whose
// epilogue writes d_tangents / d_eval)
static int32_t solve_impl(okx_program* p, const okx_solve_opts* opts, int64_t n_problems,
                          const double* d_targets, const double* d_geom_pos,
                          const double* d_geom_row_param, double* d_out_pos, okx_info* d_info,
                          void* stream, bool evaluated, double* d_tangents, double* d_eval, int32_t* plan_only = nullptr) {
  if (!p || !opts) return fail(OKX_ERR_INVALID, "null program or options");
  {
    const hipStream_t launch_stream = (hipStream_t)stream;
    attach_when_ready(p, false, &launch_stream);
  }
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (evaluated) {
    if (!p->ev_solve_u) return fail(OKX_ERR_INVALID, "evaluated solves need okx_program_enable_evaluation first%s%s", p->ev_note[0] ? ": " : "", p->ev_note);
    if (!d_tangents && !d_eval) return fail(OKX_ERR_INVALID, "an evaluated solve needs d_tangents or d_eval");
  }
  if (n_problems < 0) return fail(OKX_ERR_INVALID, "negative problem count");
  if (n_problems == 0) return OKX_OK;
  if (opts->output < OKX_OUTPUT_RECORDS || opts->output > OKX_OUTPUT_NONE) return fail(OKX_ERR_INVALID, "unknown output mode");
  if ((!d_out_pos && opts->output != OKX_OUTPUT_NONE) || !d_info) return fail(OKX_ERR_INVALID, "null output pointer");
  if (p->host.n_targets > 0 && !d_targets) return fail(OKX_ERR_INVALID, "null targets");
  if ((d_geom_pos == nullptr) != (d_geom_row_param == nullptr))
    return fail(OKX_ERR_INVALID, "geometry positions and row parameters must be given together");
  const long long spg = opts->steps_per_geometry;
  if (spg < 0 || (spg > 0 && n_problems % spg != 0))
    return fail(OKX_ERR_INVALID, "n_problems must be a multiple of steps_per_geometry");
  if (d_geom_pos && spg == 0 )
    return fail(OKX_ERR_INVALID, "a geometry table needs steps_per_geometry > 0");
  if (opts->max_iter < 1) return fail(OKX_ERR_INVALID, "max_iter must be >= 1");
  if (!(opts->lambda0 >= 0.0) || !(opts->lambda0 < 1e300)) return fail(OKX_ERR_INVALID, "lambda0 must be finite and >= 0");
  okx::SolveArgs a;
  a.targets = d_targets;
  a.geom_pos = d_geom_pos;
  a.geom_row_param = d_geom_row_param;
  a.out_pos = d_out_pos;
  a.info = d_info;
  a.n_problems = n_problems;
  a.steps_per_geometry = spg;
  a.max_iter = opts->max_iter;
  // the predicted-convergence ending is not offered along the reference's zero-gradient
  // point-on-line valley: the step length says nothing about the distance there
  bool degenerate_line = false;
  for (int i = 0; i < p->host.n_crows; ++i) degenerate_line = degenerate_line || p->host.row_type[i] == OKX_ROW_POINT_ON_LINE;
  a.confirm = (opts->confirm_full_pass != 0 || degenerate_line) ? 1 : 0;
  // Kernel choice (profiles/r01/config_sweep_v3.txt).  The packed kernel keeps more problems in
  // flight per CU (G lane groups x resident waves): measured 1.5x on saturating batches of
  // n <= 15 systems (MacPherson grid), no gain for n = 18 (DW corner), so auto = packed only
  // for n <= 15 and batches of at least 8 problems per resident slot.
  const long long single_slots = (long long)p->n_cu * p->blocks_per_cu;
  const long long packed_slots = (long long)p->n_cu * p->packed_blocks_per_cu * p->groups;
  bool use_quad = p->quad_fn_u != nullptr && (opts->kernel == 0 || opts->kernel == 3 || opts->kernel == 4);
  if (opts->kernel == 3 && !use_quad)
    return fail(OKX_ERR_INVALID, "quad kernel requested but not available: %s", p->quad_note);
  if (evaluated && !use_quad) return fail(OKX_ERR_INVALID, "evaluated solves run the generated kernels only (kernel = 0, 3 or 4)");
  const long long quad_slots = (long long)p->n_cu * p->quad_waves_per_cu * p->quad_ppw;
  // Lane kernel (one lane per problem, 64 per wavefront): auto selection from lane_min_problems on, when nothing the
  // quad kernel alone offers is asked for (fitted model, trace); kernel == 4 forces it.
  // (which body a launch needs is known once the chain length is: a body auto selection may not use sends the launch
  //  back to the quad kernel below)
  // Rounds of each kernel for this launch (one wavefront per SIMD either way): lane wave units hold 64 problems of ONE
  // geometry, so an ensemble with few steps per geometry leaves lanes idle and may be the quad kernel's after all.
  // The parallel unit is a CHAIN (a problem when chains have length 1): an explicit chain length - or chain = 1, the whole
  // span - is counted as such; chain_len = -1 (auto) sizes its chains to the kernel chosen here, from the problem count.
  const auto lane_pays = [&]() {
    const long long simds = (long long)p->n_cu * 4;
    const long long span0 = spg > 0 ? spg : n_problems, n_spans = n_problems / span0;
    long long len0 = opts->chain_len;
    if (len0 == 0) len0 = opts->chain ? span0 : 1;
    if (len0 < 1) len0 = 1;  // (auto)
    if (len0 > span0) len0 = span0;
    const long long chains_per_span = (span0 + len0 - 1) / len0;
    const long long lane_waves = n_spans * ((chains_per_span + 63) / 64);
    const long long quad_waves = (n_spans * chains_per_span + 15) / 16;
    const long long lane_rounds = (lane_waves + simds - 1) / simds, quad_rounds = (quad_waves + simds - 1) / simds;
    return 26 * lane_rounds < 19 * quad_rounds + 3;  // us per round (and chain step) of either kernel, measured on C2 / C4 shapes
  };
  bool use_lane = p->lane_fn_u != nullptr && use_quad && opts->predictor == 0 && (p->quad_trace == nullptr || okx::dev_switch("lane_timeline")) &&
                  (opts->kernel == 4 || (opts->kernel == 0 && n_problems >= p->lane_min_problems && lane_pays()));
  bool lane_auto_cold = false;  // chain_len = -1 resolved to independent solves on the lane kernel
  bool lane_nested = false;     // ... or to the nested start mode (okx_lane_nest_*: four steps per lane, 256 per wave unit)
  if (use_lane && (opts->kernel == 0 || opts->kernel == 4) && opts->chain_len == -1 && p->lane_nest[0] && !evaluated) {
    // "auto" on a sweep that fills wave units of 256 consecutive steps: the nested start mode - every step but a lane's
    // first starts from the interpolant of already solved neighbours (one full pass + the confirming evaluation)
    const long long span0 = spg > 0 ? spg : n_problems;
    lane_nested = span0 >= 256 && (span0 % 256 == 0 || span0 >= 2048);
  }
  bool lane_refined = false;    // ... or to the coarse-to-fine start (okx_lane_refc_* / okx_lane_refw_*)
  if (use_lane && !lane_nested && (opts->kernel == 0 || opts->kernel == 4) && opts->chain_len == -1 && p->lane_refine[0] && !evaluated &&
      opts->output != OKX_OUTPUT_NONE && opts->grad_tol == 0.0) {
    // every fourth step cold, the steps between from the cubic interpolant of those: spans of at least 64 coarse steps
    const long long span0 = spg > 0 ? spg : n_problems;
    lane_refined = span0 >= 256 && span0 % 4 == 0;
  }
  if (use_lane && opts->kernel == 0 && !lane_nested) {
    const long long span0 = spg > 0 ? spg : n_problems;
    long long len0 = opts->chain_len;
    if (len0 == 0) len0 = opts->chain ? span0 : 1;
    const bool cold_launch = len0 == 1 || span0 == 1;
    if (len0 == -1 && !p->lane_chain_ok && p->lane_cold_ok) {
      // "auto" may also mean independent solves: where the lane kernel's chain body spills but its independent-solve body
      // does not (the double wishbone), cold starts on the lane kernel beat the quad kernel's chains (measured on 4096
      // geometries x 256 steps: 0.50 ms against 0.71 ms)
      lane_auto_cold = true;
    } else if (cold_launch ? !p->lane_cold_ok : !p->lane_chain_ok) {
      use_lane = false;
    }
  }
  if (evaluated && use_lane) {
    // the lane form of the evaluated module has the independent-solve bodies only: chains go to the quad kernel
    const long long span0 = spg > 0 ? spg : n_problems;
    long long len0 = opts->chain_len;
    if (len0 == 0) len0 = opts->chain ? span0 : 1;
    const bool cold_launch = len0 == 1 || span0 == 1 || lane_auto_cold;
    if (!p->ev_lane_u || !cold_launch) {
      if (opts->kernel == 4) return fail(OKX_ERR_INVALID, "no evaluated lane kernel for this launch: %s", p->ev_lane_u ? "chains" : p->ev_note);
      use_lane = false;
      lane_auto_cold = false;
    }
  }
  if (opts->kernel == 4 && !use_lane)
    return fail(OKX_ERR_INVALID, "lane kernel requested but not available: %s", p->lane_note[0] ? p->lane_note : "predictor / trace in use");
  bool use_packed = false;
  if (p->packed_fn && !use_quad) {
    if (opts->kernel == 2) use_packed = true;
    else if (opts->kernel == 0) use_packed = p->nreg <= 15 && n_problems >= 8 * single_slots;
  }
  {
    const long long span = spg > 0 ? spg : n_problems;
    long long len = opts->chain_len;
    if (len == 0) len = opts->chain ? span : 1;
    if (len < 0) {  // auto: about one chain per resident problem slot, balanced inside a geometry
      const long long lane_slots = (long long)p->n_cu * 4 * 64;
      const long long slots = use_lane ? lane_slots : use_quad ? quad_slots : (use_packed ? packed_slots : single_slots);
      const long long ideal = (n_problems + slots - 1) / slots;
      if (ideal >= span) {
        len = span;
      } else {
        long long per_span = (span + ideal - 1) / ideal;
        // lane kernel: a wave unit is 64 chains of ONE span, so the chains of a span come in multiples of 64
        if (use_lane) per_span = (per_span + 63) / 64 * 64;
        if (per_span > span) per_span = span;
        len = (span + per_span - 1) / per_span;
      }
    }
    if (lane_auto_cold || (lane_refined && use_lane)) len = 1;
    if (lane_nested) len = okx::kLaneNestSteps;
    if (len < 1) len = 1;
    if (len > span) len = span;
    a.chain_len = len;
  }
  if (plan_only) {  // okx_plan_launch: what this launch would run, nothing launched
    plan_only[0] = use_lane ? 4 : use_quad ? 3 : use_packed ? 2 : 1;
    plan_only[1] = lane_nested ? -1 : (int32_t)(a.chain_len > 0x7fffffffll ? 0x7fffffffll : a.chain_len);
    return OKX_OK;
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
  if (use_quad) {
    okx::QuadArgs q;
    q.targets = a.targets;
    q.geom_pos = a.geom_pos;
    q.geom_row_param = a.geom_row_param;
    q.out_pos = a.out_pos;
    q.info = a.info;
    q.n_problems = a.n_problems;
    q.steps_per_geometry = a.steps_per_geometry;
    q.chain_len = a.chain_len;
    q.max_iter = a.max_iter;
    q.confirm = a.confirm;
    q.step_tol = a.step_tol;
    q.grad_tol = a.grad_tol;
    q.ftol = a.ftol;
    q.lambda0 = a.lambda0;
    q.residual_tolerance = a.residual_tolerance;
    const char* base = reinterpret_cast<const char*>(p->dev);
    q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
    q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
    q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
    q.trace = p->quad_trace;
    q.trace_problem = p->quad_trace_problem;
    q.predictor = (opts->predictor != 0 && !d_geom_pos) ? p->predictor_dev : nullptr;
    q.predictor_mode = opts->predictor;
    q.predictor_len = p->predictor_len;
    q.head = nullptr;
    q.out_mode = opts->output;
    if (p->quad_fn_head_u && opts->shared_first_step != 0 && opts->grad_tol == 0.0 && p->host.n_targets > 0) {
      // Shared first step: the design state's Jacobian, J^T J and damped factorisation are common to every problem
      // of a geometry, so they are evaluated once per geometry (one quad each) instead of once per chain head.
      okx::QuadHeadArgs h;
      h.geom_pos = d_geom_pos;
      h.geom_row_param = d_geom_row_param;
      h.lambda0 = opts->lambda0;
      h.design_pos = q.design_pos;
      h.row_param = q.row_param;
      h.dop_param = q.dop_param;
      void* hargs[] = {(void*)&h};
      if (!d_geom_pos) {
        double* table = nullptr;  // own geometry: once per lambda0 (the default's at program creation), then cached
        const int rc = own_head_table(p, opts->lambda0, (hipStream_t)stream, &table);
        if (rc != OKX_OK) return rc;
        q.head = table;  // (null: no table for this launch - see own_head_table)
      } else if (spg >= 4 && (n_problems / spg) * (long long)p->head_stride * 8 <= (256LL << 20)) {
        // (one table row costs about 1.3 passes of one quad: with fewer than four steps per geometry, or a table beyond
        //  256 MiB, the heads run their own first pass)
        const long long n_geom = n_problems / spg;
        bool have_scratch = n_geom <= p->head_geom_cap;
        if (!have_scratch && !stream_is_capturing((hipStream_t)stream)) {
          // grow-only scratch, replaced in stream order: launches of this program with geometry tables are stream-ordered
          // (okx.h), so the old table's readers are ahead of the free on this stream - no device-wide synchronisation.
          // Never inside a stream capture (the allocation would become a node of the graph while the pointer is cached
          // here): a captured launch that finds the scratch too small runs without the shared first step - warm the launch
          // up once outside the capture.
          if (p->head_geom_dev) {
            double* old = p->head_geom_dev;
            p->head_geom_dev = nullptr;
            p->head_geom_cap = 0;
            HIP_TRY(hipFreeAsync(old, (hipStream_t)stream));
          }
          const size_t bytes = sizeof(double) * (size_t)n_geom * (size_t)p->head_stride;
          double* fresh = nullptr;
          HIP_TRY(hipMallocAsync((void**)&fresh, bytes, (hipStream_t)stream));
          p->head_geom_dev = fresh;
          p->head_geom_cap = n_geom;
          have_scratch = true;
        }
        if (have_scratch) {
        h.head = p->head_geom_dev;
        h.n_geometries = n_geom;
        const long long head_waves = (n_geom + p->quad_ppw - 1) / p->quad_ppw;
        const long long head_cap = (long long)p->n_cu * p->quad_waves_per_cu;
        HIP_TRY(hipModuleLaunchKernel(p->quad_fn_head_g, (int)(head_waves < head_cap ? head_waves : head_cap), 1, 1, okx::kWave, 1, 1,
                                      0, (hipStream_t)stream, hargs, nullptr));
        q.head = p->head_geom_dev;
        }
      }
    }
    okx::QuadEvArgs qe;  // evaluated launches: the same arguments, then the epilogue's outputs and the roles' numbers
    qe.tan = d_tangents;
    qe.ev = d_eval;
    qe.cfg = p->ev_cfg;
    qe.cfg_r = p->ev_cfg_r;
    std::memcpy(qe.roles, p->ev_roles, sizeof(qe.roles));
    void* kargs[] = {evaluated ? (void*)&qe : (void*)&q};
    if (use_lane) {
      const long long chains_per_span = (span_ + a.chain_len - 1) / a.chain_len;
      const long long lane_units = (n_problems / span_) * ((chains_per_span + 63) / 64);
      const long long lane_cap = (long long)p->n_cu * 4;  // one wavefront per SIMD (512 registers, ~37 KB LDS)
      const int lane_grid = (int)(lane_units < lane_cap ? (lane_units < 1 ? 1 : lane_units) : lane_cap);
      hipFunction_t fn = a.chain_len == 1 ? (d_geom_pos ? p->lane_fn_g : p->lane_fn_u) : (d_geom_pos ? p->lane_chain_g : p->lane_chain_u);
      if (opts->output != OKX_OUTPUT_RECORDS) fn = p->lane_compact[(a.chain_len == 1 ? 0 : 2) + (d_geom_pos ? 1 : 0)];
      if (evaluated) fn = d_geom_pos ? p->ev_lane_g : p->ev_lane_u;
      void* ring = nullptr;
      if (lane_nested) {
        fn = p->lane_nest[(opts->output != OKX_OUTPUT_RECORDS ? 2 : 0) + (d_geom_pos ? 1 : 0)];
        HIP_TRY(hipMallocAsync(&ring, sizeof(double) * (size_t)okx::lane_nest_doubles(p->host.n) * (size_t)lane_grid, (hipStream_t)stream));
        q.predictor = static_cast<const double*>(ring);
      } else if (a.chain_len != 1 && okx::lane_chain_is_flat(p->host.n)) {
        // what a flat chain body carries from step to step (okx_quad.hpp lane_chain_is_flat): scratch of this launch,
        // allocated and freed in stream order (legal under stream capture, no device-wide synchronisation)
        HIP_TRY(hipMallocAsync(&ring, sizeof(double) * (size_t)okx::lane_flat_chain_doubles(p->host.n) * (size_t)lane_grid, (hipStream_t)stream));
        q.predictor = static_cast<const double*>(ring);
      }
      if (lane_refined && a.chain_len == 1) {
        // four launches on the stream: the coarse steps (offset 0), then offsets 1, 2, 3 from their interpolant
        const int variant = (opts->output != OKX_OUTPUT_RECORDS ? 2 : 0) + (d_geom_pos ? 1 : 0);
        for (int off = 0; off < 4; ++off) {
          q.chain_len = off;  // (the strided bodies read their offset here)
          const long long per_span = (span_ - off + 3) / 4;
          const long long units4 = (n_problems / span_) * ((per_span + 63) / 64);
          const int grid4 = (int)(units4 < lane_cap ? (units4 < 1 ? 1 : units4) : lane_cap);
          HIP_TRY(hipModuleLaunchKernel(p->lane_refine[(off ? 4 : 0) + variant], grid4, 1, 1, okx::kWave, 1, 1, 0, (hipStream_t)stream, kargs, nullptr));
        }
        return OKX_OK;
      }
      qe.q = q;
      HIP_TRY(hipModuleLaunchKernel(fn, lane_grid, 1, 1, okx::kWave, 1, 1, 0, (hipStream_t)stream, kargs, nullptr));
      if (ring) HIP_TRY(hipFreeAsync(ring, (hipStream_t)stream));
      return OKX_OK;
    }
    const long long wave_units = (units + p->quad_ppw - 1) / p->quad_ppw;
    const long long cap = (long long)p->n_cu * p->quad_waves_per_cu;
    const int grid = (int)(wave_units < cap ? (wave_units < 1 ? 1 : wave_units) : cap);
    hipFunction_t fn = d_geom_pos ? p->quad_fn_g : p->quad_fn_u;
    // independent solves from the own geometry's design state with its first-step table and nothing the general body
    // alone offers (fitted model, LM trace, gradient stop): the cold body
    // (not for programs with the reference's zero-gradient line row: their solves reject steps as a matter of course,
    //  which the cold body answers by starting over in its general loop - measured 4 % slower than the general body)
    if (p->quad_fn_cold_u && !degenerate_line && !d_geom_pos && a.chain_len == 1 && q.head != nullptr && q.predictor == nullptr && (q.trace == nullptr || okx::dev_switch("quad_timeline")) &&
        opts->grad_tol == 0.0 && !okx::dev_switch("no_cold"))
      fn = p->quad_fn_cold_u;
    if (evaluated) fn = fn == p->quad_fn_cold_u && p->ev_cold_u ? p->ev_cold_u : d_geom_pos ? p->ev_solve_g : p->ev_solve_u;
    qe.q = q;
    HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, okx::kWave, 1, 1, 0, (hipStream_t)stream, kargs, nullptr));
    return OKX_OK;
  }
  if (opts->output != OKX_OUTPUT_RECORDS)
    return fail(OKX_ERR_INVALID, "output mode %d needs a generated kernel (this launch runs the interpreter: %s)", opts->output,
                p->quad_note[0] ? p->quad_note : "kernel option");
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
  HIP_TRY(hipLaunchKernel(p->solve_fn, dim3(grid), dim3(p->threads), kargs, p->lds_bytes,
                          (hipStream_t)stream));
  return OKX_OK;
}

int32_t okx_solve_batch(okx_program* p, const okx_solve_opts* opts, int64_t n_problems,
                        const double* d_targets, const double* d_geom_pos,
                        const double* d_geom_row_param, double* d_out_pos, okx_info* d_info,
                        void* stream) {
  return solve_impl(p, opts, n_problems, d_targets, d_geom_pos, d_geom_row_param, d_out_pos, d_info, stream, false, nullptr, nullptr);
}

int32_t okx_solve_evaluated_batch(okx_program* p, const okx_solve_opts* opts, int64_t n_problems,
                                  const double* d_targets, const double* d_geom_pos, const double* d_geom_row_param,
                                  double* d_out_pos, okx_info* d_info, double* d_tangents, double* d_eval, void* stream) {
  return solve_impl(p, opts, n_problems, d_targets, d_geom_pos, d_geom_row_param, d_out_pos, d_info, stream, true, d_tangents, d_eval);
}

int32_t okx_plan_launch(okx_program* p, const okx_solve_opts* opts, int64_t n_problems, int32_t geometry_tables, int32_t evaluated,
                        int32_t* out2) {
  if (!out2) return fail(OKX_ERR_INVALID, "null output pointer");
  // the selection reads no batch array: placeholders stand for the pointers a launch of this shape would pass
  double* const some = reinterpret_cast<double*>(uintptr_t(64));
  out2[0] = out2[1] = 0;
  if (n_problems <= 0) return fail(OKX_ERR_INVALID, "a launch plan needs a positive problem count");
  return solve_impl(p, opts, n_problems, some, geometry_tables ? some : nullptr, geometry_tables ? some : nullptr, some,
                    reinterpret_cast<okx_info*>(some), nullptr, evaluated != 0, nullptr, evaluated ? some : nullptr, out2);
}

static void release_evaluation(okx_program* p) {
  if (p->ev_mod) (void)hipModuleUnload(p->ev_mod);
  if (p->ev_lane_mod) (void)hipModuleUnload(p->ev_lane_mod);
  p->ev_mod = p->ev_lane_mod = nullptr;
  p->ev_solve_u = p->ev_solve_g = p->ev_cold_u = p->ev_pos_u = p->ev_pos_g = nullptr;
  p->ev_lane_u = p->ev_lane_g = nullptr;
  p->ev_lane_pos_u = p->ev_lane_pos_g = nullptr;
}

int32_t okx_program_enable_evaluation(okx_program* p, const okx_corner_roles* roles) {
  if (!p || !roles) return fail(OKX_ERR_INVALID, "null program or roles");
  if (int32_t rc = check_corner_roles(roles, p->host.n_out, "roles")) return rc;
  attach_when_ready(p, true);  // the evaluated kernels share the solve kernels' first-step tables: those first
  okx::EvalSpec spec;
  std::string why;
  if (!okx::eval_spec_from_roles(p->host, *roles, &spec, &why)) return fail(OKX_ERR_INVALID, "%s", why.c_str());
  bool with_lane = false;
  {
    std::shared_lock<std::shared_mutex> readers(*p->kern_mutex);
    if (!p->quad_fn_u || p->quad_ppw != 16) {
      const std::string note = p->quad_note[0] ? p->quad_note : "a pair-mode program";
      readers.unlock();
      std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
      std::snprintf(p->ev_note, sizeof(p->ev_note), "no single-mode quad kernel (%.180s)", note.c_str());
      return fail(OKX_ERR_INVALID, "evaluated solves need the program's single-mode quad kernel: %s", p->ev_note);
    }
    if (p->ev_solve_u && std::memcmp(&spec, &p->ev_spec, sizeof(spec)) == 0) {
      // same role points as before: only the numbers change (launches read them as a kernel argument)
      readers.unlock();
      std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
      okx::eval_scalars_from_roles(*roles, &p->ev_cfg);
      return OKX_OK;
    }
    with_lane = p->lane_fn_u != nullptr && !okx::dev_switch("no_lane");
  }
  // generate + compile (or fetch from the cache) outside the lock: launches of this program go on meanwhile
  std::string code, lcode, lwhy;
  int lane_scratch = -1;
  if (!okx::quad_eval_build(p->host, spec, quad_waves_per_simd(), &code, &why)) {
    std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
    std::snprintf(p->ev_note, sizeof(p->ev_note), "%.250s", why.c_str());
    return fail(OKX_ERR_LIMIT, "no evaluated kernels for this program: %s", why.c_str());
  }
  const bool lane_built = with_lane && okx::lane_eval_build(p->host, spec, &lcode, &lwhy, false, &lane_scratch);
  std::unique_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (p->ev_mod) {
    HIP_TRY(hipDeviceSynchronize());  // launches in flight may still run the modules about to be replaced
    release_evaluation(p);
  }
  p->ev_note[0] = 0;
  hipModule_t mod = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
    (void)hipGetLastError();
    return fail(OKX_ERR_DEVICE, "hipModuleLoadData failed for the evaluated module");
  }
  hipFunction_t su = nullptr;
  if (hipModuleGetFunction(&su, mod, "okx_quad_evsolve_u") != hipSuccess ||
      hipModuleGetFunction(&p->ev_solve_g, mod, "okx_quad_evsolve_g") != hipSuccess ||
      hipModuleGetFunction(&p->ev_pos_u, mod, "okx_quad_evaluate_u") != hipSuccess ||
      hipModuleGetFunction(&p->ev_pos_g, mod, "okx_quad_evaluate_g") != hipSuccess) {
    (void)hipGetLastError();
    (void)hipModuleUnload(mod);
    p->ev_solve_g = p->ev_pos_u = p->ev_pos_g = nullptr;
    return fail(OKX_ERR_DEVICE, "kernel symbols missing in the evaluated module");
  }
  if (hipModuleGetFunction(&p->ev_cold_u, mod, "okx_quad_evcold_u") != hipSuccess) {
    (void)hipGetLastError();
    p->ev_cold_u = nullptr;
  }
  p->ev_mod = mod;
  p->ev_spec = spec;
  p->ev_axle = false;
  okx::eval_scalars_from_roles(*roles, &p->ev_cfg);
  // the lane form, for programs whose solves have one (failure only means the quad form serves every batch size)
  if (with_lane && !lane_built) {
    std::snprintf(p->ev_note, sizeof(p->ev_note), "no lane form: %.230s", lwhy.c_str());
  } else if (with_lane && lane_scratch > 512) {
    std::snprintf(p->ev_note, sizeof(p->ev_note), "the lane form spills %d B of scratch: not used", lane_scratch);
  } else if (with_lane) {
    hipModule_t lmod = nullptr;
    if (hipModuleLoadData(&lmod, lcode.data()) == hipSuccess &&
        hipModuleGetFunction(&p->ev_lane_u, lmod, "okx_lane_evsolve_u") == hipSuccess &&
        hipModuleGetFunction(&p->ev_lane_g, lmod, "okx_lane_evsolve_g") == hipSuccess) {
      p->ev_lane_mod = lmod;
      p->ev_lane_scratch = lane_scratch;
      // the same epilogue on given states (optional: programs whose free points are not all output points have none)
      if (hipModuleGetFunction(&p->ev_lane_pos_u, lmod, "okx_lane_evaluate_u") != hipSuccess ||
          hipModuleGetFunction(&p->ev_lane_pos_g, lmod, "okx_lane_evaluate_g") != hipSuccess) {
        (void)hipGetLastError();
        p->ev_lane_pos_u = p->ev_lane_pos_g = nullptr;
      }
    } else {
      (void)hipGetLastError();
      if (lmod) (void)hipModuleUnload(lmod);
      p->ev_lane_u = p->ev_lane_g = nullptr;
      p->ev_lane_pos_u = p->ev_lane_pos_g = nullptr;
      std::snprintf(p->ev_note, sizeof(p->ev_note), "the lane form's code object did not load");
    }
  }
  p->ev_solve_u = su;  // the gate of the evaluated launch paths
  return OKX_OK;
}

int32_t okx_program_evaluation(const okx_program* p) { return p && p->ev_solve_u ? 1 | (p->ev_lane_u ? 2 : 0) : 0; }
const char* okx_program_evaluation_note(const okx_program* p) { return p ? p->ev_note : ""; }
int32_t okx_program_eval_columns(const okx_program* p) {
  return p && p->ev_solve_u ? (p->ev_axle ? OKX_EVAL_AXLE_COLUMNS : OKX_EVAL_COLUMNS) : 0;
}

static int32_t check_axle_roles(const okx_axle_roles* roles, int32_t n_out) {
  if (int32_t rc = check_corner_roles(&roles->left, n_out, "left")) return rc;
  if (int32_t rc = check_corner_roles(&roles->right, n_out, "right")) return rc;
  if (roles->n_roles < 0 || roles->n_roles > OKX_MAX_ROTATIONS) return fail(OKX_ERR_INVALID, "n_roles must be 0 .. %d", OKX_MAX_ROTATIONS);
  return OKX_OK;
}

static void axle_numbers_from_roles(okx_program* p, const okx_axle_roles* roles) {
  okx::eval_scalars_from_roles(roles->left, &p->ev_cfg);
  okx::eval_scalars_from_roles(roles->right, &p->ev_cfg_r);
  std::memset(p->ev_roles, 0, sizeof(p->ev_roles));
  for (int k = 0; k < roles->n_roles; ++k) {
    const okx_rotation_role& r = roles->roles[k];
    okx::EvalRoleNum& n = p->ev_roles[k];
    for (int i = 0; i < 3; ++i) n.design[i] = r.design[i], n.axis_point[i] = r.axis_point[i], n.axis_dir[i] = r.axis_dir[i];
    n.scale = r.scale;
  }
}

/* okx_program_enable_evaluation for a composed axle (pair-mode program): see okx.h. */
int32_t okx_program_enable_axle_evaluation(okx_program* p, const okx_axle_roles* roles) {
  if (!p || !roles) return fail(OKX_ERR_INVALID, "null program or roles");
  if (int32_t rc = check_axle_roles(roles, p->host.n_out)) return rc;
  attach_when_ready(p, true);
  okx::AxleEvalSpec spec;
  std::memset(&spec, 0, sizeof(spec));
  std::string why;
  if (!okx::axle_eval_spec_from_roles(p->host, *roles, &spec, &why)) return fail(OKX_ERR_INVALID, "%s", why.c_str());
  {
    std::shared_lock<std::shared_mutex> readers(*p->kern_mutex);
    if (!p->quad_fn_u || p->quad_ppw != 8) {
      const std::string note = p->quad_note[0] ? p->quad_note : "a single-mode program (use okx_program_enable_evaluation)";
      readers.unlock();
      std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
      std::snprintf(p->ev_note, sizeof(p->ev_note), "no pair-mode quad kernel (%.180s)", note.c_str());
      return fail(OKX_ERR_INVALID, "an axle's evaluated solves need the program's pair-mode quad kernel: %s", p->ev_note);
    }
    if (p->ev_solve_u && p->ev_axle && std::memcmp(&spec, &p->ev_axle_spec, sizeof(spec)) == 0) {
      readers.unlock();
      std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
      axle_numbers_from_roles(p, roles);  // same points: only the numbers change (launches read them as kernel arguments)
      return OKX_OK;
    }
  }
  std::string code;
  if (!okx::quad_axle_eval_build(p->host, spec, quad_waves_per_simd(), &code, &why)) {
    std::unique_lock<std::shared_mutex> writer(*p->kern_mutex);
    std::snprintf(p->ev_note, sizeof(p->ev_note), "%.250s", why.c_str());
    return fail(OKX_ERR_LIMIT, "no evaluated kernels for this program: %s", why.c_str());
  }
  std::unique_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (p->ev_mod) {
    HIP_TRY(hipDeviceSynchronize());  // launches in flight may still run the modules about to be replaced
    release_evaluation(p);
  }
  p->ev_note[0] = 0;
  hipModule_t mod = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
    (void)hipGetLastError();
    return fail(OKX_ERR_DEVICE, "hipModuleLoadData failed for the evaluated module");
  }
  hipFunction_t su = nullptr;
  if (hipModuleGetFunction(&su, mod, "okx_quad_evsolve_u") != hipSuccess ||
      hipModuleGetFunction(&p->ev_solve_g, mod, "okx_quad_evsolve_g") != hipSuccess ||
      hipModuleGetFunction(&p->ev_pos_u, mod, "okx_quad_evaluate_u") != hipSuccess ||
      hipModuleGetFunction(&p->ev_pos_g, mod, "okx_quad_evaluate_g") != hipSuccess) {
    (void)hipGetLastError();
    (void)hipModuleUnload(mod);
    p->ev_solve_g = p->ev_pos_u = p->ev_pos_g = nullptr;
    return fail(OKX_ERR_DEVICE, "kernel symbols missing in the evaluated module");
  }
  if (hipModuleGetFunction(&p->ev_cold_u, mod, "okx_quad_evcold_u") != hipSuccess) {
    (void)hipGetLastError();
    p->ev_cold_u = nullptr;
  }
  p->ev_mod = mod;
  p->ev_axle = true;
  p->ev_axle_spec = spec;
  axle_numbers_from_roles(p, roles);
  p->ev_solve_u = su;  // the gate of the evaluated launch paths
  return OKX_OK;
}

int32_t okx_precompile_axle_evaluation(const okx_program_desc* desc, const okx_axle_roles* roles) {
  if (!desc || !roles) return fail(OKX_ERR_INVALID, "null pointer");
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  if (rc == OKX_OK) rc = check_axle_roles(roles, tmp->n_out);
  okx::AxleEvalSpec spec;
  std::memset(&spec, 0, sizeof(spec));
  std::string why, code;
  if (rc == OKX_OK && !okx::axle_eval_spec_from_roles(*tmp, *roles, &spec, &why)) rc = fail(OKX_ERR_INVALID, "%s", why.c_str());
  if (rc == OKX_OK && !okx::quad_axle_eval_build(*tmp, spec, quad_waves_per_simd(), &code, &why))
    rc = fail(why.compare(0, 14, "compile failed") == 0 ? OKX_ERR_DEVICE : OKX_ERR_LIMIT, "no evaluated kernels for this program: %s", why.c_str());
  delete tmp;
  return rc;
}

int32_t okx_evaluate_batch(okx_program* p, int64_t n_problems, int64_t steps_per_geometry, const double* d_pos,
                           const double* d_geom_pos, const double* d_geom_row_param, double* d_tangents, double* d_eval,
                           void* stream) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (!p->ev_solve_u) return fail(OKX_ERR_INVALID, "okx_evaluate_batch needs okx_program_enable_evaluation first%s%s", p->ev_note[0] ? ": " : "", p->ev_note);
  if (n_problems < 0) return fail(OKX_ERR_INVALID, "negative problem count");
  if (n_problems == 0) return OKX_OK;
  if (!d_pos || (!d_tangents && !d_eval)) return fail(OKX_ERR_INVALID, "null pointer");
  if ((d_geom_pos == nullptr) != (d_geom_row_param == nullptr))
    return fail(OKX_ERR_INVALID, "geometry positions and row parameters must be given together");
  if (steps_per_geometry < 0 || (d_geom_pos && steps_per_geometry == 0) ||
      (steps_per_geometry > 0 && n_problems % steps_per_geometry != 0))
    return fail(OKX_ERR_INVALID, "bad steps_per_geometry");
  const char* base = reinterpret_cast<const char*>(p->dev);
  {
    // Lane form (one lane per state, 64 per wavefront: a third of the quad form's instructions per state) once the batch
    // gives every SIMD a wave unit; an ensemble's wave units hold states of ONE geometry, so few steps per geometry leave
    // lanes idle and stay with the quad form.
    const long long span = steps_per_geometry > 0 ? steps_per_geometry : n_problems;
    const long long units = (n_problems / span) * ((span + 63) / 64);
    const bool fills = units >= (long long)p->n_cu * 4 && n_problems >= 48 * units;
    if (p->ev_lane_pos_u && !okx::dev_switch("evaluate_quad") && (fills || okx::dev_switch("evaluate_lane"))) {
      okx::QuadEvArgs qe{};
      okx::QuadArgs& a = qe.q;
      a.targets = d_pos;  // (the GIVEN bodies read the records through this pointer: okx_lanegen.cpp)
      a.geom_pos = d_geom_pos;
      a.geom_row_param = d_geom_row_param;
      a.out_pos = nullptr;
      a.info = nullptr;
      a.n_problems = n_problems;
      a.steps_per_geometry = steps_per_geometry;
      a.chain_len = 1;
      a.max_iter = 0;
      a.confirm = 0;
      a.step_tol = a.ftol = a.lambda0 = a.residual_tolerance = 0.0;
      a.grad_tol = 0.0;
      a.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
      a.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
      a.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
      a.trace = nullptr;
      a.trace_problem = 0;
      a.predictor = nullptr;
      a.predictor_mode = 0;
      a.predictor_len = 0;
      a.head = nullptr;
      a.out_mode = OKX_OUTPUT_NONE;
      qe.tan = d_tangents;
      qe.ev = d_eval;
      qe.cfg = p->ev_cfg;
      const long long cap = (long long)p->n_cu * 4;
      void* kargs[] = {(void*)&qe};
      HIP_TRY(hipModuleLaunchKernel(d_geom_pos ? p->ev_lane_pos_g : p->ev_lane_pos_u, (int)(units < cap ? units : cap), 1, 1, okx::kWave, 1, 1,
                                    0, (hipStream_t)stream, kargs, nullptr));
      return OKX_OK;
    }
  }
  okx::QuadEvPosArgs q;
  q.pos = d_pos;
  q.geom_pos = d_geom_pos;
  q.geom_row_param = d_geom_row_param;
  q.tan = d_tangents;
  q.ev = d_eval;
  q.n_problems = n_problems;
  q.steps_per_geometry = steps_per_geometry > 0 ? steps_per_geometry : n_problems;
  q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
  q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
  q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
  q.cfg = p->ev_cfg;
  q.cfg_r = p->ev_cfg_r;
  std::memcpy(q.roles, p->ev_roles, sizeof(q.roles));
  const long long waves = (n_problems + p->quad_ppw - 1) / p->quad_ppw;
  // (a streaming launch: several rounds' worth of workgroups - but an axle's own-geometry body keeps its chain constants and
  //  fixed points across a persistent loop: one wavefront per SIMD)
  const long long cap = (long long)p->n_cu * p->quad_waves_per_cu * (p->ev_axle && !d_geom_pos ? 1 : 8);
  void* kargs[] = {(void*)&q};
  HIP_TRY(hipModuleLaunchKernel(d_geom_pos ? p->ev_pos_g : p->ev_pos_u, (int)(waves < cap ? waves : cap), 1, 1, okx::kWave, 1, 1, 0,
                                (hipStream_t)stream, kargs, nullptr));
  return OKX_OK;
}

int32_t okx_precompile_evaluation(const okx_program_desc* desc, const okx_corner_roles* roles) {
  if (!desc || !roles) return fail(OKX_ERR_INVALID, "null pointer");
  okx::DevProgram* tmp = new (std::nothrow) okx::DevProgram;
  if (!tmp) return fail(OKX_ERR_ALLOC, "out of host memory");
  int rc = okx::build_dev_program(desc, tmp, g_err, (int)sizeof(g_err));
  okx::EvalSpec spec;
  std::string why, code;
  if (rc == OKX_OK && !okx::eval_spec_from_roles(*tmp, *roles, &spec, &why)) rc = fail(OKX_ERR_INVALID, "%s", why.c_str());
  if (rc == OKX_OK && !okx::quad_eval_build(*tmp, spec, quad_waves_per_simd(), &code, &why))
    rc = fail(why.compare(0, 14, "compile failed") == 0 ? OKX_ERR_DEVICE : OKX_ERR_LIMIT, "no evaluated kernels for this program: %s", why.c_str());
  if (rc == OKX_OK && tmp->n_free <= okx::kLaneMaxFree) {
    std::string lcode, lwhy;
    (void)okx::lane_eval_build(*tmp, spec, &lcode, &lwhy);  // (programs / variants it does not fit simply have no lane form)
  }
  delete tmp;
  return rc;
}

int32_t okx_program_has_predictor(const okx_program* p) { return p && p->predictor_dev ? 1 : 0; }

/* Fits the chain-head predictor of a program with a quad kernel over the target box [lo, hi] (absolute
   target values, host arrays of n_targets; lo[t] == hi[t]: that target is held): solves the program at the
   (degree + 1)^d Chebyshev nodes of the d varying targets (one cold-start launch, synchronous), takes the
   tensor Chebyshev coefficients of every free coordinate by discrete orthogonality and keeps the terms up to
   total degree `degree` (<= 0: 7).  Launches with opts.predictor != 0 on the program's own geometry start
   every chain head at that polynomial (targets clamped to the box) instead of the design state.  Refitting
   replaces the previous model. */
int32_t okx_program_fit_predictor(okx_program* p, const double* lo, const double* hi, int32_t degree, void* stream) {
  if (!p || !lo || !hi) return fail(OKX_ERR_INVALID, "null pointer");
  attach_when_ready(p, false);
  {
    std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);  // (released before the node solve takes it itself)
    if (!p->quad_fn_u) return fail(OKX_ERR_INVALID, "the predictor belongs to the quad kernel: %s", p->quad_note);
    if (p->quad_ppw != 16) return fail(OKX_ERR_INVALID, "pair-mode kernels carry no predictor (register-bound)");
  }
  const okx::DevProgram& H = p->host;
  const int T = H.n_targets;
  if (T < 1) return fail(OKX_ERR_INVALID, "program has no targets");
  const int D = degree <= 0 ? okx::kPredictorDegree : degree;
  if (D > okx::kPredictorMaxDegree) return fail(OKX_ERR_INVALID, "degree must be <= %d", okx::kPredictorMaxDegree);
  std::vector<int> out_of(H.n_points, -1);
  for (int k = 0; k < H.n_out; ++k) out_of[H.out_point[k]] = k;
  for (int f = 0; f < H.n_free; ++f)
    if (out_of[H.free_point[f]] < 0) return fail(OKX_ERR_INVALID, "free point %d is not an output point", H.free_point[f]);
  std::vector<double> mid(T), half(T);
  std::vector<int> deg(T), vary;
  for (int t = 0; t < T; ++t) {
    if (!(hi[t] >= lo[t])) return fail(OKX_ERR_INVALID, "target %d: hi < lo", t);
    mid[t] = 0.5 * (lo[t] + hi[t]);
    half[t] = 0.5 * (hi[t] - lo[t]);
    deg[t] = half[t] > 1e-9 ? D : 0;
    if (deg[t]) vary.push_back(t);
  }
  const int d = (int)vary.size(), N = D + 1;
  long long S = 1;
  for (int k = 0; k < d; ++k) {
    S *= N;
    if (S > 32768) return fail(OKX_ERR_LIMIT, "%d varying targets at degree %d need too many nodes", d, D);
  }
  std::vector<double> node(N);
  for (int a = 0; a < N; ++a) node[a] = cos(3.14159265358979323846 * (a + 0.5) / N);
  // node k <-> digits a_0 .. a_{d-1} (dimension 0 slowest)
  auto digit = [&](long long k, int j) {
    for (int q = d - 1; q > j; --q) k /= N;
    return (int)(k % N);
  };
  std::vector<double> targets((size_t)S * T);
  for (long long k = 0; k < S; ++k) {
    for (int t = 0; t < T; ++t) targets[(size_t)k * T + t] = mid[t];
    for (int j = 0; j < d; ++j) targets[(size_t)k * T + vary[j]] = mid[vary[j]] + half[vary[j]] * node[digit(k, j)];
  }
  // one scratch allocation for the node solve: targets | positions | info records
  const size_t out_doubles = (size_t)S * H.n_out * 3;
  const size_t bytes_t = targets.size() * sizeof(double), bytes_out = out_doubles * sizeof(double);
  const size_t bytes_info = (size_t)S * sizeof(okx_info);
  char* d_scratch = nullptr;
  HIP_TRY(hipMalloc((void**)&d_scratch, bytes_t + bytes_out + bytes_info));
  double* d_t = reinterpret_cast<double*>(d_scratch);
  double* d_out = reinterpret_cast<double*>(d_scratch + bytes_t);
  okx_info* d_info = reinterpret_cast<okx_info*>(d_scratch + bytes_t + bytes_out);
  std::vector<double> out(out_doubles);
  std::vector<okx_info> info((size_t)S);
  int32_t rc = OKX_OK;
  {
    okx_solve_opts o;
    okx_default_opts(&o);
    o.chain_len = 1;
    o.kernel = 3;
    o.confirm_full_pass = 1;  // end on a computed correction: the fit wants every digit
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(d_t, targets.data(), bytes_t, hipMemcpyHostToDevice, st) != hipSuccess)
      rc = fail(OKX_ERR_DEVICE, "copy failed");
    if (rc == OKX_OK) rc = okx_solve_batch(p, &o, S, d_t, nullptr, nullptr, d_out, d_info, stream);
    if (rc == OKX_OK &&
        (hipMemcpyAsync(out.data(), d_out, bytes_out, hipMemcpyDeviceToHost, st) != hipSuccess ||
         hipMemcpyAsync(info.data(), d_info, bytes_info, hipMemcpyDeviceToHost, st) != hipSuccess ||
         hipStreamSynchronize(st) != hipSuccess))
      rc = fail(OKX_ERR_DEVICE, "node solve failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(d_scratch);
  if (rc != OKX_OK) return rc;
  for (long long k = 0; k < S; ++k)
    if (!(info[k].flags & OKX_INFO_CONVERGED) || (info[k].flags & (OKX_INFO_FAILED | OKX_INFO_RESIDUAL_EXCEEDED)))
      return fail(OKX_ERR_INVALID, "node %lld of the target box did not converge", k);
  // Chebyshev values at the nodes
  std::vector<double> cheb((size_t)N * N);  // [i][a] = T_i(node_a)
  for (int a = 0; a < N; ++a) {
    cheb[a] = 1.0;
    if (N > 1) cheb[(size_t)N + a] = node[a];
    for (int i = 2; i < N; ++i) cheb[(size_t)i * N + a] = 2.0 * node[a] * cheb[(size_t)(i - 1) * N + a] - cheb[(size_t)(i - 2) * N + a];
  }
  // Table in the kernel's layout (okx_quadgen.cpp): per target (mid, 1 / half-range, degree), the
  // total-degree cap, the table length, then one [free][4] block per term in the kernel's loop order:
  // target 0 outermost, index i_t <= degree_t and <= the remaining total degree.
  const int nv = H.n_free * 4;
  std::vector<double> table((size_t)3 * T + 2, 0.0);
  for (int t = 0; t < T; ++t) {
    table[3 * t] = mid[t];
    table[3 * t + 1] = deg[t] ? 1.0 / half[t] : 0.0;
    table[3 * t + 2] = (double)deg[t];
  }
  table[3 * T] = (double)D;
  std::vector<int> idx(T, 0), slot_of(T, -1);
  for (int j = 0; j < d; ++j) slot_of[vary[j]] = j;
  std::vector<double> block(nv);
  std::function<void(int, int)> walk = [&](int t, int budget) {
    if (t == T) {
      std::fill(block.begin(), block.end(), 0.0);
      double scale = 1.0;
      for (int j = 0; j < d; ++j) scale *= (idx[vary[j]] == 0 ? 1.0 : 2.0) / N;
      for (long long k = 0; k < S; ++k) {
        double w = scale;
        for (int j = 0; j < d; ++j) w *= cheb[(size_t)idx[vary[j]] * N + digit(k, j)];
        for (int f = 0; f < H.n_free; ++f) {
          const int o = out_of[H.free_point[f]];
          for (int c = 0; c < 3; ++c) block[f * 4 + c] += w * out[((size_t)k * H.n_out + o) * 3 + c];
        }
      }
      table.insert(table.end(), block.begin(), block.end());
      return;
    }
    for (int i = 0; i <= deg[t] && i <= budget; ++i) {
      idx[t] = i;
      walk(t + 1, budget - i);
    }
    idx[t] = 0;
  };
  walk(0, D);
  table[3 * T + 1] = (double)table.size();
  double* d_table = nullptr;
  HIP_TRY(hipMalloc(&d_table, table.size() * sizeof(double)));
  if (hipMemcpy(d_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d_table);
    return fail(OKX_ERR_DEVICE, "copy failed");
  }
  if (p->predictor_dev) {
    HIP_TRY(hipDeviceSynchronize());  // launches in flight may still read the old table
    (void)hipFree(p->predictor_dev);
  }
  p->predictor_dev = d_table;
  p->predictor_len = (long long)table.size();
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
  const okx::DevProgram* dev = p->dev;
  void* kargs[] = {(void*)&dev, (void*)&a};
  HIP_TRY(hipLaunchKernel(p->eval_fn, dim3(grid_for(p, n_problems)), dim3(p->threads), kargs, p->lds_bytes,
                          (hipStream_t)stream));
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
  const okx::DevProgram* dev = p->dev;
  void* kargs[] = {(void*)&dev, (void*)&a};
  HIP_TRY(hipLaunchKernel(p->eval_fn, dim3(grid_for(p, n_problems)), dim3(p->threads), kargs, p->lds_bytes,
                          (hipStream_t)stream));
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

int32_t okx_expand_positions_batch(okx_program* p, int64_t n_problems, int64_t steps_per_geometry, const double* d_free,
                                   const double* d_geom_pos, double* d_out_pos, void* stream) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  {
    const hipStream_t launch_stream = (hipStream_t)stream;
    attach_when_ready(p, false, &launch_stream);
  }
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (n_problems < 0) return fail(OKX_ERR_INVALID, "negative problem count");
  if (n_problems == 0) return OKX_OK;
  if (!d_free || !d_out_pos) return fail(OKX_ERR_INVALID, "null pointer");
  if (steps_per_geometry < 0 || (d_geom_pos && steps_per_geometry == 0) ||
      (steps_per_geometry > 0 && n_problems % steps_per_geometry != 0))
    return fail(OKX_ERR_INVALID, "bad steps_per_geometry");
  if (p->quad_fn_expand) {  // generated form: 16 states per wavefront, coalesced records
    okx::QuadExpandArgs q;
    q.free = d_free;
    q.geom_pos = d_geom_pos;
    q.out_pos = d_out_pos;
    q.n_problems = n_problems;
    q.steps_per_geometry = steps_per_geometry > 0 ? steps_per_geometry : n_problems;
    const char* base = reinterpret_cast<const char*>(p->dev);
    q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
    q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
    q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
    const long long waves = (n_problems + p->quad_ppw - 1) / p->quad_ppw;
    // (pair mode: a persistent grid - the wavefront reads its fixed points once and walks its wave units)
    const long long grid_cap = p->quad_ppw == 8 ? (long long)p->n_cu * 32 : 65536;
    void* kargs[] = {(void*)&q};
    HIP_TRY(hipModuleLaunchKernel(p->quad_fn_expand, (int)(waves < grid_cap ? waves : grid_cap), 1, 1, okx::kWave, 1, 1, 0,
                                  (hipStream_t)stream, kargs, nullptr));
    return OKX_OK;
  }
  okx::ExpandArgs a;
  a.free = d_free;
  a.geom_pos = d_geom_pos;
  a.out_pos = d_out_pos;
  a.n_problems = n_problems;
  a.steps_per_geometry = steps_per_geometry > 0 ? steps_per_geometry : n_problems;
  const long long blocks = (n_problems + okx::kExpandThreads - 1) / okx::kExpandThreads;
  if (blocks > 0x7fffffffll) return fail(OKX_ERR_INVALID, "too many problems for one launch");
  // derived positions [3 n_derived][64] and the point -> source table
  const size_t expand_lds = sizeof(double) * 3 * p->host.n_derived * okx::kExpandThreads + sizeof(int) * p->host.n_points;
  hipLaunchKernelGGL(okx::okx_expand_kernel, dim3((unsigned)blocks), dim3(okx::kExpandThreads), expand_lds, (hipStream_t)stream,
                     (const okx::DevProgram*)p->dev, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

int32_t okx_tangent_batch(okx_program* p, int64_t n_problems, int64_t steps_per_geometry, const double* d_pos,
                          const double* d_geom_pos, const double* d_geom_row_param, double* d_tangents,
                          okx_tangent_info* d_tinfo, void* stream) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  {
    const hipStream_t launch_stream = (hipStream_t)stream;
    attach_when_ready(p, false, &launch_stream);
  }
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (n_problems < 0) return fail(OKX_ERR_INVALID, "negative problem count");
  if (n_problems == 0) return OKX_OK;
  if (!d_pos || !d_tangents || !d_tinfo) return fail(OKX_ERR_INVALID, "null pointer");
  if ((d_geom_pos == nullptr) != (d_geom_row_param == nullptr))
    return fail(OKX_ERR_INVALID, "geometry positions and row parameters must be given together");
  if (steps_per_geometry < 0 || (d_geom_pos && steps_per_geometry == 0) ||
      (steps_per_geometry > 0 && n_problems % steps_per_geometry != 0))
    return fail(OKX_ERR_INVALID, "bad steps_per_geometry");
  if (p->host.n_targets == 0) return OKX_OK;
  if (!p->quad_fn_tan_u || okx::dev_switch("tangent_generic")) {  // (tests: the interpreter's tangent kernel on a program with a generated one)
    // generic interpreter form: one wavefront per state (programs without a quad kernel)
    okx::TangentArgs t;
    t.pos = d_pos;
    t.geom_pos = d_geom_pos;
    t.geom_row_param = d_geom_row_param;
    t.tan = d_tangents;
    t.tinfo = d_tinfo;
    t.n_problems = n_problems;
    t.steps_per_geometry = steps_per_geometry;
    for (int k = 0; k < p->host.n_free; ++k) {
      t.free_out[k] = -1;
      for (int o = 0; o < p->host.n_out; ++o)
        if (p->host.out_point[o] == p->host.free_point[k]) t.free_out[k] = o;
      if (t.free_out[k] < 0)
        return fail(OKX_ERR_INVALID, "tangents need every free point among the output points (point %d is not)",
                    p->host.free_point[k]);
    }
    const size_t lds = p->lds_bytes + sizeof(double) * 3 * (size_t)p->host.n_points;
    if (lds > 160 * 1024) return fail(OKX_ERR_LIMIT, "tangent kernel needs %zu bytes of LDS", lds);
    const okx::DevProgram* dev = p->dev;
    void* kargs[] = {(void*)&dev, (void*)&t};
    HIP_TRY(hipLaunchKernel(p->tangent_fn, dim3(grid_for(p, n_problems)), dim3(p->threads), kargs, lds,
                            (hipStream_t)stream));
    return OKX_OK;
  }
  okx::QuadTanArgs q;
  q.pos = d_pos;
  q.geom_pos = d_geom_pos;
  q.geom_row_param = d_geom_row_param;
  q.tan = d_tangents;
  q.tinfo = d_tinfo;
  q.n_problems = n_problems;
  q.steps_per_geometry = steps_per_geometry;
  const char* base = reinterpret_cast<const char*>(p->dev);
  q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
  q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
  q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
  const long long waves = (n_problems + p->quad_ppw - 1) / p->quad_ppw;
  const long long cap = (long long)p->n_cu * p->quad_waves_per_cu;
  void* kargs[] = {(void*)&q};
  HIP_TRY(hipModuleLaunchKernel(d_geom_pos ? p->quad_fn_tan_g : p->quad_fn_tan_u, (int)(waves < cap ? waves : cap), 1, 1,
                                okx::kWave, 1, 1, 0, (hipStream_t)stream, kargs, nullptr));
  return OKX_OK;
}

static_assert(sizeof(okx_corner_roles) == 128, "okx_corner_roles layout is part of the ABI (ctypes mirror in metrics.py)");

static int32_t check_corner_roles(const okx_corner_roles* roles, int32_t n_out, const char* who) {
  const int32_t idx[6] = {roles->wheel_center, roles->contact_patch, roles->axle_inboard,
                          roles->axle_outboard, roles->steer_lower, roles->steer_upper};
  for (int k = 0; k < 6; ++k)
    if (idx[k] < 0 || idx[k] >= n_out) return fail(OKX_ERR_INVALID, "%s: role %d is not an output point", who, k);
  if (!(roles->side_sign == 1.0 || roles->side_sign == -1.0))
    return fail(OKX_ERR_INVALID, "%s: side_sign must be +-1", who);
  const int n_axis = roles->instant_axis_kind == OKX_IA_TWO_PLANES ? 6
                     : roles->instant_axis_kind == OKX_IA_PLANE_AND_STRUT ? 4
                     : roles->instant_axis_kind == OKX_IA_NONE ? 0 : -1;
  if (n_axis < 0) return fail(OKX_ERR_INVALID, "%s: unknown instant_axis_kind %d", who, roles->instant_axis_kind);
  for (int k = 0; k < n_axis; ++k)
    if (roles->instant_axis_point[k] < 0 || roles->instant_axis_point[k] >= n_out)
      return fail(OKX_ERR_INVALID, "%s: instant-axis point %d is not an output point", who, k);
  if ((roles->damper_top < 0) != (roles->damper_bottom < 0) || roles->damper_top >= n_out || roles->damper_bottom >= n_out)
    return fail(OKX_ERR_INVALID, "%s: damper points must both be output points or both be -1", who);
  if (roles->rack_attachment >= n_out) return fail(OKX_ERR_INVALID, "%s: rack attachment is not an output point", who);
  for (int32_t v : {roles->axle_position, roles->driven_axle})
    if (v != OKX_AXLE_UNSET && v != OKX_AXLE_FRONT && v != OKX_AXLE_REAR)
      return fail(OKX_ERR_INVALID, "%s: axle_position / driven_axle must be OKX_AXLE_*", who);
  return OKX_OK;
}

int32_t okx_axis_rotation_batch(const okx_rotation_role* roles, int32_t n_roles, int64_t n_states, int32_t n_out,
                                int32_t n_targets, const double* d_pos, const double* d_tangents, double* d_angles,
                                double* d_dangles, void* stream) {
  if (!roles || !d_pos || !d_angles) return fail(OKX_ERR_INVALID, "null pointer");
  if (n_roles < 1 || n_roles > OKX_MAX_ROTATIONS) return fail(OKX_ERR_INVALID, "1..%d rotations per call", OKX_MAX_ROTATIONS);
  if (n_states < 0 || n_out <= 0 || n_targets < 0) return fail(OKX_ERR_INVALID, "bad dimension");
  if ((d_tangents == nullptr) != (d_dangles == nullptr))
    return fail(OKX_ERR_INVALID, "tangents and derivative output must be given together");
  okx::RotationArgs a;
  for (int k = 0; k < n_roles; ++k) {
    if (roles[k].point < 0 || roles[k].point >= n_out) return fail(OKX_ERR_INVALID, "rotation %d: not an output point", k);
    if (roles[k].kind < OKX_ROLE_AXIS_ROTATION || roles[k].kind > OKX_ROLE_MIDPOINT_COORDINATE)
      return fail(OKX_ERR_INVALID, "rotation %d: unknown kind %d", k, roles[k].kind);
    if (roles[k].kind != OKX_ROLE_AXIS_ROTATION && (roles[k].point_b < 0 || roles[k].point_b >= n_out))
      return fail(OKX_ERR_INVALID, "rotation %d: second point is not an output point", k);
    const double* d = roles[k].axis_dir;
    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(std::fabs(len - 1.0) <= 1e-9)) return fail(OKX_ERR_INVALID, "rotation %d: axis_dir must be a unit vector", k);
    a.roles[k] = roles[k];
  }
  if (n_states == 0) return OKX_OK;
  a.n_roles = n_roles;
  a.pos = d_pos;
  a.tan = d_tangents;
  a.angles = d_angles;
  a.dangles = d_dangles;
  a.n_states = n_states;
  a.n_out = n_out;
  a.n_targets = n_targets;
  const long long blocks = (n_states + 255) / 256;
  hipLaunchKernelGGL(okx::okx_axis_rotation_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

int32_t okx_axle_metrics_batch(const okx_corner_roles* left, const okx_corner_roles* right, int64_t n_states,
                               int32_t n_out, const double* d_pos, double* d_metrics, void* stream) {
  if (!left || !right || !d_pos || !d_metrics) return fail(OKX_ERR_INVALID, "null pointer");
  if (n_states < 0 || n_out <= 0) return fail(OKX_ERR_INVALID, "bad dimension");
  if (int32_t rc = check_corner_roles(left, n_out, "left")) return rc;
  if (int32_t rc = check_corner_roles(right, n_out, "right")) return rc;
  if (n_states == 0) return OKX_OK;
  okx::AxleMetricsArgs a;
  a.left = *left;
  a.right = *right;
  a.pos = d_pos;
  a.metrics = d_metrics;
  a.n_states = n_states;
  a.n_out = n_out;
  const long long blocks = (n_states + 255) / 256;
  hipLaunchKernelGGL(okx::okx_axle_metrics_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

int32_t okx_corner_metrics_batch(const okx_corner_roles* roles, int64_t n_states, int32_t n_out, int32_t n_targets,
                                 const double* d_pos, const double* d_tangents, double* d_metrics,
                                 double* d_dmetrics, void* stream) {
  if (!roles || !d_pos || !d_metrics) return fail(OKX_ERR_INVALID, "null pointer");
  if (n_states < 0 || n_out <= 0 || n_targets < 0) return fail(OKX_ERR_INVALID, "bad dimension");
  if ((d_tangents == nullptr) != (d_dmetrics == nullptr))
    return fail(OKX_ERR_INVALID, "tangents and derivative output must be given together");
  if (int32_t rc = check_corner_roles(roles, n_out, "roles")) return rc;
  if (n_states == 0) return OKX_OK;
  okx::MetricsArgs a;
  a.roles = *roles;
  a.pos = d_pos;
  a.tan = d_tangents;
  a.metrics = d_metrics;
  a.dmetrics = d_dmetrics;
  a.n_states = n_states;
  a.n_out = n_out;
  a.n_targets = n_targets;
  // Records of up to 21 points (corners): 64 states per wavefront, records, tangent rows and results staged through LDS
  // (okx_corner_metrics_tiled); longer records (the corners of a composed axle): one thread per state.  The same
  // formulas either way: same bits.
  okx::TileArgs ta;
  ta.rec = 3u * (uint32_t)n_out;
  ta.rec_inv = (uint32_t)((1ull << 32) / ta.rec) + 1u;
  ta.stride = ta.rec | 1u;
  const size_t lds_bytes = sizeof(double) * okx::kTileStates * (ta.stride > OKX_METRIC_COUNT ? ta.stride : OKX_METRIC_COUNT);
  const long long tiles = (n_states + okx::kTileStates - 1) / okx::kTileStates;
  if (ta.rec <= 63 && tiles < 0x7fffffffll) {
    if (d_tangents)
      hipLaunchKernelGGL(okx::okx_corner_metrics_tiled<true>, dim3((unsigned)tiles), dim3(okx::kTileStates), lds_bytes, (hipStream_t)stream, a, ta);
    else
      hipLaunchKernelGGL(okx::okx_corner_metrics_tiled<false>, dim3((unsigned)tiles), dim3(okx::kTileStates), lds_bytes, (hipStream_t)stream, a, ta);
    HIP_TRY(hipGetLastError());
    return OKX_OK;
  }
  const long long blocks = (n_states + 255) / 256;
  hipLaunchKernelGGL(okx::okx_corner_metrics_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

int32_t okx_camber_shim_batch(const okx_shim_roles* roles, int64_t n_geometries, int32_t n_points, double* d_points,
                              const double* d_shim, okx_shim_info* d_info, void* stream) {
  if (!roles || !d_points || !d_shim) return fail(OKX_ERR_INVALID, "null pointer");
  if (n_geometries < 0 || n_points <= 0) return fail(OKX_ERR_INVALID, "bad dimension");
  auto bad = [&](int32_t k) { return k < 0 || k >= n_points; };
  if (bad(roles->upper_outboard) || bad(roles->lower_outboard) || bad(roles->upper_inboard_front) ||
      bad(roles->upper_inboard_rear) || bad(roles->heading_inboard) || bad(roles->heading_outboard))
    return fail(OKX_ERR_INVALID, "shim role is not a point of the table");
  if (roles->n_upright_points < 0 || roles->n_upright_points > OKX_SHIM_MAX_POINTS || roles->n_rocker_points < 0 ||
      roles->n_rocker_points > OKX_SHIM_MAX_POINTS)
    return fail(OKX_ERR_INVALID, "at most %d upright / rocker points", OKX_SHIM_MAX_POINTS);
  for (int k = 0; k < roles->n_upright_points; ++k)
    if (bad(roles->upright_point[k]) || roles->upright_point[k] == roles->lower_outboard)
      return fail(OKX_ERR_INVALID, "upright point %d is not a movable point of the table", k);
  if (roles->rocker != 0 && roles->rocker != 1) return fail(OKX_ERR_INVALID, "rocker must be 0 or 1");
  if (roles->rocker) {
    if (bad(roles->rocker_axis_a) || bad(roles->rocker_axis_b) || bad(roles->pushrod_inboard) ||
        bad(roles->pushrod_outboard))
      return fail(OKX_ERR_INVALID, "rocker coupling role is not a point of the table");
    for (int k = 0; k < roles->n_rocker_points; ++k)
      if (bad(roles->rocker_point[k])) return fail(OKX_ERR_INVALID, "rocker point %d is not a point of the table", k);
  }
  if (n_geometries == 0) return OKX_OK;
  okx::shim::ShimArgs a;
  a.roles = *roles;
  a.points = d_points;
  a.shim = d_shim;
  a.info = d_info;
  a.n_geometries = n_geometries;
  a.n_points = n_points;
  const long long blocks = (n_geometries + 63) / 64;
  hipLaunchKernelGGL(okx::shim::okx_camber_shim_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return OKX_OK;
}

/* Diagnostic (okx_debug.h): record the LM passes of ONE problem of this program's subsequent quad-kernel
   solves into d_trace [256][8] = (mode, trial cost, accepted cost, lambda, step, gain ratio, accepted, done);
   pass a null pointer to switch it off. */
int32_t okx_debug_quad_trace(okx_program* p, double* d_trace, int64_t problem) {
  if (!p) return fail(OKX_ERR_INVALID, "null program");
  p->quad_trace = d_trace;
  p->quad_trace_problem = d_trace ? problem : -1;
  return OKX_OK;
}

/* Test hook: what the quad kernel's straight-line code computes at given free vectors d_x [B][n]:
   d_r [B][m], d_ata [B][n][n] (each structurally non-zero off-diagonal block is written ONCE, on one side of the
   diagonal, and the diagonal blocks in full: clear the
   buffer first), d_atr [B][n] and the damped step d_dx [B][n] = -(J^T J + lambda I)^-1 J^T r from
   its LDL^T (NaN when a pivot is not positive). */
int32_t okx_debug_quad_eval(okx_program* p, int64_t n_problems, const double* d_x, const double* d_targets,
                            double lambda, double* d_r, double* d_ata, double* d_atr, double* d_dx,
                            void* stream) {
  if (!p || !d_x || !d_r || !d_ata || !d_atr || !d_dx) return fail(OKX_ERR_INVALID, "null pointer");
  attach_when_ready(p, false);
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (!p->quad_fn_eval) return fail(OKX_ERR_INVALID, "no quad kernel: %s", p->quad_note);
  if (n_problems <= 0) return OKX_OK;
  okx::QuadEvalArgs q;
  q.x = d_x;
  q.targets = d_targets;
  q.r = d_r;
  q.ata = d_ata;
  q.atr = d_atr;
  q.dx = d_dx;
  q.lambda = lambda;
  q.n_problems = n_problems;
  const char* base = reinterpret_cast<const char*>(p->dev);
  q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
  q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
  q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
  const long long waves = (n_problems + 15) / 16;
  void* kargs[] = {(void*)&q};
  HIP_TRY(hipModuleLaunchKernel(p->quad_fn_eval, (int)(waves < 4096 ? waves : 4096), 1, 1, okx::kWave, 1, 1, 0,
                                (hipStream_t)stream, kargs, nullptr));
  return OKX_OK;
}

/* Test hook: okx_debug_quad_eval's quantities as the LANE kernel's straight-line code computes them. */
int32_t okx_debug_lane_eval(okx_program* p, int64_t n_problems, const double* d_x, const double* d_targets,
                            double lambda, double* d_r, double* d_ata, double* d_atr, double* d_dx,
                            void* stream) {
  if (!p || !d_x || !d_r || !d_ata || !d_atr || !d_dx) return fail(OKX_ERR_INVALID, "null pointer");
  attach_when_ready(p, false);
  std::shared_lock<std::shared_mutex> kernels(*p->kern_mutex);
  if (!p->lane_fn_eval) return fail(OKX_ERR_INVALID, "no lane kernel: %s", p->lane_note);
  if (n_problems <= 0) return OKX_OK;
  okx::QuadEvalArgs q;
  q.x = d_x;
  q.targets = d_targets;
  q.r = d_r;
  q.ata = d_ata;
  q.atr = d_atr;
  q.dx = d_dx;
  q.lambda = lambda;
  q.n_problems = n_problems;
  const char* base = reinterpret_cast<const char*>(p->dev);
  q.design_pos = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, design_pos));
  q.row_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, row_param));
  q.dop_param = reinterpret_cast<const double*>(base + offsetof(okx::DevProgram, dop_param));
  const long long waves = (n_problems + 63) / 64;
  void* kargs[] = {(void*)&q};
  HIP_TRY(hipModuleLaunchKernel(p->lane_fn_eval, (int)(waves < 1024 ? waves : 1024), 1, 1, okx::kWave, 1, 1, 0,
                                (hipStream_t)stream, kargs, nullptr));
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
  a.confirm = 1;
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
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
    HIP_TRY(hipLaunchKernel((const void*)fn, dim3(grid_for(p, (n_problems + 2) / 3)), dim3(okx::kWave),
                            kargs, p->packed_lds_bytes, (hipStream_t)stream));
    return OKX_OK;
  }
  void* kargs[] = {(void*)&dev, (void*)&a};
  solve_kernel_t fn = okx::okx_solve_kernel<18, true>;
  HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
  HIP_TRY(hipLaunchKernel((const void*)fn, dim3(grid_for(p, n_problems)), dim3(okx::kWave), kargs,
                          p->lds_bytes, (hipStream_t)stream));
  return OKX_OK;
}

/* Plan introspection for CPU-side tests: fills counts without touching a device. */
int32_t okx_debug_plan_stats(const okx_program_desc* desc, int32_t* out8) {
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
