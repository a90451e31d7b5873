/*
 * okx_debug.h — diagnostic and test-hook exports of libokx.so.
 *
 * NOT part of the drop-in boundary (include/okx.h): nothing here replaces a reference
 * interface.  These entry points exist for the parity tests (rung R1b: the normal equations
 * as the kernels form them), for the CPU-side plan tests and for the profiling tools under
 * tools/.  They follow okx.h's conventions (plain C, d_* = device pointers, okx_status
 * return codes) but carry no compatibility promise across ABI versions.
 */
#ifndef OKX_DEBUG_H
#define OKX_DEBUG_H

#include "okx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* J^T J [B][n][n] and J^T r [B][n] exactly as the generic interpreter kernels form them from
 * their sparse plans, at free vectors d_x [B][n] (d_r [B][m] receives the residuals). */
int32_t okx_debug_normal_equations(okx_program* prog, int64_t n_problems, const double* d_x,
                                   const double* d_targets, double* d_r, double* d_ata, double* d_atr,
                                   void* stream);

/* What the quad kernel's straight-line code computes at free vectors d_x [B][n]: d_r [B][m],
 * d_ata [B][n][n] (each structurally non-zero off-diagonal block written ONCE, on one side of the
 * diagonal, diagonal blocks in full: clear the buffer first), d_atr [B][n] and the damped step
 * d_dx [B][n] = -(J^T J + lambda I)^-1 J^T r from its LDL^T (NaN when a pivot is not positive). */
int32_t okx_debug_quad_eval(okx_program* prog, int64_t n_problems, const double* d_x, const double* d_targets,
                            double lambda, double* d_r, double* d_ata, double* d_atr, double* d_dx,
                            void* stream);

/* The same quantities as the LANE kernel's straight-line code computes them (okx.h: lane kernel). */
int32_t okx_debug_lane_eval(okx_program* prog, int64_t n_problems, const double* d_x, const double* d_targets,
                            double lambda, double* d_r, double* d_ata, double* d_atr, double* d_dx,
                            void* stream);

/* Record the LM passes of ONE problem of this program's subsequent quad-kernel solves into
 * d_trace [256][8] = (mode, trial cost, accepted cost, lambda, step, gain ratio, accepted, done);
 * a null pointer switches it off.  Per program (no process-global state). */
int32_t okx_debug_quad_trace(okx_program* prog, double* d_trace, int64_t problem);

/* okx_solve_batch for an n = 18 program on the instrumented instantiation of the generic kernel:
 * per-phase cycle sums of workgroup 0 in d_phase_cycles[12] (0 staging, 1 problem setup,
 * 2 x -> positions + derived points, 3 rows, 4 reductions + LM logic, 5 normal equations,
 * 6 factorisation, 7 substitutions, 8 output). */
int32_t okx_debug_phase_profile(okx_program* prog, const okx_solve_opts* opts, int64_t n_problems,
                                const double* d_targets, double* d_out_pos, okx_info* d_info,
                                unsigned long long* d_phase_cycles, void* stream);

/* Plan introspection without a device: out8 = (n, m, block pairs, J^T J contributions, active
 * derived ops, Jacobian row stride, lda, LDS bytes of the generic kernel). */
int32_t okx_debug_plan_stats(const okx_program_desc* desc, int32_t* out8);

/* Scratch (private segment) bytes of the runtime-specialised solve kernels this program would use, read from the code
 * object's metadata (no device needed; compiles into the kernel cache like okx_precompile): 0 = nothing spilt. */
int32_t okx_debug_kernel_scratch(const okx_program_desc* desc, int32_t* scratch_bytes);

/* The same for the lane kernel: out3 = (scratch bytes of the independent-solve bodies, of the chain bodies, emission
 * variant kept by the build). */
int32_t okx_debug_lane_scratch(const okx_program_desc* desc, int32_t* out3);

#ifdef __cplusplus
}
#endif
#endif /* OKX_DEBUG_H */
