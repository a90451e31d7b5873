// okx_quad.hpp — runtime-specialised "quad" solve kernel: source generation (okx_quadgen.cpp)
// and hiprtc compilation / caching / loading (okx_jit.cpp).
#pragma once

#include <string>
#include <vector>

#include "okx_plan.hpp"

namespace okx {

// Developer switches.  ONE environment variable, OKX_DEV, holds a comma-separated list of `name` or `name=value` items;
// nothing in it is needed to run the product (the two product variables are OKX_KERNEL_CACHE - where compiled kernels are
// kept - and OKX_VERBOSE).  The names that exist are listed in tools/README.md, each is covered by
// tests/test_dev_switches.py (generated source still compiles and is deterministic); the switches that change generated
// source are part of its text and therefore of the kernel cache key.
//   generators   quad_mark, quad_timeline, quad_no_light, quad_no_head, quad_no_fast, quad_two_waves, pair_no_head, pair_first_order_head,
//                pair_lds_homes, pair_cold_lds, lane_mark, lane_timeline, lane_lds_tables, lane_nested, lane_refine
//   library      no_quad, no_lane, no_cold, tangent_generic, evaluate_quad, evaluate_lane, keep_source
bool dev_switch(const char* name);

// default / largest degree of the chain-head predictor's Chebyshev series
constexpr int kPredictorDegree = 7;
constexpr int kPredictorMaxDegree = 12;
constexpr int kPredictorLdsDoubles = 1024;  // single mode: tables up to this size are staged through LDS (8 KB per wavefront)
constexpr int kQuadMaxFree = 9;          // n <= 27 unknowns (a corner with pushrod, rocker and coil-over): up to 8 free points the lane-owned
                                         // rows of J^T J stay in registers, the ninth costs ~400 B of scratch (3.2e8 solves/s against 1.5e7 on the interpreter)
constexpr int kQuadMaxJoins = 4;          // pair mode: rows joining the two halves (rack; T-bar crossbar length and centre plane; rocker-to-rocker heave link)
constexpr int kQuadMaxFreePerSide = 11;  // pair mode (two identical halves, one quad each): free points per half (rocker corner + droplink + heave pickup)
constexpr int kLaneMaxFree = 6;          // lane kernel (one lane per problem): n <= 18 unknowns, lower triangle of J^T J <= 171 doubles

// Kernel arguments of the generated kernels (mirrors `struct QArgs` in the generated source).
struct QuadArgs {
  const double* targets;
  const double* geom_pos;
  const double* geom_row_param;
  double* out_pos;
  okx_info* info;
  long long n_problems, steps_per_geometry, chain_len;
  int max_iter, confirm;
  double step_tol, grad_tol, ftol, lambda0, residual_tolerance;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
  double* trace;            // diagnostic: [256][8] per-pass record of one problem, or null
  long long trace_problem;
  const double* predictor;  // chain-head polynomial model (okx_program_fit_predictor) or null
  long long predictor_mode; // 2: every chain step starts from the model
  long long predictor_len;  // doubles in the table
  const double* head;       // per-geometry first-step table (okx_quad_head_u/_g) or null
  long long out_mode;       // okx_solve_opts.output: 0 records, 1 free coordinates, 2 nothing
};

// Arguments of the generated `okx_quad_head_u/_g` (mirrors `struct QHeadArgs`): one quad per geometry evaluates the
// design state once and tabulates the first Levenberg-Marquardt step of every chain head of that geometry.
struct QuadHeadArgs {
  const double* geom_pos;
  const double* geom_row_param;
  double* head;
  long long n_geometries;
  double lambda0;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
};
// doubles per geometry in that table (0: the program has no quad kernel): Q[k][F][4] per half, M[j][k], N[j][k] for
// k = constraint gradient + one column per program target, 8 scalars (okx_quadgen.cpp)
int quad_head_stride(const DevProgram& program);

// Arguments of the generated parity kernel `okx_quad_eval` (mirrors `struct QEvalArgs`).
struct QuadEvalArgs {
  const double* x;
  const double* targets;
  double* r;
  double* ata;
  double* atr;
  double* dx;
  double lambda;
  long long n_problems;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
};

// Arguments of the generated `okx_quad_expand` (mirrors `struct QExpandArgs`).
struct QuadExpandArgs {
  const double* free;
  const double* geom_pos;
  double* out_pos;
  long long n_problems, steps_per_geometry;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
};

// Arguments of the generated tangent kernels `okx_quad_tangent_u/_g` (mirrors `struct QTanArgs`).
struct QuadTanArgs {
  const double* pos;
  const double* geom_pos;
  const double* geom_row_param;
  double* tan;
  okx_tangent_info* tinfo;
  long long n_problems, steps_per_geometry;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
};

// ---- evaluated solve (okx_solve_evaluated_batch): tangents and the metric catalog as the solve kernels' epilogue ----
// The role POINTS of okx_corner_roles (indices into the program's output list, -1 = absent) are compile-time constants of
// the generated "evaluated" module; the numeric part travels as a kernel argument.
constexpr int kEvalSlots = 15;  // wheel centre, contact patch, axle outboard / inboard, steer lower / upper, damper top / bottom, 6 instant-axis points, rack
struct EvalSpec {
  int wheel_center, contact_patch, axle_inboard, axle_outboard, steer_lower, steer_upper;
  int ia_kind, ia_point[6];
  int damper_top, damper_bottom, rack;
};
struct EvalScalars {  // mirrors `struct EvCfg` of the generated source (okx_evalsrc.cpp)
  double side_sign, design_wheel_center_z, design_contact_patch_z, design_rack_y, wheelbase, cg_z, front_brake_bias;
  int axle_position, driven_axle;
};
bool eval_spec_from_roles(const DevProgram& P, const okx_corner_roles& roles, EvalSpec* spec, std::string* why);
void eval_scalars_from_roles(const okx_corner_roles& roles, EvalScalars* scalars);
int eval_slot_point(const EvalSpec& spec, int slot);         // output-list index of a role slot (-1: absent)
std::string eval_metrics_source(const EvalSpec& spec);       // role #defines + the catalog on duals
// ---- composed axles (pair mode): both corners' catalogs, the axle-scope metrics and the hardware / rotation roles ----
// The evaluated module of a pair-mode program (okx_program_enable_axle_evaluation) is specialised to BOTH corners' role
// points and to the points (and kinds) of up to OKX_MAX_ROTATIONS roles of the kinds of okx_rotation_role; their numbers
// travel as kernel arguments.  Row layout of d_eval: OKX_EVAL_AXLE_COLUMNS per row (okx.h).
constexpr int kEvalAxleColumns = 64;
struct EvalRoleSpec { int kind, point, point_b; };
struct AxleEvalSpec {
  EvalSpec side[2];
  int n_roles;
  EvalRoleSpec role[8];
};
struct EvalRoleNum { double design[3], axis_point[3], axis_dir[3], scale; };  // mirrors `struct EvRoleNum` of the generated source
bool axle_eval_spec_from_roles(const DevProgram& P, const okx_axle_roles& roles, AxleEvalSpec* spec, std::string* why);
std::string eval_roles_source();  // the role kinds of okx_rotation_role on duals (pair-mode modules only)
// Arguments of the evaluated solve kernels okx_quad_evsolve_u/_g, okx_quad_evcold_u, okx_lane_evsolve_u/_g (mirrors
// `struct QEvArgs`): the solve's own arguments, then what the epilogue writes and the roles' numeric part.  The generated
// corner modules declare the struct up to `cfg`; pair-mode modules read the right corner's numbers and the roles' as well.
struct QuadEvArgs {
  QuadArgs q;
  double* tan;    // [B][T][n_out][3] or null
  double* ev;     // [B][1 + T][OKX_EVAL_COLUMNS] (pair mode: OKX_EVAL_AXLE_COLUMNS) or null
  EvalScalars cfg;
  EvalScalars cfg_r;
  EvalRoleNum roles[8];
};
// Arguments of okx_quad_evaluate_u/_g (mirrors `struct QEvPosArgs`): the same epilogue on given solved states.
struct QuadEvPosArgs {
  const double* pos;
  const double* geom_pos;
  const double* geom_row_param;
  double* tan;
  double* ev;
  long long n_problems, steps_per_geometry;
  const double* design_pos;
  const double* row_param;
  const double* dop_param;
  EvalScalars cfg;
  EvalScalars cfg_r;       // (pair-mode modules only, like QuadEvArgs)
  EvalRoleNum roles[8];
};

// A program made of two structurally identical halves joined by one distance row (the composed
// axle), seen as its half ("side") program plus the index maps of both sides (okx_pairview.cpp).
struct PairView {
  DevProgram side;             // side-0 sub-program in its own point / row numbering; targets = union of both sides
  std::vector<int> pt[2];      // side point -> program point
  std::vector<int> row[2];     // side constraint row -> program constraint row
  std::vector<int> tgt[2];     // side target -> program target index, -1: that side has no such target
  std::vector<int> dop[2];     // side derived op -> program derived op
  std::vector<int> out[2];     // side output k -> index in the program's output list, -1: not written by that side
  std::vector<int> shared_out; // output-list indices of fixed points neither half owns (written once)
  std::vector<int> shared_pt;  // their program point indices
  int couple_point;            // side point (free) joined to its mirror image by the (first) coupling row
  int couple_row;              // program constraint row of that distance
  // Every row joining the halves, in program row order (joins[0] = couple_point / couple_row): a distance between a
  // free point and its mirror image (the rack, a T-bar's crossbar) or the midpoint of such a pair on a plane (the
  // T-bar's centre line).  One joining row: a 2 x 2 Woodbury system; k rows: 2k x 2k (okx_quadgen.cpp).
  struct Join { int row, point, type; };
  std::vector<Join> joins;
  int n_prog_points, n_prog_crows, n_prog_targets, n_prog_out;
};
bool build_pair_view(const DevProgram& P, PairView* pv, std::string* why);

// Emits the HIP source of the kernel specialised to `P`.  Returns false (and says why) when the
// program uses a feature the generator has no code path for; the caller then keeps the generic
// interpreter kernels of okx_kernels.hip.
// `lds_homes` (pair mode only): the chain constants and the fixed points live in LDS instead of registers - the
// fallback for a half program whose register-resident variant spills (see quad_build).
// `eval` non-null: the EVALUATED module of the program instead - the solve bodies with the tangent / metric epilogue
// (kernels okx_quad_evsolve_u/_g, okx_quad_evcold_u) and the same epilogue on given states (okx_quad_evaluate_u/_g);
// single mode only.
// `axle_eval` non-null (pair-mode programs only): the evaluated module of a composed axle - the same kernel names; each half's
// quad evaluates its own corner's catalog, the left one also the axle-scope metrics, both the roles.
bool quad_generate(const DevProgram& P, int waves_per_simd, std::string* src, std::string* why, bool lds_homes = false,
                   const EvalSpec* eval = nullptr, const AxleEvalSpec* axle_eval = nullptr);

// Emits the HIP source of the LANE kernel specialised to `P` (okx_lanegen.cpp): one lane per problem, 64 problems per
// wavefront, for batches that fill the chip several times over.  Kernels okx_lane_solve_u/_g (arguments: QuadArgs) and
// okx_lane_eval (QuadEvalArgs).  Returns false (and says why) when the program does not fit one lane's registers.
// `variant` in [0, lane_variant_count()): the same arithmetic with other hints to the compiler (see lane_build).
// `eval` non-null: the evaluated module (okx_lane_evsolve_u/_g: independent solves with the tangent / metric epilogue).
bool lane_generate(const DevProgram& P, std::string* src, std::string* why, int variant = 0, const EvalSpec* eval = nullptr);
// How the lane kernel walks chains.  While x, dx and the chain history (two previous solutions) leave at least 16 of the
// 80 LDS slots to the factor's parked rows, a lane keeps its chain in LDS and loops over its steps (MacPherson: 60 + 20).
// Otherwise (double wishbone: 72 + 8) the chain body IS the independent-solve body in one flat loop over
// (wave unit, chain step), and everything a chain carries from step to step - the last three solutions, whether they
// converged, the damping - lives in a per-launch global scratch of lane_flat_chain_doubles() per wavefront, passed in
// QuadArgs.predictor (the lane kernel has no fitted model).  OKX_LANE_FLAT_CHAIN=0/1 forces either.
bool lane_chain_is_flat(int n_vars);
inline long long lane_flat_chain_doubles(int n_vars) { return 3LL * (n_vars + 2) * 64; }
// The NESTED start mode of the lane kernel (okx_lane_nest_*; okx_lanegen.cpp): a lane owns four consecutive steps, solved in
// the order 0, 2, 1, 3, each after the first started from the interpolant of the steps its wave unit has already solved;
// the launch's scratch holds, per wavefront, four entries of (n_vars + 3) slots x 64 lanes.
constexpr int kLaneNestSteps = 4;
inline long long lane_nest_doubles(int n_vars) { return (long long)kLaneNestSteps * (n_vars + 3) * 64; }
int lane_variant_count();
bool lane_variants_same_arithmetic(int a, int b);  // same operations in the same order: bit-identical results
// lane_generate + quad_compile over the emission variants: keeps the first variant whose independent-solve bodies
// (okx_lane_solve_*) do not spill, else the one that spills least.  The choice is remembered next to the code objects
// (<hash of variant 0's source>.lanevar in the kernel cache), so a later call compiles nothing.  `variant_out` (may be
// null) receives the variant kept.  `good_enough_scratch`: the search stops at the first variant whose independent-solve
// bodies spill at most that many bytes - 0 for okx_precompile (the full search, whose result is remembered), the 256 B
// auto selection accepts for okx_program_create (a program nobody precompiled must not wait for eight hiprtc runs).
// `overrides` (may be null): the register allocator's result differs from kernel to kernel of one module, so after a FULL
// search every solve / chain kernel that spills less in another variant's module than in the kept one is listed here with
// that module (the caller loads it and takes this one kernel from it); remembered in the same file.  Only variants with the
// kept one's arithmetic are candidates: a program's kernels give the same bits whichever output they write.
struct LaneOverride {
  std::string kernel;  // e.g. "okx_lane_solve_g"
  std::string code;    // the code object to take it from
  int variant, scratch;
};
bool lane_build(const DevProgram& P, std::string* src, std::string* code, std::string* why, bool ignore_cached = false,
                int* variant_out = nullptr, int good_enough_scratch = 0, bool cache_only = false,
                std::vector<LaneOverride>* overrides = nullptr);
// Scratch bytes of the kernel called exactly `name` (-1: no such kernel).
int quad_code_kernel_scratch_bytes(const std::string& code, const char* name);

// Compiles `src` for gfx950 with hiprtc (no device needed) or fetches it from the on-disk cache
// (<dir of libokx.so>/_kcache/<hash>.okxc, override with OKX_KERNEL_CACHE).  Returns the code
// object in `code`; false + message on failure.
// `ignore_cached` recompiles and overwrites the cache entry (used once when a cached object fails to load).
// `cache_only`: never run the compiler - a source that is not in the cache fails with kNotCached in `err` (what
// okx_program_create asks first: a miss is compiled on a host thread while the interpreter kernels serve the program).
extern const char* const kNotCached;
bool quad_compile(const std::string& src, std::string* code, std::string* err, bool ignore_cached = false, bool cache_only = false);

// Largest private-segment (scratch) size among the kernels of a code object whose name starts with `prefix`, read from
// the code object's metadata note; -1 when no such kernel is found.
int quad_code_scratch_bytes(const std::string& code, const char* prefix);
int quad_code_lds_bytes(const std::string& code, const char* prefix);  // static LDS of those kernels

// quad_generate + quad_compile.  A pair-mode program is first generated with its chain constants in registers; if the
// compiler then spills in the solve kernels (a larger half program than the BASELINE axle), the variant with LDS homes
// is generated and compiled instead.  `src` receives the source of the variant that was kept.
bool quad_build(const DevProgram& P, int waves_per_simd, std::string* src, std::string* code, std::string* why,
                bool ignore_cached = false, bool cache_only = false);
// The evaluated modules of a program for one set of metric roles: quad (single mode) and, when the program has a lane
// kernel, lane (the emission variant with the least scratch in okx_lane_evsolve_*; remembered in the cache like lane_build's).
bool quad_eval_build(const DevProgram& P, const EvalSpec& spec, int waves_per_simd, std::string* code, std::string* why, bool cache_only = false);
bool lane_eval_build(const DevProgram& P, const EvalSpec& spec, std::string* code, std::string* why, bool cache_only = false, int* scratch_out = nullptr);
bool quad_axle_eval_build(const DevProgram& P, const AxleEvalSpec& spec, int waves_per_simd, std::string* code, std::string* why, bool cache_only = false);

}  // namespace okx
