/*
 * okx.h — C-ABI of the MI355X-native batched suspension-kinematics constraint solver.
 *
 * This is the drop-in boundary for ONE hot path of nickmccleery/open-kinematics: the
 * per-sweep-step nonlinear least-squares solve.  The reference has no FFI for this path
 * (it is pure Python); the entry points below are what a binding for it would bind, and
 * each cites the reference interface it replaces (paths relative to the reference root,
 * src/kinematics/core/...).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*.
 *   - every pointer named d_* is a DEVICE-ACCESSIBLE pointer: normally HBM; pinned host
 *     memory (hipHostMalloc / hipHostRegister, mapped) is accepted as well - okx_solve_batch's
 *     d_targets, d_out_pos and d_info then cross PCIe as the kernel reads / stores them, with
 *     no copy command on either side (bench.py `e2e.zero_copy`; the host may read the outputs
 *     once an event recorded after the call has completed).  Everything else is host memory
 *     that is only read during the call.
 *   - all entry points return OKX_OK (0) or a negative okx_status; they never throw.
 *     okx_last_error() returns a thread-local message for the last failure.
 *   - all floating point is IEEE fp64; lengths in millimetres, angles in radians.
 *
 * The "constraint program" is the flattened form of what the reference passes to
 * solve_suspension_sweep() (solver.py:654-660): initial SuspensionState (state.py:23-72),
 * list[Constraint] (constraints.py), DerivedPointsManager spec (points/derived/) and the
 * target rows of a SweepConfig (targeting.py:51-104).
 */
#ifndef OKX_H
#define OKX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKX_ABI_VERSION 6   /* 6: the evaluated solve of composed axles (okx_axle_roles, okx_program_enable_axle_evaluation, okx_precompile_axle_evaluation, okx_program_eval_columns).  5: the evaluated solve (okx_program_enable_evaluation, okx_solve_evaluated_batch, okx_evaluate_batch, okx_precompile_evaluation).  4: okx_rotation_role.kind / point_b (hardware metrics of composed axles), okx_program_has_cold_body, okx_program_ready.  3: okx_solve_opts.output (+ reserved); lane kernel entry points.  2: confirm_full_pass; diagnostics moved to okx_debug.h */

/* Hard limits of one problem (one wavefront owns one problem). */
#define OKX_MAX_VARS 126     /* n = 3 * free points (one thread per variable: one wavefront up to 63, two beyond) */
#define OKX_MAX_ROWS 128     /* m = constraint rows + target rows                    */
#define OKX_MAX_POINTS 96    /* fixed + free + derived                               */
#define OKX_MAX_TARGETS 8
#define OKX_ROW_PARAMS 8     /* doubles per constraint row                           */
#define OKX_ROW_POINTS 4     /* point slots per constraint row                       */

typedef enum okx_status {
  OKX_OK = 0,
  OKX_ERR_INVALID = -1,      /* malformed program / argument                         */
  OKX_ERR_LIMIT = -2,        /* exceeds an OKX_MAX_* limit                           */
  OKX_ERR_UNDERDETERMINED = -3, /* n_vars > n_rows (solver.py:116-121)               */
  OKX_ERR_DEVICE = -4,       /* HIP runtime failure (no GPU, launch error, ...)      */
  OKX_ERR_ALLOC = -5
} okx_status;

/*
 * Constraint row types.  One scalar residual per row, defined exactly as
 * constraints.py does (softnorm(s) = sqrt(s + 1e-12) - 1e-6, soft_math.py:16-27);
 * Jacobian rows follow jacobians.py / tools/generate_jacobians.py.
 * Point slots p0..p3 and params q0..q7 per type:
 */
typedef enum okx_row_type {
  OKX_ROW_DISTANCE = 0,          /* constraints.py:89-134   p0=p1,p1=p2        q0=L                */
  OKX_ROW_SPHERICAL = 1,         /* constraints.py:137-170  p0,p1              —                   */
  OKX_ROW_ANGLE = 2,             /* constraints.py:173-243  v1s,v1e,v2s,v2e    q0=alpha            */
  OKX_ROW_THREE_POINT_ANGLE = 3, /* constraints.py:246-308  p1,p2(vertex),p3   q0=alpha            */
  OKX_ROW_VECTORS_PARALLEL = 4,  /* constraints.py:311-371  v1s,v1e,v2s,v2e    —                   */
  OKX_ROW_VECTORS_PERPENDICULAR = 5, /* constraints.py:374-429 same            —                   */
  OKX_ROW_EQUAL_DISTANCE = 6,    /* constraints.py:432-477  p1,p2,p3,p4        —                   */
  OKX_ROW_FIXED_AXIS = 7,        /* constraints.py:480-516  p0                 q0=axis(0/1/2) q1=value */
  OKX_ROW_POINT_ON_LINE = 8,     /* constraints.py:519-576  p0                 q0..2=line point q3..5=line dir */
  OKX_ROW_POINT_ON_PLANE = 9,    /* constraints.py:579-627  p0                 q0..2=plane point q3..5=normal  */
  OKX_ROW_MIDPOINT_ON_PLANE = 10,/* constraints.py:630-666  a,b                q0..2=plane point q3..5=normal  */
  OKX_ROW_COPLANAR = 11,         /* constraints.py:669-709  p1..p4             —                   */
  OKX_ROW_SCALAR_TRIPLE = 12,    /* constraints.py:712-733  p1..p4             q0=V q1=scale       */
  /*
   * Extension (not a reference class): one Cartesian component of
   * cross(p - line_point, line_dir).  Three such rows replace one POINT_ON_LINE row
   * when the host flattens with line_mode="pinned"; sum of squares = d^2 instead of
   * softnorm(d^2)^2.  Same minimiser whenever the point can reach the line;
   * it removes the reference row's zero-gradient degeneracy (sensitivity.py:83-87,146-174
   * does the same with explicit pins).
   */
  OKX_ROW_LINE_PIN = 13,         /* p0  q0..2=line point q3..5=line dir q6=component(0/1/2)        */
  OKX_ROW_TYPE_COUNT = 14
} okx_row_type;

/*
 * Derived-point ops (points/derived/definitions.py), evaluated in the given
 * (topological) order; closed-form 3x3 chain-rule blocks replace the reference's
 * dual-number pass (manager.py:271-324).
 */
typedef enum okx_dop_type {
  OKX_DOP_MIDPOINT = 0,      /* definitions.py:76-89    out = a + (b - a)/2               pts=(a,b)           */
  OKX_DOP_ALONG = 1,         /* definitions.py:24-33,92-155 out = base + normalize(a - b)*c pts=(base,a,b) q=c */
  OKX_DOP_CONTACT_PATCH = 2, /* definitions.py:36-73,158-180 out = wc + normalize(-Z - ((-Z).ax)ax)*R,
                                ax = normalize(axo - axi)   pts=(wc,axi,axo) q=R                              */
  OKX_DOP_TYPE_COUNT = 3
} okx_dop_type;

/* Flattened constraint program (host arrays; copied by okx_program_create). */
typedef struct okx_program_desc {
  int32_t abi_version;        /* = OKX_ABI_VERSION */
  int32_t n_points;           /* P: all points, any order */
  int32_t n_free;             /* F: free points; n_vars = 3F */
  int32_t n_derived;          /* D */
  int32_t n_rows;             /* Mc: constraint rows (targets excluded) */
  int32_t n_targets;          /* T: target rows appended after the constraint rows */
  int32_t n_out;              /* points written per solved problem */
  int32_t reserved;
  const int32_t* free_point;  /* [F]      point index of variable block k (state.py:46-50 order) */
  const int32_t* dop_type;    /* [D]      okx_dop_type */
  const int32_t* dop_out;     /* [D]      output point index */
  const int32_t* dop_pts;     /* [D][4]   input point indices, -1 = unused */
  const double* dop_param;    /* [D]      scalar parameter */
  const int32_t* row_type;    /* [Mc]     okx_row_type */
  const int32_t* row_pts;     /* [Mc][4]  point indices, -1 = unused */
  const double* row_param;    /* [Mc][8]  design-state parameters (geometry 0) */
  const int32_t* tgt_point;   /* [T]      targeted point */
  const double* tgt_dir;      /* [T][3]   unit direction (targeting.py:135-148) */
  const int32_t* out_point;   /* [n_out]  output point indices (Suspension.output_points()) */
  const double* design_pos;   /* [P][3]   design positions incl. derived (geometry 0) */
} okx_program_desc;

typedef struct okx_program okx_program; /* opaque, device-resident */

/* Levenberg–Marquardt controls.  Zero-initialise then call okx_default_opts(). */
typedef struct okx_solve_opts {
  int32_t max_iter;       /* LM iterations per problem (default 100)                       */
  int32_t chain;          /* 0: every problem starts from its geometry's design state
                             (independent problems, one wavefront each);
                             1: reference semantics (solver.py:716,774): problems of one
                             geometry form a sequential chain, step k starts at step k-1's
                             solution; one wavefront walks one chain.                      */
  int64_t steps_per_geometry; /* problems [g*S, (g+1)*S) use geometry g; 0 = single geometry */
  int64_t chain_len;      /* > 0: consecutive problems are grouped into chains of this length
                             (never across geometries); one wavefront walks one chain and
                             warm-starts each problem from its predecessor, the chain head
                             starts from the design state.  1 = independent cold starts,
                             0 = take the length from `chain` (0 -> 1, 1 -> whole sweep),
                             -1 = auto: one chain per resident wavefront.                  */
  double step_tol;        /* stop when max|dx| <= step_tol (mm, default 1e-11)             */
  double grad_tol;        /* > 0: stop when max|J^T r| <= grad_tol.  < 0: MINPACK's gtol (what SolverConfig.gtol means,
                             solver.py:158-169): stop when max_j |(J^T r)_j| / (|J_j| |r|) <= -grad_tol (lmder's gnorm:
                             the cosine between the residual and the Jacobian's columns).  0 (default): unused.  A launch
                             with a gradient stop runs the general bodies without the shared first step. */
  double ftol;            /* stop when an accepted step reduces the cost by <= ftol*cost,
                             actual and predicted (MINPACK's ftol test; default 1e-10):
                             terminates infeasible targets at their compromise point so
                             that the residual_tolerance check can reject them
                             (solver.py:732-747)                                           */
  double lambda0;         /* initial damping relative to max diag(J^T J) (default 1e-6)    */
  double residual_tolerance; /* informational: info.flags bit1 set if max|r| exceeds it
                                (solver.py:735-747, default 1e-3)                          */
  int32_t kernel;         /* 0 auto (the runtime-specialised quad kernel when the program has one,
                             else 1 or 2 by size), 1 generic interpreter, one problem per
                             wavefront, 2 generic lane-group packed (several small problems per
                             wavefront; falls back to 1 when a problem needs more than 32
                             lanes), 3 quad kernel (OKX_ERR_INVALID when the program has none),
                             4 lane kernel (one lane per problem, 64 per wavefront; auto picks it for
                             batches of at least okx_program_lane_threshold() problems)          */
  int32_t confirm_full_pass; /* quad kernel only.  0 (default): a step predicted to land within
                             step_tol of the solution (damping contraction lambda / min pivot and
                             the observed quadratic contraction, both with a 100x margin) is
                             applied and confirmed by a residual-only evaluation (cost must not
                             rise) instead of a full Jacobian / factorisation pass.
                             non-zero: always end on a computed correction <= step_tol.        */
  int32_t predictor;      /* quad kernel, own-geometry launches.  non-zero: chain heads (cold starts) begin at
                             the polynomial model of okx_program_fit_predictor instead of the design state
                             (ignored until a predictor has been fitted).  Same solutions, fewer passes. */
  int32_t shared_first_step; /* quad kernel (single mode).  non-zero (default 1): a chain head starts at its geometry's
                             design state, where only the target residuals depend on the problem: the constraint
                             residuals, the Jacobian, J^T J and its damped factorisation are identical for every problem
                             of that geometry.  That first Levenberg-Marquardt pass is evaluated ONCE PER GEOMETRY (own
                             geometry: once per program and lambda0; geometry tables: a small launch ahead of the solve,
                             when there are at least four steps per geometry) and every head takes its first step
                             dx = -(Q_0 + sum_t r_t Q_t), Q_k = (J^T J + lambda I)^-1 G_k, from that table.  Same
                             iteration, same iterates up to rounding; info.nfev counts the evaluations a problem ran
                             itself.  0: every chain head runs its own first pass.  With geometry tables the scratch
                             table lives in the program: such launches of one program must be stream-ordered.  Own
                             geometry: one table per lambda0, the default's filled at okx_program_create, any other by
                             the first launch that asks for it (on its stream; launches on other streams wait for it). */
  int32_t output;         /* what okx_solve_batch writes to d_out_pos (generated kernels; the interpreter kernels only
                             know OKX_OUTPUT_RECORDS):
                             OKX_OUTPUT_RECORDS (0, default) [B][n_out][3]: every output point, the reference's state copy
                               (solver.py:763);
                             OKX_OUTPUT_FREE [B][n_free][3]: the solved free points only, in the program's free_point order
                               (144 B instead of 360 B per double-wishbone solve: what a PCIe link or an xGMI all-gather
                               wants to carry; okx_expand_positions_batch rebuilds the records, bit-identical);
                             OKX_OUTPUT_NONE: nothing (d_out_pos may be NULL) - only d_info, for feasibility scans.      */
  int32_t reserved;
} okx_solve_opts;

enum { OKX_OUTPUT_RECORDS = 0, OKX_OUTPUT_FREE = 1, OKX_OUTPUT_NONE = 2 };

/* Per-problem result, the device analogue of SolverInfo (solver.py:83-96). */
typedef struct okx_info {
  double max_residual;    /* max |r_i| at the returned state, reference row definitions    */
  double cost;            /* 0.5 * sum r_i^2 at the returned state                         */
  double last_step;       /* max|dx| of the last accepted step                             */
  int32_t iterations;     /* LM iterations used                                            */
  int32_t nfev;           /* residual evaluations (SolverInfo.nfev analogue)               */
  int32_t flags;          /* bit0 converged, bit1 max_residual > residual_tolerance,
                             bit2 damping failure / non-finite, bit3 ill-conditioned: in the last
                             factorisation (smallest pivot - damping) / largest pivot fell below
                             OKX_ILL_CONDITIONED_PIVOT_RATIO, i.e. cond(J) above ~1e6: a singular
                             configuration, typically a compromise point just beyond kinematic
                             lock-out whose residual is still inside residual_tolerance.  The
                             minimiser is only defined to ~cond(J) * eps * |x| there: advisory.
                             (quad kernel and the one-problem-per-wavefront interpreter).  Pair-mode kernels
                             test the halves' pivots and the stiffness of the mode their joining row ties
                             together.  The generated kernels test, beside the pivots (lower bounds of the
                             smallest eigenvalue), the Rayleigh quotient of the last step in J^T J + lambda I less
                             lambda (an upper bound, close where J is singular): a singular direction spread
                             over several pivots leaves every one of them well above the damping.  A solve that
                             ends on the ftol test while missing its rows by more than 1 % of
                             residual_tolerance (a compromise point) sets it as well.                       */
  int32_t reserved;
} okx_info;

#define OKX_INFO_CONVERGED 1
#define OKX_INFO_RESIDUAL_EXCEEDED 2
#define OKX_INFO_FAILED 4
#define OKX_INFO_ILL_CONDITIONED 8
#define OKX_ILL_CONDITIONED_PIVOT_RATIO 1e-12 /* (min pivot - lambda) / max pivot of the LDL^T of J^T J + lambda I */

int32_t okx_abi_version(void);
const char* okx_last_error(void);
void okx_default_opts(okx_solve_opts* opts);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int32_t okx_device_count(void);

/*
 * Replaces ResidualComputer.__init__/build_jac_plan (solver.py:187-214, :281-500):
 * validates the program, uploads it to the current HIP device and precomputes the
 * sparsity plan.  Returns OKX_ERR_UNDERDETERMINED like validate_least_squares_dimensions
 * (solver.py:99-121).
 */
int32_t okx_program_create(const okx_program_desc* desc, okx_program** out);
void okx_program_destroy(okx_program* prog);

/*
 * Replaces solve_suspension_sweep (solver.py:654-776) for a batch of B sweep-step
 * problems.  d_targets holds the ABSOLUTE target scalars (convert_targets_to_absolute,
 * solver.py:584-627, is host work).  For a perturbed-geometry ensemble pass G geometries:
 * d_geom_pos [G][P][3] design positions and d_geom_row_param [G][Mc][8] row parameters
 * (see okx_rebind_design); pass NULL for both to use the program's own geometry.
 * Outputs: d_out_pos [B][n_out][3] solved positions of the output points
 * (SuspensionState.positions, state.py:119-126) and d_info [B].
 */
int32_t okx_solve_batch(okx_program* prog, const okx_solve_opts* opts, int64_t n_problems,
                        const double* d_targets,          /* [B][T] */
                        const double* d_geom_pos,         /* [G][P][3] or NULL */
                        const double* d_geom_row_param,   /* [G][Mc][8] or NULL */
                        double* d_out_pos,                /* [B][n_out][3] */
                        okx_info* d_info,                 /* [B] */
                        void* stream);

/*
 * Which kernel family and chain length okx_solve_batch (evaluated != 0: okx_solve_evaluated_batch) would use for a launch
 * of n_problems with these options, with (geometry_tables != 0) or without per-geometry tables; nothing is launched.
 * out2[0]: the okx_solve_opts.kernel value that forces the same family (1 interpreter, 2 packed interpreter, 3 quad, 4 lane);
 * out2[1]: the chain length the launch resolves chain / chain_len to (-1: the lane kernel's nested start mode, which sizes
 * itself from steps_per_geometry alone).  The selection depends on the problem count: a caller that cuts one batch into
 * several launches and needs the BITS of the single launch (the multi-GPU ensemble's chunks, dist.py) asks once for the
 * whole batch and passes the answer to every piece.  No reference counterpart (the reference has one code path).
 */
int32_t okx_plan_launch(okx_program* prog, const okx_solve_opts* opts, int64_t n_problems, int32_t geometry_tables,
                        int32_t evaluated, int32_t* out2);

/*
 * Replaces ResidualComputer.compute / compute_jacobian (solver.py:226-275, :502-581)
 * for B free-coordinate vectors: d_x [B][n] -> d_r [B][m], d_jac [B][m][n] (row-major,
 * dense).  d_jac may be NULL.  Used by the parity tests (rung R1).
 */
int32_t okx_eval_batch(okx_program* prog, int64_t n_problems,
                       const double* d_x,                 /* [B][n] */
                       const double* d_targets,           /* [B][T] */
                       double* d_r,                       /* [B][m] */
                       double* d_jac,                     /* [B][m][n] or NULL */
                       void* stream);

/*
 * Per-geometry problem emission on device (reference: topology constraints() computing
 * design lengths/angles/volumes from the initial state, e.g. double_wishbone.py:259-308,
 * track_rod.py:60-97, attachments.py:23-121).  d_hardpoints [G][P][3] holds design
 * positions of the fixed and free points (derived entries are ignored and recomputed).
 * Writes d_geom_pos [G][P][3] (derived points filled in) and d_geom_row_param [G][Mc][8]
 * with every row's design-state target recomputed (distance L, angle alpha, triple
 * product V and scale=|V|, point-on-line / line-pin anchor = the point's design position);
 * rows whose parameters are authored constants (fixed axis, planes) are copied through.
 */
int32_t okx_rebind_design(okx_program* prog, int64_t n_geometries,
                          const double* d_hardpoints,     /* [G][P][3] */
                          double* d_geom_pos,             /* [G][P][3] */
                          double* d_geom_row_param,       /* [G][Mc][8] */
                          void* stream);

/* Health of one tangent solve, the device analogue of TangentSolveInfo (sensitivity.py:44-55):
 * the pivots of the LDL^T of J^T J stand in for the singular values of J: they lie inside
 * [s_min^2, s_max^2], so sqrt(max_pivot / min_pivot) is a lower bound of J's condition number
 * and a non-positive or vanishing pivot signals rank deficiency. */
typedef struct okx_tangent_info {
  double min_pivot;       /* smallest pivot of the LDL^T of J^T J                          */
  double max_pivot;       /* largest pivot                                                 */
  int32_t flags;          /* bit0 factorisation succeeded, bit1 rank deficient
                             (min_pivot <= n * eps * max_pivot or a non-positive pivot)    */
  int32_t reserved;
} okx_tangent_info;

#define OKX_TANGENT_OK 1
#define OKX_TANGENT_RANK_DEFICIENT 2

/*
 * Replaces compute_state_tangents / compute_sweep_tangents (sensitivity.py:57-143,
 * sweep.py:113-141) for B solved states: with J the analytical Jacobian at the state
 * (constraint rows + target rows; the pinned line rows play the role of the reference's
 * _degenerate_constraint_pins, sensitivity.py:146-174), solves J q_t = e_t in the
 * least-squares sense for every target row t (q_t = (J^T J)^-1 J^T e_t) and propagates the
 * free-point velocities to every derived point in forward mode.  d_pos [B][n_out][3] are
 * solved positions as written by okx_solve_batch (every free point must be an output
 * point); d_tangents [B][T][n_out][3] receives d(point)/d(target t); fixed points get 0.
 * Geometry tables as in okx_solve_batch.  Programs with a quad kernel use its generated
 * tangent kernel (16 states per wavefront); all others the generic interpreter form (one
 * wavefront per state).
 */
int32_t okx_tangent_batch(okx_program* prog, int64_t n_problems, int64_t steps_per_geometry,
                          const double* d_pos,            /* [B][n_out][3] */
                          const double* d_geom_pos,       /* [G][P][3] or NULL */
                          const double* d_geom_row_param, /* [G][Mc][8] or NULL */
                          double* d_tangents,             /* [B][T][n_out][3] */
                          okx_tangent_info* d_tinfo,      /* [B] */
                          void* stream);

/*
 * Corner state metrics: pure functions of solved points, evaluated for B
 * states in one streaming launch.  Roles are indices into the OUTPUT point list of the
 * program the positions come from (Suspension.wheel_axis_points(), steering_axis_points(),
 * suspensions/corner/base.py role hooks).
 */
typedef struct okx_corner_roles {
  int32_t wheel_center, contact_patch;   /* PointID.WHEEL_CENTER, CONTACT_PATCH_CENTER        */
  int32_t axle_inboard, axle_outboard;   /* wheel_axis_points()                               */
  int32_t steer_lower, steer_upper;      /* steering_axis_points() (lower, upper pivot)       */
  /* Instant axis of the upright (compute_instant_axis): OKX_IA_TWO_PLANES = upper and lower wishbone
   * planes, points (upper front, upper rear, upper outboard, lower front, lower rear, lower outboard)
   * (corner/double_wishbone.py:376-403); OKX_IA_PLANE_AND_STRUT = lower-arm plane and the plane through
   * the strut top normal to the strut, points (arm front, arm rear, ball joint, strut top, -, -)
   * (corner/macpherson.py:325-355); OKX_IA_NONE: the instant-centre family reads NaN. */
  int32_t instant_axis_kind;
  int32_t instant_axis_point[6];
  int32_t damper_top, damper_bottom;     /* damper_points() or -1, -1 (travel.py:48-62)        */
  int32_t rack_attachment;               /* rack_attachment_point() or -1 (axle_metrics.py:60-70) */
  int32_t axle_position, driven_axle;    /* OKX_AXLE_UNSET / _FRONT / _REAR (schema/config.py:79-80,140) */
  double side_sign;                      /* +1 left, -1 right (Side.lateral_sign)             */
  double design_wheel_center_z;          /* travel reference (metrics/context.py:42-45)       */
  double design_contact_patch_z;         /* ride-height reference (axle_metrics.py:33-37)     */
  double design_rack_y;                  /* rack reference (axle_metrics.py:66-69)            */
  double wheelbase, cg_z;                /* config.wheelbase, config.cg_position.z            */
  double front_brake_bias;               /* config.front_brake_bias, NaN when unset           */
} okx_corner_roles;

enum { OKX_IA_NONE = 0, OKX_IA_TWO_PLANES = 1, OKX_IA_PLANE_AND_STRUT = 2 };
enum { OKX_AXLE_UNSET = 0, OKX_AXLE_FRONT = 1, OKX_AXLE_REAR = 2 };

/* The reference's corner metric catalog (metrics/catalog.py:46-146).  A metric the reference
 * reports as None (undefined geometry, unset configuration) reads NaN. */
enum {
  OKX_METRIC_CAMBER = 0,           /* deg, metrics/angles.py:22-50            */
  OKX_METRIC_CASTER = 1,           /* deg, angles.py:53-71                    */
  OKX_METRIC_KPI = 2,              /* deg, angles.py:74-94                    */
  OKX_METRIC_ROADWHEEL_ANGLE = 3,  /* deg (= toe), angles.py:97-132           */
  OKX_METRIC_WHEEL_TRAVEL = 4,     /* mm, travel.py:19-32                     */
  OKX_METRIC_HALF_TRACK = 5,       /* mm, travel.py:35-45                     */
  OKX_METRIC_SCRUB_RADIUS = 6,     /* mm, steering_geometry.py:22-54          */
  OKX_METRIC_MECHANICAL_TRAIL = 7, /* mm, steering_geometry.py:57-76          */
  OKX_METRIC_SVIC_X = 8,           /* mm, side-view instant centre, catalog.py:95-100 */
  OKX_METRIC_SVIC_Z = 9,
  OKX_METRIC_SVSA_LENGTH = 10,     /* mm, swing_arms.py:45-59                 */
  OKX_METRIC_FVIC_Y = 11,          /* mm, front-view instant centre, catalog.py:104-109 */
  OKX_METRIC_FVIC_Z = 12,
  OKX_METRIC_FVSA_LENGTH = 13,     /* mm, signed, swing_arms.py:62-88         */
  OKX_METRIC_DAMPER_LENGTH = 14,   /* mm, travel.py:48-62                     */
  OKX_METRIC_SVSA_ANGLE = 15,      /* deg, anti_geometry.py:32-58             */
  OKX_METRIC_ANTI_DIVE = 16,       /* %, anti_geometry.py:75-116              */
  OKX_METRIC_ANTI_LIFT = 17,       /* %, anti_geometry.py:119-160             */
  OKX_METRIC_ANTI_SQUAT = 18,      /* %, anti_geometry.py:163-206             */
  OKX_METRIC_COUNT = 19
};

/*
 * Replaces compute_metrics_for_state's catalog entries (metrics/main.py:150-185, catalog.py) and,
 * given the tangents of okx_tangent_batch, the raw material of the derivative columns
 * (metrics/derivatives.py): d_dmetrics[b][t][m] = d metric_m / d target_t,
 * e.g. deriv_camber_wrt_hub_z = d_dmetrics[.][bump target][OKX_METRIC_CAMBER],
 * deriv_damper_length_wrt_hub_z likewise, and
 * deriv_wheel_center_x_wrt_hub_z = d_tangents[.][bump target][wheel_center][0].
 * (The instant-centre family has no derivative column in the reference; its entries are the
 * forward-mode derivatives of the same formulas.)  d_tangents / d_dmetrics may both be NULL.
 */
int32_t okx_corner_metrics_batch(const okx_corner_roles* roles, int64_t n_states, int32_t n_out, int32_t n_targets,
                                 const double* d_pos,       /* [B][n_out][3] */
                                 const double* d_tangents,  /* [B][T][n_out][3] or NULL */
                                 double* d_metrics,         /* [B][OKX_METRIC_COUNT] */
                                 double* d_dmetrics,        /* [B][T][OKX_METRIC_COUNT] or NULL */
                                 void* stream);

/* Axle-scope state metrics of a solved two-corner axle (metrics/axle_metrics.py:21-95). */
enum {
  OKX_AXLE_METRIC_HEAVE = 0,               /* mm, mean wheel-centre rise                        */
  OKX_AXLE_METRIC_ROLL = 1,                /* deg, atan2(left - right rise, track)              */
  OKX_AXLE_METRIC_RIDE_HEIGHT_CHANGE = 2,  /* mm, -mean contact-patch rise                      */
  OKX_AXLE_METRIC_TRACK = 3,               /* mm, |CP_y left - CP_y right|                      */
  OKX_AXLE_METRIC_ROLL_CENTER_Y = 4,       /* mm, the two contact-patch -> FVIC lines meet here */
  OKX_AXLE_METRIC_ROLL_CENTER_Z = 5,
  OKX_AXLE_METRIC_RACK_DISPLACEMENT = 6,   /* mm, left rack pickup y - design (NaN: no rack)    */
  OKX_AXLE_METRIC_COUNT = 7
};

/* left / right: roles of the two corners as indices into the AXLE program's output points. */
int32_t okx_axle_metrics_batch(const okx_corner_roles* left, const okx_corner_roles* right, int64_t n_states,
                               int32_t n_out,
                               const double* d_pos,  /* [B][n_out][3] */
                               double* d_metrics,    /* [B][OKX_AXLE_METRIC_COUNT] */
                               void* stream);

/*
 * The EVALUATED solve (SURVEY.md section 8f.1-2 as ONE launch).  Replaces solve_evaluated_sweep / evaluate_solved_sweep
 * (core/sweep.py:217-270: solve -> compute_sweep_tangents -> compute_sweep_metrics; diagnostics excluded) for a batch:
 * the generated solve kernels end every problem with an epilogue that, at the converged state still in registers,
 * evaluates the Jacobian once more, factors the undamped J^T J, substitutes once per target for the solution-manifold
 * tangent (sensitivity.py:57-143) and evaluates the corner metric catalog and its derivatives along every tangent
 * (metrics/catalog.py, metrics/derivatives.py) - no position records are re-read, no tangent tensor is materialised
 * unless asked for.
 *
 * d_eval [B][1 + T][OKX_EVAL_COLUMNS] receives, per problem,
 *   row 0      : columns 0 .. OKX_METRIC_COUNT-1 the metric values (NaN where the reference reports None),
 *                OKX_EVAL_MIN_PIVOT / _MAX_PIVOT / _TANGENT_FLAGS the health of the tangent solve (okx_tangent_info's
 *                fields; the flags as a double), the rest 0;
 *   row 1 + t  : columns 0 .. OKX_METRIC_COUNT-1 d metric / d target t, OKX_EVAL_RATE_WHEEL_CENTER_X .. _Z the wheel centre's
 *                velocity along that tangent, OKX_EVAL_RATE_RACK_Y the rack pickup's lateral velocity (NaN: no rack) - the
 *                driver rates the reference's deriv_<response>_wrt_<driver> columns divide by (derivatives.py:265-320).
 * The role POINTS of `roles` are compiled into the program's evaluated kernels (okx_program_enable_evaluation: generated
 * and compiled like the solve kernels, kept in the same cache; a miss compiles in place, 10 ... 60 s - okx_precompile_evaluation
 * fills the cache ahead of time); its numbers (side sign, vehicle data, design references) are kernel arguments and may
 * change from call to call through okx_program_enable_evaluation at no cost.  With geometry tables the design references
 * (wheel travel, ride height, rack displacement) are each geometry's own design state, taken from d_geom_pos.
 * Programs with a single-mode quad kernel (corners: up to 9 free points, every free point an output point, at least one
 * target); okx_solve_opts.output still says what goes to d_out_pos (OKX_OUTPUT_NONE: nothing - metrics only).
 */
#define OKX_EVAL_COLUMNS 24
enum {
  OKX_EVAL_MIN_PIVOT = 19, OKX_EVAL_MAX_PIVOT = 20, OKX_EVAL_TANGENT_FLAGS = 21,                 /* row 0 */
  OKX_EVAL_RATE_WHEEL_CENTER_X = 19, OKX_EVAL_RATE_WHEEL_CENTER_Y = 20, OKX_EVAL_RATE_WHEEL_CENTER_Z = 21,
  OKX_EVAL_RATE_RACK_Y = 22                                                                      /* rows 1 + t */
};
int32_t okx_program_enable_evaluation(okx_program* prog, const okx_corner_roles* roles);
/* bit0: the program has evaluated kernels for the roles last enabled; bit1: a lane form exists as well. */
int32_t okx_program_evaluation(const okx_program* prog);
const char* okx_program_evaluation_note(const okx_program* prog);
/* okx_solve_batch with the epilogue: d_tangents [B][T][n_out][3] and d_eval may each be NULL (not both). */
int32_t okx_solve_evaluated_batch(okx_program* prog, const okx_solve_opts* opts, int64_t n_problems,
                                  const double* d_targets, const double* d_geom_pos, const double* d_geom_row_param,
                                  double* d_out_pos, okx_info* d_info, double* d_tangents, double* d_eval, void* stream);
/* The same epilogue on given solved states d_pos [B][n_out][3] (evaluate_solved_sweep, core/sweep.py:217-245). */
int32_t okx_evaluate_batch(okx_program* prog, int64_t n_problems, int64_t steps_per_geometry, const double* d_pos,
                           const double* d_geom_pos, const double* d_geom_row_param, double* d_tangents, double* d_eval,
                           void* stream);
/* Generate + compile a program's evaluated kernels for `roles` into the on-disk cache (no device needed). */
int32_t okx_precompile_evaluation(const okx_program_desc* desc, const okx_corner_roles* roles);

/*
 * Signed rotation of output points about fixed axes from their design positions, in degrees
 * (metrics/kernels.py:58-76 rotation_about_fixed_axis_deg, vector_utils/geometric.py:31-52): the
 * topology-specific state metrics rocker_angle / torsion_bar_twist (corner/mechanisms.py:378-407,611-623:
 * PUSHROD_INBOARD about the rocker axis, times Side.lateral_sign) and arb_arm_angle / arb_twist
 * (axle/mechanisms.py:402-430: DROPLINK_U_BAR of each side about the U-bar axis; twist = left - right),
 * and with tangents their derivative columns (deriv_rocker_angle_wrt_hub_z, deriv_arb_twist_wrt_hub_z_*).
 * A point on its axis (either perpendicular below 1e-6) reads NaN where the reference raises.
 */
#define OKX_MAX_ROTATIONS 8
/* What a role measures (`kind`).  0 is the rotation described above; the others are the state metrics of an axle's
 * shared hardware that are not rotations of one point about a fixed axis (axle/mechanisms.py:718-815, :903-944),
 * evaluated by the same kernel, with the same derivative columns:
 *   1  rotation about the fixed axis of the MIDPOINT of output points `point` and `point_b` from `design` (a rigid T-bar's
 *      t_bar_heave_angle: crossbar midpoint about the pivot's lateral axis, mechanisms.py:763-798), degrees x scale;
 *   2  twist of the segment `point` - `point_b` about the MOVING stem from `axis_point` (the pivot) to the segment's
 *      midpoint, measured from `axis_dir` (the lateral direction) and taken relative to `design[0]` (the design twist in
 *      degrees): a T-bar's arb_twist (mechanisms.py:800-815), degrees;
 *   3  distance between `point` and `point_b` (a rocker-to-rocker heave link's heave_link_length, mechanisms.py:934-944), mm;
 *   4  coordinate of the midpoint of `point` and `point_b` along `axis_dir` from `axis_point` (t_bar_center_x and its
 *      rate, mechanisms.py:718-761), mm. */
#define OKX_ROLE_AXIS_ROTATION 0
#define OKX_ROLE_MIDPOINT_ROTATION 1
#define OKX_ROLE_STEM_TWIST 2
#define OKX_ROLE_DISTANCE 3
#define OKX_ROLE_MIDPOINT_COORDINATE 4
typedef struct okx_rotation_role {
  int32_t point;         /* output-point index of the moving pickup   */
  int32_t point_b;       /* second output point (kinds 1-4; unused by kind 0) */
  double design[3];      /* its design position (kind 2: design[0] = the design twist in degrees) */
  double axis_point[3];  /* a point on the fixed axis                  */
  double axis_dir[3];    /* unit direction of the axis                 */
  double scale;          /* Side.lateral_sign, or 1                    */
  int32_t kind;          /* OKX_ROLE_* */
  int32_t reserved;
} okx_rotation_role;

int32_t okx_axis_rotation_batch(const okx_rotation_role* roles, int32_t n_roles, int64_t n_states, int32_t n_out,
                                int32_t n_targets,
                                const double* d_pos,       /* [B][n_out][3] */
                                const double* d_tangents,  /* [B][T][n_out][3] or NULL */
                                double* d_angles,          /* [B][n_roles] */
                                double* d_dangles,         /* [B][T][n_roles] or NULL */
                                void* stream);

/*
 * The evaluated solve of a COMPOSED AXLE (two identical corners joined by rack / anti-roll bar / heave-link rows; the
 * generated kernels' pair mode): okx_solve_evaluated_batch / okx_evaluate_batch on a program enabled with
 * okx_program_enable_axle_evaluation end every problem with ONE epilogue for the whole axle - one undamped refactorisation
 * through the halves' factors and the joining rows' Woodbury system, one substitution per target (sensitivity.py:57-174,
 * the partner half's share exchanged inside the wavefront), derived-point velocities, then
 *   - BOTH corners' metric catalogs with their derivative along every target's tangent (metrics/main.py:63-185 per side),
 *   - the axle-scope metrics (metrics/axle_metrics.py:21-95) and their derivatives,
 *   - up to OKX_MAX_ROTATIONS roles of okx_rotation_role's kinds (rocker angles, U-bar arm angles, a T-bar's heave angle,
 *     twist and centre travel, a heave link's length: corner/mechanisms.py:378-407, axle/mechanisms.py:402-430,718-815,903-944)
 * - what solve -> okx_tangent_batch -> okx_corner_metrics_batch x 2 -> okx_axle_metrics_batch -> okx_axis_rotation_batch
 * compute in six launches.  Reference: core/sweep.py:113-173,217-270 for an AxleSuspension.
 * d_eval is [B][1 + T][OKX_EVAL_AXLE_COLUMNS]; T = the PROGRAM's targets.  Per row:
 *   columns OKX_EVAL_AXLE_LEFT  + 0 .. 23 : the left corner's block in the corner layout above (row 0: metric values, the
 *                                            tangent solve's pivots and flags; row 1 + t: derivatives, wheel-centre and rack rates),
 *   columns OKX_EVAL_AXLE_RIGHT + 0 .. 23 : the right corner's (its pivot columns are 0),
 *   columns OKX_EVAL_AXLE_METRICS + OKX_AXLE_METRIC_* : the axle-scope metrics (row 0) / their derivatives (row 1 + t),
 *   columns OKX_EVAL_AXLE_ROLES + k       : role k's value (row 0) / its rate along target t's tangent (row 1 + t).
 * The role POINTS (both corners' and the roles' `point` / `point_b`, and the roles' kinds) are compiled into the module;
 * every number (side signs, design references, vehicle data, the roles' axes, design positions and scales) is a kernel
 * argument.  The two corners must use the same instant-axis construction and both or neither carry damper / rack roles.
 */
#define OKX_EVAL_AXLE_COLUMNS 64
enum { OKX_EVAL_AXLE_LEFT = 0, OKX_EVAL_AXLE_RIGHT = 24, OKX_EVAL_AXLE_METRICS = 48, OKX_EVAL_AXLE_ROLES = 56 };
typedef struct okx_axle_roles {
  okx_corner_roles left, right;  /* indices into the AXLE program's output points */
  int32_t n_roles;               /* 0 .. OKX_MAX_ROTATIONS */
  int32_t reserved;
  okx_rotation_role roles[OKX_MAX_ROTATIONS];
} okx_axle_roles;
int32_t okx_program_enable_axle_evaluation(okx_program* prog, const okx_axle_roles* roles);
int32_t okx_precompile_axle_evaluation(const okx_program_desc* desc, const okx_axle_roles* roles);
/* Columns per row of d_eval for the evaluation last enabled: OKX_EVAL_COLUMNS (corner), OKX_EVAL_AXLE_COLUMNS (axle), 0 (none). */
int32_t okx_program_eval_columns(const okx_program* prog);

/*
 * Output positions from free-point coordinates: fixed points come from the program's design state (or the
 * per-geometry table of okx_rebind_design), derived points are re-evaluated (DerivedPointsManager.update,
 * points/derived/manager.py).  The receiving side of a multi-GPU exchange that ships the 3 n_free free
 * coordinates of each solve instead of its 3 n_out output coordinates; also turns stored free vectors
 * (SuspensionState.get_free_array) back into full states.
 */
int32_t okx_expand_positions_batch(okx_program* prog, int64_t n_problems, int64_t steps_per_geometry,
                                   const double* d_free,      /* [B][n_free][3], free_point order */
                                   const double* d_geom_pos,  /* [G][n_points][3] or NULL */
                                   double* d_out_pos,         /* [B][n_out][3] */
                                   void* stream);

/*
 * Chain-head predictor.  The reference warm-starts step k from step k-1 (solver.py:774) and starts a
 * sweep at the design state; a batch that fills the chip solves every step as an independent cold
 * start instead.  okx_program_fit_predictor solves the program once at the Chebyshev
 * nodes of a target box [lo, hi] (host arrays of n_targets absolute target values; lo[t] == hi[t]
 * holds target t), i.e. (degree + 1)^d cold solves for d varying targets in one synchronous launch,
 * and keeps the Chebyshev coefficients of every free coordinate up to total degree `degree` (1..12;
 * <= 0: 7).  Launches with opts.predictor != 0 on the program's own geometry start every chain head at that
 * polynomial (targets clamped to the box).  The LM iteration, its stopping rules and the results (to
 * step_tol) are unchanged; only the number of passes drops.  Needs the program's quad kernel and every
 * free point among the output points.  Fitting again replaces the model.
 */
int32_t okx_program_fit_predictor(okx_program* prog, const double* lo, const double* hi, int32_t degree, void* stream);
int32_t okx_program_has_predictor(const okx_program* prog);

/*
 * Camber-shim setup solve: the pose a double-wishbone corner takes when the split
 * upright's shim stack is changed from its design to its setup thickness, for n_geometries
 * geometries at once.  Replaces solve_camber_shim_assembly (suspensions/config/shims.py:284-501: 7 or 8
 * variables, 10 or 11 residuals, scipy MINPACK `lm`) and DoubleWishboneSuspension.apply_camber_shim
 * (corner/double_wishbone.py:501-570): the upper ball joint moves on the upper wishbone's arc, the
 * upright's attachments rotate about the lower ball joint, an upright-mounted pushrod turns the
 * rocker group.  Indices are rows of the point table the positions are given in (the program's
 * point list: okx_program_desc.design_pos).  The table is rewritten in place and can go straight
 * into okx_rebind_design.
 */
#define OKX_SHIM_MAX_POINTS 8
#define OKX_SHIM_PARAMS 11 /* per geometry: face point a (3), face point b (3), face normal (3), design, setup thickness
                              (CamberShimConfig, schema/config.py:52-70) */
typedef struct okx_shim_roles {
  int32_t upper_outboard, lower_outboard;            /* UPPER_/LOWER_WISHBONE_OUTBOARD (ball joints)      */
  int32_t upper_inboard_front, upper_inboard_rear;   /* upper wishbone axis (shims.py:386-390)            */
  int32_t heading_inboard, heading_outboard;         /* installed track rod / toe link (shims.py:399-403) */
  int32_t n_upright_points;                          /* upright_attachment_points() (double_wishbone.py:572-581) */
  int32_t upright_point[OKX_SHIM_MAX_POINTS];
  int32_t rocker;                                    /* 1: upright-mounted pushrod (CamberShimRockerCoupling, shims.py:49-56) */
  int32_t rocker_axis_a, rocker_axis_b, pushrod_inboard, pushrod_outboard;
  int32_t n_rocker_points;                           /* rocker group (mechanisms.py:247-265)              */
  int32_t rocker_point[OKX_SHIM_MAX_POINTS];
} okx_shim_roles;

typedef struct okx_shim_info {
  double residual_norm;       /* CamberShimAssemblySolution.constraint_residual_norm             */
  double max_residual;
  double upright_angle_rad;   /* upright_body_rot_angle_rad                                      */
  double rocker_angle_rad;
  double wishbone_angle_rad;
  int32_t converged;          /* max residual <= SOLVE_ACCEPT_RESIDUAL (shims.py:456-462)        */
  int32_t iterations;
} okx_shim_info;

int32_t okx_camber_shim_batch(const okx_shim_roles* roles, int64_t n_geometries, int32_t n_points,
                              double* d_points,        /* [G][n_points][3] in: authored, out: setup */
                              const double* d_shim,    /* [G][OKX_SHIM_PARAMS] */
                              okx_shim_info* d_info,   /* [G] or NULL */
                              void* stream);

/*
 * Runtime specialisation.  okx_program_create also GENERATES a HIP kernel for the program at
 * hand (straight-line residual / Jacobian / normal-equation / LDL^T code, four lanes per
 * problem), compiles it with hiprtc and loads it; the reference does the analogous thing
 * offline for its Jacobian rows (tools/generate_jacobians.py -> core/jacobians.py) and
 * interprets the rest per call (solver.py:226-275, :502-581).  Code objects are cached under
 * <directory of libokx.so>/_kcache (override: OKX_KERNEL_CACHE), keyed by the generated source.
 * Programs the generator has no code path for (more than 9 free points and no pair structure, unsupported row or
 * derived-point types) keep the generic interpreter kernels; nothing else changes for them.
 */

/* "quad" when the specialised kernel of this program is loaded, "wave" when the generic
 * interpreter kernels are in charge; okx_program_kernel_note() then says why (static strings
 * owned by the program). */
const char* okx_program_kernel(const okx_program* prog);
const char* okx_program_kernel_note(const okx_program* prog);
/* 1 when chain heads of the program's own geometry take their first Levenberg-Marquardt step from the shared first-step
 * table (okx_solve_opts.shared_first_step): okx_info.nfev of such a head does not count the design-state evaluation,
 * which was made once for all of them (the drop-in's SolverInfo.nfev adds it back, solver.py:766-771). */
int32_t okx_program_shares_first_step(const okx_program* prog);
/* 1 when launches of INDEPENDENT solves (chain length 1) on the program's own geometry with the shared first step run the
 * quad kernel's cold body `okx_quad_cold_u` (no chain history, no fitted model, no trace: tables staged through LDS, the
 * passes of an all-accepted solve under one exec mask, anything else redone through the general loop - same answers, bit
 * for bit, as `okx_quad_solve_u`).  What bench.py names as the dominant kernel of the headline launch. */
int32_t okx_program_has_cold_body(const okx_program* prog);
/* Tiered start.  okx_program_create loads a program's generated kernels when the kernel cache holds them (what
 * okx_precompile and __graft_entry__.build() are for).  When it does not, the call still returns at once: a host thread
 * runs the compiler (10 ... 80 s per module; it owns its inputs, so destroying the program does not wait for it) while the
 * interpreter kernels solve - same answers to 1e-9 mm, 20 ... 100 x slower - and the first launching call after the job
 * has finished switches the program over: it loads the job's code objects from memory (a few milliseconds on the calling
 * thread; never the compiler, never a synchronise of the legacy stream) while holding the program's kernel state
 * exclusively - launches of the same program from other threads wait for it, launches in progress finish first.  A call
 * whose stream is recording a HIP graph never switches over (module loads are illegal in a capture): the interpreter
 * serves it and a later call does.  The switch-over has two stages: the quad module is attached as soon as IT is compiled
 * (it serves every batch size), the lane module when the whole job is done.  okx_program_ready returns 1 when nothing is
 * pending any more and 0 while the job
 * runs; wait != 0 blocks until it has finished and switches over before returning (benchmarks and tests that must know
 * which kernel they time).  okx_program_kernel() / okx_program_kernel_note() report the current state. */
int32_t okx_program_ready(okx_program* prog, int32_t wait);

/* Generated source of a program's quad kernel (no device needed).  Copies at most buflen - 1
 * bytes plus a terminator into buf (buf may be NULL) and returns the size the full text
 * needs, or a negative okx_status (OKX_ERR_LIMIT: no quad kernel for this program). */
int64_t okx_quad_source(const okx_program_desc* desc, char* buf, int64_t buflen);

/* Generate + compile a program's quad kernel (and its lane kernel, when it has one) into the on-disk
 * cache (no device needed), so that a later okx_program_create only loads them. */
int32_t okx_precompile(const okx_program_desc* desc);

/*
 * The second generated kernel family: ONE LANE PER PROBLEM, 64 problems per wavefront, no cross-lane
 * operand at all (the quad kernel moves every dot product, J^T J column and pivot through DPP and idles
 * one lane in four).  About a quarter of the quad kernel's instructions per problem, but one wavefront
 * holds 64 problems and takes 25 ... 29 us whatever it holds: it pays as soon as the quad kernel (16
 * problems per wavefront, ~21 us per round) needs a second round - from okx_program_lane_threshold()
 * problems on, 16385 on an MI355X - unless an ensemble has so few steps per geometry that its lanes
 * would idle (a wave unit holds problems of ONE geometry).  Programs with at most 6 free points
 * (n <= 18) whose quad kernel is loaded have one; okx_program_lane_note() says why another has not.
 * Same algorithm, same evaluation points, same first-step tables; okx_solve_opts.kernel = 4 forces it.
 */
const char* okx_program_lane_note(const okx_program* prog);
int64_t okx_program_lane_threshold(const okx_program* prog);   /* -1: no lane kernel */
/* Which bodies auto selection uses: bit0 independent solves (chain_len 1), bit1 chains.  A body whose register
 * allocation spills more than a little stays with the quad kernel (okx_solve_opts.kernel = 4 still runs it). */
int32_t okx_program_lane_bodies(const okx_program* prog);
int64_t okx_lane_source(const okx_program_desc* desc, char* buf, int64_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* OKX_H */
