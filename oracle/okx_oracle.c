/*
 * okx_oracle.c — CPU restatement of the reference's per-sweep-step solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (open_kinematics_amd/) never
 * does and fails loudly when the HIP extension is missing.
 *
 * What is restated, and from where (paths relative to /root/reference/src/kinematics/core):
 *   residual rows        constraints.py:125-134,162-170,223-243,287-308,351-371,414-429,
 *                        466-477,508-516,560-576,616-627,657-666,698-709,731-733
 *   Jacobian rows        jacobians.py:35-51,55-122,127-188,192-262,266-318,322-367,372-403,
 *                        408-422,426-483 (the CSE bodies emitted by tools/generate_jacobians.py)
 *   softnorm             primitives/soft_math.py:16-27
 *   derived points       points/derived/definitions.py:24-33,36-73,76-180; their 3x3 chain
 *                        blocks replace the dual-number pass of manager.py:271-324
 *   residual/Jacobian    solver.py:226-275 (ResidualComputer.compute), :502-581
 *   assembly             (compute_jacobian), target rows :264-270,:560-579
 *   sweep driver         solver.py:654-776 (solve_suspension_sweep): sequential warm start,
 *                        success test, residual_tolerance acceptance, nfev / max_residual
 *   LM iteration         THIRD PARTY, not under /root/reference: SciPy (uv.lock pins
 *                        scipy 1.14.1; floor scipy>=1.14.1 in pyproject.toml:7)
 *                        scipy.optimize.least_squares(method="lm") -> MINPACK lmder
 *                        (More', Garbow, Hillstrom 1980).  lmder/lmpar/qrfac/qrsolv/enorm
 *                        below restate the published MINPACK algorithm with SciPy's call
 *                        conventions (solver.py:158-169: factor=100, diag=1/x_scale=1
 *                        i.e. mode 2, maxfev=100*n, analytical Jacobian).
 *
 * Pinning: oracle/gen_golden.py runs the real reference (imported via oracle/ref_shim.py)
 * and commits its inputs/outputs under tests/golden/; tests/test_oracle.py checks this file
 * against them (residuals/Jacobians to 1e-13 relative, solved positions per the ladder in
 * DESIGN.md).
 */
#include "../include/okx.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS_GEOM 1e-6
#define EPS_SQ (EPS_GEOM * EPS_GEOM) /* soft_math.py:19 */

static double softnorm(double s) { return sqrt(s + EPS_SQ) - EPS_GEOM; }

/* ------------------------------------------------------------------ */
/* Evaluation workspace                                                */
/* ------------------------------------------------------------------ */

typedef struct {
  const okx_program_desc* p;
  int n, m;
  int var_of_point[OKX_MAX_POINTS]; /* 3k or -1 */
  int dop_of_point[OKX_MAX_POINTS]; /* derived-op index or -1 */
  double pos[OKX_MAX_POINTS][3];
  /* d(derived point)/d(x): [D][3][n] dense */
  double* dder;
  const double* row_param; /* [Mc][8] current geometry */
} eval_ws;

static int ws_init(eval_ws* w, const okx_program_desc* p) {
  if (p->n_points > OKX_MAX_POINTS || 3 * p->n_free > OKX_MAX_VARS ||
      p->n_rows + p->n_targets > OKX_MAX_ROWS || p->n_targets > OKX_MAX_TARGETS)
    return OKX_ERR_LIMIT;
  w->p = p;
  w->n = 3 * p->n_free;
  w->m = p->n_rows + p->n_targets;
  for (int i = 0; i < p->n_points; ++i) {
    w->var_of_point[i] = -1;
    w->dop_of_point[i] = -1;
  }
  for (int k = 0; k < p->n_free; ++k) w->var_of_point[p->free_point[k]] = 3 * k;
  for (int d = 0; d < p->n_derived; ++d) w->dop_of_point[p->dop_out[d]] = d;
  w->dder = (double*)calloc((size_t)(p->n_derived > 0 ? p->n_derived : 1) * 3 * (size_t)w->n,
                            sizeof(double));
  w->row_param = p->row_param;
  memcpy(w->pos, p->design_pos, sizeof(double) * 3 * (size_t)p->n_points);
  return w->dder ? OKX_OK : OKX_ERR_ALLOC;
}

static void ws_free(eval_ws* w) {
  free(w->dder);
  w->dder = NULL;
}

static void v_sub(const double* a, const double* b, double* o) {
  o[0] = a[0] - b[0];
  o[1] = a[1] - b[1];
  o[2] = a[2] - b[2];
}
static double v_dot(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void v_cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* normalize_vector (vector_utils/generic.py:135-171): v / ||v||. */
static double v_normalize(const double* v, double* u) {
  double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  u[0] = v[0] / nrm;
  u[1] = v[1] / nrm;
  u[2] = v[2] / nrm;
  return nrm;
}

/* Accumulate dst[3][n] += B(3x3) * d(point)/dx, where the point may be fixed, free, derived. */
static void chain_add(const eval_ws* w, double* dst, const double B[3][3], int point) {
  const int n = w->n;
  int v = w->var_of_point[point];
  if (v >= 0) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) dst[r * n + v + c] += B[r][c];
    return;
  }
  int d = w->dop_of_point[point];
  if (d < 0) return; /* fixed */
  const double* src = w->dder + (size_t)d * 3 * n;
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) {
      double b = B[r][k];
      if (b == 0.0) continue;
      for (int j = 0; j < n; ++j) dst[r * n + j] += b * src[k * n + j];
    }
}

/* d normalize(v) / dv = (I - u u^T) / |v| */
static void normalize_block(const double* u, double nrm, double B[3][3]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) B[r][c] = ((r == c ? 1.0 : 0.0) - u[r] * u[c]) / nrm;
}

/* Derived points in program order (manager.py:186-197) and, if with_jac, their chain blocks. */
static void update_derived(eval_ws* w, int with_jac) {
  const okx_program_desc* p = w->p;
  const int n = w->n;
  for (int d = 0; d < p->n_derived; ++d) {
    const int32_t* pt = p->dop_pts + 4 * d;
    double* out = w->pos[p->dop_out[d]];
    double* dd = w->dder + (size_t)d * 3 * n;
    const double c = p->dop_param[d];
    if (with_jac) memset(dd, 0, sizeof(double) * 3 * (size_t)n);
    switch (p->dop_type[d]) {
      case OKX_DOP_MIDPOINT: { /* definitions.py:76-89: p1 + (p2 - p1)/2 */
        const double* a = w->pos[pt[0]];
        const double* b = w->pos[pt[1]];
        for (int k = 0; k < 3; ++k) out[k] = a[k] + (b[k] - a[k]) / 2;
        if (with_jac) {
          double H[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 0.5}};
          chain_add(w, dd, H, pt[0]);
          chain_add(w, dd, H, pt[1]);
        }
      } break;
      case OKX_DOP_ALONG: { /* definitions.py:24-33,92-155: base + normalize(a - b) * c */
        const double* base = w->pos[pt[0]];
        double v[3], u[3];
        v_sub(w->pos[pt[1]], w->pos[pt[2]], v);
        double nrm = v_normalize(v, u);
        double o0 = base[0] + u[0] * c, o1 = base[1] + u[1] * c, o2 = base[2] + u[2] * c;
        out[0] = o0;
        out[1] = o1;
        out[2] = o2;
        if (with_jac) {
          double I3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
          double B[3][3], Bn[3][3];
          normalize_block(u, nrm, B);
          for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) {
              B[r][k] *= c;
              Bn[r][k] = -B[r][k];
            }
          chain_add(w, dd, I3, pt[0]);
          chain_add(w, dd, B, pt[1]);
          chain_add(w, dd, Bn, pt[2]);
        }
      } break;
      case OKX_DOP_CONTACT_PATCH: { /* definitions.py:36-73,158-180 */
        const double* wc = w->pos[pt[0]];
        double v[3], a[3], wd[3], wu[3];
        v_sub(w->pos[pt[2]], w->pos[pt[1]], v); /* axle_outboard - axle_inboard */
        double vn = v_normalize(v, a);
        const double g[3] = {-0.0, -0.0, -1.0}; /* -1 * WorldAxisSystem.Z */
        double ga = v_dot(g, a);
        for (int k = 0; k < 3; ++k) wd[k] = g[k] - ga * a[k];
        double wn = v_normalize(wd, wu);
        for (int k = 0; k < 3; ++k) out[k] = wc[k] + wu[k] * c;
        if (with_jac) {
          double I3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
          double Na[3][3], Nw[3][3], Wa[3][3], M[3][3], T[3][3], Tn[3][3];
          normalize_block(a, vn, Na);  /* da/dv */
          normalize_block(wu, wn, Nw); /* dwu/dwd */
          /* dwd/da = -(a g^T) - (g.a) I */
          for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) Wa[r][k] = -a[r] * g[k] - (r == k ? ga : 0.0);
          for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) {
              double s = 0;
              for (int q = 0; q < 3; ++q) s += Nw[r][q] * Wa[q][k];
              M[r][k] = s;
            }
          for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) {
              double s = 0;
              for (int q = 0; q < 3; ++q) s += M[r][q] * Na[q][k];
              T[r][k] = s * c;
              Tn[r][k] = -T[r][k];
            }
          chain_add(w, dd, I3, pt[0]);
          chain_add(w, dd, T, pt[2]);
          chain_add(w, dd, Tn, pt[1]);
        }
      } break;
      default:
        break;
    }
  }
}

static void set_free(eval_ws* w, const double* x) {
  const okx_program_desc* p = w->p;
  for (int k = 0; k < p->n_free; ++k) {
    double* q = w->pos[p->free_point[k]];
    q[0] = x[3 * k];
    q[1] = x[3 * k + 1];
    q[2] = x[3 * k + 2];
  }
}

/* One constraint row: residual and partials w.r.t. its (up to 4) points. */
static double row_eval(const eval_ws* w, int i, double dp[12], int* npts) {
  const okx_program_desc* p = w->p;
  const int32_t* pt = p->row_pts + 4 * i;
  const double* q = w->row_param + OKX_ROW_PARAMS * i;
  const int type = p->row_type[i];
  memset(dp, 0, sizeof(double) * 12);
  switch (type) {
    case OKX_ROW_DISTANCE:
    case OKX_ROW_SPHERICAL: { /* constraints.py:125-134 / jacobians.py:35-51 */
      double d[3];
      v_sub(w->pos[pt[1]], w->pos[pt[0]], d);
      double s = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      double inv = 1.0 / sqrt(EPS_SQ + s);
      for (int k = 0; k < 3; ++k) {
        dp[k] = -d[k] * inv;
        dp[3 + k] = d[k] * inv;
      }
      *npts = 2;
      return type == OKX_ROW_DISTANCE ? softnorm(s) - q[0] : softnorm(s);
    }
    case OKX_ROW_ANGLE:
    case OKX_ROW_THREE_POINT_ANGLE: {
      /* constraints.py:223-243,287-308 / jacobians.py:55-122,127-188.
         theta = atan2(sqrt(|c|^2 + eps^2), v1.v2), c = v1 x v2. */
      double v1[3], v2[3], c[3];
      if (type == OKX_ROW_ANGLE) {
        v_sub(w->pos[pt[1]], w->pos[pt[0]], v1);
        v_sub(w->pos[pt[3]], w->pos[pt[2]], v2);
      } else {
        v_sub(w->pos[pt[0]], w->pos[pt[1]], v1);
        v_sub(w->pos[pt[2]], w->pos[pt[1]], v2);
      }
      v_cross(v1, v2, c);
      double t15 = EPS_SQ + c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      double s = sqrt(t15);
      double dot = v_dot(v1, v2);
      double inv = 1.0 / (t15 + dot * dot);
      double a = dot * inv / s; /* t17*t18*t25 */
      double b = s * inv;       /* t19 */
      double v2xc[3], cxv1[3], g1[3], g2[3];
      v_cross(v2, c, v2xc);
      v_cross(c, v1, cxv1);
      for (int k = 0; k < 3; ++k) {
        g1[k] = a * v2xc[k] - b * v2[k]; /* d theta / d v1 */
        g2[k] = a * cxv1[k] - b * v1[k]; /* d theta / d v2 */
      }
      if (type == OKX_ROW_ANGLE) {
        for (int k = 0; k < 3; ++k) {
          dp[k] = -g1[k];
          dp[3 + k] = g1[k];
          dp[6 + k] = -g2[k];
          dp[9 + k] = g2[k];
        }
        *npts = 4;
      } else {
        for (int k = 0; k < 3; ++k) {
          dp[k] = g1[k];
          dp[3 + k] = -g1[k] - g2[k];
          dp[6 + k] = g2[k];
        }
        *npts = 3;
      }
      return atan2(softnorm(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), dot) - q[0];
    }
    case OKX_ROW_VECTORS_PARALLEL: { /* constraints.py:351-371 / jacobians.py:192-262 */
      double v1[3], v2[3], c[3];
      v_sub(w->pos[pt[1]], w->pos[pt[0]], v1);
      v_sub(w->pos[pt[3]], w->pos[pt[2]], v2);
      v_cross(v1, v2, c);
      double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      double n1s = v_dot(v1, v1), n2s = v_dot(v2, v2);
      double sc = sqrt(EPS_SQ + c2), s1 = sqrt(EPS_SQ + n1s), s2 = sqrt(EPS_SQ + n2s);
      /* Jacobian differentiates sqrt(.+eps^2) forms (generate_jacobians.py:28-44). */
      double v2xc[3], cxv1[3];
      v_cross(v2, c, v2xc);
      v_cross(c, v1, cxv1);
      double k26 = 1.0 / (s1 * s2 * sc);
      double k19 = sc / (s2 * s1 * s1 * s1);
      double k31 = sc / (s1 * s2 * s2 * s2);
      for (int k = 0; k < 3; ++k) {
        double g1 = k26 * v2xc[k] - k19 * v1[k];
        double g2 = k26 * cxv1[k] - k31 * v2[k];
        dp[k] = -g1;
        dp[3 + k] = g1;
        dp[6 + k] = -g2;
        dp[9 + k] = g2;
      }
      *npts = 4;
      return softnorm(c2) / (softnorm(n1s) * softnorm(n2s));
    }
    case OKX_ROW_VECTORS_PERPENDICULAR: { /* constraints.py:414-429 / jacobians.py:266-318 */
      double v1[3], v2[3];
      v_sub(w->pos[pt[1]], w->pos[pt[0]], v1);
      v_sub(w->pos[pt[3]], w->pos[pt[2]], v2);
      double n1s = v_dot(v1, v1), n2s = v_dot(v2, v2), dot = v_dot(v1, v2);
      double s1 = sqrt(EPS_SQ + n1s), s2 = sqrt(EPS_SQ + n2s);
      double k16 = 1.0 / (s1 * s2);
      double k18 = dot / (s2 * s1 * s1 * s1);
      double k19 = dot / (s1 * s2 * s2 * s2);
      for (int k = 0; k < 3; ++k) {
        double g1 = k16 * v2[k] - k18 * v1[k];
        double g2 = k16 * v1[k] - k19 * v2[k];
        dp[k] = -g1;
        dp[3 + k] = g1;
        dp[6 + k] = -g2;
        dp[9 + k] = g2;
      }
      *npts = 4;
      return dot / (softnorm(n1s) * softnorm(n2s));
    }
    case OKX_ROW_EQUAL_DISTANCE: { /* constraints.py:466-477 / jacobians.py:322-367 */
      double d1[3], d2[3];
      v_sub(w->pos[pt[1]], w->pos[pt[0]], d1);
      v_sub(w->pos[pt[3]], w->pos[pt[2]], d2);
      double s1 = v_dot(d1, d1), s2 = v_dot(d2, d2);
      double i1 = 1.0 / sqrt(EPS_SQ + s1), i2 = 1.0 / sqrt(EPS_SQ + s2);
      for (int k = 0; k < 3; ++k) {
        dp[k] = -d1[k] * i1;
        dp[3 + k] = d1[k] * i1;
        dp[6 + k] = d2[k] * i2;
        dp[9 + k] = -d2[k] * i2;
      }
      *npts = 4;
      return softnorm(s1) - softnorm(s2);
    }
    case OKX_ROW_FIXED_AXIS: { /* constraints.py:508-516 / solver.py:407-416 */
      int ax = (int)q[0];
      dp[ax] = 1.0;
      *npts = 1;
      return w->pos[pt[0]][ax] - q[1];
    }
    case OKX_ROW_POINT_ON_LINE: { /* constraints.py:560-576 / jacobians.py:372-403 */
      double wv[3], c[3];
      v_sub(w->pos[pt[0]], q, wv);
      const double* ld = q + 3;
      v_cross(wv, ld, c);
      double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      double inv = 1.0 / sqrt(EPS_SQ + c2);
      double g[3];
      v_cross(ld, c, g); /* d|c|/dp * |c| = ld x c */
      for (int k = 0; k < 3; ++k) dp[k] = inv * g[k];
      *npts = 1;
      return softnorm(c2);
    }
    case OKX_ROW_LINE_PIN: { /* extension, okx.h: one component of (p - lp) x ld */
      double wv[3], c[3];
      v_sub(w->pos[pt[0]], q, wv);
      const double* ld = q + 3;
      v_cross(wv, ld, c);
      int comp = (int)q[6];
      /* c = wv x ld; dc_comp/dwv */
      if (comp == 0) {
        dp[1] = ld[2];
        dp[2] = -ld[1];
      } else if (comp == 1) {
        dp[0] = -ld[2];
        dp[2] = ld[0];
      } else {
        dp[0] = ld[1];
        dp[1] = -ld[0];
      }
      *npts = 1;
      return c[comp];
    }
    case OKX_ROW_POINT_ON_PLANE: { /* constraints.py:616-627 / solver.py:429-437 */
      double wv[3];
      v_sub(w->pos[pt[0]], q, wv);
      for (int k = 0; k < 3; ++k) dp[k] = q[3 + k];
      *npts = 1;
      return v_dot(wv, q + 3);
    }
    case OKX_ROW_MIDPOINT_ON_PLANE: { /* constraints.py:657-666 / solver.py:439-448 */
      const double* a = w->pos[pt[0]];
      const double* b = w->pos[pt[1]];
      double mid[3], wv[3];
      for (int k = 0; k < 3; ++k) mid[k] = a[k] + (b[k] - a[k]) / 2.0;
      v_sub(mid, q, wv);
      for (int k = 0; k < 3; ++k) {
        dp[k] = q[3 + k] * 0.5;
        dp[3 + k] = q[3 + k] * 0.5;
      }
      *npts = 2;
      return v_dot(wv, q + 3);
    }
    case OKX_ROW_COPLANAR:
    case OKX_ROW_SCALAR_TRIPLE: { /* constraints.py:698-709,731-733 / jacobians.py:426-483 */
      double v1[3], v2[3], v3[3], c23[3], c31[3], c12[3];
      v_sub(w->pos[pt[1]], w->pos[pt[0]], v1);
      v_sub(w->pos[pt[2]], w->pos[pt[0]], v2);
      v_sub(w->pos[pt[3]], w->pos[pt[0]], v3);
      v_cross(v2, v3, c23);
      v_cross(v3, v1, c31);
      v_cross(v1, v2, c12);
      double vol = v_dot(v1, c23);
      double sc = type == OKX_ROW_SCALAR_TRIPLE ? q[1] : 1.0;
      for (int k = 0; k < 3; ++k) {
        dp[3 + k] = c23[k] / sc;
        dp[6 + k] = c31[k] / sc;
        dp[9 + k] = c12[k] / sc;
        dp[k] = -(c23[k] + c31[k] + c12[k]) / sc;
      }
      *npts = 4;
      return type == OKX_ROW_SCALAR_TRIPLE ? (vol - q[0]) / q[1] : vol;
    }
    default:
      *npts = 0;
      return 0.0;
  }
}

/* ResidualComputer.compute (solver.py:226-275); jac may be NULL (compute_jacobian :502-581). */
static void eval_rj(eval_ws* w, const double* x, const double* targets, double* r, double* jac) {
  const okx_program_desc* p = w->p;
  const int n = w->n;
  set_free(w, x);
  update_derived(w, jac != NULL);
  if (jac) memset(jac, 0, sizeof(double) * (size_t)w->m * (size_t)n);
  for (int i = 0; i < p->n_rows; ++i) {
    double dp[12];
    int np = 0;
    r[i] = row_eval(w, i, dp, &np);
    if (!jac) continue;
    double* jr = jac + (size_t)i * n;
    const int32_t* pt = p->row_pts + 4 * i;
    for (int a = 0; a < np; ++a) {
      int point = pt[a];
      int v = w->var_of_point[point];
      if (v >= 0) {
        jr[v] += dp[3 * a];
        jr[v + 1] += dp[3 * a + 1];
        jr[v + 2] += dp[3 * a + 2];
        continue;
      }
      int d = w->dop_of_point[point];
      if (d < 0) continue;
      const double* src = w->dder + (size_t)d * 3 * n;
      for (int j = 0; j < n; ++j)
        jr[j] += dp[3 * a] * src[j] + dp[3 * a + 1] * src[n + j] + dp[3 * a + 2] * src[2 * n + j];
    }
  }
  for (int t = 0; t < p->n_targets; ++t) { /* solver.py:264-270, :560-579 */
    int point = p->tgt_point[t];
    const double* dir = p->tgt_dir + 3 * t;
    int i = p->n_rows + t;
    r[i] = v_dot(w->pos[point], dir) - targets[t];
    if (!jac) continue;
    double* jr = jac + (size_t)i * n;
    int v = w->var_of_point[point];
    if (v >= 0) {
      jr[v] = dir[0];
      jr[v + 1] = dir[1];
      jr[v + 2] = dir[2];
      continue;
    }
    int d = w->dop_of_point[point];
    if (d < 0) continue;
    const double* src = w->dder + (size_t)d * 3 * n;
    for (int j = 0; j < n; ++j) jr[j] = dir[0] * src[j] + dir[1] * src[n + j] + dir[2] * src[2 * n + j];
  }
}

/* ------------------------------------------------------------------ */
/* MINPACK restatement (third-party algorithm; see header)             */
/* ------------------------------------------------------------------ */

static double enorm(int n, const double* x, int stride) {
  const double rdwarf = 3.834e-20, rgiant = 1.304e19;
  double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
  double agiant = rgiant / (double)n;
  for (int i = 0; i < n; ++i) {
    double xabs = fabs(x[(size_t)i * stride]);
    if (xabs > rdwarf && xabs < agiant) {
      s2 += xabs * xabs;
    } else if (xabs <= rdwarf) {
      if (xabs > x3max) {
        double t = x3max / xabs;
        s3 = 1.0 + s3 * t * t;
        x3max = xabs;
      } else if (xabs != 0.0) {
        double t = xabs / x3max;
        s3 += t * t;
      }
    } else {
      if (xabs > x1max) {
        double t = x1max / xabs;
        s1 = 1.0 + s1 * t * t;
        x1max = xabs;
      } else {
        double t = xabs / x1max;
        s1 += t * t;
      }
    }
  }
  if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
  if (s2 != 0.0) {
    if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
    return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
  }
  return x3max * sqrt(s3);
}

/* a is m x n ROW-major with leading dimension n: A(i,j) = a[i*n + j]. */
#define A_(i, j) a[(size_t)(i) * n + (j)]

static void qrfac(int m, int n, double* a, int* ipvt, double* rdiag, double* acnorm, double* wa) {
  const double epsmch = DBL_EPSILON;
  for (int j = 0; j < n; ++j) {
    acnorm[j] = enorm(m, &A_(0, j), n);
    rdiag[j] = acnorm[j];
    wa[j] = rdiag[j];
    ipvt[j] = j;
  }
  int minmn = m < n ? m : n;
  for (int j = 0; j < minmn; ++j) {
    int kmax = j;
    for (int k = j; k < n; ++k)
      if (rdiag[k] > rdiag[kmax]) kmax = k;
    if (kmax != j) {
      for (int i = 0; i < m; ++i) {
        double t = A_(i, j);
        A_(i, j) = A_(i, kmax);
        A_(i, kmax) = t;
      }
      rdiag[kmax] = rdiag[j];
      wa[kmax] = wa[j];
      int k = ipvt[j];
      ipvt[j] = ipvt[kmax];
      ipvt[kmax] = k;
    }
    double ajnorm = enorm(m - j, &A_(j, j), n);
    if (ajnorm != 0.0) {
      if (A_(j, j) < 0.0) ajnorm = -ajnorm;
      for (int i = j; i < m; ++i) A_(i, j) /= ajnorm;
      A_(j, j) += 1.0;
      for (int k = j + 1; k < n; ++k) {
        double sum = 0.0;
        for (int i = j; i < m; ++i) sum += A_(i, j) * A_(i, k);
        double temp = sum / A_(j, j);
        for (int i = j; i < m; ++i) A_(i, k) -= temp * A_(i, j);
        if (rdiag[k] != 0.0) {
          temp = A_(j, k) / rdiag[k];
          double d = 1.0 - temp * temp;
          rdiag[k] *= sqrt(d > 0.0 ? d : 0.0);
          double ratio = rdiag[k] / wa[k];
          if (0.05 * ratio * ratio <= epsmch) {
            rdiag[k] = enorm(m - j - 1, &A_(j + 1, k), n);
            wa[k] = rdiag[k];
          }
        }
      }
    }
    rdiag[j] = -ajnorm;
  }
}

/* r: n x n upper triangle lives in a (row-major, ld n).  Strict lower triangle is scratch. */
static void qrsolv(int n, double* a, const int* ipvt, const double* diag, const double* qtb,
                   double* x, double* sdiag, double* wa) {
  for (int j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) A_(i, j) = A_(j, i);
    x[j] = A_(j, j);
    wa[j] = qtb[j];
  }
  for (int j = 0; j < n; ++j) {
    int l = ipvt[j];
    if (diag[l] != 0.0) {
      for (int k = j; k < n; ++k) sdiag[k] = 0.0;
      sdiag[j] = diag[l];
      double qtbpj = 0.0;
      for (int k = j; k < n; ++k) {
        if (sdiag[k] == 0.0) continue;
        double cs, sn;
        if (fabs(A_(k, k)) < fabs(sdiag[k])) {
          double cotan = A_(k, k) / sdiag[k];
          sn = 0.5 / sqrt(0.25 + 0.25 * cotan * cotan);
          cs = sn * cotan;
        } else {
          double tn = sdiag[k] / A_(k, k);
          cs = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
          sn = cs * tn;
        }
        A_(k, k) = cs * A_(k, k) + sn * sdiag[k];
        double temp = cs * wa[k] + sn * qtbpj;
        qtbpj = -sn * wa[k] + cs * qtbpj;
        wa[k] = temp;
        for (int i = k + 1; i < n; ++i) {
          temp = cs * A_(i, k) + sn * sdiag[i];
          sdiag[i] = -sn * A_(i, k) + cs * sdiag[i];
          A_(i, k) = temp;
        }
      }
    }
    sdiag[j] = A_(j, j);
    A_(j, j) = x[j];
  }
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    if (sdiag[j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa[j] = 0.0;
  }
  for (int k = 1; k <= nsing; ++k) {
    int j = nsing - k;
    double sum = 0.0;
    for (int i = j + 1; i < nsing; ++i) sum += A_(i, j) * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

static void lmpar(int n, double* a, const int* ipvt, const double* diag, const double* qtb,
                  double delta, double* par, double* x, double* sdiag, double* wa1, double* wa2) {
  const double dwarf = DBL_MIN, p1 = 0.1, p001 = 0.001;
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    wa1[j] = qtb[j];
    if (A_(j, j) == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa1[j] = 0.0;
  }
  for (int k = 1; k <= nsing; ++k) {
    int j = nsing - k;
    wa1[j] /= A_(j, j);
    double temp = wa1[j];
    for (int i = 0; i < j; ++i) wa1[i] -= A_(i, j) * temp;
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];

  int iter = 0;
  for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = enorm(n, wa2, 1);
  double fp = dxnorm - delta;
  if (fp <= p1 * delta) {
    *par = 0.0;
    return;
  }
  double parl = 0.0;
  if (nsing >= n) {
    for (int j = 0; j < n; ++j) {
      int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      double sum = 0.0;
      for (int i = 0; i < j; ++i) sum += A_(i, j) * wa1[i];
      wa1[j] = (wa1[j] - sum) / A_(j, j);
    }
    double temp = enorm(n, wa1, 1);
    parl = ((fp / delta) / temp) / temp;
  }
  for (int j = 0; j < n; ++j) {
    double sum = 0.0;
    for (int i = 0; i <= j; ++i) sum += A_(i, j) * qtb[i];
    wa1[j] = sum / diag[ipvt[j]];
  }
  double gnorm = enorm(n, wa1, 1);
  double paru = gnorm / delta;
  if (paru == 0.0) paru = dwarf / (delta < p1 ? delta : p1);
  if (*par < parl) *par = parl;
  if (*par > paru) *par = paru;
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (*par == 0.0) *par = dwarf > p001 * paru ? dwarf : p001 * paru;
    double temp = sqrt(*par);
    for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
    qrsolv(n, a, ipvt, wa1, qtb, x, sdiag, wa2);
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = enorm(n, wa2, 1);
    temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= p1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
    for (int j = 0; j < n; ++j) {
      int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      wa1[j] /= sdiag[j];
      temp = wa1[j];
      for (int i = j + 1; i < n; ++i) wa1[i] -= A_(i, j) * temp;
    }
    temp = enorm(n, wa1, 1);
    double parc = ((fp / delta) / temp) / temp;
    if (fp > 0.0 && parl < *par) parl = *par;
    if (fp < 0.0 && paru > *par) paru = *par;
    double cand = *par + parc;
    *par = parl > cand ? parl : cand;
  }
  if (iter == 0) *par = 0.0;
}

typedef struct {
  eval_ws* w;
  const double* targets;
} lm_problem;

typedef struct {
  int info;   /* MINPACK info 0..8 */
  int nfev;
  int njev;
} lm_result;

/*
 * lmder with SciPy's conventions: mode=2 with diag=1 (x_scale=1.0), factor=100,
 * maxfev = 100*n unless overridden (solver.py:158-169 passes no max_nfev).
 * x: in/out [n]; fvec: out [m]; fjac scratch [m*n].
 */
static lm_result lmder(lm_problem* pb, int m, int n, double* x, double* fvec, double* a,
                       double ftol, double xtol, double gtol, int maxfev, double factor) {
  const double epsmch = DBL_EPSILON;
  lm_result res = {0, 0, 0};
  if (n <= 0 || m < n || ftol < 0 || xtol < 0 || gtol < 0 || maxfev <= 0 || factor <= 0) return res;
  int* ipvt = (int*)malloc(sizeof(int) * (size_t)n);
  double* buf = (double*)malloc(sizeof(double) * (size_t)(6 * n + m));
  double *diag = buf, *qtf = buf + n, *wa1 = buf + 2 * n, *wa2 = buf + 3 * n, *wa3 = buf + 4 * n,
         *sdiag_unused = buf + 5 * n, *wa4 = buf + 6 * n;
  (void)sdiag_unused;
  for (int j = 0; j < n; ++j) diag[j] = 1.0;

  eval_rj(pb->w, x, pb->targets, fvec, NULL);
  res.nfev = 1;
  double fnorm = enorm(m, fvec, 1);
  double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0;
  int iter = 1;
  int info = 0;
  for (;;) {
    eval_rj(pb->w, x, pb->targets, wa4, a); /* Jacobian at x (residual recomputed, discarded) */
    res.njev++;
    qrfac(m, n, a, ipvt, wa1, wa2, wa3);
    if (iter == 1) {
      for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
      xnorm = enorm(n, wa3, 1);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
    for (int j = 0; j < n; ++j) {
      if (A_(j, j) != 0.0) {
        double sum = 0.0;
        for (int i = j; i < m; ++i) sum += A_(i, j) * wa4[i];
        double temp = -sum / A_(j, j);
        for (int i = j; i < m; ++i) wa4[i] += A_(i, j) * temp;
      }
      A_(j, j) = wa1[j];
      qtf[j] = wa4[j];
    }
    gnorm = 0.0;
    if (fnorm != 0.0) {
      for (int j = 0; j < n; ++j) {
        int l = ipvt[j];
        if (wa2[l] == 0.0) continue;
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += A_(i, j) * (qtf[i] / fnorm);
        double g = fabs(sum / wa2[l]);
        if (g > gnorm) gnorm = g;
      }
    }
    if (gnorm <= gtol) info = 4;
    if (info != 0) break;
    double ratio = 0.0;
    do {
      lmpar(n, a, ipvt, diag, qtf, delta, &par, wa1, wa2, wa3, wa4);
      for (int j = 0; j < n; ++j) {
        wa1[j] = -wa1[j];
        wa2[j] = x[j] + wa1[j];
        wa3[j] = diag[j] * wa1[j];
      }
      double pnorm = enorm(n, wa3, 1);
      if (iter == 1 && pnorm < delta) delta = pnorm;
      eval_rj(pb->w, wa2, pb->targets, wa4, NULL);
      res.nfev++;
      double fnorm1 = enorm(m, wa4, 1);
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) {
        double t = fnorm1 / fnorm;
        actred = 1.0 - t * t;
      }
      for (int j = 0; j < n; ++j) {
        wa3[j] = 0.0;
        int l = ipvt[j];
        double temp = wa1[l];
        for (int i = 0; i <= j; ++i) wa3[i] += A_(i, j) * temp;
      }
      double temp1 = enorm(n, wa3, 1) / fnorm;
      double temp2 = (sqrt(par) * pnorm) / fnorm;
      double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
      double dirder = -(temp1 * temp1 + temp2 * temp2);
      ratio = 0.0;
      if (prered != 0.0) ratio = actred / prered;
      if (ratio <= 0.25) {
        double temp;
        if (actred >= 0.0)
          temp = 0.5;
        else
          temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        double dm = delta < pnorm / 0.1 ? delta : pnorm / 0.1;
        delta = temp * dm;
        par /= temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        for (int j = 0; j < n; ++j) {
          x[j] = wa2[j];
          wa2[j] = diag[j] * x[j];
        }
        for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
        xnorm = enorm(n, wa2, 1);
        fnorm = fnorm1;
        ++iter;
      }
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
      if (delta <= xtol * xnorm) info = 2;
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
      if (info != 0) break;
      if (res.nfev >= maxfev) info = 5;
      if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
      if (delta <= epsmch * xnorm) info = 7;
      if (gnorm <= epsmch) info = 8;
      if (info != 0) break;
    } while (ratio < 1e-4);
    if (info != 0) break;
  }
  res.info = info;
  free(ipvt);
  free(buf);
  return res;
}

/* ------------------------------------------------------------------ */
/* Public oracle entry points (ctypes)                                 */
/* ------------------------------------------------------------------ */

typedef struct okx_oracle_opts {
  double ftol, xtol, gtol;   /* SolverConfig (solver.py:65-80): 1e-5, 1e-9, 1e-9 */
  double residual_tolerance; /* 1e-3 */
  int32_t max_nfev;          /* 0 -> 100*n (SciPy default for method="lm") */
  int32_t warm_start;        /* 1: step k starts from step k-1 (solver.py:774); 0: design */
} okx_oracle_opts;

typedef struct okx_oracle_info {
  double max_residual;
  int32_t nfev;
  int32_t njev;
  int32_t minpack_info;
  int32_t success; /* SciPy: status > 0  <=> minpack info in {1,2,3,4} (6,7,8 accepted too) */
} okx_oracle_info;

int32_t okx_oracle_eval(const okx_program_desc* p, int64_t n_eval, const double* x,
                        const double* targets, double* r, double* jac) {
  eval_ws w;
  int rc = ws_init(&w, p);
  if (rc != OKX_OK) return rc;
  for (int64_t b = 0; b < n_eval; ++b)
    eval_rj(&w, x + b * w.n, targets + b * p->n_targets, r + b * w.m,
            jac ? jac + b * (int64_t)w.m * w.n : NULL);
  ws_free(&w);
  return OKX_OK;
}

/* All point positions (fixed, free, derived) for a free-coordinate vector. */
int32_t okx_oracle_positions(const okx_program_desc* p, const double* x, double* pos_out) {
  eval_ws w;
  int rc = ws_init(&w, p);
  if (rc != OKX_OK) return rc;
  set_free(&w, x);
  update_derived(&w, 0);
  memcpy(pos_out, w.pos, sizeof(double) * 3 * (size_t)p->n_points);
  ws_free(&w);
  return OKX_OK;
}

/*
 * solve_suspension_sweep (solver.py:654-776): S steps, targets [S][T] absolute.
 * out_pos [S][n_out][3], out_x [S][n] (may be NULL), info [S].
 * Returns the index of the first failing step + 1 as a positive value (the reference
 * raises there, solver.py:726-747) or 0 when every step was accepted; the remaining
 * steps are still solved so that a batch caller can inspect them.
 */
int32_t okx_oracle_sweep(const okx_program_desc* p, const okx_oracle_opts* o, int64_t n_steps,
                         const double* targets, const double* x_start, double* out_pos,
                         double* out_x, okx_oracle_info* info) {
  eval_ws w;
  int rc = ws_init(&w, p);
  if (rc != OKX_OK) return rc;
  const int n = w.n, m = w.m;
  if (n > m) {
    ws_free(&w);
    return OKX_ERR_UNDERDETERMINED;
  }
  double* x = (double*)malloc(sizeof(double) * (size_t)n);
  double* x0 = (double*)malloc(sizeof(double) * (size_t)n);
  double* fvec = (double*)malloc(sizeof(double) * (size_t)m);
  double* a = (double*)malloc(sizeof(double) * (size_t)m * (size_t)n);
  for (int k = 0; k < p->n_free; ++k)
    for (int c = 0; c < 3; ++c)
      x0[3 * k + c] = x_start ? x_start[3 * k + c] : p->design_pos[3 * p->free_point[k] + c];
  memcpy(x, x0, sizeof(double) * (size_t)n);
  int first_fail = 0;
  int maxfev = o->max_nfev > 0 ? o->max_nfev : 100 * n;
  for (int64_t s = 0; s < n_steps; ++s) {
    if (!o->warm_start) memcpy(x, x0, sizeof(double) * (size_t)n);
    lm_problem pb = {&w, targets + s * p->n_targets};
    lm_result res = lmder(&pb, m, n, x, fvec, a, o->ftol, o->xtol, o->gtol, maxfev, 100.0);
    double mx = 0.0;
    for (int i = 0; i < m; ++i)
      if (fabs(fvec[i]) > mx) mx = fabs(fvec[i]);
    int success = (res.info >= 1 && res.info <= 4) || (res.info >= 6 && res.info <= 8);
    info[s].max_residual = mx;
    info[s].nfev = res.nfev;
    info[s].njev = res.njev;
    info[s].minpack_info = res.info;
    info[s].success = success && mx <= o->residual_tolerance;
    if (!info[s].success && first_fail == 0) first_fail = (int)(s + 1);
    set_free(&w, x); /* solver.py:759-760 */
    update_derived(&w, 0);
    for (int k = 0; k < p->n_out; ++k)
      memcpy(out_pos + (s * p->n_out + k) * 3, w.pos[p->out_point[k]], sizeof(double) * 3);
    if (out_x) memcpy(out_x + s * n, x, sizeof(double) * (size_t)n);
  }
  free(x);
  free(x0);
  free(fvec);
  free(a);
  ws_free(&w);
  return first_fail;
}

/*
 * Per-geometry problem emission (reference topology constraints(); see okx_rebind_design
 * in okx.h): recompute design-state row targets from hardpoints [P][3]; writes
 * pos_out [P][3] (derived filled in) and row_param_out [Mc][8].
 */
int32_t okx_oracle_rebind(const okx_program_desc* p, const double* hardpoints, double* pos_out,
                          double* row_param_out) {
  eval_ws w;
  int rc = ws_init(&w, p);
  if (rc != OKX_OK) return rc;
  memcpy(w.pos, hardpoints, sizeof(double) * 3 * (size_t)p->n_points);
  update_derived(&w, 0);
  memcpy(pos_out, w.pos, sizeof(double) * 3 * (size_t)p->n_points);
  memcpy(row_param_out, p->row_param, sizeof(double) * OKX_ROW_PARAMS * (size_t)p->n_rows);
  for (int i = 0; i < p->n_rows; ++i) {
    const int32_t* pt = p->row_pts + 4 * i;
    double* q = row_param_out + OKX_ROW_PARAMS * i;
    switch (p->row_type[i]) {
      case OKX_ROW_DISTANCE: { /* geometric.py:17-28 compute_point_point_distance */
        double d[3];
        v_sub(w.pos[pt[1]], w.pos[pt[0]], d);
        q[0] = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      } break;
      case OKX_ROW_ANGLE:
      case OKX_ROW_THREE_POINT_ANGLE: { /* geometric.py:71-104 compute_vector_vector_angle */
        double v1[3], v2[3], u1[3], u2[3], c[3];
        if (p->row_type[i] == OKX_ROW_ANGLE) {
          v_sub(w.pos[pt[1]], w.pos[pt[0]], v1);
          v_sub(w.pos[pt[3]], w.pos[pt[2]], v2);
        } else {
          v_sub(w.pos[pt[0]], w.pos[pt[1]], v1);
          v_sub(w.pos[pt[2]], w.pos[pt[1]], v2);
        }
        v_normalize(v1, u1);
        v_normalize(v2, u2);
        v_cross(u1, u2, c);
        q[0] = atan2(sqrt(v_dot(c, c)), v_dot(u1, u2));
      } break;
      case OKX_ROW_SCALAR_TRIPLE: { /* attachments.py:45-74 */
        double v1[3], v2[3], v3[3], c[3];
        v_sub(w.pos[pt[1]], w.pos[pt[0]], v1);
        v_sub(w.pos[pt[2]], w.pos[pt[0]], v2);
        v_sub(w.pos[pt[3]], w.pos[pt[0]], v3);
        v_cross(v2, v3, c);
        q[0] = v_dot(v1, c);
        q[1] = fabs(q[0]);
      } break;
      case OKX_ROW_POINT_ON_LINE:
      case OKX_ROW_LINE_PIN: /* track_rod.py:92-96: line through the design rack pickup */
        q[0] = w.pos[pt[0]][0];
        q[1] = w.pos[pt[0]][1];
        q[2] = w.pos[pt[0]][2];
        break;
      default:
        break;
    }
  }
  ws_free(&w);
  return OKX_OK;
}
