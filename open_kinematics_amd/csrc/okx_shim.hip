// okx_shim.hip — camber-shim setup solve for a batch of geometries.
//
// Reference: core/suspensions/config/shims.py — residuals :118-268 (datum A / B closure, face-normal
// alignment, heading-link length, optional upright-mounted pushrod length; 7 or 8 variables: wishbone
// angle, camber-block and upright-body rotation vectors, rocker angle), context :339-440, solve
// :442-470 (scipy MINPACK `lm`, finite-difference Jacobian) — and the pose write-back of
// corner/double_wishbone.py:501-570 / corner/mechanisms.py:247-265.
//
// One thread per geometry (a setup solve happens once per geometry, before its sweep): Levenberg-
// Marquardt on the normal equations with an exact forward-mode Jacobian (one dual evaluation per
// variable), 8 x 8 Cholesky in registers, iterated to machine precision.  The rotation
// R(rho) v = v + A rho x v + B rho x (rho x v), A = sin(t)/t, B = (1 - cos t)/t^2, is the reference's
// Rodrigues formula written so that it is smooth at rho = 0 (the seed).  The kernel is compute-bound
// and tiny next to a sweep (≈ 40 B of coefficients per flop-heavy iteration); it reads the authored
// hardpoint table and rewrites the moved points in place, ready for okx_rebind_design.
#include <hip/hip_runtime.h>
#pragma once
#include <stdint.h>

#include "../../include/okx.h"

// every function is host + device so that tests/ can drive the same code on the CPU (sanitizers, no GPU);
// the library itself only ever launches the kernel
#define OKX_SHIM_FN __host__ __device__ __forceinline__

namespace okx {
namespace shim {

struct D {  // value + derivative along one variable
  double v, d;
};
OKX_SHIM_FN D operator+(D a, D b) { return {a.v + b.v, a.d + b.d}; }
OKX_SHIM_FN D operator-(D a, D b) { return {a.v - b.v, a.d - b.d}; }
OKX_SHIM_FN D operator*(D a, D b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
OKX_SHIM_FN D operator*(double s, D a) { return {s * a.v, s * a.d}; }
OKX_SHIM_FN D operator+(double s, D a) { return {s + a.v, a.d}; }
OKX_SHIM_FN D operator-(D a, double s) { return {a.v - s, a.d}; }
OKX_SHIM_FN D root(D a) {
  const double r = sqrt(a.v);
  return {r, a.d / (2.0 * r)};
}
OKX_SHIM_FN double root(double a) { return sqrt(a); }

struct V3 {
  double x, y, z;
};
OKX_SHIM_FN V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
OKX_SHIM_FN V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
OKX_SHIM_FN V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
OKX_SHIM_FN double norm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
OKX_SHIM_FN V3 load3(const double* p, int k) { return {p[3 * k], p[3 * k + 1], p[3 * k + 2]}; }
OKX_SHIM_FN void store3(double* p, int k, V3 v) {
  p[3 * k] = v.x;
  p[3 * k + 1] = v.y;
  p[3 * k + 2] = v.z;
}

// A = sin(t)/t, B = (1 - cos t)/t^2 and their derivatives with respect to s = t^2
OKX_SHIM_FN void rotation_coefficients(double s, double* A, double* B, double* dA, double* dB) {
  if (s < 1e-6) {
    *A = 1.0 + s * (-1.0 / 6.0 + s * (1.0 / 120.0 - s * (1.0 / 5040.0)));
    *B = 0.5 + s * (-1.0 / 24.0 + s * (1.0 / 720.0 - s * (1.0 / 40320.0)));
    *dA = -1.0 / 6.0 + s * (1.0 / 60.0 - s * (1.0 / 1680.0));
    *dB = -1.0 / 24.0 + s * (1.0 / 360.0 - s * (1.0 / 13440.0));
    return;
  }
  const double t = sqrt(s), sn = sin(t), half = sin(0.5 * t);
  *A = sn / t;
  *B = 2.0 * half * half / s;
  *dA = (cos(t) - *A) / (2.0 * s);
  *dB = (0.5 * *A - *B) / s;
}

template <class S>
struct Vec {
  S x, y, z;
};
OKX_SHIM_FN D lift(double v) { return {v, 0.0}; }

// R(rho) v for a constant vector v (geometric.py:351-376)
OKX_SHIM_FN Vec<D> rotate(V3 v, D rx, D ry, D rz) {
  const D s = rx * rx + ry * ry + rz * rz;
  double A, B, dA, dB;
  rotation_coefficients(s.v, &A, &B, &dA, &dB);
  const D a = {A, dA * s.d}, b = {B, dB * s.d};
  const D cx = ry * lift(v.z) - rz * lift(v.y), cy = rz * lift(v.x) - rx * lift(v.z), cz = rx * lift(v.y) - ry * lift(v.x);
  const D ex = ry * cz - rz * cy, ey = rz * cx - rx * cz, ez = rx * cy - ry * cx;
  return {v.x + a * cx + b * ex, v.y + a * cy + b * ey, v.z + a * cz + b * ez};
}
OKX_SHIM_FN V3 rotate(V3 v, V3 r) {
  const Vec<D> out = rotate(v, lift(r.x), lift(r.y), lift(r.z));
  return {out.x.v, out.y.v, out.z.v};
}

struct Context {  // shims.py:97-116, :58-66
  double t;
  V3 n0, axis, hl_in, lbj, front, front_to_ubj, ua, ub, la, lb, l_hl;
  double hl_len;
  int rocker;
  V3 rk_point, rk_dir, rk_to_pi, lbj_to_po;
  double pr_len;
};

constexpr int kMaxVars = 8, kMaxRes = 11;

// shims.py:118-268 with variable `seed` carrying derivative 1 (seed < 0: plain values)
OKX_SHIM_FN void residuals(const Context& c, const double* x, int seed, D* r) {
  D q[kMaxVars];
#pragma unroll
  for (int k = 0; k < kMaxVars; ++k) q[k] = {x[k], k == seed ? 1.0 : 0.0};
  const Vec<D> arm = rotate(c.front_to_ubj, c.axis.x * q[0], c.axis.y * q[0], c.axis.z * q[0]);
  const D ux = c.front.x + arm.x, uy = c.front.y + arm.y, uz = c.front.z + arm.z;
  const Vec<D> nc = rotate(c.n0, q[1], q[2], q[3]), nu = rotate(c.n0, q[4], q[5], q[6]);
  const Vec<D> ca = rotate(c.ua, q[1], q[2], q[3]), cb = rotate(c.ub, q[1], q[2], q[3]);
  const Vec<D> ba = rotate(c.la, q[4], q[5], q[6]), bb = rotate(c.lb, q[4], q[5], q[6]);
  r[0] = (c.lbj.x + ba.x) - (ux + ca.x) - c.t * nc.x;
  r[1] = (c.lbj.y + ba.y) - (uy + ca.y) - c.t * nc.y;
  r[2] = (c.lbj.z + ba.z) - (uz + ca.z) - c.t * nc.z;
  r[3] = (c.lbj.x + bb.x) - (ux + cb.x) - c.t * nc.x;
  r[4] = (c.lbj.y + bb.y) - (uy + cb.y) - c.t * nc.y;
  r[5] = (c.lbj.z + bb.z) - (uz + cb.z) - c.t * nc.z;
  r[6] = nu.x - nc.x;
  r[7] = nu.y - nc.y;
  r[8] = nu.z - nc.z;
  const Vec<D> hl = rotate(c.l_hl, q[4], q[5], q[6]);
  const D hx = (c.lbj.x - c.hl_in.x) + hl.x, hy = (c.lbj.y - c.hl_in.y) + hl.y, hz = (c.lbj.z - c.hl_in.z) + hl.z;
  r[9] = root(hx * hx + hy * hy + hz * hz) - c.hl_len;
  if (c.rocker) {
    const Vec<D> pi = rotate(c.rk_to_pi, c.rk_dir.x * q[7], c.rk_dir.y * q[7], c.rk_dir.z * q[7]);
    const Vec<D> po = rotate(c.lbj_to_po, q[4], q[5], q[6]);
    const D px = (c.lbj.x - c.rk_point.x) + po.x - pi.x, py = (c.lbj.y - c.rk_point.y) + po.y - pi.y,
            pz = (c.lbj.z - c.rk_point.z) + po.z - pi.z;
    r[10] = root(px * px + py * py + pz * pz) - c.pr_len;
  } else {
    r[10] = {0.0, 0.0};
  }
}

struct ShimArgs {
  okx_shim_roles roles;
  double* points;        // [G][P][3], in: authored, out: setup
  const double* shim;    // [G][OKX_SHIM_PARAMS]
  okx_shim_info* info;   // [G] or null
  long long n_geometries;
  int n_points;
};

// geometric.py:377-400
OKX_SHIM_FN V3 rotate_about(V3 point, V3 pivot, V3 rotvec) { return pivot + rotate(point - pivot, rotvec); }

// the setup solve of geometry g
OKX_SHIM_FN void solve_one(const ShimArgs& a, long long g) {
  const okx_shim_roles& R = a.roles;
  double* pts = a.points + g * 3 * a.n_points;
  const double* sp = a.shim + g * OKX_SHIM_PARAMS;
  const V3 face_a = {sp[0], sp[1], sp[2]}, face_b = {sp[3], sp[4], sp[5]}, n0 = {sp[6], sp[7], sp[8]};
  const double design_t = sp[9], setup_t = sp[10];
  okx_shim_info info = {};
  info.converged = 1;
  if (fabs(setup_t - design_t) < 1e-6) {  // shims.py:346-357: nothing moves
    if (a.info) a.info[g] = info;
    return;
  }
  Context c;
  const V3 ubj = load3(pts, R.upper_outboard), front = load3(pts, R.upper_inboard_front),
           rear = load3(pts, R.upper_inboard_rear), hl_out = load3(pts, R.heading_outboard);
  c.t = setup_t;
  c.n0 = n0;
  c.lbj = load3(pts, R.lower_outboard);
  c.front = front;
  c.axis = (1.0 / norm(rear - front)) * (rear - front);
  c.front_to_ubj = ubj - front;
  c.hl_in = load3(pts, R.heading_inboard);
  c.hl_len = norm(hl_out - c.hl_in);
  const double half = 0.5 * design_t;
  c.ua = (face_a - half * n0) - ubj;
  c.ub = (face_b - half * n0) - ubj;
  c.la = (face_a + half * n0) - c.lbj;
  c.lb = (face_b + half * n0) - c.lbj;
  c.l_hl = hl_out - c.lbj;
  c.rocker = R.rocker;
  if (R.rocker) {
    const V3 ax_a = load3(pts, R.rocker_axis_a), ax_b = load3(pts, R.rocker_axis_b);
    const V3 pi = load3(pts, R.pushrod_inboard), po = load3(pts, R.pushrod_outboard);
    c.rk_point = ax_a;
    c.rk_dir = (1.0 / norm(ax_b - ax_a)) * (ax_b - ax_a);
    c.rk_to_pi = pi - ax_a;
    c.lbj_to_po = po - c.lbj;
    c.pr_len = norm(po - pi);
  } else {
    c.rk_point = c.rk_dir = c.rk_to_pi = c.lbj_to_po = {0.0, 0.0, 0.0};
    c.pr_len = 0.0;
  }
  const int n = R.rocker ? 8 : 7, m = R.rocker ? 11 : 10;

  double x[kMaxVars] = {0, 0, 0, 0, 0, 0, 0, 0};
  D rd[kMaxRes];
  double r[kMaxRes], J[kMaxRes][kMaxVars];
  double lambda = 0.0, nu = 2.0, cost;
  residuals(c, x, -1, rd);
  cost = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxRes; ++i) {
    r[i] = rd[i].v;
    cost += r[i] * r[i];
  }
  int it = 0, done = 0;
  for (; it < 100 && !done; ++it) {
#pragma unroll
    for (int j = 0; j < kMaxVars; ++j) {
      if (j < n) {
        residuals(c, x, j, rd);
#pragma unroll
        for (int i = 0; i < kMaxRes; ++i) J[i][j] = rd[i].d;
      } else {
#pragma unroll
        for (int i = 0; i < kMaxRes; ++i) J[i][j] = 0.0;
      }
    }
    double A[kMaxVars][kMaxVars], gr[kMaxVars];
    double amax = 0.0;
#pragma unroll
    for (int p = 0; p < kMaxVars; ++p) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < kMaxRes; ++i) s += J[i][p] * r[i];
      gr[p] = s;
#pragma unroll
      for (int q = 0; q <= p; ++q) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kMaxRes; ++i) t += J[i][p] * J[i][q];
        A[p][q] = t;
      }
      amax = fmax(amax, A[p][p]);
    }
    if (it == 0) lambda = 1e-6 * amax;
    // damped step, retried with more damping until the cost goes down
    int accepted = 0;
    for (int attempt = 0; attempt < 30 && !accepted; ++attempt) {
      double L[kMaxVars][kMaxVars], dx[kMaxVars];
      bool ok = true;
#pragma unroll
      for (int p = 0; p < kMaxVars; ++p) {
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          double s = A[p][q] + (p == q ? (p < n ? lambda : 1.0) : 0.0);
#pragma unroll
          for (int k = 0; k < q; ++k) s -= L[p][k] * L[q][k];
          if (p == q) {
            if (!(s > 0.0)) ok = false;
            L[p][p] = sqrt(s > 0.0 ? s : 1.0);
          } else {
            L[p][q] = s / L[q][q];
          }
        }
      }
      if (ok) {
#pragma unroll
        for (int p = 0; p < kMaxVars; ++p) {
          double s = -gr[p];
#pragma unroll
          for (int k = 0; k < p; ++k) s -= L[p][k] * dx[k];
          dx[p] = s / L[p][p];
        }
#pragma unroll
        for (int p = kMaxVars - 1; p >= 0; --p) {
          double s = dx[p];
#pragma unroll
          for (int k = p + 1; k < kMaxVars; ++k) s -= L[k][p] * dx[k];
          dx[p] = s / L[p][p];
        }
        double xn[kMaxVars], step = 0.0, size = 0.0, predicted = 0.0;
#pragma unroll
        for (int p = 0; p < kMaxVars; ++p) {
          xn[p] = x[p] + dx[p];
          step = fmax(step, fabs(dx[p]));
          size = fmax(size, fabs(x[p]));
          predicted += dx[p] * (lambda * dx[p] - gr[p]);
        }
        residuals(c, xn, -1, rd);
        double trial = 0.0;
#pragma unroll
        for (int i = 0; i < kMaxRes; ++i) trial += rd[i].v * rd[i].v;
        if (trial <= cost) {
          const double rho = predicted > 0.0 ? (cost - trial) / predicted : 1.0;
          const double f = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
          lambda *= fmax(1.0 / 3.0, f);
          nu = 2.0;
#pragma unroll
          for (int p = 0; p < kMaxVars; ++p) x[p] = xn[p];
#pragma unroll
          for (int i = 0; i < kMaxRes; ++i) r[i] = rd[i].v;
          // quadratic convergence: a step below 1e-9 rad leaves an error of its square; or the arithmetic's floor
          if (step <= 1e-9 * (1.0 + size) || trial == cost) done = 1;
          cost = trial;
          accepted = 1;
          continue;
        }
      }
      lambda = fmax(lambda, 1e-12 * amax) * nu;
      nu *= 2.0;
    }
    if (!accepted) done = 1;  // no descent left: at the floor of the arithmetic
    if (cost == 0.0) done = 1;
  }
  (void)m;
  double rmax = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxRes; ++i) rmax = fmax(rmax, fabs(r[i]));

  // pose write-back (double_wishbone.py:543-570)
  const V3 upright = {x[4], x[5], x[6]};
  const double angle = norm(upright);
  store3(pts, R.upper_outboard, front + rotate(c.front_to_ubj, x[0] * c.axis));
  if (angle > 1e-6)
    for (int k = 0; k < R.n_upright_points; ++k)
      store3(pts, R.upright_point[k], rotate_about(load3(pts, R.upright_point[k]), c.lbj, upright));
  if (R.rocker)  // mechanisms.py:247-265
    for (int k = 0; k < R.n_rocker_points; ++k)
      store3(pts, R.rocker_point[k], rotate_about(load3(pts, R.rocker_point[k]), c.rk_point, x[7] * c.rk_dir));
  if (a.info) {
    info.residual_norm = sqrt(cost);
    info.max_residual = rmax;
    info.upright_angle_rad = angle;
    info.rocker_angle_rad = R.rocker ? x[7] : 0.0;
    info.wishbone_angle_rad = x[0];
    info.converged = rmax <= 1e-3 ? 1 : 0;  // SOLVE_ACCEPT_RESIDUAL (primitives/constants.py:19)
    info.iterations = it;
    a.info[g] = info;
  }
}


__global__ void __launch_bounds__(64) okx_camber_shim_kernel(ShimArgs a) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < a.n_geometries) solve_one(a, g);
}

}  // namespace shim
}  // namespace okx
