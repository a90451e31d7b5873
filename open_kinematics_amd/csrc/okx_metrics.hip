// okx_metrics.hip — corner state metrics of solved states and their derivatives along the
// solution-manifold tangents.
//
// Reference: core/metrics/angles.py:22-132 (camber, caster, KPI, toe / roadwheel angle),
// travel.py:19-45 (wheel travel, half-track), steering_geometry.py:22-76 (scrub radius,
// mechanical trail) over MetricContext (context.py:82-138); the derivative columns
// (metrics/derivatives.py: d response / d driver along a tangent field) are forward-mode
// derivatives of the same formulas, which is what `Dual` below carries.
//
// One thread per solved state: six role points in (144 B), eight scalars out (64 B), plus
// 64 B per target when tangents are given — a pure streaming kernel (HBM-bound).
#include <hip/hip_runtime.h>
#pragma once
#include <stdint.h>

#include "../../include/okx.h"

namespace okx {

struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
__device__ __forceinline__ Dual operator*(double s, Dual a) { return {s * a.v, s * a.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual dsqrt(Dual a) {
  const double r = sqrt(a.v);
  return {r, a.d / (2.0 * r)};
}
__device__ __forceinline__ Dual datan2(Dual y, Dual x) {
  return {atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v)};
}
__device__ __forceinline__ Dual dabs(Dual a) { return a.v < 0.0 ? -a : a; }

struct DVec {
  Dual x, y, z;
};
__device__ __forceinline__ DVec dsub(DVec a, DVec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

__device__ __forceinline__ DVec load_point(const double* pos, const double* vel, int k) {
  DVec p;
  p.x = {pos[3 * k], vel ? vel[3 * k] : 0.0};
  p.y = {pos[3 * k + 1], vel ? vel[3 * k + 1] : 0.0};
  p.z = {pos[3 * k + 2], vel ? vel[3 * k + 2] : 0.0};
  return p;
}

// The eight metrics of one state as duals (value, derivative along `vel`).
__device__ __forceinline__ void corner_metrics(const okx_corner_roles& R, const double* pos, const double* vel,
                                               Dual out[OKX_METRIC_COUNT]) {
  const double kDeg = 57.29577951308232;  // 180 / pi (numpy rad2deg)
  const double side = R.side_sign;
  const DVec wc = load_point(pos, vel, R.wheel_center), cp = load_point(pos, vel, R.contact_patch);
  const DVec axle = dsub(load_point(pos, vel, R.axle_outboard), load_point(pos, vel, R.axle_inboard));
  const DVec lower = load_point(pos, vel, R.steer_lower), upper = load_point(pos, vel, R.steer_upper);
  const DVec steer = dsub(upper, lower);
  // angles.py:22-50: wheel_up = (axle x X) * -side = -side * (0, axle_z, -axle_y); front-view angle from Z
  const Dual up_y = (-side) * axle.z, up_z = side * axle.y;
  const Dual angle = datan2(up_y, up_z);
  out[OKX_METRIC_CAMBER] = kDeg * (side > 0.0 ? angle : -angle);
  out[OKX_METRIC_CASTER] = kDeg * datan2(-steer.x, steer.z);           // angles.py:53-71
  out[OKX_METRIC_KPI] = kDeg * datan2((-side) * steer.y, steer.z);     // angles.py:74-94
  out[OKX_METRIC_ROADWHEEL_ANGLE] =                                     // angles.py:97-132 (toe)
      kDeg * (side > 0.0 ? datan2(axle.x, axle.y) : datan2(axle.x, -axle.y));
  out[OKX_METRIC_WHEEL_TRAVEL] = wc.z - Dual{R.design_wheel_center_z, 0.0};  // travel.py:19-32
  out[OKX_METRIC_HALF_TRACK] = dabs(cp.y);                                   // travel.py:35-45
  // context.py:119-138: steering axis meets the horizontal plane through the contact patch
  const Dual t = (cp.z - lower.z) / steer.z;
  const Dual gx = lower.x + t * steer.x, gy = lower.y + t * steer.y;
  // steering_geometry.py:22-54: offset along the wheel axis projected into the ground plane
  const Dual an = dsqrt(axle.x * axle.x + axle.y * axle.y);
  out[OKX_METRIC_SCRUB_RADIUS] = -(((gx - cp.x) * axle.x + (gy - cp.y) * axle.y) / an);
  out[OKX_METRIC_MECHANICAL_TRAIL] = gx - cp.x;  // steering_geometry.py:57-76
}

struct MetricsArgs {
  okx_corner_roles roles;
  const double* pos;      // [B][n_out][3]
  const double* tan;      // [B][T][n_out][3] or null
  double* metrics;        // [B][OKX_METRIC_COUNT]
  double* dmetrics;       // [B][T][OKX_METRIC_COUNT] or null
  long long n_states;
  int n_out, n_targets;
};

__global__ void __launch_bounds__(256) okx_corner_metrics_kernel(MetricsArgs a) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.n_states) return;
  const double* pos = a.pos + b * 3 * a.n_out;
  Dual m[OKX_METRIC_COUNT];
  corner_metrics(a.roles, pos, nullptr, m);
  double* out = a.metrics + b * OKX_METRIC_COUNT;
#pragma unroll
  for (int k = 0; k < OKX_METRIC_COUNT; ++k) out[k] = m[k].v;
  if (a.tan && a.dmetrics)
    for (int t = 0; t < a.n_targets; ++t) {
      corner_metrics(a.roles, pos, a.tan + (b * a.n_targets + t) * 3 * a.n_out, m);
      double* dout = a.dmetrics + (b * a.n_targets + t) * OKX_METRIC_COUNT;
#pragma unroll
      for (int k = 0; k < OKX_METRIC_COUNT; ++k) dout[k] = m[k].d;
    }
}

}  // namespace okx
