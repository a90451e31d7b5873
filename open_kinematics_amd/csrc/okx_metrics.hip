// okx_metrics.hip — corner state metrics of solved states and their derivatives along the
// solution-manifold tangents.
//
// Reference: core/metrics/angles.py:22-132 (camber, caster, KPI, toe / roadwheel angle),
// travel.py:19-45 (wheel travel, half-track), steering_geometry.py:22-76 (scrub radius,
// mechanical trail) over MetricContext (context.py:82-138); the derivative columns
// (metrics/derivatives.py: d response / d driver along a tangent field) are forward-mode
// derivatives of the same formulas, which is what `Dual` below carries.
//
// One thread per solved state: up to fourteen role points in (336 B), nineteen scalars out (152 B),
// plus 152 B per target when tangents are given — a pure streaming kernel (HBM-bound).
// Instant centres, swing arms and the anti-geometry follow swing_arms.py / anti_geometry.py over
// the corner's instant axis; axle-scope metrics follow axle_metrics.py.
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

__device__ __forceinline__ Dual dnan() { return {__builtin_nan(""), __builtin_nan("")}; }
__device__ __forceinline__ Dual datan(Dual q) { return {atan(q.v), q.d / (1.0 + q.v * q.v)}; }
__device__ __forceinline__ DVec dcross(DVec a, DVec b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Dual ddot(DVec a, DVec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ DVec dscale(Dual s, DVec a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ DVec dadd(DVec a, DVec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ DVec dunit(DVec a, Dual norm) { return {a.x / norm, a.y / norm, a.z / norm}; }

constexpr double kEpsGeometric = 1e-6;  // primitives/constants.py:9

// vector_utils/geometric.py:216-252: unit normal n and offset d of the plane n.x + d = 0 through a, b, c
__device__ __forceinline__ bool plane_from_three_points(DVec a, DVec b, DVec c, DVec* n, Dual* d) {
  const DVec raw = dcross(dsub(b, a), dsub(c, a));
  const Dual mag = dsqrt(ddot(raw, raw));
  if (!(mag.v >= kEpsGeometric)) return false;
  *n = dunit(raw, mag);
  *d = -ddot(*n, a);
  return true;
}

// geometric.py:255-290: line (point, unit direction) where two planes meet
__device__ __forceinline__ bool intersect_two_planes(DVec n1, Dual d1, DVec n2, Dual d2, DVec* point, DVec* dir) {
  const DVec raw = dcross(n1, n2);
  const Dual m2 = ddot(raw, raw);
  if (!(m2.v >= kEpsGeometric * kEpsGeometric)) return false;
  const DVec w = dsub(dscale(d2, n1), dscale(d1, n2));
  const DVec pc = dcross(w, raw);
  *point = {pc.x / m2, pc.y / m2, pc.z / m2};
  *dir = dunit(raw, dsqrt(m2));
  return true;
}

// corner/double_wishbone.py:376-403, corner/macpherson.py:325-355
__device__ __forceinline__ bool instant_axis(const okx_corner_roles& R, const double* pos, const double* vel,
                                             DVec* point, DVec* dir) {
  const int32_t* ip = R.instant_axis_point;
  DVec n1, n2;
  Dual d1, d2;
  if (R.instant_axis_kind == OKX_IA_TWO_PLANES) {
    if (!plane_from_three_points(load_point(pos, vel, ip[0]), load_point(pos, vel, ip[1]), load_point(pos, vel, ip[2]),
                                 &n1, &d1) ||
        !plane_from_three_points(load_point(pos, vel, ip[3]), load_point(pos, vel, ip[4]), load_point(pos, vel, ip[5]),
                                 &n2, &d2))
      return false;
  } else if (R.instant_axis_kind == OKX_IA_PLANE_AND_STRUT) {
    const DVec ball = load_point(pos, vel, ip[2]), top = load_point(pos, vel, ip[3]);
    if (!plane_from_three_points(load_point(pos, vel, ip[0]), load_point(pos, vel, ip[1]), ball, &n1, &d1)) return false;
    const DVec strut = dsub(top, ball);
    n2 = dunit(strut, dsqrt(ddot(strut, strut)));
    d2 = -ddot(n2, top);
  } else {
    return false;
  }
  return intersect_two_planes(n1, d1, n2, d2, point, dir);
}

// geometric.py:316-352: the line meets the plane {coordinate `axis` = value}
__device__ __forceinline__ bool line_at_coordinate(DVec point, DVec dir, int axis, Dual value, DVec* hit) {
  const Dual comp = axis == 0 ? dir.x : (axis == 1 ? dir.y : dir.z);
  if (!(fabs(comp.v) >= kEpsGeometric)) return false;
  const Dual from = axis == 0 ? point.x : (axis == 1 ? point.y : point.z);
  *hit = dadd(point, dscale((value - from) / comp, dir));
  return true;
}

// The catalog's metrics of one state as duals (value, derivative along `vel`).
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

  // travel.py:48-62
  if (R.damper_top >= 0 && R.damper_bottom >= 0) {
    const DVec strut = dsub(load_point(pos, vel, R.damper_top), load_point(pos, vel, R.damper_bottom));
    out[OKX_METRIC_DAMPER_LENGTH] = dsqrt(ddot(strut, strut));
  } else {
    out[OKX_METRIC_DAMPER_LENGTH] = dnan();
  }

  // instant centres: the instant axis cut at the wheel centre's y (side view) and x (front view)
  for (int k = OKX_METRIC_SVIC_X; k <= OKX_METRIC_FVSA_LENGTH; ++k) out[k] = dnan();
  for (int k = OKX_METRIC_SVSA_ANGLE; k <= OKX_METRIC_ANTI_SQUAT; ++k) out[k] = dnan();
  DVec ap, ad, svic, fvic;
  if (!instant_axis(R, pos, vel, &ap, &ad)) return;
  if (line_at_coordinate(ap, ad, 0, wc.x, &fvic)) {  // double_wishbone.py:405-430
    out[OKX_METRIC_FVIC_Y] = fvic.y;
    out[OKX_METRIC_FVIC_Z] = fvic.z;
    // swing_arms.py:62-88: distance in the front view, positive when the centre is inboard of the patch
    const Dual dy = fvic.y - cp.y, dz = fvic.z - cp.z;
    const double sgn = dy.v > 0.0 ? 1.0 : (dy.v < 0.0 ? -1.0 : 0.0);
    out[OKX_METRIC_FVSA_LENGTH] = (-side * sgn) * dsqrt(dy * dy + dz * dz);
  }
  if (!line_at_coordinate(ap, ad, 1, wc.y, &svic)) return;  // double_wishbone.py:352-374
  out[OKX_METRIC_SVIC_X] = svic.x;
  out[OKX_METRIC_SVIC_Z] = svic.z;
  out[OKX_METRIC_SVSA_LENGTH] = svic.x - cp.x;  // swing_arms.py:45-59
  const Dual run = svic.x - cp.x, rise = svic.z - cp.z;
  const bool run_ok = fabs(run.v) >= kEpsGeometric;
  if (run_ok) out[OKX_METRIC_SVSA_ANGLE] = kDeg * datan(rise / run);  // anti_geometry.py:32-58
  const Dual height = Dual{R.cg_z, 0.0} - cp.z;                       // anti_geometry.py:61-72
  const bool height_ok = height.v > kEpsGeometric;
  const bool bias_set = R.front_brake_bias == R.front_brake_bias;
  const Dual lever = Dual{R.wheelbase, 0.0} / height;
  if (run_ok && height_ok && bias_set && R.axle_position == OKX_AXLE_FRONT)  // anti_geometry.py:75-116
    out[OKX_METRIC_ANTI_DIVE] = ((100.0 * R.front_brake_bias) * lever) * (rise / -run);
  if (run_ok && height_ok && bias_set && R.axle_position == OKX_AXLE_REAR)   // anti_geometry.py:119-160
    out[OKX_METRIC_ANTI_LIFT] = ((100.0 * (1.0 - R.front_brake_bias)) * lever) * (rise / run);
  if (R.driven_axle != OKX_AXLE_UNSET && R.driven_axle == R.axle_position) {  // anti_geometry.py:163-206
    const Dual drive_run = R.axle_position == OKX_AXLE_FRONT ? wc.x - svic.x : svic.x - wc.x;
    if (fabs(drive_run.v) >= kEpsGeometric && height_ok)
      out[OKX_METRIC_ANTI_SQUAT] = (100.0 * lever) * ((svic.z - wc.z) / drive_run);
  }
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

struct AxleMetricsArgs {
  okx_corner_roles left, right;
  const double* pos;  // [B][n_out][3]
  double* metrics;    // [B][OKX_AXLE_METRIC_COUNT]
  long long n_states;
  int n_out;
};

// metrics/axle_metrics.py:21-95: one thread per solved axle state.
__global__ void __launch_bounds__(256) okx_axle_metrics_kernel(AxleMetricsArgs a) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.n_states) return;
  const double* pos = a.pos + b * 3 * a.n_out;
  const double kDeg = 57.29577951308232;
  const double nan = __builtin_nan("");
  double wheel_dz[2], contact_dz[2], line[2][4];
  bool have_lines = true;
  for (int s = 0; s < 2; ++s) {
    const okx_corner_roles& R = s == 0 ? a.left : a.right;
    const double* wc = pos + 3 * R.wheel_center;
    const double* cp = pos + 3 * R.contact_patch;
    wheel_dz[s] = wc[2] - R.design_wheel_center_z;
    contact_dz[s] = cp[2] - R.design_contact_patch_z;
    DVec ap, ad, fvic;
    const DVec wcd = load_point(pos, nullptr, R.wheel_center);
    if (instant_axis(R, pos, nullptr, &ap, &ad) && line_at_coordinate(ap, ad, 0, wcd.x, &fvic)) {
      line[s][0] = cp[1];
      line[s][1] = cp[2];
      line[s][2] = fvic.y.v - cp[1];
      line[s][3] = fvic.z.v - cp[2];
    } else {
      have_lines = false;
    }
  }
  double* out = a.metrics + b * OKX_AXLE_METRIC_COUNT;
  const double track = fabs(pos[3 * a.left.contact_patch + 1] - pos[3 * a.right.contact_patch + 1]);
  out[OKX_AXLE_METRIC_HEAVE] = 0.5 * (wheel_dz[0] + wheel_dz[1]);
  out[OKX_AXLE_METRIC_ROLL] = kDeg * atan2(wheel_dz[0] - wheel_dz[1], track);
  out[OKX_AXLE_METRIC_RIDE_HEIGHT_CHANGE] = -0.5 * (contact_dz[0] + contact_dz[1]);
  out[OKX_AXLE_METRIC_TRACK] = track;
  double rcy = nan, rcz = nan;
  if (have_lines) {  // axle_metrics.py:73-95: the two contact-patch -> FVIC lines in the YZ plane
    const double den = line[0][2] * line[1][3] - line[0][3] * line[1][2];
    if (fabs(den) >= kEpsGeometric) {
      const double t = ((line[1][0] - line[0][0]) * line[1][3] - (line[1][1] - line[0][1]) * line[1][2]) / den;
      rcy = line[0][0] + t * line[0][2];
      rcz = line[0][1] + t * line[0][3];
    }
  }
  out[OKX_AXLE_METRIC_ROLL_CENTER_Y] = rcy;
  out[OKX_AXLE_METRIC_ROLL_CENTER_Z] = rcz;
  out[OKX_AXLE_METRIC_RACK_DISPLACEMENT] =
      a.left.rack_attachment >= 0 ? pos[3 * a.left.rack_attachment + 1] - a.left.design_rack_y : nan;
}

struct RotationArgs {
  okx_rotation_role roles[OKX_MAX_ROTATIONS];
  int n_roles;
  const double* pos;   // [B][n_out][3]
  const double* tan;   // [B][T][n_out][3] or null
  double* angles;      // [B][n_roles]
  double* dangles;     // [B][T][n_roles] or null
  long long n_states;
  int n_out, n_targets;
};

// metrics/kernels.py:58-76 on duals; geometric.py:47-51: undefined for a point on the axis.  The other kinds: the
// hardware metrics of a composed axle (okx.h OKX_ROLE_*; axle/mechanisms.py:718-815, :903-944), same duals.
__device__ __forceinline__ Dual axis_rotation_deg(const okx_rotation_role& R, const double* pos, const double* vel) {
  const double kDeg = 57.29577951308232;
  const DVec a = {{R.axis_dir[0], 0.0}, {R.axis_dir[1], 0.0}, {R.axis_dir[2], 0.0}};
  const DVec origin = {{R.axis_point[0], 0.0}, {R.axis_point[1], 0.0}, {R.axis_point[2], 0.0}};
  DVec moving = load_point(pos, vel, R.point);
  if (R.kind != OKX_ROLE_AXIS_ROTATION) {
    const DVec other = load_point(pos, vel, R.point_b);
    const DVec span = dsub(moving, other);  // `point` - `point_b` (left end - right end of a crossbar)
    if (R.kind == OKX_ROLE_DISTANCE) return dsqrt(ddot(span, span));
    const Dual half = {0.5, 0.0};
    const DVec mid = dadd(moving, dscale(half, dsub(other, moving)));  // a + (b - a) / 2, the reference's midpoint
    if (R.kind == OKX_ROLE_MIDPOINT_COORDINATE) return ddot(a, dsub(mid, origin));
    if (R.kind == OKX_ROLE_STEM_TWIST) {  // mechanisms.py:800-815
      DVec stem = dsub(mid, origin);
      const Dual len = dsqrt(ddot(stem, stem));
      if (!(len.v >= kEpsGeometric)) return dnan();
      stem = dunit(stem, len);
      const DVec crossbar = dsub(span, dscale(ddot(span, stem), stem));
      const Dual twist = kDeg * datan2(ddot(stem, dcross(a, crossbar)), ddot(crossbar, a));
      return {twist.v - R.design[0], twist.d};
    }
    moving = mid;  // OKX_ROLE_MIDPOINT_ROTATION: the midpoint about the fixed axis
  }
  const DVec dr = {{R.design[0] - R.axis_point[0], 0.0}, {R.design[1] - R.axis_point[1], 0.0}, {R.design[2] - R.axis_point[2], 0.0}};
  const DVec cr = dsub(moving, origin);
  const DVec dperp = dsub(dr, dscale(ddot(dr, a), a)), cperp = dsub(cr, dscale(ddot(cr, a), a));
  if (!(sqrt(ddot(dperp, dperp).v) >= kEpsGeometric) || !(sqrt(ddot(cperp, cperp).v) >= kEpsGeometric)) return dnan();
  return (R.scale * kDeg) * datan2(ddot(a, dcross(dr, cr)), ddot(dperp, cperp));
}

__global__ void __launch_bounds__(256) okx_axis_rotation_kernel(RotationArgs a) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.n_states) return;
  const double* pos = a.pos + b * 3 * a.n_out;
  for (int k = 0; k < a.n_roles; ++k) a.angles[b * a.n_roles + k] = axis_rotation_deg(a.roles[k], pos, nullptr).v;
  if (a.tan && a.dangles)
    for (int t = 0; t < a.n_targets; ++t) {
      const double* vel = a.tan + (b * a.n_targets + t) * 3 * a.n_out;
      for (int k = 0; k < a.n_roles; ++k)
        a.dangles[(b * a.n_targets + t) * a.n_roles + k] = axis_rotation_deg(a.roles[k], pos, vel).d;
    }
}

}  // namespace okx
