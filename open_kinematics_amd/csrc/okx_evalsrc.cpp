// okx_evalsrc.cpp — source text shared by the two kernel generators for the EVALUATED solve (okx_solve_evaluated_batch):
// the corner metric catalog on forward-mode duals with N tangent directions, specialised to one set of metric roles.
//
// Reference: core/sweep.py:217-270 (solve_evaluated_sweep: solve -> tangents -> metrics in one call),
// core/metrics/angles.py:22-132, travel.py:19-62, steering_geometry.py:22-76, swing_arms.py:45-88,
// anti_geometry.py:32-206 over MetricContext (context.py:82-138); the derivative columns (metrics/derivatives.py) are the
// duals' derivative parts.  Same formulas as okx_metrics.hip (the stand-alone metric kernels, IEEE division / libm), here
// with the generated kernels' Newton-refined reciprocals and roots: what a solve kernel's epilogue can afford.
//
// The role POINTS are compile-time constants of the generated module (a register-resident state has no run-time index);
// everything numeric about the roles (side sign, design references, vehicle data) is a kernel argument (`EvCfg`).
#include <cstdio>
#include <string>

#include "okx_quad.hpp"

namespace okx {

bool eval_spec_from_roles(const DevProgram& P, const okx_corner_roles& R, EvalSpec* spec, std::string* why) {
  auto bad = [&](int k, bool optional) { return optional ? (k < -1 || k >= P.n_out) : (k < 0 || k >= P.n_out); };
  const int must[6] = {R.wheel_center, R.contact_patch, R.axle_inboard, R.axle_outboard, R.steer_lower, R.steer_upper};
  for (int k = 0; k < 6; ++k)
    if (bad(must[k], false)) {
      *why = "metric role " + std::to_string(k) + " is not an output point";
      return false;
    }
  const int n_axis = R.instant_axis_kind == OKX_IA_TWO_PLANES ? 6 : R.instant_axis_kind == OKX_IA_PLANE_AND_STRUT ? 4
                     : R.instant_axis_kind == OKX_IA_NONE ? 0 : -1;
  if (n_axis < 0) {
    *why = "unknown instant_axis_kind";
    return false;
  }
  for (int k = 0; k < n_axis; ++k)
    if (bad(R.instant_axis_point[k], false)) {
      *why = "instant-axis point " + std::to_string(k) + " is not an output point";
      return false;
    }
  if ((R.damper_top < 0) != (R.damper_bottom < 0) || bad(R.damper_top, true) || bad(R.damper_bottom, true) || bad(R.rack_attachment, true)) {
    *why = "damper / rack roles must be output points or -1";
    return false;
  }
  spec->wheel_center = R.wheel_center;
  spec->contact_patch = R.contact_patch;
  spec->axle_inboard = R.axle_inboard;
  spec->axle_outboard = R.axle_outboard;
  spec->steer_lower = R.steer_lower;
  spec->steer_upper = R.steer_upper;
  spec->ia_kind = R.instant_axis_kind;
  for (int k = 0; k < 6; ++k) spec->ia_point[k] = k < n_axis ? R.instant_axis_point[k] : -1;
  spec->damper_top = R.damper_top < 0 ? -1 : R.damper_top;
  spec->damper_bottom = R.damper_bottom < 0 ? -1 : R.damper_bottom;
  spec->rack = R.rack_attachment < 0 ? -1 : R.rack_attachment;
  return true;
}

void eval_scalars_from_roles(const okx_corner_roles& R, EvalScalars* s) {
  s->side_sign = R.side_sign;
  s->design_wheel_center_z = R.design_wheel_center_z;
  s->design_contact_patch_z = R.design_contact_patch_z;
  s->design_rack_y = R.design_rack_y;
  s->wheelbase = R.wheelbase;
  s->cg_z = R.cg_z;
  s->front_brake_bias = R.front_brake_bias;
  s->axle_position = R.axle_position;
  s->driven_axle = R.driven_axle;
}

int eval_slot_point(const EvalSpec& s, int slot) {
  switch (slot) {
    case 0: return s.wheel_center;
    case 1: return s.contact_patch;
    case 2: return s.axle_outboard;
    case 3: return s.axle_inboard;
    case 4: return s.steer_lower;
    case 5: return s.steer_upper;
    case 6: return s.damper_top;
    case 7: return s.damper_bottom;
    case 14: return s.rack;
    default: return slot >= 8 && slot < 14 ? s.ia_point[slot - 8] : -1;
  }
}

namespace {
const char* kEvalBody = R"SRC(
// ---- evaluated solve: the corner metric catalog on duals with N tangent directions (okx_evalsrc.cpp) ----
struct EvCfg {  // the numeric part of okx_corner_roles
  double side_sign, design_wheel_center_z, design_contact_patch_z, design_rack_y, wheelbase, cg_z, front_brake_bias;
  int axle_position, driven_axle;
};
#define EV_SLOT_WHEEL_CENTER 0
#define EV_SLOT_CONTACT_PATCH 1
#define EV_SLOT_AXLE_OUTBOARD 2
#define EV_SLOT_AXLE_INBOARD 3
#define EV_SLOT_STEER_LOWER 4
#define EV_SLOT_STEER_UPPER 5
#define EV_SLOT_DAMPER_TOP 6
#define EV_SLOT_DAMPER_BOTTOM 7
#define EV_SLOT_IA0 8
#define EV_SLOT_RACK 14
#define EV_SLOTS 15
#define EV_EPS_GEOMETRIC 1e-6
template <int N> struct Du { double v; double d[N]; };
template <int N> DEV Du<N> du_const(double v) { Du<N> r; r.v = v; for (int i = 0; i < N; ++i) r.d[i] = 0.0; return r; }
template <int N> DEV Du<N> du_nan() { Du<N> r; r.v = __builtin_nan(""); for (int i = 0; i < N; ++i) r.d[i] = __builtin_nan(""); return r; }
template <int N> DEV Du<N> operator+(Du<N> a, Du<N> b) { Du<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> DEV Du<N> operator-(Du<N> a, Du<N> b) { Du<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> DEV Du<N> operator-(Du<N> a) { Du<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> DEV Du<N> operator*(Du<N> a, Du<N> b) { Du<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = fma(a.v, b.d[i], a.d[i] * b.v); return r; }
template <int N> DEV Du<N> operator*(double s, Du<N> a) { Du<N> r; r.v = s * a.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> DEV Du<N> operator/(Du<N> a, Du<N> b) {
  const double ib = fast_rcp(b.v);
  Du<N> r; r.v = a.v * ib;
  for (int i = 0; i < N; ++i) r.d[i] = fma(-r.v, b.d[i], a.d[i]) * ib;
  return r;
}
template <int N> DEV Du<N> du_sqrt(Du<N> a) {
  double root, inv; fast_sqrt_rsqrt(a.v, &root, &inv);
  Du<N> r; r.v = root;
  for (int i = 0; i < N; ++i) r.d[i] = 0.5 * a.d[i] * inv;
  return r;
}
// atan2 over the full circle from the [0, pi] form (fdlibm-style reduction + odd polynomial, as the solve kernels' angle rows)
DEV double ev_atan2_pos(double y, double x) {
  const double ax = fabs(x);
  if (!(y > 0.0)) return x >= 0.0 ? 0.0 : 3.14159265358979311600e+00;
  if (ax == 0.0) return 1.57079632679489655800e+00;
  double t = y * fast_rcp(ax);
  double hi, lo;
  if (t < 0.4375) { hi = 0.0; lo = 0.0; }
  else if (t < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; t = (2.0 * t - 1.0) * fast_rcp(2.0 + t); }
  else if (t < 1.1875) { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; t = (t - 1.0) * fast_rcp(t + 1.0); }
  else if (t < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; t = (t - 1.5) * fast_rcp(1.0 + 1.5 * t); }
  else { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; t = -fast_rcp(t); }
  const double z = t * t, w = z * z;
  const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
       w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
  const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
       w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  const double at = hi - ((t * (s1 + s2) - lo) - t);
  return x > 0.0 ? at : 3.14159265358979311600e+00 - (at - 1.2246467991473531772e-16);
}
DEV double ev_atan2(double y, double x) { const double a = ev_atan2_pos(fabs(y), x); return y < 0.0 ? -a : a; }
template <int N> DEV Du<N> du_atan2(Du<N> y, Du<N> x) {
  const double ir = fast_rcp(fma(x.v, x.v, y.v * y.v));
  Du<N> r; r.v = ev_atan2(y.v, x.v);
  for (int i = 0; i < N; ++i) r.d[i] = fma(x.v, y.d[i], -(y.v * x.d[i])) * ir;
  return r;
}
template <int N> DEV Du<N> du_atan(Du<N> q) {
  const double ir = fast_rcp(fma(q.v, q.v, 1.0));
  Du<N> r; r.v = ev_atan2(q.v, 1.0);
  for (int i = 0; i < N; ++i) r.d[i] = q.d[i] * ir;
  return r;
}
template <int N> DEV Du<N> du_abs(Du<N> a) { return a.v < 0.0 ? -a : a; }
template <int N> struct DV { Du<N> x, y, z; };
template <int N> DEV DV<N> dv_sub(DV<N> a, DV<N> b) { DV<N> r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
template <int N> DEV DV<N> dv_add(DV<N> a, DV<N> b) { DV<N> r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
template <int N> DEV DV<N> dv_cross(DV<N> a, DV<N> b) {
  DV<N> r; r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x; return r;
}
template <int N> DEV Du<N> dv_dot(DV<N> a, DV<N> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int N> DEV DV<N> dv_scale(Du<N> s, DV<N> a) { DV<N> r; r.x = s * a.x; r.y = s * a.y; r.z = s * a.z; return r; }
// a / |a| given the norm: one reciprocal for the three components
template <int N> DEV DV<N> dv_unit(DV<N> a, Du<N> norm) {
  const Du<N> inv = du_const<N>(1.0) / norm;
  return dv_scale(inv, a);
}
// vector_utils/geometric.py:216-252: unit normal n and offset d of the plane n.x + d = 0 through a, b, c
template <int N> DEV bool ev_plane(DV<N> a, DV<N> b, DV<N> c, DV<N>* n, Du<N>* d) {
  const DV<N> raw = dv_cross(dv_sub(b, a), dv_sub(c, a));
  const Du<N> mag = du_sqrt(dv_dot(raw, raw));
  if (!(mag.v >= EV_EPS_GEOMETRIC)) return false;
  *n = dv_unit(raw, mag);
  *d = -dv_dot(*n, a);
  return true;
}
// geometric.py:255-290: line (point, unit direction) where two planes meet
template <int N> DEV bool ev_two_planes(DV<N> n1, Du<N> d1, DV<N> n2, Du<N> d2, DV<N>* point, DV<N>* dir) {
  const DV<N> raw = dv_cross(n1, n2);
  const Du<N> m2 = dv_dot(raw, raw);
  if (!(m2.v >= EV_EPS_GEOMETRIC * EV_EPS_GEOMETRIC)) return false;
  const DV<N> w = dv_sub(dv_scale(d2, n1), dv_scale(d1, n2));
  *point = dv_scale(du_const<N>(1.0) / m2, dv_cross(w, raw));
  *dir = dv_unit(raw, du_sqrt(m2));
  return true;
}
// corner/double_wishbone.py:376-403, corner/macpherson.py:325-355
template <int N> DEV bool ev_instant_axis(const DV<N>* P, DV<N>* point, DV<N>* dir) {
  DV<N> n1, n2;
  Du<N> d1, d2;
#if EV_IA_KIND == 1
  if (!ev_plane(P[EV_SLOT_IA0 + 0], P[EV_SLOT_IA0 + 1], P[EV_SLOT_IA0 + 2], &n1, &d1) ||
      !ev_plane(P[EV_SLOT_IA0 + 3], P[EV_SLOT_IA0 + 4], P[EV_SLOT_IA0 + 5], &n2, &d2))
    return false;
#elif EV_IA_KIND == 2
  const DV<N> ball = P[EV_SLOT_IA0 + 2], top = P[EV_SLOT_IA0 + 3];
  if (!ev_plane(P[EV_SLOT_IA0 + 0], P[EV_SLOT_IA0 + 1], ball, &n1, &d1)) return false;
  const DV<N> strut = dv_sub(top, ball);
  n2 = dv_unit(strut, du_sqrt(dv_dot(strut, strut)));
  d2 = -dv_dot(n2, top);
#else
  return false;
#endif
  return ev_two_planes(n1, d1, n2, d2, point, dir);
}
// geometric.py:316-352: the line meets the plane {coordinate `axis` = value}
template <int N> DEV bool ev_line_at(DV<N> point, DV<N> dir, int axis, Du<N> value, DV<N>* hit) {
  const Du<N> comp = axis == 0 ? dir.x : (axis == 1 ? dir.y : dir.z);
  if (!(fabs(comp.v) >= EV_EPS_GEOMETRIC)) return false;
  const Du<N> from = axis == 0 ? point.x : (axis == 1 ? point.y : point.z);
  *hit = dv_add(point, dv_scale((value - from) / comp, dir));
  return true;
}
// The catalog (include/okx.h OKX_METRIC_*) of one state: values and derivatives along the N directions of its role points.
template <int N> DEV void ev_corner_metrics(const EvCfg& R, const DV<N>* P, Du<N>* out) {
  const double kDeg = 57.29577951308232;  // 180 / pi (numpy rad2deg)
  const double side = R.side_sign;
  const DV<N> wc = P[EV_SLOT_WHEEL_CENTER], cp = P[EV_SLOT_CONTACT_PATCH];
  const DV<N> axle = dv_sub(P[EV_SLOT_AXLE_OUTBOARD], P[EV_SLOT_AXLE_INBOARD]);
  const DV<N> lower = P[EV_SLOT_STEER_LOWER], upper = P[EV_SLOT_STEER_UPPER];
  const DV<N> steer = dv_sub(upper, lower);
  // angles.py:22-50: wheel_up = (axle x X) * -side = -side * (0, axle_z, -axle_y); front-view angle from Z
  const Du<N> up_y = (-side) * axle.z, up_z = side * axle.y;
  const Du<N> angle = du_atan2(up_y, up_z);
  out[0] = kDeg * (side > 0.0 ? angle : -angle);                 // camber
  out[1] = kDeg * du_atan2(-steer.x, steer.z);                   // caster, angles.py:53-71
  out[2] = kDeg * du_atan2((-side) * steer.y, steer.z);          // kpi, angles.py:74-94
  out[3] = kDeg * (side > 0.0 ? du_atan2(axle.x, axle.y) : du_atan2(axle.x, -axle.y));  // roadwheel angle, angles.py:97-132
  out[4] = wc.z - du_const<N>(R.design_wheel_center_z);          // wheel travel, travel.py:19-32
  out[5] = du_abs(cp.y);                                          // half track, travel.py:35-45
  // context.py:119-138: steering axis meets the horizontal plane through the contact patch
  const Du<N> t = (cp.z - lower.z) / steer.z;
  const Du<N> gx = lower.x + t * steer.x, gy = lower.y + t * steer.y;
  // steering_geometry.py:22-54: offset along the wheel axis projected into the ground plane
  const Du<N> an = du_sqrt(axle.x * axle.x + axle.y * axle.y);
  out[6] = -(((gx - cp.x) * axle.x + (gy - cp.y) * axle.y) / an);  // scrub radius
  out[7] = gx - cp.x;                                               // mechanical trail, steering_geometry.py:57-76
#if EV_HAS_DAMPER
  { const DV<N> strut = dv_sub(P[EV_SLOT_DAMPER_TOP], P[EV_SLOT_DAMPER_BOTTOM]);
    out[14] = du_sqrt(dv_dot(strut, strut)); }                      // damper length, travel.py:48-62
#else
  out[14] = du_nan<N>();
#endif
  for (int k = 8; k <= 13; ++k) out[k] = du_nan<N>();
  for (int k = 15; k <= 18; ++k) out[k] = du_nan<N>();
  DV<N> ap, ad, svic, fvic;
  if (!ev_instant_axis(P, &ap, &ad)) return;
  if (ev_line_at(ap, ad, 0, wc.x, &fvic)) {  // double_wishbone.py:405-430
    out[11] = fvic.y;
    out[12] = fvic.z;
    // swing_arms.py:62-88: distance in the front view, positive when the centre is inboard of the patch
    const Du<N> dy = fvic.y - cp.y, dz = fvic.z - cp.z;
    const double sgn = dy.v > 0.0 ? 1.0 : (dy.v < 0.0 ? -1.0 : 0.0);
    out[13] = (-side * sgn) * du_sqrt(dy * dy + dz * dz);
  }
  if (!ev_line_at(ap, ad, 1, wc.y, &svic)) return;  // double_wishbone.py:352-374
  out[8] = svic.x;
  out[9] = svic.z;
  out[10] = svic.x - cp.x;  // swing_arms.py:45-59
  const Du<N> run = svic.x - cp.x, rise = svic.z - cp.z;
  const bool run_ok = fabs(run.v) >= EV_EPS_GEOMETRIC;
  if (run_ok) out[15] = kDeg * du_atan(rise / run);  // anti_geometry.py:32-58
  const Du<N> height = du_const<N>(R.cg_z) - cp.z;   // anti_geometry.py:61-72
  const bool height_ok = height.v > EV_EPS_GEOMETRIC;
  const bool bias_set = R.front_brake_bias == R.front_brake_bias;
  const Du<N> lever = du_const<N>(R.wheelbase) / height;
  if (run_ok && height_ok && bias_set && R.axle_position == 1)  // anti_geometry.py:75-116 (front)
    out[16] = ((100.0 * R.front_brake_bias) * lever) * (rise / -run);
  if (run_ok && height_ok && bias_set && R.axle_position == 2)  // anti_geometry.py:119-160 (rear)
    out[17] = ((100.0 * (1.0 - R.front_brake_bias)) * lever) * (rise / run);
  if (R.driven_axle != 0 && R.driven_axle == R.axle_position) {  // anti_geometry.py:163-206
    const Du<N> drive_run = R.axle_position == 1 ? wc.x - svic.x : svic.x - wc.x;
    if (fabs(drive_run.v) >= EV_EPS_GEOMETRIC && height_ok) out[18] = (100.0 * lever) * ((svic.z - wc.z) / drive_run);
  }
}
)SRC";
}  // namespace

bool axle_eval_spec_from_roles(const DevProgram& P, const okx_axle_roles& R, AxleEvalSpec* spec, std::string* why) {
  if (!eval_spec_from_roles(P, R.left, &spec->side[0], why)) {
    *why = "left corner: " + *why;
    return false;
  }
  if (!eval_spec_from_roles(P, R.right, &spec->side[1], why)) {
    *why = "right corner: " + *why;
    return false;
  }
  const EvalSpec &a = spec->side[0], &b = spec->side[1];
  if (a.ia_kind != b.ia_kind || (a.damper_top < 0) != (b.damper_top < 0) || (a.rack < 0) != (b.rack < 0)) {
    *why = "the two corners of an evaluated axle must share the instant-axis construction and the damper / rack roles";
    return false;
  }
  if (R.n_roles < 0 || R.n_roles > 8) {
    *why = "n_roles out of range";
    return false;
  }
  spec->n_roles = R.n_roles;
  for (int k = 0; k < 8; ++k) spec->role[k] = {0, -1, -1};
  for (int k = 0; k < R.n_roles; ++k) {
    const okx_rotation_role& r = R.roles[k];
    if (r.kind < OKX_ROLE_AXIS_ROTATION || r.kind > OKX_ROLE_MIDPOINT_COORDINATE) {
      *why = "unknown role kind";
      return false;
    }
    const bool two = r.kind != OKX_ROLE_AXIS_ROTATION;
    if (r.point < 0 || r.point >= P.n_out || (two && (r.point_b < 0 || r.point_b >= P.n_out))) {
      *why = "role " + std::to_string(k) + " names a point outside the output list";
      return false;
    }
    spec->role[k] = {r.kind, r.point, two ? r.point_b : -1};
  }
  return true;
}

namespace {
// okx_metrics.hip axis_rotation_deg on the evaluated modules' duals (metrics/kernels.py:58-76, geometric.py:31-52;
// axle/mechanisms.py:718-815,903-944 for the two-point kinds).  KIND is a compile-time constant of the call site.
const char* kRolesBody = R"SRC(
struct EvRoleNum { double design[3], axis_point[3], axis_dir[3], scale; };
template <int N> DEV DV<N> dv_const(double x, double y, double z) { DV<N> r; r.x = du_const<N>(x); r.y = du_const<N>(y); r.z = du_const<N>(z); return r; }
template <int KIND, int N> DEV Du<N> ev_role(const EvRoleNum& R, DV<N> moving, DV<N> other) {
  const double kDeg = 57.29577951308232;
  const DV<N> a = dv_const<N>(R.axis_dir[0], R.axis_dir[1], R.axis_dir[2]);
  const DV<N> origin = dv_const<N>(R.axis_point[0], R.axis_point[1], R.axis_point[2]);
  if (KIND != 0) {
    const DV<N> span = dv_sub(moving, other);
    if (KIND == 3) return du_sqrt(dv_dot(span, span));
    const DV<N> mid = dv_add(moving, dv_scale(du_const<N>(0.5), dv_sub(other, moving)));
    if (KIND == 4) return dv_dot(a, dv_sub(mid, origin));
    if (KIND == 2) {
      DV<N> stem = dv_sub(mid, origin);
      const Du<N> len = du_sqrt(dv_dot(stem, stem));
      if (!(len.v >= EV_EPS_GEOMETRIC)) return du_nan<N>();
      stem = dv_unit(stem, len);
      const DV<N> crossbar = dv_sub(span, dv_scale(dv_dot(span, stem), stem));
      Du<N> twist = kDeg * du_atan2(dv_dot(stem, dv_cross(a, crossbar)), dv_dot(crossbar, a));
      twist.v -= R.design[0];
      return twist;
    }
    moving = mid;
  }
  const DV<N> dr = dv_const<N>(R.design[0] - R.axis_point[0], R.design[1] - R.axis_point[1], R.design[2] - R.axis_point[2]);
  const DV<N> cr = dv_sub(moving, origin);
  const DV<N> dperp = dv_sub(dr, dv_scale(dv_dot(dr, a), a)), cperp = dv_sub(cr, dv_scale(dv_dot(cr, a), a));
  const double dn = dv_dot(dperp, dperp).v, cn = dv_dot(cperp, cperp).v;
  if (!(dn >= EV_EPS_GEOMETRIC * EV_EPS_GEOMETRIC) || !(cn >= EV_EPS_GEOMETRIC * EV_EPS_GEOMETRIC)) return du_nan<N>();
  return (R.scale * kDeg) * du_atan2(dv_dot(a, dv_cross(dr, cr)), dv_dot(dperp, cperp));
}
)SRC";
}  // namespace

std::string eval_roles_source() { return kRolesBody; }

std::string eval_metrics_source(const EvalSpec& spec) {
  char line[128];
  std::string out;
  std::snprintf(line, sizeof(line), "#define EV_IA_KIND %d\n#define EV_HAS_DAMPER %d\n#define EV_HAS_RACK %d\n", spec.ia_kind,
                spec.damper_top >= 0 ? 1 : 0, spec.rack >= 0 ? 1 : 0);
  out += line;
  out += "// metric roles (output-list indices):";
  for (int s = 0; s < kEvalSlots; ++s) out += " " + std::to_string(eval_slot_point(spec, s));
  out += "\n";
  out += kEvalBody;
  return out;
}

}  // namespace okx
