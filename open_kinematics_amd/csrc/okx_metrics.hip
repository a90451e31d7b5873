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

// Where a state's role points come from.  The metric formulas name their points by ROLE SLOT (a compile-time constant at
// every use); a source turns a slot into the point as a dual vector.
enum RoleSlot {
  SLOT_WHEEL_CENTER, SLOT_CONTACT_PATCH, SLOT_AXLE_OUTBOARD, SLOT_AXLE_INBOARD, SLOT_STEER_LOWER, SLOT_STEER_UPPER,
  SLOT_DAMPER_TOP, SLOT_DAMPER_BOTTOM, SLOT_INSTANT_AXIS_0,  // ... SLOT_INSTANT_AXIS_0 + 5
  SLOT_COUNT = SLOT_INSTANT_AXIS_0 + 6
};
__device__ __forceinline__ int role_point(const okx_corner_roles& R, int slot) {
  switch (slot) {
    case SLOT_WHEEL_CENTER: return R.wheel_center;
    case SLOT_CONTACT_PATCH: return R.contact_patch;
    case SLOT_AXLE_OUTBOARD: return R.axle_outboard;
    case SLOT_AXLE_INBOARD: return R.axle_inboard;
    case SLOT_STEER_LOWER: return R.steer_lower;
    case SLOT_STEER_UPPER: return R.steer_upper;
    case SLOT_DAMPER_TOP: return R.damper_top;
    case SLOT_DAMPER_BOTTOM: return R.damper_bottom;
    default: return R.instant_axis_point[slot - SLOT_INSTANT_AXIS_0];
  }
}
// ... the record (and a tangent row, or none) read where the point is used: one thread per state
struct RecordPoints {
  const okx_corner_roles& R;
  const double* pos;
  const double* vel;
  __device__ __forceinline__ DVec at(int slot) const { return load_point(pos, vel, role_point(R, slot)); }
};
// ... positions gathered into registers once per state, the tangent row (or none) read where it is used (the tiled kernels)
struct GatheredPoints {
  const okx_corner_roles& R;
  double pv[SLOT_COUNT][3];
  const double* vel;
  __device__ __forceinline__ void gather(const double* pos) {
#pragma unroll
    for (int s = 0; s < SLOT_COUNT; ++s) {
      const int k = role_point(R, s);  // (an absent role, -1: its value is never used)
#pragma unroll
      for (int c = 0; c < 3; ++c) pv[s][c] = pos[3 * (k < 0 ? 0 : k) + c];
    }
  }
  __device__ __forceinline__ DVec at(int slot) const {
    const int k = role_point(R, slot);
    DVec p;
    p.x = {pv[slot][0], vel ? vel[3 * k] : 0.0};
    p.y = {pv[slot][1], vel ? vel[3 * k + 1] : 0.0};
    p.z = {pv[slot][2], vel ? vel[3 * k + 2] : 0.0};
    return p;
  }
};

// corner/double_wishbone.py:376-403, corner/macpherson.py:325-355
template <class Points>
__device__ __forceinline__ bool instant_axis(const okx_corner_roles& R, const Points& pts, DVec* point, DVec* dir) {
  constexpr int ip = SLOT_INSTANT_AXIS_0;
  DVec n1, n2;
  Dual d1, d2;
  if (R.instant_axis_kind == OKX_IA_TWO_PLANES) {
    if (!plane_from_three_points(pts.at(ip + 0), pts.at(ip + 1), pts.at(ip + 2), &n1, &d1) ||
        !plane_from_three_points(pts.at(ip + 3), pts.at(ip + 4), pts.at(ip + 5), &n2, &d2))
      return false;
  } else if (R.instant_axis_kind == OKX_IA_PLANE_AND_STRUT) {
    const DVec ball = pts.at(ip + 2), top = pts.at(ip + 3);
    if (!plane_from_three_points(pts.at(ip + 0), pts.at(ip + 1), ball, &n1, &d1)) return false;
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
template <class Points>
__device__ __forceinline__ void corner_metrics(const okx_corner_roles& R, const Points& pts, Dual out[OKX_METRIC_COUNT]) {
  const double kDeg = 57.29577951308232;  // 180 / pi (numpy rad2deg)
  const double side = R.side_sign;
  const DVec wc = pts.at(SLOT_WHEEL_CENTER), cp = pts.at(SLOT_CONTACT_PATCH);
  const DVec axle = dsub(pts.at(SLOT_AXLE_OUTBOARD), pts.at(SLOT_AXLE_INBOARD));
  const DVec lower = pts.at(SLOT_STEER_LOWER), upper = pts.at(SLOT_STEER_UPPER);
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
    const DVec strut = dsub(pts.at(SLOT_DAMPER_TOP), pts.at(SLOT_DAMPER_BOTTOM));
    out[OKX_METRIC_DAMPER_LENGTH] = dsqrt(ddot(strut, strut));
  } else {
    out[OKX_METRIC_DAMPER_LENGTH] = dnan();
  }

  // instant centres: the instant axis cut at the wheel centre's y (side view) and x (front view)
  for (int k = OKX_METRIC_SVIC_X; k <= OKX_METRIC_FVSA_LENGTH; ++k) out[k] = dnan();
  for (int k = OKX_METRIC_SVSA_ANGLE; k <= OKX_METRIC_ANTI_SQUAT; ++k) out[k] = dnan();
  DVec ap, ad, svic, fvic;
  if (!instant_axis(R, pts, &ap, &ad)) return;
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
  corner_metrics(a.roles, RecordPoints{a.roles, pos, nullptr}, m);
  double* out = a.metrics + b * OKX_METRIC_COUNT;
#pragma unroll
  for (int k = 0; k < OKX_METRIC_COUNT; ++k) out[k] = m[k].v;
  if (a.tan && a.dmetrics)
    for (int t = 0; t < a.n_targets; ++t) {
      corner_metrics(a.roles, RecordPoints{a.roles, pos, a.tan + (b * a.n_targets + t) * 3 * a.n_out}, m);
      double* dout = a.dmetrics + (b * a.n_targets + t) * OKX_METRIC_COUNT;
#pragma unroll
      for (int k = 0; k < OKX_METRIC_COUNT; ++k) dout[k] = m[k].d;
    }
}

// ---- the same metrics, records staged through LDS (the forms okx_corner_metrics_batch launches for corners: records of
// up to 42 points) ----
// One thread per state reads its own 360-byte record at a 360-byte stride: every load instruction of the kernel above
// touches 64 different lines and a workgroup's records (92 KB) do not stay in the L1 (29 - 45 % of HBM,
// profiles/r04/EXPERIMENTS.md section 9).  Here a wavefront takes a TILE of 64 consecutive states: their records are one
// contiguous block, copied into LDS with coalesced loads (row stride odd in doubles: the per-state reads that follow are
// conflict-free).  Every lane gathers its state's fourteen role points into registers, which frees the buffer: the
// tangent rows of one target at a time ([B][T][n_out][3]: 64 segments of one record each) go through the same 23 KB,
// and so do the 19 results per state on their way out (one contiguous block per tile).  Same arithmetic, same bits.
constexpr int kTileStates = 64;

// e / d for e < 2^16 by one multiply-high (inv = 2^32 / d + 1, exact in that range)
__device__ __forceinline__ uint32_t tile_div(uint32_t e, uint32_t inv) { return __umulhi(e, inv); }

// `n` doubles at `src` (rows of `rec` doubles, back to back) -> dst[row * stride + col]
__device__ __forceinline__ void stage_contiguous(double* dst, uint32_t stride, const double* __restrict__ src, uint32_t n,
                                                 uint32_t rec, uint32_t rec_inv, int lane) {
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src);
    const uint32_t n2 = n >> 1;
    constexpr int kBatch = 6;  // loads in flight per lane and batch (6 KB per wavefront)
    for (uint32_t i0 = 0; i0 < n2; i0 += kTileStates * kBatch) {
      double2 v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const uint32_t i = i0 + kTileStates * u + lane;
        if (i < n2) v[u] = s2[i];
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const uint32_t i = i0 + kTileStates * u + lane;
        if (i < n2) {
          const uint32_t e = 2 * i, q = tile_div(e, rec_inv), c = e - q * rec;
          dst[q * stride + c] = v[u].x;
          const bool wrap = c + 1 == rec;
          dst[(wrap ? q + 1 : q) * stride + (wrap ? 0 : c + 1)] = v[u].y;
        }
      }
    }
    if ((n & 1) && lane == 0) {
      const uint32_t e = n - 1, q = tile_div(e, rec_inv);
      dst[q * stride + (e - q * rec)] = src[e];
    }
  } else {
    for (uint32_t e = lane; e < n; e += kTileStates) {
      const uint32_t q = tile_div(e, rec_inv);
      dst[q * stride + (e - q * rec)] = src[e];
    }
  }
}

// a tile's results, staged as [64][19], to their contiguous block in HBM
__device__ __forceinline__ void store_tile_results(double* __restrict__ dst, const double* stage, uint32_t n, int lane) {
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    for (uint32_t i = lane; i < n / 2; i += kTileStates) reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(stage)[i];
    if ((n & 1) && lane == 0) dst[n - 1] = stage[n - 1];
  } else {
    for (uint32_t e = lane; e < n; e += kTileStates) dst[e] = stage[e];
  }
}

struct TileArgs {
  uint32_t rec, rec_inv, stride;  // doubles per record, 2^32 / rec + 1, LDS row stride (odd)
};

// One tile per workgroup.  Values alone: 100 registers, the LDS buffer bounds the occupancy at seven wavefronts per CU.
// With derivative columns (1 + T evaluations per state, the dual ones at ~270 registers): one wavefront per SIMD.
// (Measured and dropped, profiles/r04/EXPERIMENTS.md section 9: the dual kernel held to 256 registers for two
//  wavefronts per SIMD - 44 B of scratch, slower; persistent wavefronts prefetching the next phase's rows into
//  registers - 160 B of scratch, slower.)
template <bool TAN>
__global__ void __launch_bounds__(kTileStates) okx_corner_metrics_tiled(MetricsArgs a, TileArgs ta) {
  extern __shared__ double okx_tile_lds[];  // [64][max(stride, 19)]: records, then tangent rows of one target / results
  double* const buf = okx_tile_lds;
  const int lane = threadIdx.x;
  const long long b0 = (long long)blockIdx.x * kTileStates;
  const long long left = a.n_states - b0;
  const uint32_t rows = left < kTileStates ? (uint32_t)left : kTileStates;
  const uint32_t rec = ta.rec, stride = ta.stride;
  // (a ragged last tile: the lanes beyond it work on the tile's last record and store nothing)
  const uint32_t my_row = (lane < (int)rows ? lane : (int)rows - 1) * stride;
  stage_contiguous(buf, stride, a.pos + b0 * rec, rows * rec, rec, ta.rec_inv, lane);
  __syncthreads();
  GatheredPoints pts{a.roles};
  pts.gather(buf + my_row);
  pts.vel = nullptr;
  Dual m[OKX_METRIC_COUNT];
  corner_metrics(a.roles, pts, m);
  __syncthreads();  // (every lane has gathered its points: the buffer is free)
#pragma unroll
  for (int k = 0; k < OKX_METRIC_COUNT; ++k) buf[lane * OKX_METRIC_COUNT + k] = m[k].v;
  __syncthreads();
  store_tile_results(a.metrics + b0 * OKX_METRIC_COUNT, buf, rows * OKX_METRIC_COUNT, lane);
  if (TAN) {
    const int T = a.n_targets;
    for (int t = 0; t < T; ++t) {
      // tangent rows of target t: state q's row starts at ((b0 + q) T + t) rec
      const double* __restrict__ src = a.tan + (b0 * T + t) * rec;
      const uint32_t n = rows * rec, hop = (uint32_t)(T - 1) * rec;
      constexpr int kBatch = 8;
      for (uint32_t e0 = 0; e0 < n; e0 += kTileStates * kBatch) {
        double v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const uint32_t e = e0 + kTileStates * u + lane;
          if (e < n) v[u] = src[e + tile_div(e, ta.rec_inv) * hop];
        }
        if (e0 == 0) __syncthreads();  // (the results staged in the buffer have been read; the loads above are in flight)
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const uint32_t e = e0 + kTileStates * u + lane;
          if (e < n) {
            const uint32_t q = tile_div(e, ta.rec_inv);
            buf[q * stride + (e - q * rec)] = v[u];
          }
        }
      }
      __syncthreads();
      pts.vel = buf + my_row;
      corner_metrics(a.roles, pts, m);
      __syncthreads();  // (every lane has read its tangent row)
#pragma unroll
      for (int k = 0; k < OKX_METRIC_COUNT; ++k) buf[lane * OKX_METRIC_COUNT + k] = m[k].d;
      __syncthreads();
      double* __restrict__ dst = a.dmetrics + (b0 * T + t) * OKX_METRIC_COUNT;
      const uint32_t no = rows * OKX_METRIC_COUNT, ohop = (uint32_t)(T - 1) * OKX_METRIC_COUNT;
      for (uint32_t e = lane; e < no; e += kTileStates) dst[e + (e / OKX_METRIC_COUNT) * ohop] = buf[e];
    }
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
    if (instant_axis(R, RecordPoints{R, pos, nullptr}, &ap, &ad) && line_at_coordinate(ap, ad, 0, wcd.x, &fvic)) {
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
