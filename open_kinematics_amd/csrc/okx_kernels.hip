// okx_kernels.hip — hand-written gfx950 (CDNA4) kernels of the batched constraint solver.
//
// Execution model: ONE WAVEFRONT (64 lanes) OWNS ONE SWEEP-STEP PROBLEM, one wavefront per
// workgroup, persistent workgroups striding over the batch.  Everything a problem needs
// between its first and last Levenberg-Marquardt iteration lives in that workgroup's LDS
// slice (positions, block-sparse Jacobian, n x n normal matrix) or in lane registers
// (lane j owns variable j: x_j, g_j, dx_j); HBM is touched only to read the targets
// (8*T bytes) and to write the solved points (24*n_out bytes) and the info record.
//
//   rows        lane i evaluates constraint row i (and i+64): residual + partials
//               (reference constraints.py / jacobians.py), chain rule through derived
//               points with closed-form 3x3 blocks (reference manager.py:271-324 uses
//               dual numbers), scattered into a block-sparse row (<= 6 blocks of 3).
//   normal eq.  J^T J and J^T r from host-built contribution plans (okx_plan.cpp): only
//               structurally non-zero 3x3 blocks are formed.
//   solve       wavefront-cooperative Cholesky of (J^T J + lambda I) in LDS, lane i owns
//               row i; triangular solves broadcast the pivot unknown with v_readlane.
//   LM          Nielsen gain-ratio damping; accept/reject decided uniformly by the wave.
//
// No MFMA: systems are tens of unknowns, fp64, block-sparse — see DESIGN.md §5.
#include <hip/hip_runtime.h>
#pragma once
#include <stdint.h>

#include "okx_plan.hpp"

namespace okx {

#define OKX_EPS_SQ 1e-12
#define OKX_EPS 1e-6

struct SolveArgs {
  const double* targets;         // [B][T]
  const double* geom_pos;        // [G][P][3] or null
  const double* geom_row_param;  // [G][Mc][8] or null
  double* out_pos;               // [B][n_out][3]
  okx_info* info;                // [B]
  long long n_problems;
  long long steps_per_geometry;  // 0: single geometry
  int max_iter;
  int confirm;                   // non-zero: always end on a computed correction (no predicted-convergence test)
  long long chain_len;           // problems per warm-started chain (>= 1)
  double step_tol, grad_tol, ftol, lambda0, residual_tolerance;
  unsigned long long* phase_cycles;  // diagnostic build only: [8] per-phase cycle sums of block 0
};

// ------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------

// Broadcast lane `k`'s value; k must be wave-uniform (v_readlane_b32 x2).
__device__ __forceinline__ double wave_bcast(double v, int k) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, k);
  hi = __builtin_amdgcn_readlane(hi, k);
  return __hiloint2double(hi, lo);
}

// One DPP data movement of a double; lanes without a valid source (or masked rows) get 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Wave-wide reductions in the DPP network (no LDS crossbar): row_shr 1/2/4/8 builds the
// per-16-lane row totals, row_bcast:15 / row_bcast:31 chain them into lane 63.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0x111, 0xf>(v);
  v += dpp_move<0x112, 0xf>(v);
  v += dpp_move<0x114, 0xf>(v);
  v += dpp_move<0x118, 0xf>(v);
  v += dpp_move<0x142, 0xa>(v);
  v += dpp_move<0x143, 0xc>(v);
  return wave_bcast(v, 63);
}

// Maximum of NON-NEGATIVE values (identity 0).
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move<0x111, 0xf>(v));
  v = fmax(v, dpp_move<0x112, 0xf>(v));
  v = fmax(v, dpp_move<0x114, 0xf>(v));
  v = fmax(v, dpp_move<0x118, 0xf>(v));
  v = fmax(v, dpp_move<0x142, 0xa>(v));
  v = fmax(v, dpp_move<0x143, 0xc>(v));
  return wave_bcast(v, 63);
}

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// Reductions over the W threads that share one problem: one wavefront (W = 64: the DPP network alone) or, for programs of
// more than 63 variables, a workgroup of two wavefronts whose partial results meet in two LDS slots (`red`).  Every
// thread of the group gets the same bits, so the control flow built on them stays uniform.
template <int W>
__device__ __forceinline__ double grp_sum(double* red, double v) {
  v = wave_sum(v);
  if constexpr (W > kWave) {
    __syncthreads();  // the readers of the previous reduction are done with red[]
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    v = red[0] + red[1];
  }
  return v;
}

template <int W>
__device__ __forceinline__ double grp_max(double* red, double v) {
  v = wave_max(v);
  if constexpr (W > kWave) {
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    v = fmax(red[0], red[1]);
  }
  return v;
}

// Diagnostic phase timer (only alive in the PROFILE instantiation; a null pointer folds away).
struct Prof {
  unsigned long long phase[12];
  unsigned long long t;
};
__device__ __forceinline__ void stamp(Prof* prof, int slot) {
  if (prof) {
    const unsigned long long now = __builtin_readcyclecounter();
    prof->phase[slot] += now - prof->t;
    prof->t = now;
  }
}

__device__ __forceinline__ double sel3(int r, double a, double b, double c) {
  return r == 0 ? a : (r == 1 ? b : c);
}

__device__ __forceinline__ double softnorm(double s) { return sqrt(s + OKX_EPS_SQ) - OKX_EPS; }

// 1/x to ~1 ulp from the v_rcp_f64 seed and two Newton steps.  The IEEE divide sequence
// (v_div_scale / v_div_fmas / v_div_fixup) sits on every dependent chain of this kernel at
// ~200 cycles; operands here are well-scaled lengths and pivots (never denormal, never huge),
// which is all the scaling/fix-up steps protect against.
__device__ __forceinline__ double fast_rcp(double x) {
#if defined(OKX_IEEE_MATH)
  return 1.0 / x;
#endif
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  e = fma(-x, r, 1.0);
  return fma(e, r, r);
}

// atan2(y, x) for y >= 0 (the angle rows pass y = softnorm(|v1 x v2|^2) >= 0), result in
// [0, pi].  The math-library atan2 costs 44 VGPRs in this kernel (182 -> 138 without it,
// measured with -Rpass-analysis), which alone decides between 3 waves/SIMD with spills and
// without.  This is the classic fdlibm scheme (Sun Microsystems' freely distributable
// s_atan.c / e_atan2.c): four-interval argument reduction and an odd polynomial of degree 21,
// < 1 ulp, evaluated with Horner chains so it needs a handful of registers.
__device__ __forceinline__ double lean_atan2_pos(double y, double x) {
  const double ax = fabs(x);
  if (!(y > 0.0)) return x >= 0.0 ? 0.0 : 3.14159265358979311600e+00;
  if (ax == 0.0) return 1.57079632679489655800e+00;
  double t = y * fast_rcp(ax);  // t = |y / x| >= 0
  double hi, lo;
  if (t < 0.4375) {
    hi = 0.0;
    lo = 0.0;
  } else if (t < 0.6875) {
    hi = 4.63647609000806093515e-01;
    lo = 2.26987774529616870924e-17;
    t = (2.0 * t - 1.0) * fast_rcp(2.0 + t);
  } else if (t < 1.1875) {
    hi = 7.85398163397448278999e-01;
    lo = 3.06161699786838301793e-17;
    t = (t - 1.0) * fast_rcp(t + 1.0);
  } else if (t < 2.4375) {
    hi = 9.82793723247329054082e-01;
    lo = 1.39033110312309984516e-17;
    t = (t - 1.5) * fast_rcp(1.0 + 1.5 * t);
  } else {
    hi = 1.57079632679489655800e+00;
    lo = 6.12323399573676603587e-17;
    t = -fast_rcp(t);
  }
  const double z = t * t, w = z * z;
  const double s1 = z * (3.33333333333329318027e-01 +
                         w * (1.42857142725034663711e-01 +
                              w * (9.09088713343650656196e-02 +
                                   w * (6.66107313738753120669e-02 +
                                        w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
  const double s2 = w * (-1.99999999998764832476e-01 +
                         w * (-1.11111104054623557880e-01 +
                              w * (-7.69187620504482999495e-02 +
                                   w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  const double at = hi - ((t * (s1 + s2) - lo) - t);  // atan(|y/x|) in [0, pi/2]
  return x > 0.0 ? at : 3.14159265358979311600e+00 - (at - 1.2246467991473531772e-16);
}

// sqrt(x) and 1/sqrt(x) together (Goldschmidt from the v_rsq_f64 seed), x > 0 well scaled.
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double* root, double* inv) {
#if defined(OKX_IEEE_MATH)
  *root = sqrt(x);
  *inv = 1.0 / *root;
  return;
#endif
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  *root = g;
  *inv = h + h;
}

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ------------------------------------------------------------------------------------
// LDS carve-up (offsets in doubles; sizes from DevProgram)
// ------------------------------------------------------------------------------------

constexpr int kColBuf = 26;         // N + 1 entries for N <= 24, padded to an even count
constexpr int kRowMetaStride = 15;  // ints per row: type, pts[4], nblk, 4 x PointRef (odd stride)

struct Lds {
  double* pos;    // [P][3]
  double* rowq;   // [m][8]   row parameters of the current geometry
  double* dblk;   // [n_active][kDepMax][3][3]
  double* js;     // [2][m][js_stride]
  double* rb;     // [2][m]
  double* A;      // [n(n-1)/2] packed strict triangle, entry (i, j < i) at i(i-1)/2 + j:
                  //            J^T J before the factorisation loads it, then the factor L in place
  double* dA;     // [n]       diag(J^T J)
  double* tv;     // [T]       targets of the current problem
  double* col;    // [kColBuf] pivot column being broadcast by the factorisation
  double* zbuf;   // [kColBuf] right-hand side in / z out
  double* red;    // [4]       two-wavefront groups: partial reductions [0..1], pivot / broadcast slots [2..3]
  // program tables staged once per workgroup (static for the whole launch)
  int* rowmeta;              // [m][kRowMetaStride]
  int* item_dst;             // [n_work]
  const unsigned int* item_terms;  // GLOBAL (L1/L2 resident) [n_work][kc]  offA | offB << 16
  const unsigned int* grad_terms;  // GLOBAL [n][kg]       offJ | row << 16
};

// One Jacobian buffer = m rows + the always-zero slot the padded plan terms point at; one
// residual buffer = m residuals + a zero.
__host__ __device__ inline int js_buf_doubles(const DevProgram& P) { return P.zero_off + 1; }
__host__ __device__ inline int rb_buf_doubles(const DevProgram& P) { return P.m + 1; }
// programs of more than 63 variables (two wavefronts per problem, LDL^T in LDS) carry the right-hand side as row n
__host__ __device__ inline bool wide_program(const DevProgram& P) { return P.n > kWave - 1; }
__host__ __device__ inline int tri_doubles(const DevProgram& P) {
  const int rows = wide_program(P) ? P.n + 1 : P.n;
  return rows * (rows - 1) / 2 + 1;
}
__host__ __device__ inline int tri(int i, int j) { return i * (i - 1) / 2 + j; }  // j < i

__host__ __device__ inline int lds_table_doubles(const DevProgram& P) {
  int ints = P.m * kRowMetaStride + P.n_work + 2;
  return (ints + 1) / 2;
}

__host__ __device__ inline int lds_doubles(const DevProgram& P) {
  int s = 0;
  s += P.n_points * 3;
  s += P.m * 8;
  s += (P.n_active > 0 ? P.n_active : 1) * kDepMax * 9;
  s += js_buf_doubles(P);
  s += 2 * rb_buf_doubles(P);
  s += tri_doubles(P);
  s += P.n;
  s += kMaxTargets;
  s += 2 * kColBuf + 4;
  s = (s + 1) & ~1;
  s += lds_table_doubles(P);
  return (s + 1) & ~1;
}

// Shared, launch-static tables: 16-byte aligned term tables first (read with ds_read_b128).
__device__ __forceinline__ void carve_tables(int* q, const DevProgram* P, Lds* S) {
  S->item_terms = P->item_terms;  // launch-static: read straight from HBM through L1/L2,
  S->grad_terms = P->grad_terms;  // every load of an item is issued at once (no dependent walk)
  S->item_dst = q;
  q += P->n_work;
  S->rowmeta = q;
}

__device__ __forceinline__ Lds carve(double* base, const DevProgram* P) {
  Lds S;
  double* p = base;
  S.pos = p;
  p += P->n_points * 3;
  S.rowq = p;
  p += P->m * 8;
  S.dblk = p;
  p += (P->n_active > 0 ? P->n_active : 1) * kDepMax * 9;
  S.js = p;
  p += js_buf_doubles(*P);
  S.rb = p;
  p += 2 * rb_buf_doubles(*P);
  S.A = p;
  p += tri_doubles(*P);
  S.dA = p;
  p += P->n;
  S.tv = p;
  p += kMaxTargets;
  S.col = p;
  p += kColBuf;
  S.zbuf = p;
  p += kColBuf;
  S.red = p;
  p += 4;
  p = base + (((p - base) + 1) & ~1);
  carve_tables(reinterpret_cast<int*>(p), P, &S);
  return S;
}

// Copy the static program tables into LDS (once per persistent workgroup).
__device__ __forceinline__ void stage_program(const DevProgram* P, const Lds& S, int lane, int W = kWave) {
  for (int i = lane; i < P->m; i += W) {
    int* r = S.rowmeta + i * kRowMetaStride;
    r[0] = P->row_type[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[1 + k] = P->row_pts[i][k];
    r[5] = P->row_nblk[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r[6 + 2 * k] = (int)P->row_in[i][k].w0;
      r[7 + 2 * k] = (int)P->row_in[i][k].w1;
    }
    r[14] = (int)P->row_first[i];
  }
  for (int w = lane; w < P->n_work; w += W) S.item_dst[w] = P->item_dst[w];
  __syncthreads();
}

// Per-problem slice: zero the matrix (structural zeros stay zero) and the padding slots.
__device__ __forceinline__ void init_slice(const DevProgram* P, const Lds& S, int l, int W) {
  for (int e = l; e < tri_doubles(*P); e += W) S.A[e] = 0.0;
  if (l < 2) S.rb[l * rb_buf_doubles(*P) + P->m] = 0.0;
  if (l == 0) S.js[P->zero_off] = 0.0;
  __syncthreads();
}

// ------------------------------------------------------------------------------------
// derived points
// ------------------------------------------------------------------------------------

// Position of one derived op (reference points/derived/definitions.py); uniform over lanes.
// `at(pt)`: the current position of point pt
template <class At>
__device__ __forceinline__ V3 dop_position_at(int type, const int* pts, double c, At at,
                                              V3* u_out, double* nrm_out, V3* a_out,
                                              double* vn_out, double* ga_out) {
  if (type == OKX_DOP_MIDPOINT) {  // definitions.py:76-89
    V3 a = at(pts[0]), b = at(pts[1]);
    return {a.x + (b.x - a.x) / 2, a.y + (b.y - a.y) / 2, a.z + (b.z - a.z) / 2};
  }
  if (type == OKX_DOP_ALONG) {  // definitions.py:24-33, :92-155
    V3 base = at(pts[0]);
    V3 v = sub(at(pts[1]), at(pts[2]));
    double nrm, inrm;
    fast_sqrt_rsqrt(v.x * v.x + v.y * v.y + v.z * v.z, &nrm, &inrm);
    V3 u = {v.x * inrm, v.y * inrm, v.z * inrm};
    *u_out = u;
    *nrm_out = inrm;  // callers want 1/|v|
    return {base.x + u.x * c, base.y + u.y * c, base.z + u.z * c};
  }
  // OKX_DOP_CONTACT_PATCH: definitions.py:36-73, :158-180
  V3 wc = at(pts[0]);
  V3 v = sub(at(pts[2]), at(pts[1]));
  double vn, ivn;
  fast_sqrt_rsqrt(v.x * v.x + v.y * v.y + v.z * v.z, &vn, &ivn);
  V3 a = {v.x * ivn, v.y * ivn, v.z * ivn};
  double ga = -a.z;  // (0,0,-1) . a
  V3 wd = {-ga * a.x, -ga * a.y, -1.0 - ga * a.z};
  double wn, iwn;
  fast_sqrt_rsqrt(wd.x * wd.x + wd.y * wd.y + wd.z * wd.z, &wn, &iwn);
  V3 wu = {wd.x * iwn, wd.y * iwn, wd.z * iwn};
  *u_out = wu;
  *nrm_out = iwn;  // callers want the inverse norms
  *a_out = a;
  *vn_out = ivn;
  *ga_out = ga;
  return {wc.x + wu.x * c, wc.y + wu.y * c, wc.z + wu.z * c};
}
__device__ __forceinline__ V3 dop_position(int type, const int* pts, double c, const double* pos,
                                           V3* u_out, double* nrm_out, V3* a_out,
                                           double* vn_out, double* ga_out) {
  return dop_position_at(type, pts, c, [pos](int pt) { return ld3(pos + 3 * pt); }, u_out, nrm_out, a_out, vn_out, ga_out);
}

// Row r of the local 3x3 block d(out)/d(input s) of a derived op (inrm, ivn: INVERSE norms).
__device__ __forceinline__ V3 dop_local_row(int type, int s, int r, double c, V3 u, double inrm,
                                            V3 a, double ivn, double ga) {
  V3 e = {r == 0 ? 1.0 : 0.0, r == 1 ? 1.0 : 0.0, r == 2 ? 1.0 : 0.0};
  if (type == OKX_DOP_MIDPOINT) return {0.5 * e.x, 0.5 * e.y, 0.5 * e.z};
  if (type == OKX_DOP_ALONG) {
    if (s == 0) return e;
    // d normalize(v)/dv = (I - u u^T)/|v|, scaled by c; input 2 enters v with a minus sign
    double ur = sel3(r, u.x, u.y, u.z);
    double k = (s == 1 ? c : -c) * inrm;
    return {k * (e.x - ur * u.x), k * (e.y - ur * u.y), k * (e.z - ur * u.z)};
  }
  // contact patch: input 0 = wheel centre (identity), 1 = axle inboard (-T), 2 = axle outboard (+T)
  if (s == 0) return e;
  // T = c * Nw * Wa * Na,  Nw = (I - wu wu^T)/wn (u,nrm hold wu,wn), Wa = a e_z^T - ga I,
  // Na = (I - a a^T)/vn
  double wr = sel3(r, u.x, u.y, u.z);
  V3 nw = {(e.x - wr * u.x) * inrm, (e.y - wr * u.y) * inrm, (e.z - wr * u.z) * inrm};
  double nwa = dot(nw, a);
  V3 mrow = {-ga * nw.x, -ga * nw.y, nwa - ga * nw.z};  // M[r][q] = d_q2 (Nw_r.a) - ga Nw[r][q]
  double ma = dot(mrow, a);
  double k = (s == 2 ? c : -c) * ivn;
  return {k * (mrow.x - ma * a.x), k * (mrow.y - ma * a.y), k * (mrow.z - ma * a.z)};
}

// Evaluate the listed derived ops in order.  WITH_BLOCKS also fills the chain-rule blocks
// d(out)/d(free block) of active ops: lane (je*9 + r*3 + col) computes one entry.
// `l` is the lane's index inside its problem group and `W` the group width (64 when one
// wavefront owns one problem); S carries the group's own LDS slice.
template <bool WITH_BLOCKS>
__device__ __forceinline__ void derived_update(const DevProgram* P, const Lds& S, int l, int W,
                                               bool active_only) {
  const int count = active_only ? P->n_active : P->n_derived;
  for (int idx = 0; idx < count; ++idx) {
    const int e = active_only ? P->active_op[idx] : idx;
    const int type = P->dop_type[e];
    const double c = P->dop_param[e];
    V3 u = {0, 0, 0}, a = {0, 0, 0};
    double nrm = 1.0, vn = 1.0, ga = 0.0;
    V3 o = dop_position(type, P->dop_pts[e], c, S.pos, &u, &nrm, &a, &vn, &ga);
    if (WITH_BLOCKS) {
      const int act = P->dop_active[e];
      const int nblk = P->dop_nblk[e];
      for (int ent = l; act >= 0 && ent < 9 * nblk; ent += W) {
        const int je = ent / 9, rc = ent % 9, r = rc / 3, col = rc % 3;
        double val = 0.0;
        const int nin = type == OKX_DOP_MIDPOINT ? 2 : 3;
        for (int s = 0; s < nin; ++s) {
          const PointRef ref = P->dop_in[e][s];
          if (ref.kind() == kRefFixed) continue;
          V3 row = dop_local_row(type, s, r, c, u, nrm, a, vn, ga);
          if (ref.kind() == kRefFree) {
            if (ref.slot() == je) val += sel3(col, row.x, row.y, row.z);
          } else {
            const double* src = S.dblk + P->dop_active[ref.src()] * (kDepMax * 9);
            for (int js = 0; js < ref.nsrc(); ++js)
              if (ref.map(js) == je) {
                const double* B = src + js * 9;
                val += row.x * B[0 + col] + row.y * B[3 + col] + row.z * B[6 + col];
              }
          }
        }
        S.dblk[act * (kDepMax * 9) + ent] = val;
      }
    }
    wave_sync();  // inputs of this op were read by every lane before its output is written
    if (l < 3) S.pos[3 * P->dop_out[e] + l] = sel3(l, o.x, o.y, o.z);
    wave_sync();
  }
}

// ------------------------------------------------------------------------------------
// constraint rows
// ------------------------------------------------------------------------------------

// Residual and partial derivatives of one row (reference constraints.py / jacobians.py).
// dp[3*s + k] = d r / d (coordinate k of the row's point slot s).
template <bool WITH_J>
__device__ __forceinline__ double row_eval(int type, const int* pts, const double* q,
                                           const double* pos, const double* tv, double* dp) {
#pragma unroll
  for (int k = 0; k < 12; ++k) dp[k] = 0.0;
  switch (type) {
    case OKX_ROW_DISTANCE:
    case OKX_ROW_SPHERICAL: {  // constraints.py:125-134,162-170; jacobians.py:35-51
      V3 d = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[0]));
      double s = d.x * d.x + d.y * d.y + d.z * d.z;
      double root, inv;
      fast_sqrt_rsqrt(s + OKX_EPS_SQ, &root, &inv);
      if (WITH_J) {
        dp[0] = -d.x * inv, dp[1] = -d.y * inv, dp[2] = -d.z * inv;
        dp[3] = d.x * inv, dp[4] = d.y * inv, dp[5] = d.z * inv;
      }
      double r = root - OKX_EPS;
      return type == OKX_ROW_DISTANCE ? r - q[0] : r;
    }
    case OKX_ROW_ANGLE:
    case OKX_ROW_THREE_POINT_ANGLE: {  // constraints.py:223-243,287-308; jacobians.py:55-188
      V3 v1, v2;
      if (type == OKX_ROW_ANGLE) {
        v1 = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[0]));
        v2 = sub(ld3(pos + 3 * pts[3]), ld3(pos + 3 * pts[2]));
      } else {
        v1 = sub(ld3(pos + 3 * pts[0]), ld3(pos + 3 * pts[1]));
        v2 = sub(ld3(pos + 3 * pts[2]), ld3(pos + 3 * pts[1]));
      }
      V3 c = cross(v1, v2);
      double c2 = c.x * c.x + c.y * c.y + c.z * c.z;
      double t15 = OKX_EPS_SQ + c2;
      double s, is;
      fast_sqrt_rsqrt(t15, &s, &is);
      double dt = dot(v1, v2);
      if (WITH_J) {
        double inv = fast_rcp(t15 + dt * dt);
        double ka = dt * inv * is, kb = s * inv;
        V3 w1 = cross(v2, c), w2 = cross(c, v1);
        V3 g1 = {ka * w1.x - kb * v2.x, ka * w1.y - kb * v2.y, ka * w1.z - kb * v2.z};
        V3 g2 = {ka * w2.x - kb * v1.x, ka * w2.y - kb * v1.y, ka * w2.z - kb * v1.z};
        if (type == OKX_ROW_ANGLE) {
          dp[0] = -g1.x, dp[1] = -g1.y, dp[2] = -g1.z;
          dp[3] = g1.x, dp[4] = g1.y, dp[5] = g1.z;
          dp[6] = -g2.x, dp[7] = -g2.y, dp[8] = -g2.z;
          dp[9] = g2.x, dp[10] = g2.y, dp[11] = g2.z;
        } else {
          dp[0] = g1.x, dp[1] = g1.y, dp[2] = g1.z;
          dp[3] = -g1.x - g2.x, dp[4] = -g1.y - g2.y, dp[5] = -g1.z - g2.z;
          dp[6] = g2.x, dp[7] = g2.y, dp[8] = g2.z;
        }
      }
      return lean_atan2_pos(s - OKX_EPS, dt) - q[0];
    }
    case OKX_ROW_VECTORS_PARALLEL: {  // constraints.py:351-371; jacobians.py:192-262
      V3 v1 = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[0]));
      V3 v2 = sub(ld3(pos + 3 * pts[3]), ld3(pos + 3 * pts[2]));
      V3 c = cross(v1, v2);
      double c2 = dot(c, c), n1 = dot(v1, v1), n2 = dot(v2, v2);
      double sc = sqrt(OKX_EPS_SQ + c2), s1 = sqrt(OKX_EPS_SQ + n1), s2 = sqrt(OKX_EPS_SQ + n2);
      if (WITH_J) {
        V3 w1 = cross(v2, c), w2 = cross(c, v1);
        double k26 = 1.0 / (s1 * s2 * sc), k19 = sc / (s2 * s1 * s1 * s1),
               k31 = sc / (s1 * s2 * s2 * s2);
        V3 g1 = {k26 * w1.x - k19 * v1.x, k26 * w1.y - k19 * v1.y, k26 * w1.z - k19 * v1.z};
        V3 g2 = {k26 * w2.x - k31 * v2.x, k26 * w2.y - k31 * v2.y, k26 * w2.z - k31 * v2.z};
        dp[0] = -g1.x, dp[1] = -g1.y, dp[2] = -g1.z, dp[3] = g1.x, dp[4] = g1.y, dp[5] = g1.z;
        dp[6] = -g2.x, dp[7] = -g2.y, dp[8] = -g2.z, dp[9] = g2.x, dp[10] = g2.y, dp[11] = g2.z;
      }
      return (sc - OKX_EPS) / ((s1 - OKX_EPS) * (s2 - OKX_EPS));
    }
    case OKX_ROW_VECTORS_PERPENDICULAR: {  // constraints.py:414-429; jacobians.py:266-318
      V3 v1 = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[0]));
      V3 v2 = sub(ld3(pos + 3 * pts[3]), ld3(pos + 3 * pts[2]));
      double n1 = dot(v1, v1), n2 = dot(v2, v2), dt = dot(v1, v2);
      double s1 = sqrt(OKX_EPS_SQ + n1), s2 = sqrt(OKX_EPS_SQ + n2);
      if (WITH_J) {
        double k16 = 1.0 / (s1 * s2), k18 = dt / (s2 * s1 * s1 * s1),
               k19 = dt / (s1 * s2 * s2 * s2);
        V3 g1 = {k16 * v2.x - k18 * v1.x, k16 * v2.y - k18 * v1.y, k16 * v2.z - k18 * v1.z};
        V3 g2 = {k16 * v1.x - k19 * v2.x, k16 * v1.y - k19 * v2.y, k16 * v1.z - k19 * v2.z};
        dp[0] = -g1.x, dp[1] = -g1.y, dp[2] = -g1.z, dp[3] = g1.x, dp[4] = g1.y, dp[5] = g1.z;
        dp[6] = -g2.x, dp[7] = -g2.y, dp[8] = -g2.z, dp[9] = g2.x, dp[10] = g2.y, dp[11] = g2.z;
      }
      return dt / ((s1 - OKX_EPS) * (s2 - OKX_EPS));
    }
    case OKX_ROW_EQUAL_DISTANCE: {  // constraints.py:466-477; jacobians.py:322-367
      V3 d1 = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[0]));
      V3 d2 = sub(ld3(pos + 3 * pts[3]), ld3(pos + 3 * pts[2]));
      double r1 = sqrt(OKX_EPS_SQ + dot(d1, d1)), r2 = sqrt(OKX_EPS_SQ + dot(d2, d2));
      if (WITH_J) {
        double i1 = 1.0 / r1, i2 = 1.0 / r2;
        dp[0] = -d1.x * i1, dp[1] = -d1.y * i1, dp[2] = -d1.z * i1;
        dp[3] = d1.x * i1, dp[4] = d1.y * i1, dp[5] = d1.z * i1;
        dp[6] = d2.x * i2, dp[7] = d2.y * i2, dp[8] = d2.z * i2;
        dp[9] = -d2.x * i2, dp[10] = -d2.y * i2, dp[11] = -d2.z * i2;
      }
      return (r1 - OKX_EPS) - (r2 - OKX_EPS);
    }
    case OKX_ROW_FIXED_AXIS: {  // constraints.py:508-516; solver.py:407-416
      int ax = (int)q[0];
      if (WITH_J) dp[0] = ax == 0 ? 1.0 : 0.0, dp[1] = ax == 1 ? 1.0 : 0.0, dp[2] = ax == 2 ? 1.0 : 0.0;
      V3 p = ld3(pos + 3 * pts[0]);
      return sel3(ax, p.x, p.y, p.z) - q[1];
    }
    case OKX_ROW_POINT_ON_LINE:
    case OKX_ROW_LINE_PIN: {  // constraints.py:560-576; jacobians.py:372-403; okx.h (pin)
      V3 w = sub(ld3(pos + 3 * pts[0]), ld3(q));
      V3 ld = ld3(q + 3);
      V3 c = cross(w, ld);
      if (type == OKX_ROW_POINT_ON_LINE) {
        double c2 = dot(c, c);
        double root, inv;
        fast_sqrt_rsqrt(OKX_EPS_SQ + c2, &root, &inv);
        if (WITH_J) {
          V3 g = cross(ld, c);
          dp[0] = inv * g.x, dp[1] = inv * g.y, dp[2] = inv * g.z;
        }
        return root - OKX_EPS;
      }
      int comp = (int)q[6];
      if (WITH_J) {
        // c = w x ld: dc_x = (0, ld.z, -ld.y), dc_y = (-ld.z, 0, ld.x), dc_z = (ld.y, -ld.x, 0)
        dp[0] = sel3(comp, 0.0, -ld.z, ld.y);
        dp[1] = sel3(comp, ld.z, 0.0, -ld.x);
        dp[2] = sel3(comp, -ld.y, ld.x, 0.0);
      }
      return sel3(comp, c.x, c.y, c.z);
    }
    case OKX_ROW_POINT_ON_PLANE: {  // constraints.py:616-627; solver.py:429-437
      V3 w = sub(ld3(pos + 3 * pts[0]), ld3(q));
      V3 nn = ld3(q + 3);
      if (WITH_J) dp[0] = nn.x, dp[1] = nn.y, dp[2] = nn.z;
      return dot(w, nn);
    }
    case OKX_ROW_MIDPOINT_ON_PLANE: {  // constraints.py:657-666; solver.py:439-448
      V3 a = ld3(pos + 3 * pts[0]), b = ld3(pos + 3 * pts[1]);
      V3 mid = {a.x + (b.x - a.x) / 2.0, a.y + (b.y - a.y) / 2.0, a.z + (b.z - a.z) / 2.0};
      V3 nn = ld3(q + 3);
      if (WITH_J) {
        dp[0] = dp[3] = 0.5 * nn.x;
        dp[1] = dp[4] = 0.5 * nn.y;
        dp[2] = dp[5] = 0.5 * nn.z;
      }
      return dot(sub(mid, ld3(q)), nn);
    }
    case OKX_ROW_COPLANAR:
    case OKX_ROW_SCALAR_TRIPLE: {  // constraints.py:698-709,731-733; jacobians.py:426-483
      V3 p1 = ld3(pos + 3 * pts[0]);
      V3 v1 = sub(ld3(pos + 3 * pts[1]), p1), v2 = sub(ld3(pos + 3 * pts[2]), p1),
         v3 = sub(ld3(pos + 3 * pts[3]), p1);
      V3 c23 = cross(v2, v3);
      double vol = dot(v1, c23);
      double sc = type == OKX_ROW_SCALAR_TRIPLE ? q[1] : 1.0;
      if (WITH_J) {
        V3 c31 = cross(v3, v1), c12 = cross(v1, v2);
        dp[3] = c23.x / sc, dp[4] = c23.y / sc, dp[5] = c23.z / sc;
        dp[6] = c31.x / sc, dp[7] = c31.y / sc, dp[8] = c31.z / sc;
        dp[9] = c12.x / sc, dp[10] = c12.y / sc, dp[11] = c12.z / sc;
        dp[0] = -(c23.x + c31.x + c12.x) / sc;
        dp[1] = -(c23.y + c31.y + c12.y) / sc;
        dp[2] = -(c23.z + c31.z + c12.z) / sc;
      }
      return type == OKX_ROW_SCALAR_TRIPLE ? (vol - q[0]) / q[1] : vol;
    }
    case kRowTarget: {  // solver.py:264-270, :560-579
      V3 dir = ld3(q);
      if (WITH_J) dp[0] = dir.x, dp[1] = dir.y, dp[2] = dir.z;
      return dot(ld3(pos + 3 * pts[0]), dir) - tv[(int)q[3]];
    }
    default:
      return 0.0;
  }
}

// Evaluate row i into buffer `buf`: residual -> rb, block-sparse Jacobian row -> js.
template <bool WITH_J>
__device__ __forceinline__ double row_pass(const DevProgram* P, const Lds& S, int i, int buf) {
  double dp[12];
  const int* meta = S.rowmeta + i * kRowMetaStride;
  const int type = meta[0];
  const int pts[4] = {meta[1], meta[2], meta[3], meta[4]};
  const double* q = S.rowq + 8 * i;
  double r = row_eval<WITH_J>(type, pts, q, S.pos, S.tv, dp);
  S.rb[buf * rb_buf_doubles(*P) + i] = r;
  if (WITH_J) {
    double* jr = S.js + (size_t)i * P->js_stride;  // one Jacobian buffer: a rejected trial never needs the old J
    const unsigned first = (unsigned)meta[14];  // bit (4 s + j): this write is the first to its slot
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      PointRef ref;
      ref.w0 = (unsigned)meta[6 + 2 * s];
      ref.w1 = (unsigned)meta[7 + 2 * s];
      if (ref.kind() == kRefFixed) continue;
      const double d0 = dp[3 * s], d1 = dp[3 * s + 1], d2 = dp[3 * s + 2];
      if (ref.kind() == kRefFree) {
        double* dst = jr + 3 * ref.slot();
        if ((first >> (4 * s)) & 1) {  // plain stores: no LDS read on the dependent chain
          dst[0] = d0;
          dst[1] = d1;
          dst[2] = d2;
        } else {
          dst[0] += d0;
          dst[1] += d1;
          dst[2] += d2;
        }
      } else {  // point_partial @ block (solver.py:554-558)
        const double* src = S.dblk + P->dop_active[ref.src()] * (kDepMax * 9);
        for (int js = 0; js < ref.nsrc(); ++js) {
          const double* B = src + 9 * js;
          double* dst = jr + 3 * ref.map(js);
          const double v0 = d0 * B[0] + d1 * B[3] + d2 * B[6];
          const double v1 = d0 * B[1] + d1 * B[4] + d2 * B[7];
          const double v2 = d0 * B[2] + d1 * B[5] + d2 * B[8];
          if ((first >> (4 * s + js)) & 1) {
            dst[0] = v0;
            dst[1] = v1;
            dst[2] = v2;
          } else {
            dst[0] += v0;
            dst[1] += v1;
            dst[2] += v2;
          }
        }
      }
    }
  }
  return r;
}

// Evaluate every row of one problem at the free vector held one-variable-per-lane in `x`.
// Returns this lane's partial sum of squared residuals (the caller reduces over the group).
template <bool WITH_J>
__device__ __forceinline__ double eval_rows(const DevProgram* P, const Lds& S, int l, int W,
                                            double x, int xaddr, int buf, Prof* prof = nullptr) {
  wave_sync();
  if (l < P->n) S.pos[xaddr] = x;
  wave_sync();
  derived_update<WITH_J>(P, S, l, W, true);
  stamp(prof, 2);
  double ss = 0.0;
  for (int i = l; i < P->m; i += W) {
    const double r = row_pass<WITH_J>(P, S, i, buf);
    ss += r * r;
  }
  wave_sync();
  stamp(prof, 3);
  return ss;
}

// One problem per group of W threads (a wavefront, or two for a wide program): returns 0.5 * sum r^2 (uniform).
template <bool WITH_J, int W = kWave>
__device__ __forceinline__ double evaluate(const DevProgram* P, const Lds& S, int lane, double x,
                                           int xaddr, int buf, Prof* prof = nullptr) {
  return 0.5 * grp_sum<W>(S.red, eval_rows<WITH_J>(P, S, lane, W, x, xaddr, buf, prof));
}

// ------------------------------------------------------------------------------------
// normal equations:  A (strict upper) + dA (diag) = J^T J,   g = J^T r   (lane j owns g_j)
// ------------------------------------------------------------------------------------

__device__ __forceinline__ double build_normal(const DevProgram* P, const Lds& S, int l, int W,
                                               int buf, bool is_var) {
  const double* js = S.js;
  const double* rb = S.rb + buf * rb_buf_doubles(*P);
  const int kc = P->kc, kg = P->kg;
  // the triangle still holds the previous factor: clear it, then scatter the new J^T J entries
  for (int e = l; e < tri_doubles(*P); e += W) S.A[e] = 0.0;
  wave_sync();
  for (int w = l; w < P->n_work; w += W) {
    const uint4* terms = reinterpret_cast<const uint4*>(S.item_terms + (size_t)w * kc);
    double acc = 0.0;
#pragma unroll
    for (int c4 = 0; c4 < kItemTermsMax / 4; ++c4) {
      if (4 * c4 < kc) {  // uniform
        const uint4 t = terms[c4];
        acc += js[t.x & 0xffff] * js[t.x >> 16];
        acc += js[t.y & 0xffff] * js[t.y >> 16];
        acc += js[t.z & 0xffff] * js[t.z >> 16];
        acc += js[t.w & 0xffff] * js[t.w >> 16];
      }
    }
    const int dst = S.item_dst[w];
    if (dst < 0)
      S.dA[-dst - 1] = acc;
    else
      S.A[dst] = acc;
  }
  double g = 0.0;
  if (is_var) {
    const uint4* terms = reinterpret_cast<const uint4*>(S.grad_terms + (size_t)l * kg);
#pragma unroll
    for (int c4 = 0; c4 < kGradTermsMax / 4; ++c4) {
      if (4 * c4 < kg) {
        const uint4 t = terms[c4];
        g += js[t.x & 0xffff] * rb[t.x >> 16];
        g += js[t.y & 0xffff] * rb[t.y >> 16];
        g += js[t.z & 0xffff] * rb[t.z >> 16];
        g += js[t.w & 0xffff] * rb[t.w >> 16];
      }
    }
  }
  wave_sync();
  return g;
}

// Register-resident LDL^T (n <= N <= 63).  Lane i keeps row i of the
// matrix in N statically indexed registers; every cross-lane operand is a v_readlane
// broadcast, so the factorisation never waits on LDS.  Rows >= n are padded with identity.
// The unit-lower factor is also streamed to LDS (fire and forget) because the backward
// substitution needs COLUMN access, i.e. row k of L as seen by lane i < k.
// Returns false (uniformly) when a pivot is not positive; otherwise *dx = -(A + lambda I)^-1 g.
//
// Measured alternatives (profiles/r01/ldlt_variants.md): carrying the right-hand side as an
// extra row, a branch-free pivot test, a software-pipelined pivot reciprocal and broadcasting
// the pivot column through LDS were all slower in this kernel — each v_readlane operand needs an
// SGPR pair, and without the per-column exit branch below acting as a scheduling fence hipcc
// hoists later columns' broadcasts until the scalar file spills (v_writelane) inside the loop.
template <int N>
__device__ __forceinline__ bool ldlt_solve_reg(const DevProgram* P, const Lds& S, int lane,
                                               double lambda, double g, double* dx,
                                               double* pivot_min = nullptr, double* pivot_max = nullptr) {
  const int n = P->n;
  double a[N];
  const bool live = lane < n;
  const double diag = live ? S.dA[live ? lane : 0] + lambda : 1.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v = 0.0;
    if (live && j < lane) v = S.A[tri(lane, j)];
    if (j == lane) v = diag;
    a[j] = v;
  }
  double dinv = 0.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double pivot = wave_bcast(a[k], k);
    if (!(pivot > 0.0)) {
      ok = false;
      break;
    }
    if (pivot_min && k < n) {  // tangent solve only: conditioning record
      *pivot_min = fmin(*pivot_min, pivot);
      *pivot_max = fmax(*pivot_max, pivot);
    }
    const double rinv = fast_rcp(pivot);
    const double lk = a[k] * rinv;  // L[lane][k] for lanes > k
#pragma unroll
    for (int j = k + 1; j < N; ++j) a[j] -= lk * wave_bcast(a[k], j);
    if (lane == k) dinv = rinv;
    if (lane > k) {
      a[k] = lk;
      if (live) S.A[tri(lane, k)] = lk;
    }
  }
  if (!ok) return false;
  wave_sync();  // factor visible for the column reads of the backward substitution
  double b = live ? -g : 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {  // L y = -g (unit lower)
    const double yk = wave_bcast(b, k);
    if (lane > k) b -= a[k] * yk;
  }
  b *= dinv;  // D z = y
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {  // L^T dx = z; L[k][lane] (k > lane) read column-wise from LDS
    const double dk = wave_bcast(b, k);
    const double lkl = (k < n && lane < k) ? S.A[tri(k, lane < k ? lane : 0)] : 0.0;
    b -= lkl * dk;
  }
  *dx = b;
  return true;
}

// The same solve for programs of 64 ... 126 variables: W = 128 threads, thread i owns row i of the packed triangle in
// LDS (its diagonal entry in a register) and thread n the right-hand side, carried as row n of the matrix so that the
// forward substitution and the division by D happen inside the factorisation (row n ends as z = D^-1 L^-1 (-g)).
// Right-looking, two barriers per column: (a) the pivot is published, every row below scales its own column entry;
// (b) every row subtracts its multiple of the scaled column from the rest of itself.  The backward substitution
// publishes one finished unknown per barrier (two slots, alternating).  This is the capacity path - a T-bar axle
// with a heave link has 66 variables - not a tuned one: ~3 n barriers per step.
template <int W>
__device__ __forceinline__ bool ldlt_solve_wide(const DevProgram* P, const Lds& S, int tid, double lambda,
                                                double g, double* dx, double* pivot_min = nullptr,
                                                double* pivot_max = nullptr) {
  const int n = P->n;
  double diag = tid < n ? S.dA[tid] + lambda : 0.0;
  if (tid < n) S.A[tri(n, tid)] = -g;
  if (tid == 0) S.red[2] = diag;
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    __syncthreads();  // (a) pivot k and every update of column k are visible
    const double pivot = S.red[2 + (k & 1)];
    if (!(pivot > 0.0)) {
      ok = false;
      break;
    }
    if (pivot_min) {
      *pivot_min = fmin(*pivot_min, pivot);
      *pivot_max = fmax(*pivot_max, pivot);
    }
    const double rinv = 1.0 / pivot;
    const bool below = tid > k && tid <= n;
    double c = 0.0;
    if (below) {
      c = S.A[tri(tid, k)];
      S.A[tri(tid, k)] = c * rinv;  // L[tid][k]
    }
    __syncthreads();  // (b) the scaled column is visible
    if (below) {
      double* row = S.A + tri(tid, 0);
      for (int j = k + 1; j < tid; ++j) row[j] -= c * S.A[tri(j, k)];
      diag -= c * c * rinv;
      if (tid == k + 1) S.red[2 + ((k + 1) & 1)] = diag;
    }
  }
  __syncthreads();
  if (!ok) return false;
  double b = tid < n ? S.A[tri(n, tid)] : 0.0;  // z
  for (int k = n - 1; k >= 1; --k) {  // L^T dx = z, row k of L is contiguous: L[k][tid], tid < k
    if (tid == k) S.red[2 + (k & 1)] = b;
    __syncthreads();
    if (tid < k) b -= S.A[tri(k, tid)] * S.red[2 + (k & 1)];
  }
  __syncthreads();
  *dx = b;
  return true;
}

// The factorisation a kernel instantiation uses: register rows for one wavefront, LDS rows for two.
template <int NREG, int W>
__device__ __forceinline__ bool ldlt_solve(const DevProgram* P, const Lds& S, int lane, double lambda, double g,
                                           double* dx, double* pivot_min = nullptr, double* pivot_max = nullptr) {
  if constexpr (W > kWave) {
    return ldlt_solve_wide<W>(P, S, lane, lambda, g, dx, pivot_min, pivot_max);
  } else {
    return ldlt_solve_reg<NREG>(P, S, lane, lambda, g, dx, pivot_min, pivot_max);
  }
}

// Threads per problem of the instantiation for rows of NREG entries.
template <int NREG>
struct GroupWidth {
  static constexpr int value = NREG > kWave - 1 ? 2 * kWave : kWave;
};

// ------------------------------------------------------------------------------------
// problem setup helpers
// ------------------------------------------------------------------------------------

__device__ __forceinline__ void load_geometry(const DevProgram* P, const Lds& S, int l, int W,
                                              const double* gpos, const double* gparam,
                                              bool enable = true) {
  wave_sync();
  const double* src = gpos ? gpos : &P->design_pos[0][0];
  const double* par = gparam ? gparam : &P->row_param[0][0];
  const int nc = 8 * P->n_crows;
  if (enable) {
    for (int e = l; e < 3 * P->n_points; e += W) S.pos[e] = src[e];
    for (int e = l; e < nc; e += W) S.rowq[e] = par[e];
    for (int e = nc + l; e < 8 * P->m; e += W) S.rowq[e] = P->row_param[0][e];
  }
  wave_sync();
}

// max |r_i| in the reference's row definitions: the three pin components of a line are
// reported as the single softnorm point-on-line residual (constraints.py:560-576).
__device__ __forceinline__ double reference_abs_residual(const DevProgram* P, const Lds& S,
                                                         int i, int buf) {
  double r = S.rb[buf * rb_buf_doubles(*P) + i];
  const int* meta = S.rowmeta + i * kRowMetaStride;
  if (meta[0] == OKX_ROW_LINE_PIN) {
    const double* q = S.rowq + 8 * i;
    if ((int)q[6] != 0) return 0.0;
    V3 c = cross(sub(ld3(S.pos + 3 * meta[1]), ld3(q)), ld3(q + 3));
    r = softnorm(dot(c, c));
  }
  return fabs(r);
}

// ------------------------------------------------------------------------------------
// the solve kernel
// ------------------------------------------------------------------------------------

// PROFILE = true is a separate diagnostic instantiation: s_memtime stamps around the phases of
// block 0, summed into args.phase_cycles (never used by the product path or the bench).
#define OKX_STAMP(slot) stamp(prof, slot);

#ifndef OKX_WAVES_PER_SIMD
#define OKX_WAVES_PER_SIMD 2
#endif


// Register budget: rows of up to 24 entries fit 3 waves/SIMD (168 VGPRs); longer rows need the
// 256-register budget of 2 waves/SIMD (the LDS slice of such problems allows <= 5 waves/CU anyway).
template <int NREG, bool PROFILE>
__global__ void __launch_bounds__(GroupWidth<NREG>::value, NREG <= 24 ? OKX_WAVES_PER_SIMD : (NREG <= 48 ? 2 : 1)) okx_solve_kernel(const DevProgram* __restrict__ P,
                                                          SolveArgs args) {
  constexpr int W = GroupWidth<NREG>::value;  // threads per problem: one wavefront, two for NREG > 63
  Prof prof_store;
  Prof* prof = nullptr;
  if constexpr (PROFILE) {
    for (int k = 0; k < 12; ++k) prof_store.phase[k] = 0;
    prof_store.t = __builtin_readcyclecounter();
    prof = &prof_store;
  }
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const Lds S = carve(lds_base, P);
  const int n = P->n, m = P->m, T = P->n_targets;
  const int xaddr = lane < n ? 3 * P->free_point[lane / 3] + lane % 3 : 0;
  stage_program(P, S, lane, W);
  init_slice(P, S, lane, W);
  OKX_STAMP(0)

  const long long spg = args.steps_per_geometry;
  // chains of `chain_len` consecutive problems, never across a geometry boundary
  const long long span = spg > 0 ? spg : args.n_problems;           // problems per geometry
  const long long unit_len = args.chain_len;                        // 1 <= chain_len <= span
  const long long chains_per_span = (span + unit_len - 1) / unit_len;
  const long long n_units = (args.n_problems / span) * chains_per_span;
  long long loaded_geom = -1;

  for (long long unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    double x = 0.0, x_prev = 0.0;
    int hist = 0;  // consecutive solved predecessors in this chain (predictor needs two)
    double lambda_carry = 0.0;  // damping a converged chain step ended with (0: none)
    const long long span_idx = unit / chains_per_span;
    const long long first = span_idx * span + (unit % chains_per_span) * unit_len;
    const long long last = first + unit_len < (span_idx + 1) * span ? first + unit_len : (span_idx + 1) * span;
    for (long long b = first; b < last; ++b) {
      const long long step = b - first;
      const long long geom = spg > 0 ? b / spg : 0;
      if (geom != loaded_geom) {
        load_geometry(P, S, lane, W,
                      args.geom_pos ? args.geom_pos + geom * 3 * P->n_points : nullptr,
                      args.geom_row_param ? args.geom_row_param + geom * 8 * P->n_crows : nullptr);
        loaded_geom = geom;
        // a new geometry always restarts from its own design state
        if (lane < n) x = S.pos[xaddr];
      } else if (step == 0) {
        const double* src = args.geom_pos ? args.geom_pos + geom * 3 * P->n_points
                                          : &P->design_pos[0][0];
        if (lane < n) x = src[xaddr];
      }
      wave_sync();
      // Secant predictor inside a chain: with two solved predecessors, start from the linear
      // extrapolation x_{k-1} + alpha (x_{k-1} - x_{k-2}), alpha = projection of this step's
      // target increment on the previous one (1 for a uniform sweep, clamped to [0, 2]; a grid
      // wrap-around gives alpha <= 0 -> plain warm start).  The start point only changes the
      // iteration count, never the minimiser.
      double t_new = 0.0, t_old = 0.0, t_old2 = 0.0;
      if (lane < T) {
        t_new = args.targets[b * T + lane];
        t_old = step >= 1 ? args.targets[(b - 1) * T + lane] : t_new;
        t_old2 = step >= 2 ? args.targets[(b - 2) * T + lane] : t_old;
        S.tv[lane] = t_new;
      }
      if (hist >= 2) {
        const double num = grp_sum<W>(S.red, (t_new - t_old) * (t_old - t_old2));
        const double den = grp_sum<W>(S.red, (t_old - t_old2) * (t_old - t_old2));
        double alpha = den > 0.0 ? num / den : 0.0;
        alpha = fmin(fmax(alpha, 0.0), 2.0);
        const double xp = x + alpha * (x - x_prev);
        x_prev = x;
        x = xp;
      } else {
        x_prev = x;
      }
      wave_sync();
      OKX_STAMP(1)

      // Levenberg-Marquardt.  One evaluation site: `xt` is the point being evaluated,
      // `x` the last accepted point; the first pass accepts unconditionally.
      int cur = 1;  // buffer of the accepted point (first evaluation writes buffer 0)
      int nfev = 0, iters = 0, flags = 0;
      double F = 0.0, g = 0.0, dx = 0.0, lambda = 0.0, dmax = 0.0, nu = 2.0;
      double last_step = 0.0, step_len = 0.0, prev_step = 0.0;
      double piv_lo = 0.0, piv_hi = 0.0;  // pivot range of the last successful factorisation
      double xt = x;
      bool first = true;
      for (;;) {
        const double Ft = evaluate<true, W>(P, S, lane, xt, xaddr, cur ^ 1, prof);
        ++nfev;
        bool accept;
        bool stop = false;
        double rho = 1.0;
        if (first) {
          accept = true;
        } else {
          const double pred = 0.5 * grp_sum<W>(S.red, lane < n ? dx * (lambda * dx - g) : 0.0);
          const bool finite = Ft == Ft && step_len == step_len && Ft < 1e300;
          const bool small = finite && step_len <= 1e-8 && Ft <= F * (1.0 + 1e-6) + 1e-28;
          rho = (finite && pred > 0.0) ? (F - Ft) / pred : -1.0;
          accept = rho > 1e-4 || small;
          if (finite && step_len <= args.step_tol) {
            accept = small;
            flags |= OKX_INFO_CONVERGED;
            stop = true;
          } else if (accept && finite && F - Ft <= args.ftol * F && pred <= args.ftol * F) {
            // cost has stopped moving (MINPACK's ftol test): a compromise point of an
            // infeasible target, or the rounding floor of a feasible one
            flags |= OKX_INFO_CONVERGED;
            stop = true;
          }
        }
        if (accept) {
          x = xt;
          F = Ft;
          cur ^= 1;
          if (!first) last_step = step_len;
          if (!stop) {
            OKX_STAMP(4)
            g = build_normal(P, S, lane, W, cur, lane < n);
            OKX_STAMP(5)
            if (first) {
              dmax = grp_max<W>(S.red, lane < n ? S.dA[lane] : 0.0);
              lambda = args.lambda0 * dmax;
              // a warm-started chain step continues with the damping its predecessor ended with
              if (lambda_carry > 0.0) lambda = fmin(lambda, lambda_carry);
            } else if (rho > 1e-4) {
              // Nielsen's update; an accurate quadratic model (gain ratio > 0.9) drops the damping
              // by 10 (Marquardt), so the last steps are Gauss-Newton steps (MINPACK: par = 0)
              const double t = 2.0 * rho - 1.0;
              lambda *= rho > 0.9 ? 0.1 : fmax(1.0 / 3.0, 1.0 - t * t * t);
            }
            nu = 2.0;
            // gradient stop: > 0 the absolute form, < 0 MINPACK's scaled form max_j |(J^T r)_j| / (|J_j| |r|) (lmder's gnorm)
            double gmeasure = 0.0;
            if (args.grad_tol > 0.0) gmeasure = lane < n ? fabs(g) : 0.0;
            else if (args.grad_tol < 0.0) {
              const double cn = lane < n ? S.dA[lane] * 2.0 * F : 0.0;
              gmeasure = cn > 0.0 ? fabs(g) / sqrt(cn) : 0.0;
            }
            if (args.grad_tol != 0.0 && grp_max<W>(S.red, gmeasure) <= fabs(args.grad_tol)) {
              flags |= OKX_INFO_CONVERGED;
              stop = true;
            }
          }
        } else if (!stop) {
          lambda *= nu;
          nu *= 2.0;
        }
        first = false;
        if (stop) break;
        if (iters >= args.max_iter) break;
        if (!accept) {
          // Rejected trial (rare): the single Jacobian buffer now holds J(xt) and the packed
          // triangle holds the factor, so the accepted point's J and J^T J are rebuilt.
          evaluate<true, W>(P, S, lane, x, xaddr, cur, prof);
          ++nfev;
          g = build_normal(P, S, lane, W, cur, lane < n);
        }
        ++iters;
        OKX_STAMP(4)
        // damped normal equations; enlarge lambda until the factorisation succeeds
        bool ok = false;
        double pmin = 1e300, pmax = 0.0;
        for (int tries = 0; tries < 60; ++tries) {
          if (!(lambda < 1e30)) break;
          pmin = 1e300;
          ok = ldlt_solve<NREG, W>(P, S, lane, lambda, g, &dx, &pmin, &pmax);
          stamp(prof, 6);
          if (ok) break;
          lambda = fmax(lambda * 10.0, 1e-12 * dmax);
          build_normal(P, S, lane, W, cur, lane < n);  // the failed factor overwrote J^T J
        }
        if (!ok) {
          flags |= OKX_INFO_FAILED;
          break;
        }
        piv_lo = pmin - lambda;  // what the damping did not put there
        piv_hi = pmax;
        OKX_STAMP(7)
        step_len = grp_max<W>(S.red, lane < n ? fabs(dx) : 0.0);
        if (step_len <= args.step_tol) {
          // the Newton-type correction is already below tolerance: x is the answer and the
          // residuals in hand belong to it (no confirming evaluation of x + dx)
          flags |= OKX_INFO_CONVERGED;
          last_step = step_len;
          break;
        }
        xt = x + dx;
        // Predicted next correction (rho |dx| + C |dx|^2: damping contraction lambda / min pivot and
        // the observed quadratic contraction, both x 100; profiles/r02/DESIGN_r02.md §5.1 "Ending a solve"): when it is
        // within step_tol the step is applied and confirmed by a residual-only evaluation instead of
        // a full Jacobian / factorisation pass.
        if (!args.confirm) {
          const double cq = prev_step > 0.0 ? fmax(100.0 * step_len / (prev_step * prev_step), 1e-3) : 1.0;
          const double rho_lin = 100.0 * lambda / pmin;
          if (step_len <= 1e-3 && (rho_lin + cq * step_len) * step_len <= args.step_tol) {
            const double Fl = evaluate<false, W>(P, S, lane, xt, xaddr, cur ^ 1, prof);
            ++nfev;
            if (Fl == Fl && Fl <= F * (1.0 + 1e-6) + 1e-28) {
              x = xt;
              F = Fl;
              cur ^= 1;
              last_step = step_len;
              flags |= OKX_INFO_CONVERGED;
              break;
            }
            // the cost rose: the same point goes through a full pass (its Jacobian is needed anyway)
          }
        }
        prev_step = step_len;
      }
      OKX_STAMP(4)

      // final state: free points, then every derived point (incl. output-only ones)
      wave_sync();
      if (lane < n) S.pos[xaddr] = x;
      wave_sync();
      derived_update<false>(P, S, lane, W, false);
      double ra = 0.0;
      for (int i = lane; i < m; i += W) ra = fmax(ra, reference_abs_residual(P, S, i, cur));
      const double max_res = grp_max<W>(S.red, ra);
      if (max_res > args.residual_tolerance) flags |= OKX_INFO_RESIDUAL_EXCEEDED;
      if (piv_hi > 0.0 && piv_lo <= OKX_ILL_CONDITIONED_PIVOT_RATIO * piv_hi) flags |= OKX_INFO_ILL_CONDITIONED;
      double* out = args.out_pos + b * 3 * P->n_out;
      for (int e = lane; e < 3 * P->n_out; e += W) out[e] = S.pos[3 * P->out_point[e / 3] + e % 3];
      if (lane == 0) {
        okx_info inf;
        inf.max_residual = max_res;
        inf.cost = F;
        inf.last_step = last_step;
        inf.iterations = iters;
        inf.nfev = nfev;
        inf.flags = flags;
        inf.reserved = 0;
        args.info[b] = inf;
      }
      // a chain never continues from a state that failed to converge
      if (!(flags & OKX_INFO_CONVERGED) || (flags & OKX_INFO_FAILED)) {
        const double* src = args.geom_pos ? args.geom_pos + geom * 3 * P->n_points
                                          : &P->design_pos[0][0];
        if (lane < n) x = src[xaddr];
        hist = 0;
        lambda_carry = 0.0;
      } else {
        if (hist < 2) ++hist;
        lambda_carry = lambda;
      }
      OKX_STAMP(8)
    }
  }
  if constexpr (PROFILE) {
    if (blockIdx.x == 0 && lane == 0 && args.phase_cycles)
      for (int k = 0; k < 12; ++k) args.phase_cycles[k] = prof_store.phase[k];
  }
}

// ------------------------------------------------------------------------------------
// residual / dense Jacobian evaluation (parity rung R1) and normal equations (debug)
// ------------------------------------------------------------------------------------

struct EvalArgs {
  const double* x;        // [B][n]
  const double* targets;  // [B][T]
  double* r;              // [B][m]
  double* jac;            // [B][m][n] or null
  double* ata;            // [B][n][n] or null  (J^T J, full symmetric)
  double* atr;            // [B][n] or null     (J^T r)
  long long n_problems;
};

template <int W>
__global__ void __launch_bounds__(W) okx_eval_kernel(const DevProgram* __restrict__ P, EvalArgs args) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const Lds S = carve(lds_base, P);
  const int n = P->n, m = P->m, T = P->n_targets;
  const int xaddr = lane < n ? 3 * P->free_point[lane / 3] + lane % 3 : 0;
  stage_program(P, S, lane, W);
  init_slice(P, S, lane, W);
  load_geometry(P, S, lane, W, nullptr, nullptr);
  for (long long b = blockIdx.x; b < args.n_problems; b += gridDim.x) {
    wave_sync();
    if (lane < T) S.tv[lane] = args.targets[b * T + lane];
    const double x = lane < n ? args.x[b * n + lane] : 0.0;
    evaluate<true, W>(P, S, lane, x, xaddr, 0);
    for (int i = lane; i < m; i += W) args.r[b * m + i] = S.rb[i];
    if (args.jac) {
      double* J = args.jac + b * (long long)m * n;
      for (int e = lane; e < m * n; e += W) J[e] = 0.0;
      wave_sync();
      for (int i = lane; i < m; i += W) {
        const double* jr = S.js + (size_t)i * P->js_stride;
        for (int s = 0; s < P->row_nblk[i]; ++s)
          for (int k = 0; k < 3; ++k) J[(long long)i * n + 3 * P->row_blk[i][s] + k] = jr[3 * s + k];
      }
    }
    if (args.ata || args.atr) {
      const double g = build_normal(P, S, lane, W, 0, lane < n);
      if (args.atr && lane < n) args.atr[b * n + lane] = g;
      if (args.ata) {
        double* M = args.ata + b * (long long)n * n;
        for (int e = lane; e < n * n; e += W) {
          const int i = e / n, j = e % n;
          M[e] = i == j ? S.dA[i] : (i < j ? S.A[tri(j, i)] : S.A[tri(i, j)]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// solution-manifold tangents (reference sensitivity.py:57-143), generic interpreter form
// ------------------------------------------------------------------------------------

struct TangentArgs {
  const double* pos;             // [B][n_out][3] solved positions (okx_solve_batch output)
  const double* geom_pos;        // [G][P][3] or null
  const double* geom_row_param;  // [G][Mc][8] or null
  double* tan;                   // [B][T][n_out][3]
  okx_tangent_info* tinfo;       // [B]
  long long n_problems;
  long long steps_per_geometry;
  int free_out[kMaxFree];        // index of every free point inside the output point list
};

// Velocity of one derived point from the velocities of its inputs (forward mode, closed form;
// the reference propagates dual numbers: sensitivity.py:127-131).  Uniform over lanes.
__device__ __forceinline__ V3 dop_velocity(int type, const int* pts, double c, const double* pos,
                                           const double* vel) {
  if (type == OKX_DOP_MIDPOINT) {
    V3 a = ld3(vel + 3 * pts[0]), b = ld3(vel + 3 * pts[1]);
    return {a.x + (b.x - a.x) / 2, a.y + (b.y - a.y) / 2, a.z + (b.z - a.z) / 2};
  }
  if (type == OKX_DOP_ALONG) {  // out = base + c u,  u = w / |w|,  w = p1 - p2
    V3 vb = ld3(vel + 3 * pts[0]);
    V3 w = sub(ld3(pos + 3 * pts[1]), ld3(pos + 3 * pts[2]));
    V3 dw = sub(ld3(vel + 3 * pts[1]), ld3(vel + 3 * pts[2]));
    double nrm, inrm;
    fast_sqrt_rsqrt(dot(w, w), &nrm, &inrm);
    V3 u = {w.x * inrm, w.y * inrm, w.z * inrm};
    const double ud = dot(u, dw), k = c * inrm;
    return {vb.x + k * (dw.x - u.x * ud), vb.y + k * (dw.y - u.y * ud), vb.z + k * (dw.z - u.z * ud)};
  }
  // contact patch: out = wc + R wu, wu = wd / |wd|, wd = a_z a - e_z, a = (axo - axi) / |axo - axi|
  V3 vw = ld3(vel + 3 * pts[0]);
  V3 v = sub(ld3(pos + 3 * pts[2]), ld3(pos + 3 * pts[1]));
  V3 dv = sub(ld3(vel + 3 * pts[2]), ld3(vel + 3 * pts[1]));
  double vn, ivn;
  fast_sqrt_rsqrt(dot(v, v), &vn, &ivn);
  V3 a = {v.x * ivn, v.y * ivn, v.z * ivn};
  const double adv = dot(a, dv);
  V3 da = {ivn * (dv.x - a.x * adv), ivn * (dv.y - a.y * adv), ivn * (dv.z - a.z * adv)};
  V3 wd = {a.z * a.x, a.z * a.y, a.z * a.z - 1.0};
  V3 dwd = {da.z * a.x + a.z * da.x, da.z * a.y + a.z * da.y, da.z * a.z + a.z * da.z};
  double wn, iwn;
  fast_sqrt_rsqrt(dot(wd, wd), &wn, &iwn);
  V3 wu = {wd.x * iwn, wd.y * iwn, wd.z * iwn};
  const double wdw = dot(wu, dwd), k = c * iwn;
  return {vw.x + k * (dwd.x - wu.x * wdw), vw.y + k * (dwd.y - wu.y * wdw), vw.z + k * (dwd.z - wu.z * wdw)};
}

// One wavefront per solved state: rows + J^T J at the state, undamped LDL^T, one solve per target
// with right-hand side J^T e_t (the target row's own Jacobian entries), derived-point velocities.
// Dynamic LDS = the solve kernel's slice + a [P][3] velocity table.
template <int NREG>
__global__ void __launch_bounds__(GroupWidth<NREG>::value, NREG <= 24 ? OKX_WAVES_PER_SIMD : (NREG <= 48 ? 2 : 1))
okx_tangent_kernel(const DevProgram* __restrict__ P, TangentArgs args) {
  constexpr int W = GroupWidth<NREG>::value;
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const Lds S = carve(lds_base, P);
  double* vel = lds_base + P->lds_doubles;
  const int n = P->n, T = P->n_targets;
  const int xaddr = lane < n ? 3 * P->free_point[lane / 3] + lane % 3 : 0;
  stage_program(P, S, lane, W);
  init_slice(P, S, lane, W);
  long long loaded_geom = -1;
  for (long long b = blockIdx.x; b < args.n_problems; b += gridDim.x) {
    const long long geom = args.steps_per_geometry > 0 ? b / args.steps_per_geometry : 0;
    if (geom != loaded_geom) {
      load_geometry(P, S, lane, W, args.geom_pos ? args.geom_pos + geom * 3 * P->n_points : nullptr,
                    args.geom_row_param ? args.geom_row_param + geom * 8 * P->n_crows : nullptr);
      loaded_geom = geom;
    }
    wave_sync();
    if (lane < T) S.tv[lane] = 0.0;  // target values do not enter the Jacobian
    const double x = lane < n ? args.pos[(b * P->n_out + args.free_out[lane / 3]) * 3 + lane % 3] : 0.0;
    evaluate<true, W>(P, S, lane, x, xaddr, 0);
    derived_update<false>(P, S, lane, W, false);  // every derived point (inputs of the velocity pass)
    double pmin = 1e300, pmax = 0.0;
    bool all_ok = true;
    for (int t = 0; t < T; ++t) {
      build_normal(P, S, lane, W, 0, lane < n);  // the previous factorisation overwrote J^T J
      // (J^T e_t)_j = J[target row t][j]
      const int row = P->n_crows + t;
      double rhs = 0.0;
      if (lane < n) {
        const int blk = lane / 3;
        for (int s = 0; s < P->row_nblk[row]; ++s)
          if (P->row_blk[row][s] == blk) rhs = S.js[(size_t)row * P->js_stride + 3 * s + lane % 3];
      }
      double q = 0.0;
      const bool ok = ldlt_solve<NREG, W>(P, S, lane, 0.0, -rhs, &q, &pmin, &pmax);
      all_ok = all_ok && ok;
      wave_sync();
      for (int e = lane; e < 3 * P->n_points; e += W) vel[e] = 0.0;
      wave_sync();
      if (lane < n) vel[xaddr] = q;
      wave_sync();
      for (int e = 0; e < P->n_derived; ++e) {
        V3 v = dop_velocity(P->dop_type[e], P->dop_pts[e], P->dop_param[e], S.pos, vel);
        wave_sync();
        if (lane < 3) vel[3 * P->dop_out[e] + lane] = sel3(lane, v.x, v.y, v.z);
        wave_sync();
      }
      double* out = args.tan + ((b * T + t) * P->n_out) * 3;
      for (int e = lane; e < 3 * P->n_out; e += W)
        out[e] = ok ? vel[3 * P->out_point[e / 3] + e % 3] : __builtin_nan("");
    }
    if (lane == 0) {
      okx_tangent_info ti;
      ti.min_pivot = pmin;
      ti.max_pivot = pmax;
      ti.flags = (all_ok ? OKX_TANGENT_OK : 0) |
                 ((!all_ok || pmin <= n * 2.220446049250313e-16 * pmax) ? OKX_TANGENT_RANK_DEFICIENT : 0);
      ti.reserved = 0;
      args.tinfo[b] = ti;
    }
  }
}

// ------------------------------------------------------------------------------------
// per-geometry problem emission (okx_rebind_design)
// ------------------------------------------------------------------------------------

// Full output positions from free-point coordinates: fixed points from the (per-geometry) design table, derived
// points re-evaluated in program order.  One thread per problem — the receiving side of the multi-GPU exchange,
// which ships the 3 n_free free coordinates of a solve instead of its 3 n_out output coordinates.  The interpreter
// form (programs without a generated okx_quad_expand: composed axles).  A point is read where it is used - a fixed one
// from the table (wave-uniform for the program's own geometry), a free one from the thread's input row, a derived one
// from LDS ([component][thread]: conflict-free) - so that no thread carries a private table of every position
// (round 3's form: 2.3 KB of scratch per thread, 0.9 TB/s on axle states).
struct ExpandArgs {
  const double* free;      // [B][n_free][3]
  const double* geom_pos;  // [G][P][3] or null (program's own geometry)
  double* out_pos;         // [B][n_out][3]
  long long n_problems, steps_per_geometry;
};

constexpr int kExpandThreads = 64;

__global__ void __launch_bounds__(kExpandThreads) okx_expand_kernel(const DevProgram* __restrict__ P, ExpandArgs a) {
  extern __shared__ double okx_expand_lds[];  // [3 n_derived][64] derived positions, then the point -> source table
  const int lane = threadIdx.x;
  const int np = P->n_points, nd = P->n_derived;
  double* const dv = okx_expand_lds;
  int* const source = reinterpret_cast<int*>(okx_expand_lds + 3 * nd * kExpandThreads);  // -1 fixed, k free point k, 1000 + e derived op e
  for (int p = lane; p < np; p += kExpandThreads) source[p] = -1;
  __syncthreads();
  for (int k = lane; k < P->n_free; k += kExpandThreads) source[P->free_point[k]] = k;
  for (int e = lane; e < nd; e += kExpandThreads) source[P->dop_out[e]] = 1000 + e;
  __syncthreads();
  long long b = (long long)blockIdx.x * kExpandThreads + lane;
  const bool valid = b < a.n_problems;
  if (!valid) b = a.n_problems - 1;
  const double* base = a.geom_pos ? a.geom_pos + (b / a.steps_per_geometry) * 3 * np : &P->design_pos[0][0];
  const double* x = a.free + b * 3 * P->n_free;
  auto at = [&](int pt) -> V3 {
    const int s = source[pt];  // (wave-uniform)
    if (s < 0) return ld3(base + 3 * pt);
    if (s < 1000) return ld3(x + 3 * s);
    const double* d = dv + 3 * (s - 1000) * kExpandThreads + lane;
    return {d[0], d[kExpandThreads], d[2 * kExpandThreads]};
  };
  for (int e = 0; e < nd; ++e) {
    V3 u, av;
    double nrm, vn, ga;
    const V3 out = dop_position_at(P->dop_type[e], P->dop_pts[e], P->dop_param[e], at, &u, &nrm, &av, &vn, &ga);
    double* d = dv + 3 * e * kExpandThreads + lane;
    d[0] = out.x;
    d[kExpandThreads] = out.y;
    d[2 * kExpandThreads] = out.z;
  }
  if (!valid) return;
  double* o = a.out_pos + b * 3 * P->n_out;
  for (int k = 0; k < P->n_out; ++k) {
    const V3 v = at(P->out_point[k]);
    o[3 * k] = v.x;
    o[3 * k + 1] = v.y;
    o[3 * k + 2] = v.z;
  }
}

struct RebindArgs {
  const double* hardpoints;  // [G][P][3]
  double* geom_pos;          // [G][P][3]
  double* geom_row_param;    // [G][Mc][8]
  long long n_geometries;
};

__global__ void __launch_bounds__(kWave) okx_rebind_kernel(const DevProgram* __restrict__ P,
                                                           RebindArgs args) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const Lds S = carve(lds_base, P);
  for (long long gidx = blockIdx.x; gidx < args.n_geometries; gidx += gridDim.x) {
    load_geometry(P, S, lane, kWave, args.hardpoints + gidx * 3 * P->n_points, nullptr);
    derived_update<false>(P, S, lane, kWave, false);
    double* gp = args.geom_pos + gidx * 3 * P->n_points;
    for (int e = lane; e < 3 * P->n_points; e += kWave) gp[e] = S.pos[e];
    for (int i = lane; i < P->n_crows; i += kWave) {
      double q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = S.rowq[8 * i + k];
      const int* pts = P->row_pts[i];
      const int type = P->row_type[i];
      if (type == OKX_ROW_DISTANCE) {  // geometric.py:17-28
        V3 d = sub(ld3(S.pos + 3 * pts[1]), ld3(S.pos + 3 * pts[0]));
        q[0] = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
      } else if (type == OKX_ROW_ANGLE || type == OKX_ROW_THREE_POINT_ANGLE) {
        // geometric.py:71-104 compute_vector_vector_angle
        V3 v1, v2;
        if (type == OKX_ROW_ANGLE) {
          v1 = sub(ld3(S.pos + 3 * pts[1]), ld3(S.pos + 3 * pts[0]));
          v2 = sub(ld3(S.pos + 3 * pts[3]), ld3(S.pos + 3 * pts[2]));
        } else {
          v1 = sub(ld3(S.pos + 3 * pts[0]), ld3(S.pos + 3 * pts[1]));
          v2 = sub(ld3(S.pos + 3 * pts[2]), ld3(S.pos + 3 * pts[1]));
        }
        double n1 = sqrt(dot(v1, v1)), n2 = sqrt(dot(v2, v2));
        V3 u1 = {v1.x / n1, v1.y / n1, v1.z / n1}, u2 = {v2.x / n2, v2.y / n2, v2.z / n2};
        V3 c = cross(u1, u2);
        q[0] = atan2(sqrt(dot(c, c)), dot(u1, u2));
      } else if (type == OKX_ROW_SCALAR_TRIPLE) {  // attachments.py:45-74
        V3 p1 = ld3(S.pos + 3 * pts[0]);
        V3 v1 = sub(ld3(S.pos + 3 * pts[1]), p1), v2 = sub(ld3(S.pos + 3 * pts[2]), p1),
           v3 = sub(ld3(S.pos + 3 * pts[3]), p1);
        q[0] = dot(v1, cross(v2, v3));
        q[1] = fabs(q[0]);
      } else if (type == OKX_ROW_POINT_ON_LINE || type == OKX_ROW_LINE_PIN) {  // track_rod.py:92-96
        q[0] = S.pos[3 * pts[0]], q[1] = S.pos[3 * pts[0] + 1], q[2] = S.pos[3 * pts[0] + 2];
      }
      double* dst = args.geom_row_param + (gidx * P->n_crows + i) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) dst[k] = q[k];
    }
  }
}

}  // namespace okx
