// okx_packed.hip — lane-group packed solve kernel for small systems.
//
// A double-wishbone corner has 18 unknowns and 21 residual rows: one problem keeps at most
// 21 of a wavefront's 64 lanes busy and the one-problem-per-wavefront kernel
// (okx_kernels.hip) is instruction-issue bound at that utilisation (profiles/r01).  Here a
// wavefront hosts G = floor(64 / W) INDEPENDENT problems side by side, W = max(m, n + 1)
// lanes each (3 for the DW corner, 4 for n <= 15 variants, 2 up to W = 32).  Each group has its
// own LDS slice and its own Levenberg-Marquardt state; the groups run the same instruction
// stream in lockstep (a finished group idles until its neighbours are done).
//
// Cross-lane traffic is group-local, so the v_readlane broadcasts of the single-problem
// kernel do not apply.  The factorisation instead publishes each pivot column through LDS
// (one ds_write per column, broadcast reads back), and carries the right-hand side as row n
// of the matrix so that the forward substitution and the D^-1 scaling fall out of the
// trailing updates; only the backward substitution broadcasts one unknown per step
// (G v_readlane pairs + a select).  Group-wide sums/maxima are accumulated from an LDS
// scratch line in a fixed order, so every lane of a group sees bit-identical values and
// takes identical accept/reject decisions.
#pragma once

#include "okx_kernels.hip"

namespace okx {

// Extra per-group LDS behind the single-problem layout: column buffer + rhs/z buffer.
__host__ __device__ inline int packed_group_doubles(const DevProgram& P) {
  int s = 0;
  s += P.n_points * 3;
  s += P.m * 8;
  s += (P.n_active > 0 ? P.n_active : 1) * kDepMax * 9;
  s += js_buf_doubles(P);
  s += 2 * rb_buf_doubles(P);
  s += tri_doubles(P);
  s += P.n;
  s += kMaxTargets;
  s += 2 * kColBuf;
  return (s + 1) & ~1;
}

__host__ __device__ inline int packed_lds_doubles(const DevProgram& P, int groups) {
  return groups * packed_group_doubles(P) + 2 * kWave + lds_table_doubles(P) + 2;
}

struct PackedLds {
  Lds S;           // group slice (per-lane pointers) + shared tables
  double* scratch; // [2][64] wave-wide reduction scratch
};

__device__ __forceinline__ PackedLds carve_packed(double* base, const DevProgram* P, int g,
                                                  int groups) {
  PackedLds L;
  double* p = base + (size_t)g * packed_group_doubles(*P);
  L.S.pos = p;
  p += P->n_points * 3;
  L.S.rowq = p;
  p += P->m * 8;
  L.S.dblk = p;
  p += (P->n_active > 0 ? P->n_active : 1) * kDepMax * 9;
  L.S.js = p;
  p += js_buf_doubles(*P);
  L.S.rb = p;
  p += 2 * rb_buf_doubles(*P);
  L.S.A = p;
  p += tri_doubles(*P);
  L.S.dA = p;
  p += P->n;
  L.S.tv = p;
  p += kMaxTargets;
  L.S.col = p;
  p += kColBuf;
  L.S.zbuf = p;
  double* shared = base + (size_t)groups * packed_group_doubles(*P);
  L.scratch = shared;
  carve_tables(reinterpret_cast<int*>(shared + 2 * kWave), P, &L.S);
  return L;
}

// Sum / max of `v` over the lanes of each group; every lane of a group gets the same bits.
__device__ __forceinline__ double group_sum(double v, double* scratch, int lane, int gbase, int W) {
  scratch[lane] = v;
  wave_sync();
  double s = 0.0;
  for (int j = 0; j < W; ++j) s += scratch[gbase + j];
  wave_sync();
  return s;
}

__device__ __forceinline__ double group_max(double v, double* scratch, int lane, int gbase, int W) {
  scratch[lane] = v;
  wave_sync();
  double s = 0.0;
  for (int j = 0; j < W; ++j) s = fmax(s, scratch[gbase + j]);
  wave_sync();
  return s;
}

// Two sums in one LDS round trip.
__device__ __forceinline__ void group_sum2(double v0, double v1, double* scratch, int lane,
                                           int gbase, int W, double* s0, double* s1) {
  scratch[lane] = v0;
  scratch[kWave + lane] = v1;
  wave_sync();
  double a = 0.0, b = 0.0;
  for (int j = 0; j < W; ++j) {
    a += scratch[gbase + j];
    b += scratch[kWave + gbase + j];
  }
  wave_sync();
  *s0 = a;
  *s1 = b;
}

// Value of local lane k of every lane's own group (k wave-uniform).
template <int G>
__device__ __forceinline__ double group_bcast(double v, int k, int W, int g) {
  double r = wave_bcast(v, k);
  if (G > 1) {
    const double r1 = wave_bcast(v, W + k);
    r = g == 1 ? r1 : r;
  }
  if (G > 2) {
    const double r2 = wave_bcast(v, 2 * W + k);
    r = g == 2 ? r2 : r;
  }
  if (G > 3) {
    const double r3 = wave_bcast(v, 3 * W + k);
    r = g == 3 ? r3 : r;
  }
  return r;
}

// LDL^T with the pivot column broadcast through LDS.  Lane i keeps row i in N statically
// indexed registers; per column every lane publishes its entry with one ds_write and reads the
// column back as broadcast ds_reads, so all operands stay in VGPRs (the v_readlane variant
// below needs an SGPR pair per operand and spills scalars inside the loop).  Lanes n..N-1 are
// identity padding rows, lane N carries the right-hand side -g, so the forward substitution and
// the D^-1 scaling fall out of the trailing updates.  Branch-free, statically indexed.
// BCAST(v, k): value of (group-local) lane k.
template <int N, typename Bcast>
__device__ __forceinline__ bool ldlt_solve_lds(const DevProgram* P, const Lds& S, int l,
                                               double lambda, double grad, double* dx, Bcast bcast,
                                               Prof* prof = nullptr) {
  const int n = P->n;
  const bool is_row = l < n, is_pad = l >= n && l < N, is_rhs = l == N;
  if (l < N) S.zbuf[l] = is_row ? -grad : 0.0;
  wave_sync();
  double a[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v = 0.0;
    if (is_row && j < l) v = S.A[tri(l, j)];
    if (is_row && j == l) v = S.dA[l] + lambda;
    if (is_pad && j == l) v = 1.0;
    if (is_rhs) v = S.zbuf[j];
    a[j] = v;
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (l >= k && l <= N) S.col[l] = a[k];
    wave_sync();
    const double pivot = S.col[k];
    const bool good = pivot > 0.0;
    ok = ok && good;
    const double rinv = fast_rcp(good ? pivot : 1.0);
    const double lk = a[k] * rinv;  // L[l][k] for rows > k; z_k for the rhs row
#pragma unroll
    for (int j = k + 1; j < N; ++j) a[j] -= lk * S.col[j];
    if (l > k) {
      a[k] = lk;
      if (is_row) S.A[tri(l, k)] = lk;
    }
    wave_sync();  // the column buffer is reused by the next column
  }
  stamp(prof, 6);
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (is_rhs) S.zbuf[k] = a[k];
  wave_sync();
  double b = is_row ? S.zbuf[l] : 0.0;
  double lt[N];  // lt[k] = L[k][l] for k > l (row k of the factor, read column-wise)
#pragma unroll
  for (int k = 0; k < N; ++k) lt[k] = (is_row && k < n && l < k) ? S.A[tri(k, l)] : 0.0;
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {  // L^T dx = z
    const double dk = bcast(b, k);
    b -= lt[k] * dk;
  }
  *dx = b;
  stamp(prof, 7);
  return ok;
}

template <int N, int G, bool PROFILE>
__global__ void __launch_bounds__(kWave, 2) okx_solve_packed_kernel(const DevProgram* __restrict__ P,
                                                                 SolveArgs args, int W) {
  Prof prof_store;
  Prof* prof = nullptr;
  if constexpr (PROFILE) {
    for (int k = 0; k < 12; ++k) prof_store.phase[k] = 0;
    prof_store.t = __builtin_readcyclecounter();
    prof = &prof_store;
  }
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const bool valid = lane < G * W;
  const int g = valid ? lane / W : G - 1;
  const int gbase = g * W;
  const int l = valid ? lane - gbase : (1 << 20);  // idle lanes fall out of every l-bounded loop
  const PackedLds L = carve_packed(lds_base, P, g, G);
  const Lds& S = L.S;
  const int n = P->n, m = P->m, T = P->n_targets;
  const bool is_var = l < n;
  const int xaddr = is_var ? 3 * P->free_point[l / 3] + l % 3 : 0;

  // shared tables (all lanes cooperate), then every group initialises its own slice
  stage_program(P, S, lane);
  init_slice(P, S, l, W);
  OKX_STAMP(0)

  const long long spg = args.steps_per_geometry;
  // chains of `chain_len` consecutive problems, never across a geometry boundary
  const long long span = spg > 0 ? spg : args.n_problems;           // problems per geometry
  const long long unit_len = args.chain_len;                        // 1 <= chain_len <= span
  const long long chains_per_span = (span + unit_len - 1) / unit_len;
  const long long n_units = (args.n_problems / span) * chains_per_span;
  long long loaded_geom = -1;

  for (long long unit0 = (long long)blockIdx.x * G; unit0 < n_units;
       unit0 += (long long)gridDim.x * G) {
    const long long unit = unit0 + g;
    const bool unit_ok = valid && unit < n_units;
    double x = 0.0, x_prev = 0.0;
    int hist = 0;  // consecutive solved predecessors of this group's chain
    const long long span_idx = unit_ok ? unit / chains_per_span : 0;
    const long long first = span_idx * span + (unit_ok ? unit % chains_per_span : 0) * unit_len;
    const long long last = first + unit_len < (span_idx + 1) * span ? first + unit_len : (span_idx + 1) * span;
    for (long long step = 0; step < unit_len; ++step) {
      const bool has_unit = unit_ok && first + step < last;
      const long long b = has_unit ? first + step : 0;
      const long long geom = spg > 0 ? b / spg : 0;
      const bool reload = has_unit && geom != loaded_geom;
      const double* gp = args.geom_pos ? args.geom_pos + geom * 3 * P->n_points : nullptr;
      const double* gq = args.geom_row_param ? args.geom_row_param + geom * 8 * P->n_crows : nullptr;
      if (__any(reload)) load_geometry(P, S, l, W, gp, gq, reload);
      if (reload) {
        loaded_geom = geom;
        if (is_var) x = S.pos[xaddr];  // a new geometry restarts from its own design state
      } else if (has_unit && step == 0) {
        const double* src = gp ? gp : &P->design_pos[0][0];
        if (is_var) x = src[xaddr];
      }
      wave_sync();
      // secant predictor per group (see okx_solve_kernel): extrapolate from two solved predecessors
      double t_new = 0.0, t_old = 0.0, t_old2 = 0.0;
      if (has_unit && l < T) {
        t_new = args.targets[b * T + l];
        t_old = step >= 1 ? args.targets[(b - 1) * T + l] : t_new;
        t_old2 = step >= 2 ? args.targets[(b - 2) * T + l] : t_old;
        S.tv[l] = t_new;
      }
      {
        double num, den;
        group_sum2((t_new - t_old) * (t_old - t_old2), (t_old - t_old2) * (t_old - t_old2), L.scratch,
                   lane, gbase, W, &num, &den);
        if (hist >= 2) {
          double alpha = den > 0.0 ? num / den : 0.0;
          alpha = fmin(fmax(alpha, 0.0), 2.0);
          const double xp = x + alpha * (x - x_prev);
          x_prev = x;
          x = xp;
        } else {
          x_prev = x;
        }
      }
      wave_sync();
      OKX_STAMP(1)

      int cur = 1, nfev = 0, iters = 0, flags = 0;
      double F = 0.0, grad = 0.0, dx = 0.0, lambda = 0.0, dmax = 0.0, nu = 2.0;
      double last_step = 0.0, step_len = 0.0;
      double xt = x;
      bool first = true;
      bool done = !has_unit;
      for (;;) {
        const double ss = eval_rows<true>(P, S, l, W, xt, xaddr, cur ^ 1, prof);
        double Ft, pred;
        group_sum2(valid ? ss : 0.0, is_var ? dx * (lambda * dx - grad) : 0.0, L.scratch, lane, gbase,
                   W, &Ft, &pred);
        Ft *= 0.5;
        pred *= 0.5;
        if (!done) ++nfev;
        bool accept, stop = false;
        double rho = 1.0;
        if (first) {
          accept = true;
        } else {
          const bool finite = Ft == Ft && step_len == step_len && Ft < 1e300;
          const bool small = finite && step_len <= 1e-8 && Ft <= F * (1.0 + 1e-6) + 1e-28;
          rho = (finite && pred > 0.0) ? (F - Ft) / pred : -1.0;
          accept = rho > 1e-4 || small;
          if (finite && step_len <= args.step_tol) {
            accept = small;
            if (!done) flags |= OKX_INFO_CONVERGED;
            stop = true;
          } else if (accept && finite && F - Ft <= args.ftol * F && pred <= args.ftol * F) {
            if (!done) flags |= OKX_INFO_CONVERGED;
            stop = true;
          }
        }
        accept = accept && !done;
        if (accept) {
          x = xt;
          F = Ft;
          cur ^= 1;
          if (!first) last_step = step_len;
        }
        const bool rebuild = accept && !stop;
        OKX_STAMP(4)
        if (__any(rebuild)) {
          // groups that did not accept keep their matrix: evaluate into a dummy predicate
          const double gnew = build_normal(P, S, rebuild ? l : (1 << 20), W, cur, rebuild && is_var);
          if (rebuild) grad = gnew;
        }
        OKX_STAMP(5)
        if (rebuild) {
          nu = 2.0;
          if (!first && rho > 1e-4) {
            const double t = 2.0 * rho - 1.0;
            lambda *= fmax(1.0 / 3.0, 1.0 - t * t * t);
          }
        }
        if (first) {
          dmax = group_max(is_var ? S.dA[is_var ? l : 0] : 0.0, L.scratch, lane, gbase, W);
          lambda = args.lambda0 * dmax;
        }
        if (!accept && !stop && !done && !first) {
          lambda *= nu;
          nu *= 2.0;
        }
        const bool rejected = !accept && !stop && !done && !first;
        first = false;
        if (stop) done = true;
        if (!done && iters >= args.max_iter) done = true;
        if (__all(done)) break;
        if (__any(rejected && !done)) {
          // a group rejected its trial (rare): its Jacobian buffer holds J(xt) and its triangle
          // the factor.  Every group re-evaluates at its accepted point (identical values for
          // the groups that did accept) and the rejecting groups rebuild J^T J.
          eval_rows<true>(P, S, l, W, x, xaddr, cur, prof);
          if (rejected && !done) ++nfev;
          const bool again = rejected && !done;
          const double gnew = build_normal(P, S, again ? l : (1 << 20), W, cur, again && is_var);
          if (again) grad = gnew;
        }
        if (!done) ++iters;
        OKX_STAMP(4)
        // damped normal equations; a group whose factorisation fails retries with more damping
        bool have = done;
        double dx_new = 0.0;
        for (int tries = 0; tries < 60; ++tries) {
          double cand;
          const bool ok = ldlt_solve_lds<N>(P, S, l, lambda, grad, &cand,
                                            [W, g](double v, int k) { return group_bcast<G>(v, k, W, g); });
          if (!have) {
            if (ok) {
              dx_new = cand;
              have = true;
            } else {
              lambda = fmax(lambda * 10.0, 1e-12 * dmax);
              if (!(lambda < 1e30)) {
                flags |= OKX_INFO_FAILED;
                done = true;
                have = true;
              }
            }
          }
          if (__all(have)) break;
          // groups still without a factor re-scatter J^T J (the failed factor overwrote it)
          build_normal(P, S, !have ? l : (1 << 20), W, cur, false);
        }
        if (!have) {
          flags |= OKX_INFO_FAILED;
          done = true;
        }
        OKX_STAMP(6)
        dx = done ? 0.0 : dx_new;
        step_len = group_max(is_var ? fabs(dx) : 0.0, L.scratch, lane, gbase, W);
        if (!done && step_len <= args.step_tol) {  // correction below tolerance: x is the answer
          flags |= OKX_INFO_CONVERGED;
          last_step = step_len;
          done = true;
          dx = 0.0;
        }
        if (__all(done)) break;
        xt = x + dx;
      }

      OKX_STAMP(4)
      // final state: free points, then every derived point (incl. output-only ones)
      wave_sync();
      if (is_var) S.pos[xaddr] = x;
      wave_sync();
      derived_update<false>(P, S, l, W, false);
      double ra = 0.0;
      for (int i = l; i < m; i += W) ra = fmax(ra, reference_abs_residual(P, S, i, cur));
      const double max_res = group_max(valid ? ra : 0.0, L.scratch, lane, gbase, W);
      if (max_res > args.residual_tolerance) flags |= OKX_INFO_RESIDUAL_EXCEEDED;
      if (has_unit) {
        double* out = args.out_pos + b * 3 * P->n_out;
        for (int e = l; e < 3 * P->n_out; e += W) out[e] = S.pos[3 * P->out_point[e / 3] + e % 3];
        if (l == 0) {
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
        if (!(flags & OKX_INFO_CONVERGED) || (flags & OKX_INFO_FAILED)) {
          const double* src = gp ? gp : &P->design_pos[0][0];
          if (is_var) x = src[xaddr];
          hist = 0;
        } else if (hist < 2) {
          ++hist;
        }
      }
      OKX_STAMP(8)
    }
  }
  if constexpr (PROFILE) {
    if (blockIdx.x == 0 && lane == 0 && args.phase_cycles)
      for (int k = 0; k < 12; ++k) args.phase_cycles[k] = prof_store.phase[k];
  }
}

}  // namespace okx
