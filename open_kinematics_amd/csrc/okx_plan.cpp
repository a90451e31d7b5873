// okx_plan.cpp — host-side construction of the device program and its sparsity plans.
// Replaces ResidualComputer.__init__ / build_jac_plan (reference core/solver.py:187-214,
// :281-500) and DerivedPointsManager._get_computation_plan (points/derived/manager.py:199-250).
#include "okx_plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

namespace okx {

namespace {

int points_of_row_type(int type) {
  switch (type) {
    case OKX_ROW_DISTANCE:
    case OKX_ROW_SPHERICAL:
    case OKX_ROW_MIDPOINT_ON_PLANE:
      return 2;
    case OKX_ROW_THREE_POINT_ANGLE:
      return 3;
    case OKX_ROW_ANGLE:
    case OKX_ROW_VECTORS_PARALLEL:
    case OKX_ROW_VECTORS_PERPENDICULAR:
    case OKX_ROW_EQUAL_DISTANCE:
    case OKX_ROW_COPLANAR:
    case OKX_ROW_SCALAR_TRIPLE:
      return 4;
    case OKX_ROW_FIXED_AXIS:
    case OKX_ROW_POINT_ON_LINE:
    case OKX_ROW_POINT_ON_PLANE:
    case OKX_ROW_LINE_PIN:
    case kRowTarget:
      return 1;
    default:
      return -1;
  }
}

int inputs_of_dop_type(int type) {
  switch (type) {
    case OKX_DOP_MIDPOINT:
      return 2;
    case OKX_DOP_ALONG:
    case OKX_DOP_CONTACT_PATCH:
      return 3;
    default:
      return -1;
  }
}

#define FAIL(code, ...)                      \
  do {                                       \
    std::snprintf(err, errlen, __VA_ARGS__); \
    return code;                             \
  } while (0)

}  // namespace

int build_dev_program(const okx_program_desc* d, DevProgram* out, char* err, int errlen) {
  if (!d || !out) FAIL(OKX_ERR_INVALID, "null descriptor");
  if (d->abi_version != OKX_ABI_VERSION)
    FAIL(OKX_ERR_INVALID, "ABI version mismatch: got %d, library is %d", d->abi_version,
         OKX_ABI_VERSION);
  const int P = d->n_points, F = d->n_free, D = d->n_derived, Mc = d->n_rows, T = d->n_targets;
  const int n = 3 * F, m = Mc + T;
  if (P <= 0 || F <= 0 || D < 0 || Mc < 0 || T < 0 || d->n_out < 0)
    FAIL(OKX_ERR_INVALID, "negative or empty dimension");
  if (P > kMaxPoints || n > kMaxVars || D > kMaxDerived || m > kMaxRows || T > kMaxTargets ||
      d->n_out > kMaxPoints)
    FAIL(OKX_ERR_LIMIT, "problem exceeds one-wavefront limits (points=%d vars=%d rows=%d)", P, n,
         m);
  if (n > m)
    FAIL(OKX_ERR_UNDERDETERMINED,
         "System is underdetermined (n_vars=%d > m_res=%d). The solve method "
         "(Levenberg-Marquardt) requires at least as many residuals as variables.",
         n, m);

  std::memset(out, 0, sizeof(DevProgram));
  out->n_points = P;
  out->n_free = F;
  out->n = n;
  out->n_derived = D;
  out->n_crows = Mc;
  out->n_targets = T;
  out->m = m;
  out->n_out = d->n_out;

  std::vector<int> block_of_point(P, -1), dop_of_point(P, -1);
  for (int k = 0; k < F; ++k) {
    int p = d->free_point[k];
    if (p < 0 || p >= P || block_of_point[p] >= 0) FAIL(OKX_ERR_INVALID, "bad free_point[%d]", k);
    block_of_point[p] = k;
    out->free_point[k] = p;
  }
  for (int k = 0; k < d->n_out; ++k) {
    int p = d->out_point[k];
    if (p < 0 || p >= P) FAIL(OKX_ERR_INVALID, "bad out_point[%d]", k);
    out->out_point[k] = p;
  }
  std::memcpy(out->design_pos, d->design_pos, sizeof(double) * 3 * P);

  // ---- derived ops: dependency block lists and input references ----
  auto make_ref = [&](int point, const std::vector<int>& owner_blocks, PointRef* ref) -> bool {
    ref->w0 = kRefFixed;
    ref->w1 = 0;
    if (point < 0) return true;
    if (block_of_point[point] >= 0) {
      auto it = std::find(owner_blocks.begin(), owner_blocks.end(), block_of_point[point]);
      if (it == owner_blocks.end()) return false;
      ref->w0 = kRefFree | ((uint32_t)(it - owner_blocks.begin()) << 8);
      return true;
    }
    int e = dop_of_point[point];
    if (e < 0) return true;  // fixed point
    ref->w0 = kRefDerived | ((uint32_t)e << 16) | ((uint32_t)out->dop_nblk[e] << 24);
    for (int j = 0; j < out->dop_nblk[e]; ++j) {
      auto it = std::find(owner_blocks.begin(), owner_blocks.end(), out->dop_blk[e][j]);
      if (it == owner_blocks.end()) return false;
      ref->w1 |= (uint32_t)(it - owner_blocks.begin()) << (8 * j);
    }
    return true;
  };
  auto blocks_of_point = [&](int point, std::vector<int>& acc) {
    if (point < 0) return;
    if (block_of_point[point] >= 0) {
      acc.push_back(block_of_point[point]);
    } else if (dop_of_point[point] >= 0) {
      int e = dop_of_point[point];
      for (int j = 0; j < out->dop_nblk[e]; ++j) acc.push_back(out->dop_blk[e][j]);
    }
  };
  auto uniq = [](std::vector<int>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };

  for (int e = 0; e < D; ++e) {
    int type = d->dop_type[e];
    int nin = inputs_of_dop_type(type);
    if (nin < 0) FAIL(OKX_ERR_INVALID, "unknown derived op type %d", type);
    int o = d->dop_out[e];
    if (o < 0 || o >= P || block_of_point[o] >= 0 || dop_of_point[o] >= 0)
      FAIL(OKX_ERR_INVALID, "derived op %d: bad output point", e);
    out->dop_type[e] = type;
    out->dop_out[e] = o;
    out->dop_param[e] = d->dop_param[e];
    out->dop_active[e] = -1;
    std::vector<int> blocks;
    for (int s = 0; s < 4; ++s) {
      int p = d->dop_pts[4 * e + s];
      out->dop_pts[e][s] = s < nin ? p : -1;
      if (s >= nin) continue;
      if (p < 0 || p >= P) FAIL(OKX_ERR_INVALID, "derived op %d: bad input point", e);
      if (dop_of_point[p] < 0 && block_of_point[p] < 0) {
        // fixed input, or a derived point defined later (order violation)
        for (int later = e; later < D; ++later)
          if (d->dop_out[later] == p)
            FAIL(OKX_ERR_INVALID, "derived op %d reads point %d before it is computed", e, p);
      }
      blocks_of_point(p, blocks);
    }
    uniq(blocks);
    if ((int)blocks.size() > kDepMax)
      FAIL(OKX_ERR_LIMIT, "derived op %d depends on %d free points (max %d)", e,
           (int)blocks.size(), kDepMax);
    out->dop_nblk[e] = (int)blocks.size();
    for (size_t j = 0; j < blocks.size(); ++j) out->dop_blk[e][j] = blocks[j];
    for (int s = 0; s < nin; ++s)
      if (!make_ref(out->dop_pts[e][s], blocks, &out->dop_in[e][s]))
        FAIL(OKX_ERR_INVALID, "derived op %d: inconsistent dependency map", e);
    dop_of_point[o] = e;
  }

  // ---- rows: constraints then targets ----
  int max_nblk = 1;
  for (int i = 0; i < m; ++i) {
    int type;
    int pts[4] = {-1, -1, -1, -1};
    if (i < Mc) {
      type = d->row_type[i];
      if (type < 0 || type >= OKX_ROW_TYPE_COUNT) FAIL(OKX_ERR_INVALID, "row %d: bad type", i);
      for (int s = 0; s < 4; ++s) pts[s] = d->row_pts[4 * i + s];
      std::memcpy(out->row_param[i], d->row_param + OKX_ROW_PARAMS * i,
                  sizeof(double) * OKX_ROW_PARAMS);
      if (type == OKX_ROW_FIXED_AXIS) {
        int ax = (int)out->row_param[i][0];
        if (ax < 0 || ax > 2) FAIL(OKX_ERR_INVALID, "row %d: bad axis", i);
      }
      if (type == OKX_ROW_LINE_PIN) {
        int c = (int)out->row_param[i][6];
        if (c < 0 || c > 2) FAIL(OKX_ERR_INVALID, "row %d: bad line-pin component", i);
      }
      if (type == OKX_ROW_SCALAR_TRIPLE && !(out->row_param[i][1] > 0.0))
        FAIL(OKX_ERR_INVALID, "row %d: scale must be strictly positive", i);
    } else {
      int t = i - Mc;
      type = kRowTarget;
      pts[0] = d->tgt_point[t];
      out->row_param[i][0] = d->tgt_dir[3 * t + 0];
      out->row_param[i][1] = d->tgt_dir[3 * t + 1];
      out->row_param[i][2] = d->tgt_dir[3 * t + 2];
      out->row_param[i][3] = (double)t;
    }
    int np = points_of_row_type(type);
    out->row_type[i] = type;
    std::vector<int> blocks;
    for (int s = 0; s < 4; ++s) {
      if (s >= np) {
        out->row_pts[i][s] = -1;
        continue;
      }
      if (pts[s] < 0 || pts[s] >= P) FAIL(OKX_ERR_INVALID, "row %d: bad point slot %d", i, s);
      out->row_pts[i][s] = pts[s];
      blocks_of_point(pts[s], blocks);
    }
    uniq(blocks);
    if ((int)blocks.size() > kRowBlkMax)
      FAIL(OKX_ERR_LIMIT, "row %d touches %d free points (max %d)", i, (int)blocks.size(),
           kRowBlkMax);
    out->row_nblk[i] = (int)blocks.size();
    max_nblk = std::max(max_nblk, (int)blocks.size());
    for (size_t j = 0; j < blocks.size(); ++j) out->row_blk[i][j] = blocks[j];
    for (int s = 0; s < np; ++s) {
      if (!make_ref(out->row_pts[i][s], blocks, &out->row_in[i][s]))
        FAIL(OKX_ERR_INVALID, "row %d: inconsistent dependency map", i);
      // mark the derived chain this row needs while iterating
      if (out->row_in[i][s].kind() == kRefDerived) out->dop_active[out->row_in[i][s].src()] = 0;
    }
  }
  // propagate activity backwards through derived inputs, then number the active ops
  for (int e = D - 1; e >= 0; --e) {
    if (out->dop_active[e] < 0) continue;
    int nin = inputs_of_dop_type(out->dop_type[e]);
    for (int s = 0; s < nin; ++s)
      if (out->dop_in[e][s].kind() == kRefDerived) out->dop_active[out->dop_in[e][s].src()] = 0;
  }
  int n_active = 0;
  for (int e = 0; e < D; ++e)
    if (out->dop_active[e] >= 0) {
      out->dop_active[e] = n_active;
      out->active_op[n_active++] = e;
    }
  out->n_active = n_active;

  // ---- J^T J pair plan and J^T r plan ----
  std::map<std::pair<int, int>, std::vector<uint16_t>> pairs;
  std::vector<std::vector<uint16_t>> gl(F);
  for (int i = 0; i < m; ++i) {
    for (int a = 0; a < out->row_nblk[i]; ++a) {
      gl[out->row_blk[i][a]].push_back((uint16_t)(i | (a << 7)));
      for (int b = 0; b <= a; ++b) {
        int p = out->row_blk[i][a], q = out->row_blk[i][b];  // p >= q (sorted lists)
        pairs[{q, p}].push_back((uint16_t)(i | (b << 7) | (a << 10)));
      }
    }
  }
  // every diagonal block must exist (otherwise a variable is unconstrained: J^T J singular
  // by structure); keep it as an empty pair so the matrix entry is defined.
  for (int k = 0; k < F; ++k) pairs[{k, k}];
  if ((int)pairs.size() > kMaxPairs) FAIL(OKX_ERR_LIMIT, "too many block pairs");
  int np = 0, nc = 0;
  for (auto& kv : pairs) {
    out->pair_p[np] = kv.first.first;   // row block (p <= q: upper triangle)
    out->pair_q[np] = kv.first.second;  // column block
    out->pair_start[np] = nc;
    if (nc + (int)kv.second.size() > kMaxContrib) FAIL(OKX_ERR_LIMIT, "too many contributions");
    for (uint16_t c : kv.second) out->contrib[nc++] = c;
    ++np;
  }
  out->pair_start[np] = nc;
  out->n_pairs = np;
  out->n_items = np * 9;
  int ng = 0;
  for (int k = 0; k < F; ++k) {
    out->g_start[k] = ng;
    for (uint16_t c : gl[k]) out->g_contrib[ng++] = c;
  }
  out->g_start[F] = ng;

  // ---- LDS layout sizes ----
  out->js_stride = 3 * max_nblk + 1;                 // odd number of doubles: conflict-free rows
  int lda = n;
  while (lda % 4 != 2) ++lda;                        // lda = 2 (mod 4): see okx_kernels.hip
  out->lda = lda;

  // ---- flattened work items ----
  int nw = 0;
  for (int pr = 0; pr < np; ++pr) {
    const int bp = out->pair_p[pr], bq = out->pair_q[pr];
    const int start = out->pair_start[pr], count = out->pair_start[pr + 1] - start;
    if (start >= (1 << 12) || count >= (1 << 8)) FAIL(OKX_ERR_LIMIT, "plan item out of packing range");
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        if (bp == bq && a > b) continue;
        const int row = 3 * bp + a, col = 3 * bq + b;
        out->item_dst[nw] = row == col ? -(1 + row) : col * (col - 1) / 2 + row;
        out->item_desc[nw] = (uint32_t)start | ((uint32_t)count << 12) | ((uint32_t)a << 20) |
                             ((uint32_t)b << 22);
        ++nw;
      }
  }
  out->n_work = nw;
  out->n_contrib = nc;
  out->n_gcontrib = ng;

  // ---- batched-load tables ----
  const int stride = out->js_stride;
  out->zero_off = m * stride;
  int kc = 1, kg = 1;
  for (int pr = 0; pr < np; ++pr) kc = std::max(kc, out->pair_start[pr + 1] - out->pair_start[pr]);
  for (int k = 0; k < F; ++k) kg = std::max(kg, out->g_start[k + 1] - out->g_start[k]);
  kc = (kc + 3) / 4 * 4;
  kg = (kg + 3) / 4 * 4;
  if (kc > kItemTermsMax || kg > kGradTermsMax)
    FAIL(OKX_ERR_LIMIT, "a free point takes part in too many rows (%d / %d product terms)", kc, kg);
  if (out->zero_off + 1 >= (1 << 16)) FAIL(OKX_ERR_LIMIT, "Jacobian buffer too large for 16-bit offsets");
  out->kc = kc;
  out->kg = kg;
  for (int w = 0; w < nw; ++w) {
    const uint32_t desc = out->item_desc[w];
    const int start = desc & 0xfff, count = (desc >> 12) & 0xff, a = (desc >> 20) & 3, b = (desc >> 22) & 3;
    for (int c = 0; c < kc; ++c) {
      uint32_t term = (uint32_t)out->zero_off | ((uint32_t)out->zero_off << 16);
      if (c < count) {
        const int pk = out->contrib[start + c];
        const int row = pk & 127, sp = (pk >> 7) & 7, sq = (pk >> 10) & 7;
        term = (uint32_t)(row * stride + 3 * sp + a) | ((uint32_t)(row * stride + 3 * sq + b) << 16);
      }
      out->item_terms[w * kc + c] = term;
    }
  }
  for (int v = 0; v < n; ++v) {
    const int blk = v / 3, a = v % 3;
    const int start = out->g_start[blk], count = out->g_start[blk + 1] - start;
    for (int c = 0; c < kg; ++c) {
      uint32_t term = (uint32_t)out->zero_off | ((uint32_t)m << 16);
      if (c < count) {
        const int pk = out->g_contrib[start + c];
        const int row = pk & 127, sl = (pk >> 7) & 7;
        term = (uint32_t)(row * stride + 3 * sl + a) | ((uint32_t)row << 16);
      }
      out->grad_terms[v * kg + c] = term;
    }
  }
  // first-writer masks of the row scatter (program order: point slot, then producer block)
  for (int i = 0; i < m; ++i) {
    bool written[kRowBlkMax] = {false, false, false, false, false, false};
    uint32_t mask = 0;
    const int npts = points_of_row_type(out->row_type[i]);
    for (int sl = 0; sl < npts; ++sl) {
      const PointRef ref = out->row_in[i][sl];
      if (ref.kind() == kRefFree) {
        if (!written[ref.slot()]) mask |= 1u << (4 * sl);
        written[ref.slot()] = true;
      } else if (ref.kind() == kRefDerived) {
        for (int j = 0; j < ref.nsrc(); ++j) {
          if (!written[ref.map(j)]) mask |= 1u << (4 * sl + j);
          written[ref.map(j)] = true;
        }
      }
    }
    out->row_first[i] = mask;
  }
  return OKX_OK;
}

}  // namespace okx
