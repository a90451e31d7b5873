// okx_lanegen.cpp — source generator of the "lane" solve kernel: ONE LANE OWNS ONE PROBLEM.
//
// Second member of the runtime-specialised kernel family (the first is the quad kernel of okx_quadgen.cpp).
// Same objective, same residual / Jacobian definitions (reference core/constraints.py, core/jacobians.py,
// core/solver.py:226-275, :502-581), same Levenberg-Marquardt policy and the same evaluation points as the quad
// kernel; only the parallel decomposition differs:
//
//   * quad kernel: four lanes own one problem, lane c holds Cartesian component c.  Every dot product, cross
//     product, J^T J column and pivot crosses lanes through DPP (two v_mov_b32_dpp per double), one lane in four
//     idles, and the three working lanes repeat all scalar arithmetic.  16 problems per wavefront: the right
//     shape when a batch only just fills the chip (16384 problems = one wavefront per SIMD).
//   * lane kernel (this file): a problem lives entirely in the registers (and a little LDS) of ONE lane, 64
//     problems per wavefront.  No cross-lane operand exists: the instruction stream of a pass is fp64 arithmetic
//     on statically named registers, about a quarter of the quad kernel's instructions per problem.  The price
//     is the register file: the lower triangle of J^T J alone is 135 doubles for the double wishbone, so the
//     kernel runs one wavefront per SIMD with the cold part of the state (accepted point, step in hand, chain
//     history) in LDS, [slot][lane] so that every access is conflict-free.  It is the right shape for batches of
//     at least 64 problems per SIMD (grids, ensembles: BASELINE configs 4 and 5).
//
// Geometry data (fixed points, row parameters, the first-step table) is WAVE-UNIFORM here: a wavefront's 64
// problems always belong to one geometry (work units never straddle a span), so those values are read once per
// wave unit and live in scalar registers; fp64 VALU instructions take them as scalar operands.
//
// The first-step tables are the quad module's (okx_quad_head_u/_g); this generator uses the same block
// elimination order, so the table layout is shared.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "okx_plan.hpp"
#include "okx_quad.hpp"

namespace okx {
namespace {

// Three scalar expressions (names of doubles); an empty string is a structural zero.  `sg` folds a sign into uses.
struct S3 {
  std::string c[3];
  int sg = 1;
};

// d(derived point) / d(free block): s * I, or a full 3 x 3 block m[r][k] = d out_r / d free_k.
struct LBlk {
  bool scaled = true;
  std::string s;
  std::string m[3][3];
};

struct LBlkTerm {
  LBlk b;
  int sg;
};

class LGen {
 public:
  explicit LGen(const DevProgram& prog) : P(prog) {
    blk_of_point.assign(P.n_points, -1);
    dop_of_point.assign(P.n_points, -1);
    perm = lane_elimination_order(P);
    layout_tables();
    for (int F = 0; F < P.n_free; ++F) blk_of_point[fp(F)] = F;
    for (int e = 0; e < P.n_derived; ++e) dop_of_point[P.dop_out[e]] = e;
  }

  // Same greedy minimum-degree order as the quad generator (okx_quadgen.cpp, Gen::elimination_order): the
  // first-step table of okx_quad_head_* is laid out in that block order.
  static std::vector<int> lane_elimination_order(const DevProgram& P) {
    const int nf = P.n_free;
    std::vector<std::set<int>> adj(nf);
    for (int i = 0; i < P.m; ++i)
      for (int a = 0; a < P.row_nblk[i]; ++a)
        for (int b = 0; b < P.row_nblk[i]; ++b)
          if (a != b) adj[P.row_blk[i][a]].insert(P.row_blk[i][b]);
    std::vector<bool> gone(nf, false);
    std::vector<int> perm;
    for (int step = 0; step < nf; ++step) {
      int best = -1;
      for (int k = 0; k < nf; ++k)
        if (!gone[k] && (best < 0 || adj[k].size() < adj[best].size())) best = k;
      perm.push_back(best);
      gone[best] = true;
      for (int u : adj[best]) {
        adj[u].erase(best);
        for (int w : adj[best])
          if (w != u) adj[u].insert(w);
      }
      adj[best].clear();
    }
    return perm;
  }

  const DevProgram& P;
  std::vector<int> perm;
  int fp(int F) const { return P.free_point[perm[F]]; }
  std::string out, why;
  std::vector<int> blk_of_point, dop_of_point;
  int uid = 0;
  std::map<int, std::map<int, LBlk>> dblk;  // active derived op -> free block -> chain block
  bool nz[kMaxVars][kMaxVars] = {};         // scalar-level structure of the lower triangle (block-dense)
  bool fill[kMaxVars][kMaxVars] = {};       // ... after symbolic factorisation

  void f(const char* fmt, ...) {
    char buf[2048];
    va_list ap, again;
    va_start(ap, fmt);
    va_copy(again, ap);
    const int need = std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (need >= (int)sizeof(buf)) {
      std::string big((size_t)need + 1, '\0');
      std::vsnprintf(&big[0], big.size(), fmt, again);
      big.resize((size_t)need);
      out += big;
    } else if (need > 0) {
      out += buf;
    }
    va_end(again);
    out += '\n';
  }
  std::string tmp(const char* base) { return "_" + std::string(base) + std::to_string(uid++); }

  // ---- scalar vector algebra on named doubles ----
  static S3 pt(int p) {
    S3 v;
    for (int k = 0; k < 3; ++k) v.c[k] = "p" + std::to_string(p) + "_" + std::to_string(k);
    return v;
  }
  static std::string sgn(const S3& a, int k) { return a.sg < 0 ? "(-" + a.c[k] + ")" : a.c[k]; }
  static S3 neg(S3 a) {
    a.sg = -a.sg;
    return a;
  }
  // sum of products as one nested fma expression; skips structural zeros; "" when everything is zero
  static std::string sum_expr(const std::vector<std::pair<std::string, std::string>>& prods, const std::string& init = "") {
    std::string e = init;
    for (auto it = prods.rbegin(); it != prods.rend(); ++it) {
      if (it->first.empty() || it->second.empty()) continue;
      if (e.empty()) e = it->first + " * " + it->second;
      else e = "fma(" + it->first + ", " + it->second + ", " + e + ")";
    }
    return e;
  }
  std::string emit(const char* base, const std::string& expr) {
    if (expr.empty()) return "";
    std::string t = tmp(base);
    f("    double %s = %s;", t.c_str(), expr.c_str());
    return t;
  }
  S3 sub(const S3& a, const S3& b) {
    S3 r;
    for (int k = 0; k < 3; ++k) {
      if (a.c[k].empty() && b.c[k].empty()) continue;
      if (b.c[k].empty()) r.c[k] = emit("v", sgn(a, k));
      else if (a.c[k].empty()) r.c[k] = emit("v", "-" + sgn(b, k));
      else r.c[k] = emit("v", sgn(a, k) + " - " + sgn(b, k));
    }
    return r;
  }
  // |v|^2 in the QUAD kernels' order of operations (qsum of the per-lane squares: (x x + y y) + z z, every product and sum
  // rounded, okx_quadgen.cpp) - the final state's derived points are written with it, so that a record never depends on
  // which kernel family wrote it: lane solve, quad solve and okx_expand_positions_batch of the same free coordinates give
  // the same bits (what dist.ShardedEnsemble relies on: one rank writes records, N ranks gather coordinates and expand)
  bool quad_order = false;
  std::string norm2(const S3& v) {
    if (!quad_order) return dot(v, v);
    std::string e;
    for (int k = 0; k < 3; ++k) {
      if (v.c[k].empty()) continue;
      const std::string sq = "okx_mul_rn(" + v.c[k] + ", " + v.c[k] + ")";
      e = e.empty() ? sq : "okx_add_rn(" + e + ", " + sq + ")";
    }
    return e.empty() ? std::string("0.0") : emit("d", e);
  }
  std::string dot(const S3& a, const S3& b, const std::string& init = "") {
    std::vector<std::pair<std::string, std::string>> pr;
    const int sg = a.sg * b.sg;
    for (int k = 0; k < 3; ++k)
      if (!a.c[k].empty() && !b.c[k].empty()) pr.push_back({sg < 0 ? "(-" + a.c[k] + ")" : a.c[k], b.c[k]});
    const std::string e = sum_expr(pr, init);
    if (e.empty()) return "0.0";
    return emit("d", e);
  }
  // (a x b)_k = a_{k+1} b_{k+2} - a_{k+2} b_{k+1}
  S3 cross(const S3& a, const S3& b) {
    S3 r;
    r.sg = a.sg * b.sg;
    for (int k = 0; k < 3; ++k) {
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      const bool t1 = !a.c[k1].empty() && !b.c[k2].empty(), t2 = !a.c[k2].empty() && !b.c[k1].empty();
      if (t1 && t2) r.c[k] = emit("cx", "fma(" + a.c[k1] + ", " + b.c[k2] + ", -(" + a.c[k2] + " * " + b.c[k1] + "))");
      else if (t1) r.c[k] = emit("cx", a.c[k1] + " * " + b.c[k2]);
      else if (t2) r.c[k] = emit("cx", "-(" + a.c[k2] + " * " + b.c[k1] + ")");
    }
    return r;
  }
  S3 scale(const std::string& k, const S3& a) {
    S3 r;
    r.sg = a.sg;
    for (int i = 0; i < 3; ++i)
      if (!a.c[i].empty()) r.c[i] = emit("s", k + " * " + a.c[i]);
    return r;
  }
  static S3 unit(int axis) {
    S3 r;
    r.c[axis] = "1.0";
    return r;
  }

  // ---- chain constants: wave-uniform values (row parameters, derived-op parameters, fixed points) ----
  // The geometry's tables (positions, constraint-row parameters; the program's target-row and derived-op parameters) are
  // STAGED into LDS once per wave unit with coalesced loads (lane k fetches entry k) and read where they are used: one
  // ds_read_b64 each, every lane the same address (a broadcast), through an index the optimiser cannot see through
  // (`kz`, an opaque zero refreshed at the top of every pass).  As named values loaded ahead of the loops (the first
  // version) they are ~50 loop invariants that the compiler parks in vector registers and spills; as ~200 same-address
  // global loads per wave unit (the second) the ensemble kernel spent 43 % of its cycles waiting (profiles/r03).
  // A constant's name is a macro: `#define hs3_0 gl[GQ0 + 24 + kz]`.
  std::set<std::string> defines;   // #define name gl[offset + kz]
  std::set<std::string> undefs;
  int gl_gp0 = 0, gl_gq0 = 0, gl_tq0 = 0, gl_dp0 = 0, gl_size = 0;
  void layout_tables() {
    gl_gp0 = 0;
    gl_gq0 = gl_gp0 + 3 * P.n_points;
    gl_tq0 = gl_gq0 + 8 * P.n_crows;
    gl_dp0 = gl_tq0 + 8 * P.n_targets;
    gl_size = gl_dp0 + (P.n_derived > 0 ? P.n_derived : 1);
  }
  std::map<std::pair<int, int>, std::string> hoisted_names;
  void add_const(const char* name, int offset) {
    char line[256];
    std::snprintf(line, sizeof(line), "#define %s GL(%d)\n", name, offset);
    defines.insert(line);
    std::snprintf(line, sizeof(line), "#undef %s\n", name);
    undefs.insert(line);
  }
  std::string dp(int e) {
    auto key = std::make_pair(-1 - e, 0);
    auto it = hoisted_names.find(key);
    if (it != hoisted_names.end()) return it->second;
    char name[48];
    std::snprintf(name, sizeof(name), "hd%d", e);
    add_const(name, gl_dp0 + e);
    hoisted_names[key] = name;
    return name;
  }
  int pin_leader(int i) const {
    if (i >= P.n_crows || P.row_type[i] != OKX_ROW_LINE_PIN) return i;
    for (int j = 0; j < i; ++j) {
      if (P.row_type[j] != OKX_ROW_LINE_PIN || P.row_pts[j][0] != P.row_pts[i][0]) continue;
      bool same = true;
      for (int k = 0; k < 6; ++k) same = same && P.row_param[j][k] == P.row_param[i][k];
      if (same) return j;
    }
    return i;
  }
  std::string rp(int i, int k) {
    i = pin_leader(i);
    auto key = std::make_pair(i, k);
    auto it = hoisted_names.find(key);
    if (it != hoisted_names.end()) return it->second;
    char name[48];
    std::snprintf(name, sizeof(name), "hs%d_%d", i, k);
    add_const(name, i < P.n_crows ? gl_gq0 + 8 * i + k : gl_tq0 + 8 * (i - P.n_crows) + k);
    hoisted_names[key] = name;
    return name;
  }
  S3 rpv(int i, int k0) {
    S3 v;
    for (int k = 0; k < 3; ++k) v.c[k] = rp(i, k0 + k);
    return v;
  }
  int target_of_row(int i) const { return (int)P.row_param[i][3]; }
  std::map<int, S3> pin_cross_;

  // ---- derived points ----
  std::map<int, LBlk> blocks_of_point(int p) {
    std::map<int, LBlk> r;
    if (blk_of_point[p] >= 0) {
      LBlk b;
      b.scaled = true;
      b.s = "1.0";
      r[blk_of_point[p]] = b;
    } else if (dop_of_point[p] >= 0) {
      auto it = dblk.find(dop_of_point[p]);
      if (it != dblk.end()) r = it->second;
    }
    return r;
  }
  std::map<int, LBlk> combine(std::map<int, std::vector<LBlkTerm>>& acc) {
    std::map<int, LBlk> res;
    for (auto& kv : acc) {
      std::string ssum;
      std::vector<LBlkTerm*> gen;
      for (auto& t : kv.second) {
        if (t.b.scaled) {
          if (!ssum.empty()) ssum += t.sg < 0 ? " - " : " + ";
          else if (t.sg < 0) ssum += "-";
          ssum += "(" + t.b.s + ")";
        } else {
          gen.push_back(&t);
        }
      }
      LBlk b;
      if (gen.empty()) {
        b.scaled = true;
        b.s = emit("ks", ssum);
      } else {
        b.scaled = false;
        for (int r = 0; r < 3; ++r)
          for (int k = 0; k < 3; ++k) {
            std::string e;
            for (auto* t : gen) {
              if (t->b.m[r][k].empty()) continue;
              if (!e.empty()) e += t->sg < 0 ? " - " : " + ";
              else if (t->sg < 0) e += "-";
              e += t->b.m[r][k];
            }
            if (!ssum.empty() && r == k) e += (e.empty() ? "(" : " + (") + ssum + ")";
            b.m[r][k] = emit("B", e);
          }
      }
      res[kv.first] = b;
    }
    return res;
  }

  bool derived_op(int e, bool with_blocks) {
    const int type = P.dop_type[e];
    const int* pts = P.dop_pts[e];
    const S3 o = pt(P.dop_out[e]);
    f("    // derived op %d (type %d) -> point %d", e, type, P.dop_out[e]);
    if (type == OKX_DOP_MIDPOINT) {  // definitions.py:76-89
      const S3 a = pt(pts[0]), b = pt(pts[1]);
      for (int k = 0; k < 3; ++k) f("    %s = fma(%s - %s, 0.5, %s);", o.c[k].c_str(), b.c[k].c_str(), a.c[k].c_str(), a.c[k].c_str());
      if (with_blocks) {
        std::map<int, std::vector<LBlkTerm>> acc;
        for (int s = 0; s < 2; ++s)
          for (auto& kv : blocks_of_point(pts[s])) {
            LBlk b2 = kv.second;
            if (b2.scaled) {
              b2.s = "0.5 * (" + b2.s + ")";
            } else {
              for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k)
                  if (!b2.m[r][k].empty()) b2.m[r][k] = emit("B", "0.5 * " + b2.m[r][k]);
            }
            acc[kv.first].push_back({b2, 1});
          }
        dblk[e] = combine(acc);
      }
      return true;
    }
    if (type == OKX_DOP_ALONG) {  // definitions.py:24-33, :92-155: out = base + normalize(a - b) * c
      const S3 v = sub(pt(pts[1]), pt(pts[2]));
      const std::string s2 = norm2(v);
      const std::string nrm = tmp("nr"), inrm = tmp("in");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", nrm.c_str(), inrm.c_str(), s2.c_str(), nrm.c_str(), inrm.c_str());
      const S3 u = scale(inrm, v);
      const S3 base = pt(pts[0]);
      for (int k = 0; k < 3; ++k) f("    %s = fma(%s, %s, %s);", o.c[k].c_str(), u.c[k].c_str(), dp(e).c_str(), base.c[k].c_str());
      if (with_blocks) {
        std::map<int, std::vector<LBlkTerm>> acc;
        for (auto& kv : blocks_of_point(pts[0])) acc[kv.first].push_back({kv.second, 1});
        const std::string k = emit("k", dp(e) + " * " + inrm);
        for (int s = 1; s <= 2; ++s) {
          const int sg = s == 1 ? 1 : -1;
          for (auto& kv : blocks_of_point(pts[s])) {
            LBlk nb;
            nb.scaled = false;
            const LBlk& b = kv.second;
            if (b.scaled) {  // k s (I - u u^T): symmetric
              const std::string ks = emit("ks", k + " * (" + b.s + ")");
              for (int r = 0; r < 3; ++r)
                for (int c = 0; c <= r; ++c) {
                  const std::string uu = u.c[r] + " * " + u.c[c];
                  nb.m[r][c] = emit("B", r == c ? ks + " * (1.0 - " + uu + ")" : "-(" + ks + " * (" + uu + "))");
                  nb.m[c][r] = nb.m[r][c];
                }
            } else {  // k (B - u (u^T B))
              for (int c = 0; c < 3; ++c) {
                const std::string ub = emit("ub", sum_expr({{u.c[0], b.m[0][c]}, {u.c[1], b.m[1][c]}, {u.c[2], b.m[2][c]}}));
                for (int r = 0; r < 3; ++r) {
                  std::string e2;
                  if (!b.m[r][c].empty() && !ub.empty()) e2 = k + " * fma(-" + u.c[r] + ", " + ub + ", " + b.m[r][c] + ")";
                  else if (!b.m[r][c].empty()) e2 = k + " * " + b.m[r][c];
                  else if (!ub.empty()) e2 = "-(" + k + " * " + u.c[r] + " * " + ub + ")";
                  nb.m[r][c] = emit("B", e2);
                }
              }
            }
            acc[kv.first].push_back({nb, sg});
          }
        }
        dblk[e] = combine(acc);
      }
      return true;
    }
    if (type == OKX_DOP_CONTACT_PATCH) {  // definitions.py:36-73, :158-180
      const S3 v = sub(pt(pts[2]), pt(pts[1]));
      const std::string vv = norm2(v);
      const std::string vn = tmp("vn"), ivn = tmp("iv");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", vn.c_str(), ivn.c_str(), vv.c_str(), vn.c_str(), ivn.c_str());
      const S3 ax = scale(ivn, v);
      const std::string az = ax.c[2];
      S3 wd;  // -ga a - e_z with ga = -a_z
      wd.c[0] = emit("wd", az + " * " + ax.c[0]);
      wd.c[1] = emit("wd", az + " * " + ax.c[1]);
      wd.c[2] = emit("wd", "fma(" + az + ", " + ax.c[2] + ", -1.0)");
      const std::string ww = norm2(wd);
      const std::string wn = tmp("wn"), iwn = tmp("iw");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", wn.c_str(), iwn.c_str(), ww.c_str(), wn.c_str(), iwn.c_str());
      const S3 wc = pt(pts[0]);
      const std::string rk = emit("rk", iwn + " * " + dp(e));
      if (quad_order && !with_blocks) {  // (wd / |wd|) * R + wc, the unit vector rounded first (okx_quadgen.cpp)
        const S3 wu = scale(iwn, wd);
        for (int k = 0; k < 3; ++k) f("    %s = fma(%s, %s, %s);", o.c[k].c_str(), wu.c[k].c_str(), dp(e).c_str(), wc.c[k].c_str());
        return true;
      }
      for (int k = 0; k < 3; ++k) f("    %s = fma(%s, %s, %s);", o.c[k].c_str(), wd.c[k].c_str(), rk.c_str(), wc.c[k].c_str());
      if (with_blocks) {
        // d out / d axo = T = R Nw Wa Na (axi: -T, wheel centre: I), Na = (I - a a^T)/|v|, Wa = a_z I + a e_z^T,
        // Nw = (I - wu wu^T)/|wd|; applied column by column to the input's own block.
        const S3 wu = scale(iwn, wd);
        auto apply_t = [&](const std::string b[3], std::string t[3]) {
          const std::string adb = emit("ad", sum_expr({{ax.c[0], b[0]}, {ax.c[1], b[1]}, {ax.c[2], b[2]}}));
          std::string nn[3], w[3];
          for (int r = 0; r < 3; ++r) {
            std::string e2;
            if (!b[r].empty() && !adb.empty()) e2 = ivn + " * fma(-" + ax.c[r] + ", " + adb + ", " + b[r] + ")";
            else if (!b[r].empty()) e2 = ivn + " * " + b[r];
            else if (!adb.empty()) e2 = "-(" + ivn + " * " + ax.c[r] + " * " + adb + ")";
            nn[r] = emit("n", e2);
          }
          for (int r = 0; r < 3; ++r) w[r] = emit("w", sum_expr({{az, nn[r]}, {ax.c[r], nn[2]}}));
          const std::string wdw = emit("ww", sum_expr({{wu.c[0], w[0]}, {wu.c[1], w[1]}, {wu.c[2], w[2]}}));
          for (int r = 0; r < 3; ++r) {
            std::string e2;
            if (!w[r].empty() && !wdw.empty()) e2 = rk + " * fma(-" + wu.c[r] + ", " + wdw + ", " + w[r] + ")";
            else if (!w[r].empty()) e2 = rk + " * " + w[r];
            else if (!wdw.empty()) e2 = "-(" + rk + " * " + wu.c[r] + " * " + wdw + ")";
            t[r] = emit("B", e2);
          }
        };
        std::map<int, std::vector<LBlkTerm>> acc;
        for (auto& kv : blocks_of_point(pts[0])) acc[kv.first].push_back({kv.second, 1});
        for (int sidx = 1; sidx <= 2; ++sidx) {
          const int sg = sidx == 2 ? 1 : -1;
          for (auto& kv : blocks_of_point(pts[sidx])) {
            const LBlk& b = kv.second;
            LBlk nb;
            nb.scaled = false;
            for (int c = 0; c < 3; ++c) {
              std::string col[3], t[3];
              for (int r = 0; r < 3; ++r) col[r] = b.scaled ? (r == c ? "(" + b.s + ")" : std::string()) : b.m[r][c];
              apply_t(col, t);
              for (int r = 0; r < 3; ++r) nb.m[r][c] = t[r];
            }
            acc[kv.first].push_back({nb, sg});
          }
        }
        dblk[e] = combine(acc);
      }
      return true;
    }
    why = "unknown derived op";
    return false;
  }

  // ---- directional derivatives of the derived points (forward mode, closed form; the evaluated module's epilogue) ----
  // vel[d][p]: velocity of point p along direction d as three names ("" = structurally zero: fixed points and what
  // only depends on them).  Replaces the reference's dual-number pass (sensitivity.py:127-131, primitives/dual.py).
  S3 add(const S3& a, const S3& b) {
    S3 r;
    for (int k = 0; k < 3; ++k) {
      if (a.c[k].empty() && b.c[k].empty()) continue;
      if (b.c[k].empty()) r.c[k] = emit("v", sgn(a, k));
      else if (a.c[k].empty()) r.c[k] = emit("v", sgn(b, k));
      else r.c[k] = emit("v", sgn(a, k) + " + " + sgn(b, k));
    }
    return r;
  }
  // k * (a - u * s) per component, s a scalar expression ("" = zero)
  S3 scaled_reject(const std::string& k, const S3& a, const S3& u, const std::string& s) {
    S3 r;
    for (int i = 0; i < 3; ++i) {
      const bool ha = !a.c[i].empty(), hu = !u.c[i].empty() && !s.empty() && s != "0.0";
      if (ha && hu) r.c[i] = emit("jv", k + " * fma(-" + u.c[i] + ", " + s + ", " + sgn(a, i) + ")");
      else if (ha) r.c[i] = emit("jv", k + " * " + sgn(a, i));
      else if (hu) r.c[i] = emit("jv", "-(" + k + " * " + u.c[i] + " * " + s + ")");
    }
    return r;
  }
  static bool is_zero(const S3& a) { return a.c[0].empty() && a.c[1].empty() && a.c[2].empty(); }
  bool derived_jvp(int e, std::vector<std::vector<S3>>& vel) {
    const int type = P.dop_type[e];
    const int* pts = P.dop_pts[e];
    const int o = P.dop_out[e];
    const int D = (int)vel.size();
    f("    // velocity of derived point %d (op %d, type %d)", o, e, type);
    if (type == OKX_DOP_MIDPOINT) {
      for (int d = 0; d < D; ++d) {
        const S3 &a = vel[d][pts[0]], &b = vel[d][pts[1]];
        S3 r;
        for (int k = 0; k < 3; ++k) {
          if (a.c[k].empty() && b.c[k].empty()) continue;
          if (a.c[k].empty()) r.c[k] = emit("jv", "0.5 * " + sgn(b, k));
          else if (b.c[k].empty()) r.c[k] = emit("jv", "0.5 * " + sgn(a, k));
          else r.c[k] = emit("jv", "fma(" + sgn(b, k) + " - " + sgn(a, k) + ", 0.5, " + sgn(a, k) + ")");
        }
        vel[d][o] = r;
      }
      return true;
    }
    if (type == OKX_DOP_ALONG) {  // out = base + c u, u = w / |w|, w = a - b
      const S3 w = sub(pt(pts[1]), pt(pts[2]));
      const std::string ww = dot(w, w);
      std::string nrm, inrm;
      sqrt_rsqrt(ww, &nrm, &inrm);
      const S3 u = scale(inrm, w);
      const std::string k = emit("k", dp(e) + " * " + inrm);
      for (int d = 0; d < D; ++d) {
        const S3 dw = sub(vel[d][pts[1]], vel[d][pts[2]]);
        S3 r = vel[d][pts[0]];
        if (!is_zero(dw)) {
          const std::string ud = dot(u, dw);
          r = add(vel[d][pts[0]], scaled_reject(k, dw, u, ud));
        }
        vel[d][o] = r;
      }
      return true;
    }
    if (type == OKX_DOP_CONTACT_PATCH) {  // out = wc + R wu, wu = wd / |wd|, wd = a_z a - e_z, a = v / |v|, v = axo - axi
      const S3 v = sub(pt(pts[2]), pt(pts[1]));
      const std::string vv = dot(v, v);
      std::string vn, ivn;
      sqrt_rsqrt(vv, &vn, &ivn);
      const S3 ax = scale(ivn, v);
      const std::string az = ax.c[2];
      S3 wd;
      wd.c[0] = emit("wd", az + " * " + ax.c[0]);
      wd.c[1] = emit("wd", az + " * " + ax.c[1]);
      wd.c[2] = emit("wd", "fma(" + az + ", " + ax.c[2] + ", -1.0)");
      const std::string ww = dot(wd, wd);
      std::string wn, iwn;
      sqrt_rsqrt(ww, &wn, &iwn);
      const S3 wu = scale(iwn, wd);
      const std::string rk = emit("rk", iwn + " * " + dp(e));
      for (int d = 0; d < D; ++d) {
        const S3 dv = sub(vel[d][pts[2]], vel[d][pts[1]]);
        S3 r = vel[d][pts[0]];
        if (!is_zero(dv)) {
          const std::string adv = dot(ax, dv);
          const S3 dax = scaled_reject(ivn, dv, ax, adv);
          const std::string daz = dax.c[2];  // (never structurally zero: a_z' has a part along every component of dv)
          S3 dwd;
          for (int i = 0; i < 3; ++i) {
            std::string ex;
            if (!daz.empty()) ex = daz + " * " + ax.c[i];
            if (!dax.c[i].empty()) ex = ex.empty() ? az + " * " + dax.c[i] : "fma(" + az + ", " + dax.c[i] + ", " + ex + ")";
            dwd.c[i] = emit("dw", ex);
          }
          const std::string wdw = dot(wu, dwd);
          r = add(vel[d][pts[0]], scaled_reject(rk, dwd, wu, wdw));
        }
        vel[d][o] = r;
      }
      return true;
    }
    why = "unknown derived op";
    return false;
  }

  // ---- rows ----
  struct RowOut {
    std::string r;
    std::vector<std::pair<int, S3>> partial;  // point -> d r / d point
    std::string absres;
  };

  void sqrt_rsqrt(const std::string& x, std::string* root, std::string* inv) {
    *root = tmp("rt");
    *inv = tmp("iv");
    f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", root->c_str(), inv->c_str(), x.c_str(), root->c_str(), inv->c_str());
  }

  bool row(int i, RowOut* ro) {
    const int type = P.row_type[i];
    const int* pts = P.row_pts[i];
    f("    // row %d (type %d)", i, type);
    const std::string r = "r" + std::to_string(i);
    ro->r = r;
    ro->absres = "fabs(" + r + ")";
    switch (type) {
      case OKX_ROW_DISTANCE:
      case OKX_ROW_SPHERICAL: {  // constraints.py:125-134,162-170; jacobians.py:35-51
        const S3 d = sub(pt(pts[1]), pt(pts[0]));
        const std::string s = dot(d, d, "EPS_SQ");
        std::string root, inv;
        sqrt_rsqrt(s, &root, &inv);
        const S3 g = scale(inv, d);
        if (type == OKX_ROW_DISTANCE)
          f("    const double %s = (%s - EPS) - %s;", r.c_str(), root.c_str(), rp(i, 0).c_str());
        else
          f("    const double %s = %s - EPS;", r.c_str(), root.c_str());
        ro->partial.push_back({pts[0], neg(g)});
        ro->partial.push_back({pts[1], g});
        return true;
      }
      case OKX_ROW_ANGLE:
      case OKX_ROW_THREE_POINT_ANGLE: {  // constraints.py:223-243,287-308; jacobians.py:55-188
        S3 v1, v2;
        if (type == OKX_ROW_ANGLE) {
          v1 = sub(pt(pts[1]), pt(pts[0]));
          v2 = sub(pt(pts[3]), pt(pts[2]));
        } else {
          v1 = sub(pt(pts[0]), pt(pts[1]));
          v2 = sub(pt(pts[2]), pt(pts[1]));
        }
        const S3 cv = cross(v1, v2);
        const std::string t15 = dot(cv, cv, "EPS_SQ");
        const std::string dt = dot(v1, v2);
        std::string s, is;
        sqrt_rsqrt(t15, &s, &is);
        const std::string inv = emit("iv", "fast_rcp(fma(" + dt + ", " + dt + ", " + t15 + "))");
        const std::string ka = emit("ka", dt + " * " + inv + " * " + is), kb = emit("kb", s + " * " + inv);
        const S3 w1 = cross(v2, cv), w2 = cross(cv, v1);
        S3 g1, g2;
        for (int k = 0; k < 3; ++k) {
          g1.c[k] = emit("g", "fma(" + ka + ", " + w1.c[k] + ", -(" + kb + " * " + v2.c[k] + "))");
          g2.c[k] = emit("g", "fma(" + ka + ", " + w2.c[k] + ", -(" + kb + " * " + v1.c[k] + "))");
        }
        f("    const double %s = lean_atan2_pos(%s - EPS, %s) - %s;", r.c_str(), s.c_str(), dt.c_str(), rp(i, 0).c_str());
        if (type == OKX_ROW_ANGLE) {
          ro->partial.push_back({pts[0], neg(g1)});
          ro->partial.push_back({pts[1], g1});
          ro->partial.push_back({pts[2], neg(g2)});
          ro->partial.push_back({pts[3], g2});
        } else {
          S3 gm;
          for (int k = 0; k < 3; ++k) gm.c[k] = emit("g", "-" + g1.c[k] + " - " + g2.c[k]);
          ro->partial.push_back({pts[0], g1});
          ro->partial.push_back({pts[1], gm});
          ro->partial.push_back({pts[2], g2});
        }
        return true;
      }
      case OKX_ROW_VECTORS_PARALLEL:
      case OKX_ROW_VECTORS_PERPENDICULAR: {  // constraints.py:351-371,414-429; jacobians.py:192-318
        const S3 v1 = sub(pt(pts[1]), pt(pts[0])), v2 = sub(pt(pts[3]), pt(pts[2]));
        const std::string n1 = dot(v1, v1, "EPS_SQ"), n2 = dot(v2, v2, "EPS_SQ");
        const std::string s1 = emit("s", "sqrt(" + n1 + ")"), s2 = emit("s", "sqrt(" + n2 + ")");
        S3 g1, g2;
        if (type == OKX_ROW_VECTORS_PARALLEL) {
          const S3 cv = cross(v1, v2);
          const std::string c2 = dot(cv, cv, "EPS_SQ");
          const std::string sc = emit("s", "sqrt(" + c2 + ")");
          const S3 w1 = cross(v2, cv), w2 = cross(cv, v1);
          const std::string k26 = emit("k", "1.0 / (" + s1 + " * " + s2 + " * " + sc + ")");
          const std::string k19 = emit("k", sc + " / (" + s2 + " * " + s1 + " * " + s1 + " * " + s1 + ")");
          const std::string k31 = emit("k", sc + " / (" + s1 + " * " + s2 + " * " + s2 + " * " + s2 + ")");
          for (int k = 0; k < 3; ++k) {
            g1.c[k] = emit("g", k26 + " * " + w1.c[k] + " - " + k19 + " * " + v1.c[k]);
            g2.c[k] = emit("g", k26 + " * " + w2.c[k] + " - " + k31 + " * " + v2.c[k]);
          }
          f("    const double %s = (%s - EPS) / ((%s - EPS) * (%s - EPS));", r.c_str(), sc.c_str(), s1.c_str(), s2.c_str());
        } else {
          const std::string dt = dot(v1, v2);
          const std::string k16 = emit("k", "1.0 / (" + s1 + " * " + s2 + ")");
          const std::string k18 = emit("k", dt + " / (" + s2 + " * " + s1 + " * " + s1 + " * " + s1 + ")");
          const std::string k19 = emit("k", dt + " / (" + s1 + " * " + s2 + " * " + s2 + " * " + s2 + ")");
          for (int k = 0; k < 3; ++k) {
            g1.c[k] = emit("g", k16 + " * " + v2.c[k] + " - " + k18 + " * " + v1.c[k]);
            g2.c[k] = emit("g", k16 + " * " + v1.c[k] + " - " + k19 + " * " + v2.c[k]);
          }
          f("    const double %s = %s / ((%s - EPS) * (%s - EPS));", r.c_str(), dt.c_str(), s1.c_str(), s2.c_str());
        }
        ro->partial.push_back({pts[0], neg(g1)});
        ro->partial.push_back({pts[1], g1});
        ro->partial.push_back({pts[2], neg(g2)});
        ro->partial.push_back({pts[3], g2});
        return true;
      }
      case OKX_ROW_EQUAL_DISTANCE: {  // constraints.py:466-477; jacobians.py:322-367
        const S3 d1 = sub(pt(pts[1]), pt(pts[0])), d2 = sub(pt(pts[3]), pt(pts[2]));
        const std::string s1 = dot(d1, d1, "EPS_SQ"), s2 = dot(d2, d2, "EPS_SQ");
        std::string r1, i1, r2, i2;
        sqrt_rsqrt(s1, &r1, &i1);
        sqrt_rsqrt(s2, &r2, &i2);
        const S3 g1 = scale(i1, d1), g2 = scale(i2, d2);
        f("    const double %s = (%s - EPS) - (%s - EPS);", r.c_str(), r1.c_str(), r2.c_str());
        ro->partial.push_back({pts[0], neg(g1)});
        ro->partial.push_back({pts[1], g1});
        ro->partial.push_back({pts[2], g2});
        ro->partial.push_back({pts[3], neg(g2)});
        return true;
      }
      case OKX_ROW_FIXED_AXIS: {  // constraints.py:508-516; solver.py:407-416
        const int ax = (int)P.row_param[i][0];
        f("    const double %s = %s - %s;", r.c_str(), pt(pts[0]).c[ax].c_str(), rp(i, 1).c_str());
        ro->partial.push_back({pts[0], unit(ax)});
        return true;
      }
      case OKX_ROW_POINT_ON_LINE:
      case OKX_ROW_LINE_PIN: {  // constraints.py:560-576; jacobians.py:372-403; okx.h (pin)
        const S3 lp = rpv(i, 0), ld = rpv(i, 3);
        S3 cv;
        const int leader = pin_leader(i);
        if (type == OKX_ROW_LINE_PIN && pin_cross_.count(leader)) {
          cv = pin_cross_[leader];
        } else {
          const S3 w = sub(pt(pts[0]), lp);
          cv = cross(w, ld);
          if (type == OKX_ROW_LINE_PIN) pin_cross_[leader] = cv;
        }
        if (type == OKX_ROW_POINT_ON_LINE) {
          const std::string c2 = dot(cv, cv, "EPS_SQ");
          std::string root, inv;
          sqrt_rsqrt(c2, &root, &inv);
          const S3 gx = cross(ld, cv);
          const S3 g = scale(inv, gx);
          f("    const double %s = %s - EPS;", r.c_str(), root.c_str());
          ro->partial.push_back({pts[0], g});
          return true;
        }
        const int comp = (int)P.row_param[i][6];
        // r = e_comp . (w x ld) = w . (ld x e_comp)
        const S3 g = cross(ld, unit(comp));
        f("    const double %s = %s;", r.c_str(), cv.c[comp].c_str());
        ro->partial.push_back({pts[0], g});
        if (comp == 0) {
          const std::string c2 = dot(cv, cv, "EPS_SQ");
          ro->absres = "fabs(sqrt(" + c2 + ") - EPS)";
        } else {
          ro->absres.clear();
        }
        return true;
      }
      case OKX_ROW_POINT_ON_PLANE: {  // constraints.py:616-627; solver.py:429-437
        const S3 pp = rpv(i, 0), nn = rpv(i, 3);
        const S3 w = sub(pt(pts[0]), pp);
        const std::string d = dot(w, nn);
        f("    const double %s = %s;", r.c_str(), d.c_str());
        ro->partial.push_back({pts[0], nn});
        return true;
      }
      case OKX_ROW_MIDPOINT_ON_PLANE: {  // constraints.py:657-666; solver.py:439-448
        const S3 pp = rpv(i, 0), nn = rpv(i, 3);
        const S3 a = pt(pts[0]), b = pt(pts[1]);
        S3 mid;
        for (int k = 0; k < 3; ++k) mid.c[k] = emit("m", "fma(" + b.c[k] + " - " + a.c[k] + ", 0.5, " + a.c[k] + ")");
        const S3 w = sub(mid, pp);
        const std::string d = dot(w, nn);
        const S3 h = scale("0.5", nn);
        f("    const double %s = %s;", r.c_str(), d.c_str());
        ro->partial.push_back({pts[0], h});
        ro->partial.push_back({pts[1], h});
        return true;
      }
      case OKX_ROW_COPLANAR:
      case OKX_ROW_SCALAR_TRIPLE: {  // constraints.py:698-709,731-733; jacobians.py:426-483
        const S3 v1 = sub(pt(pts[1]), pt(pts[0])), v2 = sub(pt(pts[2]), pt(pts[0])), v3 = sub(pt(pts[3]), pt(pts[0]));
        const S3 c23 = cross(v2, v3), c31 = cross(v3, v1), c12 = cross(v1, v2);
        const std::string vol = dot(v1, c23);
        S3 g1 = c23, g2 = c31, g3 = c12;
        if (type == OKX_ROW_SCALAR_TRIPLE) {
          const std::string isc = emit("is", "fast_rcp(" + rp(i, 1) + ")");
          g1 = scale(isc, c23), g2 = scale(isc, c31), g3 = scale(isc, c12);
          f("    const double %s = (%s - %s) * %s;", r.c_str(), vol.c_str(), rp(i, 0).c_str(), isc.c_str());
        } else {
          f("    const double %s = %s;", r.c_str(), vol.c_str());
        }
        S3 g0;
        for (int k = 0; k < 3; ++k) g0.c[k] = emit("g", "-(" + g1.c[k] + " + " + g2.c[k] + " + " + g3.c[k] + ")");
        ro->partial.push_back({pts[0], g0});
        ro->partial.push_back({pts[1], g1});
        ro->partial.push_back({pts[2], g2});
        ro->partial.push_back({pts[3], g3});
        return true;
      }
      case kRowTarget: {  // solver.py:264-270, :560-579
        const S3 dir = rpv(i, 0);
        const std::string d = dot(pt(pts[0]), dir);
        f("    const double %s = %s - tv%d;", r.c_str(), d.c_str(), target_of_row(i));
        ro->partial.push_back({pts[0], dir});
        return true;
      }
      default:
        why = "row type " + std::to_string(type) + " has no lane code path yet";
        return false;
    }
  }

  static std::string A(int i, int j) { return "A" + std::to_string(i) + "_" + std::to_string(j); }
  static std::string L(int i, int j) { return "L" + std::to_string(i) + "_" + std::to_string(j); }
  static std::string gn(int i) { return "gn" + std::to_string(i); }

  bool emit_rows_residual_only() {
    f("    double ss = 0.0, mres_new = 0.0;");
    for (int i = 0; i < P.m; ++i) {
      RowOut ro;
      if (!row(i, &ro)) return false;
      f("    ss = fma(%s, %s, ss);", ro.r.c_str(), ro.r.c_str());
      if (!ro.absres.empty()) f("    mres_new = fmax(mres_new, %s);", ro.absres.c_str());
    }
    return true;
  }

  std::map<int, std::vector<std::pair<int, S3>>> target_j;  // target index -> (free block, d r / d block)
  // J^T J is NOT accumulated where the rows are: the lower triangle (135 doubles for the double wishbone) plus the
  // state would need ~540 registers.  Every row keeps its gradient (3 doubles for a distance row) and the products
  // of entry (i, j) are recorded here; emit_factor assembles column j just before it eliminates it, so that what is
  // live is the finished factor columns + the Schur complement in progress + the gradients still to be consumed.
  // Only the diagonal is accumulated at the rows (the first damping of a solve is scaled by its largest entry).
  bool pin_acc = true;  // opaque use of the touched accumulators after each row (keeps the contributions at the row)
  // Homes of the rows' gradients between the rows and the factorisation.  A gradient component is written once where its
  // row is and read back once per block column that multiplies it: up to j_lds_slots of them in LDS slots
  // [j_lds_base + k][lane] (the independent-solve body has 40-odd slots to spare), the rest stay in registers.
  int j_lds_base = 0, j_lds_slots = 0;
  std::map<std::string, int> j_home;  // gradient component name -> LDS slot
  void home_gradients(const std::vector<std::pair<int, S3>>& jv) {
    for (auto& fv : jv)
      for (int a = 0; a < 3; ++a) {
        const std::string& nm = fv.second.c[a];
        if (nm.empty() || nm == "1.0" || j_home.count(nm) || nm.compare(0, 2, "hs") == 0) continue;  // (constants have a home already)
        if ((int)j_home.size() >= j_lds_slots) return;
        const int slot = j_lds_base + (int)j_home.size();
        j_home[nm] = slot;
        f("    lds[%d + lane] = %s;", 64 * slot, nm.c_str());
      }
  }
  struct Prod { std::string a, b; int sg; };
  std::map<std::pair<int, int>, std::vector<Prod>> ata_terms;  // (i, j), i > j
  // late_diag: the diagonal is assembled at its column too (18 doubles less between the rows and the factorisation); the
  // largest diagonal entry, which scales the damping of a solve that starts without a first-step table, is then formed
  // from the same products in the (rare) branch that needs it.
  bool late_diag = false;
  std::map<int, std::vector<std::string>> diag_terms;  // i -> gradient components whose squares make (J^T J)_ii
  bool early_ata = false;  // parity kernel: additionally accumulate the whole lower triangle as E{i}_{j} at the rows
  std::set<std::string> early_declared;
  static std::string E(int i, int j) { return "E" + std::to_string(i) + "_" + std::to_string(j); }

  // Rows: r_i, cost, max |r|, gradient gn{i} = (J^T r)_i, diagonal A{i}_{i} of J^T J (elimination order).
  bool emit_rows() {
    const int n = 3 * P.n_free;
    for (int i = 0; i < n; ++i) {
      if (late_diag) f("    double %s = 0.0;", gn(i).c_str());
      else f("    double %s = 0.0, %s = 0.0;", gn(i).c_str(), A(i, i).c_str());
      nz[i][i] = true;
    }
    f("    double ss = 0.0, mres_new = 0.0;");
    for (int i = 0; i < P.m; ++i) {
      RowOut ro;
      if (!row(i, &ro)) return false;
      f("    ss = fma(%s, %s, ss);", ro.r.c_str(), ro.r.c_str());
      if (!ro.absres.empty()) f("    mres_new = fmax(mres_new, %s);", ro.absres.c_str());
      // point partials -> free blocks (chain rule through derived points: point_partial @ block, solver.py:554-558)
      std::map<int, std::vector<S3>> terms;
      for (auto& pp : ro.partial) {
        const int p = pp.first;
        const S3& g = pp.second;
        if (blk_of_point[p] >= 0) {
          terms[blk_of_point[p]].push_back(g);
        } else if (dop_of_point[p] >= 0) {
          auto it = dblk.find(dop_of_point[p]);
          if (it == dblk.end()) {
            why = "row reads a derived point without chain blocks";
            return false;
          }
          for (auto& kv : it->second) {
            const LBlk& b = kv.second;
            S3 t;
            t.sg = g.sg;
            for (int k = 0; k < 3; ++k) {
              if (b.scaled) {
                if (!g.c[k].empty()) t.c[k] = emit("j", "(" + b.s + ") * " + g.c[k]);
              } else {
                t.c[k] = emit("j", sum_expr({{g.c[0], b.m[0][k]}, {g.c[1], b.m[1][k]}, {g.c[2], b.m[2][k]}}));
              }
            }
            terms[kv.first].push_back(t);
          }
        }
      }
      std::vector<std::pair<int, S3>> jv;
      for (auto& kv : terms) {
        if (kv.second.size() == 1) {
          jv.push_back({kv.first, kv.second[0]});
        } else {
          S3 t;
          for (int k = 0; k < 3; ++k) {
            std::string e;
            for (auto& s : kv.second) {
              if (s.c[k].empty()) continue;
              if (!e.empty()) e += s.sg < 0 ? " - " : " + ";
              else if (s.sg < 0) e += "-";
              e += s.c[k];
            }
            t.c[k] = emit("j", e);
          }
          jv.push_back({kv.first, t});
        }
      }
      if (P.row_type[i] == kRowTarget) target_j[target_of_row(i)] = jv;
      std::vector<std::string> touched;
      for (auto& fv : jv)
        for (int a = 0; a < 3; ++a) {
          if (fv.second.c[a].empty()) continue;
          const int ia = 3 * fv.first + a;
          f("    %s = fma(%s, %s, %s);", gn(ia).c_str(), sgn(fv.second, a).c_str(), ro.r.c_str(), gn(ia).c_str());
          touched.push_back(gn(ia));
          if (late_diag) {
            diag_terms[ia].push_back(fv.second.c[a]);
          } else {
            f("    %s = fma(%s, %s, %s);", A(ia, ia).c_str(), fv.second.c[a].c_str(), fv.second.c[a].c_str(), A(ia, ia).c_str());
            touched.push_back(A(ia, ia));
          }
        }
      for (size_t ia = 0; ia < jv.size(); ++ia)
        for (size_t ib = 0; ib <= ia; ++ib) {
          const int F = jv[ia].first, G = jv[ib].first;
          const S3& jF = jv[ia].second;
          const S3& jG = jv[ib].second;
          const int sg = jF.sg * jG.sg;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < (F == G ? a : 3); ++b) {
              if (jF.c[a].empty() || jG.c[b].empty()) continue;
              const int i2 = 3 * F + a, j2 = 3 * G + b;
              nz[i2][j2] = true;
              ata_terms[{i2, j2}].push_back({jF.c[a], jG.c[b], sg});
              if (early_ata) {
                const std::string en = E(i2, j2);
                if (!early_declared.count(en)) {
                  early_declared.insert(en);
                  f("    double %s = %s%s * %s;", en.c_str(), sg < 0 ? "-" : "", jF.c[a].c_str(), jG.c[b].c_str());
                } else {
                  f("    %s = fma(%s%s, %s, %s);", en.c_str(), sg < 0 ? "-" : "", jF.c[a].c_str(), jG.c[b].c_str(), en.c_str());
                }
              }
            }
        }
      if (pin_acc)
        for (auto& an : touched) f("    asm volatile(\"\" : \"+v\"(%s));", an.c_str());
      home_gradients(jv);
    }
    return true;
  }

  // (J^T J)_ii as one expression of the rows' gradient components (late_diag)
  std::string diag_expr(int i, const std::function<std::string(const std::string&)>& ref) {
    std::vector<std::pair<std::string, std::string>> pr;
    for (const std::string& nm : diag_terms[i]) pr.push_back({ref(nm), ref(nm)});
    const std::string e = sum_expr(pr);
    return e.empty() ? "0.0" : e;
  }

  // Structure of the factor at scalar granularity (symbolic right-looking elimination).
  void symbolic() {
    const int n = 3 * P.n_free;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) fill[i][j] = j <= i && nz[i][j];
    for (int j = 0; j < n; ++j) {
      std::vector<int> rows;
      for (int i = j + 1; i < n; ++i)
        if (fill[i][j]) rows.push_back(i);
      for (size_t a = 0; a < rows.size(); ++a)
        for (size_t b = 0; b <= a; ++b) fill[rows[a]][rows[b]] = true;
    }
  }

  // LDL^T of (J^T J + lambda I) with the forward substitution of ONE right-hand side fused in, LEFT-LOOKING, column by
  // column.  Column j: scale row j (L_jk = C_jk / d_k, k < j), y_j = rhs_j - sum_k L_jk y_k, then every entry of the
  // column at once: C_ij = (J^T J)_ij [assembled here from the rows' gradients] - sum_k C_ik L_jk.  Nothing of the
  // trailing matrix exists before its column's turn: what is live is the window of unscaled entries C_ik with row > j
  // >= column, the gradients still to be consumed, y and the pivots.  A finished ROW j is cold until the backward
  // substitution reads it once (last row first): all but the last rows are parked in accumulation registers by hand
  // (v_accvgpr_write / _read through "a"-class operands).  Left to the compiler the register allocator shuffles the
  // whole factor through the AGPRs (measured: 912 moves for 1407 fp64 instructions in this block) and spills.
  // Leaves L{i}_{j} (parked or resident), dinv{j}, y{i}, ok, pmin, pmax.
  int col_fence = 3;     // scheduling barrier after every col_fence columns (keeps the late assembly late)
  bool launder = true;
  void fence() { f("    __builtin_amdgcn_sched_barrier(0);"); }
  int resident_rows = 1 << 20; // rows of the factor (counted from the last) that are NOT parked in accumulation registers by hand: all of them
  std::set<std::pair<int, int>> parked;
  // ... or in LDS, where the body has slots to spare: [l_lds_base + k][lane], first rows first (they wait longest)
  int l_lds_base = 0, l_lds_slots = 0;
  std::map<std::pair<int, int>, int> l_home;
  void emit_factor(const std::vector<std::string>& rhs) { emit_factor_multi({rhs}, {"y"}); }
  // ... with several right-hand sides: set s forward-substitutes into {yp[s]}{j} (the evaluated module's epilogue: one
  // per target, J^T e_t)
  void emit_factor_multi(const std::vector<std::vector<std::string>>& rhs_sets, const std::vector<std::string>& yp) {
    const int n = 3 * P.n_free;
    symbolic();
    parked.clear();
    l_home.clear();
    f("    // ---- damped normal equations: LDL^T + forward substitution, left-looking ----");
    fence();
    if (launder) {
      // every value the factorisation takes over from the rows is redefined here (an empty asm, no instruction): nothing
      // of the factorisation can be computed ahead of this point
      std::set<std::string> seen;
      for (auto& kv : ata_terms)
        for (const Prod& t : kv.second)
          for (const std::string* nm : {&t.a, &t.b})
            if (!j_home.count(*nm) && nm->compare(0, 1, "_") == 0 && seen.insert(*nm).second)
              f("    asm volatile(\"\" : \"+v\"(%s));", nm->c_str());
      for (int i = 0; i < n; ++i) {
        if (late_diag) f("    asm volatile(\"\" : \"+v\"(%s));", gn(i).c_str());
        else f("    asm volatile(\"\" : \"+v\"(%s), \"+v\"(%s));", gn(i).c_str(), A(i, i).c_str());
      }
    }
    f("    bool ok = true;");
    f("    double pmin = 1e300, pmax = 0.0;");
    std::map<std::string, std::string> alias;  // gradient component -> its reloaded copy of the current block column
    auto ref = [&](const std::string& nm) {
      auto it = alias.find(nm);
      return it == alias.end() ? nm : it->second;
    };
    for (int j = 0; j < n; ++j) {
      f("    // column %d", j);
      if (j % 3 == 0 && !j_home.empty()) {  // gradients of this block column's products come back from LDS
        alias.clear();
        for (int jj = j; jj < j + 3 && jj < n && late_diag; ++jj)
          for (const std::string& nm : diag_terms[jj]) {
            auto h = j_home.find(nm);
            if (h == j_home.end() || alias.count(nm)) continue;
            const std::string cp = nm + "_c" + std::to_string(j / 3);
            f("    const double %s = lds[%d + lane + kz];", cp.c_str(), 64 * h->second);
            alias[nm] = cp;
          }
        for (int jj = j; jj < j + 3 && jj < n; ++jj)
          for (int i = jj + 1; i < n; ++i) {
            auto it = ata_terms.find({i, jj});
            if (it == ata_terms.end()) continue;
            for (const Prod& t : it->second)
              for (const std::string* nm : {&t.a, &t.b}) {
                auto h = j_home.find(*nm);
                if (h == j_home.end() || alias.count(*nm)) continue;
                const std::string cp = *nm + "_c" + std::to_string(j / 3);
                f("    const double %s = lds[%d + lane + kz];", cp.c_str(), 64 * h->second);  // (opaque index: no store-to-load forwarding)
                alias[*nm] = cp;
              }
          }
      }
      std::vector<int> cols;  // k < j with L_jk structurally non-zero
      for (int k = 0; k < j; ++k)
        if (fill[j][k]) cols.push_back(k);
      for (int k : cols) f("    const double %s = C%d_%d * dinv%d;", L(j, k).c_str(), j, k, k);
      for (size_t si = 0; si < rhs_sets.size(); ++si) {
        std::vector<std::pair<std::string, std::string>> pr;
        for (int k : cols) pr.push_back({"(-" + L(j, k) + ")", yp[si] + std::to_string(k)});
        const std::string e = sum_expr(pr, rhs_sets[si][j] == "0.0" ? "" : rhs_sets[si][j]);
        f("    const double %s%d = %s;", yp[si].c_str(), j, e.empty() ? "0.0" : e.c_str());
      }
      for (int i = j; i < n; ++i) {
        if (!fill[i][j]) continue;
        // (J^T J)_ij from the gradients, then the earlier columns
        std::vector<std::pair<std::string, std::string>> pr;
        for (int k : cols)
          if (fill[i][k]) pr.push_back({"(-C" + std::to_string(i) + "_" + std::to_string(k) + ")", L(j, k)});
        std::string init;
        if (i == j) {
          if (late_diag) {
            std::vector<std::pair<std::string, std::string>> ap;
            for (const std::string& nm : diag_terms[j]) ap.push_back({ref(nm), ref(nm)});
            init = sum_expr(ap, "lambda");
          } else {
            init = A(j, j) + " + lambda";
          }
        } else {
          auto it = ata_terms.find({i, j});
          if (it != ata_terms.end()) {
            std::vector<std::pair<std::string, std::string>> ap;
            for (const Prod& t : it->second) ap.push_back({t.sg < 0 ? "(-" + ref(t.a) + ")" : ref(t.a), ref(t.b)});
            init = sum_expr(ap);
          }
        }
        const std::string e = sum_expr(pr, init);
        f("    const double C%d_%d = %s;", i, j, e.empty() ? "0.0" : e.c_str());
      }
      f("    ok = ok && C%d_%d > 0.0;", j, j);
      f("    pmin = fmin(pmin, C%d_%d); pmax = fmax(pmax, C%d_%d);", j, j, j, j);
      f("    const double dinv%d = pivot_rcp(C%d_%d);", j, j, j);
      for (int k : cols) {
        if ((int)l_home.size() >= l_lds_slots) break;
        const int slot = l_lds_base + (int)l_home.size();
        l_home[{j, k}] = slot;
        f("    lds[%d + lane] = %s;", 64 * slot, L(j, k).c_str());
      }
      if (j < n - resident_rows)
        for (int k : cols) {
          if (l_home.count({j, k})) continue;
          f("    int %s_lo, %s_hi; park(%s, %s_lo, %s_hi);", L(j, k).c_str(), L(j, k).c_str(), L(j, k).c_str(), L(j, k).c_str(), L(j, k).c_str());
          parked.insert({j, k});
        }
      if (col_fence > 0 && (j + 1) % col_fence == 0 && j + 1 < n) fence();
    }
    fence();
  }

  // D z = y, L^T x = z, row-oriented: x_j is final once every later row has been scattered; row j of L is read once.
  void emit_backward(const char* outn, const char* yp = "y") {
    const int n = 3 * P.n_free;
    f("    // ---- D z = y, L^T x = z (row by row, last row first) ----");
    for (int i = 0; i < n; ++i) f("    double %s%d = %s%d * dinv%d;", outn, i, yp, i, i);
    for (int j = n - 1; j >= 0; --j) {
      for (int k = 0; k < j; ++k) {
        if (!fill[j][k]) continue;
        if (l_home.count({j, k}))
          f("    %s%d = fma(-lds[%d + lane + kz], %s%d, %s%d);", outn, k, 64 * l_home[{j, k}], outn, j, outn, k);
        else if (parked.count({j, k}))
          f("    %s%d = fma(-unpark(%s_lo, %s_hi), %s%d, %s%d);", outn, k, L(j, k).c_str(), L(j, k).c_str(), outn, j, outn, k);
        else
          f("    %s%d = fma(-%s, %s%d, %s%d);", outn, k, L(j, k).c_str(), outn, j, outn, k);
      }
    }
  }
};

const char* kLanePreamble = R"SRC(
// Generated by okx_lanegen.cpp for one constraint program — do not edit.
typedef struct { double max_residual, cost, last_step; int iterations, nfev, flags, reserved; } okx_info;
// One wavefront per workgroup: its LDS instructions execute in program order, so what separates a lane's LDS writes from
// another lane's reads is an ordering for the COMPILER.  __syncthreads() is more than that: a workgroup-scope release
// fence, i.e. a wait for every global store the wavefront has in flight - the record stores of a wave unit (4 us of HBM
// time for 64 records) would be waited for before the next unit may start instead of draining behind it.
typedef const __attribute__((address_space(4))) double* okx_cptr;  // loads through it are scalar-cache loads (s_load)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
struct QArgs {
  const double* targets; const double* geom_pos; const double* geom_row_param;
  double* out_pos; okx_info* info;
  long long n_problems, steps_per_geometry, chain_len;
  int max_iter, confirm;
  double step_tol, grad_tol, ftol, lambda0, residual_tolerance;
  const double* design_pos; const double* row_param; const double* dop_param;
  double* trace; long long trace_problem;
  const double* predictor; long long predictor_mode; long long predictor_len;
  const double* head;
  long long out_mode;   // okx_solve_opts.output: 0 records of every output point, 1 the free points only, 2 nothing
};
#define EPS_SQ 1e-12
#define EPS 1e-6
#define DEV __device__ __forceinline__
#define INFO_CONVERGED 1
#define INFO_RESIDUAL_EXCEEDED 2
#define INFO_FAILED 4
#define INFO_ILL_CONDITIONED 8
#define ILL_CONDITIONED_PIVOT_RATIO 1e-12
// A wave-uniform double as a scalar-register value (every lane loaded the same address).
DEV double uni(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
DEV long long uni64(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}
DEV double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  e = fma(-x, r, 1.0);
  return fma(e, r, r);
}
// v_rcp_f64 / v_rsq_f64 deliver 2^-24.3 (measured on MI355X, profiles/r02/README.md): one Newton step for a pivot.
DEV double pivot_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
// a product / a sum that is rounded where it stands (never fused into a neighbour): the final state's squared norms are
// summed in the quad kernels' order with them
DEV double okx_mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
DEV double okx_add_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
// sqrt(x) to the last bit or so and 1 / sqrt(x) to 4e-15: one Goldschmidt step, then the residual correction.
DEV void fast_sqrt_rsqrt(double x, double* root, double* inv) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  *root = g;
  *inv = h + h;
}
// atan2(y, x), y >= 0, result in [0, pi]: fdlibm-style reduction + odd polynomial (see okx_kernels.hip).
DEV double lean_atan2_pos(double y, double x) {
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
DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// A double parked in two accumulation registers by hand (the factor's finished rows): never a candidate for the
// register allocator's own spilling, one write and one read per half.
DEV void park(double v, int& lo, int& hi) {
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(v)));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(v)));
}
DEV double unpark(int lo, int hi) {
  int l, h;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
  return __hiloint2double(h, l);
}
)SRC";

}  // namespace

int lane_variant_count() { return 24; }

namespace {
// Emission variants: the same arithmetic, differing only in hints to the compiler (opaque uses after each row, a
// redefinition of the factorisation's inputs at its top, where the scheduling barriers of the factorisation sit) - except
// `late_diag`, which assembles the diagonal of J^T J in another order (other rounding).
const struct { bool pin, launder, late_diag; int col_fence; } kVariants[12] = {
    {false, false, false, 3}, {true, true, false, 3}, {false, true, false, 3}, {true, false, false, 3},
    {false, false, false, 1}, {false, false, false, 6}, {false, false, true, 3}, {false, false, false, 0},
    {false, false, false, 2}, {false, false, true, 1}, {false, true, true, 3}, {true, false, true, 6}};
}  // namespace

// Variants 12 .. 23 are variants 0 .. 11 with a small program's state and factor back in LDS (lane_generate): where a value is
// kept does not enter the arithmetic.
bool lane_variants_same_arithmetic(int a, int b) {
  const int n = lane_variant_count();
  return a >= 0 && a < n && b >= 0 && b < n && kVariants[a % 12].late_diag == kVariants[b % 12].late_diag;
}

bool lane_chain_is_flat(int n_vars) {
  return 80 - 4 * n_vars < 16;  // fewer than 16 of the 80 LDS slots left for the factor's rows beside x, dx, xp, xq
}

bool lane_generate(const DevProgram& P, std::string* src, std::string* why, int variant, const EvalSpec* es) {
  // EV: the lane form of the evaluated module (okx_solve_evaluated_batch): the independent-solve bodies end every wave
  // unit with the tangent / metric epilogue (see the quad form in okx_quadgen.cpp; here a lane holds a whole problem, so the
  // catalog runs on duals with ALL the targets' directions at once, its role points straight from the lane's registers)
  const bool EV = es != nullptr;
  // Emission variants (kVariants above): the register allocator's result for an 18-unknown program sits at the edge of the
  // 512-register file and is not monotonic in any of the hints (0 ... 250 B of scratch across them for the double
  // wishbone, and not the same variant for every kernel of the module), so lane_build (okx_jit.cpp) compiles them in this
  // order, keeps the first one whose independent-solve kernels do not spill - or, after a full search, the one that spills
  // least, with single kernels taken from other variants of the same arithmetic.
  if (variant < 0 || variant >= lane_variant_count()) variant = 0;
  const auto& V = kVariants[variant % 12];
  // Programs of up to 15 variables (the MacPherson corner) have registers to spare: variants 0 .. 11 keep the accepted
  // point, the step in hand and the whole factor of their independent-solve body in registers, variants 12 .. 23 are the
  // same hints with the LDS layout every larger program has (a 15-variable program with one row of each class spills
  // 272 B in registers and nothing in LDS).  For larger programs the second dozen would repeat the first: not generated.
  const bool small_in_registers = 3 * P.n_free <= 15 && variant < 12;
  if (3 * P.n_free > 15 && variant >= 12) {
    *why = "variant " + std::to_string(variant) + " is variant " + std::to_string(variant - 12) + " for this program";
    return false;
  }
  if (P.n_free > kLaneMaxFree) {
    *why = "more than " + std::to_string(kLaneMaxFree) + " free points: the lower triangle of J^T J does not fit one lane's registers";
    return false;
  }
  if (P.n_targets > kMaxTargets || P.n_targets < 1) {
    *why = "needs 1.." + std::to_string(kMaxTargets) + " targets";
    return false;
  }
  const int nf = P.n_free, n = 3 * nf, NP = P.n_points, T = P.n_targets;
  // The pass (rows, factorisation, backward substitution) is generated once per body: the independent-solve body has
  // LDS slots to spare for the rows' gradients, the chain body has not.  `ev` is the chain body's generator (no LDS
  // homes) and the owner of the chain constants; `evc` the independent-solve body's.
  const int kColdStateSlots = 2 * n;                               // x, dx
  // first-step table layout (shared with okx_quadgen.cpp: quad_head_stride)
  const int HK = T + 1;
  const int head_off = 4 * nf * HK + 2 * HK * HK;
  const int head_s_off = head_off + 8;                                   // second-order vectors S_st, [pair][F][4]
  const int head_stride = head_s_off + 4 * nf * (HK - 1) * HK / 2;
  // The geometry's tables are read through the scalar cache where they are used (see the chain constants above); the
  // developer switch lane_lds_tables brings back round 3's staging into LDS (and takes its share of the 40 KiB).
  const bool scalar_tables = !dev_switch("lane_lds_tables");
  const int all_table_doubles = 3 * P.n_points + 8 * (P.n_crows + P.n_targets) + (P.n_derived > 0 ? P.n_derived : 1);
  const int table_doubles = scalar_tables ? 0 : all_table_doubles;
  int cold_j_slots = (40 * 1024 - 8 * table_doubles - 256) / 512 - kColdStateSlots;
  if (cold_j_slots < 0) cold_j_slots = 0;
  // Independent solves on PER-GEOMETRY tables get a body of their own: every wave unit (or every few) has another
  // geometry, the wavefronts of a CU read 3 KB each of different tables, and those reads miss the 16 KB scalar cache - a
  // round trip to L2 per batch of reads, in every pass (measured: the scalar tables gave the ensemble kernel 2.5 % where
  // 15 % of its time was staging).  There the tables - and the first-step table, in an area of its own - are staged into
  // LDS once per geometry, every load of the batch in flight together, and a wavefront takes a contiguous block of wave
  // units (developer switch lane_g_scalar: one body for both, as for the chains).
  const bool split_g = scalar_tables && !dev_switch("lane_g_scalar");
  int g_l_slots = (40 * 1024 - 8 * (all_table_doubles + head_stride) - 256) / 512 - kColdStateSlots;
  if (g_l_slots < 0) g_l_slots = 0;
  struct PassSrc { std::string eval, factor, subst; };
  auto make_pass = [&](LGen& gen, PassSrc* out) -> bool {
    // (measured on the double wishbone, scratch bytes of the independent-solve bodies _u / _g: pins + launder 0 / 188, pins
    //  only 160 / -, launder only 96 / 96, neither 0 / 0: the register allocator's result is not monotonic in anything)
    gen.pin_acc = V.pin;
    gen.launder = V.launder;
    gen.col_fence = V.col_fence;
    for (int e = 0; e < P.n_derived; ++e) gen.dp(e);
    gen.f("    // ---- active derived points with chain-rule blocks ----");
    for (int idx = 0; idx < P.n_active; ++idx)
      if (!gen.derived_op(P.active_op[idx], true)) return false;
    if (!gen.emit_rows()) return false;
    out->eval = gen.out;
    gen.out.clear();
    std::vector<std::string> rhs;
    for (int i = 0; i < n; ++i) rhs.push_back("-" + LGen::gn(i));
    gen.emit_factor(rhs);
    out->factor = gen.out;
    gen.out.clear();
    gen.emit_backward("nx");
    out->subst = gen.out;
    gen.out.clear();
    return true;
  };
  LGen ev(P);
  PassSrc pass_chain, pass_cold;
  if (!make_pass(ev, &pass_chain)) {
    *why = ev.why;
    return false;
  }
  LGen evc(P);
  evc.hoisted_names = ev.hoisted_names;
  // Measured on the double wishbone (scratch bytes of okx_lane_solve_u): the spare slots given to the factor's rows
  // 104 B, to the rows' gradients 432 B, hand-parked AGPR rows on top of either 250 - 1000 B (the allocator needs the
  // accumulation registers for its own spilling).  So: the factor's first rows in LDS, nothing parked by hand.
  // (a program of up to 15 variables - the MacPherson corner - holds its whole factor in registers: 369 of 512 without a
  //  parked row; C4 cold 0.105 -> 0.097 ms: what a parked row costs a lone wavefront is its round trip, not its instruction)
  int cold_l_slots = small_in_registers ? 0 : cold_j_slots;
  cold_j_slots -= cold_l_slots;
  if (small_in_registers) cold_j_slots = 0;  // (... nor a gradient)
  evc.j_lds_base = kColdStateSlots;
  evc.j_lds_slots = cold_j_slots;
  evc.l_lds_base = kColdStateSlots + cold_j_slots;
  evc.l_lds_slots = cold_l_slots;
  evc.late_diag = V.late_diag;
  if (!make_pass(evc, &pass_cold)) {
    *why = evc.why;
    return false;
  }
  LGen evg(P);
  PassSrc pass_g;
  if (split_g) {
    evg.hoisted_names = ev.hoisted_names;
    evg.j_lds_base = kColdStateSlots;
    evg.j_lds_slots = 0;
    evg.l_lds_base = kColdStateSlots;
    evg.l_lds_slots = small_in_registers ? 0 : g_l_slots;  // (a small program's factor stays in registers here too)
    evg.late_diag = V.late_diag;
    if (!make_pass(evg, &pass_g)) {
      *why = evg.why;
      return false;
    }
  }

  // evaluated module: the epilogue's pass - J at the solved state, the undamped LDL^T with one forward substitution per
  // target (right-hand side J^T e_t, the target row's gradient), one backward substitution each: tq{t}_{i} = d x_i / d target t
  struct EpiSrc { std::string eval, factor, subst; };
  auto make_epilogue_pass = [&](LGen& gen, const LGen& like, EpiSrc* out) -> bool {
    gen.uid = 700000;
    gen.hoisted_names = ev.hoisted_names;
    gen.j_lds_base = like.j_lds_base;
    gen.j_lds_slots = like.j_lds_slots;
    gen.l_lds_base = like.l_lds_base;
    gen.l_lds_slots = like.l_lds_slots;
    gen.late_diag = like.late_diag;
    gen.pin_acc = false;   // (J^T r is not wanted here: without the opaque uses the compiler drops its accumulation)
    gen.launder = false;
    gen.col_fence = V.col_fence;
    gen.f("    // ---- active derived points with chain-rule blocks ----");
    for (int idx = 0; idx < P.n_active; ++idx)
      if (!gen.derived_op(P.active_op[idx], true)) return false;
    if (!gen.emit_rows()) return false;
    out->eval = gen.out;
    gen.out.clear();
    std::vector<std::vector<std::string>> rhs_sets;
    std::vector<std::string> yp;
    for (int t = 0; t < T; ++t) {
      std::vector<std::string> rhs(n, "0.0");
      auto it = gen.target_j.find(t);
      if (it != gen.target_j.end())
        for (auto& fv : it->second)
          for (int a = 0; a < 3; ++a)
            if (!fv.second.c[a].empty()) rhs[3 * fv.first + a] = LGen::sgn(fv.second, a);
      rhs_sets.push_back(rhs);
      yp.push_back("yq" + std::to_string(t) + "_");
    }
    gen.emit_factor_multi(rhs_sets, yp);
    out->factor = gen.out;
    gen.out.clear();
    for (int t = 0; t < T; ++t) gen.emit_backward(("tq" + std::to_string(t) + "_").c_str(), yp[t].c_str());
    out->subst = gen.out;
    gen.out.clear();
    return true;
  };
  LGen epc(P), epg(P);
  EpiSrc epi_cold, epi_g;
  if (EV) {
    std::vector<int> oi(NP, -1);
    for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
    for (int F = 0; F < nf; ++F)
      if (oi[ev.fp(F)] < 0) {
        *why = "an evaluated module needs every free point among the output points";
        return false;
      }
    if (!make_epilogue_pass(epc, evc, &epi_cold) || (split_g && !make_epilogue_pass(epg, evg, &epi_g))) {
      *why = epc.why.empty() ? epg.why : epc.why;
      return false;
    }
  }

  // confirming evaluation (residuals only); not for programs with the reference's zero-gradient point-on-line row
  bool light_ok = true;
  for (int i = 0; i < P.n_crows; ++i) light_ok = light_ok && P.row_type[i] != OKX_ROW_POINT_ON_LINE;
  std::string light_src;
  if (light_ok) {
    LGen lt(P);
    lt.uid = 300000;
    lt.hoisted_names = ev.hoisted_names;
    for (int idx = 0; idx < P.n_active; ++idx)
      if (!lt.derived_op(P.active_op[idx], false)) light_ok = false;
    if (light_ok && !lt.emit_rows_residual_only()) light_ok = false;
    light_src = lt.out;
  }
  // final state: every derived point
  LGen fin(P);
  fin.quad_order = true;
  fin.uid = 100000;
  fin.hoisted_names = ev.hoisted_names;
  for (int e = 0; e < P.n_derived; ++e)
    if (!fin.derived_op(e, false)) {
      *why = fin.why;
      return false;
    }
  const std::string final_src = fin.out;

  std::vector<bool> used(NP, false);
  for (int k = 0; k < P.n_out; ++k) used[P.out_point[k]] = true;
  for (int i = 0; i < P.m; ++i)
    for (int s = 0; s < 4; ++s)
      if (P.row_pts[i][s] >= 0) used[P.row_pts[i][s]] = true;
  for (int e = 0; e < P.n_derived; ++e) {
    used[P.dop_out[e]] = true;
    for (int s = 0; s < 4; ++s)
      if (P.dop_pts[e][s] >= 0) used[P.dop_pts[e][s]] = true;
  }
  for (int k = 0; k < nf; ++k) used[P.free_point[k]] = true;
  auto is_fixed = [&](int p) { return ev.blk_of_point[p] < 0 && ev.dop_of_point[p] < 0; };

  // the fixed points are chain constants too (macros p{k}_{c} -> cl[...])
  for (int p = 0; p < NP; ++p)
    if (used[p] && is_fixed(p))
      for (int c = 0; c < 3; ++c) {
        char name[32];
        std::snprintf(name, sizeof(name), "p%d_%d", p, c);
        ev.add_const(name, ev.gl_gp0 + 3 * p + c);
      }
  const int gl_size = ev.gl_size;
  // coalesced staging of the geometry's tables (and, in the solve bodies, of its first-step table) into LDS
  auto stage_tables = [&](LGen& gg, const char* indent) {
    gg.f("%sfor (int k = lane; k < %d; k += 64) gl[%d + k] = gp[k];", indent, 3 * NP, ev.gl_gp0);
    gg.f("%sfor (int k = lane; k < %d; k += 64) gl[%d + k] = gq[k];", indent, 8 * P.n_crows, ev.gl_gq0);
    gg.f("%sif (lane < %d) gl[%d + lane] = a.row_param[%d + lane];", indent, 8 * T, ev.gl_tq0, 8 * P.n_crows);
    gg.f("%sif (lane < %d) gl[%d + lane] = a.dop_param[lane];", indent, P.n_derived, ev.gl_dp0);
  };
  // The same as ONE batch: every load first (lane k fetching entries k, k + 64, ... of each table, the index clamped so
  // that no load sits under a branch), then the LDS writes; the first-step table too (under `with_head`), to gl[head_into].
  // (a `for` with a load and an LDS store per trip is compiled to one memory round trip per trip: seven to HBM-resident
  //  tables were 13 900 cycles per wave unit of the ensemble kernel)
  struct Piece { const char* src; int src_off, count, dst; };
  auto stage_pieces = [&](LGen& gg, const char* indent, const std::vector<Piece>& pieces, int id) -> std::string {
    std::string stores;
    char line[160];
    for (const Piece& pc : pieces)
      for (int k0 = 0; k0 < pc.count; k0 += 64, ++id) {
        const int left = pc.count - k0;
        if (left >= 64) {
          gg.f("%sconst double sv%d = %s[%d + lane];", indent, id, pc.src, pc.src_off + k0);
          std::snprintf(line, sizeof(line), "%sgl[%d + lane] = sv%d;\n", indent, pc.dst + k0, id);
        } else {
          gg.f("%sconst double sv%d = %s[%d + (lane < %d ? lane : %d)];", indent, id, pc.src, pc.src_off + k0, left, left - 1);
          std::snprintf(line, sizeof(line), "%sif (lane < %d) gl[%d + lane] = sv%d;\n", indent, left, pc.dst + k0, id);
        }
        stores += line;
      }
    return stores;
  };
  auto stage_tables_batched = [&](LGen& gg, const char* indent, int head_into) {
    const std::string stores = stage_pieces(gg, indent, {{"gp", 0, 3 * NP, ev.gl_gp0}, {"gq", 0, 8 * P.n_crows, ev.gl_gq0},
                                                         {"a.row_param", 8 * P.n_crows, 8 * T, ev.gl_tq0}, {"a.dop_param", 0, P.n_derived, ev.gl_dp0}}, 0);
    gg.f("%sif (with_head) {", indent);
    const std::string deeper = std::string(indent) + "  ";
    gg.out += stage_pieces(gg, deeper.c_str(), {{"hsrc", 0, head_stride, head_into}}, 100);
    gg.f("%s}", indent);
    gg.out += stores;
  };
  const bool marks = dev_switch("lane_mark");  // `s_nop 11..16` between the sections of a pass (tools/lane_isa.sh)
  // developer build: every wave unit of the independent-solve body stamps the shader clock into a.trace[16 wu + k] - 0 unit
  // start, 1 tables staged, 16 state set up, 2 first step in hand, 3 passes done (4 / 5: full / confirming passes it ran,
  // 6 ... 11: cycles of its last full pass by section), 13 final state, 14 info stored, 17 records in LDS, 15 records
  // stored; 32 slots per wave unit (tools/lane_timeline.py)
  const bool timeline = dev_switch("lane_timeline");

  LGen g(P);
  g.out += kLanePreamble;
  g.f("");
  if (EV) {
    g.out += eval_metrics_source(*es);
    g.f("struct QEvArgs { QArgs q; double* tan; double* ev; EvCfg cfg; };");
  }
  {
    // every table entry a constant's name can stand for (unused macros cost nothing)
    std::set<std::string> defs;
    for (LGen* gen : {&ev, &evc, &evg, &epc, &epg}) defs.insert(gen->defines.begin(), gen->defines.end());
    char line[96];
    for (int i = 0; i < P.m; ++i)
      for (int k = 0; k < 8; ++k) {
        std::snprintf(line, sizeof(line), "#define hs%d_%d GL(%d)\n", i, k,
                      i < P.n_crows ? ev.gl_gq0 + 8 * i + k : ev.gl_tq0 + 8 * (i - P.n_crows) + k);
        defs.insert(line);
      }
    for (int e = 0; e < P.n_derived; ++e) {
      std::snprintf(line, sizeof(line), "#define hd%d GL(%d)\n", e, ev.gl_dp0 + e);
      defs.insert(line);
    }
    for (const std::string& d : defs) g.out += d;
  }
  auto PF = [&](int i) { return "p" + std::to_string(ev.fp(i / 3)) + "_" + std::to_string(i % 3); };
  const char* refresh_kz = "asm volatile(\"\" : \"+v\"(kz));";

  // One body per start mode.  COLD: every problem is an independent solve from the design state (chain_len 1): no
  // chain loop, no history, 36 LDS slots.  CHAIN: consecutive problems of a lane form a chain with secant / quadratic
  // extrapolation: the history xp / xq takes another 36 slots.
  std::string lds_why;
  const bool flat_chain = lane_chain_is_flat(n);
  // `ch`: the chain loop inside the body, history in LDS.  `fl` (flat chain, see okx_quad.hpp lane_chain_is_flat): the
  // independent-solve body walked over (wave unit, chain step) pairs, chain state in the launch's global scratch.
  LGen& evc_ = evc;
  const PassSrc& pass_cold_ = pass_cold;
  // `gb`: the independent-solve body of per-geometry launches (tables and first-step table staged in LDS, see split_g)
  // `ns` (with `fl`): the NESTED start mode - the warm-started form of a sweep for a kernel whose lanes run in lockstep.
  // A lane owns four consecutive steps of its span, as in a flat chain of four, and a wave unit therefore 256 consecutive
  // steps of one geometry; but the four are solved in the order 0, 2, 1, 3, and every one after the first starts from the
  // Lagrange interpolant (in the step index) of the up to six nearest steps the wave unit has already solved - its own and
  // its neighbour lanes', read from the launch's scratch [step of four][slot][lane].  Step 0 of every lane is a cold start
  // (first-step table); steps 2, 1 and 3 start ~1e-8 mm from their solutions instead of a secant's ~1e-4: one full pass
  // and one confirming evaluation.  What a chain carries beside the point - the damping it ended with, the contraction
  // constant it observed - comes from the lane's own step 0.
  // Coarse-to-fine start (developer switch lane_refine; okx_api.hip solve_impl): two more instantiations of the
  // independent-solve bodies with a compile-time STRIDE - SUB = 4: lane l of a wave unit solves step 4 l + offset of its
  // span (offset = a.chain_len, 0 .. 3) - one as it is (the coarse launch: every fourth step, cold from the first-step
  // table) and one WARM: the start is the cubic Lagrange interpolant, in the step index, of the four nearest coarse steps,
  // read from the OUTPUT buffer the coarse launch wrote; no first-step table, the first pass evaluates the guess.
  const bool refine = !EV && dev_switch("lane_refine");
  auto body = [&](bool ch, bool fl, bool gb, bool ns = false) -> bool {
    LGen& evc = gb ? evg : evc_;
    const PassSrc& pass_cold = gb ? pass_g : pass_cold_;
    int n_slots = 0;
    auto slot_ref = [&](const std::string& name) { return "double& " + name + " = lds[" + std::to_string(64 * n_slots++) + " + lane];"; };
    // A program of up to 15 variables (the MacPherson corner) keeps the accepted point and the step in hand in registers
    // too: 30 LDS slots less, and - what counts for a lone wavefront - their round trips out of the passes.  7 % more
    // instructions (a value parked in accumulation registers costs two moves, one in LDS one access) and C4 cold
    // 0.0973 -> 0.0889 ms; with the factor's rows out of LDS as well (below) 2.46e9 -> 2.85e9 solves/s.
    const bool reg_state = !ch && !fl && small_in_registers;  // (the chain body, its history on top, spills 56 B with it)
    std::string state_decl;
    for (int i = 0; i < n; ++i) {
      if (reg_state) state_decl += "    double x" + std::to_string(i) + ", dx" + std::to_string(i) + ";";
      else state_decl += "    " + slot_ref("x" + std::to_string(i)) + " " + slot_ref("dx" + std::to_string(i));
      if (ch) state_decl += " " + slot_ref("xp" + std::to_string(i)) + " " + slot_ref("xq" + std::to_string(i));
      state_decl += "\n";
    }
    const int state_doubles_before_l = 64 * n_slots;  // x, dx (and the chain history): live from the prologue on
    if (!ch) n_slots = evc.l_lds_base + evc.l_lds_slots > n_slots + (int)evc.j_home.size() ? evc.l_lds_base + evc.l_lds_slots : n_slots + (int)evc.j_home.size();  // + the rows' gradients / the factor's first rows
    const int state_doubles = 64 * n_slots;
    const int stage_doubles = (ch || fl) && !ns ? 0 : 64 * 3 * P.n_out;
    int lds_doubles = state_doubles > stage_doubles ? state_doubles : stage_doubles;
    if (EV && lds_doubles < 64 * ((3 * P.n_out) | 1)) lds_doubles = 64 * ((3 * P.n_out) | 1);  // (tangent rows [lane][record | 1]; result rows [lane][25])
    if (EV && lds_doubles < 64 * 25) lds_doubles = 64 * 25;
    // the first-step table of the wave unit's geometry is staged behind the state (the area the factor's rows are parked
    // in later): the prologue's 100-odd table reads are LDS broadcasts instead of same-address global loads
    const int head_l0 = state_doubles_before_l;
    const bool cold = !ch && !fl;
    const bool sc = scalar_tables && !gb;  // no tables in LDS at all: read where they are used through the scalar cache
    if (!sc && !gb && lds_doubles < head_l0 + head_stride) lds_doubles = head_l0 + head_stride;
    const int gl_doubles = sc ? 0 : gl_size + (gb ? head_stride : 0);
    const std::string refresh_s = sc ? std::string(refresh_kz) + " asm volatile(\"\" : \"+s\"(kzs));" : std::string(refresh_kz);
    const char* const refresh_kz = refresh_s.c_str();  // (shadows the LDS-only form: this body's passes refresh both opaque zeros)
    if ((lds_doubles + gl_doubles) * 8 > 40 * 1024) {
      lds_why = "per-wavefront LDS state exceeds 40 KiB";
      return false;
    }
    const bool tl = timeline && ((!ch && !fl) || ns);  // (nested mode: one row per unit-step, `it`)
    auto stamp = [&](int slot) {
      if (tl) g.f("    if (a.trace && lane == 0) a.trace[%s * 32 + %d] = (double)__builtin_readcyclecounter();", ns ? "it" : "wu", slot);
    };
    auto mark = [&](int k) {
      if (marks && !ch && !fl) g.f("    __builtin_amdgcn_sched_barrier(0); asm volatile(\"s_nop %d\"); __builtin_amdgcn_sched_barrier(0);", 10 + k);
      // (scheduling barriers on both sides: without them the compiler moves a section's arithmetic across the clock read)
      if (tl) g.f("    __builtin_amdgcn_sched_barrier(0); { const long long tl_now = __builtin_readcyclecounter(); tl_sec%d = (double)(tl_now - tl_at); tl_at = tl_now; } __builtin_amdgcn_sched_barrier(0);", k);
    };
    // FULL: the kernel that writes full records (okx_solve_opts.output = 0) is compiled on its own, exactly as it was
    // before the compact outputs existed: the register allocator's result for the double wishbone is that fragile
    // (the same body with the output mode as a run-time switch: 0 -> 248 B of scratch).
    g.f("#undef GL");
    if (sc) {
      // a table entry by its place in the (former) LDS image: positions, constraint-row parameters, target-row parameters,
      // derived-op parameters - the offset is a literal, the chain of conditions folds to one array
      g.f("#define GL(o) ((o) < %d ? gpc[(o) + kzs] : (o) < %d ? gqc[(o) - %d + kzs] : (o) < %d ? rpc[(o) - %d + kzs] : dpc[(o) - %d + kzs])",
          ev.gl_gq0, ev.gl_tq0, ev.gl_gq0, ev.gl_dp0, ev.gl_tq0 - 8 * P.n_crows, ev.gl_dp0);
    } else
      g.f("#define GL(o) gl[(o) + kz]");
    // (evaluated module: GIVEN = the wave unit's states are read from records - a.targets points at them - instead of solved;
    //  the body is then its final state and the epilogue: okx_evaluate_batch's lane form)
    const bool sub_body = refine && !ch && !fl;  // (the independent-solve bodies carry the stride / warm-start parameters)
    g.f("template <bool PG, bool FULL%s> DEV void okx_lane_body_%s(const QArgs& a%s) {", EV ? ", bool GIVEN" : sub_body ? ", int SUB, bool WARM" : "",
        ns ? (gb ? "nestg" : "nest") : ch || fl ? "chain" : gb ? "coldg" : "cold", EV ? ", const QEvArgs& ea" : "");
    if (!EV) g.f("  constexpr bool GIVEN = false;");
    if (!sub_body) g.f("  constexpr int SUB = 1; constexpr bool WARM = false;");
    g.f("  const int lane = threadIdx.x;");
    g.f("  __shared__ double lds[%d];", lds_doubles);
    if (sc) g.f("  int kzs = 0;  // an opaque zero in a scalar register: a table read inside a pass is a load of that pass, not a loop invariant");
    else
    g.f("  __shared__ double gl[%d];  // the wave unit's geometry tables: positions, row parameters, derived-op parameters%s", gl_doubles,
        gb ? ", first-step table" : "");
    g.f("  int kz = 0;");
    g.f("  const long long spg = a.steps_per_geometry;");
    g.f("  const long long span = spg > 0 ? spg : a.n_problems;");
    g.f("  const long long n_spans = spg > 0 ? a.n_problems / span : 1;");
    if (ch || fl) {
      g.f("  const long long unit_len = a.chain_len;");
      g.f("  const long long chains_per_span = (span + unit_len - 1) / unit_len;");
    } else {
      g.f("  const long long unit_len = 1;");
      g.f("  const long long sub_off = SUB == 1 ? 0 : a.chain_len;  // (strided bodies: the offset rides in the unused chain length)");
      g.f("  const long long chains_per_span = SUB == 1 ? span : (span - sub_off + SUB - 1) / SUB;");
    }
    g.f("  const long long waves_per_span = (chains_per_span + 63) / 64;");
    g.f("  const long long n_wave_units = n_spans * waves_per_span;");
    if (fl) {
      // every wavefront owns a contiguous block of wave units and walks (wave unit, step) pairs in order: the steps of a
      // lane's chain are consecutive iterations of ONE loop whose body carries nothing from one iteration to the next
      g.f("  const long long wu_per_wave = (n_wave_units + gridDim.x - 1) / gridDim.x;");
      g.f("  const long long wu_lo = blockIdx.x * wu_per_wave, wu_hi = wu_lo + wu_per_wave < n_wave_units ? wu_lo + wu_per_wave : n_wave_units;");
      g.f("  double* const ring = const_cast<double*>(a.predictor) + (long long)blockIdx.x * %lld + lane;  // [entry][slot][lane]", ns ? lane_nest_doubles(n) : lane_flat_chain_doubles(n));
      g.f("  long long staged_span = -1;");
      if (gb) g.f("  const bool with_head_all = a.head != nullptr && a.grad_tol == 0.0;");
      g.f("  for (long long it = wu_lo * unit_len; it < wu_hi * unit_len; ++it) {");
      g.f("    const long long wu = uni64(it / unit_len);");
      g.f("    const int step = (int)uni64(it - wu * unit_len);  // wave-uniform: every lane of the wave unit is at this step of its chain");
      if (ns) g.f("    const int sidx = step == 1 ? 2 : step == 2 ? 1 : step;  // which of its four steps a lane solves now: 0, 2, 1, 3");
      else g.f("    const int sidx = step;");
    } else if (cold) {
      // Own geometry: the tables are staged once per wavefront, the wave units dealt out round-robin (a unit's cost goes with
      // its place in the sweep: neighbours to different wavefronts).  Per-geometry tables: every wavefront takes a contiguous
      // block of wave units, so that the units of one geometry follow each other and its tables are staged once.
      g.f("  const long long wu_per_wave = (n_wave_units + gridDim.x - 1) / gridDim.x;");
      g.f("  const long long wu_lo = PG ? blockIdx.x * wu_per_wave : blockIdx.x, wu_step = PG ? 1 : gridDim.x;");
      g.f("  const long long wu_hi = PG ? (wu_lo + wu_per_wave < n_wave_units ? wu_lo + wu_per_wave : n_wave_units) : n_wave_units;");
      if (gb) {
        g.f("  const bool with_head = !GIVEN && !WARM && a.head != nullptr && a.grad_tol == 0.0;");
        g.f("  long long staged_span = -1;");
      }
      {  // (own geometry: round k gives wavefront w the unit k G + (w + 131 k) mod G, see okx_quadgen.cpp; C4 cold +1.3 %)
        g.f("  const unsigned wu_g = gridDim.x, wu_rot_step = 131u %% wu_g;");
        g.f("  unsigned wu_rot = blockIdx.x;");
        g.f("  for (long long wu0 = PG ? wu_lo : 0; wu0 < wu_hi; wu0 += wu_step, wu_rot = wu_rot + wu_rot_step >= wu_g ? wu_rot + wu_rot_step - wu_g : wu_rot + wu_rot_step) {");
        g.f("    const long long wu = PG ? wu0 : wu0 + wu_rot;");
        g.f("    if (wu >= wu_hi) continue;");
      }
    } else
    g.f("  for (long long wu = blockIdx.x; wu < n_wave_units; wu += gridDim.x) {");
    stamp(0);
    if (tl) g.f("    long long tl_at = 0; double tl_sec1 = 0.0, tl_sec2 = 0.0, tl_sec3 = 0.0, tl_sec4 = 0.0, tl_sec5 = 0.0, tl_sec6 = 0.0, tl_full = 0.0, tl_light = 0.0;");
    // (the 64-bit division runs on the vector ALU; its result goes to scalar registers so that every table address
    //  derived from it is scalar arithmetic, not a pair of vector registers kept alive through the solve)
    g.f("    const long long span_idx = uni64(n_spans > 1 ? wu / waves_per_span : 0);  // wave-uniform: one geometry per wave unit");
    g.f("    const long long wave_in_span = uni64(wu - span_idx * waves_per_span);");
    g.f("    long long chain_in_span = wave_in_span * 64 + lane;");
    g.f("    const bool have = chain_in_span < chains_per_span;");
    g.f("    if (!have) chain_in_span = chains_per_span - 1;");
    if (!ch && !fl) g.f("    const long long first_b = span_idx * span + (SUB == 1 ? chain_in_span : chain_in_span * SUB + sub_off);");
    else
    g.f("    const long long first_b = span_idx * span + chain_in_span * unit_len;");
    if (ch || fl) g.f("    const long long last_b = first_b + unit_len < (span_idx + 1) * span ? first_b + unit_len : (span_idx + 1) * span;");
    g.f("    const double* gp = PG ? a.geom_pos + span_idx * %d : a.design_pos;", 3 * NP);
    g.f("    const double* gq = PG ? a.geom_row_param + span_idx * %d : a.row_param;", 8 * P.n_crows);
    g.f("    (void)gq;");
    if (ch)
      for (int t = 0; t < T; ++t) g.f("    double tn%d = a.targets[first_b * %d + %d], tp%d = 0.0, tq%d = 0.0, tr%d = 0.0;", t, T, t, t, t, t);
    else if (fl) {
      g.f("    const bool valid = have && first_b + sidx < last_b;");
      g.f("    const long long bb = valid ? first_b + sidx : last_b - 1;");
      for (int t = 0; t < T; ++t) g.f("    const double tn%d = a.targets[bb * %d + %d];", t, T, t);
    } else
      for (int t = 0; t < T; ++t) g.f("    const double tn%d = GIVEN ? 0.0 : a.targets[first_b * %d + %d];", t, T, t);
    if (sc) {
      g.f("    const okx_cptr gpc = (okx_cptr)gp, gqc = (okx_cptr)gq, rpc = (okx_cptr)a.row_param, dpc = (okx_cptr)a.dop_param;");
      g.f("    (void)gqc; (void)rpc; (void)dpc;");
      g.f("    const bool with_head = !GIVEN && !WARM && a.head != nullptr && a.grad_tol == 0.0%s;", fl ? " && step == 0" : "");
      g.f("    WAVE_SYNC();  // (the previous wave unit's last LDS reads are done)");
    } else {
    g.f("    // the geometry's tables (and its first-step table) into LDS, lane k fetching entry k");
    g.f("    WAVE_SYNC();  // (the previous wave unit's last reads of these areas are done)");
    if (fl && gb) {  // (the nested mode's per-geometry kernels: tables and first-step table staged once per geometry, like coldg)
      g.f("    if (span_idx != staged_span) {  // wave-uniform: the tables stay while the geometry does");
      g.f("      const bool with_head = with_head_all;");
      g.f("      const double* hsrc = a.head + span_idx * %d;", head_stride);
      stage_tables_batched(g, "      ", gl_size);
      g.f("      staged_span = span_idx;");
      g.f("    }");
      g.f("    const bool with_head = with_head_all && step == 0;");
    } else if (fl) {
      g.f("    if (span_idx != staged_span) {  // wave-uniform: the tables stay while the geometry does");
      stage_tables(g, "      ");
      g.f("      staged_span = span_idx;");
      g.f("    }");
    } else if (gb) {
      g.f("    if (span_idx != staged_span) {  // wave-uniform: the tables stay while the geometry does");
      g.f("      const double* hsrc = a.head + span_idx * %d;", head_stride);
      stage_tables_batched(g, "      ", gl_size);
      g.f("      staged_span = span_idx;");
      g.f("    }");
    } else
    stage_tables(g, "    ");
    if (!gb) {
    g.f("    const bool with_head = !GIVEN && !WARM && a.head != nullptr && a.grad_tol == 0.0%s;", fl ? " && step == 0" : "");
    g.f("    if (with_head) {");
    g.f("      const double* hp = a.head + (PG ? span_idx * %d : 0);", head_stride);
    g.f("      for (int k = lane; k < %d; k += 64) lds[%d + k] = hp[k];", head_stride, head_l0);
    g.f("    }");
    }
    g.f("    WAVE_SYNC();");
    }
    stamp(1);
    g.f("    %s", refresh_kz);
    for (int p = 0; p < NP; ++p) {
      if (!used[p] || is_fixed(p)) continue;
      for (int c = 0; c < 3; ++c)
        if (sc || gb) g.f("    double p%d_%d = GL(%d);", p, c, ev.gl_gp0 + 3 * p + c);
        else g.f("    double p%d_%d = gp[%d];", p, c, 3 * p + c);
    }
    g.out += state_decl;
    for (int i = 0; i < n; ++i) {
      g.f("    x%d = %s; dx%d = 0.0;", i, PF(i).c_str(), i);
      if (ch) g.f("    xp%d = %s; xq%d = %s;", i, PF(i).c_str(), i, PF(i).c_str());
    }
    if (EV && !ch && !fl) {
      // given states: the free coordinates from the lane's record (18 loads in flight; the wave unit's records are one
      // contiguous block, so every line fetched is used by the unit)
      std::vector<int> oi(NP, -1);
      for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
      bool all_out = true;
      for (int F = 0; F < nf; ++F) all_out = all_out && oi[ev.fp(F)] >= 0;
      if (all_out) {
        g.f("    if (GIVEN) {");
        g.f("      const double* rec = a.targets + first_b * %d;  // (okx_lane_evaluate_*: a.targets points at the records)", 3 * P.n_out);
        for (int i = 0; i < n; ++i) g.f("      x%d = rec[%d];", i, 3 * oi[ev.fp(i / 3)] + i % 3);
        g.f("    }");
      } else {
        g.f("    static_assert(!GIVEN, \"a free point is not among the output points\");");
      }
    }
    if (sub_body) {
      // warm start: cubic Lagrange through the four nearest coarse steps (steps 4 m of this span, already in the output
      // buffer), the stencil shifted inwards at the ends of the span; loads unconditional, from clamped node indices
      std::vector<int> oi(NP, -1);
      for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
      g.f("    if (WARM) {");
      g.f("      const long long n_nodes = (span + SUB - 1) / SUB;");
      g.f("      long long nb0 = chain_in_span - 1; if (nb0 > n_nodes - 4) nb0 = n_nodes - 4; if (nb0 < 0) nb0 = 0;");
      g.f("      const double wu_ = (double)(chain_in_span - nb0) + (double)sub_off * %.17g;", 0.25);
      g.f("      const double lw0 = -(wu_ - 1.0) * (wu_ - 2.0) * (wu_ - 3.0) * %.17g, lw1 = wu_ * (wu_ - 2.0) * (wu_ - 3.0) * 0.5;", 1.0 / 6.0);
      g.f("      const double lw2 = -wu_ * (wu_ - 1.0) * (wu_ - 3.0) * 0.5, lw3 = wu_ * (wu_ - 1.0) * (wu_ - 2.0) * %.17g;", 1.0 / 6.0);
      g.f("      const int wrec = FULL ? %d : %d;", 3 * P.n_out, n);
      g.f("      const double* nd0 = a.out_pos + (span_idx * span + nb0 * SUB) * wrec;");
      g.f("      const double* nd1 = nd0 + SUB * wrec; const double* nd2 = nd1 + SUB * wrec; const double* nd3 = nd2 + SUB * wrec;");
      for (int i = 0; i < n; ++i) {
        const int full_idx = oi[ev.fp(i / 3)] >= 0 ? 3 * oi[ev.fp(i / 3)] + i % 3 : 0;
        const int free_idx = 3 * ev.perm[i / 3] + i % 3;
        g.f("      { const int wi = FULL ? %d : %d; x%d = fma(lw0, nd0[wi], fma(lw1, nd1[wi], fma(lw2, nd2[wi], lw3 * nd3[wi]))); }", full_idx, free_idx, i);
      }
      g.f("    }");
    }
    stamp(16);
    // the design state is a solved state of its own design targets: it seeds the chain's history
    for (int i = P.n_crows; i < P.m; ++i) {
      const int t = ev.target_of_row(i);
      const S3 dir = ev.rpv(i, 0);
      const S3 q = LGen::pt(P.row_pts[i][0]);
      g.f("    const double td%d = fma(%s, %s, fma(%s, %s, %s * %s));", t, q.c[0].c_str(), dir.c[0].c_str(), q.c[1].c_str(),
          dir.c[1].c_str(), q.c[2].c_str(), dir.c[2].c_str());
    }
    if (ch)
      for (int t = 0; t < T; ++t) g.f("    tp%d = td%d;", t, t);
    // shared first step of the unit's first problem, table of okx_quad_head_u/_g
    g.f("    bool head_ready = false;");
    g.f("    double hstep = 0.0, hN = 0.0, hM = 0.0, hss = 0.0, hmr = 0.0, hs0 = 0.0, hs1 = 0.0, hs4 = 0.0, hs5 = 0.0;");
    g.f("    if (with_head) {");
    if (sc) g.f("      const okx_cptr hp = (okx_cptr)(a.head + (PG ? span_idx * %d : 0)) + kzs;", head_stride);
    else if (gb) g.f("      const double* hp = gl + %d + kz;  // the staged table", gl_size);
    else g.f("      const double* hp = lds + %d + kz;  // the staged table", head_l0);
    g.f("      const double hr0 = 1.0;");
    for (int k = 1; k < HK; ++k) g.f("      const double hr%d = td%d - tn%d;", k, k - 1, k - 1);
    // first-order step d1 and, when the table carries them (scalar 6), the second-order correction
    // d2 = -1/2 sum_st w_s w_t S_st (okx_quadgen.cpp, okx_quad_head_*), taken while 2 |d2| <= 0.75 |d1|
    g.f("      double hst1 = 0.0, hst2 = 0.0;");
    for (int i = 0; i < n; ++i) {
      std::string e, e2;
      for (int k = 0; k < HK; ++k)
        e += (k ? " + hr" : "hr") + std::to_string(k) + " * hp[" + std::to_string(4 * (k * nf + i / 3) + i % 3) + "]";
      int pi = 0;
      for (int s2 = 1; s2 < HK; ++s2)
        for (int t2 = s2; t2 < HK; ++t2, ++pi)
          e2 += (pi ? " + " : "") + std::string(s2 == t2 ? "0.5" : "1.0") + " * hr" + std::to_string(s2) + " * hr" + std::to_string(t2) + " * hp[" +
                std::to_string(head_s_off + 4 * (pi * nf + i / 3) + i % 3) + "]";
      g.f("      const double hxa%d = -(%s), hxb%d = -(%s);", i, e.c_str(), i, e2.c_str());
      g.f("      hst1 = fmax(hst1, fabs(hxa%d)); hst2 = fmax(hst2, fabs(hxb%d));", i, i);
    }
    g.f("      const double hw2 = (hp[%d] > 0.5 && hst2 <= 0.375 * hst1) ? 1.0 : 0.0;", head_off + 6);
    for (int i = 0; i < n; ++i)
      g.f("      { const double hx = fma(hw2, hxb%d, hxa%d); dx%d = hx; hstep = fmax(hstep, fabs(hx)); hN = fma(hx, hx, hN); }", i, i, i);
    for (int j = 0; j < HK; ++j)
      for (int k = j; k < HK; ++k)
        g.f("      hM = fma(%shr%d * hr%d, hp[%d], hM);", j == k ? "" : "2.0 * ", j, k, head_off - 2 * HK * HK + j * HK + k);
    g.f("      hss = hp[%d]; hmr = hp[%d];", head_off + 2, head_off + 3);
    for (int k = 1; k < HK; ++k) g.f("      hss = fma(hr%d, hr%d, hss); hmr = fmax(hmr, fabs(hr%d));", k, k, k);
    g.f("      hs0 = hp[%d]; hs1 = hp[%d]; hs4 = hp[%d]; hs5 = hp[%d];", head_off, head_off + 1, head_off + 4, head_off + 5);
    g.f("      head_ready = true;");
    g.f("    }");
    if (ch) {
      g.f("    int hist = 1;");
      g.f("    double lambda_carry = 0.0;");
      g.f("    for (long long b = first_b; wave_any(have && b < last_b); ++b) {");
      g.f("      const bool valid = have && b < last_b;");
      g.f("      const long long bb = valid ? b : last_b - 1;");
      g.f("      const long long nb = b + 1 < last_b ? b + 1 : last_b - 1;");
      for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
      for (int t = 0; t < T; ++t) g.f("      tn%d = a.targets[nb * %d + %d];", t, T, t);
      // extrapolation along the chain: secant / quadratic through the last solved states
      g.f("      if (hist >= 2) {");
      g.f("        double num = 0.0, den = 0.0, nn = 0.0, num2 = 0.0, den2 = 0.0;");
      for (int t = 0; t < T; ++t) {
        g.f("        { const double dn = tv%d - tp%d, dold = tp%d - tq%d, dolder = tq%d - tr%d;", t, t, t, t, t, t);
        g.f("          num = fma(dn, dold, num); den = fma(dold, dold, den); nn = fma(dn, dn, nn);");
        g.f("          num2 = fma(dold, dolder, num2); den2 = fma(dolder, dolder, den2); }");
      }
      g.f("        double alpha = den > 0.0 ? num * fast_rcp(den) : 0.0;");
      g.f("        alpha = fmin(fmax(alpha, 0.0), 2.0);");
      g.f("        const double beta = den2 > 0.0 ? num2 * fast_rcp(den2) : 0.0;");
      g.f("        const bool line = hist >= 3 && alpha > 0.0 && beta >= 1e-3 && beta <= 2.0 && num * num >= 0.98 * nn * den && num2 * num2 >= 0.98 * den * den2;");
      g.f("        const double bq = line ? fast_rcp(beta) : 1.0;");
      g.f("        const double r1q = fast_rcp(1.0 + bq);");
      g.f("        const double l0 = line ? (alpha + 1.0) * (alpha + 1.0 + bq) * r1q : 1.0 + alpha;");
      g.f("        const double l1 = line ? -alpha * (alpha + 1.0 + bq) * beta : -alpha;");
      g.f("        const double l2 = line ? alpha * (alpha + 1.0) * r1q * beta : 0.0;");
      for (int i = 0; i < n; ++i)
        g.f("        { const double xo = x%d, xpo = xp%d; const double xn = fma(l0, xo, fma(l1, xpo, l2 * xq%d)); xq%d = xpo; xp%d = xo; x%d = xn; }", i, i, i, i, i, i);
      g.f("      } else {");
      for (int i = 0; i < n; ++i) g.f("        { const double xo = x%d; xq%d = xp%d; xp%d = xo; }", i, i, i, i);
      g.f("      }");
    } else if (fl && ns) {
      // Nested start: the Lagrange interpolant, in the step index, of the nearest steps this wave unit has solved (own and
      // neighbour lanes'), over whichever of eight candidates exist - four on either side inside the wave unit, so that the
      // lanes at its ends still have four or five on one side: a unit-step takes the passes of its slowest lane.  Entry
      // [step of four]: n coordinates, converged?, damping, contraction constant.  No candidate (a neighbourhood that
      // failed): the design state, like a chain that restarts.
      const int E = (n + 3) * 64, K = 8;
      struct Cand { int dl, q, o; };
      const Cand cand[3][K] = {
          {{-3, 0, -14}, {-2, 0, -10}, {-1, 0, -6}, {0, 0, -2}, {1, 0, 2}, {2, 0, 6}, {3, 0, 10}, {4, 0, 14}},          // second step (third of the four)
          {{-2, 0, -9}, {-2, 1, -7}, {-1, 0, -5}, {-1, 1, -3}, {0, 0, -1}, {0, 1, 1}, {1, 0, 3}, {1, 1, 5}},            // third step (second of the four)
          {{-1, 2, -6}, {-1, 1, -5}, {0, 0, -3}, {0, 2, -2}, {0, 1, -1}, {1, 0, 1}, {1, 2, 2}, {1, 1, 3}}};             // fourth step
      g.f("    {");
      for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
      g.f("      double lambda_carry = 0.0, cq_carry = 0.0;");
      g.f("      if (step > 0) {");
      // (entries stored by other lanes of THIS wavefront, through the same L1: its vector-memory instructions execute in
      //  order, so what separates the stores from these loads is an ordering for the compiler - an agent-scope release
      //  would wait for every record store in flight)
      g.f("        WAVE_SYNC();");
      // (opaque: derived from a loop invariant, every candidate address of every coordinate would be hoisted out of the
      //  unit-step loop, kept alive across the passes and spilled - a kilobyte of scratch, read back at memory latency)
      g.f("        const double* rb = ring - lane;  // this wavefront's entries [step][slot][lane]");
      g.f("        asm volatile(\"\" : \"+v\"(rb));");
      // Few round trips per unit-step: the lane's own first entry (damping, contraction constant), the candidates' flags and
      // their coordinates - half of the coordinates at a time - are loaded unconditionally (clamped lane indices: every
      // address is a valid entry) and pinned as a batch; flags, weights and selects afterwards.  (As written first - flags,
      // then per coordinate the loads under `v ? load : 0` - the compiler issued 20-odd dependent round trips to the
      // Infinity Cache: 14 us per unit-step, tools/lane_timeline.py c5nest.)
      g.f("        const double own_ok = rb[%d + lane], own_lambda = rb[%d + lane], own_cq = rb[%d + lane];", 64 * n, 64 * (n + 1), 64 * (n + 2));
      g.f("        bool v0 = false, v1 = false, v2 = false, v3 = false, v4 = false, v5 = false, v6 = false, v7 = false;");
      g.f("        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0, w5 = 0.0, w6 = 0.0, w7 = 0.0;");
      for (int q = 1; q <= 3; ++q) {
        g.f("        %sif (step == %d) {", q > 1 ? "else " : "", q);
        std::string fpin = "          asm volatile(\"\" : ";
        for (int j = 0; j < K; ++j) {
          const Cand& c = cand[q - 1][j];
          g.f("          const int lc%d = lane + (%d) < 0 ? 0 : (lane + (%d) > 63 ? 63 : lane + (%d));", j, c.dl, c.dl, c.dl);
          g.f("          const double* en%d = rb + %d + lc%d;", j, c.q * E, j);
          g.f("          double f%d = en%d[%d];", j, j, 64 * n);
          fpin += std::string(j ? ", " : "") + "\"+v\"(f" + std::to_string(j) + ")";
        }
        for (int half = 0; half < 2; ++half) {
          const int i0 = half * ((n + 1) / 2), i1 = half ? n : (n + 1) / 2;
          g.f("          __builtin_amdgcn_sched_barrier(0);");
          g.f("          {");
          for (int i = i0; i < i1; ++i)
            for (int j = 0; j < K; ++j) g.f("            double t%d_%d = en%d[%d];", j, i, j, 64 * i);
          if (half == 0) g.out += fpin + ");\n";
          for (int i = i0; i < i1; i += 3) {  // (an asm statement takes 30 operands)
            std::string pin = "            asm volatile(\"\" : ";
            bool first = true;
            for (int k = i; k < i + 3 && k < i1; ++k)
              for (int j = 0; j < K; ++j) {
                pin += std::string(first ? "" : ", ") + "\"+v\"(t" + std::to_string(j) + "_" + std::to_string(k) + ")";
                first = false;
              }
            g.out += pin + ");\n";
          }
          if (half == 0) {
            for (int j = 0; j < K; ++j) {
              const Cand& c = cand[q - 1][j];
              g.f("            v%d = lane + (%d) >= 0 && lane + (%d) < 64 && f%d > 0.5;", j, c.dl, c.dl, j);
            }
            for (int j = 0; j < K; ++j) {
              std::string w = "1.0";
              for (int k = 0; k < K; ++k) {
                if (k == j) continue;
                char buf[96];
                std::snprintf(buf, sizeof(buf), " * (v%d ? %.17g : 1.0)", k, (0.0 - cand[q - 1][k].o) / (double)(cand[q - 1][j].o - cand[q - 1][k].o));
                w += buf;
              }
              g.f("            w%d = v%d ? %s : 0.0;", j, j, w.c_str());
            }
          }
          g.f("            if (v0 || v1 || v2 || v3 || v4 || v5 || v6 || v7) {");
          for (int i = i0; i < i1; ++i) {
            std::string e;
            for (int j = 0; j < K; ++j) e += std::string(j ? " + " : "") + "w" + std::to_string(j) + " * (v" + std::to_string(j) + " ? t" + std::to_string(j) + "_" + std::to_string(i) + " : 0.0)";
            g.f("              x%d = %s;", i, e.c_str());
          }
          g.f("            }");
          g.f("          }");
          if (half == 0) continue;
        }
        g.f("        }");
      }
      g.f("        if (own_ok > 0.5) { lambda_carry = own_lambda; cq_carry = own_cq; }  // (the lane's own first step)");
      g.f("      }");
    } else if (fl) {
      // Start of a chain step: from the ring of this lane's chain - the last three steps' solutions, whether each
      // converged, the damping the last one ended with - exactly what the looping chain body keeps in x / xp / xq, hist
      // and lambda_carry (the design state stands in for solutions the chain does not have yet; a predecessor that did
      // not converge restarts the chain from the design state).
      const int E = (n + 2) * 64;
      g.f("    {");
      for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
      g.f("      double lambda_carry = 0.0;");
      g.f("      if (step > 0) {");
      g.f("        const double* r1 = ring + ((step + 2) %% 3) * %d;", E);
      g.f("        const double* r2 = ring + ((step + 1) %% 3) * %d;", E);
      g.f("        const double* r3 = ring + (step %% 3) * %d;", E);
      g.f("        const bool ok1 = r1[%d] > 0.5, ok2 = ok1 && step >= 2 && r2[%d] > 0.5, ok3 = ok2 && step >= 3 && r3[%d] > 0.5;", 64 * n, 64 * n, 64 * n);
      g.f("        if (ok1) {");
      g.f("          lambda_carry = r1[%d];", 64 * (n + 1));
      g.f("          const long long b2 = step >= 2 ? bb - 2 : bb, b3 = step >= 3 ? bb - 3 : bb;");
      for (int t = 0; t < T; ++t)
        g.f("          const double tp%d = a.targets[(bb - 1) * %d + %d], tq%d = ok2 ? a.targets[b2 * %d + %d] : td%d, tr%d = ok3 ? a.targets[b3 * %d + %d] : td%d;",
            t, T, t, t, T, t, t, t, T, t, t);
      g.f("          double num = 0.0, den = 0.0, nn = 0.0, num2 = 0.0, den2 = 0.0;");
      for (int t = 0; t < T; ++t) {
        g.f("          { const double dn = tv%d - tp%d, dold = tp%d - tq%d, dolder = tq%d - tr%d;", t, t, t, t, t, t);
        g.f("            num = fma(dn, dold, num); den = fma(dold, dold, den); nn = fma(dn, dn, nn);");
        g.f("            num2 = fma(dold, dolder, num2); den2 = fma(dolder, dolder, den2); }");
      }
      g.f("          double alpha = den > 0.0 ? num * fast_rcp(den) : 0.0;");
      g.f("          alpha = fmin(fmax(alpha, 0.0), 2.0);");
      g.f("          const double beta = den2 > 0.0 ? num2 * fast_rcp(den2) : 0.0;");
      g.f("          const bool line = ok2 && alpha > 0.0 && beta >= 1e-3 && beta <= 2.0 && num * num >= 0.98 * nn * den && num2 * num2 >= 0.98 * den * den2;");
      g.f("          const double bq = line ? fast_rcp(beta) : 1.0;");
      g.f("          const double r1q = fast_rcp(1.0 + bq);");
      g.f("          const double l0 = line ? (alpha + 1.0) * (alpha + 1.0 + bq) * r1q : 1.0 + alpha;");
      g.f("          const double l1 = line ? -alpha * (alpha + 1.0 + bq) * beta : -alpha;");
      g.f("          const double l2 = line ? alpha * (alpha + 1.0) * r1q * beta : 0.0;");
      for (int i = 0; i < n; ++i)
        g.f("          { const double xd = x%d, xo = r1[%d], xpo = ok2 ? r2[%d] : xd, xqo = ok3 ? r3[%d] : xd; x%d = fma(l0, xo, fma(l1, xpo, l2 * xqo)); }",
            i, 64 * i, 64 * i, 64 * i, i);
      g.f("        }");
      g.f("      }");
    } else {
      g.f("    {");
      g.f("      const bool valid = have;");
      g.f("      const long long bb = first_b;");
      for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
    }
    g.f("      double Fc = 0.0, lambda = 0.0, nu = 2.0, dmax = 0.0, step_len = 0.0, last_step = 0.0, mres = 0.0, pred = 0.0;");
    g.f("      int nfev = 0, iters = 0, flags = 0, nfail = 0;");
    g.f("      int mode = 0;  // 0 first evaluation, 1 trial point, 2 re-evaluation of the accepted point");
    g.f("      bool done = GIVEN || !valid, want_light = false;");
    g.f("      double prev_sl = 0.0, piv_lo = 0.0, piv_hi = 0.0;");
    if (ns) g.f("      double cq_seen = 0.0;");
    g.f("      if (head_ready%s) {", ch ? " && b == first_b" : "");
    g.f("        const bool at_design = valid%s && hs4 > 0.5;", ch ? " && hist == 1" : "");
    g.f("        if (at_design) {");
    g.f("          Fc = 0.5 * hss; mres = hmr; dmax = hs0; lambda = a.lambda0 * dmax;");
    g.f("          step_len = hstep; pred = 0.5 * fma(lambda, hN, hM); iters = 1; mode = 1;");
    g.f("          piv_lo = hs1 - lambda; piv_hi = hs5;");
    g.f("          if (hstep <= a.step_tol) { flags |= INFO_CONVERGED; last_step = hstep; done = true; }");
    g.f("          else {");
    g.f("            want_light = hstep <= 1e-3 && (100.0 * lambda * fast_rcp(hs1) + hstep) * hstep <= a.step_tol;");
    g.f("            prev_sl = hstep;");
    g.f("          }");
    g.f("        } else {");
    for (int i = 0; i < n; ++i) g.f("          dx%d = 0.0;", i);
    g.f("        }");
    g.f("      }%s", ch ? " else {" : "");
    if (ch) {
      for (int i = 0; i < n; ++i) g.f("        dx%d = 0.0;", i);
      g.f("      }");
    }
    // Two nested loops: the inner one runs full passes while any lane needs one; when every active lane only has a step to
    // confirm, the outer loop takes the residual-only pass and comes back (a lane whose step is not confirmed goes on with
    // full passes).  Same order of evaluations as one loop with the confirming pass as a branch at its top.
    const bool nested = light_ok;
    stamp(2);
    if (nested) {
      g.f("      while (wave_any(!done)) {");
      g.f("    %s", refresh_kz);
      g.f("    if (a.confirm == 0 && !wave_any(!done && !want_light)) {");
      if (tl) g.f("      tl_light += 1.0;");
      for (int i = 0; i < n; ++i) g.f("      %s = x%d + dx%d;", PF(i).c_str(), i, i);
      g.out += light_src;
      g.f("      const double Fl = 0.5 * ss;");
      g.f("      if (!done) {");
      g.f("        ++nfev;");
      g.f("        if (Fl == Fl && Fl <= Fc * (1.0 + 1e-6) + 1e-28) {");
      for (int i = 0; i < n; ++i) g.f("          x%d = %s;", i, PF(i).c_str());
      g.f("          Fc = Fl; mres = mres_new; last_step = step_len; flags |= INFO_CONVERGED; done = true;");
      g.f("        } else {");
      g.f("          want_light = false;");
      g.f("        }");
      g.f("      }");
      g.f("    }");
      g.f("    while (wave_any(!done) && (a.confirm != 0 || wave_any(!done && !want_light))) {");
      g.f("    %s", refresh_kz);
      g.f("    want_light = false;");
      if (tl) g.f("    __builtin_amdgcn_sched_barrier(0); tl_full += 1.0; tl_at = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);");
    } else {
    g.f("      while (wave_any(!done)) {");
    g.f("    %s", refresh_kz);
    if (light_ok) {
      g.f("    if (a.confirm == 0 && !wave_any(!done && !want_light)) {");
      for (int i = 0; i < n; ++i) g.f("      %s = x%d + dx%d;", PF(i).c_str(), i, i);
      g.out += light_src;
      g.f("      const double Fl = 0.5 * ss;");
      g.f("      if (!done) {");
      g.f("        ++nfev;");
      g.f("        if (Fl == Fl && Fl <= Fc * (1.0 + 1e-6) + 1e-28) {");
      for (int i = 0; i < n; ++i) g.f("          x%d = %s;", i, PF(i).c_str());
      g.f("          Fc = Fl; mres = mres_new; last_step = step_len; flags |= INFO_CONVERGED; done = true;");
      g.f("        } else {");
      g.f("          want_light = false;");
      g.f("        }");
      g.f("      }");
      g.f("      continue;");
      g.f("    }");
      g.f("    want_light = false;");
    }
    }
    mark(1);
    for (int i = 0; i < n; ++i) g.f("    %s = mode == 2 ? x%d : x%d + dx%d;", PF(i).c_str(), i, i, i);
    g.out += (ch ? pass_chain : pass_cold).eval;  // (the flat chain body runs the independent-solve body's pass)
    mark(2);
    g.f("    const double Ft = 0.5 * ss;");
    g.f("    bool accept = true, stop = false, compromise = false;");
    g.f("    double rho = 1.0;");
    g.f("    if (mode == 1) {");
    g.f("      const bool finite = Ft == Ft && step_len == step_len && Ft < 1e300;");
    g.f("      const bool small = finite && step_len <= 1e-8 && Ft <= Fc * (1.0 + 1e-6) + 1e-28;");
    g.f("      rho = (finite && pred > 0.0) ? (Fc - Ft) * fast_rcp(pred) : -1.0;");
    g.f("      accept = rho > 1e-4 || small;");
    g.f("      if (finite && step_len <= a.step_tol) { accept = small; stop = true; }");
    g.f("      else if (accept && finite && Fc - Ft <= a.ftol * Fc && pred <= a.ftol * Fc) { stop = true; compromise = true; }");
    g.f("    }");
    g.f("    double diag = 0.0, gm = 0.0;");
    g.f("    if (wave_any(mode == 0)) {");
    if (!ch && evc.late_diag) {
      auto same = [](const std::string& nm) { return nm; };
      for (int i = 0; i < n; ++i) g.f("      diag = fmax(diag, %s);", evc.diag_expr(i, same).c_str());
    } else {
      for (int i = 0; i < n; ++i) g.f("      diag = fmax(diag, %s);", LGen::A(i, i).c_str());
    }
    g.f("    }");
    // gradient stop (okx_solve_opts.grad_tol): > 0 the absolute form max |J^T r|; < 0 MINPACK's scaled form
    // max_j |(J^T r)_j| / (|J_j| |r|) (lmder's gnorm, what the reference's gtol means: solver.py:158-169)
    g.f("    if (a.grad_tol > 0.0) {");
    for (int i = 0; i < n; ++i) g.f("      gm = fmax(gm, fabs(%s));", LGen::gn(i).c_str());
    g.f("    } else if (a.grad_tol < 0.0) {");
    g.f("      const double rr = 2.0 * Ft;");
    for (int i = 0; i < n; ++i) {
      const std::string aii = (!ch && evc.late_diag) ? evc.diag_expr(i, [](const std::string& nm) { return nm; }) : LGen::A(i, i);
      g.f("      { const double cn = %s * rr; gm = fmax(gm, cn > 0.0 ? fabs(%s) * __builtin_amdgcn_rsq(cn) : 0.0); }", aii.c_str(), LGen::gn(i).c_str());
    }
    g.f("    }");
    g.f("    if (!done) {");
    g.f("      ++nfev;");
    g.f("      if (stop) flags |= INFO_CONVERGED;");
    // a solve that ends on the cost test without meeting its rows sits at a compromise point (okx_quadgen.cpp): advisory bit
    g.f("      if (compromise && mres_new > 0.01 * a.residual_tolerance) flags |= INFO_ILL_CONDITIONED;");
    g.f("      if (accept) {");
    g.f("        if (mode != 2) {");
    for (int i = 0; i < n; ++i) g.f("          x%d = %s;", i, PF(i).c_str());
    g.f("          if (mode == 1) last_step = step_len;");
    g.f("          nu = 2.0;");
    g.f("        }");
    g.f("        Fc = Ft; mres = mres_new;");
    g.f("        if (!stop) {");
    g.f("          if (mode == 0) {");
    g.f("            dmax = diag; lambda = a.lambda0 * dmax;");
    if (ch || fl) g.f("            if (lambda_carry > 0.0) lambda = fmin(lambda, lambda_carry);");
    g.f("          }");
    g.f("          else if (mode == 1 && rho > 1e-4) {");
    g.f("            const double t = 2.0 * rho - 1.0;");
    g.f("            lambda *= (rho > 0.99 && (step_len <= 1.0 || Ft <= 1e-2)) ? 1e-3 : (rho > 0.9 ? 0.1 : fmax(1.0 / 3.0, 1.0 - t * t * t));");
    g.f("          }");
    g.f("          if (a.grad_tol != 0.0 && gm <= fabs(a.grad_tol)) { flags |= INFO_CONVERGED; stop = true; }");
    g.f("        }");
    g.f("      } else if (!stop) {");
    g.f("        lambda *= nu; nu *= 2.0;");
    g.f("      }");
    g.f("      if (stop || iters >= a.max_iter) done = true;");
    g.f("    }");
    g.f("    const bool solve_now = !done && accept;");
    g.f("    if (!done && !accept) mode = 2;");
    g.f("    if (wave_any(solve_now)) {");
    mark(3);
    g.out += (ch ? pass_chain : pass_cold).factor;
    mark(4);
    g.out += (ch ? pass_chain : pass_cold).subst;
    mark(5);
    g.f("    double sl = 0.0, pr = 0.0, dd = 0.0;");
    for (int i = 0; i < n; ++i) g.f("    sl = fmax(sl, fabs(nx%d));", i);
    for (int i = 0; i < n; ++i) g.f("    pr = fma(nx%d, fma(lambda, nx%d, -%s), pr); dd = fma(nx%d, nx%d, dd);", i, i, LGen::gn(i).c_str(), i, i);
    g.f("    pr = 0.5 * pr;");
    // Rayleigh quotient of the step in J^T J + lambda I, less lambda: an upper bound of what the damping did not put on the
    // weakest direction (okx_quadgen.cpp); joins the pivots in the conditioning test
    g.f("    const double rq = dd > 0.0 ? (2.0 * pr - lambda * dd) * fast_rcp(dd) - lambda : 1e300;");
    g.f("    if (solve_now) {");
    g.f("      ++iters;");
    g.f("      if (ok) {");
    g.f("        piv_lo = fmin(pmin - lambda, rq); piv_hi = pmax;");
    for (int i = 0; i < n; ++i) g.f("        dx%d = nx%d;", i, i);
    g.f("        step_len = sl; pred = pr;");
    g.f("        if (sl <= a.step_tol) { flags |= INFO_CONVERGED; last_step = sl; done = true; }");
    g.f("        else {");
    if (ns) {
      // (a warm-started step has no earlier step of its own to read the quadratic contraction from: the constant its lane
      //  observed on the first of its four steps - same mechanism, same place on the solution manifold - stands in for
      //  the 1 / mm a cold problem's first step is given)
      g.f("          const double cq = prev_sl > 0.0 ? fmax(3.0 * sl * fast_rcp(prev_sl * prev_sl), 1e-3) : (cq_carry > 0.0 ? cq_carry : 1.0);");
      g.f("          if (prev_sl > 0.0) cq_seen = fmax(cq_seen, cq);");
    } else
    g.f("          const double cq = prev_sl > 0.0 ? fmax(3.0 * sl * fast_rcp(prev_sl * prev_sl), 1e-3) : 1.0;");
    g.f("          const double rho_lin = 100.0 * lambda * fast_rcp(pmin);");
    g.f("          want_light = sl <= 1e-3 && (rho_lin + cq * sl) * sl <= a.step_tol;");
    if (ns) {
      // Nested mode, interpolated starts: the start is ~1e-8 mm from the solution and this Gauss-Newton step lands within
      // a THOUSANDTH of step_tol of it by the same prediction that otherwise asks for a confirming evaluation.  With that
      // margin the step is taken as it is - one pass per step instead of a pass and a confirming pass (which costs a lone
      // wavefront most of a full pass: tools/lane_timeline.py c5nest); cost and max_residual in the info record are then
      // those of the point the step started from (~1e-8 off the rows' final values).  okx_solve_opts.confirm_full_pass
      // switches it off with the confirming evaluations.
      g.f("          if (step > 0 && a.confirm == 0 && sl <= 1e-5 && (rho_lin + cq * sl) * sl <= 1e-3 * a.step_tol) {");
      for (int i = 0; i < n; ++i) g.f("            x%d = x%d + nx%d;", i, i, i);
      g.f("            last_step = sl; flags |= INFO_CONVERGED; done = true; want_light = false;");
      g.f("          }");
    }
    g.f("          prev_sl = sl;");
    g.f("        }");
    g.f("        mode = 1;");
    g.f("      } else {");
    g.f("        lambda = fmax(lambda * 10.0, 1e-12 * dmax);");
    g.f("        if (++nfail > 60 || !(lambda < 1e30)) { flags |= INFO_FAILED; done = true; }");
    g.f("        mode = 2;");
    g.f("      }");
    g.f("    }");
    mark(6);
    g.f("    }  // any lane solves");
    if (nested) g.f("    }  // full passes");
    g.f("      }  // LM passes");
    stamp(3);
    if (tl) {
      g.f("    if (a.trace && lane == 0) { double* tr = a.trace + %s * 32; tr[4] = tl_full; tr[5] = tl_light; tr[6] = tl_sec1; tr[7] = tl_sec2;", ns ? "it" : "wu");
      g.f("      tr[8] = tl_sec3; tr[9] = tl_sec4; tr[10] = tl_sec5; tr[11] = tl_sec6; }");
    }
    // final state and output
    g.f("      {");
    g.f("    %s", refresh_kz);
    for (int i = 0; i < n; ++i) g.f("    %s = x%d;", PF(i).c_str(), i);
    if (EV) g.f("    {  // (the evaluated module needs every derived point: records or not)");
    else
    g.f("    if (FULL) {  // the derived points only matter to the full records");
    g.out += final_src;
    g.f("    }");
    g.f("    if (mres > a.residual_tolerance) flags |= INFO_RESIDUAL_EXCEEDED;");
    g.f("    if (piv_hi > 0.0 && piv_lo <= ILL_CONDITIONED_PIVOT_RATIO * piv_hi) flags |= INFO_ILL_CONDITIONED;");
    if (ch) {
      // chain bookkeeping
      for (int t = 0; t < T; ++t) g.f("    tr%d = tq%d; tq%d = tp%d; tp%d = tv%d;", t, t, t, t, t, t);
      g.f("    if (!(flags & INFO_CONVERGED) || (flags & INFO_FAILED)) {");
      for (int i = 0; i < n; ++i)
        if (sc) g.f("      x%d = GL(%d);", i, ev.gl_gp0 + 3 * ev.fp(i / 3) + i % 3);
        else g.f("      x%d = gp[%d];", i, 3 * ev.fp(i / 3) + i % 3);
      g.f("      hist = 1; lambda_carry = 0.0;");
      for (int t = 0; t < T; ++t) g.f("      tp%d = td%d;", t, t);
      g.f("    } else {");
      g.f("      if (hist < 3) ++hist;");
      g.f("      lambda_carry = lambda;");
      g.f("    }");
    }
    if (fl && ns) {
      g.f("    { double* r0 = ring + step * %d;  // this step's entry: solution, converged?, damping, contraction constant", (n + 3) * 64);
      g.f("      _Pragma(\"unroll 6\")");
      g.f("      for (int i = 0; i < %d; ++i) r0[64 * i] = lds[128 * i + lane];  // (x{i})", n);
      g.f("      r0[%d] = ((flags & INFO_CONVERGED) && !(flags & INFO_FAILED)) ? 1.0 : 0.0; r0[%d] = lambda; r0[%d] = cq_seen > 0.0 ? cq_seen : cq_carry;", 64 * n, 64 * (n + 1), 64 * (n + 2));
      g.f("      WAVE_SYNC(); }");
    } else if (fl) {
      g.f("    { double* r0 = ring + (step %% 3) * %d;  // this step's entry: solution, converged?, damping", (n + 2) * 64);
      for (int i = 0; i < n; ++i) g.f("      r0[%d] = x%d;", 64 * i, i);
      g.f("      r0[%d] = ((flags & INFO_CONVERGED) && !(flags & INFO_FAILED)) ? 1.0 : 0.0; r0[%d] = lambda; }", 64 * n, 64 * (n + 1));
    }
    stamp(13);
    g.f("    if (valid && !GIVEN) {");
    g.f("      okx_info inf; inf.max_residual = mres; inf.cost = Fc; inf.last_step = last_step;");
    g.f("      inf.iterations = iters; inf.nfev = nfev; inf.flags = flags; inf.reserved = 0;");
    g.f("      a.info[bb] = inf;");
    g.f("    }");
    stamp(14);
    if (!ch && !fl) {
      // Records of independent solves: the 64 problems of a wave unit are consecutive, their records one contiguous
      // block: transposed through LDS (which the state no longer needs) and written as full 16-byte-per-lane rows.
      // (okx_solve_opts.output: the full record, the free points alone in the program's free_point order, or nothing)
      if (EV) {  // (the evaluated module's kernels take the output mode at run time)
        g.f("      const bool full_rec = a.out_mode == 0;");
        g.f("      if (a.out_mode != 2) {");
        g.f("      const int rec = full_rec ? %d : %d;", 3 * P.n_out, n);
        g.f("      WAVE_SYNC();");
        g.f("      if (full_rec) {");
      } else {
      g.f("      if (FULL || a.out_mode == 1) {");
      g.f("      const int rec = FULL ? %d : %d;", 3 * P.n_out, n);
      g.f("      WAVE_SYNC();");
      g.f("      if (FULL) {");
      }
      g.f("      double* st = lds + lane * %d;", 3 * P.n_out);
      for (int k = 0; k < P.n_out; ++k)
        for (int c = 0; c < 3; ++c) g.f("      st[%d] = p%d_%d;", 3 * k + c, P.out_point[k], c);
      g.f("      } else {");
      g.f("      double* st = lds + lane * %d;", n);
      for (int i = 0; i < n; ++i) g.f("      st[%d] = %s;", 3 * ev.perm[i / 3] + i % 3, PF(i).c_str());
      g.f("      }");
      g.f("      WAVE_SYNC();");
      stamp(17);
      if (sub_body) {
        // strided bodies: the wave unit's records lie SUB problems apart - lane = column of a record, pointer bumps
        g.f("      if (SUB != 1) {");
        g.f("        const int per = 64 / rec > 0 ? 64 / rec : 1;");
        g.f("        const int sub = lane / rec, col = lane - sub * rec;");
        g.f("        const long long sb = span_idx * span + wave_in_span * 64 * SUB + sub_off, sl_ = (span_idx + 1) * span;");
        g.f("        const long long n_rows = (sl_ - sb + SUB - 1) / SUB;");
        g.f("        const int rows = (int)(n_rows < 64 ? (n_rows > 0 ? n_rows : 0) : 64);");
        g.f("        double* op = a.out_pos + (sb + (long long)sub * SUB) * rec + col;");
        g.f("        const double* ip = lds + sub * rec + col;");
        g.f("        if (sub < per) {");
        g.f("          _Pragma(\"unroll 8\")");
        g.f("          for (int r = sub; r < rows; r += per) { *op = *ip; op += (long long)per * SUB * rec; ip += per * rec; }");
        g.f("        }");
        g.f("      } else {");
      }
      g.f("      const long long base_b = span_idx * span + wave_in_span * 64;");
      g.f("      const long long rem = (span_idx + 1) * span - base_b;");
      g.f("      const int n_doubles = (int)(rem < 64 ? rem : 64) * rec;");
      g.f("      double* dst = a.out_pos + base_b * rec;");
      g.f("      double2* dst2 = reinterpret_cast<double2*>(dst);");
      g.f("      const double2* src2 = reinterpret_cast<const double2*>(lds);");
      // (measured: the same copy with a fixed trip count, unrolled - the compiler hoists its 23 store addresses out of the
      //  wave-unit loop and spills them; no faster where it did not)
      g.f("      if ((reinterpret_cast<unsigned long long>(dst) & 15ull) == 0ull) {");
      g.f("        for (int i = lane; i < n_doubles / 2; i += 64) dst2[i] = src2[i];");
      g.f("        if ((n_doubles & 1) && lane == 0) dst[n_doubles - 1] = lds[n_doubles - 1];");
      g.f("      } else {");
      g.f("        for (int i = lane; i < n_doubles; i += 64) dst[i] = lds[i];");
      g.f("      }");
      if (sub_body) g.f("      }");
      g.f("      WAVE_SYNC();");
      g.f("      }");
      stamp(15);
      if (EV) {
        // ---- evaluated epilogue (see okx_quadgen.cpp for the quad form): J at the solved state, undamped LDL^T with one
        // substitution per target, the points' velocities in forward mode, then the metric catalog on duals with all T
        // directions - role points and their velocities straight from this lane's registers.  Rows leave through LDS
        // ([lane][25]: an odd stride) as 192-byte fragments of the [problem][1 + T][24] records.
        LGen& eg = gb ? epg : epc;
        const EpiSrc& ep = gb ? epi_g : epi_cold;
        const int REC = 3 * P.n_out, RS = REC | 1, EVC = OKX_EVAL_COLUMNS;
        g.f("      {");
        g.f("      WAVE_SYNC();  // (the record copy's last LDS reads are done)");
        g.f("      %s", refresh_kz);
        g.out += ep.eval;
        g.f("      const double lambda = 0.0;  // (an undamped factorisation; shadows the solve's damping)");
        g.out += ep.factor;
        g.out += ep.subst;
        eg.out.clear();
        std::vector<std::vector<S3>> vel(T, std::vector<S3>(NP));
        for (int t = 0; t < T; ++t)
          for (int F = 0; F < nf; ++F)
            for (int c = 0; c < 3; ++c) vel[t][eg.fp(F)].c[c] = "tq" + std::to_string(t) + "_" + std::to_string(3 * F + c);
        for (int e = 0; e < P.n_derived; ++e)
          if (!eg.derived_jvp(e, vel)) {
            lds_why = eg.why;
            return false;
          }
        const std::string jvp_src = eg.out;
        eg.out.clear();
        g.f("      const double ev_flags = (ok ? 1.0 : 0.0) + ((!ok || pmin <= %d * 2.220446049250313e-16 * pmax) ? 2.0 : 0.0);", n);
        g.f("      const long long ev_base = span_idx * span + wave_in_span * 64;");
        g.f("      const long long ev_rem = (span_idx + 1) * span - ev_base;");
        g.f("      const int ev_rows = (int)(ev_rem < 64 ? ev_rem : 64);");
        g.f("      if (ea.tan != nullptr) {  // the tangents themselves: [problem][target][record], one target at a time through LDS");
        g.out += jvp_src;
        for (int t = 0; t < T; ++t) {
          g.f("        WAVE_SYNC();");
          g.f("        { double* st = lds + lane * %d;", RS);
          for (int k = 0; k < P.n_out; ++k)
            for (int c = 0; c < 3; ++c) {
              const std::string& nm = vel[t][P.out_point[k]].c[c];
              if (nm.empty()) g.f("          st[%d] = ok ? 0.0 : __builtin_nan(\"\");", 3 * k + c);
              else g.f("          st[%d] = ok ? %s : __builtin_nan(\"\");", 3 * k + c, nm.c_str());
            }
          g.f("        }");
          g.f("        WAVE_SYNC();");
          g.f("        for (int i = lane; i < ev_rows * %d; i += 64) { const int j = i / %d, col = i - j * %d; ea.tan[((ev_base + j) * %d + %d) * %d + col] = lds[j * %d + col]; }",
              REC, REC, REC, T, t, REC, RS);
        }
        g.f("        WAVE_SYNC();");
        g.f("      }");
        g.f("      if (ea.ev != nullptr) {");
        g.out += jvp_src;
        g.f("        EvCfg cfg = ea.cfg;");
        g.f("        if (PG) {  // an ensemble's design references are its geometry's own");
        g.f("          cfg.design_wheel_center_z = GL(%d); cfg.design_contact_patch_z = GL(%d);", ev.gl_gp0 + 3 * P.out_point[es->wheel_center] + 2,
            ev.gl_gp0 + 3 * P.out_point[es->contact_patch] + 2);
        if (es->rack >= 0) g.f("          cfg.design_rack_y = GL(%d);", ev.gl_gp0 + 3 * P.out_point[es->rack] + 1);
        g.f("        }");
        g.f("        DV<%d> RP[EV_SLOTS];", T);
        for (int sl = 0; sl < kEvalSlots; ++sl) {
          const int k = eval_slot_point(*es, sl);
          for (int c = 0; c < 3; ++c) {
            if (k < 0) {
              g.f("        RP[%d].%c = du_const<%d>(0.0);", sl, "xyz"[c], T);
              continue;
            }
            const int pnt = P.out_point[k];
            g.f("        RP[%d].%c.v = p%d_%d;", sl, "xyz"[c], pnt, c);
            for (int t = 0; t < T; ++t) {
              const std::string& nm = vel[t][pnt].c[c];
              g.f("        RP[%d].%c.d[%d] = %s;", sl, "xyz"[c], t, nm.empty() ? "0.0" : ("(ok ? " + nm + " : __builtin_nan(\"\"))").c_str());
            }
          }
        }
        g.f("        Du<%d> em[%d];", T, OKX_METRIC_COUNT);
        g.f("        ev_corner_metrics<%d>(cfg, RP, em);", T);
        for (int r = 0; r <= T; ++r) {
          g.f("        WAVE_SYNC();");
          g.f("        { double* st = lds + lane * 25;");
          if (r == 0) {
            for (int k = 0; k < OKX_METRIC_COUNT; ++k) g.f("          st[%d] = em[%d].v;", k, k);
            g.f("          st[19] = pmin; st[20] = pmax; st[21] = ev_flags; st[22] = 0.0; st[23] = 0.0;");
          } else {
            for (int k = 0; k < OKX_METRIC_COUNT; ++k) g.f("          st[%d] = em[%d].d[%d];", k, k, r - 1);
            g.f("          st[19] = RP[EV_SLOT_WHEEL_CENTER].x.d[%d]; st[20] = RP[EV_SLOT_WHEEL_CENTER].y.d[%d]; st[21] = RP[EV_SLOT_WHEEL_CENTER].z.d[%d];", r - 1, r - 1, r - 1);
            if (es->rack >= 0) g.f("          st[22] = RP[EV_SLOT_RACK].y.d[%d]; st[23] = 0.0;", r - 1);
            else g.f("          st[22] = __builtin_nan(\"\"); st[23] = 0.0;");
          }
          g.f("        }");
          g.f("        WAVE_SYNC();");
          g.f("        for (int i = lane; i < ev_rows * %d; i += 64) { const int j = i / %d, c2 = i - j * %d;", EVC / 2, EVC / 2, EVC / 2);
          g.f("          double2 v2; v2.x = lds[j * 25 + 2 * c2]; v2.y = lds[j * 25 + 2 * c2 + 1];");
          g.f("          reinterpret_cast<double2*>(ea.ev + ((ev_base + j) * %d + %d) * %d)[c2] = v2; }", 1 + T, r, EVC);
        }
        g.f("        WAVE_SYNC();");
        g.f("      }");
        g.f("      }");
      }
    } else if (ns) {
      // nested mode: the wave unit's records of this step lie four problems apart - each one a contiguous run of `rec`
      // doubles: through LDS (which the state no longer needs), every record written by consecutive lanes
      g.f("    if (FULL || a.out_mode == 1) {");
      g.f("      constexpr int rec = FULL ? %d : %d;", 3 * P.n_out, n);
      g.f("      WAVE_SYNC();");
      g.f("      if (FULL) {");
      g.f("      double* st = lds + lane * %d;", 3 * P.n_out);
      for (int k = 0; k < P.n_out; ++k)
        for (int c = 0; c < 3; ++c) g.f("      st[%d] = p%d_%d;", 3 * k + c, P.out_point[k], c);
      g.f("      } else {");
      g.f("      double* st = lds + lane * %d;", n);
      for (int i = 0; i < n; ++i) g.f("      st[%d] = %s;", 3 * ev.perm[i / 3] + i % 3, PF(i).c_str());
      g.f("      }");
      g.f("      WAVE_SYNC();");
      stamp(17);
      g.f("      const long long base_b = span_idx * span + wave_in_span * 64 * unit_len + sidx, lim_b = (span_idx + 1) * span;");
      // lane l copies column l % rec of record (l / rec) of every group of 64 / rec records: one LDS read, one store and a
      // pointer bump per group (no division, no address arithmetic inside the loop); a record leaves as one contiguous run
      g.f("      constexpr int per = 64 / rec > 0 ? 64 / rec : 1;  // records per group");
      g.f("      const int sub = lane / rec, col = lane - sub * rec;");
      g.f("      const long long n_rows = (lim_b - base_b + unit_len - 1) / unit_len;  // records of this step inside the span");
      g.f("      const int rows = (int)(n_rows < 64 ? (n_rows > 0 ? n_rows : 0) : 64);");
      g.f("      double* op = a.out_pos + (base_b + (long long)sub * unit_len) * rec + col;");
      g.f("      const double* ip = lds + sub * rec + col;");
      g.f("      if (sub < per) {");
      g.f("        _Pragma(\"unroll 8\")");
      g.f("        for (int r = sub; r < rows; r += per) { *op = *ip; op += (long long)per * unit_len * rec; ip += per * rec; }");
      g.f("      }");
      g.f("      WAVE_SYNC();");
      g.f("    }");
      stamp(15);
    } else {
      // chains: a lane's problems are far apart in memory, every lane stores its own record
      g.f("    if (valid && FULL) {");
      g.f("      double* o = a.out_pos + bb * %d;", 3 * P.n_out);
      for (int k = 0; k < P.n_out; ++k)
        for (int c = 0; c < 3; ++c) g.f("      o[%d] = p%d_%d;", 3 * k + c, P.out_point[k], c);
      g.f("    } else if (valid && !FULL && a.out_mode == 1) {");
      g.f("      double* o = a.out_pos + bb * %d;", n);
      for (int i = 0; i < n; ++i) g.f("      o[%d] = %s;", 3 * ev.perm[i / 3] + i % 3, PF(i).c_str());
      g.f("    }");
    }
    g.f("      }");
    g.f("    }  // chain steps");
    g.f("  }  // wave units");
    g.f("}");
    g.f("");
    return true;
  };
  if (EV) {
    if (!body(false, false, false) || (split_g && !body(false, false, true))) {
      *why = lds_why;
      return false;
    }
    g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_evsolve_u(QEvArgs ea) { okx_lane_body_cold<false, true, false>(ea.q, ea); }");
    g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_evsolve_g(QEvArgs ea) { okx_lane_body_%s<true, true, false>(ea.q, ea); }", split_g ? "coldg" : "cold");
    {  // okx_evaluate_batch's lane form (reference core/sweep.py:217-245 evaluate_solved_sweep): the epilogue on given states
      bool all_out = true;
      std::vector<bool> is_out(NP, false);
      for (int k = 0; k < P.n_out; ++k) is_out[P.out_point[k]] = true;
      for (int F = 0; F < nf; ++F) all_out = all_out && is_out[evc.fp(F)];
      if (all_out) {
        g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_evaluate_u(QEvArgs ea) { okx_lane_body_cold<false, true, true>(ea.q, ea); }");
        g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_evaluate_g(QEvArgs ea) { okx_lane_body_%s<true, true, true>(ea.q, ea); }", split_g ? "coldg" : "cold");
      }
    }
    *src = g.out;
    return true;
  }
  if (!body(false, false, false) || (split_g && !body(false, false, true)) || !(flat_chain ? body(false, true, false) : body(true, false, false))) {
    *why = lds_why;
    return false;
  }
  // The nested start mode is generated on request only (developer switch lane_nested): measured on BASELINE config 5 it
  // halves the evaluations per solve (2.97 -> 1.5) and does not shorten the launch (0.47 -> 0.49 ms) - a wave unit steps at
  // the pace of its slowest lane, a confirming pass costs a lone wavefront most of a full one, and every unit-step carries
  // ~19 k cycles that are not passes (profiles/r05/EXPERIMENTS.md section 5).  chain_len = -1 keeps resolving to
  // independent solves on the lane kernel.
  const bool nested = light_ok && dev_switch("lane_nested");
  // (two bodies: own geometry reads its tables through the scalar cache; per-geometry launches stage each geometry's tables
  //  and first-step table in LDS once per wave unit - its four unit-steps share them - like the independent-solve body)
  if (nested && (!body(false, true, false, true) || (split_g && !body(false, true, true, true)))) {
    *why = lds_why;
    return false;
  }
  // ---- parity / debug kernel: r, J^T J, J^T r at given x, and the damped step for a given lambda ----
  g.f("struct QEvalArgs { const double* x; const double* targets; double* r; double* ata; double* atr; double* dx;");
  g.f("  double lambda; long long n_problems; const double* design_pos; const double* row_param; const double* dop_param; };");
  g.f("#undef GL\n#define GL(o) gl[(o) + kz]");
  g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_eval(QEvalArgs a) {");
  g.f("  const double* gp = a.design_pos; const double* gq = a.row_param; (void)gq;");
  g.f("  __shared__ double gl[%d];", gl_size);
  g.f("  const int lane = threadIdx.x;");
  g.f("  int kz = 0;");
  stage_tables(g, "  ");
  g.f("  __syncthreads();");
  g.f("  %s", refresh_kz);
  g.f("  for (long long wu = blockIdx.x; wu * 64 < a.n_problems; wu += gridDim.x) {");
  g.f("    long long bb = wu * 64 + threadIdx.x; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;");
  for (int p = 0; p < NP; ++p)
    if (used[p] && !is_fixed(p))
      for (int c = 0; c < 3; ++c) g.f("    double p%d_%d = gp[%d];", p, c, 3 * p + c);
  for (int i = 0; i < n; ++i) g.f("    %s = a.x[bb * %d + %d];", PF(i).c_str(), n, 3 * ev.perm[i / 3] + i % 3);
  for (int t = 0; t < T; ++t) g.f("    const double tv%d = a.targets[bb * %d + %d];", t, T, t);
  {
    // the parity kernel additionally accumulates the whole lower triangle of J^T J where the rows are (E{i}_{j}): the
    // solve kernels never hold it in that form
    LGen ee(P);
    ee.uid = 500000;
    ee.early_ata = true;
    ee.pin_acc = false;
    ee.hoisted_names = ev.hoisted_names;
    for (int idx = 0; idx < P.n_active; ++idx) (void)ee.derived_op(P.active_op[idx], true);
    (void)ee.emit_rows();
    g.out += ee.out;
    ee.out.clear();
    g.f("    if (valid) {");
    for (int i = 0; i < P.m; ++i) g.f("      a.r[bb * %d + %d] = r%d;", P.m, i, i);
    for (int i = 0; i < n; ++i) {
      const int pi = 3 * ev.perm[i / 3] + i % 3;
      g.f("      a.atr[bb * %d + %d] = %s;", n, pi, LGen::gn(i).c_str());
      g.f("      a.ata[(bb * %d + %d) * %d + %d] = %s;", n, pi, n, pi, LGen::A(i, i).c_str());
      for (int j = 0; j < i; ++j)
        if (ee.nz[i][j] && ee.early_ata) {
          const int pj = 3 * ev.perm[j / 3] + j % 3;
          g.f("      a.ata[(bb * %d + %d) * %d + %d] = %s;", n, pi, n, pj, LGen::E(i, j).c_str());
          if (i / 3 == j / 3) g.f("      a.ata[(bb * %d + %d) * %d + %d] = %s;", n, pj, n, pi, LGen::E(i, j).c_str());
        }
    }
    g.f("    }");
    g.f("    const double lambda = a.lambda;");
    std::vector<std::string> rhs;
    for (int i = 0; i < n; ++i) rhs.push_back("-" + LGen::gn(i));
    ee.col_fence = ev.col_fence;
    ee.resident_rows = ev.resident_rows;
    ee.emit_factor(rhs);
    ee.emit_backward("nx");
    g.out += ee.out;
  }
  g.f("    if (valid) {");
  for (int i = 0; i < n; ++i) g.f("      a.dx[bb * %d + %d] = ok ? nx%d : __builtin_nan(\"\");", n, 3 * ev.perm[i / 3] + i % 3, i);
  g.f("    }");
  g.f("  }");
  g.f("}");
  g.f("");
  for (const char* body : {"solve", "chain"})
    for (const char* geo : {"u", "g"})
      for (const char* out : {"", "_c"})   // _c: compact outputs (free coordinates or nothing)
        g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_%s_%s%s(QArgs a) { okx_lane_body_%s<%s, %s%s>(a); }", body, geo,
            out, body[0] != 's' ? "chain" : (split_g && geo[0] == 'g') ? "coldg" : "cold", geo[0] == 'g' ? "true" : "false", out[0] ? "false" : "true",
            refine && body[0] == 's' ? ", 1, false" : "");
  if (refine)
    for (const char* kind : {"refc", "refw"})   // coarse launch (every fourth step, cold) / warm launches (the steps between)
      for (const char* geo : {"u", "g"})
        for (const char* out : {"", "_c"})
          g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_%s_%s%s(QArgs a) { okx_lane_body_%s<%s, %s, 4, %s>(a); }", kind, geo, out,
              (split_g && geo[0] == 'g') ? "coldg" : "cold", geo[0] == 'g' ? "true" : "false", out[0] ? "false" : "true", kind[3] == 'w' ? "true" : "false");
  if (nested)
    for (const char* geo : {"u", "g"})
      for (const char* out : {"", "_c"})
        g.f("extern \"C\" __global__ void __launch_bounds__(64, 1) okx_lane_nest_%s%s(QArgs a) { okx_lane_body_%s<%s, %s>(a); }", geo, out,
            split_g && geo[0] == 'g' ? "nestg" : "nest", geo[0] == 'g' ? "true" : "false", out[0] ? "false" : "true");
  (void)ev.undefs;  // (the macros live to the end of the translation unit: one program per module)
  *src = g.out;
  return true;
}

}  // namespace okx
