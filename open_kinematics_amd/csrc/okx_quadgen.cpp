// okx_quadgen.cpp — source generator of the "quad" solve kernel: a gfx950 kernel specialised to
// ONE constraint program, compiled at okx_program_create time with hiprtc (okx_jit.cpp).
//
// Why generate code.  The reference itself generates its Jacobian rows (tools/generate_jacobians.py
// -> core/jacobians.py) and then interprets the problem structure in Python on every call
// (ResidualComputer.compute / compute_jacobian, core/solver.py:226-275, :502-581).  The generic
// kernels in okx_kernels.hip interpret the same structure on the GPU (one wavefront per problem,
// tables in LDS); here the structure is resolved at program-creation time instead, so that the
// kernel is straight-line fp64 code with every operand in a statically named register.
//
// Execution model of the generated kernel (CDNA4, 64-wide wavefronts):
//   * FOUR LANES OWN ONE PROBLEM ("quad"), 16 problems per wavefront, all in lockstep.
//     Lane c in {0,1,2} of a quad owns Cartesian component c of every point and therefore the
//     variables x[3F+c], the Jacobian columns 3F+c and the rows 3F+c of J^T J / of its LDL^T
//     factor, for every free point F.  Lane 3 carries zeros.
//   * All three lanes run the same instruction stream on the same block structure, so there
//     is no index table, no LDS and no divergence inside a problem.  Cross-lane operands
//     (dot products, cross products, J^T J columns, pivots) move through DPP quad_perm
//     (v_mov_b32_dpp, 2 per double), never through LDS.
//   * Structural zeros of J^T J and of the factor (block level, symbolic fill-in) are
//     resolved here: the emitted LDL^T touches only blocks that can be non-zero.
//   * Levenberg-Marquardt control flow is predicated per quad; a wavefront iterates until its
//     16 problems are done.
//
// Same objective, same residual/Jacobian definitions and same LM policy as okx_solve_kernel
// (DESIGN.md §4); only the parallel decomposition differs.
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

// A per-lane fp64 value held in a named variable, with a sign folded into its uses.
struct LV {
  std::string n;
  int sg = 1;
};

// d(derived point)/d(free block) as seen by the quad: lane c holds COLUMN c of the 3x3 block
// (col[r] = B[r][c]), or the block is s * I.
struct Blk {
  bool scaled = true;
  std::string s;       // scaled: scalar expression
  std::string col[3];  // general: variable names
};

struct BlkTerm {
  Blk b;
  int sg;
};

class Gen {
 public:
  explicit Gen(const DevProgram& prog, const PairView* pair = nullptr) : P(prog), pv(pair) {
    blk_of_point.assign(P.n_points, -1);
    dop_of_point.assign(P.n_points, -1);
    elimination_order();
    for (int F = 0; F < P.n_free; ++F) blk_of_point[fp(F)] = F;
    for (int e = 0; e < P.n_derived; ++e) dop_of_point[P.dop_out[e]] = e;
  }

  // Block F of the generated code (x{F}, rows 3F+c of J^T J, elimination step F of the LDL^T) is
  // the program's free point perm[F]: a greedy minimum-degree order on the block graph of J^T J
  // (ties: program order), so that leaf chains (rack pickup, pushrod / rocker / drop-link) are
  // eliminated before the upright's clique and create no fill-in.
  std::vector<int> perm;
  int fp(int F) const { return P.free_point[perm[F]]; }
  void elimination_order() {
    const int nf = P.n_free;
    std::vector<std::set<int>> adj(nf);
    for (int i = 0; i < P.m; ++i)
      for (int a = 0; a < P.row_nblk[i]; ++a)
        for (int b = 0; b < P.row_nblk[i]; ++b)
          if (a != b) adj[P.row_blk[i][a]].insert(P.row_blk[i][b]);
    std::vector<bool> gone(nf, false);
    perm.clear();
    // Pair mode with ONE joining row: the joined point's block is eliminated LAST.  The right-hand side of the joining row's
    // own system (D~ z = w) is then zero everywhere but in the last block, so z needs no substitution of its own: the
    // coupling correction touches the last block only and rides on the backward substitution of the step itself
    // (emit_substitute's `last_block_hook`; round 6 - a pass of the axle kernel: two substitutions -> one).
    int held_back = -1;
    if (pv && pv->joins.size() == 1)
      for (int k = 0; k < nf; ++k)
        if (P.free_point[k] == pv->couple_point) held_back = k;
    for (int step = 0; step < nf; ++step) {
      int best = -1;
      for (int k = 0; k < nf; ++k)
        if (!gone[k] && k != held_back && (best < 0 || adj[k].size() < adj[best].size())) best = k;
      if (best < 0) best = held_back;
      perm.push_back(best);
      gone[best] = true;
      for (int u : adj[best]) {
        adj[u].erase(best);
        for (int w : adj[best])
          if (w != u) adj[u].insert(w);
      }
      adj[best].clear();
    }
  }

  const DevProgram& P;
  const PairView* pv;  // non-null: P is the side program of a two-sided problem (okx_pairview.cpp)
  std::string out;
  std::string why;

  // Index of program data as seen by a lane: one literal, or in pair mode a select on the side bit
  // q1 (all such accesses are chain-constant loads or record stores, never inside the LM passes).
  static std::string sel(int i0, int i1) {
    if (i0 == i1) return std::to_string(i0);
    return "(q1 ? " + std::to_string(i1) + " : " + std::to_string(i0) + ")";
  }
  std::string point3(int p) const {  // 3 * point index into gp / design_pos
    return pv ? sel(3 * pv->pt[0][p], 3 * pv->pt[1][p]) : std::to_string(3 * p);
  }
  std::string point4(int p) const {  // 4 * program point index (predictor table)
    return pv ? sel(4 * pv->pt[0][p], 4 * pv->pt[1][p]) : std::to_string(4 * p);
  }
  std::string crow8(int i, int k) const {  // constraint row parameter offset into gq
    return pv ? sel(8 * pv->row[0][i] + k, 8 * pv->row[1][i] + k) : std::to_string(8 * i + k);
  }
  int target_of_row(int i) const { return (int)P.row_param[i][3]; }
  std::string trow8(int i, int k) const {  // target row parameter offset into a.row_param
    if (!pv) return std::to_string(8 * i + k);
    const int t = target_of_row(i);
    int t0 = pv->tgt[0][t], t1 = pv->tgt[1][t];
    if (t0 < 0) t0 = t1;
    if (t1 < 0) t1 = t0;
    return sel(8 * (pv->n_prog_crows + t0) + k, 8 * (pv->n_prog_crows + t1) + k);
  }
  std::string target_slot(int t) const {  // column of a.targets
    if (!pv) return std::to_string(t);
    int t0 = pv->tgt[0][t], t1 = pv->tgt[1][t];
    if (t0 < 0) t0 = t1;
    if (t1 < 0) t1 = t0;
    return sel(t0, t1);
  }
  std::string target_enable(int t) const {  // 1.0 where this side really has target t
    if (!pv) return "1.0";
    const bool e0 = pv->tgt[0][t] >= 0, e1 = pv->tgt[1][t] >= 0;
    if (e0 && e1) return "1.0";
    return e0 ? "(q1 ? 0.0 : 1.0)" : "(q1 ? 1.0 : 0.0)";
  }
  std::string dop_slot(int e) const { return pv ? sel(pv->dop[0][e], pv->dop[1][e]) : std::to_string(e); }
  std::vector<int> blk_of_point, dop_of_point;
  int uid = 0;
  std::map<std::string, std::string> rot1_, rot2_;
  std::map<std::string, std::vector<std::string>> bc_;
  std::map<int, std::map<int, Blk>> dblk;  // active derived op -> free block -> chain block
  bool nz[kMaxFree][kMaxFree] = {};

  void f(const char* fmt, ...) {
    char buf[1024];
    va_list ap, again;
    va_start(ap, fmt);
    va_copy(again, ap);
    const int need = std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (need >= (int)sizeof(buf)) {  // a long combined expression: format again into an exact-size buffer
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
  static std::string f_str(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
  }
  std::string tmp(const char* base) { return "_" + std::string(base) + std::to_string(uid++); }
  static std::string sx(const LV& v) { return (v.sg < 0 ? "-" : "") + v.n; }
  static std::string pn(int p) { return "p" + std::to_string(p); }
  void reset_caches() {
    rot1_.clear();
    rot2_.clear();
    bc_.clear();
  }

  // ---- quad data movement ----
  std::string bcast(const std::string& n, int r) {
    std::vector<std::string>& slot = bc_[n];
    if (slot.empty()) slot.resize(3);
    if (slot[r].empty()) {
      std::string t = tmp("b");
      f("    const double %s = QB%d(%s);", t.c_str(), r, n.c_str());
      slot[r] = t;
    }
    return slot[r];
  }
  std::string rot1(const std::string& n) {
    auto it = rot1_.find(n);
    if (it != rot1_.end()) return it->second;
    std::string t = tmp("ra");
    f("    const double %s = QR1(%s);", t.c_str(), n.c_str());
    rot1_[n] = t;
    return t;
  }
  std::string rot2(const std::string& n) {
    auto it = rot2_.find(n);
    if (it != rot2_.end()) return it->second;
    std::string t = tmp("rb");
    f("    const double %s = QR2(%s);", t.c_str(), n.c_str());
    rot2_[n] = t;
    return t;
  }
  // lane c: (a x b)_c = a_{c+1} b_{c+2} - a_{c+2} b_{c+1}
  std::string cross(const std::string& a, const std::string& b) {
    std::string a1 = rot1(a), a2 = rot2(a), b1 = rot1(b), b2 = rot2(b);
    std::string t = tmp("cx");
    f("    const double %s = %s * %s - %s * %s;", t.c_str(), a1.c_str(), b2.c_str(), a2.c_str(), b1.c_str());
    return t;
  }
  std::string vsub(const std::string& a, const std::string& b) {
    std::string t = tmp("v");
    f("    const double %s = %s - %s;", t.c_str(), a.c_str(), b.c_str());
    return t;
  }
  std::string dot(const std::string& a, const std::string& b) {
    std::string t = tmp("d");
    f("    const double %s = qsum(%s * %s);", t.c_str(), a.c_str(), b.c_str());
    return t;
  }

  // ---- derived points ----
  std::map<int, Blk> blocks_of_point(int p) {
    std::map<int, Blk> r;
    if (blk_of_point[p] >= 0) {
      Blk b;
      b.scaled = true;
      b.s = "1.0";
      r[blk_of_point[p]] = b;
    } else if (dop_of_point[p] >= 0) {
      auto it = dblk.find(dop_of_point[p]);
      if (it != dblk.end()) r = it->second;
    }
    return r;
  }

  // Sum block terms per free block into one Blk each.
  std::map<int, Blk> combine(std::map<int, std::vector<BlkTerm>>& acc) {
    std::map<int, Blk> res;
    for (auto& kv : acc) {
      std::string ssum;
      std::vector<BlkTerm*> gen;
      for (auto& t : kv.second) {
        if (t.b.scaled) {
          if (!ssum.empty()) ssum += t.sg < 0 ? " - " : " + ";
          else if (t.sg < 0) ssum += "-";
          ssum += "(" + t.b.s + ")";
        } else {
          gen.push_back(&t);
        }
      }
      Blk b;
      if (gen.empty()) {
        b.scaled = true;
        std::string t = tmp("ks");
        f("    const double %s = %s;", t.c_str(), ssum.c_str());
        b.s = t;
      } else {
        b.scaled = false;
        for (int r = 0; r < 3; ++r) {
          std::string e;
          for (auto* t : gen) {
            if (!e.empty()) e += t->sg < 0 ? " - " : " + ";
            else if (t->sg < 0) e += "-";
            e += t->b.col[r];
          }
          if (!ssum.empty()) e += " + (" + ssum + ") * e" + std::to_string(r);
          std::string t = tmp("B");
          f("    const double %s = %s;", t.c_str(), e.c_str());
          b.col[r] = t;
        }
      }
      res[kv.first] = b;
    }
    return res;
  }

  bool derived_op(int e, bool with_blocks) {
    const int type = P.dop_type[e];
    const int* pts = P.dop_pts[e];
    const std::string o = pn(P.dop_out[e]);
    f("    // derived op %d (type %d) -> point %d", e, type, P.dop_out[e]);
    if (type == OKX_DOP_MIDPOINT) {  // definitions.py:76-89
      f("    %s = %s + (%s - %s) * 0.5;", o.c_str(), pn(pts[0]).c_str(), pn(pts[1]).c_str(), pn(pts[0]).c_str());
      if (with_blocks) {
        std::map<int, std::vector<BlkTerm>> acc;
        for (int s = 0; s < 2; ++s)
          for (auto& kv : blocks_of_point(pts[s])) {
            Blk b = kv.second;
            if (b.scaled) {
              b.s = "0.5 * (" + b.s + ")";
            } else {
              for (int r = 0; r < 3; ++r) {
                std::string t = tmp("B");
                f("    const double %s = 0.5 * %s;", t.c_str(), b.col[r].c_str());
                b.col[r] = t;
              }
            }
            acc[kv.first].push_back({b, 1});
          }
        dblk[e] = combine(acc);
      }
      return true;
    }
    if (type == OKX_DOP_ALONG) {  // definitions.py:24-33, :92-155: out = base + normalize(a - b) * c
      std::string v = vsub(pn(pts[1]), pn(pts[2]));
      std::string s2 = dot(v, v);
      std::string nrm = tmp("nr"), inrm = tmp("in"), u = tmp("u");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", nrm.c_str(), inrm.c_str(), s2.c_str(), nrm.c_str(),
        inrm.c_str());
      f("    const double %s = %s * %s;", u.c_str(), v.c_str(), inrm.c_str());
      f("    %s = %s + %s * %s;", o.c_str(), pn(pts[0]).c_str(), u.c_str(), dp(e).c_str());
      if (with_blocks) {
        std::map<int, std::vector<BlkTerm>> acc;
        for (auto& kv : blocks_of_point(pts[0])) acc[kv.first].push_back({kv.second, 1});
        std::string k = tmp("k");
        f("    const double %s = %s * %s;", k.c_str(), dp(e).c_str(), inrm.c_str());
        std::string ub[3];
        bool have_ub = false;
        for (int s = 1; s <= 2; ++s) {
          const int sg = s == 1 ? 1 : -1;
          for (auto& kv : blocks_of_point(pts[s])) {
            if (!have_ub) {
              for (int r = 0; r < 3; ++r) ub[r] = bcast(u, r);
              have_ub = true;
            }
            Blk nb;
            nb.scaled = false;
            const Blk& b = kv.second;
            if (b.scaled) {  // k s (e_r - u_r u_c)
              std::string ks = tmp("ks");
              f("    const double %s = %s * (%s);", ks.c_str(), k.c_str(), b.s.c_str());
              for (int r = 0; r < 3; ++r) {
                std::string t = tmp("B");
                f("    const double %s = %s * (e%d - %s * %s);", t.c_str(), ks.c_str(), r, ub[r].c_str(), u.c_str());
                nb.col[r] = t;
              }
            } else {  // k (b_r - u_r (u . b))
              std::string ud = tmp("ud");
              f("    const double %s = %s * %s + %s * %s + %s * %s;", ud.c_str(), ub[0].c_str(), b.col[0].c_str(),
                ub[1].c_str(), b.col[1].c_str(), ub[2].c_str(), b.col[2].c_str());
              for (int r = 0; r < 3; ++r) {
                std::string t = tmp("B");
                f("    const double %s = %s * (%s - %s * %s);", t.c_str(), k.c_str(), b.col[r].c_str(), ub[r].c_str(),
                  ud.c_str());
                nb.col[r] = t;
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
      std::string v = vsub(pn(pts[2]), pn(pts[1]));
      std::string vv = dot(v, v);
      std::string vn = tmp("vn"), ivn = tmp("iv"), ax = tmp("ax");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", vn.c_str(), ivn.c_str(), vv.c_str(), vn.c_str(),
        ivn.c_str());
      f("    const double %s = %s * %s;", ax.c_str(), v.c_str(), ivn.c_str());
      std::string az = bcast(ax, 2);
      std::string wd = tmp("wd");
      f("    const double %s = %s * %s - e2;", wd.c_str(), az.c_str(), ax.c_str());  // -ga a - e_z, ga = -a_z
      std::string ww = dot(wd, wd);
      std::string wn = tmp("wn"), iwn = tmp("iw");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", wn.c_str(), iwn.c_str(), ww.c_str(), wn.c_str(),
        iwn.c_str());
      f("    %s = %s + (%s * %s) * %s;", o.c_str(), pn(pts[0]).c_str(), wd.c_str(), iwn.c_str(), dp(e).c_str());
      if (with_blocks) {
        // d out / d axo = T = R Nw Wa Na (axi: -T, wheel centre: I) with Na = (I - a a^T)/|v|,
        // Wa = a_z I + a e_z^T, Nw = (I - wu wu^T)/|wd|.  Lane c forms column c: T applied to the
        // lane-held column of the input's own block (e_c for a free input).
        std::string wu = tmp("wu");
        f("    const double %s = %s * %s;", wu.c_str(), wd.c_str(), iwn.c_str());
        std::string ab[3], wb[3];
        for (int r = 0; r < 3; ++r) ab[r] = bcast(ax, r), wb[r] = bcast(wu, r);
        auto apply_t = [&](const std::string b[3], std::string t[3]) {
          std::string adb = tmp("ad");
          f("    const double %s = %s * %s + %s * %s + %s * %s;", adb.c_str(), ab[0].c_str(), b[0].c_str(), ab[1].c_str(),
            b[1].c_str(), ab[2].c_str(), b[2].c_str());
          std::string nn[3], w[3];
          for (int r = 0; r < 3; ++r) {
            nn[r] = tmp("n");
            f("    const double %s = %s * (%s - %s * %s);", nn[r].c_str(), ivn.c_str(), b[r].c_str(), ab[r].c_str(), adb.c_str());
          }
          for (int r = 0; r < 3; ++r) {
            w[r] = tmp("w");
            f("    const double %s = %s * %s + %s * %s;", w[r].c_str(), az.c_str(), nn[r].c_str(), ab[r].c_str(), nn[2].c_str());
          }
          std::string wdw = tmp("ww");
          f("    const double %s = %s * %s + %s * %s + %s * %s;", wdw.c_str(), wb[0].c_str(), w[0].c_str(), wb[1].c_str(),
            w[1].c_str(), wb[2].c_str(), w[2].c_str());
          for (int r = 0; r < 3; ++r) {
            t[r] = tmp("B");
            f("    const double %s = %s * %s * (%s - %s * %s);", t[r].c_str(), dp(e).c_str(), iwn.c_str(), w[r].c_str(),
              wb[r].c_str(), wdw.c_str());
          }
        };
        std::map<int, std::vector<BlkTerm>> acc;
        for (auto& kv : blocks_of_point(pts[0])) acc[kv.first].push_back({kv.second, 1});
        for (int sidx = 1; sidx <= 2; ++sidx) {
          const int sg = sidx == 2 ? 1 : -1;  // pts[2] = axle outboard (+T), pts[1] = axle inboard (-T)
          for (auto& kv : blocks_of_point(pts[sidx])) {
            const Blk& b = kv.second;
            std::string col[3];
            if (b.scaled) {
              for (int r = 0; r < 3; ++r) {
                col[r] = tmp("c");
                f("    const double %s = (%s) * e%d;", col[r].c_str(), b.s.c_str(), r);
              }
            } else {
              for (int r = 0; r < 3; ++r) col[r] = b.col[r];
            }
            Blk nb;
            nb.scaled = false;
            apply_t(col, nb.col);
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

  // ---- rows ----
  struct RowOut {
    std::string r;                            // residual (uniform in the quad)
    std::vector<std::pair<int, LV>> partial;  // point -> d r / d point (lane component)
    std::string absres;                       // |r| in the reference's row definition ("" = skip)
  };

  // Derived-op parameter (axial offset, tyre radius ...): program constant, loaded once.
  std::string dp(int e) {
    auto key = std::make_pair(-1 - e, 0);
    auto it = hoisted_names.find(key);
    if (it != hoisted_names.end()) return it->second;
    char name[48], line[200];
    std::snprintf(name, sizeof(name), "hd%d", e);
    const std::string home = scalar_home(name);
    std::snprintf(line, sizeof(line), "      %s%s = a.dop_param[%s];\n", sdecl(), home.c_str(), dop_slot(e).c_str());
    hoisted += line;
    hoisted_names[key] = home;
    return home;
  }

  // Scalar row parameter (length, angle, volume ...): constant for a whole chain, so it is loaded
  // once in front of the LM loop like the lane-component parameters below.  (Left inside the loop
  // these loads are vector loads - the kernel also stores to global memory, so the compiler may not
  // use the scalar cache - and a lone wavefront per SIMD exposes their L2 latency every pass.)
  std::string rp(int i, int k) {
    i = pin_leader(i);
    auto key = std::make_pair(i, 100 + k);
    auto it = hoisted_names.find(key);
    if (it != hoisted_names.end()) return it->second;
    char name[48], line[200];
    std::snprintf(name, sizeof(name), "hs%d_%d", i, k);
    const std::string home = scalar_home(name);
    if (i < P.n_crows)
      std::snprintf(line, sizeof(line), "      %s%s = gq[%s];\n", sdecl(), home.c_str(), crow8(i, k).c_str());
    else
      std::snprintf(line, sizeof(line), "      %s%s = a.row_param[%s];\n", sdecl(), home.c_str(), trow8(i, k).c_str());
    hoisted += line;
    hoisted_names[key] = home;
    return home;
  }
  // Lane-component load of three consecutive row parameters (0 in lane 3).  These are constant
  // for a whole chain (geometry), so they are emitted into `hoisted`, which the kernel places
  // in front of the Levenberg-Marquardt loop, one load per distinct (row, offset).
  std::string hoisted;
  std::map<std::pair<int, int>, std::string> hoisted_names;
  // Pair mode keeps the chain constants in LDS instead of registers (the half program of an axle
  // has ~50 of them and the kernel is register-bound): scalars in hsl[slot][quad-side], lane
  // components in hql[slot][lane]; a lane only ever reads what it (or a lane of its own quad, with
  // the same value) wrote, in program order, so no barrier is involved.
  bool lds_constants = false;

  int n_scalar_slots = 0, n_lane_slots = 0;
  // (the scalars and the lane components can be sent back to registers separately: experiment switches)
  bool scalars_in_regs = false, lanes_in_regs = false;
  std::string scalar_home(const char* name) {
    if (!lds_constants || scalars_in_regs) return name;
    return "hsl[" + std::to_string(16 * n_scalar_slots++) + " + qs]";
  }
  std::string lane_home(const char* name) {
    if (!lds_constants || lanes_in_regs) return name;
    return "hql[" + std::to_string(64 * n_lane_slots++) + " + lane]";
  }
  const char* sdecl() const { return lds_constants && !scalars_in_regs ? "" : "const double "; }
  const char* ldecl() const { return lds_constants && !lanes_in_regs ? "" : "const double "; }
  std::string rpv(int i, int k0) {
    i = pin_leader(i);
    auto key = std::make_pair(i, k0);
    auto it = hoisted_names.find(key);
    if (it != hoisted_names.end()) return it->second;
    char name[48], line[200];
    std::snprintf(name, sizeof(name), "hq%d_%d", i, k0);
    const std::string home = lane_home(name);
    if (i < P.n_crows)
      std::snprintf(line, sizeof(line), "      %s%s = ld3(gq + %s + cc, c);\n", ldecl(), home.c_str(), crow8(i, k0).c_str());
    else  // target direction; zero on a side that does not carry this target (pair mode)
      std::snprintf(line, sizeof(line), "      %s%s = ld3(a.row_param + %s + cc, c) * %s;\n", ldecl(), home.c_str(),
                    trow8(i, k0).c_str(), target_enable(target_of_row(i)).c_str());
    hoisted += line;
    hoisted_names[key] = home;
    return home;
  }
  // The three LINE_PIN rows that one point-on-line constraint flattens into (same point, same
  // line in the program's own geometry, components 0/1/2) share their line parameters and their
  // cross product: the first of them is the group's leader.  Per-geometry tables keep them equal
  // (okx_rebind_design writes the same anchor for every pin of a point).
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
  std::map<int, std::string> pin_cross_;  // leader row -> (p - line point) x line dir

  bool row(int i, RowOut* ro) {
    const int type = P.row_type[i];
    const int* pts = P.row_pts[i];
    f("    // row %d (type %d)", i, type);
    std::string r = "r" + std::to_string(i);
    ro->r = r;
    ro->absres = "fabs(" + r + ")";
    switch (type) {
      case OKX_ROW_DISTANCE:
      case OKX_ROW_SPHERICAL: {  // constraints.py:125-134,162-170; jacobians.py:35-51
        std::string d = vsub(pn(pts[1]), pn(pts[0]));
        std::string s = dot(d, d);
        std::string root = tmp("rt"), inv = tmp("iv"), g = tmp("g");
        f("    double %s, %s; fast_sqrt_rsqrt(%s + EPS_SQ, &%s, &%s);", root.c_str(), inv.c_str(), s.c_str(),
          root.c_str(), inv.c_str());
        f("    const double %s = %s * %s;", g.c_str(), d.c_str(), inv.c_str());
        if (type == OKX_ROW_DISTANCE)
          f("    const double %s = (%s - EPS) - %s;", r.c_str(), root.c_str(), rp(i, 0).c_str());
        else
          f("    const double %s = %s - EPS;", r.c_str(), root.c_str());
        ro->partial.push_back({pts[0], {g, -1}});
        ro->partial.push_back({pts[1], {g, 1}});
        return true;
      }
      case OKX_ROW_ANGLE:
      case OKX_ROW_THREE_POINT_ANGLE: {  // constraints.py:223-243,287-308; jacobians.py:55-188
        std::string v1, v2;
        if (type == OKX_ROW_ANGLE) {
          v1 = vsub(pn(pts[1]), pn(pts[0]));
          v2 = vsub(pn(pts[3]), pn(pts[2]));
        } else {
          v1 = vsub(pn(pts[0]), pn(pts[1]));
          v2 = vsub(pn(pts[2]), pn(pts[1]));
        }
        std::string cv = cross(v1, v2);
        std::string c2 = dot(cv, cv);
        std::string dt = dot(v1, v2);
        std::string s = tmp("s"), is = tmp("is"), t15 = tmp("t");
        f("    const double %s = EPS_SQ + %s;", t15.c_str(), c2.c_str());
        f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", s.c_str(), is.c_str(), t15.c_str(), s.c_str(),
          is.c_str());
        std::string inv = tmp("iv"), ka = tmp("ka"), kb = tmp("kb");
        f("    const double %s = fast_rcp(%s + %s * %s);", inv.c_str(), t15.c_str(), dt.c_str(), dt.c_str());
        f("    const double %s = %s * %s * %s, %s = %s * %s;", ka.c_str(), dt.c_str(), inv.c_str(), is.c_str(),
          kb.c_str(), s.c_str(), inv.c_str());
        std::string w1 = cross(v2, cv), w2 = cross(cv, v1);
        std::string g1 = tmp("g"), g2 = tmp("g");
        f("    const double %s = %s * %s - %s * %s;", g1.c_str(), ka.c_str(), w1.c_str(), kb.c_str(), v2.c_str());
        f("    const double %s = %s * %s - %s * %s;", g2.c_str(), ka.c_str(), w2.c_str(), kb.c_str(), v1.c_str());
        f("    const double %s = lean_atan2_pos<%s>(%s - EPS, %s, %s) - %s;", r.c_str(), pv ? "true" : "false", s.c_str(), dt.c_str(), pv ? "atl" : "nullptr",
          rp(i, 0).c_str());
        if (type == OKX_ROW_ANGLE) {
          ro->partial.push_back({pts[0], {g1, -1}});
          ro->partial.push_back({pts[1], {g1, 1}});
          ro->partial.push_back({pts[2], {g2, -1}});
          ro->partial.push_back({pts[3], {g2, 1}});
        } else {
          std::string gm = tmp("g");
          f("    const double %s = -%s - %s;", gm.c_str(), g1.c_str(), g2.c_str());
          ro->partial.push_back({pts[0], {g1, 1}});
          ro->partial.push_back({pts[1], {gm, 1}});
          ro->partial.push_back({pts[2], {g2, 1}});
        }
        return true;
      }
      case OKX_ROW_VECTORS_PARALLEL:
      case OKX_ROW_VECTORS_PERPENDICULAR: {  // constraints.py:351-371,414-429; jacobians.py:192-318
        std::string v1 = vsub(pn(pts[1]), pn(pts[0])), v2 = vsub(pn(pts[3]), pn(pts[2]));
        std::string n1 = dot(v1, v1), n2 = dot(v2, v2);
        std::string s1 = tmp("s"), s2 = tmp("s"), g1 = tmp("g"), g2 = tmp("g");
        f("    const double %s = sqrt(EPS_SQ + %s), %s = sqrt(EPS_SQ + %s);", s1.c_str(), n1.c_str(), s2.c_str(), n2.c_str());
        if (type == OKX_ROW_VECTORS_PARALLEL) {
          std::string cv = cross(v1, v2);
          std::string c2 = dot(cv, cv);
          std::string sc = tmp("s");
          f("    const double %s = sqrt(EPS_SQ + %s);", sc.c_str(), c2.c_str());
          std::string w1 = cross(v2, cv), w2 = cross(cv, v1);
          std::string k26 = tmp("k"), k19 = tmp("k"), k31 = tmp("k");
          f("    const double %s = 1.0 / (%s * %s * %s), %s = %s / (%s * %s * %s * %s), %s = %s / (%s * %s * %s * %s);",
            k26.c_str(), s1.c_str(), s2.c_str(), sc.c_str(), k19.c_str(), sc.c_str(), s2.c_str(), s1.c_str(), s1.c_str(),
            s1.c_str(), k31.c_str(), sc.c_str(), s1.c_str(), s2.c_str(), s2.c_str(), s2.c_str());
          f("    const double %s = %s * %s - %s * %s, %s = %s * %s - %s * %s;", g1.c_str(), k26.c_str(), w1.c_str(),
            k19.c_str(), v1.c_str(), g2.c_str(), k26.c_str(), w2.c_str(), k31.c_str(), v2.c_str());
          f("    const double %s = (%s - EPS) / ((%s - EPS) * (%s - EPS));", r.c_str(), sc.c_str(), s1.c_str(), s2.c_str());
        } else {
          std::string dt = dot(v1, v2);
          std::string k16 = tmp("k"), k18 = tmp("k"), k19 = tmp("k");
          f("    const double %s = 1.0 / (%s * %s), %s = %s / (%s * %s * %s * %s), %s = %s / (%s * %s * %s * %s);",
            k16.c_str(), s1.c_str(), s2.c_str(), k18.c_str(), dt.c_str(), s2.c_str(), s1.c_str(), s1.c_str(), s1.c_str(),
            k19.c_str(), dt.c_str(), s1.c_str(), s2.c_str(), s2.c_str(), s2.c_str());
          f("    const double %s = %s * %s - %s * %s, %s = %s * %s - %s * %s;", g1.c_str(), k16.c_str(), v2.c_str(),
            k18.c_str(), v1.c_str(), g2.c_str(), k16.c_str(), v1.c_str(), k19.c_str(), v2.c_str());
          f("    const double %s = %s / ((%s - EPS) * (%s - EPS));", r.c_str(), dt.c_str(), s1.c_str(), s2.c_str());
        }
        ro->partial.push_back({pts[0], {g1, -1}});
        ro->partial.push_back({pts[1], {g1, 1}});
        ro->partial.push_back({pts[2], {g2, -1}});
        ro->partial.push_back({pts[3], {g2, 1}});
        return true;
      }
      case OKX_ROW_EQUAL_DISTANCE: {  // constraints.py:466-477; jacobians.py:322-367
        std::string d1 = vsub(pn(pts[1]), pn(pts[0])), d2 = vsub(pn(pts[3]), pn(pts[2]));
        std::string s1 = dot(d1, d1), s2 = dot(d2, d2);
        std::string r1 = tmp("rt"), i1 = tmp("iv"), r2 = tmp("rt"), i2 = tmp("iv");
        f("    double %s, %s, %s, %s; fast_sqrt_rsqrt(%s + EPS_SQ, &%s, &%s); fast_sqrt_rsqrt(%s + EPS_SQ, &%s, &%s);",
          r1.c_str(), i1.c_str(), r2.c_str(), i2.c_str(), s1.c_str(), r1.c_str(), i1.c_str(), s2.c_str(),
          r2.c_str(), i2.c_str());
        std::string g1 = tmp("g"), g2 = tmp("g");
        f("    const double %s = %s * %s, %s = %s * %s;", g1.c_str(), d1.c_str(), i1.c_str(), g2.c_str(), d2.c_str(),
          i2.c_str());
        f("    const double %s = (%s - EPS) - (%s - EPS);", r.c_str(), r1.c_str(), r2.c_str());
        ro->partial.push_back({pts[0], {g1, -1}});
        ro->partial.push_back({pts[1], {g1, 1}});
        ro->partial.push_back({pts[2], {g2, 1}});
        ro->partial.push_back({pts[3], {g2, -1}});
        return true;
      }
      case OKX_ROW_FIXED_AXIS: {  // constraints.py:508-516; solver.py:407-416
        const int ax = (int)P.row_param[i][0];
        std::string pa = bcast(pn(pts[0]), ax);
        f("    const double %s = %s - %s;", r.c_str(), pa.c_str(), rp(i, 1).c_str());
        ro->partial.push_back({pts[0], {"e" + std::to_string(ax), 1}});
        return true;
      }
      case OKX_ROW_POINT_ON_LINE:
      case OKX_ROW_LINE_PIN: {  // constraints.py:560-576; jacobians.py:372-403; okx.h (pin)
        std::string lp = rpv(i, 0), ld = rpv(i, 3);
        std::string cv;
        const int leader = pin_leader(i);
        if (type == OKX_ROW_LINE_PIN && pin_cross_.count(leader)) {
          cv = pin_cross_[leader];
        } else {
          std::string w = vsub(pn(pts[0]), lp);
          cv = cross(w, ld);
          if (type == OKX_ROW_LINE_PIN) pin_cross_[leader] = cv;
        }
        if (type == OKX_ROW_POINT_ON_LINE) {
          std::string c2 = dot(cv, cv);
          std::string root = tmp("rt"), inv = tmp("iv");
          f("    double %s, %s; fast_sqrt_rsqrt(EPS_SQ + %s, &%s, &%s);", root.c_str(), inv.c_str(), c2.c_str(),
            root.c_str(), inv.c_str());
          std::string gx = cross(ld, cv);
          std::string g = tmp("g");
          f("    const double %s = %s * %s;", g.c_str(), inv.c_str(), gx.c_str());
          f("    const double %s = %s - EPS;", r.c_str(), root.c_str());
          ro->partial.push_back({pts[0], {g, 1}});
          return true;
        }
        const int comp = (int)P.row_param[i][6];
        // r = e_comp . (w x ld) = w . (ld x e_comp)
        std::string g = cross(ld, "e" + std::to_string(comp));
        std::string rc = bcast(cv, comp);
        f("    const double %s = %s;", r.c_str(), rc.c_str());
        ro->partial.push_back({pts[0], {g, 1}});
        if (comp == 0) {
          // reported as the reference's single softnorm residual (constraints.py:560-576)
          std::string c2 = dot(cv, cv);
          ro->absres = "fabs(lean_sqrt(" + c2 + " + EPS_SQ) - EPS)";  // (a reported quantity: 2^-48 is plenty, an IEEE sqrt is ~20 instructions)
        } else {
          ro->absres.clear();
        }
        return true;
      }
      case OKX_ROW_POINT_ON_PLANE: {  // constraints.py:616-627; solver.py:429-437
        std::string pp = rpv(i, 0), nn = rpv(i, 3);
        std::string w = vsub(pn(pts[0]), pp);
        std::string d = dot(w, nn);
        f("    const double %s = %s;", r.c_str(), d.c_str());
        ro->partial.push_back({pts[0], {nn, 1}});
        return true;
      }
      case OKX_ROW_MIDPOINT_ON_PLANE: {  // constraints.py:657-666; solver.py:439-448
        std::string pp = rpv(i, 0), nn = rpv(i, 3);
        std::string mid = tmp("m");
        f("    const double %s = %s + (%s - %s) * 0.5;", mid.c_str(), pn(pts[0]).c_str(), pn(pts[1]).c_str(),
          pn(pts[0]).c_str());
        std::string w = vsub(mid, pp);
        std::string d = dot(w, nn);
        std::string h = tmp("h");
        f("    const double %s = 0.5 * %s;", h.c_str(), nn.c_str());
        f("    const double %s = %s;", r.c_str(), d.c_str());
        ro->partial.push_back({pts[0], {h, 1}});
        ro->partial.push_back({pts[1], {h, 1}});
        return true;
      }
      case OKX_ROW_COPLANAR:
      case OKX_ROW_SCALAR_TRIPLE: {  // constraints.py:698-709,731-733; jacobians.py:426-483
        std::string v1 = vsub(pn(pts[1]), pn(pts[0])), v2 = vsub(pn(pts[2]), pn(pts[0])),
                    v3 = vsub(pn(pts[3]), pn(pts[0]));
        std::string c23 = cross(v2, v3), c31 = cross(v3, v1), c12 = cross(v1, v2);
        std::string vol = dot(v1, c23);
        std::string g1 = c23, g2 = c31, g3 = c12;
        if (type == OKX_ROW_SCALAR_TRIPLE) {
          std::string isc = tmp("is");
          f("    const double %s = fast_rcp(%s);", isc.c_str(), rp(i, 1).c_str());
          g1 = tmp("g"), g2 = tmp("g"), g3 = tmp("g");
          f("    const double %s = %s * %s, %s = %s * %s, %s = %s * %s;", g1.c_str(), c23.c_str(), isc.c_str(),
            g2.c_str(), c31.c_str(), isc.c_str(), g3.c_str(), c12.c_str(), isc.c_str());
          f("    const double %s = (%s - %s) * %s;", r.c_str(), vol.c_str(), rp(i, 0).c_str(), isc.c_str());
        } else {
          f("    const double %s = %s;", r.c_str(), vol.c_str());
        }
        std::string g0 = tmp("g");
        f("    const double %s = -(%s + %s + %s);", g0.c_str(), g1.c_str(), g2.c_str(), g3.c_str());
        ro->partial.push_back({pts[0], {g0, 1}});
        ro->partial.push_back({pts[1], {g1, 1}});
        ro->partial.push_back({pts[2], {g2, 1}});
        ro->partial.push_back({pts[3], {g3, 1}});
        return true;
      }
      case kRowTarget: {  // solver.py:264-270, :560-579
        std::string dir = rpv(i, 0);
        std::string d = dot(pn(pts[0]), dir);
        f("    const double %s = %s - %s * tv%d;", r.c_str(), d.c_str(), target_enable(target_of_row(i)).c_str(),
          target_of_row(i));
        ro->partial.push_back({pts[0], {dir, 1}});
        return true;
      }
      default:
        why = "row type " + std::to_string(type) + " has no quad code path yet";
        return false;
    }
  }

  static std::string A(int F, int G, int k) {
    return "A" + std::to_string(F) + "_" + std::to_string(G) + "_" + std::to_string(k);
  }
  static std::string Ln(int F, int G, int k) {
    return "L" + std::to_string(F) + "_" + std::to_string(G) + "_" + std::to_string(k);
  }

  // Rows + normal equations: r_i, cost, max |r|, J^T J blocks A{F}_{G}_{k}, gradient gn{F}.
  // Residuals only (cost and max |r|): the confirming evaluation of a step that is predicted to
  // land within tolerance.  Row code is shared with emit_rows; the unused partials fold away.
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

  // pin_ata / pin_atr: every row's J^T J and J^T r contributions are made where the row is (an opaque use after the
  // row).  Left alone the optimiser sinks all of them into the branch that factors, and keeps every row's gradient and
  // its broadcasts alive until then - in AGPRs, at a v_accvgpr move per use (measured: DW corner -4 %, MacPherson -7 %,
  // axle -6 % per sweep; tools/quad_sections.py shows where a pass's instructions are).
  bool pin_ata = false, pin_atr = false;
  // OKX_QUAD_MARK=1: `s_nop 11..17` between the sections of a pass (tools/quad_sections.py counts the instructions
  // in between; scheduling barriers keep the sections apart, so the marked kernel is for counting, not for timing)
  bool marks = false;
  bool tl_marks = false;  // OKX_QUAD_TIMELINE=1: section stamps inside a pass (OKX_TL is defined per kernel body)
  void mark(int n) {
    if (marks) f("    __builtin_amdgcn_sched_barrier(0); asm volatile(\"s_nop %d\"); __builtin_amdgcn_sched_barrier(0);", 10 + n);
    if (tl_marks && n >= 5) f("    OKX_TL(%d)", n);
  }
  // jtv_rhs non-empty: only J^T v is formed, for every listed right-hand side q (v_i = the variable {q}{i}, result
  // {q}g{F}): the second Jacobian pass of the first-step table's second-order terms (okx_quad_head_*).
  std::vector<std::string> jtv_rhs;
  std::function<std::string(int, int)> jtv_value;  // (rhs index, row) -> expression of v_i, read where the row is
  bool jtv_only = true;                            // false: the normal evaluation AND the extra J^T v accumulations
  bool emit_rows() {
    const int nf = P.n_free;
    const bool jtv = !jtv_rhs.empty() && jtv_only;
    for (int F = 0; F < nf; ++F) {
      for (const std::string& q : jtv_rhs) f("    double %sg%d = 0.0;", q.c_str(), F);
      if (!jtv) f("    double gn%d = 0.0;", F);
      nz[F][F] = true;
    }
    if (!jtv) f("    double ss = 0.0, mres_new = 0.0;");
    std::set<std::string> declared;
    for (int i = 0; i < P.m; ++i) {
      RowOut ro;
      mark(1);
      if (!row(i, &ro)) return false;
      if (!jtv) {
      f("    ss = fma(%s, %s, ss);", ro.r.c_str(), ro.r.c_str());
      if (!ro.absres.empty()) f("    mres_new = fmax(mres_new, %s);", ro.absres.c_str());
      }
      mark(2);
      // point partials -> free blocks
      std::map<int, std::vector<LV>> terms;
      for (auto& pp : ro.partial) {
        const int pt = pp.first;
        const LV& gp = pp.second;
        if (blk_of_point[pt] >= 0) {
          terms[blk_of_point[pt]].push_back(gp);
        } else if (dop_of_point[pt] >= 0) {
          auto it = dblk.find(dop_of_point[pt]);
          if (it == dblk.end()) {
            why = "row reads a derived point without chain blocks";
            return false;
          }
          for (auto& kv : it->second) {
            const Blk& b = kv.second;
            std::string t = tmp("j");
            if (b.scaled) {
              f("    const double %s = (%s) * %s;", t.c_str(), b.s.c_str(), gp.n.c_str());
            } else {  // point_partial @ block (solver.py:554-558): lane c forms column c
              std::string b0 = bcast(gp.n, 0), b1 = bcast(gp.n, 1), b2 = bcast(gp.n, 2);
              f("    const double %s = %s * %s + %s * %s + %s * %s;", t.c_str(), b0.c_str(), b.col[0].c_str(),
                b1.c_str(), b.col[1].c_str(), b2.c_str(), b.col[2].c_str());
            }
            terms[kv.first].push_back({t, gp.sg});
          }
        }
      }
      std::vector<std::pair<int, LV>> jv;
      for (auto& kv : terms) {
        if (kv.second.size() == 1) {
          jv.push_back({kv.first, kv.second[0]});
        } else {
          std::string e;
          for (auto& t : kv.second) {
            if (!e.empty()) e += t.sg < 0 ? " - " : " + ";
            else if (t.sg < 0) e += "-";
            e += t.n;
          }
          std::string t = tmp("j");
          f("    const double %s = %s;", t.c_str(), e.c_str());
          jv.push_back({kv.first, {t, 1}});
        }
      }
      for (size_t qi = 0; qi < jtv_rhs.size(); ++qi) {
        const std::string& q = jtv_rhs[qi];
        if (jtv_value) f("    const double %s%d = %s;", q.c_str(), i, jtv_value((int)qi, i).c_str());
        for (auto& fv : jv) f("    %sg%d = fma(%s, %s%d, %sg%d);", q.c_str(), fv.first, sx(fv.second).c_str(), q.c_str(), i, q.c_str(), fv.first);
      }
      if (jtv) continue;
      if (P.row_type[i] == kRowTarget) target_j[(int)P.row_param[i][3]] = jv;
      // J^T r and J^T J (lower block triangle: F >= G)
      for (auto& fv : jv) f("    gn%d = fma(%s, %s, gn%d);", fv.first, sx(fv.second).c_str(), ro.r.c_str(), fv.first);
      if (pin_atr)
        for (auto& fv : jv) f("    asm volatile(\"\" : \"+v\"(gn%d));", fv.first);
      mark(3);
      std::vector<std::string> touched;
      for (size_t ia = 0; ia < jv.size(); ++ia)
        for (size_t ib = 0; ib <= ia; ++ib) {
          const int F = jv[ia].first, G = jv[ib].first;
          const LV& jF = jv[ia].second;
          const LV& jG = jv[ib].second;
          nz[F][G] = true;
          for (int k = 0; k < 3; ++k) {
            std::string b = bcast(jG.n, k);
            const int sg = jF.sg * jG.sg;
            std::string an = A(F, G, k);
            if (!declared.count(an)) {
              declared.insert(an);
              f("    double %s = %s%s * %s;", an.c_str(), sg < 0 ? "-" : "", jF.n.c_str(), b.c_str());
            } else {
              f("    %s = fma(%s%s, %s, %s);", an.c_str(), sg < 0 ? "-" : "", jF.n.c_str(), b.c_str(), an.c_str());
            }
            touched.push_back(an);
          }
        }
      if (pin_ata)
        for (auto& an : touched) f("    asm volatile(\"\" : \"+v\"(%s));", an.c_str());
      mark(4);
    }
    // diagonal blocks that no row touched still exist (as zeros)
    for (int F = 0; F < nf && !jtv; ++F)
      for (int k = 0; k < 3; ++k)
        if (!declared.count(A(F, F, k))) f("    double %s = 0.0;", A(F, F, k).c_str());
    return true;
  }

  // LDL^T of (J^T J + lambda I), forward / diagonal / backward substitution -> dx{F}.
  void emit_solve() {
    mark(5);
    emit_factor();
    mark(6);
    // right-hand side -g, result nx{F}
    std::vector<std::string> rhs;
    for (int F = 0; F < P.n_free; ++F) rhs.push_back("-gn" + std::to_string(F));
    emit_substitute(rhs, "nx");
    mark(7);
  }

  // LDL^T of (J^T J + lambda I) in registers; leaves L{F}_{G}_{k}, dinv{F}, ok, pmin, pmax.
  void emit_factor() {
    const int nf = P.n_free;
    bool fill[kMaxFree][kMaxFree];
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G < nf; ++G) fill[F][G] = nz[F][G];
    f("    // ---- damped normal equations: LDL^T, lane c owns rows 3F+c ----");
    for (int F = 0; F < nf; ++F)
      for (int k = 0; k < 3; ++k) f("    %s = fma(lambda, e%d, %s);", A(F, F, k).c_str(), k, A(F, F, k).c_str());
    f("    bool ok = true;");
    f("    double pmin = 1e300, pmax = 0.0;  // smallest / largest pivot (conditioning of the tangent solve)");
    for (int F = 0; F < nf; ++F) f("    double dinv%d = 0.0;", F);
    for (int G = 0; G < nf; ++G)
      for (int k = 0; k < 3; ++k) {
        f("    { // column %d", 3 * G + k);
        f("    const double piv = QB%d(%s);", k, A(G, G, k).c_str());
        f("    ok = ok && piv > 0.0;  // a failed factor is never used: no need to sanitise the pivot");
        f("    pmin = fmin(pmin, piv); pmax = fmax(pmax, piv);");
        f("    const double rinv = pivot_rcp(piv);");
        f("    dinv%d = fma(e%d, rinv, dinv%d);", G, k, G);
        // factor entries of this column (rows below the pivot)
        if (k < 2) f("    %s = c > %d ? %s * rinv : 0.0;", Ln(G, G, k).c_str(), k, A(G, G, k).c_str());
        for (int F = G + 1; F < nf; ++F)
          if (fill[F][G]) f("    %s = %s * rinv;", Ln(F, G, k).c_str(), A(F, G, k).c_str());
        // trailing update: A[v][u] -= L[v][w] * A[u][w]
        for (int H = G; H < nf; ++H) {
          if (H > G && !fill[H][G]) continue;
          for (int j = (H == G ? k + 1 : 0); j < 3; ++j) {
            f("    { const double cu = QB%d(%s);", j, A(H, G, k).c_str());
            for (int F = H; F < nf; ++F) {
              if (F > G && !fill[F][G]) continue;
              if (F == G && k == 2) continue;
              fill[F][H] = true;  // fill-in: the block becomes structurally non-zero
              f("      %s = fma(-%s, cu, %s);", A(F, H, j).c_str(), Ln(F, G, k).c_str(), A(F, H, j).c_str());
            }
            f("    }");
          }
        }
        f("    }");
      }
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G < nf; ++G) fillf[F][G] = fill[F][G];
  }

  // Forward / diagonal / backward substitution with the factor in registers:
  // {out}{F} = (J^T J + lambda I)^-1 rhs[F]   (rhs: expression per free block, lane component).
  // `last_block_hook` (may be null): text placed right after block nf - 1 of the backward pass is finished - {out}{nf-1} then
  // holds the last block of the solution and may still be corrected before the earlier blocks are substituted from it.
  void emit_substitute(const std::vector<std::string>& rhs, const char* out, const std::string* last_block_hook = nullptr) {
    const int nf = P.n_free;
    f("    // ---- L y = rhs (unit lower, block by block) ----");
    for (int F = 0; F < nf; ++F) f("    double y%d = %s;", F, rhs[F].c_str());
    for (int G = 0; G < nf; ++G) {
      f("    { const double yb0 = QB0(y%d); y%d = fma(-%s, yb0, y%d);", G, G, Ln(G, G, 0).c_str(), G);
      f("      const double yb1 = QB1(y%d); y%d = fma(-%s, yb1, y%d);", G, G, Ln(G, G, 1).c_str(), G);
      f("      const double yb2 = QB2(y%d);", G);
      for (int F = G + 1; F < nf; ++F)
        if (fillf[F][G])
          f("      y%d = fma(-%s, yb0, fma(-%s, yb1, fma(-%s, yb2, y%d)));", F, Ln(F, G, 0).c_str(), Ln(F, G, 1).c_str(),
            Ln(F, G, 2).c_str(), F);
      f("    }");
    }
    f("    // ---- D z = y, L^T x = z ----");
    for (int G = nf - 1; G >= 0; --G) {
      f("    double %s%d = y%d * dinv%d;", out, G, G, G);
      bool any = false;
      for (int F = G + 1; F < nf; ++F) any = any || fillf[F][G];
      if (any) {
        f("    { double s0 = 0.0, s1 = 0.0, s2 = 0.0;");
        for (int F = G + 1; F < nf; ++F)
          if (fillf[F][G])
            f("      s0 = fma(%s, %s%d, s0); s1 = fma(%s, %s%d, s1); s2 = fma(%s, %s%d, s2);", Ln(F, G, 0).c_str(), out, F,
              Ln(F, G, 1).c_str(), out, F, Ln(F, G, 2).c_str(), out, F);
        f("      s0 = qsum(s0); s1 = qsum(s1); s2 = qsum(s2);");
        f("      %s%d = fma(-e0, s0, fma(-e1, s1, fma(-e2, s2, %s%d))); }", out, G, out, G);  // lane 3 stays 0
      }
      f("    { const double xb2 = QB2(%s%d), l21 = QB2(%s), l20 = QB2(%s);", out, G, Ln(G, G, 1).c_str(),
        Ln(G, G, 0).c_str());
      f("      %s%d = fma(-fma(e1, l21, e0 * l20), xb2, %s%d);", out, G, out, G);
      f("      const double xb1 = QB1(%s%d), l10 = QB1(%s);", out, G, Ln(G, G, 0).c_str());
      f("      %s%d = fma(-(e0 * l10), xb1, %s%d); }", out, G, out, G);
      if (G == nf - 1 && last_block_hook) this->out += *last_block_hook;
    }
  }
  // D~_FF-only solve of a right-hand side that lives in the LAST block alone (everything before it is zero, nothing comes
  // after it): the in-block forward, diagonal and backward steps of emit_substitute for block nf - 1, result in `name`.
  std::string last_block_solve(const std::string& rhs, const std::string& name) {
    const int G = P.n_free - 1;
    std::string t;
    char line[512];
    auto add = [&](const char* fmt, auto... args) { std::snprintf(line, sizeof(line), fmt, args...); t += line; t += "\n"; };
    add("    double %s = %s;", name.c_str(), rhs.c_str());
    add("    { const double zb0 = QB0(%s); %s = fma(-%s, zb0, %s);", name.c_str(), name.c_str(), Ln(G, G, 0).c_str(), name.c_str());
    add("      const double zb1 = QB1(%s); %s = fma(-%s, zb1, %s); }", name.c_str(), name.c_str(), Ln(G, G, 1).c_str(), name.c_str());
    add("    %s = %s * dinv%d;", name.c_str(), name.c_str(), G);
    add("    { const double xb2 = QB2(%s), l21 = QB2(%s), l20 = QB2(%s);", name.c_str(), Ln(G, G, 1).c_str(), Ln(G, G, 0).c_str());
    add("      %s = fma(-fma(e1, l21, e0 * l20), xb2, %s);", name.c_str(), name.c_str());
    add("      const double xb1 = QB1(%s), l10 = QB1(%s);", name.c_str(), Ln(G, G, 0).c_str());
    add("      %s = fma(-(e0 * l10), xb1, %s); }", name.c_str(), name.c_str());
    return t;
  }

  // Directional derivative of one derived op (forward mode, closed form): velocity of the output
  // point from the velocities `vn(p)` of its inputs, at the positions p{k}.  Replaces the
  // reference's dual-number pass (sensitivity.py:127-131, primitives/dual.py).
  bool derived_jvp(int e, const std::string& vp) {
    const int type = P.dop_type[e];
    const int* pts = P.dop_pts[e];
    auto vn = [&](int p) { return vp + std::to_string(p); };
    const std::string o = vn(P.dop_out[e]);
    if (type == OKX_DOP_MIDPOINT) {
      f("    const double %s = %s + (%s - %s) * 0.5;", o.c_str(), vn(pts[0]).c_str(), vn(pts[1]).c_str(), vn(pts[0]).c_str());
      return true;
    }
    if (type == OKX_DOP_ALONG) {  // out = base + c u, u = w / |w|, w = a - b
      std::string w = vsub(pn(pts[1]), pn(pts[2]));
      std::string dw = vsub(vn(pts[1]), vn(pts[2]));
      std::string ww = dot(w, w);
      std::string nrm = tmp("nr"), inrm = tmp("in"), u = tmp("u");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", nrm.c_str(), inrm.c_str(), ww.c_str(), nrm.c_str(), inrm.c_str());
      f("    const double %s = %s * %s;", u.c_str(), w.c_str(), inrm.c_str());
      std::string ud = dot(u, dw);
      f("    const double %s = %s + %s * %s * (%s - %s * %s);", o.c_str(), vn(pts[0]).c_str(), dp(e).c_str(), inrm.c_str(),
        dw.c_str(), u.c_str(), ud.c_str());
      return true;
    }
    if (type == OKX_DOP_CONTACT_PATCH) {  // out = wc + R wu, wu = wd / |wd|, wd = -ga ax - e_z, ga = -ax_z
      std::string v = vsub(pn(pts[2]), pn(pts[1]));
      std::string dv = vsub(vn(pts[2]), vn(pts[1]));
      std::string vv = dot(v, v);
      std::string vnm = tmp("vn"), ivn = tmp("iv"), ax = tmp("ax");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", vnm.c_str(), ivn.c_str(), vv.c_str(), vnm.c_str(), ivn.c_str());
      f("    const double %s = %s * %s;", ax.c_str(), v.c_str(), ivn.c_str());
      std::string adv = dot(ax, dv);
      std::string dax = tmp("da");
      f("    const double %s = %s * (%s - %s * %s);", dax.c_str(), ivn.c_str(), dv.c_str(), ax.c_str(), adv.c_str());
      std::string az = bcast(ax, 2), daz = bcast(dax, 2);
      std::string wd = tmp("wd"), dwd = tmp("dw");
      f("    const double %s = %s * %s - e2;", wd.c_str(), az.c_str(), ax.c_str());
      f("    const double %s = %s * %s + %s * %s;", dwd.c_str(), daz.c_str(), ax.c_str(), az.c_str(), dax.c_str());
      std::string ww = dot(wd, wd);
      std::string wn = tmp("wn"), iwn = tmp("iw"), wu = tmp("wu");
      f("    double %s, %s; fast_sqrt_rsqrt(%s, &%s, &%s);", wn.c_str(), iwn.c_str(), ww.c_str(), wn.c_str(), iwn.c_str());
      f("    const double %s = %s * %s;", wu.c_str(), wd.c_str(), iwn.c_str());
      std::string wdw = dot(wu, dwd);
      f("    const double %s = %s + %s * %s * (%s - %s * %s);", o.c_str(), vn(pts[0]).c_str(), dp(e).c_str(), iwn.c_str(),
        dwd.c_str(), wu.c_str(), wdw.c_str());
      return true;
    }
    why = "unknown derived op";
    return false;
  }

  std::map<int, std::vector<std::pair<int, LV>>> target_j;  // target index -> (free block, d r / d block)

  bool fillf[kMaxFree][kMaxFree] = {};
};

const char* kPreamble = R"SRC(
// Generated by okx_quadgen.cpp for one constraint program — do not edit.
typedef struct { double max_residual, cost, last_step; int iterations, nfev, flags, reserved; } okx_info;
struct QArgs {
  const double* targets; const double* geom_pos; const double* geom_row_param;
  double* out_pos; okx_info* info;
  long long n_problems, steps_per_geometry, chain_len;
  int max_iter, confirm;   // confirm != 0: never end on the predicted-convergence test (always a full pass)
  double step_tol, grad_tol, ftol, lambda0, residual_tolerance;
  const double* design_pos; const double* row_param; const double* dop_param;
  double* trace; long long trace_problem;   // diagnostic: 8 doubles per LM pass of one problem (null: off)
  const double* predictor;  // polynomial model of the solution over the fitted target range for chain heads (null: off)
  long long predictor_mode; // 2: every chain step starts from the model, not only the heads
  long long predictor_len;  // doubles in the table
  const double* head;       // per-geometry first-step table of okx_quad_head_u/_g (null: every chain head takes its own first pass)
  long long out_mode;       // okx_solve_opts.output: 0 records of every output point, 1 the free points only, 2 nothing
};
#define EPS_SQ 1e-12
#define EPS 1e-6
#define DEV __device__ __forceinline__
// lane component load: lanes 0..2 read their component, lane 3 (address clamped to component 2 by the caller)
// gets zero.  The load itself is unconditional: no exec-mask branch per value.
DEV double ld3(const double* p, int c) { const double v = *p; return c < 3 ? v : 0.0; }
#define INFO_CONVERGED 1
#define INFO_RESIDUAL_EXCEEDED 2
#define INFO_FAILED 4
#define INFO_ILL_CONDITIONED 8
#define ILL_CONDITIONED_PIVOT_RATIO 1e-12

// DPP quad_perm of a double (2 x v_mov_b32_dpp): lane l of every quad reads lane sel[l].
// (mov_dpp, not update_dpp: every lane of a quad_perm has a valid source, so there is no "old"
// value to preserve and no v_mov to initialise it.)
// (The same permutations through the LDS crossbar - ds_swizzle_b32 in quad-permute mode, which issues on the LDS pipe
// instead of the vector ALU - were measured: only the J^T J accumulation's broadcasts 28.9 vs 28.5 us on C2, all of
// them 39.9 us.  One wavefront per SIMD cannot hide the crossbar's latency.)
template <int CTRL> DEV double qperm(double v) {
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define QB0(v) qperm<0x00>(v)   /* broadcast lane 0 of the quad */
#define QB1(v) qperm<0x55>(v)
#define QB2(v) qperm<0xAA>(v)
#define QR1(v) qperm<0xC9>(v)   /* lane c reads component (c+1)%3; lane 3 reads itself */
#define QR2(v) qperm<0xD2>(v)   /* lane c reads component (c+2)%3 */
// Sum / max over the quad, BIT-IDENTICAL in its four lanes: a butterfly of commutative adds.
// Contraction must stay off here: fusing the caller's product into the first add
// (fma(a, b, neighbour)) would round differently in the two lanes of a pair, and every
// per-problem decision (accept / reject / stop) relies on the four lanes agreeing exactly.
DEV double qsum(double v) {
#pragma clang fp contract(off)
  double w = v + qperm<0xB1>(v);
  w = w + qperm<0x4E>(w);
  return w;
}
DEV double qmax(double v) { v = fmax(v, qperm<0xB1>(v)); v = fmax(v, qperm<0x4E>(v)); return v; }

DEV double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  e = fma(-x, r, 1.0);
  return fma(e, r, r);
}
// v_rcp_f64 / v_rsq_f64 deliver 2^-24.3 (measured on MI355X over 2^20 arguments, profiles/r02/README.md).
// Reciprocal of a pivot: ONE Newton step (2^-48.6).  The factor only steers the Levenberg-Marquardt step (and the
// tangents to 1e-9): a 2e-15 relative error in L is far inside what the damping already does to it.
DEV double pivot_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
// sqrt(x) to the last bit or so and 1 / sqrt(x) to 4e-15: one Goldschmidt step (both to ~1.5 * 2^-48.4), then the
// residual correction of the root (x - g^2 is exact in the fma), which squares its error.
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
// sqrt(x), x > 0 and normal, to ~2^-48: one Goldschmidt step (values that are only reported).
DEV double lean_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y, h = 0.5 * y;
  return fma(g, fma(-h, g, 0.5), g);
}
// atan2(y, x), y >= 0, result in [0, pi]: fdlibm-style reduction + odd polynomial (see okx_kernels.hip).
// In the register-bound pair kernels (TAB) the eleven polynomial coefficients are READ WHERE THEY ARE USED, from a table in LDS (`tab`, filled once per
// workgroup from kAtanCoef; every lane reads the same address: a broadcast), through an index the optimiser cannot see
// through.  As literals they are loop invariants: the compiler materialises them ahead of the Levenberg-Marquardt loop,
// runs out of scalar registers, parks them in vector registers and - in the register-bound pair kernel - spills them
// to scratch, from where every angle row then re-reads them one dependent load at a time (profiles/r02/c3_*).
__constant__ double kAtanCoef[12] = {
    3.33333333333329318027e-01, 1.42857142725034663711e-01, 9.09088713343650656196e-02, 6.66107313738753120669e-02,
    4.97687799461593236017e-02, 1.62858201153657823623e-02,
    -1.99999999998764832476e-01, -1.11111104054623557880e-01, -7.69187620504482999495e-02, -5.83357013379057348645e-02,
    -3.65315727442169155270e-02, 0.0};
template <bool TAB> DEV double lean_atan2_pos(double y, double x, const double* tab) {
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
  double s1, s2;
  if (TAB) {
    int k0 = 0; asm volatile("" : "+v"(k0));
    const double* ct = tab + k0;
    s1 = z * (ct[0] + w * (ct[1] + w * (ct[2] + w * (ct[3] + w * (ct[4] + w * ct[5])))));
    s2 = w * (ct[6] + w * (ct[7] + w * (ct[8] + w * (ct[9] + w * ct[10]))));
  } else {  // kernels with registers to spare keep the coefficients as literals
    s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
         w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
         w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  }
  const double at = hi - ((t * (s1 + s2) - lo) - t);
  return x > 0.0 ? at : 3.14159265358979311600e+00 - (at - 1.2246467991473531772e-16);
}
DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
)SRC";

}  // namespace

int quad_head_stride(const DevProgram& program) {
  // mirrors the layout quad_generate() gives the first-step table (head_cols / head_off / head_stride there)
  if (program.n_free <= kQuadMaxFree) {
    const int k = program.n_targets + 1, pairs = program.n_targets * (program.n_targets + 1) / 2;
    return 4 * program.n_free * k + 2 * k * k + 8 + 4 * program.n_free * pairs;  // Q, Gram matrices, scalars, second-order S
  }
  PairView pv;
  std::string why;
  if (!build_pair_view(program, &pv, &why)) return 0;
  const int k = pv.n_prog_targets + 1, pairs = pv.n_prog_targets * (pv.n_prog_targets + 1) / 2;
  return 2 * 4 * pv.side.n_free * k + 2 * k * k + 8 + 2 * 4 * pv.side.n_free * pairs;  // both halves' Q and S blocks
}

bool quad_generate(const DevProgram& program, int waves_per_simd, std::string* src, std::string* why, bool lds_homes,
                   const EvalSpec* es, const AxleEvalSpec* aes) {
  // Small programs: one quad per problem.  Larger ones only when they are two identical halves
  // joined by one distance row (composed axle): one quad per half, a 2 x 2 Woodbury correction for the joint.
  PairView pair_store;
  const PairView* pv = nullptr;
  // EV: the evaluated module (okx_solve_evaluated_batch) - the solve bodies end in an epilogue that solves for the
  // solution-manifold tangents at the converged state and evaluates the metric catalog along them (okx_evalsrc.cpp)
  // EVP: the evaluated module of a composed axle (pair mode): both corners' catalogs, the axle-scope metrics and the roles
  const bool EVP = aes != nullptr;
  if (EVP) es = &aes->side[0];  // (the catalog's compile-time switches: the same for both corners, axle_eval_spec_from_roles)
  const bool EV = es != nullptr;
  if (EV && !EVP && program.n_free > kQuadMaxFree) {
    *why = "a pair-mode program takes its metric roles through okx_program_enable_axle_evaluation (both corners')";
    return false;
  }
  if (EVP && program.n_free <= kQuadMaxFree) {
    *why = "axle roles need a pair-mode program (two identical halves)";
    return false;
  }
  if (program.n_free > kQuadMaxFree) {
    std::string pair_why;
    if (!build_pair_view(program, &pair_store, &pair_why) || pair_store.side.n_free > kQuadMaxFreePerSide) {
      *why = "more than " + std::to_string(kQuadMaxFree) + " free points and not a pair of identical halves (" +
             (pair_why.empty() ? "halves too large" : pair_why) + ")";
      return false;
    }
    pv = &pair_store;
  }
  const DevProgram& P = pv ? pv->side : program;
  if (P.n_targets > kMaxTargets) {
    *why = "too many targets";
    return false;
  }
  Gen g(P, pv);
  const int nf = P.n_free, NP = P.n_points, T = P.n_targets;
  const int PPW = pv ? 8 : 16;                                  // problems per wavefront
  const int prog_points = pv ? pv->n_prog_points : NP;          // strides of the caller's tables
  const int prog_crows = pv ? pv->n_prog_crows : P.n_crows;
  const int prog_targets = pv ? pv->n_prog_targets : T;
  const int prog_out = pv ? pv->n_prog_out : P.n_out;

  // ---- evaluation body (rows + normal equations), generated first to learn the sparsity ----
  Gen ev(P, pv);
  // Pair mode: the accepted point, the chain history and the step in hand live in LDS (`lds_constants` switches that
  // layout on); the chain constants and the fixed points stay in registers unless the caller asks for LDS homes (the
  // fallback of quad_build for half programs that would spill) - measured on the axle grid: -8.5 % chained, -10 % cold
  // against everything in LDS (130 -> 60 LDS round trips per pass, which a lone wavefront cannot hide).
  if (dev_switch("pair_lds_homes")) lds_homes = true;  // (tests: the fallback layout of a pair program that would spill)
  ev.lds_constants = pv != nullptr;
  ev.scalars_in_regs = ev.lanes_in_regs = pv != nullptr && !lds_homes;
  const bool fixed_in_regs = pv != nullptr && !lds_homes;
  ev.pin_ata = ev.pin_atr = true;
  ev.marks = dev_switch("quad_mark");
  ev.tl_marks = dev_switch("quad_timeline");
  for (int e = 0; e < P.n_derived; ++e) ev.dp(e);  // every derived-op parameter is chain-constant
  ev.f("    // ---- active derived points with chain-rule blocks ----");
  for (int idx = 0; idx < P.n_active; ++idx)
    if (!ev.derived_op(P.active_op[idx], true)) {
      *why = ev.why;
      return false;
    }
  ev.mark(8);  // (derived points done)
  if (!ev.emit_rows()) {
    *why = ev.why;
    return false;
  }
  ev.mark(9);  // (rows done)
  std::string eval_src = ev.out;
  ev.out.clear();
  ev.emit_solve();
  std::string solve_src = ev.out;

  // confirming evaluation (residuals only).  Not offered for programs with the reference's
  // zero-gradient point-on-line row: along that row's valley the step length says nothing about
  // the distance to the minimiser, so those always take full passes.
  bool light_ok = !dev_switch("quad_no_light");  // (developer switch: a kernel without the residual-only confirming pass, for instruction counts)
  for (int i = 0; i < P.n_crows; ++i) light_ok = light_ok && P.row_type[i] != OKX_ROW_POINT_ON_LINE;
  std::string light_src;
  if (light_ok) {
    Gen lt(P, pv);
    lt.uid = 300000;
    lt.hoisted_names = ev.hoisted_names;  // same chain-constant loads, already emitted
    for (int idx = 0; idx < P.n_active; ++idx)
      if (!lt.derived_op(P.active_op[idx], false)) light_ok = false;
    if (light_ok && !lt.emit_rows_residual_only()) light_ok = false;
    if (!lt.hoisted.empty()) light_ok = false;  // would need loads the main body did not hoist
    light_src = lt.out;
  }

  // Shared first step of the chain heads (single mode): layout of one geometry's table, in doubles:
  // Q[k][F][4] (lane components, 0 in slot 3), M[j][k] = Q_j . G_k, N[j][k] = Q_j . Q_k, then
  // dmax, min pivot, sum of squared constraint residuals, max |constraint residual|, ok, max pivot, 0, 0.
  // Column k = 0 is the constraint rows' own gradient G_0 = Jc^T rc at the design state (the reference's distance
  // rows carry softnorm's -1e-6 offset there, constraints.py:125-134, so it is small but not zero) with weight 1;
  // column k = t + 1 belongs to target t: G_k = J^T e_t, weight = that target's residual.  Q_k = (J^T J + lambda I)^-1 G_k.
  // Pair mode carries the first-order table (OKX_PAIR_NO_HEAD=1 leaves it out).  Measured on the axle grid, round 3:
  // cold starts 5.57 -> 4.72 evaluations, 0.483 -> 0.455 ms; chained grids unchanged (0.211 vs 0.212 ms) - in round 2
  // the block's registers still cost the chained grid 3 %, before the LM scalars and constants had homes in LDS.
  const bool head_ok = T >= 1 && !dev_switch("quad_no_head") && (!pv || !dev_switch("pair_no_head"));
  // columns of the table: the constraint gradient, then one per PROGRAM target (pair mode: a side target stands for one
  // program target per half that carries it; the column's weight is that half's residual, its Q spans both halves)
  struct HeadCol { int t, side, prog_t; };
  std::vector<HeadCol> head_cols;
  head_cols.push_back({-1, -1, -1});
  for (int pt = 0; pt < prog_targets; ++pt)
    for (int t = 0; t < T; ++t) {
      if (!pv) { if (t == pt) head_cols.push_back({t, -1, pt}); continue; }
      for (int sd = 0; sd < 2; ++sd)
        if (pv->tgt[sd][t] == pt) head_cols.push_back({t, sd, pt});
    }
  const int HK = (int)head_cols.size();
  const int head_side = 4 * nf * HK;                       // doubles of one half's Q block
  const int head_off = (pv ? 2 : 1) * head_side + 2 * HK * HK;
  // Second-order terms of the shared first step: S_st = (J^T J + lambda I)^-1 J^T r''(Q_s, Q_t) for the target
  // columns s <= t, [pair][F][4] after the scalars (pair mode: one such block per half, the left half's first);
  // scalar 6 says how many pairs the table carries.
  std::vector<std::pair<int, int>> head_pairs;
  if (!(pv && dev_switch("pair_first_order_head")))
    for (int s2 = 1; s2 < HK; ++s2)
      for (int t2 = s2; t2 < HK; ++t2) head_pairs.push_back({s2, t2});
  const int NPAIR = (int)head_pairs.size();
  const int head_s_off = head_off + 8;
  const int head_s_side = 4 * nf * (HK - 1) * HK / 2;      // doubles of one half's S block
  const int head_stride = head_off + 8 + (pv ? 2 : 1) * head_s_side;
  bool has_atan = false;
  for (int i = 0; i < P.n_crows; ++i)
    has_atan = has_atan || P.row_type[i] == OKX_ROW_ANGLE || P.row_type[i] == OKX_ROW_THREE_POINT_ANGLE;
  const std::string atan_decl = !(has_atan && pv) ? std::string() :
      "  __shared__ double atl[12];  // atan2's polynomial coefficients, read at their use (lean_atan2_pos)\n"
      "  if (threadIdx.x < 12) atl[threadIdx.x] = kAtanCoef[threadIdx.x];\n"
      "  __syncthreads();\n";
  // Pair mode: the quad-uniform Levenberg-Marquardt scalars and the chain's target history live in LDS, one slot per
  // quad side like the chain constants (all lanes of a quad write the same value).  Left in registers the compiler
  // spills them to scratch, and the pass re-reads ~60 of them from there at memory latency (profiles/r02/c3_*).
  bool pair_state_lds = pv != nullptr;  // (set per body at the top of emit_body)
  int n_state_slots = 0;
  auto state_ref = [&](const std::string& name, const std::string& init) {
    // declaration of one per-quad scalar: a register, or a reference into lms[slot][quad side]
    if (!pair_state_lds) return "double " + name + " = " + init + ";";
    return "double& " + name + " = lms[" + std::to_string(16 * n_state_slots++) + " + qs]; " + name + " = " + init + ";";
  };

  // ---- pair mode: the row joining the two halves (distance between a point and its mirror image) ----
  // Each side sees d = partner - own; residual and cost terms are bit-identical on both sides
  // (squares of opposite-signed differences, commutative sums), the Jacobian entries live in the
  // joined point's block only; each half adds its own part of the rank-one term and a 2 x 2 Woodbury system the rest.
  std::string couple_eval, couple_light, couple_hoist;
  int FU = -1;
  const int NK = pv ? (int)pv->joins.size() : 0;  // rows joining the halves
  std::vector<int> FUj;                            // their joined point's block
  for (int j = 0; j < NK; ++j) FUj.push_back(ev.blk_of_point[pv->joins[j].point]);
  if (pv && NK > 1) {
    // k joining rows (T-bar axle: rack length, crossbar length, crossbar midpoint on the centre plane): residual rc{j},
    // gradient in the joined point's block cu{j} (lane component), both bit-identical on the two halves.
    FU = FUj[0];
    char buf[1024];
    std::string resid, grads;
    for (int j = 0; j < NK; ++j) {
      const PairView::Join& join = pv->joins[j];
      if (join.type == OKX_ROW_DISTANCE) {
        const std::string nm = "hcL" + std::to_string(j);
        const std::string home = ev.scalar_home(nm.c_str());
        std::snprintf(buf, sizeof(buf), "      %s%s = gq[%d];  // length of joining row %d\n", ev.sdecl(), home.c_str(), 8 * join.row, j);
        couple_hoist += buf;
        if (home != nm) couple_hoist += "#define " + nm + " " + home + "\n";
        std::snprintf(buf, sizeof(buf),
                      "    const double cd%d = xq(p%d) - p%d;\n"
                      "    const double cs%d = qsum(cd%d * cd%d);\n"
                      "    double crt%d, cinv%d; fast_sqrt_rsqrt(cs%d + EPS_SQ, &crt%d, &cinv%d);\n"
                      "    const double rc%d = (crt%d - EPS) - %s;\n",
                      j, join.point, join.point, j, j, j, j, j, j, j, j, j, j, nm.c_str());
        resid += buf;
        std::snprintf(buf, sizeof(buf), "    const double cu%d = -cd%d * cinv%d;  // d rc%d / d (own joined point), lane component\n", j, j, j, j);
        grads += buf;
      } else {  // midpoint of the pair on a plane: r = n . (a + (b - a) / 2 - p0), a = the first half's point (constraints.py:657-666)
        std::snprintf(buf, sizeof(buf), "      const double hcN%d = ld3(gq + %d + cc, c), hcP%d = ld3(gq + %d + cc, c);  // plane normal / point of joining row %d\n",
                      j, 8 * join.row + 3, j, 8 * join.row, j);
        couple_hoist += buf;
        std::snprintf(buf, sizeof(buf),
                      "    const double cpx%d = xq(p%d);  // (exchanged by every lane: never inside a select on the side bit)\n"
                      "    const double cpa%d = q1 ? cpx%d : p%d, cpb%d = q1 ? p%d : cpx%d;\n"
                      "    const double rc%d = qsum(hcN%d * ((cpa%d + (cpb%d - cpa%d) * 0.5) - hcP%d));\n",
                      j, join.point, j, j, join.point, j, join.point, j, j, j, j, j, j, j);
        resid += buf;
        std::snprintf(buf, sizeof(buf), "    const double cu%d = 0.5 * hcN%d;  // d rc%d / d (own joined point), lane component\n", j, j, j);
        grads += buf;
      }
    }
    std::string sumsq = "(ss + xq(ss))", maxr = "fmax(mres_new, xq(mres_new))";
    for (int j = 0; j < NK; ++j) {
      sumsq = "(" + sumsq + " + rc" + std::to_string(j) + " * rc" + std::to_string(j) + ")";
      maxr = "fmax(" + maxr + ", fabs(rc" + std::to_string(j) + "))";
    }
    const std::string tail = "    ss = " + sumsq + ";\n    mres_new = " + maxr + ";\n";
    couple_light = resid + tail;
    couple_eval = resid + grads;
    for (int j = 0; j < NK; ++j) {
      std::snprintf(buf, sizeof(buf), "    gn%d = fma(cu%d, rc%d, gn%d);\n", FUj[j], j, j, FUj[j]);
      couple_eval += buf;
    }
    couple_eval += tail;
  } else if (pv) {
    FU = ev.blk_of_point[pv->couple_point];
    char buf[1024];
    const std::string hcl_home = ev.scalar_home("hcL");
    std::snprintf(buf, sizeof(buf), "      %s%s = gq[%d];  // length of the joining row\n", ev.sdecl(), hcl_home.c_str(), 8 * pv->couple_row);
    couple_hoist = buf;
    if (hcl_home != "hcL") couple_hoist += "#define hcL " + hcl_home + "\n";
    std::snprintf(buf, sizeof(buf),
                  "    const double cd = xq(p%d) - p%d;\n"
                  "    const double cs = qsum(cd * cd);\n"
                  "    double crt, cinv; fast_sqrt_rsqrt(cs + EPS_SQ, &crt, &cinv);\n"
                  "    const double rc = (crt - EPS) - hcL;\n",
                  pv->couple_point, pv->couple_point);
    couple_light = buf;
    couple_light += "    ss = (ss + xq(ss)) + rc * rc;\n    mres_new = fmax(fmax(mres_new, xq(mres_new)), fabs(rc));\n";
    couple_eval = buf;
    std::snprintf(buf, sizeof(buf),
                  "    const double cu = -cd * cinv;  // d rc / d (own joined point), lane component\n"
                  "    gn%d = fma(cu, rc, gn%d);\n", FU, FU);
    couple_eval += buf;
    couple_eval += "    ss = (ss + xq(ss)) + rc * rc;\n    mres_new = fmax(fmax(mres_new, xq(mres_new)), fabs(rc));\n";
  }

  // ---- k > 1 joining rows: the pieces every kernel body shares (text; the single-row bodies keep their own) ----
  // Per half: Dt = D + sum_j w_j w_j^T (own parts of the joining rows' rank-one terms, block-diagonal), Z = Dt^-1 W
  // (k substitutions), G = W^T Z (k x k, symmetric), H = the partner's G.  A solution y of Dt y = b becomes the coupled
  // system's x = y - Z a' with a' = (I - H G)^-1 (t - H s), s = W^T y of this half, t the partner's (derivation:
  // x_o = y_o - Z_o a_p, a_o = s_o - G_o a_p, a_p = s_p - G_p a_o).  For k = 1 this is the 2 x 2 form of the bodies below.
  auto sfmt = [](const char* format, auto... args) {
    char line[512];
    std::snprintf(line, sizeof(line), format, args...);
    return std::string(line);
  };
  auto join_rank_one_src = [&]() {
    std::string out;
    for (int j = 0; j < NK; ++j)
      for (int k = 0; k < 3; ++k)
        out += sfmt("    %s = fma(cu%d, QB%d(cu%d), %s);\n", Gen::A(FUj[j], FUj[j], k).c_str(), j, k, j, Gen::A(FUj[j], FUj[j], k).c_str());
    return out;
  };
  auto join_z_src = [&]() {  // after the factorisation: nz{j}_{F}, smG / smH / smM (leaves ev.out cleared)
    std::string out;
    for (int j = 0; j < NK; ++j) {
      std::vector<std::string> rhs_w;
      for (int F = 0; F < nf; ++F) rhs_w.push_back(F == FUj[j] ? "cu" + std::to_string(j) : "0.0");
      for (int F = 0; F < nf; ++F) out += sfmt("    double nz%d_%d;\n", j, F);
      ev.out.clear();
      ev.emit_substitute(rhs_w, "sz");
      out += "    {\n" + ev.out;
      for (int F = 0; F < nf; ++F) out += sfmt("    nz%d_%d = sz%d;\n", j, F, F);
      out += "    }\n";
    }
    ev.out.clear();
    for (int i = 0; i < NK; ++i)
      for (int j = i; j < NK; ++j) {
        out += sfmt("    const double smG%d_%d = qsum(cu%d * nz%d_%d), smH%d_%d = xq(smG%d_%d);\n", i, j, i, j, FUj[i], i, j, i, j);
        if (j != i) out += sfmt("    const double smG%d_%d = smG%d_%d, smH%d_%d = smH%d_%d;\n", j, i, i, j, j, i, i, j);
      }
    // (I - H G)^-1 by Gauss-Jordan on [N | I], unrolled here (no pivoting: N = I - (two matrices of norm < 1))
    for (int i = 0; i < NK; ++i)
      for (int j = 0; j < 2 * NK; ++j) {
        std::string e = j < NK ? (i == j ? "1.0" : "0.0") : (j - NK == i ? "1.0" : "0.0");
        if (j < NK)
          for (int l = 0; l < NK; ++l) e += sfmt(" - smH%d_%d * smG%d_%d", i, l, l, j);
        out += sfmt("    double smA%d_%d = %s;\n", i, j, e.c_str());
      }
    for (int piv = 0; piv < NK; ++piv) {
      out += sfmt("    { const double smr = 1.0 / smA%d_%d;", piv, piv);
      for (int j = 0; j < 2 * NK; ++j) out += sfmt(" smA%d_%d *= smr;", piv, j);
      out += " }\n";
      for (int i = 0; i < NK; ++i) {
        if (i == piv) continue;
        out += sfmt("    { const double smf = smA%d_%d;", i, piv);
        for (int j = 0; j < 2 * NK; ++j) out += sfmt(" smA%d_%d = fma(-smf, smA%d_%d, smA%d_%d);", i, j, piv, j, i, j);
        out += " }\n";
      }
    }
    return out;
  };
  // corrects the solution whose block F is in_name(F); out_stmt(F, expression) is the statement that receives block F
  auto join_correct_src = [&](const std::function<std::string(int)>& in_name,
                              const std::function<std::string(int, const std::string&)>& out_stmt) {
    std::string out;
    for (int i = 0; i < NK; ++i)
      out += sfmt("    const double smS%d = qsum(cu%d * %s), smT%d = xq(smS%d);\n", i, i, in_name(FUj[i]).c_str(), i, i);
    for (int i = 0; i < NK; ++i) {
      std::string e = "smT" + std::to_string(i);
      for (int l = 0; l < NK; ++l) e += sfmt(" - smH%d_%d * smS%d", i, l, l);
      out += sfmt("    const double smR%d = %s;\n", i, e.c_str());
    }
    for (int i = 0; i < NK; ++i) {
      std::string e;
      for (int l = 0; l < NK; ++l) e += sfmt("%ssmA%d_%d * smR%d", l ? " + " : "", i, NK + l, l);
      out += sfmt("    const double sma%d = %s;\n", i, e.c_str());
    }
    for (int F = 0; F < nf; ++F) {
      std::string e = in_name(F);
      for (int j = 0; j < NK; ++j) e = sfmt("fma(-nz%d_%d, sma%d, %s)", j, F, j, e.c_str());
      out += out_stmt(F, e);
    }
    return out;
  };

  // which points must live in registers
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

  // ---- evaluated pair module: where a program output point lives (which half, which of that half's MOVING points) ----
  // The epilogue stages the velocities of the moving points only - [problem][target][half][moving point][3] - fixed points'
  // are zero; a program record element maps to an offset in one problem-target block (kVelMap, -1: a fixed point).
  std::vector<int> mov_index(NP, -1);  // side point -> index among the half's moving (free or derived) points
  int MVN = 0;
  if (EVP)
    for (int p = 0; p < NP; ++p)
      if (used[p] && (g.blk_of_point[p] >= 0 || g.dop_of_point[p] >= 0)) mov_index[p] = MVN++;
  const int MV = 3 * MVN;
  struct OutLoc { int side, point; };  // of a program output index (side -1: a fixed point neither half moves)
  std::vector<OutLoc> out_loc(prog_out, OutLoc{-1, -1});
  if (EVP)
    for (int sd = 0; sd < 2; ++sd)
      for (int k = 0; k < P.n_out; ++k)
        if (pv->out[sd][k] >= 0 && (sd == 0 || pv->out[1][k] != pv->out[0][k])) out_loc[pv->out[sd][k]] = {sd, P.out_point[k]};
  auto vel_offset = [&](int k) {  // offset of output point k's velocity inside a [half][moving point][3] block, -1: fixed
    const OutLoc& l = out_loc[k];
    if (l.side < 0 || mov_index[l.point] < 0) return -1;
    return l.side * MV + 3 * mov_index[l.point];
  };
  g.out += kPreamble;
  if (EV) {
    g.out += eval_metrics_source(*es);
    if (EVP) {
      g.out += eval_roles_source();
      g.f("struct QEvArgs { QArgs q; double* tan; double* ev; EvCfg cfg; EvCfg cfg_r; EvRoleNum roles[8]; };");
      std::string table = "__constant__ short kVelMap[" + std::to_string(3 * prog_out) + "] = {";
      for (int k = 0; k < prog_out; ++k)
        for (int cc2 = 0; cc2 < 3; ++cc2) {
          const int off = vel_offset(k);
          table += (k || cc2 ? ", " : "") + std::to_string(off < 0 ? -1 : off + cc2);
        }
      g.out += table + "};  // record element -> offset in a problem-target block of the staged velocities\n";
    } else
    g.f("struct QEvArgs { QArgs q; double* tan; double* ev; EvCfg cfg; };");
    g.f("#define EV_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\"); } while (0)");
  }
  if (ev.tl_marks) g.f("#define OKX_TL(k)");
  g.f("");
  if (pv) {
    g.f("// Two quads own one problem: `xq` reads the other quad's lane with the same component");
    g.f("// (ds_swizzle, lane ^ 4: data path only, no LDS memory); PSUM / PMAX reduce over both quads and");
    g.f("// are bit-identical in all eight lanes (commutative combination of the two quad results).");
    g.f("DEV double xq(double v) {");
    g.f("  int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x101F);");
    g.f("  int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x101F);");
    g.f("  return __hiloint2double(hi, lo);");
    g.f("}");
    if (EVP) g.f("DEV Du<1> du_xq(Du<1> a) { Du<1> r; r.v = xq(a.v); r.d[0] = xq(a.d[0]); return r; }");
    g.f("DEV double PSUM(double v) { const double s = qsum(v); return s + xq(s); }");
    g.f("DEV double PMAX(double v) { const double s = qmax(v); return fmax(s, xq(s)); }");
    g.f("#define PJOIN_SUM(v) ((v) + xq(v))");
  } else {
    g.f("#define PSUM(v) qsum(v)");
    g.f("#define PMAX(v) qmax(v)");
    g.f("#define PJOIN_SUM(v) (v)");
  }
  g.f("");
  // The solve body is emitted twice in single mode: the general one (chains, fitted model, trace) and a COLD one for
  // launches of independent solves from the design state with a first-step table (chain_len 1: the headline shape) -
  // no chain history, no model, no trace, the first step applied from registers.  Same passes, same answers.
  const std::string lds_decl =
      "  const int qs = lane >> 2;  // quad(-side) slot of this lane inside the wavefront\n"
      "  __shared__ double hsl[" + std::to_string(16 * (ev.n_scalar_slots + 1)) + "];  // chain-constant scalars [slot][quad]\n"
      "  __shared__ double hql[" + std::to_string(64 * (ev.n_lane_slots + 1)) + "];  // chain-constant lane components [slot][lane]\n";
  std::string final_src;
  bool body_failed = false;
  // ---- evaluated module: the epilogue every body (and okx_quad_evaluate_*) ends a problem with ----
  // At the solved state (every point p{k} in registers, the output record staged in `stage`): J once more, undamped
  // J^T J = L D L^T, one substitution per target for the solution-manifold tangent q_t = (J^T J)^-1 J^T e_t
  // (reference sensitivity.py:57-143; the pinned line rows play the part of its _degenerate_constraint_pins), the
  // velocity of every point by the derived ops' closed-form directional derivatives, staged in LDS [quad][t][record].
  // Then the metric catalog (okx_evalsrc.cpp) ONCE per quad: lane c evaluates direction c - 1 - lane 0 the values (a zero
  // tangent), lanes 1 .. T the derivative along target 0 .. T - 1 - on duals with one tangent component, its role points
  // gathered from the staged record and its own direction's velocities.  Results leave as one contiguous block per
  // wavefront: [problem][1 + T][OKX_EVAL_COLUMNS] (row 0: values + the factorisation's pivots, row 1 + t: d / d target t,
  // then the wheel centre's and the rack pickup's rates - the drivers of the reference's derivative columns).
  // `contig`: C expression, true when the wavefront's problems are consecutive (bb = wu * 16 + quad).
  const int REC = 3 * P.n_out, EVC = OKX_EVAL_COLUMNS;
  std::string epi_factor_src;
  if (EV) {
    if (T < 1) {
      *why = "an evaluated module needs at least one target";
      return false;
    }
    std::vector<int> oi(NP, -1);
    for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
    for (int F = 0; F < nf; ++F)
      if (oi[ev.fp(F)] < 0) {
        *why = "an evaluated module needs every free point among the output points";
        return false;
      }
    ev.out.clear();
    ev.emit_factor();
    epi_factor_src = ev.out;
    ev.out.clear();
  }
  // ---- pair mode (a composed axle): ONE epilogue for the whole axle ----
  // Each half's quad evaluates its half of J at the solved state; the undamped system is solved as in the passes (each
  // half's own factor with its share of the joining rows' rank-one terms, the 2k x 2k Woodbury system for the rest), one
  // substitution per PROGRAM target; the moving points' velocities are staged in LDS [problem][target][half][point][3].
  // Then lane c of EACH quad evaluates direction c - 1 (lane 0 the values) of ITS corner's catalog - role points from the
  // staged record, the left / right role indices selected on the side bit -; the left quad combines both corners'
  // travel, contact patch and front-view instant centre (the partner's through ds_swizzle) into the axle-scope metrics
  // (metrics/axle_metrics.py:21-95); the roles of okx_rotation_role's kinds are evaluated on the same duals.  Every lane
  // stores its own row segment of d_eval [problem][1 + T][OKX_EVAL_AXLE_COLUMNS].
  const int TP = prog_targets, RECP = 3 * prog_out, EVA = kEvalAxleColumns;
  struct EvJob { int t, side, prog_t; };
  std::vector<EvJob> ev_jobs;
  if (EVP)
    for (int t = 0; t < T; ++t)
      for (int sd = 0; sd < 2; ++sd)
        if (pv->tgt[sd][t] >= 0) ev_jobs.push_back({t, sd, pv->tgt[sd][t]});
  auto pair_epilogue_src = [&](const std::string& contig) -> std::string {
    ev.out.clear();
    ev.reset_caches();
    ev.f("    {  // ---- evaluated epilogue (axle): tangents at the solved state, both corners' metrics, axle metrics, roles ----");
    ev.out += eval_src;
    ev.out += couple_eval;
    ev.f("    const double lambda = 0.0;  // (an undamped factorisation; shadows the solve's damping)");
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G <= F; ++G)
        if (ev.fillf[F][G])
          for (int k = 0; k < 3; ++k) {
            if (!(F == G && k == 2)) ev.f("    double %s;", Gen::Ln(F, G, k).c_str());
            if (!ev.nz[F][G]) ev.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
          }
    if (NK > 1) ev.out += join_rank_one_src();
    else
      for (int k = 0; k < 3; ++k)
        ev.f("    %s = fma(cu, QB%d(cu), %s);", Gen::A(FU, FU, k).c_str(), k, Gen::A(FU, FU, k).c_str());
    ev.out += epi_factor_src;
    ev.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
    ev.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
    if (NK > 1) {
      const std::string keep = ev.out;  // (join_z_src works through ev.out)
      const std::string z = join_z_src();
      ev.out = keep + z;
    } else {
      // (the joined point is eliminated last: the joining row's own system is a solve in the last block, and every target's
      //  coupling correction is made in that block as the backward substitution delivers it - see the passes' solve step)
      ev.out += ev.last_block_solve("cu", "smz");
      ev.f("    const double sm_g = qsum(cu * smz), sm_gp = xq(sm_g);");
      ev.f("    const double sm_det = 1.0 - sm_g * sm_gp;");
    }
    ev.f("    if (c == 0 && !q1) vok[quad] = ok ? 0.0 : __builtin_nan(\"\");  // a fixed point's velocity: zero, or NaN with the rest");
    for (const EvJob& job : ev_jobs) {
      const int t = job.t;
      ev.f("    {  // program target %d: (J^T J) q = J^T e_t, then the velocity of every moving point", job.prog_t);
      ev.f("    const double ms = q1 == %d ? 1.0 : 0.0;  // the half that carries this target", job.side);
      std::vector<std::string> rhs(nf, "0.0");
      auto it = ev.target_j.find(t);
      if (it != ev.target_j.end())
        for (auto& fv : it->second) rhs[fv.first] = "(ms * " + Gen::sx(fv.second) + ")";
      if (NK > 1) {
        ev.emit_substitute(rhs, "ty");
        ev.out += join_correct_src([&](int F) { return "ty" + std::to_string(F); },
                                   [&](int F, const std::string& e) { return sfmt("    const double tq%d = %s;\n", F, e.c_str()); });
      } else {
        std::string hook = sfmt("    const double sm_s = qsum(cu * tq%d);\n", FU);
        hook += "    const double sm_c = (xq(sm_s) - sm_gp * sm_s) / sm_det;\n";
        hook += sfmt("    tq%d = fma(-smz, sm_c, tq%d);\n", FU, FU);
        ev.emit_substitute(rhs, "tq", &hook);
      }
      const std::string vp = "w" + std::to_string(job.prog_t) + "_";
      for (int p = 0; p < NP; ++p) {
        if (!used[p] || ev.dop_of_point[p] >= 0) continue;
        if (ev.blk_of_point[p] >= 0) ev.f("    const double %s%d = tq%d;", vp.c_str(), p, ev.blk_of_point[p]);
        else ev.f("    const double %s%d = 0.0;", vp.c_str(), p);
      }
      for (int e = 0; e < P.n_derived; ++e)
        if (!ev.derived_jvp(e, vp)) return std::string();
      ev.f("    if (c < 3) {");
      ev.f("      double* vs = vst + ((quad * %d + %d) * 2 + q1) * %d + c;", TP, job.prog_t, MV);
      for (int p = 0; p < NP; ++p)
        if (mov_index[p] >= 0) ev.f("      vs[%d] = ok ? %s%d : __builtin_nan(\"\");", 3 * mov_index[p], vp.c_str(), p);
      ev.f("    }");
      ev.f("    }");
    }
    ev.f("    EV_WAVE_SYNC();");
    // tangents [B][T][n_out][3], when asked for: the staged velocities expanded to whole records (fixed points: zero)
    ev.f("    if (ea.tan != nullptr) {");
    ev.f("      if (%s) {", contig.c_str());
    ev.f("        const long long ev_rem = a.n_problems - wu * %d;", PPW);
    ev.f("        const int n_el = (int)(ev_rem < %d ? ev_rem : %d) * %d;", PPW, PPW, TP * RECP);
    ev.f("        double* dst = ea.tan + wu * %d * %d;", PPW, TP * RECP);
    ev.f("        for (int i = lane; i < n_el; i += 64) {");
    ev.f("          const int blk = i / %d, el = i - blk * %d, off = kVelMap[el];  // blk = problem * T + target", RECP, RECP);
    ev.f("          dst[i] = off >= 0 ? vst[blk * %d + off] : vok[blk / %d];", 2 * MV, TP);
    ev.f("        }");
    ev.f("      } else if (valid) {  // chains: a problem's tangents by its own eight lanes");
    ev.f("        double* dst = ea.tan + bb * %d;", TP * RECP);
    ev.f("        for (int i = lane & 7; i < %d; i += 8) {", TP * RECP);
    ev.f("          const int blk = i / %d, el = i - blk * %d, off = kVelMap[el];", RECP, RECP);
    ev.f("          dst[i] = off >= 0 ? vst[(quad * %d + blk) * %d + off] : vok[quad];", TP, 2 * MV);
    ev.f("        }");
    ev.f("      }");
    ev.f("    }");
    ev.f("    if (ea.ev != nullptr) {");
    ev.f("      EvCfg cfg = ea.cfg;");
    for (const char* field : {"side_sign", "design_wheel_center_z", "design_contact_patch_z", "design_rack_y", "wheelbase", "cg_z", "front_brake_bias"})
      ev.f("      cfg.%s = q1 ? ea.cfg_r.%s : ea.cfg.%s;", field, field, field);
    {
      const EvalSpec &sl = aes->side[0], &sr = aes->side[1];
      auto gp3 = [&](int kL, int kR, int comp) {  // a design coordinate of this half's role point in the geometry table
        return Gen::sel(3 * program.out_point[kL] + comp, 3 * program.out_point[kR] + comp);
      };
      ev.f("      if (PG) {  // an ensemble's design references are its geometry's own");
      ev.f("        cfg.design_wheel_center_z = gp[%s]; cfg.design_contact_patch_z = gp[%s];", gp3(sl.wheel_center, sr.wheel_center, 2).c_str(),
           gp3(sl.contact_patch, sr.contact_patch, 2).c_str());
      if (sl.rack >= 0) ev.f("        cfg.design_rack_y = gp[%s];", gp3(sl.rack, sr.rack, 1).c_str());
      ev.f("      }");
    }
    ev.f("      const double ev_flags = (ok ? 1.0 : 0.0) + ((!ok || pmin <= %d * 2.220446049250313e-16 * pmax) ? 2.0 : 0.0);", 3 * nf * 2);
    ev.f("      _Pragma(\"unroll 1\")");
    ev.f("      for (int ep = 0; 4 * ep <= %d; ++ep) {", TP);
    ev.f("        const int evt = 4 * ep + c - 1;  // this lane's direction: -1 the values, t the derivative along program target t");
    ev.f("        const bool dir = evt >= 0 && evt < %d;", TP);
    ev.f("        const double* ps = stage + quad * %d;", RECP);
    ev.f("        const double* vq = vst + (quad * %d + (dir ? evt : 0)) * %d;  // this direction's block: [half][moving point][3]", TP, 2 * MV);
    // a point as a dual vector: position from the staged record, velocity from the staged block (zero for the value lane
    // and for fixed points); kL / kR = the output index this lane reads when it sits in the left / right half
    auto load_point = [&](const std::string& dst, int kL, int kR) {
      const int oL = vel_offset(kL), oR = vel_offset(kR);
      for (int cc2 = 0; cc2 < 3; ++cc2) {
        const std::string pos = kL == kR ? sfmt("ps[%d]", 3 * kL + cc2) : sfmt("ps[q1 ? %d : %d]", 3 * kR + cc2, 3 * kL + cc2);
        std::string vel;
        if (oL < 0 && oR < 0) vel = "0.0";
        else if (oL == oR) vel = sfmt("dir ? vq[%d] : 0.0", oL + cc2);
        else if (oL >= 0 && oR >= 0) vel = sfmt("dir ? vq[q1 ? %d : %d] : 0.0", oR + cc2, oL + cc2);
        else vel = sfmt("(dir && (q1 ? %s : %s)) ? vq[%d] : 0.0", oR >= 0 ? "true" : "false", oL >= 0 ? "true" : "false", (oL >= 0 ? oL : oR) + cc2);
        ev.f("        %s.%c.v = %s; %s.%c.d[0] = %s;", dst.c_str(), "xyz"[cc2], pos.c_str(), dst.c_str(), "xyz"[cc2], vel.c_str());
      }
    };
    ev.f("        DV<1> RP[EV_SLOTS];");
    for (int sl = 0; sl < kEvalSlots; ++sl) {
      const int kL = eval_slot_point(aes->side[0], sl), kR = eval_slot_point(aes->side[1], sl);
      if (kL < 0 || kR < 0) {
        ev.f("        RP[%d].x = du_const<1>(0.0); RP[%d].y = du_const<1>(0.0); RP[%d].z = du_const<1>(0.0);", sl, sl, sl);
        continue;
      }
      load_point("RP[" + std::to_string(sl) + "]", kL, kR);
    }
    ev.f("        Du<1> em[%d];", OKX_METRIC_COUNT);
    ev.f("        ev_corner_metrics<1>(cfg, RP, em);");
    // axle-scope metrics: own corner = this lane's half, partner = the other half's lane of the same direction
    ev.f("        Du<1> am[8];");
    ev.f("        {");
    ev.f("          const Du<1> ow = em[%d], oc = RP[EV_SLOT_CONTACT_PATCH].z - du_const<1>(cfg.design_contact_patch_z);", OKX_METRIC_WHEEL_TRAVEL);
    ev.f("          const Du<1> oy = RP[EV_SLOT_CONTACT_PATCH].y, oz = RP[EV_SLOT_CONTACT_PATCH].z;");
    ev.f("          const Du<1> ofy = em[%d] - oy, ofz = em[%d] - oz;  // contact patch -> front-view instant centre (NaN: none)", OKX_METRIC_FVIC_Y, OKX_METRIC_FVIC_Z);
    ev.f("          const Du<1> pw = du_xq(ow), pc = du_xq(oc), py = du_xq(oy), pz = du_xq(oz), pfy = du_xq(ofy), pfz = du_xq(ofz);");
    ev.f("          const Du<1> trk = du_abs(oy - py);");
    ev.f("          am[%d] = 0.5 * (ow + pw);", OKX_AXLE_METRIC_HEAVE);
    ev.f("          am[%d] = 57.29577951308232 * du_atan2(ow - pw, trk);", OKX_AXLE_METRIC_ROLL);
    ev.f("          am[%d] = (-0.5) * (oc + pc);", OKX_AXLE_METRIC_RIDE_HEIGHT_CHANGE);
    ev.f("          am[%d] = trk;", OKX_AXLE_METRIC_TRACK);
    ev.f("          am[%d] = du_nan<1>(); am[%d] = du_nan<1>(); am[7] = du_const<1>(0.0);", OKX_AXLE_METRIC_ROLL_CENTER_Y, OKX_AXLE_METRIC_ROLL_CENTER_Z);
    ev.f("          const Du<1> den = ofy * pfz - ofz * pfy;");
    ev.f("          if (fabs(den.v) >= EV_EPS_GEOMETRIC) {  // (false for NaN: a corner without a front-view instant centre)");
    ev.f("            const Du<1> tt = ((py - oy) * pfz - (pz - oz) * pfy) / den;");
    ev.f("            am[%d] = oy + tt * ofy; am[%d] = oz + tt * ofz;", OKX_AXLE_METRIC_ROLL_CENTER_Y, OKX_AXLE_METRIC_ROLL_CENTER_Z);
    ev.f("          }");
    if (aes->side[0].rack >= 0) ev.f("          am[%d] = RP[EV_SLOT_RACK].y - du_const<1>(cfg.design_rack_y);", OKX_AXLE_METRIC_RACK_DISPLACEMENT);
    else ev.f("          am[%d] = du_nan<1>();", OKX_AXLE_METRIC_RACK_DISPLACEMENT);
    ev.f("        }");
    // roles: two consecutive roles of one kind are evaluated side by side (the left quad the first, the right quad the
    // second); a role without such a partner is evaluated by both quads and stored by the left one
    ev.f("        Du<1> rv[8];");
    ev.f("        for (int k = 0; k < 8; ++k) rv[k] = du_const<1>(0.0);");
    std::vector<std::pair<int, bool>> role_slots;  // (first role, paired)
    for (int k = 0; k < aes->n_roles;) {
      const bool paired = k + 1 < aes->n_roles && aes->role[k].kind == aes->role[k + 1].kind;
      role_slots.push_back({k, paired});
      k += paired ? 2 : 1;
    }
    for (auto& rs : role_slots) {
      const int k = rs.first, k2 = rs.second ? k + 1 : k;
      const EvalRoleSpec &ra = aes->role[k], &rb = aes->role[k2];
      ev.f("        {  // role %d%s (kind %d)", k, rs.second ? " / the next" : "", ra.kind);
      ev.f("          DV<1> ra_, rb_;");
      load_point("ra_", ra.point, rb.point);
      if (ra.kind != OKX_ROLE_AXIS_ROTATION) load_point("rb_", ra.point_b, rb.point_b);
      else ev.f("          rb_ = ra_;");
      if (rs.second) ev.f("          const EvRoleNum rn = q1 ? ea.roles[%d] : ea.roles[%d];", k2, k);
      else ev.f("          const EvRoleNum rn = ea.roles[%d];", k);
      ev.f("          rv[%d] = ev_role<%d, 1>(rn, ra_, rb_);", k, ra.kind);
      ev.f("        }");
    }
    ev.f("        if (valid && evt < %d) {", TP);
    ev.f("          double* eo = ea.ev + (bb * %d + evt + 1) * %d;", 1 + TP, EVA);
    ev.f("          double* co = eo + (q1 ? %d : 0);  // this half's corner block", OKX_EVAL_COLUMNS);
    ev.f("          if (evt < 0) {");
    for (int k = 0; k < OKX_METRIC_COUNT; ++k) ev.f("            co[%d] = em[%d].v;", k, k);
    ev.f("            co[19] = q1 ? 0.0 : pmin; co[20] = q1 ? 0.0 : pmax; co[21] = q1 ? 0.0 : ev_flags; co[22] = 0.0; co[23] = 0.0;");
    ev.f("          } else {");
    for (int k = 0; k < OKX_METRIC_COUNT; ++k) ev.f("            co[%d] = em[%d].d[0];", k, k);
    ev.f("            co[19] = RP[EV_SLOT_WHEEL_CENTER].x.d[0]; co[20] = RP[EV_SLOT_WHEEL_CENTER].y.d[0]; co[21] = RP[EV_SLOT_WHEEL_CENTER].z.d[0];");
    ev.f("            co[22] = %s; co[23] = 0.0;", aes->side[0].rack >= 0 ? "RP[EV_SLOT_RACK].y.d[0]" : "__builtin_nan(\"\")");
    ev.f("          }");
    ev.f("          if (!q1) {");
    ev.f("            for (int k = 0; k < 8; ++k) eo[%d + k] = evt < 0 ? am[k].v : am[k].d[0];", 48);
    for (int k = aes->n_roles; k < 8; ++k) ev.f("            eo[%d] = 0.0;  // (no such role)", 56 + k);
    ev.f("          }");
    ev.f("        }");
    // (the role columns after the zero fill: the left quad's stores are ordered, the right quad writes other columns)
    for (auto& rs : role_slots) {
      const int k = rs.first;
      if (rs.second)
        ev.f("        if (valid && evt < %d) ea.ev[(bb * %d + evt + 1) * %d + %d + q1] = evt < 0 ? rv[%d].v : rv[%d].d[0];", TP, 1 + TP, EVA, 56 + k, k, k);
      else
        ev.f("        if (valid && evt < %d && !q1) ea.ev[(bb * %d + evt + 1) * %d + %d] = evt < 0 ? rv[%d].v : rv[%d].d[0];", TP, 1 + TP, EVA, 56 + k, k, k);
    }
    ev.f("      }");
    ev.f("    }");
    ev.f("    EV_WAVE_SYNC();  // (stage / vst are reused by the next problem of this wavefront)");
    ev.f("    }");
    std::string text = ev.out;
    ev.out.clear();
    ev.reset_caches();
    return text;
  };
  auto epilogue_src = [&](const std::string& contig) -> std::string {
    if (EVP) return pair_epilogue_src(contig);
    ev.out.clear();
    ev.reset_caches();
    ev.f("    {  // ---- evaluated epilogue: tangents at the solved state, metrics and their derivatives along them ----");
    ev.out += eval_src;
    ev.f("    const double lambda = 0.0;  // (an undamped factorisation; shadows the solve's damping)");
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G <= F; ++G)
        if (ev.fillf[F][G])
          for (int k = 0; k < 3; ++k) {
            if (!(F == G && k == 2)) ev.f("    double %s;", Gen::Ln(F, G, k).c_str());
            if (!ev.nz[F][G]) ev.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
          }
    ev.out += epi_factor_src;
    ev.f("    __shared__ __attribute__((aligned(16))) double vst[16 * %d];  // velocities [quad][target][record]", T * REC);
    for (int t = 0; t < T; ++t) {
      ev.f("    {  // target %d: (J^T J) q = J^T e_t, then the velocity of every point", t);
      std::vector<std::string> rhs(nf, "0.0");
      auto it = ev.target_j.find(t);
      if (it != ev.target_j.end())
        for (auto& fv : it->second) rhs[fv.first] = Gen::sx(fv.second);
      ev.emit_substitute(rhs, "tq");
      const std::string vp = "w" + std::to_string(t) + "_";
      for (int p = 0; p < NP; ++p) {
        if (!used[p] || ev.dop_of_point[p] >= 0) continue;
        if (ev.blk_of_point[p] >= 0) ev.f("    const double %s%d = tq%d;", vp.c_str(), p, ev.blk_of_point[p]);
        else ev.f("    const double %s%d = 0.0;", vp.c_str(), p);
      }
      for (int e = 0; e < P.n_derived; ++e)
        if (!ev.derived_jvp(e, vp)) return std::string();
      ev.f("    if (c < 3) {");
      ev.f("      double* vs = vst + (quad * %d + %d) * %d + c;", T, t, REC);
      for (int k = 0; k < P.n_out; ++k) ev.f("      vs[%d] = ok ? %s%d : __builtin_nan(\"\");", 3 * k, vp.c_str(), P.out_point[k]);
      ev.f("    }");
      ev.f("    }");
    }
    ev.f("    EV_WAVE_SYNC();");
    ev.f("    const long long ev_rem = a.n_problems - wu * 16;");
    ev.f("    const int ev_rows = (int)(ev_rem < 16 ? ev_rem : 16);  // problems of a contiguous wavefront block");
    auto copy_out = [&](const char* dst, const char* stage_name, int per_problem) {
      ev.f("      if (%s) {", contig.c_str());
      ev.f("        const int n_doubles = ev_rows * %d;", per_problem);
      ev.f("        double2* dst = reinterpret_cast<double2*>(%s + wu * 16 * %d);", dst, per_problem);
      ev.f("        const double2* src = reinterpret_cast<const double2*>(%s);", stage_name);
      ev.f("        for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
      ev.f("        if ((n_doubles & 1) && lane == 0) (%s + wu * 16 * %d)[n_doubles - 1] = %s[n_doubles - 1];", dst, per_problem, stage_name);
      ev.f("      } else if (valid) {  // chains: a quad's problems are far apart in memory");
      ev.f("        double* dst = %s + bb * %d;", dst, per_problem);
      ev.f("        const double* src = %s + quad * %d;", stage_name, per_problem);
      ev.f("        for (int i = c; i < %d; i += 4) dst[i] = src[i];", per_problem);
      ev.f("      }");
    };
    ev.f("    if (ea.tan != nullptr) {");
    copy_out("ea.tan", "vst", T * REC);
    ev.f("    }");
    ev.f("    if (ea.ev != nullptr) {");
    ev.f("      __shared__ __attribute__((aligned(16))) double est[16 * %d];  // [quad][row][column]", (1 + T) * EVC);
    ev.f("      EvCfg cfg = ea.cfg;");
    ev.f("      if (PG) {  // an ensemble's design references are its geometry's own");
    ev.f("        cfg.design_wheel_center_z = gp[%d]; cfg.design_contact_patch_z = gp[%d];", 3 * P.out_point[es->wheel_center] + 2,
         3 * P.out_point[es->contact_patch] + 2);
    if (es->rack >= 0) ev.f("        cfg.design_rack_y = gp[%d];", 3 * P.out_point[es->rack] + 1);
    ev.f("      }");
    ev.f("      const double ev_flags = (ok ? 1.0 : 0.0) + ((!ok || pmin <= %d * 2.220446049250313e-16 * pmax) ? 2.0 : 0.0);", 3 * nf);
    ev.f("      _Pragma(\"unroll 1\")");
    ev.f("      for (int ep = 0; 4 * ep <= %d; ++ep) {", T);
    ev.f("        const int evt = 4 * ep + c - 1;  // this lane's direction: -1 the values, t the derivative along target t");
    ev.f("        const bool dir = evt >= 0 && evt < %d;", T);
    ev.f("        const double* ps = stage + quad * %d;", REC);
    ev.f("        const double* vs = vst + (quad * %d + (dir ? evt : 0)) * %d;", T, REC);
    ev.f("        DV<1> RP[EV_SLOTS];");
    for (int sl = 0; sl < kEvalSlots; ++sl) {
      const int k = eval_slot_point(*es, sl);
      if (k < 0) {
        ev.f("        RP[%d].x = du_const<1>(0.0); RP[%d].y = du_const<1>(0.0); RP[%d].z = du_const<1>(0.0);", sl, sl, sl);
        continue;
      }
      for (int cc2 = 0; cc2 < 3; ++cc2)
        ev.f("        RP[%d].%c.v = ps[%d]; RP[%d].%c.d[0] = dir ? vs[%d] : 0.0;", sl, "xyz"[cc2], 3 * k + cc2, sl, "xyz"[cc2], 3 * k + cc2);
    }
    ev.f("        Du<1> em[%d];", OKX_METRIC_COUNT);
    ev.f("        ev_corner_metrics<1>(cfg, RP, em);");
    ev.f("        if (evt < %d) {", T);
    ev.f("          double* eo = est + (quad * %d + evt + 1) * %d;", 1 + T, EVC);
    ev.f("          if (evt < 0) {");
    for (int k = 0; k < OKX_METRIC_COUNT; ++k) ev.f("            eo[%d] = em[%d].v;", k, k);
    ev.f("            eo[19] = pmin; eo[20] = pmax; eo[21] = ev_flags; eo[22] = 0.0; eo[23] = 0.0;");
    ev.f("          } else {");
    for (int k = 0; k < OKX_METRIC_COUNT; ++k) ev.f("            eo[%d] = em[%d].d[0];", k, k);
    ev.f("            eo[19] = RP[EV_SLOT_WHEEL_CENTER].x.d[0]; eo[20] = RP[EV_SLOT_WHEEL_CENTER].y.d[0]; eo[21] = RP[EV_SLOT_WHEEL_CENTER].z.d[0];");
    ev.f("            eo[22] = %s; eo[23] = 0.0;", es->rack >= 0 ? "RP[EV_SLOT_RACK].y.d[0]" : "__builtin_nan(\"\")");
    ev.f("          }");
    ev.f("        }");
    ev.f("      }");
    ev.f("      EV_WAVE_SYNC();");
    copy_out("ea.ev", "est", (1 + T) * EVC);
    ev.f("    }");
    ev.f("    EV_WAVE_SYNC();  // (vst / est are reused by the next problem of this wavefront)");
    ev.f("    }");
    std::string text = ev.out;
    ev.out.clear();
    ev.reset_caches();
    return text;
  };
  auto emit_body = [&](const bool CD) {
  // One damped step from the normal equations in hand: declarations of the factor's registers, then (single mode) LDL^T +
  // substitutions, or (pair mode) each half's factorisation and the Woodbury system of the joining rows.  Leaves nx{F}, ok,
  // pmin, pmax (pair mode also pcoup, kc).  Emitted in the general loop and in the cold body's fast loop: the same text.
  auto emit_solve_step = [&]() {
  // declare factor / fill-in registers
  for (int F = 0; F < nf; ++F)
    for (int G = 0; G <= F; ++G)
      if (ev.fillf[F][G]) {
        for (int k = 0; k < 3; ++k) {
          if (!(F == G && k == 2)) g.f("    double %s;", Gen::Ln(F, G, k).c_str());
          if (!ev.nz[F][G]) g.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
        }
      }
  // the diagonal-block factor entry for k == 2 is never needed (no row below inside the block)
  if (!pv) {
    g.out += solve_src;
  } else {
    // (D + w w^T) dx = -g with D = blockdiag of the two halves' damped J^T J and w = (w_L, w_R) the joining row's
    // Jacobian.  D alone is nearly singular once the damping has decayed (the partner's rack pickup slides along
    // its line), so each half takes its own part of the rank-one term, Dt = D + blockdiag(w_L w_L^T, w_R w_R^T),
    // and the off-diagonal coupling u v^T + v u^T (u = (w_L, 0), v = (0, w_R)) goes through a 2 x 2 Woodbury
    // system: dx = y - z c, Dt y = -g, Dt z = w (per half), c = (s_partner - g_partner s_own) / (1 - g_own g_partner)
    // with g = w.z and s = w.y of each half.  (Plain Sherman-Morrison on D cancels catastrophically there.)
    if (NK > 1) {
      g.out += join_rank_one_src();
      ev.out.clear();
      ev.emit_factor();
      g.out += ev.out;
      g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
      g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
      std::vector<std::string> rhs_g;
      for (int F = 0; F < nf; ++F) rhs_g.push_back("-gn" + std::to_string(F));
      for (int F = 0; F < nf; ++F) g.f("    double ny%d;", F);
      ev.out.clear();
      ev.emit_substitute(rhs_g, "sy");
      g.f("    {");
      g.out += ev.out;
      for (int F = 0; F < nf; ++F) g.f("    ny%d = sy%d;", F, F);
      g.f("    }");
      g.out += join_z_src();
      for (int F = 0; F < nf; ++F) g.f("    double nx%d;", F);
      g.f("    {");
      g.out += join_correct_src([&](int F) { return "ny" + std::to_string(F); },
                                [&](int F, const std::string& e) { return sfmt("    nx%d = %s;\n", F, e.c_str()); });
      g.f("    }");
      // tied modes: one per joining row, each judged as in the single-row case (the halves' compliances along w_j in parallel)
      std::string kc = "1e300";
      for (int j = 0; j < NK; ++j)
        kc = sfmt("fmin(%s, (1.0 - smG%d_%d) * fast_rcp(smG%d_%d) + (1.0 - smH%d_%d) * fast_rcp(smH%d_%d))", kc.c_str(), j, j, j, j, j, j, j, j);
      g.f("    const double kc = %s;", kc.c_str());
      g.f("    const double pcoup = fmax(kc - 2.0 * lambda, 0.0);");
      g.f("    pmin = fmin(pmin, fmax(kc, 0.0));");
    } else {
    for (int k = 0; k < 3; ++k)
      g.f("    %s = fma(cu, QB%d(cu), %s);", Gen::A(FU, FU, k).c_str(), k, Gen::A(FU, FU, k).c_str());
    ev.out.clear();
    ev.mark(5);
    ev.emit_factor();
    ev.mark(6);
    g.out += ev.out;
    g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
    g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
    // ONE substitution: D~ y = -g as it stands; the joining row's own system D~ z = w has its right-hand side in the last
    // block alone (the joined point is eliminated last), so z's last block is a 3 x 3 solve, the correction x = y - z c is
    // made in that block the moment the backward pass has produced it, and the earlier blocks substitute from the corrected
    // block (L^-T is linear: x = L^-T (yh - c zh), zh zero but in its last block).
    std::vector<std::string> rhs_g;
    for (int F = 0; F < nf; ++F) rhs_g.push_back("-gn" + std::to_string(F));
    // (what does not depend on y - z's last block, the halves' g = w . z, the 2 x 2 system's determinant - ahead of the
    //  substitution, where its exchange with the partner half overlaps the forward pass; the hook keeps one exchange)
    g.out += ev.last_block_solve("cu", "smz");
    g.f("    const double sm_g = qsum(cu * smz), sm_gp = xq(sm_g);");
    g.f("    const double sm_idet = fast_rcp(1.0 - sm_g * sm_gp);");
    std::string hook = sfmt("    const double sm_s = qsum(cu * nx%d);\n", FU);
    hook += "    const double sm_k = (xq(sm_s) - sm_gp * sm_s) * sm_idet;\n";
    hook += sfmt("    nx%d = fma(-smz, sm_k, nx%d);\n", FU, FU);
    ev.out.clear();
    ev.emit_substitute(rhs_g, "nx", &hook);
    ev.mark(7);
    g.out += ev.out;
    ev.out.clear();
    // Conditioning of the COUPLED system.  Each half was regularised with its own part of the joining row (Dt = D + w w^T,
    // |w| = 1), so the halves' pivots say nothing about the one mode the joining row ties together: both joined points
    // moving along w.  With s = w^T D^-1 w of a half (the compliance of that half along w), g = w^T Dt^-1 w = s / (1 + s),
    // so 1 / s = (1 - g) / g, and the stiffness of the tied mode is the two halves' in parallel: (1 - g_L) / g_L +
    // (1 - g_R) / g_R.  It joins the pivots in the ill-conditioned test (and in the predicted-convergence bound).
    // With damping each half's share is at least lambda (s <= 1 / lambda), so what the damping did not put there is
    // kc - 2 lambda; `pcoup` carries it to the conditioning test, kc itself bounds the predicted convergence with the pivots.
    g.f("    const double kc = (1.0 - sm_g) * fast_rcp(sm_g) + (1.0 - sm_gp) * fast_rcp(sm_gp);");
    g.f("    const double pcoup = fmax(kc - 2.0 * lambda, 0.0);");
    g.f("    pmin = fmin(pmin, fmax(kc, 0.0));");
    }
  }
  };
  n_state_slots = 0;  // (pair mode: each body numbers its LDS homes from zero)
  // Pair mode keeps the quad-uniform Levenberg-Marquardt scalars in LDS in the GENERAL body (chains: in registers the compiler
  // spills them and the pass re-reads ~60 of them from scratch).  The cold body has no chain history to carry and keeps them
  // in registers since round 6: its "LM decision" and "step norms" sections were chains of dependent LDS round trips, 3.3 k of
  // a 22 k-cycle pass (C3 cold 0.3057 -> 0.2962 ms, A/B on one box; 0 B scratch, 34 KB LDS).  Only where it measured a gain:
  // one joining row, halves of up to ten free points, the plain module (the T-bar and heave-link axles lost 2 - 6 % with it,
  // the evaluated cold body 2 %: their register files are full).  pair_cold_lds: the old layout everywhere.
  pair_state_lds = pv != nullptr && !(CD && NK == 1 && nf <= 10 && !EV && !dev_switch("pair_cold_lds"));
  const bool tl_body = CD && ev.tl_marks;
  if (tl_body)  // sections of the SECOND full pass of a wavefront go to a second table behind the first: a.trace[16 (waves + w) + k]
    g.f("#undef OKX_TL\n#define OKX_TL(k) if (a.trace && tl_pass == 4 && (threadIdx.x & 63) == 0) a.trace[(gridDim.x + blockIdx.x) * 16 + (k)] = (double)__builtin_readcyclecounter();");
  if (CD) {
    g.f("DEV void okx_quad_cold_body(const QArgs& a%s) {", EV ? ", const QEvArgs& ea" : "");
    g.f("  constexpr bool PG = false;  // the program's own geometry only (per-geometry tables differ from quad to quad)");
  } else {
    g.f("template <bool PG> DEV void okx_quad_body(const QArgs& a%s) {", EV ? ", const QEvArgs& ea" : "");
  }
  // (developer build, OKX_QUAD_TIMELINE=1: wavefront w stamps the shader clock into a.trace[16 w + k] - 0 entry, 1 loads
  //  consumed / first step in hand, 2 ... 11 top of each LM pass, 13 passes done, 14 records stored; tools/quad_timeline.py)
  const bool timeline = CD && dev_switch("quad_timeline");
  auto stamp = [&](const char* slot) {
    if (timeline) g.f("    if (a.trace && (threadIdx.x & 63) == 0) a.trace[blockIdx.x * 16 + (%s)] = (double)__builtin_readcyclecounter();", slot);
  };
  if (CD) {
    // every kernel argument the body reads, asked for at once: left alone the compiler fetches them in three dependent
    // scalar loads (each a round trip a lone wavefront waits for)
    g.f("  asm volatile(\"\" :: \"s\"(a.targets), \"s\"(a.out_pos), \"s\"(a.info), \"s\"(a.n_problems), \"s\"(a.steps_per_geometry), \"s\"(a.head),"
        " \"s\"(a.design_pos), \"s\"(a.row_param), \"s\"(a.dop_param), \"s\"(a.out_mode), \"s\"(a.max_iter), \"s\"(a.confirm));");
  }
  stamp("0");
  // (the device's 100 MHz real-time counter, the same on every XCD, at entry and at the end: slots 0 / 1 of the second half)
  if (timeline) g.f("    if (a.trace && (threadIdx.x & 63) == 0) a.trace[(gridDim.x + blockIdx.x) * 16 + 0] = (double)__builtin_amdgcn_s_memrealtime();");
  if (pv)
    g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 3, q1_lane = (lane >> 2) & 1, cc = c < 3 ? c : 2;");
  else
    g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
  g.out += atan_decl;
  g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
  if (!pv && !CD) g.f("  __shared__ double pls[%d];  // LDS copy of the chain-head predictor's table", kPredictorLdsDoubles);
  // Cold body: every table the prologue reads - first-step table, design positions, row and derived-op parameters - is
  // staged into LDS with a handful of coalesced loads per lane and read from there.  Read directly, each of the ~85
  // values is a 64-lane load of 8 ... 32 distinct bytes: the four wavefronts of a CU queue ~330 of them on its one L1,
  // ~4000 cycles before the first pass starts (timeline stamps, tools/quad_timeline.py).
  const int cs_head = 0, cs_pos = (head_stride + 1) / 2 * 2, cs_rp = cs_pos + (3 * prog_points + 1) / 2 * 2,
            cs_dp = cs_rp + 8 * (prog_crows + prog_targets), cs_end = cs_dp + (program.n_derived + 1) / 2 * 2 + 2;
  if (CD) {
    g.f("  __shared__ __attribute__((aligned(16))) double cst[%d];  // [first-step table | design positions | row parameters | derived-op parameters]", cs_end);
    for (int t = 0; t < T; ++t) g.f("  double tpre%d;", t);
    g.f("  {");
    // the targets of this wavefront's first unit travel with the tables (after the staging they would be a round trip of their own)
    g.f("    long long tu0 = (long long)blockIdx.x * %d + quad; if (tu0 >= a.n_problems) tu0 = a.n_problems - 1;", PPW);
    if (pv) g.f("    const int q1 = q1_lane; (void)q1;");
    for (int t = 0; t < T; ++t) g.f("    tpre%d = a.targets[tu0 * %d + %s];", t, prog_targets, ev.target_slot(t).c_str());
    g.f("    const double2* h2 = reinterpret_cast<const double2*>(a.head);  // (hipMalloc'ed: 256-byte aligned)");
    g.f("    double2* c2 = reinterpret_cast<double2*>(cst);");
    // every load first (clamped indices: no branch), then the stores: one round trip
    struct Piece { const char* src; int n, off; };
    const Piece pieces[] = {{"a.design_pos", 3 * prog_points, cs_pos}, {"a.row_param", 8 * (prog_crows + prog_targets), cs_rp}, {"a.dop_param", program.n_derived, cs_dp}};  // (the PROGRAM's tables: both halves')
    const int n2 = (head_stride + 1) / 2;  // (the table's allocation is rounded up to an even count of doubles)
    for (int k = 0; 64 * k < n2; ++k) g.f("    double2 sh%d = h2[min(lane + %d, %d)];", k, 64 * k, n2 - 1);
    for (int q = 0; q < 3; ++q)
      for (int k = 0; 64 * k < pieces[q].n; ++k) g.f("    double sp%d_%d = %s[min(lane + %d, %d)];", q, k, pieces[q].src, 64 * k, pieces[q].n - 1);
    {  // (an opaque use of everything loaded: the compiler would sink each load into the branch that stores it)
      std::string pin = "    asm volatile(\"\" : ";
      bool first = true;
      for (int t = 0; t < T; ++t) {
        pin += std::string(first ? "" : ", ") + "\"+v\"(tpre" + std::to_string(t) + ")";
        first = false;
      }
      for (int k = 0; 64 * k < n2; ++k) {
        pin += std::string(first ? "" : ", ") + "\"+v\"(sh" + std::to_string(k) + ".x), \"+v\"(sh" + std::to_string(k) + ".y)";
        first = false;
      }
      g.out += pin + ");\n";  // (an asm statement takes 30 operands: the three tables' values in one of their own)
      pin = "    asm volatile(\"\" : ";
      first = true;
      for (int q = 0; q < 3; ++q)
        for (int k = 0; 64 * k < pieces[q].n; ++k) {
          pin += std::string(first ? "" : ", ") + "\"+v\"(sp" + std::to_string(q) + "_" + std::to_string(k) + ")";
          first = false;
        }
      g.out += pin + ");\n";
    }
    for (int k = 0; 64 * k < n2; ++k) g.f("    if (lane + %d < %d) c2[lane + %d] = sh%d;", 64 * k, n2, 64 * k, k);
    for (int q = 0; q < 3; ++q)
      for (int k = 0; 64 * k < pieces[q].n; ++k) g.f("    if (lane + %d < %d) cst[%d + lane] = sp%d_%d;", 64 * k, pieces[q].n, pieces[q].off + 64 * k, q, k);
    // (one wavefront per workgroup: its LDS instructions execute in order, only the compiler must keep them in order)
    g.f("    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
    g.f("  }");
    g.f("  const double* const c_rp = cst + %d; const double* const c_dp = cst + %d; (void)c_dp; (void)c_rp;", cs_rp, cs_dp);
  }
  (void)cs_head;
  if (ev.lds_constants) {
    g.out += lds_decl;
    if (EVP && CD) {
      // evaluated axle, cold body: the epilogue's staged velocities take the place of what the passes no longer need once a
      // unit is solved and stored - x / dx, the per-quad scalars, the free-coordinate stage - so that the kernel stays within
      // the 40 KB that let four wavefronts share a CU (every unit re-initialises all of it)
      auto even = [](int n) { return (n + 1) / 2 * 2; };
      const int stage_n = even(PPW * RECP), xsl_n = even(64 * 2 * nf), lms_n = pair_state_lds ? even(16 * (5 * T + 12)) : 0, fst_n = even(PPW * 3 * program.n_free);
      const int vst_n = even(PPW * TP * 2 * MV);
      const int union_n = xsl_n + lms_n + fst_n > vst_n ? xsl_n + lms_n + fst_n : vst_n;
      g.f("  __shared__ __attribute__((aligned(16))) double arena[%d];  // [record stage | x, dx, per-quad scalars, free-coordinate stage  /  staged velocities]", stage_n + union_n);
      g.f("  double* const stage = arena; double* const xsl = arena + %d; double* const lms = arena + %d; double* const fstage = arena + %d;",
          stage_n, stage_n + xsl_n, stage_n + xsl_n + lms_n);
      g.f("  double* const vst = arena + %d;  // velocities [problem][target][half][moving point][3] (%d doubles)", stage_n, vst_n);
      g.f("  __shared__ double vok[%d];", PPW);
    } else {
    if (EVP) {
      g.f("  __shared__ __attribute__((aligned(16))) double stage[%d];", PPW * RECP);
      g.f("  __shared__ __attribute__((aligned(16))) double vst[%d];  // velocities [problem][target][half][moving point][3]", PPW * TP * 2 * MV);
      g.f("  __shared__ double vok[%d];", PPW);
    }
    g.f("  __shared__ double xsl[%d];  // accepted point, chain history and the step in hand [block][lane]", 64 * (CD ? 2 : 4) * nf);
    }
    int n_fixed = 0;
    for (int p = 0; p < NP; ++p) n_fixed += ev.blk_of_point[p] < 0 && ev.dop_of_point[p] < 0;
    if (!fixed_in_regs) g.f("  __shared__ double psl[%d];  // fixed points [point][lane]", 64 * (n_fixed > 0 ? n_fixed : 1));
  } else {
    if (pair_state_lds) g.f("  const int qs = lane >> 2;  // quad(-side) slot of this lane inside the wavefront");
    if (!CD) g.f("  __shared__ double xql[%d];  // third chain-history point [block][lane] (registers are full)", 64 * nf);
  }
  g.f("  const long long spg = a.steps_per_geometry;");
  g.f("  const long long span = spg > 0 ? spg : a.n_problems;");
  g.f("  const long long unit_len = %s;", CD ? "1" : "a.chain_len");
  if (CD) {
    g.f("  const long long n_units = a.n_problems; (void)span;");
  } else {
  g.f("  const long long chains_per_span = unit_len == 1 ? span : (span + unit_len - 1) / unit_len;");
  g.f("  const long long n_units = spg > 0 ? (a.n_problems / span) * chains_per_span : chains_per_span;");
  }
  if (CD && pv) {
    // (pair mode; the single-mode cold body - the headline's one unit per wavefront - keeps the plain loop: with this one it
    //  measured 16.3 -> 16.7 us per C2 sweep, 22 more scalar-register spills in its code)
    // round k of the grid-stride loop gives wavefront w the unit k G + (w + 131 k) mod G: a wavefront's units are spread
    // over the sweep instead of sitting G apart (on a 256-wide grid with G = 1024 that is the same columns - the same
    // distance from the design state, the same number of passes - in every round: C3 cold 0.3375 -> 0.3307 ms; chains,
    // whose neighbours in memory are neighbours in time, lose 1.6 % with it and keep the plain stride)
    g.f("  const unsigned wu_g = gridDim.x, wu_rot_step = 131u %% wu_g;");
    g.f("  unsigned wu_rot = blockIdx.x;");
    g.f("  for (long long wu0 = 0; wu0 * %d < n_units; wu0 += wu_g, wu_rot = wu_rot + wu_rot_step >= wu_g ? wu_rot + wu_rot_step - wu_g : wu_rot + wu_rot_step) {", PPW);
    g.f("    const long long wu = wu0 + wu_rot;");
    g.f("    if (wu * %d >= n_units) continue;", PPW);
  } else
  g.f("  for (long long wu = blockIdx.x; wu * %d < n_units; wu += gridDim.x) {", PPW);
  if (pv) {
    // the side bit as a value the optimiser cannot see through: every per-side table index is then computed where it
    // is used (one v_cndmask) instead of being hoisted to the kernel's top, kept live and spilled
    g.f("    int q1 = q1_lane; asm volatile(\"\" : \"+v\"(q1));");
  }
  g.f("    long long unit = wu * %d + quad;", PPW);
  g.f("    const bool have = unit < n_units;");
  g.f("    if (!have) unit = n_units - 1;");
  // (64-bit divisions are ~100 instructions each: only ensembles with chains need them)
  if (CD) {  // a unit is a problem; only the per-geometry variant needs to know whose (one 64-bit division)
    g.f("    const long long span_idx = PG ? unit / spg : 0;");
  } else {
  g.f("    long long span_idx = 0, chain_in_span = unit;");
  g.f("    if (spg > 0) {");
  g.f("      if (unit_len == 1) { span_idx = unit / spg; chain_in_span = unit - span_idx * spg; }");
  g.f("      else { span_idx = unit / chains_per_span; chain_in_span = unit - span_idx * chains_per_span; }");
  g.f("    }");
  }
  if (pair_state_lds && CD) {
    g.f("    const long long first_b = unit, geom = 0; (void)geom; (void)span_idx;");
    g.f("    const double* gp = cst + %d;", cs_pos);
    g.f("    const double* gq = c_rp;");
  } else if (pair_state_lds) {
    // the chain's index bookkeeping and the table pointers (quad-uniform too) in LDS as well: [slot][quad side]
    g.f("    __shared__ long long lmi[%d];", 16 * 5);
    g.f("    long long& first_b = lmi[0 + qs]; first_b = span_idx * span + chain_in_span * unit_len;");
    g.f("    long long& last_b = lmi[16 + qs]; last_b = first_b + unit_len < (span_idx + 1) * span ? first_b + unit_len : (span_idx + 1) * span;");
    g.f("    long long& geom = lmi[32 + qs]; geom = span_idx;");
    g.f("#define gp (PG ? a.geom_pos + geom * %d : a.design_pos)", 3 * prog_points);
    g.f("#define gq (PG ? a.geom_row_param + geom * %d : a.row_param)", 8 * prog_crows);
  } else {
  if (CD) {
    g.f("    const long long first_b = unit;");
  } else {
  g.f("    const long long first_b = span_idx * span + chain_in_span * unit_len;");
  g.f("    const long long last_b = first_b + unit_len < (span_idx + 1) * span ? first_b + unit_len : (span_idx + 1) * span;");
  }
  g.f("    const long long geom = span_idx;");
  if (CD) {
    g.f("    const double* gp = cst + %d;", cs_pos);
    g.f("    const double* gq = c_rp;");
  } else {
  g.f("    const double* gp = PG ? a.geom_pos + geom * %d : a.design_pos;", 3 * prog_points);
  g.f("    const double* gq = PG ? a.geom_row_param + geom * %d : a.row_param;", 8 * prog_crows);
  }
  }
  // Every load of the prologue is issued before the first dependent instruction: first-step targets, chain
  // constants, points and (single mode) the predictor table's copy into LDS share one round trip.
  if (pair_state_lds && !(EVP && CD)) g.f("    __shared__ double lms[%d];  // per-quad scalars [slot][quad side]", 16 * (5 * T + 12));
  for (int t = 0; t < T; ++t) {
    char init[160];
    if (CD)  // the first unit's targets came with the staged tables; later units of the grid-stride loop load theirs here
      std::snprintf(init, sizeof(init), "wu == (long long)blockIdx.x ? tpre%d : a.targets[first_b * %d + %s]", t, prog_targets, ev.target_slot(t).c_str());
    else
    std::snprintf(init, sizeof(init), "a.targets[first_b * %d + %s]", prog_targets, ev.target_slot(t).c_str());
    g.f("    %s %s %s %s", state_ref("tn" + std::to_string(t), init).c_str(), state_ref("tp" + std::to_string(t), "0.0").c_str(),
        state_ref("tq" + std::to_string(t), "0.0").c_str(), state_ref("tr" + std::to_string(t), "0.0").c_str());
  }
  g.f("    // chain-constant lane-component parameters (line points / directions, target directions)");
  if (CD) {  // the same loads, from the staged copies
    std::string h = ev.hoisted;
    for (const auto& sub : {std::make_pair(std::string("a.row_param"), std::string("c_rp")), std::make_pair(std::string("a.dop_param"), std::string("c_dp"))})
      for (size_t at = h.find(sub.first); at != std::string::npos; at = h.find(sub.first, at + sub.second.size())) h.replace(at, sub.first.size(), sub.second);
    g.out += h;
  } else {
  g.out += ev.hoisted;
  }
  g.out += couple_hoist;
  // point registers; in the register-bound pair kernel the fixed points (read once or twice per pass, never
  // written) live in LDS instead: the compiler would otherwise park them in scratch
  {
    int slot = 0;
    for (int p = 0; p < NP; ++p) {
      if (!used[p]) continue;
      const bool fixed = ev.blk_of_point[p] < 0 && ev.dop_of_point[p] < 0;
      if (ev.lds_constants && fixed && !fixed_in_regs)
        g.f("    double& p%d = psl[%d + lane]; p%d = ld3(gp + %s + cc, c);", p, 64 * slot++, p, ev.point3(p).c_str());
      else
        g.f("    double p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    }
  }
  for (int F = 0; F < nf; ++F) {
    if (ev.lds_constants && CD) {  // cold pair body: the accepted point and the step in hand, no chain history
      g.f("    double& x%d = xsl[%d + lane]; double& dx%d = xsl[%d + lane]; x%d = p%d; dx%d = 0.0;", F, 64 * F, F, 64 * (nf + F), F, ev.fp(F), F);
    } else if (ev.lds_constants) {  // cold per-pass state lives in LDS (register-bound kernel): plain references, same code below
      g.f("    double& x%d = xsl[%d + lane]; double& xp%d = xsl[%d + lane]; double& xq%d = xsl[%d + lane];", F, 64 * (2 * F), F,
          64 * (2 * F + 1), F, 64 * (2 * nf + F));
      g.f("    double& dx%d = xsl[%d + lane];", F, 64 * (3 * nf + F));
      g.f("    x%d = p%d; xp%d = x%d; xq%d = x%d; dx%d = 0.0;", F, ev.fp(F), F, F, F, F, F);
    } else {
      if (CD) g.f("    double x%d = p%d, dx%d = 0.0;", F, ev.fp(F), F);
      else
        g.f("    double x%d = p%d, xp%d = x%d, dx%d = 0.0; double& xq%d = xql[%d + lane]; xq%d = x%d;", F, ev.fp(F), F, F, F, F, 64 * F, F, F);
    }
  }
  if (CD && head_ok) {
    // the first-step table's entries travel with the loads above: ONE batch, nothing computed in between (a scheduling
    // barrier keeps the arithmetic below from being interleaved, which would issue the rest of the loads a round trip later)
    // (pair mode: each half reads its own Q and S blocks of the staged table where they are used - LDS reads, nothing to batch)
    g.f("    const double* hp = cst;");
    g.f("    const double* hqb = hp%s;", pv ? (" + (q1 ? " + std::to_string(head_side) + " : 0)").c_str() : "");
    g.f("    const double* hsb = hp%s; (void)hsb;", pv ? (" + (q1 ? " + std::to_string(head_s_side) + " : 0)").c_str() : "");
    if (!pv)
    for (int k = 0; k < HK; ++k)
      for (int F = 0; F < nf; ++F) g.f("    const double hq%d_%d = hqb[%d + c];", k, F, 4 * (k * nf + F));
    for (int j = 0; j < HK; ++j)
      for (int k = j; k < HK; ++k) g.f("    const double hm%d_%d = hp[%d];", j, k, head_off - 2 * HK * HK + j * HK + k);
    for (int i = 0; i < 7; ++i) g.f("    const double hs%d = hp[%d];", i, head_off + i);
    if (!pv)
    for (int pi = 0; pi < NPAIR; ++pi)
      for (int F = 0; F < nf; ++F) g.f("    const double hS%d_%d = hsb[%d + c];", pi, F, head_s_off + 4 * (pi * nf + F));
    g.f("    __builtin_amdgcn_sched_barrier(0);");
  }
  // The design state is a solved state too (of its own design targets): it seeds the chain's history, so the
  // second step of a chain already extrapolates (secant through design and head) and the third quadratically.
  {
    ev.reset_caches();
    ev.out.clear();
    for (int i = P.n_crows; i < P.m; ++i) {
      const int t = ev.target_of_row(i);
      const std::string d = ev.dot(Gen::pn(P.row_pts[i][0]), ev.rpv(i, 0));
      if (pair_state_lds)
        ev.f("    %s  // target %d at the design state (zero on a half that does not carry it)", state_ref("td" + std::to_string(t), d).c_str(), t);
      else
      ev.f("    const double td%d = %s;  // target %d at the design state (zero on a half that does not carry it)", t, d.c_str(), t);
    }
    g.out += ev.out;
    ev.out.clear();
    ev.reset_caches();
  }
  for (int t = 0; t < T; ++t) g.f("    tp%d = td%d;", t, t);
  if (head_ok) {
    // Shared first step of the unit's FIRST problem, taken here, in the unit prologue, so that the
    // table loads travel with the prologue's other loads (inside the chain loop the compiler serialises them - two
    // loads, wait, fma, next load: a dozen dependent L2 round trips, ~3 us per unit) and nothing of the table stays
    // live across the chain loop: the step and its seven scalars go to LDS and are picked up by the first chain step.
    if (CD) {
      // cold body: the launch guarantees the table (okx_solve_batch), the step and its scalars stay in registers
      for (int F = 0; F < nf; ++F) g.f("    double hcx%d;", F);
      g.f("    double hc_step, hc_M, hc_N, hc_ss, hc_mr, hc_dmax, hc_pmin, hc_pmax, hc_ok;");
      g.f("    {");
    } else {
    g.f("    __shared__ double hxl[%d];  // first step of the unit's head problem [block][lane]", 64 * nf);
    g.f("    __shared__ double hsc[%d];  // its scalars [slot][quad]: step length, dx.g, |dx|^2, cost x 2, max |r|, dmax, min / max pivot, ok", 16 * 9);
    g.f("    bool head_ready = false;");
    g.f("    if (a.head != nullptr && a.grad_tol == 0.0 && (PG || a.predictor == nullptr)) {");
    }
    if (!CD) {
    g.f("      const double* hp = a.head + (PG ? geom * %d : 0);", head_stride);
    // pair mode: each half reads its own Q and S blocks; the Gram matrices and scalars belong to the whole problem
    g.f("      const double* hqb = hp%s;", pv ? (" + (q1 ? " + std::to_string(head_side) + " : 0)").c_str() : "");
    g.f("      const double* hsb = hp%s;", pv ? (" + (q1 ? " + std::to_string(head_s_side) + " : 0)").c_str() : "");
    }
    // single mode: every table load issued at once (they travel with the prologue's other loads).  Pair mode: 2 x 100
    // values at once do not fit beside the chain's invariants (152 B of scratch); there the entries are read where they
    // are used, two blocks between scheduling barriers.
    if (!pv && !CD)
      for (int k = 0; k < HK; ++k)
        for (int F = 0; F < nf; ++F) g.f("      const double hq%d_%d = hqb[%d + c];", k, F, 4 * (k * nf + F));
    if (!CD) {
    for (int j = 0; j < HK; ++j)
      for (int k = j; k < HK; ++k) g.f("      const double hm%d_%d = hp[%d];", j, k, head_off - 2 * HK * HK + j * HK + k);
    for (int i = 0; i < 6; ++i) g.f("      const double hs%d = hp[%d];", i, head_off + i);
    }
    g.f("      const double hr0 = 1.0;  // weight of the constraint rows' own gradient");
    for (int k = 1; k < HK; ++k) {
      const HeadCol& col = head_cols[k];
      if (!pv)
        g.f("      const double hr%d = td%d - tn%d;  // target residual of the first problem at the design state", k, col.t, col.t);
      else  // the residual of the half that carries this program target, known to both halves
        g.f("      const double hr%d = q1 == %d ? td%d - tn%d : xq(td%d - tn%d);", k, col.side, col.t, col.t, col.t, col.t);
    }
    g.f("      double hstep = 0.0, hN = 0.0, hM = 0.0, hss = hs2, hmr = hs3;");
    if (NPAIR > 0) {
      // first-order step d1 and the second-order correction d2 = -1/2 sum_st w_s w_t S_st (see okx_quad_head_*); d2 is
      // taken while it is a correction, 2 |d2| <= 0.75 |d1| (Transtrum & Sethna's acceptance rule)
      if (!CD) g.f("      const double hs6 = hp[%d];", head_off + 6);
      g.f("      double hst1 = 0.0, hst2 = 0.0;");
      if (pv)
        for (int pi = 0; pi < NPAIR; ++pi)
          g.f("      const double hv%d = %s * hr%d * hr%d;", pi, head_pairs[pi].first == head_pairs[pi].second ? "0.5" : "1.0",
              head_pairs[pi].first, head_pairs[pi].second);
      for (int F = 0; F < nf; ++F) {
        std::string e, e2;
        for (int k = 0; k < HK; ++k)
          e += (k ? " + hr" : "hr") + std::to_string(k) + " * " +
               (pv ? "hqb[" + std::to_string(4 * (k * nf + F)) + " + c]" : "hq" + std::to_string(k) + "_" + std::to_string(F));
        for (int pi = 0; pi < NPAIR; ++pi) {
          const int s2 = head_pairs[pi].first, t2 = head_pairs[pi].second;
          const std::string w = pv ? "hv" + std::to_string(pi)
                                   : std::string(s2 == t2 ? "0.5" : "1.0") + " * hr" + std::to_string(s2) + " * hr" + std::to_string(t2);
          e2 += (pi ? " + " : "") + w + (CD && !pv ? " * hS" + std::to_string(pi) + "_" + std::to_string(F)
                                            : " * hsb[" + std::to_string(head_s_off + 4 * (pi * nf + F)) + " + c]");
        }
        g.f("      const double hxa%d = -(%s), hxb%d = -(%s);", F, e.c_str(), F, e2.c_str());
        g.f("      hst1 = fmax(hst1, fabs(hxa%d)); hst2 = fmax(hst2, fabs(hxb%d));", F, F);
        if (pv && !CD) {  // the first-order part waits in LDS, where the finished step goes anyway
          g.f("      hxl[%d + lane] = hxa%d;", 64 * F, F);
          if (F % 2 == 1) g.f("      __builtin_amdgcn_sched_barrier(0);");
        }
      }
      g.f("      hst1 = PMAX(hst1); hst2 = PMAX(hst2);");
      g.f("      const double hw2 = (hs6 > 0.5 && hst2 <= 0.375 * hst1) ? 1.0 : 0.0;");
      for (int F = 0; F < nf; ++F)
        if (CD)
          g.f("      { const double hx = fma(hw2, hxb%d, hxa%d); hcx%d = hx; hstep = fmax(hstep, fabs(hx)); hN = fma(hx, hx, hN); }", F, F, F);
        else
        g.f("      { const double hx = fma(hw2, hxb%d, %s); hxl[%d + lane] = hx; hstep = fmax(hstep, fabs(hx)); hN = fma(hx, hx, hN); }", F,
            pv ? ("hxl[" + std::to_string(64 * F) + " + lane]").c_str() : ("hxa" + std::to_string(F)).c_str(), 64 * F);
    } else {
    for (int F = 0; F < nf; ++F) {
      std::string e;
      for (int k = 0; k < HK; ++k)
        e += (k ? " + hr" : "hr") + std::to_string(k) + " * " +
             (pv ? "hqb[" + std::to_string(4 * (k * nf + F)) + " + c]" : "hq" + std::to_string(k) + "_" + std::to_string(F));
      if (CD) g.f("      { const double hx = -(%s); hcx%d = hx; hstep = fmax(hstep, fabs(hx)); hN = fma(hx, hx, hN); }", e.c_str(), F);
      else
      g.f("      { const double hx = -(%s); hxl[%d + lane] = hx; hstep = fmax(hstep, fabs(hx)); hN = fma(hx, hx, hN); }", e.c_str(), 64 * F);
    }
    }
    g.f("      hstep = PMAX(hstep); hN = PSUM(hN);");
    for (int j = 0; j < HK; ++j)
      for (int k = j; k < HK; ++k)
        g.f("      hM = fma(%shr%d * hr%d, hm%d_%d, hM);", j == k ? "" : "2.0 * ", j, k, j, k);  // M is symmetric: Q_j . G_k = G_j^T (A + lambda I)^-1 G_k
    for (int k = 1; k < HK; ++k) g.f("      hss = fma(hr%d, hr%d, hss); hmr = fmax(hmr, fabs(hr%d));", k, k, k);
    if (CD) {
      g.f("      hc_step = hstep; hc_M = hM; hc_N = hN; hc_ss = hss; hc_mr = hmr; hc_dmax = hs0; hc_pmin = hs1; hc_pmax = hs5; hc_ok = hs4;");
      g.f("    }");
    } else {
    g.f("      const int hq_ = lane >> 2;");
    g.f("      hsc[0 + hq_] = hstep; hsc[16 + hq_] = hM; hsc[32 + hq_] = hN; hsc[48 + hq_] = hss; hsc[64 + hq_] = hmr;");
    g.f("      hsc[80 + hq_] = hs0; hsc[96 + hq_] = hs1; hsc[112 + hq_] = hs5; hsc[128 + hq_] = hs4;");
    g.f("      head_ready = true;");
    g.f("    }");
    }
  }
  if (CD) {
    g.f("    const int hist = 1; const bool cold = false; const double lambda_carry = 0.0; (void)hist; (void)cold; (void)lambda_carry;");
    if (pair_state_lds) g.f("    __shared__ int lmk[%d];  // per-quad counters [slot][quad side]", 16 * 6);
  } else {
  if (pair_state_lds) {
    g.f("    __shared__ int lmk[%d];  // per-quad counters [slot][quad side]", 16 * 6);
    g.f("    int& hist = lmk[0 + qs]; hist = 1;                // solved states in the history (the design state counts)");
    g.f("    int& steps_done = lmk[16 + qs]; steps_done = 0;   // steps of this chain solved since its (re)start");
  } else {
  g.f("    int hist = 1;        // solved states in the history (the design state counts)");
  g.f("    int steps_done = 0;  // steps of this chain solved since its (re)start");
  }
  g.f("    bool cold = false;  // the previous chain step failed: restart from the design state, not the predictor");
  g.f("    %s  // damping a converged chain step ended with (0: none)", state_ref("lambda_carry", "0.0").c_str());
  }
  // targets: the next step's values are fetched while the current step is being solved, and the two
  // previous steps' values (secant predictor) stay in registers
  if (!pv && !CD) {
    g.f("    const bool model_lds = !PG && a.predictor != nullptr && a.predictor_len <= %d;", kPredictorLdsDoubles);
    g.f("    if (model_lds) {");
    g.f("      for (int k = lane; k < (int)a.predictor_len; k += 64) pls[k] = a.predictor[k];");
    g.f("      __syncthreads();");
    g.f("    }");
  }
  if (CD) {
    g.f("    {  // the unit's one problem");
    g.f("      const bool valid = have;");
    g.f("      const long long bb = first_b;");
    for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
    g.f("      const bool from_model = false;");
  } else {
  g.f("    for (long long b = first_b; wave_any(have && b < last_b); ++b) {");
  g.f("      const bool valid = have && b < last_b;");
  g.f("      const long long bb = valid ? b : last_b - 1;");
  g.f("      const long long nb = b + 1 < last_b ? b + 1 : last_b - 1;");
  for (int t = 0; t < T; ++t) g.f("      const double tv%d = tn%d;", t, t);
  for (int t = 0; t < T; ++t) g.f("      tn%d = a.targets[nb * %d + %s];", t, prog_targets, ev.target_slot(t).c_str());
  g.f("      bool from_model = false;");
  }
  // Chain heads (and the step after, which has no secant history yet) start from the polynomial model fitted
  // by okx_program_fit_predictor instead of the design state / the previous solution:
  // x(t) = sum_k coef_k prod_t T_{k_t}(u_t), Chebyshev polynomials in the targets normalised to the fitted box
  // (clamped to it).  Table: per target (mid, 1 / half-range, degree), the total-degree cap, the table length
  // (unused here), then one [free points][4] block per term in the loop order below (lane 3 reads the zero pad).  The loops
  // are runtime loops on purpose: unrolled over two varying targets the evaluation costs the whole kernel
  // ~20 % (registers, code size) whether it runs or not.  Own-geometry launches only; a restart after a failed
  // step goes back to the design state.  Not generated in pair mode: that kernel is register-bound and the mere
  // presence of the block cost the axle 18 % on chained grids for a 3 % gain (profiles/r01/config_sweep_pred.txt).
  if (!CD && !pv) {
    const int TT = prog_targets;
    std::vector<int> ordinal(program.n_points, 0);  // program point -> its free ordinal
    for (int k = 0; k < program.n_free; ++k) ordinal[program.free_point[k]] = k;
    g.f("      if (!PG && a.predictor != nullptr && (steps_done < 2 || a.predictor_mode == 2) && !cold) {");
    for (int F = 0; F < nf; ++F) g.f("        double pa%d = 0.0;", F);
    auto up = [&](int t, const char* what) { return t == 0 ? std::string(what) + "S" : std::string(what) + std::to_string(t - 1); };
    // header + term loops, reading through `pp` (global memory or the LDS copy made in the prologue)
    auto evaluate = [&](const char* base) {
      g.f("          const double* pp = %s;", base);
      for (int t = 0; t < TT; ++t) {
        if (pv)  // the halves carry their own target lists: read the program's targets
          g.f("          const double pu%d = fmin(fmax((a.targets[bb * %d + %d] - pp[%d]) * pp[%d], -1.0), 1.0);", t, TT, t, 3 * t, 3 * t + 1);
        else
          g.f("          const double pu%d = fmin(fmax((tv%d - pp[%d]) * pp[%d], -1.0), 1.0);", t, t, 3 * t, 3 * t + 1);
        g.f("          const int pD%d = (int)pp[%d];", t, 3 * t + 2);
      }
      g.f("          const int prS = (int)pp[%d];  // total-degree budget", 3 * TT);
      g.f("          const double pwS = 1.0;");
      g.f("          const double* pq = pp + %d;", 3 * TT + 2);
      for (int t = 0; t < TT; ++t) {  // one loop level per target: pc = T_i(u), pn = T_{i+1}(u)
        g.f("          { double pc%d = 1.0, pn%d = pu%d;", t, t, t);
        if (t == TT - 1) {  // innermost level: a simple trip count, so that four terms' loads are in flight together
          g.f("          const int pe%d = pD%d < %s ? pD%d : %s;", t, t, up(t, "pr").c_str(), t, up(t, "pr").c_str());
          g.f("          _Pragma(\"unroll 4\")");
          g.f("          for (int pi%d = 0; pi%d <= pe%d; ++pi%d) {", t, t, t, t);
        } else
        g.f("          for (int pi%d = 0; pi%d <= pD%d && pi%d <= %s; ++pi%d) {", t, t, t, t, up(t, "pr").c_str(), t);
        g.f("            const double pw%d = %s * pc%d; const int pr%d = %s - pi%d;", t, up(t, "pw").c_str(), t, t, up(t, "pr").c_str(), t);
      }
      g.f("            (void)pr%d;", TT - 1);
      for (int F = 0; F < nf; ++F) {
        const int pt = ev.fp(F);
        std::string off = pv ? Gen::sel(4 * ordinal[pv->pt[0][pt]], 4 * ordinal[pv->pt[1][pt]]) : std::to_string(4 * ordinal[pt]);
        g.f("            pa%d = fma(pw%d, pq[%s + c], pa%d);", F, TT - 1, off.c_str(), F);
      }
      g.f("            pq += %d;", 4 * program.n_free);
      for (int t = TT - 1; t >= 0; --t) {
        g.f("            { const double nn = fma(2.0 * pu%d, pn%d, -pc%d); pc%d = pn%d; pn%d = nn; }", t, t, t, t, t, t);
        g.f("          } }");
      }
    };
    if (!pv) {
      g.f("        if (model_lds) {");
      evaluate("pls");
      g.f("        } else {");
      evaluate("a.predictor");
      g.f("        }");
    } else {
      g.f("        {");
      evaluate("a.predictor");
      g.f("        }");
    }
    for (int F = 0; F < nf; ++F) g.f("        xq%d = xp%d; xp%d = x%d; x%d = pa%d;", F, F, F, F, F, F);
    g.f("        from_model = true;");
    g.f("      }");
  }
  // Extrapolation along the chain from the solved states x (step k-1), xp (k-2), xq (k-3):
  // two states -> secant x + alpha (x - xp), alpha = the new target increment over the old one; three states on
  // one line of target space with comparable spacing -> the quadratic through them (error O(h^3) instead of
  // O(h^2): one full pass then suffices at the step sizes of the grids and ensembles).  The history shifts either way.
  if (!CD) {
  g.f("      if (!from_model && hist >= 2) {");
  g.f("        double num = 0.0, den = 0.0, nn = 0.0, num2 = 0.0, den2 = 0.0;");
  for (int t = 0; t < T; ++t) {
    const std::string en = ev.target_enable(t);
    g.f("        { const double dn = %s * (tv%d - tp%d), dold = tp%d - tq%d, dolder = tq%d - tr%d;", en.c_str(), t, t, t, t, t, t);
    g.f("          num = fma(dn, dold, num); den = fma(%s * dold, dold, den); nn = fma(dn, tv%d - tp%d, nn);", en.c_str(), t, t);
    g.f("          num2 = fma(%s * dold, dolder, num2); den2 = fma(%s * dolder, dolder, den2); }", en.c_str(), en.c_str());
  }
  g.f("        num = PJOIN_SUM(num); den = PJOIN_SUM(den); nn = PJOIN_SUM(nn); num2 = PJOIN_SUM(num2); den2 = PJOIN_SUM(den2);  // pair mode: targets of both halves");
  g.f("        double alpha = den > 0.0 ? num * fast_rcp(den) : 0.0;  // (Newton-refined reciprocals here and below: an IEEE fp64 division is ~25 instructions)");
  g.f("        alpha = fmin(fmax(alpha, 0.0), 2.0);");
  g.f("        const double beta = den2 > 0.0 ? num2 * fast_rcp(den2) : 0.0;  // old increment over the one before");
  g.f("        const bool line = hist >= 3 && alpha > 0.0 && beta >= 1e-3 && beta <= 2.0 && num * num >= 0.98 * nn * den && num2 * num2 >= 0.98 * den * den2;");
  g.f("        const double bq = line ? fast_rcp(beta) : 1.0;  // spacings in units of the old increment: new = alpha, old = 1, older = bq (1 / bq = beta)");
  g.f("        const double r1q = fast_rcp(1.0 + bq);");
  g.f("        const double l0 = line ? (alpha + 1.0) * (alpha + 1.0 + bq) * r1q : 1.0 + alpha;");
  g.f("        const double l1 = line ? -alpha * (alpha + 1.0 + bq) * beta : -alpha;");
  g.f("        const double l2 = line ? alpha * (alpha + 1.0) * r1q * beta : 0.0;");
  for (int F = 0; F < nf; ++F)
    g.f("        { const double xn = fma(l0, x%d, fma(l1, xp%d, l2 * xq%d)); xq%d = xp%d; xp%d = x%d; x%d = xn; }", F, F, F, F, F, F, F, F);
  g.f("      } else if (!from_model) {");
  for (int F = 0; F < nf; ++F) g.f("        xq%d = xp%d; xp%d = x%d;", F, F, F, F);
  g.f("      }");
  }
  if (pair_state_lds) {
    // (declared once per chain step: the slots are the same every time)
    const int first_slot = n_state_slots;
    g.f("      double lambda = 0.0;");
    for (const char* name : {"Fc", "dmax", "step_len", "last_step", "mres", "pred", "prev_sl", "piv_lo", "piv_hi"})
      g.f("      %s", state_ref(name, "0.0").c_str());
    g.f("      %s", state_ref("nu", "2.0").c_str());
    n_state_slots = first_slot + 10;
  } else
  g.f("      double Fc = 0.0, lambda = 0.0, nu = 2.0, dmax = 0.0, step_len = 0.0, last_step = 0.0, mres = 0.0, pred = 0.0;");
  if (pair_state_lds) {
    g.f("      int& nfev = lmk[32 + qs]; nfev = 0; int& iters = lmk[48 + qs]; iters = 0; int& nfail = lmk[64 + qs]; nfail = 0;");
    g.f("      int flags = 0;");
  } else
  g.f("      int nfev = 0, iters = 0, flags = 0, nfail = 0;");
  g.f("      int mode = 0;  // 0 first evaluation, 1 trial point, 2 re-evaluation of the accepted point");
  g.f("      bool done = !valid, want_light = false;");
  if (!pair_state_lds) g.f("      double prev_sl = 0.0, piv_lo = 0.0, piv_hi = 0.0;  // pivot range of the last successful factorisation");
  for (int F = 0; F < nf; ++F) g.f("      dx%d = 0.0;", F);
  if (head_ok) {
    // Shared first step.  A chain head starts at its geometry's design state, where the constraint
    // residuals vanish and the Jacobian, J^T J and its damped factorisation are the same for EVERY problem of that
    // geometry: only the target residuals differ.  The first LM step is therefore dx = -sum_t r_t Q_t with
    // Q_t = (J^T J + lambda I)^-1 J^T e_t tabulated once per geometry (okx_quad_head_*), and the problem enters the
    // loop below exactly where its own first pass would have left it: trial point x + dx in hand (mode 1), cost and
    // damping of the design state, predicted reduction 0.5 (lambda |dx|^2 - dx . g) from the table's Gram matrices.
    if (CD) {
    g.f("#define HEAD_APPLY \\");
    const size_t head_apply_from = g.out.size();
    g.f("      {");
    g.f("        const bool at_design = valid && hc_ok > 0.5;  // the table is good");
    g.f("        if (at_design) {");
    for (int F = 0; F < nf; ++F) g.f("          dx%d = hcx%d;", F, F);
    g.f("          const double hstep = hc_step;");
    g.f("          Fc = 0.5 * hc_ss; mres = hc_mr; dmax = hc_dmax; lambda = a.lambda0 * dmax;");
    g.f("          step_len = hstep; pred = 0.5 * fma(lambda, hc_N, hc_M); iters = 1; mode = 1;");
    g.f("          piv_lo = hc_pmin - lambda; piv_hi = hc_pmax;");
    g.f("          if (hstep <= a.step_tol) { flags |= INFO_CONVERGED; last_step = hstep; done = true; }");
    g.f("          else {");
    g.f("            want_light = hstep <= 1e-3 && (100.0 * lambda * fast_rcp(hc_pmin) + hstep) * hstep <= a.step_tol;");
    g.f("            prev_sl = hstep;");
    g.f("          }");
    g.f("        }");
    g.f("      }");
    {  // (a macro body: strip the comments, continue every line but the last)
      std::string body = g.out.substr(head_apply_from), cont;
      g.out.resize(head_apply_from);
      size_t pos = 0;
      while (pos < body.size()) {
        size_t eol = body.find('\n', pos);
        std::string line = body.substr(pos, eol - pos);
        const size_t cm = line.find("//");
        if (cm != std::string::npos) line.resize(cm);
        pos = eol + 1;
        cont += line + (pos < body.size() ? " \\\n" : "\n");
      }
      g.out += cont;
    }
    g.f("      HEAD_APPLY");
    } else {
    g.f("      if (head_ready && b == first_b) {  // wave-uniform: every quad of the wavefront is at its unit's first problem");
    g.f("        const int hq_ = lane >> 2;");
    g.f("        const bool at_design = valid && hist == 1 && !from_model && hsc[128 + hq_] > 0.5;  // x is the design state, the table is good");
    g.f("        if (at_design) {");
    for (int F = 0; F < nf; ++F) g.f("          dx%d = hxl[%d + lane];", F, 64 * F);
    g.f("          const double hstep = hsc[0 + hq_];");
    g.f("          Fc = 0.5 * hsc[48 + hq_]; mres = hsc[64 + hq_]; dmax = hsc[80 + hq_]; lambda = a.lambda0 * dmax;");
    g.f("          step_len = hstep; pred = 0.5 * fma(lambda, hsc[32 + hq_], hsc[16 + hq_]); iters = 1; mode = 1;");
    g.f("          piv_lo = hsc[96 + hq_] - lambda; piv_hi = hsc[112 + hq_];");
    g.f("          if (hstep <= a.step_tol) { flags |= INFO_CONVERGED; last_step = hstep; done = true; }");
    g.f("          else {");
    g.f("            want_light = hstep <= 1e-3 && (100.0 * lambda * fast_rcp(hsc[96 + hq_]) + hstep) * hstep <= a.step_tol;");
    g.f("            prev_sl = hstep;");
    g.f("          }");
    g.f("        }");
    g.f("      }");
    }
  }
  if (timeline) g.f("      int tl_pass = 2;");
  stamp("1");
  const bool fast_loop = CD && !dev_switch("quad_no_fast");  // (developer switch: the cold body with the general loop only)
  if (fast_loop) {
    // ---- the cold body's fast loop ----
    // The passes of a cold start whose every trial point is accepted - what a sweep inside the reach does - written for
    // exactly that: the quads that are still iterating run under ONE exec mask per pass (no per-statement predication),
    // no modes, no rejected-step bookkeeping.  The first quad that needs anything else (a rejected or non-finite trial
    // point, a stop on the cost test, the iteration cap, a failed factorisation, a confirming pass whose cost rose, no
    // usable table) hands its WAVEFRONT over to the general loop below, in place: the state the fast loop keeps IS the
    // general loop's state (mode 1, trial step in hand), so that loop takes the very evaluation the fast loop was about to
    // judge once more and goes on as the general body would have - same decisions, same counts, same answers.
    // Same evaluation, factorisation and update formulas as there (eval_src / the solve step / light_src are the same text).
    g.f("      bool hand_over = wave_any(valid && mode != 1 && !done);  // a quad without a usable first step");
    // full passes while some quad that is still iterating has no step in hand that is predicted to be its last ...
    g.f("      const bool lights = a.confirm == 0;  // confirming passes are on");
    g.f("      while (!hand_over && wave_any(!done && !(lights && want_light))) {");
    stamp("tl_pass < 12 ? tl_pass++ : 12");
    g.f("        if (!done) {");
    g.f("    want_light = false;");
    for (int F = 0; F < nf; ++F) g.f("    p%d = x%d + dx%d;", ev.fp(F), F, F);
    g.out += eval_src;
    g.out += couple_eval;
    g.f("    const double Ft = 0.5 * ss;");
    g.f("    const double rho = (Fc - Ft) * fast_rcp(pred);");
    g.f("    const bool plain = Ft < 1e300 && pred > 0.0 && rho > 1e-4 && !(Fc - Ft <= a.ftol * Fc && pred <= a.ftol * Fc) && iters < a.max_iter;");
    g.f("    if (wave_any(!plain)) { hand_over = true; break; }  // (nothing of this evaluation has been used yet)");
    for (int F = 0; F < nf; ++F) g.f("    x%d = p%d;", F, ev.fp(F));
    g.f("    last_step = step_len; Fc = Ft; mres = mres_new; ++nfev;");
    g.f("    { const double t = 2.0 * rho - 1.0;");
    g.f("      lambda *= (rho > 0.99 && (step_len <= 1.0 || Ft <= 1e-2)) ? 1e-3 : (rho > 0.9 ? 0.1 : fmax(1.0 / 3.0, 1.0 - t * t * t)); }");
    if (timeline) g.f("    OKX_TL(10)");
    emit_solve_step();
    g.f("    double sl = 0.0, pr = 0.0, dd = 0.0;");
    for (int F = 0; F < nf; ++F) g.f("    sl = fmax(sl, fabs(nx%d));", F);
    g.f("    sl = PMAX(sl);");
    for (int F = 0; F < nf; ++F) g.f("    pr = fma(nx%d, fma(lambda, nx%d, -gn%d), pr); dd = fma(nx%d, nx%d, dd);", F, F, F, F, F);
    g.f("    pr = 0.5 * PSUM(pr);");
    g.f("    dd = PSUM(dd);");
    g.f("    const double rq = dd > 0.0 ? (2.0 * pr - lambda * dd) * fast_rcp(dd) - lambda : 1e300;");
    g.f("    ++iters;");
    // every pivot positive = the smallest one is (the eighteen separate tests of the general loop are never asked for here);
    // a NaN anywhere in the matrix reaches the step, hence dd.  (Pair mode: `ok` is part of the solve text - both halves
    // must factor - so it is there anyway.)  A quad whose factorisation failed is left as the general loop leaves it.
    g.f("    const bool factored = %s && dd == dd;", pv ? "ok" : "pmin > 0.0");
    g.f("    if (factored) {");
    if (pv) g.f("      piv_lo = fmin(fmin(pmin - lambda, pcoup), rq); piv_hi = pmax;");
    else g.f("      piv_lo = fmin(pmin - lambda, rq); piv_hi = pmax;");
    for (int F = 0; F < nf; ++F) g.f("      dx%d = nx%d;", F, F);
    g.f("      step_len = sl; pred = pr;");
    g.f("      if (sl <= a.step_tol) { flags |= INFO_CONVERGED; last_step = sl; done = true; }");
    g.f("      else {");
    g.f("        const double cq = prev_sl > 0.0 ? fmax(3.0 * sl * fast_rcp(prev_sl * prev_sl), 1e-3) : 1.0;");
    g.f("        const double rho_lin = 100.0 * lambda * fast_rcp(pmin);");
    g.f("        want_light = sl <= 1e-3 && (rho_lin + cq * sl) * sl <= a.step_tol;");
    g.f("        prev_sl = sl;");
    g.f("      }");
    g.f("    } else {");
    g.f("      lambda = fmax(lambda * 10.0, 1e-12 * dmax);");
    g.f("      if (++nfail > 60 || !(lambda < 1e30)) { flags |= INFO_FAILED; done = true; }");
    g.f("      mode = 2;");
    g.f("    }");
    g.f("    if (wave_any(!factored)) { hand_over = true; break; }");
    if (timeline) g.f("    OKX_TL(11)");
    g.f("        }  // quads still iterating");
    g.f("      }  // full passes");
    // ... then ONE confirming pass (residuals only) for the quads that are left, all of which want it (the general loop's
    // rule: a confirming pass only when every quad still iterating asks for one).  A quad whose cost rose there goes on
    // with full passes: in the general loop, with the rest of its wavefront.
    if (light_ok) {
      g.f("      if (!hand_over && wave_any(!done)) {");
      stamp("tl_pass < 12 ? tl_pass++ : 12");
      g.f("        if (!done) {");
      for (int F = 0; F < nf; ++F) g.f("      p%d = x%d + dx%d;", ev.fp(F), F, F);
      g.out += light_src;
      g.out += couple_light;
      g.f("      const double Fl = 0.5 * ss;");
      g.f("      ++nfev;");
      g.f("      if (Fl == Fl && Fl <= Fc * (1.0 + 1e-6) + 1e-28) {");
      for (int F = 0; F < nf; ++F) g.f("        x%d = p%d;", F, ev.fp(F));
      g.f("        Fc = Fl; mres = mres_new; last_step = step_len; flags |= INFO_CONVERGED; done = true;");
      g.f("      } else {");
      g.f("        want_light = false;");
      g.f("      }");
      g.f("        }");
      g.f("      }");
    }
    // (whatever is not done by now - handed over, or a confirming pass that did not confirm - is the general loop's)
  }
  g.f("      while (wave_any(!done)) {");
  stamp("tl_pass < 12 ? tl_pass++ : 12");
  if (light_ok) {
    // Confirming pass: every active problem of this wavefront has a step in hand that is
    // predicted to land within step_tol of its solution.  Apply it, evaluate the residuals only
    // (no Jacobian, no factorisation) and finish if the cost did not rise; otherwise the same
    // point goes through a full pass next.
    g.f("    if (a.confirm == 0 && !wave_any(!done && !want_light)) {");
    for (int F = 0; F < nf; ++F) g.f("      p%d = x%d + dx%d;", ev.fp(F), F, F);
    g.out += light_src;
    g.out += couple_light;
    g.f("      const double Fl = 0.5 * ss;");
    g.f("      if (!done) {");
    g.f("        ++nfev;");
    g.f("        if (Fl == Fl && Fl <= Fc * (1.0 + 1e-6) + 1e-28) {");
    for (int F = 0; F < nf; ++F) g.f("          x%d = p%d;", F, ev.fp(F));
    g.f("          Fc = Fl; mres = mres_new; last_step = step_len; flags |= INFO_CONVERGED; done = true;");
    g.f("        } else {");
    g.f("          want_light = false;");
    g.f("        }");
    g.f("      }");
    g.f("      continue;");
    g.f("    }");
    g.f("    want_light = false;  // mixed wavefront: everybody takes the full pass");
  }
  // evaluation point
  for (int F = 0; F < nf; ++F) g.f("    p%d = mode == 2 ? x%d : x%d + dx%d;", ev.fp(F), F, F, F);
  g.out += eval_src;
  g.out += couple_eval;
  g.f("    const double Ft = 0.5 * ss;");
  // LM decision (mirrors okx_solve_kernel)
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
  g.f("    if (wave_any(mode == 0)) {  // largest diagonal entry scales the first damping");
  for (int F = 0; F < nf; ++F)
    g.f("      diag = fmax(diag, c == 0 ? %s : (c == 1 ? %s : (c == 2 ? %s : 0.0)));", Gen::A(F, F, 0).c_str(),
        Gen::A(F, F, 1).c_str(), Gen::A(F, F, 2).c_str());
  g.f("      diag = PMAX(diag);");
  g.f("    }");
  // gradient stop (okx_solve_opts.grad_tol): > 0 the absolute form max |J^T r|; < 0 MINPACK's scaled form
  // max_j |(J^T r)_j| / (|J_j| |r|) (lmder's gnorm, what the reference's gtol means: solver.py:158-169)
  g.f("    if (a.grad_tol > 0.0) {");
  for (int F = 0; F < nf; ++F) g.f("      gm = fmax(gm, fabs(gn%d));", F);
  g.f("      gm = PMAX(gm);");
  g.f("    } else if (a.grad_tol < 0.0) {");
  g.f("      const double rr = 2.0 * Ft;");
  for (int F = 0; F < nf; ++F)
    g.f("      { const double cn = (c == 0 ? %s : (c == 1 ? %s : (c == 2 ? %s : 0.0))) * rr; gm = fmax(gm, cn > 0.0 ? fabs(gn%d) * __builtin_amdgcn_rsq(cn) : 0.0); }",
        Gen::A(F, F, 0).c_str(), Gen::A(F, F, 1).c_str(), Gen::A(F, F, 2).c_str(), F);
  g.f("      gm = PMAX(gm);");
  g.f("    }");
  g.f("    if (!done) {");
  g.f("      ++nfev;");
  g.f("      if (stop) flags |= INFO_CONVERGED;");
  // A solve that ends on the cost test (no further reduction, actual or predicted) WITHOUT meeting its rows to a hundredth
  // of the acceptance tolerance sits at a compromise point: a local minimum with a residual, i.e. beyond kinematic
  // lock-out, where J is singular.  The pivot / Rayleigh tests cannot certify that while the damping is above the weak
  // direction's curvature (rocker axle in rebound just beyond lock-out: cond(J) 9.6e8, smallest pivot 1.9 lambda), so the
  // ending itself raises the advisory bit.
  g.f("      if (compromise && mres_new > 0.01 * a.residual_tolerance) flags |= INFO_ILL_CONDITIONED;");
  g.f("      if (accept) {");
  g.f("        if (mode != 2) {");
  for (int F = 0; F < nf; ++F) g.f("          x%d = p%d;", F, ev.fp(F));
  g.f("          if (mode == 1) last_step = step_len;");
  g.f("          nu = 2.0;");
  g.f("        }");
  g.f("        Fc = Ft; mres = mres_new;");
  g.f("        if (!stop) {");
  g.f("          if (mode == 0) {");
  g.f("            // a warm-started chain step continues with the damping its predecessor ended with");
  g.f("            dmax = diag; lambda = a.lambda0 * dmax;");
  g.f("            if (lambda_carry > 0.0) lambda = fmin(lambda, lambda_carry);");
  g.f("            // a start from the fitted model is a near-converged start: the damping only adds a linear");
  g.f("            // contraction floor there (MacPherson grid: 3.0 -> 2.0 evaluations); it grows back if a step fails");
  g.f("            if (from_model) lambda *= 1e-3;");
  g.f("          }");
  g.f("          else if (mode == 1 && rho > 1e-4) {");
  g.f("            // Nielsen's update; a step whose gain ratio shows the quadratic model to be accurate");
  g.f("            // (rho > 0.9) drops the damping by 10 (Marquardt), one that matches it to a percent NEAR the solution - a step");
  g.f("            // of at most 1 mm, or rows met to ~0.1 mm (cost <= 1e-2) - by 1000 (far from the solution the collapse");
  g.f("            // costs dozens of rejected steps: MacPherson cold starts at 0.99 of the rack's reach), so that the");
  g.f("            // final steps are Gauss-Newton steps without a linear contraction floor, like MINPACK's par = 0");
  g.f("            // (after the second-order shared first step two such steps finish a cold start: the damping must");
  g.f("            //  be out of the way by the second)");
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
  if (!CD) {
  g.f("    if (a.trace && valid && bb == a.trace_problem && c == 0 && nfev < 256) {");
  g.f("      double* tr = a.trace + 8 * nfev;");
  g.f("      tr[0] = mode; tr[1] = Ft; tr[2] = Fc; tr[3] = lambda; tr[4] = step_len; tr[5] = rho; tr[6] = accept ? 1.0 : 0.0; tr[7] = done ? 1.0 : 0.0;");
  g.f("    }");
  }
  if (timeline) g.f("    OKX_TL(10)");
  g.f("    const bool solve_now = !done && accept;");
  g.f("    if (!done && !accept) mode = 2;");
  g.f("    if (wave_any(solve_now)) {");
  emit_solve_step();
  g.f("    double sl = 0.0, pr = 0.0, dd = 0.0;");
  for (int F = 0; F < nf; ++F) g.f("    sl = fmax(sl, fabs(nx%d));", F);
  g.f("    sl = PMAX(sl);");
  for (int F = 0; F < nf; ++F) g.f("    pr = fma(nx%d, fma(lambda, nx%d, -gn%d), pr); dd = fma(nx%d, nx%d, dd);", F, F, F, F, F);
  g.f("    pr = 0.5 * PSUM(pr);  // predicted cost reduction of this step (gain-ratio denominator)");
  // Rayleigh quotient of the step in the damped matrix M = J^T J + lambda I: dx^T M dx / dx^T dx = -dx.g / |dx|^2, an UPPER
  // bound of M's smallest eigenvalue - and, since dx = -M^-1 g amplifies the weakest direction most, a close one where J
  // is (nearly) singular.  The pivots bound that eigenvalue from below only (every pivot >= lambda_min(M)): a singular
  // direction spread over several pivots leaves all of them well above the damping (rocker axle in rebound just beyond
  // lock-out: cond(J) 8.6e8, smallest pivot 8.7e-6 above lambda).  rq = quotient - lambda joins the conditioning test.
  g.f("    dd = PSUM(dd);");
  g.f("    const double rq = dd > 0.0 ? (2.0 * pr - lambda * dd) * fast_rcp(dd) - lambda : 1e300;");
  g.f("    if (solve_now) {");
  g.f("      ++iters;");
  g.f("      if (ok) {");
  if (pv) {
    g.f("        piv_lo = fmin(fmin(pmin - lambda, pcoup), rq); piv_hi = pmax;  // what the damping did not put there (halves' pivots, tied mode, Rayleigh bound)");
  }
  else g.f("        piv_lo = fmin(pmin - lambda, rq); piv_hi = pmax;  // what the damping did not put there (pivots, Rayleigh bound)");
  for (int F = 0; F < nf; ++F) g.f("        dx%d = nx%d;", F, F);
  g.f("        step_len = sl; pred = pr;");
  g.f("        if (sl <= a.step_tol) { flags |= INFO_CONVERGED; last_step = sl; done = true; }");
  g.f("        else {");
  g.f("          // Next correction predicted as rho |dx| + C |dx|^2: the damping's linear contraction");
  g.f("          // rho = lambda / sigma_min(J^T J), bounded with the smallest pivot (x 100), plus the");
  g.f("          // Gauss-Newton curvature term with C = 3 x the observed |dx| / |dx-|^2, or 1 / mm on a");
  g.f("          // problem's first step (two orders above a linkage's curvature / stiffness ratio).");
  g.f("          const double cq = prev_sl > 0.0 ? fmax(3.0 * sl * fast_rcp(prev_sl * prev_sl), 1e-3) : 1.0;");
  g.f("          const double rho_lin = 100.0 * lambda * fast_rcp(pmin);");
  g.f("          want_light = sl <= 1e-3 && (rho_lin + cq * sl) * sl <= a.step_tol;");
  g.f("          prev_sl = sl;");
  g.f("        }");
  g.f("        mode = 1;");
  g.f("      } else {");
  g.f("        lambda = fmax(lambda * 10.0, 1e-12 * dmax);");
  g.f("        if (++nfail > 60 || !(lambda < 1e30)) { flags |= INFO_FAILED; done = true; }");
  g.f("        mode = 2;");
  g.f("      }");
  g.f("    }");
  if (timeline) g.f("    OKX_TL(11)");
  g.f("    }  // any quad solves");
  g.f("      }  // LM passes");
  stamp("13");
  // final state and output
  g.f("      {");
  for (int F = 0; F < nf; ++F) g.f("    p%d = x%d;", ev.fp(F), F);
  Gen fin(P, pv);
  fin.uid = 100000;
  fin.hoisted_names = ev.hoisted_names;
  for (int e = 0; e < P.n_derived; ++e)
    if (!fin.derived_op(e, false)) {
      *why = fin.why;
      body_failed = true;
      return;
    }
  if (EV) g.f("    {  // (the evaluated module needs every derived point: records or not)");
  else
  g.f("    if (a.out_mode == 0) {  // the derived points only matter to the full records");
  g.out += fin.out;
  g.f("    }");
  final_src = fin.out;
  g.f("    if (mres > a.residual_tolerance) flags |= INFO_RESIDUAL_EXCEEDED;");
  g.f("    if (piv_hi > 0.0 && piv_lo <= ILL_CONDITIONED_PIVOT_RATIO * piv_hi) flags |= INFO_ILL_CONDITIONED;");
  // Record stores.  Independent problems (chain_len 1): the 16 problems of a wavefront are
  // consecutive, so their records form one contiguous block; it is transposed through LDS and
  // written with full-width 16-byte-per-lane stores.  Chains: a quad's problems are far apart in
  // memory, each lane stores its own 8-byte components.
  {
    // okx_solve_opts.output = OKX_OUTPUT_FREE: the solved free points alone, [n_free][3] in the program's free_point order
    std::vector<int> ordinal(program.n_points, 0);
    for (int k = 0; k < program.n_free; ++k) ordinal[program.free_point[k]] = k;
    if (CD) {
      // cold body: the wavefront's free coordinates form one contiguous block (16 consecutive problems): through LDS and out in
      // 16-byte-per-lane rows like the records - whole cache lines for HBM, whole packets for a caller's pinned host buffer
      // (lane-by-lane 8-byte stores leave 24-byte fragments whose merging on the way out depends on timing)
      if (!EVP) g.f("    __shared__ __attribute__((aligned(16))) double fstage[%d * %d];", PPW, 3 * program.n_free);
      g.f("    if (a.out_mode == 1) {");
      g.f("      if (c < 3) {");
      if (pv) g.f("        int q1f = q1; asm volatile(\"\" : \"+v\"(q1f));");
      g.f("        double* st = fstage + quad * %d + c;", 3 * program.n_free);
      for (int F = 0; F < nf; ++F) {
        const int pt = ev.fp(F);
        if (pv) g.f("        st[q1f ? %d : %d] = x%d;", 3 * ordinal[pv->pt[1][pt]], 3 * ordinal[pv->pt[0][pt]], F);
        else g.f("        st[%d] = x%d;", 3 * ordinal[pt], F);
      }
      g.f("      }");
      g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
      g.f("      const long long rem = a.n_problems - wu * %d;", PPW);
      g.f("      const int n_doubles = (int)(rem < %d ? rem : %d) * %d;", PPW, PPW, 3 * program.n_free);
      g.f("      double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * %d * %d);", PPW, 3 * program.n_free);
      g.f("      const double2* src = reinterpret_cast<const double2*>(fstage);");
      g.f("      for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
      g.f("      if ((n_doubles & 1) && lane == 0) a.out_pos[wu * %d * %d + n_doubles - 1] = fstage[n_doubles - 1];", PPW, 3 * program.n_free);
      g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();");
      g.f("    }");
      g.f("    if (false) { long long bf = 0;");
    } else {
    g.f("    if (a.out_mode == 1 && valid && c < 3) {");
    g.f("      long long bf = bb; asm volatile(\"\" : \"+v\"(bf));");
    }
    g.f("      double* o = a.out_pos + bf * %d + c;", 3 * program.n_free);
    for (int F = 0; F < nf; ++F) {
      const int pt = ev.fp(F);
      const std::string off = pv ? Gen::sel(3 * ordinal[pv->pt[0][pt]], 3 * ordinal[pv->pt[1][pt]]) : std::to_string(3 * ordinal[pt]);
      g.f("      o[%s] = x%d;", off.c_str(), F);
    }
    g.f("    }");
  }
  if (EVP) {
    // evaluated axle: both halves' records are staged in any case (the epilogue gathers both corners' roles from them) and
    // written as the output mode says - whole rows where the wavefront's eight records are contiguous
    g.f("    if (c < 3) {");
    g.f("      int q1s = q1; asm volatile(\"\" : \"+v\"(q1s));");
    g.f("      double* st = stage + quad * %d + c;", RECP);
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 >= 0) g.f("      st[q1s ? %d : %d] = p%d;", 3 * k1, 3 * k0, P.out_point[k]);
      else g.f("      if (!q1s) st[%d] = p%d;", 3 * k0, P.out_point[k]);
    }
    for (size_t k = 0; k < pv->shared_out.size(); ++k)
      g.f("      if (!q1s) st[%d] = gp[%d + c];", 3 * pv->shared_out[k], 3 * pv->shared_pt[k]);
    g.f("    }");
    g.f("    EV_WAVE_SYNC();");
    g.f("    if (a.out_mode == 0) {");
    g.f("      if (unit_len == 1) {");
    g.f("        const long long rem = a.n_problems - wu * %d;", PPW);
    g.f("        const int n_doubles = (int)(rem < %d ? rem : %d) * %d;", PPW, PPW, RECP);
    g.f("        double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * %d * %d);", PPW, RECP);
    g.f("        const double2* src = reinterpret_cast<const double2*>(stage);");
    g.f("        for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
    g.f("        if ((n_doubles & 1) && lane == 0) a.out_pos[wu * %d * %d + n_doubles - 1] = stage[n_doubles - 1];", PPW, RECP);
    g.f("      } else if (valid) {  // chains: a problem's record by its own eight lanes");
    g.f("        double* o = a.out_pos + bb * %d;", RECP);
    g.f("        for (int i = lane & 7; i < %d; i += 8) o[i] = stage[quad * %d + i];", RECP, RECP);
    g.f("      }");
    g.f("    }");
    g.f("    if (false) {");
  } else if (EV) {
    // evaluated module: the record is staged in any case (the epilogue gathers the metric roles from it) and written as
    // the output mode says - whole rows where the wavefront's records are contiguous
    g.f("    __shared__ __attribute__((aligned(16))) double stage[16 * %d];", 3 * P.n_out);
    g.f("    if (c < 3) {");
    g.f("      double* st = stage + quad * %d + c;", 3 * P.n_out);
    for (int k = 0; k < P.n_out; ++k) g.f("      st[%d] = p%d;", 3 * k, P.out_point[k]);
    g.f("    }");
    g.f("    EV_WAVE_SYNC();");
    g.f("    if (a.out_mode == 0) {");
    g.f("      if (unit_len == 1) {");
    g.f("        const long long rem = a.n_problems - wu * 16;");
    g.f("        const int n_doubles = (int)(rem < 16 ? rem : 16) * %d;", 3 * P.n_out);
    g.f("        double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * 16 * %d);", 3 * P.n_out);
    g.f("        const double2* src = reinterpret_cast<const double2*>(stage);");
    g.f("        for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
    g.f("        if ((n_doubles & 1) && lane == 0) a.out_pos[wu * 16 * %d + n_doubles - 1] = stage[n_doubles - 1];", 3 * P.n_out);
    g.f("      } else if (valid && c < 3) {");
    g.f("        double* o = a.out_pos + bb * %d + c;", 3 * P.n_out);
    for (int k = 0; k < P.n_out; ++k) g.f("        o[%d] = p%d;", 3 * k, P.out_point[k]);
    g.f("      }");
    g.f("    }");
    g.f("    if (false) {");
  } else if (pv && CD) {
    // cold pair body: the wavefront's eight records are one contiguous block as well - staged and written in whole rows
    // (the general body's lane-by-lane 8-byte stores leave 24-byte fragments: 1.3x the bytes on the way to HBM)
    g.f("    if (a.out_mode == 0) {");
    g.f("      __shared__ __attribute__((aligned(16))) double stage[%d * %d];", PPW, 3 * prog_out);
    g.f("      if (c < 3) {");
    g.f("        int q1s = q1; asm volatile(\"\" : \"+v\"(q1s));");
    g.f("        double* st = stage + quad * %d + c;", 3 * prog_out);
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 >= 0) g.f("        st[q1s ? %d : %d] = p%d;", 3 * k1, 3 * k0, P.out_point[k]);
      else g.f("        if (!q1s) st[%d] = p%d;", 3 * k0, P.out_point[k]);
    }
    for (size_t k = 0; k < pv->shared_out.size(); ++k)
      g.f("        if (!q1s) st[%d] = gp[%d + c];", 3 * pv->shared_out[k], 3 * pv->shared_pt[k]);
    g.f("      }");
    g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
    g.f("      const long long rem = a.n_problems - wu * %d;", PPW);
    g.f("      const int n_doubles = (int)(rem < %d ? rem : %d) * %d;", PPW, PPW, 3 * prog_out);
    g.f("      double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * %d * %d);", PPW, 3 * prog_out);
    g.f("      const double2* src = reinterpret_cast<const double2*>(stage);");
    g.f("      for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
    g.f("      if ((n_doubles & 1) && lane == 0) a.out_pos[wu * %d * %d + n_doubles - 1] = stage[n_doubles - 1];", PPW, 3 * prog_out);
    g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();");
    g.f("    }");
    g.f("    if (false) {");
  } else if (pv) {
    g.f("    if (a.out_mode == 0 && valid && c < 3) {");
    // (the record's address is rebuilt from an opaque copy of the problem index: as an induction variable the compiler
    //  keeps one strength-reduced 64-bit address per output point alive across the chain loop and spills all of them)
    g.f("      long long bo = bb; asm volatile(\"\" : \"+v\"(bo));");
    // The two halves' records usually sit a constant distance apart in the output list (left block, right block):
    // one select on the side bit then serves every store.  Otherwise each store selects its own index, on a fresh
    // opaque copy of the side bit so that the selects are made here instead of being kept (and spilled) as invariants.
    long long delta = 0;
    bool uniform_delta = true, first = true;
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 < 0) continue;
      if (first) delta = 3LL * (k1 - k0), first = false;
      else uniform_delta = uniform_delta && delta == 3LL * (k1 - k0);
    }
    g.f("      int q1s = q1; asm volatile(\"\" : \"+v\"(q1s));");
    if (uniform_delta)
      g.f("      double* o = a.out_pos + bo * %d + c + (q1s ? %lld : 0);", 3 * prog_out, delta);
    else
      g.f("      double* o = a.out_pos + bo * %d + c;", 3 * prog_out);
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 >= 0 && uniform_delta)
        g.f("      o[%d] = p%d;", 3 * k0, P.out_point[k]);
      else if (k1 >= 0)
        g.f("      o[q1s ? %d : %d] = p%d;", 3 * k1, 3 * k0, P.out_point[k]);
      else
        g.f("      if (!q1s) o[%d] = p%d;", 3 * k0, P.out_point[k]);
    }
    for (size_t k = 0; k < pv->shared_out.size(); ++k)
      g.f("      if (!q1s) o[%d] = gp[%d + c];", 3 * pv->shared_out[k], 3 * pv->shared_pt[k]);
    g.f("    }");
    g.f("    if (false) {");
  } else {
  g.f("    if (a.out_mode != 0) {");
  g.f("    } else if (unit_len == 1) {");
  g.f("      __shared__ double stage[16 * %d];", 3 * P.n_out);
  g.f("      if (c < 3) {");
  g.f("        double* st = stage + quad * %d + c;", 3 * P.n_out);
  for (int k = 0; k < P.n_out; ++k) g.f("        st[%d] = p%d;", 3 * k, P.out_point[k]);
  g.f("      }");
  if (CD) g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
  else
  g.f("      __syncthreads();");
  g.f("      const long long rem = a.n_problems - wu * 16;");
  g.f("      const int n_doubles = (int)(rem < 16 ? rem : 16) * %d;", 3 * P.n_out);
  g.f("      double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * 16 * %d);", 3 * P.n_out);
  g.f("      const double2* src = reinterpret_cast<const double2*>(stage);");
  g.f("      for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
  g.f("      if ((n_doubles & 1) && lane == 0) a.out_pos[wu * 16 * %d + n_doubles - 1] = stage[n_doubles - 1];", 3 * P.n_out);
  // (cold body: no workgroup barrier here - it is a release fence, i.e. a wait for every record store of this unit to
  //  complete, ~1500 cycles at the end of the ONLY unit most wavefronts have; one wavefront's LDS accesses are ordered anyway)
  if (CD) g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();");
  else
  g.f("      __syncthreads();  // stage is reused by the next unit of this wavefront");
  g.f("    } else if (valid && c < 3) {");
  g.f("      double* o = a.out_pos + bb * %d + c;", 3 * P.n_out);
  for (int k = 0; k < P.n_out; ++k) g.f("      o[%d] = p%d;", 3 * k, P.out_point[k]);
  }
  g.f("    }");
  stamp("14");
  if (CD) {
    // the sixteen 40-byte records of the wavefront are contiguous too: one staged block, 8 bytes per lane
    g.f("    __shared__ __attribute__((aligned(16))) okx_info istage[%d];", PPW);
    g.f("    if (c == 0%s) {", pv ? " && !q1" : "");
    g.f("      okx_info inf; inf.max_residual = mres; inf.cost = Fc; inf.last_step = last_step;");
    g.f("      inf.iterations = iters; inf.nfev = nfev; inf.flags = flags; inf.reserved = 0;");
    g.f("      istage[quad] = inf;");
    g.f("    }");
    g.f("    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
    g.f("    {");
    g.f("      const long long rem = a.n_problems - wu * %d;", PPW);
    g.f("      const int n_words = (int)(rem < %d ? rem : %d) * 5;  // doubles of info records", PPW, PPW);
    g.f("      double* dst = reinterpret_cast<double*>(a.info + wu * %d);", PPW);
    g.f("      const double* src = reinterpret_cast<const double*>(istage);");
    g.f("      if (lane < n_words) dst[lane] = src[lane];");
    g.f("      if (lane + 64 < n_words) dst[lane + 64] = src[lane + 64];");
    g.f("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();");
    g.f("    }");
  } else {
  g.f("    if (valid && c == 0%s) {", pv ? " && !q1" : "");
  g.f("      okx_info inf; inf.max_residual = mres; inf.cost = Fc; inf.last_step = last_step;");
  g.f("      inf.iterations = iters; inf.nfev = nfev; inf.flags = flags; inf.reserved = 0;");
  g.f("      a.info[bb] = inf;");
  g.f("    }");
  }
  if (EV) {
    const std::string epi = epilogue_src(CD ? "true" : "unit_len == 1");
    if (epi.empty()) {
      *why = ev.why;
      body_failed = true;
      return;
    }
    g.out += epi;
  }
  stamp("15");
  if (timeline) g.f("    if (a.trace && (threadIdx.x & 63) == 0) a.trace[(gridDim.x + blockIdx.x) * 16 + 1] = (double)__builtin_amdgcn_s_memrealtime();");
  // chains never continue from a state that failed to converge
  if (!CD) {
  for (int t = 0; t < T; ++t) g.f("    tr%d = tq%d; tq%d = tp%d; tp%d = tv%d;", t, t, t, t, t, t);
  g.f("    if (!(flags & INFO_CONVERGED) || (flags & INFO_FAILED)) {");
  if (pv) {  // restart addresses are built here, from a fresh opaque copy of the side bit, not carried through the kernel
    g.f("      int q1r = q1; asm volatile(\"\" : \"+v\"(q1r));");
    g.f("      { const int q1 = q1r; (void)q1;");
  }
  for (int F = 0; F < nf; ++F) g.f("      x%d = ld3(gp + %s + cc, c);", F, ev.point3(ev.fp(F)).c_str());
  if (pv) g.f("      }");
  g.f("      hist = 1; steps_done = 0; lambda_carry = 0.0; cold = true;");
  for (int t = 0; t < T; ++t) g.f("      tp%d = td%d;", t, t);
  g.f("    } else {");
  g.f("      if (hist < 3) ++hist;");
  g.f("      ++steps_done;");
  g.f("      cold = false;");
  g.f("      lambda_carry = lambda;");
  g.f("    }");
  }
  g.f("      }");
  g.f("    }  // chain steps");
  g.f("  }  // wave units");
  g.f("}");
  if (tl_body) g.f("#undef OKX_TL\n#define OKX_TL(k)");
  if (CD) g.f("#undef HEAD_APPLY");
  if (pair_state_lds) g.f("#undef gp\n#undef gq");
  g.f("");
  };  // emit_body
  emit_body(false);
  if (body_failed) return false;
  const bool cold_body = head_ok;
  if (cold_body) emit_body(true);
  if (body_failed) return false;
  if (EVP) {
    // ---- the axle epilogue on GIVEN solved states (okx_evaluate_batch; reference core/sweep.py:217-245 for an AxleSuspension):
    //      one quad per half - its fixed points from the geometry, its free points from the record, its derived points
    //      re-evaluated by the solve kernel's own final-state code -, both halves' records staged, then the epilogue ----
    std::vector<int> oi(NP, -1);
    for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
    for (int F = 0; F < nf; ++F) {
      const int k = oi[ev.fp(F)];
      if (k < 0 || pv->out[0][k] < 0 || pv->out[1][k] < 0) {
        *why = "an evaluated module needs every free point of both halves among the output points";
        return false;
      }
    }
    g.f("struct QEvPosArgs { const double* pos; const double* geom_pos; const double* geom_row_param; double* tan; double* ev;");
    g.f("  long long n_problems, steps_per_geometry; const double* design_pos; const double* row_param; const double* dop_param; EvCfg cfg; EvCfg cfg_r; EvRoleNum roles[8]; };");
    // two bodies: on the program's own geometry the chain constants and the fixed points are the same for every state - read
    // once per wavefront, ahead of a persistent loop over its wave units (okx_evaluate_batch caps that grid at one wavefront
    // per SIMD); with geometry tables they belong to the wave unit
    for (const bool pg : {false, true}) {
    g.f("DEV void okx_quad_evaluate_body_%s(const QEvPosArgs& a) {", pg ? "g" : "u");
    g.f("  constexpr bool PG = %s;", pg ? "true" : "false");
    g.f("  const QEvPosArgs& ea = a;");
    g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 3, q1 = (lane >> 2) & 1, cc = c < 3 ? c : 2;");
    g.out += atan_decl;
    g.out += lds_decl;
    g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
    g.f("  __shared__ __attribute__((aligned(16))) double stage[%d];", PPW * RECP);
    g.f("  __shared__ __attribute__((aligned(16))) double vst[%d];  // velocities [problem][target][half][moving point][3]", PPW * TP * 2 * MV);
    g.f("  __shared__ double vok[%d];", PPW);
    auto constants = [&]() {
      g.out += ev.hoisted;
      g.out += couple_hoist;
      for (int p = 0; p < NP; ++p)
        if (used[p]) g.f("    double p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    };
    if (!pg) {
      g.f("  const double* gp = a.design_pos;");
      g.f("  const double* gq = a.row_param;");
      constants();
    }
    g.f("  for (long long wu = blockIdx.x; wu * %d < a.n_problems; wu += gridDim.x) {", PPW);
    g.f("    long long bb = wu * %d + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;", PPW);
    if (pg) {
      g.f("    const long long geom = bb / a.steps_per_geometry;");
      g.f("    const double* gp = a.geom_pos + geom * %d;", 3 * prog_points);
      g.f("    const double* gq = a.geom_row_param + geom * %d;", 8 * prog_crows);
      constants();
    }
    for (int F = 0; F < nf; ++F) {
      const int k = oi[ev.fp(F)];
      g.f("    p%d = ld3(a.pos + bb * %d + %s + cc, c);", ev.fp(F), RECP, Gen::sel(3 * pv->out[0][k], 3 * pv->out[1][k]).c_str());
    }
    for (int t = 0; t < T; ++t) g.f("    const double tv%d = 0.0;  // target values do not enter the Jacobian", t);
    g.f("    {");
    g.out += final_src;
    g.f("    }");
    g.f("    if (c < 3) {");
    g.f("      double* st = stage + quad * %d + c;", RECP);
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 >= 0) g.f("      st[q1 ? %d : %d] = p%d;", 3 * k1, 3 * k0, P.out_point[k]);
      else g.f("      if (!q1) st[%d] = p%d;", 3 * k0, P.out_point[k]);
    }
    for (size_t k = 0; k < pv->shared_out.size(); ++k)
      g.f("      if (!q1) st[%d] = gp[%d + c];", 3 * pv->shared_out[k], 3 * pv->shared_pt[k]);
    g.f("    }");
    g.f("    EV_WAVE_SYNC();");
    {
      const std::string epi = epilogue_src("true");
      if (epi.empty()) {
        *why = ev.why;
        return false;
      }
      g.out += epi;
    }
    g.f("  }");
    g.f("}");
    }
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evaluate_u(QEvPosArgs a) { okx_quad_evaluate_body_u(a); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evaluate_g(QEvPosArgs a) { okx_quad_evaluate_body_g(a); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evsolve_u(QEvArgs ea) { okx_quad_body<false>(ea.q, ea); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evsolve_g(QEvArgs ea) { okx_quad_body<true>(ea.q, ea); }", waves_per_simd);
    if (cold_body)
      g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evcold_u(QEvArgs ea) { okx_quad_cold_body(ea.q, ea); }", waves_per_simd);
    *src = g.out;
    return true;
  }
  if (EV) {
    // ---- the same epilogue on GIVEN solved states (okx_evaluate_batch; reference core/sweep.py:217-245,
    //      evaluate_solved_sweep): free points from the records, fixed points from the geometry, every derived point
    //      re-evaluated - one launch instead of tangents -> metrics, no tangent tensor in between unless asked for ----
    g.f("struct QEvPosArgs { const double* pos; const double* geom_pos; const double* geom_row_param; double* tan; double* ev;");
    g.f("  long long n_problems, steps_per_geometry; const double* design_pos; const double* row_param; const double* dop_param; EvCfg cfg; };");
    g.f("template <bool PG> DEV void okx_quad_evaluate_body(const QEvPosArgs& a) {");
    g.f("  const QEvPosArgs& ea = a;");
    g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
    g.out += atan_decl;
    g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
    g.f("  __shared__ __attribute__((aligned(16))) double stage[16 * %d];", 3 * P.n_out);
    g.f("  for (long long wu = blockIdx.x; wu * 16 < a.n_problems; wu += gridDim.x) {");
    g.f("    long long bb = wu * 16 + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;");
    g.f("    const long long geom = PG ? bb / a.steps_per_geometry : 0;");
    g.f("    const double* gp = PG ? a.geom_pos + geom * %d : a.design_pos;", 3 * NP);
    g.f("    const double* gq = PG ? a.geom_row_param + geom * %d : a.row_param;", 8 * P.n_crows);
    g.out += ev.hoisted;
    for (int p = 0; p < NP; ++p)
      if (used[p]) g.f("    double p%d = ld3(gp + %d + cc, c);", p, 3 * p);
    {
      std::vector<int> oi(NP, -1);
      for (int k = 0; k < P.n_out; ++k) oi[P.out_point[k]] = k;
      for (int F = 0; F < nf; ++F) g.f("    p%d = ld3(a.pos + bb * %d + %d + cc, c);", ev.fp(F), 3 * P.n_out, 3 * oi[ev.fp(F)]);
    }
    for (int t = 0; t < T; ++t) g.f("    const double tv%d = 0.0;  // target values do not enter the Jacobian", t);
    g.f("    {");
    g.out += final_src;
    g.f("    }");
    g.f("    if (c < 3) {");
    g.f("      double* st = stage + quad * %d + c;", 3 * P.n_out);
    for (int k = 0; k < P.n_out; ++k) g.f("      st[%d] = p%d;", 3 * k, P.out_point[k]);
    g.f("    }");
    g.f("    EV_WAVE_SYNC();");
    {
      const std::string epi = epilogue_src("true");
      if (epi.empty()) {
        *why = ev.why;
        return false;
      }
      g.out += epi;
    }
    g.f("  }");
    g.f("}");
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evaluate_u(QEvPosArgs a) { okx_quad_evaluate_body<false>(a); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evaluate_g(QEvPosArgs a) { okx_quad_evaluate_body<true>(a); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evsolve_u(QEvArgs ea) { okx_quad_body<false>(ea.q, ea); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evsolve_g(QEvArgs ea) { okx_quad_body<true>(ea.q, ea); }", waves_per_simd);
    if (cold_body)
      g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_evcold_u(QEvArgs ea) { okx_quad_cold_body(ea.q, ea); }", waves_per_simd);
    *src = g.out;
    return true;
  }
  if (pv) {
    // ---- pair mode: output positions from free coordinates (okx_expand_positions_batch; the receiving side of a multi-GPU
    //      exchange expands every gathered step).  One quad per half: its fixed points from the design table, its free points
    //      from the input row, its derived points re-evaluated by the solve kernel's own final-state code (same bits); the
    //      wavefront's eight records leave through LDS as whole rows, like the cold pair body's ----
    std::vector<int> ordinal(program.n_points, 0);
    for (int k = 0; k < program.n_free; ++k) ordinal[program.free_point[k]] = k;
    g.f("struct QExpandArgs { const double* free; const double* geom_pos; double* out_pos; long long n_problems, steps_per_geometry;");
    g.f("  const double* design_pos; const double* row_param; const double* dop_param; };");
    g.f("extern \"C\" __global__ void __launch_bounds__(64) okx_quad_expand(QExpandArgs a) {");
    g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 3, q1 = (lane >> 2) & 1, cc = c < 3 ? c : 2;");
    g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
    g.f("  (void)e0; (void)e1; (void)e2;");
    g.out += lds_decl;
    g.f("  __shared__ __attribute__((aligned(16))) double stage[%d * %d];", PPW, 3 * prog_out);
    g.f("  __shared__ __attribute__((aligned(16))) double fin[%d * %d];  // the wavefront's input rows", PPW, 3 * program.n_free);
    // own geometry: the fixed points and the derived-op parameters are the same for every problem - read once per
    // wavefront, ahead of the loop over its wave units (a persistent grid: okx_expand_positions_batch caps it)
    g.f("  const bool own = a.geom_pos == nullptr;");
    g.f("  const double* gp = a.design_pos;");
    g.f("  const double* gq = a.row_param; (void)gq;");
    g.out += ev.hoisted;
    for (int p = 0; p < NP; ++p)
      if (used[p]) g.f("  double p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    g.f("  for (long long wu = blockIdx.x; wu * %d < a.n_problems; wu += gridDim.x) {", PPW);
    g.f("    long long bb = wu * %d + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;", PPW);
    g.f("    {  // input rows: one contiguous block, 16 bytes per lane, then each lane its components from LDS");
    g.f("      const long long rem_in = a.n_problems - wu * %d;", PPW);
    g.f("      const int n_in = (int)(rem_in < %d ? rem_in : %d) * %d;", PPW, PPW, 3 * program.n_free);
    g.f("      const double* src = a.free + wu * %d * %d;", PPW, 3 * program.n_free);
    g.f("      if ((reinterpret_cast<unsigned long long>(src) & 15ull) == 0ull) {");
    g.f("        for (int i = lane; i < n_in / 2; i += 64) reinterpret_cast<double2*>(fin)[i] = reinterpret_cast<const double2*>(src)[i];");
    g.f("        if ((n_in & 1) && lane == 0) fin[n_in - 1] = src[n_in - 1];");
    g.f("      } else {");
    g.f("        for (int i = lane; i < n_in; i += 64) fin[i] = src[i];");
    g.f("      }");
    g.f("    }");
    g.f("    if (!own) {");
    g.f("      const long long geom = bb / a.steps_per_geometry;");
    g.f("      gp = a.geom_pos + geom * %d;", 3 * prog_points);
    for (int p = 0; p < NP; ++p)
      if (used[p] && ev.blk_of_point[p] < 0 && ev.dop_of_point[p] < 0) g.f("      p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    g.f("    }");
    g.f("    __syncthreads();");
    for (int F = 0; F < nf; ++F) {
      const int pt = ev.fp(F);
      g.f("    p%d = ld3(fin + (valid ? quad : 0) * %d + %s + cc, c);", pt, 3 * program.n_free,
          Gen::sel(3 * ordinal[pv->pt[0][pt]], 3 * ordinal[pv->pt[1][pt]]).c_str());
    }
    g.out += final_src;
    g.f("    if (c < 3) {");
    g.f("      double* st = stage + quad * %d + c;", 3 * prog_out);
    for (int k = 0; k < P.n_out; ++k) {
      const int k0 = pv->out[0][k], k1 = pv->out[1][k];
      if (k1 >= 0) g.f("      st[q1 ? %d : %d] = p%d;", 3 * k1, 3 * k0, P.out_point[k]);
      else g.f("      if (!q1) st[%d] = p%d;", 3 * k0, P.out_point[k]);
    }
    for (size_t k = 0; k < pv->shared_out.size(); ++k)
      g.f("      if (!q1) st[%d] = gp[%d + c];", 3 * pv->shared_out[k], 3 * pv->shared_pt[k]);
    g.f("    }");
    g.f("    __syncthreads();");
    g.f("    const long long rem = a.n_problems - wu * %d;", PPW);
    g.f("    const int n_doubles = (int)(rem < %d ? rem : %d) * %d;", PPW, PPW, 3 * prog_out);
    g.f("    double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * %d * %d);", PPW, 3 * prog_out);
    g.f("    const double2* src = reinterpret_cast<const double2*>(stage);");
    g.f("    for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
    g.f("    if ((n_doubles & 1) && lane == 0) a.out_pos[wu * %d * %d + n_doubles - 1] = stage[n_doubles - 1];", PPW, 3 * prog_out);
    g.f("    __syncthreads();");
    g.f("  }");
    g.f("}");
    g.f("");
  }
  if (!pv) {
  // ---- output positions from free coordinates (okx_expand_positions_batch): fixed points from the design table,
  //      every derived point re-evaluated, records written like the solve kernel's ----
  g.f("struct QExpandArgs { const double* free; const double* geom_pos; double* out_pos; long long n_problems, steps_per_geometry;");
  g.f("  const double* design_pos; const double* row_param; const double* dop_param; };");
  g.f("extern \"C\" __global__ void __launch_bounds__(64) okx_quad_expand(QExpandArgs a) {");
  g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
  g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
  g.f("  (void)e0; (void)e1; (void)e2;");
  if (ev.lds_constants) g.out += lds_decl;
  g.f("  __shared__ double stage[16 * %d];", 3 * P.n_out);
  g.f("  for (long long wu = blockIdx.x; wu * 16 < a.n_problems; wu += gridDim.x) {");
  g.f("    long long bb = wu * 16 + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;");
  g.f("    const long long geom = a.geom_pos != nullptr ? bb / a.steps_per_geometry : 0;");
  g.f("    const double* gp = a.geom_pos != nullptr ? a.geom_pos + geom * %d : a.design_pos;", 3 * NP);
  g.f("    const double* gq = a.row_param; (void)gq;");
  g.out += ev.hoisted;
  for (int p = 0; p < NP; ++p)
    if (used[p]) g.f("    double p%d = ld3(gp + %d + cc, c);", p, 3 * p);
  for (int F = 0; F < nf; ++F) g.f("    p%d = ld3(a.free + bb * %d + %d + cc, c);", ev.fp(F), 3 * nf, 3 * ev.perm[F]);
  g.out += final_src;
  g.f("    if (c < 3) {");
  g.f("      double* st = stage + quad * %d + c;", 3 * P.n_out);
  for (int k = 0; k < P.n_out; ++k) g.f("      st[%d] = p%d;", 3 * k, P.out_point[k]);
  g.f("    }");
  g.f("    __syncthreads();");
  g.f("    const long long rem = a.n_problems - wu * 16;");
  g.f("    const int n_doubles = (int)(rem < 16 ? rem : 16) * %d;", 3 * P.n_out);
  g.f("    double2* dst = reinterpret_cast<double2*>(a.out_pos + wu * 16 * %d);", 3 * P.n_out);
  g.f("    const double2* src = reinterpret_cast<const double2*>(stage);");
  g.f("    for (int i = lane; i < n_doubles / 2; i += 64) dst[i] = src[i];");
  g.f("    if ((n_doubles & 1) && lane == 0) a.out_pos[wu * 16 * %d + n_doubles - 1] = stage[n_doubles - 1];", 3 * P.n_out);
  g.f("    __syncthreads();");
  g.f("  }");
  g.f("}");
  g.f("");
  // ---- parity / debug kernel: r, J^T J, J^T r at given x, and the damped step for a given lambda ----
  g.f("struct QEvalArgs { const double* x; const double* targets; double* r; double* ata; double* atr; double* dx;");
  g.f("  double lambda; long long n_problems; const double* design_pos; const double* row_param; const double* dop_param; };");
  g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_eval(QEvalArgs a) {", waves_per_simd);
  g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
  g.out += atan_decl;
  g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
  if (ev.lds_constants) g.out += lds_decl;
  g.f("  const double* gp = a.design_pos; const double* gq = a.row_param;");
  g.out += ev.hoisted;
  g.f("  for (long long wu = blockIdx.x; wu * 16 < a.n_problems; wu += gridDim.x) {");
  g.f("    long long bb = wu * 16 + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;");
  for (int p = 0; p < NP; ++p)
    if (used[p]) g.f("    double p%d = c < 3 ? gp[%d + cc] : 0.0;", p, 3 * p);
  for (int F = 0; F < nf; ++F) g.f("    p%d = c < 3 ? a.x[bb * %d + %d + cc] : 0.0;", ev.fp(F), 3 * nf, 3 * ev.perm[F]);
  for (int t = 0; t < T; ++t) g.f("    const double tv%d = a.targets[bb * %d + %d];", t, T, t);
  g.out += eval_src;
  g.f("    if (valid && c == 0) {");
  for (int i = 0; i < P.m; ++i) g.f("      a.r[bb * %d + %d] = r%d;", P.m, i, i);
  g.f("    }");
  g.f("    if (valid && c < 3) {");
  for (int F = 0; F < nf; ++F) {
    g.f("      a.atr[bb * %d + %d + c] = gn%d;", 3 * nf, 3 * ev.perm[F], F);
    for (int G = 0; G <= F; ++G)
      if (ev.nz[F][G])
        for (int k = 0; k < 3; ++k)
          g.f("      a.ata[(bb * %d + %d + c) * %d + %d] = %s;", 3 * nf, 3 * ev.perm[F], 3 * nf, 3 * ev.perm[G] + k,
              Gen::A(F, G, k).c_str());
  }
  g.f("    }");
  g.f("    const double lambda = a.lambda;");
  for (int F = 0; F < nf; ++F)
    for (int G = 0; G <= F; ++G)
      if (ev.fillf[F][G]) {
        for (int k = 0; k < 3; ++k) {
          if (!(F == G && k == 2)) g.f("    double %s;", Gen::Ln(F, G, k).c_str());
          if (!ev.nz[F][G]) g.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
        }
      }
  g.out += solve_src;
  g.f("    if (valid && c < 3) {");
  for (int F = 0; F < nf; ++F)
    g.f("      a.dx[bb * %d + %d + c] = ok ? nx%d : __builtin_nan(\"\");", 3 * nf, 3 * ev.perm[F], F);
  g.f("    }");
  g.f("  }");
  g.f("}");
  g.f("");
  }
  // ---- solution-manifold tangents (reference sensitivity.py:57-143): one kernel, B solved states ----
  std::vector<int> out_index(NP, -1);
  for (int k = 0; k < P.n_out; ++k) out_index[P.out_point[k]] = k;
  bool tangent_ok = T > 0;
  for (int F = 0; F < nf; ++F) {
    const int k = out_index[ev.fp(F)];
    tangent_ok = tangent_ok && k >= 0 && (!pv || (pv->out[0][k] >= 0 && pv->out[1][k] >= 0));
  }
  if (tangent_ok) {
    ev.out.clear();
    ev.uid = 200000;
    for (int e = 0; e < P.n_derived; ++e)
      if (P.dop_active[e] < 0 && !ev.derived_op(e, false)) {
        *why = ev.why;
        return false;
      }
    std::string rest_src = ev.out;
    ev.out.clear();
    ev.emit_factor();
    std::string factor_src = ev.out;
    g.f("typedef struct { double min_pivot, max_pivot; int flags, reserved; } okx_tangent_info;");
    g.f("struct QTanArgs { const double* pos; const double* geom_pos; const double* geom_row_param; double* tan;");
    g.f("  okx_tangent_info* tinfo; long long n_problems, steps_per_geometry;");
    g.f("  const double* design_pos; const double* row_param; const double* dop_param; };");
    g.f("template <bool PG> DEV void okx_quad_tangent_body(const QTanArgs& a) {");
    if (pv)
      g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 3, q1 = (lane >> 2) & 1, cc = c < 3 ? c : 2;");
    else
      g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
    g.out += atan_decl;
    if (ev.lds_constants) g.out += lds_decl;
    g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
    g.f("  for (long long wu = blockIdx.x; wu * %d < a.n_problems; wu += gridDim.x) {", PPW);
    g.f("    long long bb = wu * %d + quad; const bool valid = bb < a.n_problems; if (!valid) bb = a.n_problems - 1;", PPW);
    g.f("    const long long geom = a.steps_per_geometry > 0 ? bb / a.steps_per_geometry : 0;");
    g.f("    const double* gp = PG ? a.geom_pos + geom * %d : a.design_pos;", 3 * prog_points);
    g.f("    const double* gq = PG ? a.geom_row_param + geom * %d : a.row_param;", 8 * prog_crows);
    g.out += ev.hoisted;
    g.out += couple_hoist;
    for (int p = 0; p < NP; ++p)
      if (used[p]) g.f("    double p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    for (int F = 0; F < nf; ++F) {
      const int k = out_index[ev.fp(F)];
      const std::string off = pv ? Gen::sel(3 * pv->out[0][k], 3 * pv->out[1][k]) : std::to_string(3 * k);
      g.f("    p%d = ld3(a.pos + bb * %d + %s + cc, c);", ev.fp(F), 3 * prog_out, off.c_str());
    }
    for (int t = 0; t < T; ++t) g.f("    const double tv%d = 0.0;  // target values do not enter the Jacobian", t);
    g.out += eval_src;
    g.out += couple_eval;
    g.out += rest_src;
    g.f("    const double lambda = 0.0;");
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G <= F; ++G)
        if (ev.fillf[F][G]) {
          for (int k = 0; k < 3; ++k) {
            if (!(F == G && k == 2)) g.f("    double %s;", Gen::Ln(F, G, k).c_str());
            if (!ev.nz[F][G]) g.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
          }
        }
    if (pv) {
      // The joining row couples the halves: A = D + w w^T with D = blockdiag(J^T J of each half), w = (w_L, w_R).
      // Undamped, D alone is singular (the partner's rack pickup slides freely along its line), so each half
      // takes its own part of the rank-one term, Dt = D + blockdiag(w_L w_L^T, w_R w_R^T), and the remaining
      // off-diagonal coupling u v^T + v u^T (u = (w_L, 0), v = (0, w_R)) goes through a 2 x 2 Woodbury system:
      // q = y - z c,  Dt y = rhs,  Dt z = w (per half),  c = (s_partner - g_partner s_own) / (1 - g_own g_partner),
      // g = w.z and s = w.y of each half.
      if (NK > 1) g.out += join_rank_one_src();
      else
      for (int k = 0; k < 3; ++k)
        g.f("    %s = fma(cu, QB%d(cu), %s);", Gen::A(FU, FU, k).c_str(), k, Gen::A(FU, FU, k).c_str());
    }
    g.out += factor_src;
    if (pv && NK > 1) {
      g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
      g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
      g.out += join_z_src();
    } else if (pv) {
      g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
      g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
      std::vector<std::string> rhs_w;
      for (int F = 0; F < nf; ++F) rhs_w.push_back(F == FU ? "cu" : "0.0");
      for (int F = 0; F < nf; ++F) g.f("    double nz%d;", F);
      ev.out.clear();
      ev.emit_substitute(rhs_w, "sz");
      g.f("    {");
      g.out += ev.out;
      for (int F = 0; F < nf; ++F) g.f("    nz%d = sz%d;", F, F);
      g.f("    }");
      g.f("    const double sm_g = qsum(cu * nz%d), sm_gp = xq(sm_g);", FU);
      g.f("    const double sm_det = 1.0 - sm_g * sm_gp;");
    }
    // one solve per PROGRAM target; in pair mode a side target stands for one program target per half
    struct Job { int t, side, prog_t; };
    std::vector<Job> jobs;
    for (int t = 0; t < T; ++t) {
      if (!pv) { jobs.push_back({t, -1, t}); continue; }
      for (int sd = 0; sd < 2; ++sd)
        if (pv->tgt[sd][t] >= 0) jobs.push_back({t, sd, pv->tgt[sd][t]});
    }
    for (const Job& job : jobs) {
      const int t = job.t;
      g.f("    {  // program target %d: (J^T J) q = J^T e_t, then the velocity of every point", job.prog_t);
      if (pv) g.f("    const double ms = q1 == %d ? 1.0 : 0.0;  // the half that carries this target", job.side);
      std::vector<std::string> rhs(nf, "0.0");
      auto it = ev.target_j.find(t);
      if (it != ev.target_j.end())
        for (auto& fv : it->second) rhs[fv.first] = pv ? "(ms * " + Gen::sx(fv.second) + ")" : Gen::sx(fv.second);
      ev.out.clear();
      ev.emit_substitute(rhs, pv ? "ty" : "tq");
      if (pv && NK > 1) {
        ev.out += join_correct_src([&](int F) { return "ty" + std::to_string(F); },
                                   [&](int F, const std::string& e) { return sfmt("    const double tq%d = %s;\n", F, e.c_str()); });
      } else if (pv) {
        ev.f("    const double sm_s = qsum(cu * ty%d);", FU);
        ev.f("    const double sm_c = (xq(sm_s) - sm_gp * sm_s) / sm_det;");
        for (int F = 0; F < nf; ++F) ev.f("    const double tq%d = fma(-nz%d, sm_c, ty%d);", F, F, F);  // (q1 is the lane's half)
      }
      const std::string vp = "w" + std::to_string(job.prog_t) + "_";
      for (int p = 0; p < NP; ++p) {
        if (!used[p] || ev.dop_of_point[p] >= 0) continue;
        if (ev.blk_of_point[p] >= 0)
          ev.f("    const double %s%d = tq%d;", vp.c_str(), p, ev.blk_of_point[p]);
        else
          ev.f("    const double %s%d = 0.0;", vp.c_str(), p);
      }
      for (int e = 0; e < P.n_derived; ++e)
        if (!ev.derived_jvp(e, vp)) {
          *why = ev.why;
          return false;
        }
      g.out += ev.out;
      g.f("    if (valid && c < 3) {");
      g.f("      double* o = a.tan + (bb * %d + %d) * %d + c;", prog_targets, job.prog_t, 3 * prog_out);
      for (int k = 0; k < P.n_out; ++k) {
        if (!pv) {
          g.f("      o[%d] = ok ? %s%d : __builtin_nan(\"\");", 3 * k, vp.c_str(), P.out_point[k]);
          continue;
        }
        const int k0 = pv->out[0][k], k1 = pv->out[1][k];
        if (k1 >= 0)
          g.f("      o[%s] = ok ? %s%d : __builtin_nan(\"\");", Gen::sel(3 * k0, 3 * k1).c_str(), vp.c_str(), P.out_point[k]);
        else
          g.f("      if (!q1) o[%d] = ok ? %s%d : __builtin_nan(\"\");", 3 * k0, vp.c_str(), P.out_point[k]);
      }
      if (pv)
        for (size_t k = 0; k < pv->shared_out.size(); ++k)
          g.f("      if (!q1) o[%d] = ok ? 0.0 : __builtin_nan(\"\");  // chassis point of neither half", 3 * pv->shared_out[k]);
      g.f("    }");
      g.f("    }");
    }
    g.f("    if (valid && c == 0%s) {", pv ? " && !q1" : "");
    g.f("      okx_tangent_info ti; ti.min_pivot = pmin; ti.max_pivot = pmax; ti.reserved = 0;");
    g.f("      ti.flags = (ok ? 1 : 0) | ((!ok || pmin <= %d * 2.220446049250313e-16 * pmax) ? 2 : 0);", 3 * nf * (pv ? 2 : 1));
    g.f("      a.tinfo[bb] = ti;");
    g.f("    }");
    g.f("  }");
    g.f("}");
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_tangent_u(QTanArgs a) { okx_quad_tangent_body<false>(a); }",
        waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_tangent_g(QTanArgs a) { okx_quad_tangent_body<true>(a); }",
        waves_per_simd);
    g.f("");
  }
  if (head_ok) {
    // ---- first-step table: one quad (pair mode: one quad pair) per geometry evaluates the design state once (rows,
    //      J^T J, damped LDL^T) and substitutes once per column; what every chain head of that geometry starts from ----
    ev.out.clear();
    ev.uid = 400000;
    ev.reset_caches();
    ev.emit_factor();
    const std::string factor_src = ev.out;
    g.f("struct QHeadArgs { const double* geom_pos; const double* geom_row_param; double* head; long long n_geometries; double lambda0;");
    g.f("  const double* design_pos; const double* row_param; const double* dop_param; };");
    g.f("template <bool PG> DEV void okx_quad_head_body(const QHeadArgs& a) {");
    if (pv)
      g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 3, q1 = (lane >> 2) & 1, cc = c < 3 ? c : 2;");
    else
      g.f("  const int lane = threadIdx.x, c = lane & 3, quad = lane >> 2, cc = c < 3 ? c : 2;");
    g.out += atan_decl;
    g.f("  const double e0 = c == 0 ? 1.0 : 0.0, e1 = c == 1 ? 1.0 : 0.0, e2 = c == 2 ? 1.0 : 0.0;");
    if (ev.lds_constants) g.out += lds_decl;
    g.f("  for (long long wu = blockIdx.x; wu * %d < a.n_geometries; wu += gridDim.x) {", PPW);
    g.f("    long long geom = wu * %d + quad; const bool valid = geom < a.n_geometries; if (!valid) geom = a.n_geometries - 1;", PPW);
    g.f("    const double* gp = PG ? a.geom_pos + geom * %d : a.design_pos;", 3 * prog_points);
    g.f("    const double* gq = PG ? a.geom_row_param + geom * %d : a.row_param;", 8 * prog_crows);
    g.out += ev.hoisted;
    g.out += couple_hoist;
    for (int p = 0; p < NP; ++p)
      if (used[p]) g.f("    double p%d = ld3(gp + %s + cc, c);", p, ev.point3(p).c_str());
    // targets at their design values: the target rows vanish, ss / mres_new are the constraint rows' alone
    {
      ev.reset_caches();
      ev.out.clear();
      for (int i = P.n_crows; i < P.m; ++i) {
        const int t = ev.target_of_row(i);
        const std::string d = ev.dot(Gen::pn(P.row_pts[i][0]), ev.rpv(i, 0));
        ev.f("    const double tv%d = %s;", t, d.c_str());
      }
      g.out += ev.out;
      ev.out.clear();
      ev.reset_caches();
    }
    g.out += eval_src;
    g.out += couple_eval;
    const int n_fd_dirs = NPAIR > 0 ? (HK - 1) + (HK - 1) * (HK - 2) / 2 : 0;
    const int fd_rows = P.m + NK;  // the joining rows follow each half's own rows
    auto rcn = [&](int j) { return NK > 1 ? "rc" + std::to_string(j) : std::string("rc"); };
    auto cun = [&](int j) { return NK > 1 ? "cu" + std::to_string(j) : std::string("cu"); };
    if (NPAIR > 0) {
      // second differences of the rows per direction, [direction][row][quad] in LDS (63 quad-uniform doubles would
      // otherwise sit in registers beside the factor); every one starts at -2 r(design state), while the r_i are at hand
      // (pair mode: the joining row is row P.m of each half's list - both halves hold the same value)
      g.f("    __shared__ double hDl[%d];", n_fd_dirs * fd_rows * 16);
      g.f("    const int hdq = lane >> 2;");
      for (int d = 0; d < n_fd_dirs; ++d) {
        for (int i = 0; i < P.m; ++i) g.f("    hDl[%d + hdq] = -2.0 * r%d;", (d * fd_rows + i) * 16, i);
        for (int j = 0; j < NK; ++j) g.f("    hDl[%d + hdq] = -2.0 * %s;", (d * fd_rows + P.m + j) * 16, rcn(j).c_str());
      }
    }
    g.f("    double diag = 0.0;");
    for (int F = 0; F < nf; ++F)
      g.f("    diag = fmax(diag, c == 0 ? %s : (c == 1 ? %s : (c == 2 ? %s : 0.0)));", Gen::A(F, F, 0).c_str(),
          Gen::A(F, F, 1).c_str(), Gen::A(F, F, 2).c_str());
    g.f("    diag = PMAX(diag);");
    g.f("    const double lambda = a.lambda0 * diag;");
    for (int F = 0; F < nf; ++F)
      for (int G = 0; G <= F; ++G)
        if (ev.fillf[F][G]) {
          for (int k = 0; k < 3; ++k) {
            if (!(F == G && k == 2)) g.f("    double %s;", Gen::Ln(F, G, k).c_str());
            if (!ev.nz[F][G]) g.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
          }
        }
    if (pv && NK > 1) g.out += join_rank_one_src();
    else
    if (pv)  // each half takes its own part of the joining row's rank-one term (see the solve kernel)
      for (int k = 0; k < 3; ++k)
        g.f("    %s = fma(cu, QB%d(cu), %s);", Gen::A(FU, FU, k).c_str(), k, Gen::A(FU, FU, k).c_str());
    g.out += factor_src;
    if (pv && NK > 1) {
      g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
      g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
      g.out += join_z_src();
    } else if (pv) {
      g.f("    ok = ok && xq(ok ? 1.0 : 0.0) > 0.5;  // both halves must factor");
      g.f("    pmin = fmin(pmin, xq(pmin)); pmax = fmax(pmax, xq(pmax));");
      std::vector<std::string> rhs_w;
      for (int F = 0; F < nf; ++F) rhs_w.push_back(F == FU ? "cu" : "0.0");
      for (int F = 0; F < nf; ++F) g.f("    double nz%d;", F);
      ev.out.clear();
      ev.emit_substitute(rhs_w, "sz");
      g.f("    {");
      g.out += ev.out;
      for (int F = 0; F < nf; ++F) g.f("    nz%d = sz%d;", F, F);
      g.f("    }");
      g.f("    const double sm_g = qsum(cu * nz%d), sm_gp = xq(sm_g);", FU);
      g.f("    const double sm_det = 1.0 - sm_g * sm_gp;");
    }
    g.f("    double* ho = a.head + geom * %d%s;", head_stride, pv ? (" + (q1 ? " + std::to_string(head_side) + " : 0)").c_str() : "");
    g.f("    double* hs = a.head + geom * %d;  // Gram matrices and scalars (written once per geometry)", head_stride);
    std::vector<std::vector<std::string>> rhs_of(HK, std::vector<std::string>(nf, "0.0"));
    for (int F = 0; F < nf; ++F) rhs_of[0][F] = "gn" + std::to_string(F);  // constraint rows' gradient (target rows vanish here)
    for (int k = 0; k < HK; ++k) {
      const HeadCol& col = head_cols[k];
      if (k > 0) {
        auto it = ev.target_j.find(col.t);
        if (it != ev.target_j.end())
          for (auto& fv : it->second) rhs_of[k][fv.first] = pv ? "(hm" + std::to_string(k) + " * " + Gen::sx(fv.second) + ")" : Gen::sx(fv.second);
        if (pv) g.f("    const double hm%d = q1 == %d ? 1.0 : 0.0;  // the half that carries program target %d", k, col.side, col.prog_t);
      }
      ev.out.clear();
      const std::string outn = "hq" + std::to_string(k) + "_";
      ev.emit_substitute(rhs_of[k], outn.c_str());
      g.f("    // column %d: Q_k = (J^T J + lambda I)^-1 G_k", k);
      // emit_substitute declares y{F} afresh: one scope per column, results copied out
      for (int F = 0; F < nf; ++F) g.f("    double hQ%d_%d;", k, F);
      g.f("    {");
      g.out += ev.out;
      if (pv && NK > 1) {
        g.out += join_correct_src([&](int F) { return outn + std::to_string(F); },
                                  [&](int F, const std::string& e) { return sfmt("    hQ%d_%d = %s;\n", k, F, e.c_str()); });
      } else if (pv) {  // the coupling between the halves: 2 x 2 Woodbury, as in the solve kernel
        g.f("    const double sm_s = qsum(cu * %s%d);", outn.c_str(), FU);
        g.f("    const double sm_c = (xq(sm_s) - sm_gp * sm_s) / sm_det;");
        for (int F = 0; F < nf; ++F) g.f("    hQ%d_%d = fma(-nz%d, sm_c, %s%d);", k, F, F, outn.c_str(), F);
      } else {
        for (int F = 0; F < nf; ++F) g.f("    hQ%d_%d = %s%d;", k, F, outn.c_str(), F);
      }
      g.f("    }");
      g.f("    if (valid) {");
      for (int F = 0; F < nf; ++F) g.f("      ho[%d + c] = c < 3 ? hQ%d_%d : 0.0;", 4 * (k * nf + F), k, F);
      g.f("    }");
    }
    for (int j = 0; j < HK; ++j)
      for (int k = 0; k < HK; ++k) {
        std::string em, en;
        for (int F = 0; F < nf; ++F) {
          if (rhs_of[k][F] != "0.0") em += (em.empty() ? "" : " + ") + ("hQ" + std::to_string(j) + "_" + std::to_string(F)) + " * (" + rhs_of[k][F] + ")";
          en += (en.empty() ? "" : " + ") + ("hQ" + std::to_string(j) + "_" + std::to_string(F)) + " * hQ" + std::to_string(k) + "_" + std::to_string(F);
        }
        if (em.empty()) em = "0.0";
        g.f("    { const double vm = PSUM(%s), vn = PSUM(%s);", em.c_str(), en.c_str());
        g.f("      if (valid && c == 0%s) { hs[%d] = vm; hs[%d] = vn; } }", pv ? " && !q1" : "", head_off - 2 * HK * HK + j * HK + k,
            head_off - HK * HK + j * HK + k);
      }
    if (NPAIR > 0) {
      // ---- second-order terms of the first step (geodesic acceleration, Transtrum & Sethna 2012): the first step
      // d1 = -sum_k w_k Q_k is the Gauss-Newton step of the LINEARISED rows; the rows' curvature along it,
      // r''(d1, d1) = sum_st w_s w_t r''(Q_s, Q_t), gives the correction d2 = -1/2 (J^T J + lambda I)^-1 J^T r''(d1, d1),
      // which is a quadratic form in the weights with per-geometry coefficient vectors S_st = M^-1 J^T r''(Q_s, Q_t).
      // r'' by central second differences of the residual vector along Q_s, Q_t and Q_s + Q_t (the constraint column's
      // Q_0 is ~1e-6 mm: its second-order share is far below the solve's tolerance and is left out); then one more pass
      // over the rows' gradients at the design state for J^T r'' and one substitution per pair.
      const double kFdStep = 0.25;  // displacement of the finite differences in units of Q (mm per mm of target residual)
      for (int F = 0; F < nf; ++F) g.f("    const double hx0_%d = p%d;", F, ev.fp(F));
      // directions: 0 .. T-1 the target columns, then the sums of two
      struct Dir { int s, t; };
      std::vector<Dir> dirs;
      for (int s2 = 1; s2 < HK; ++s2) dirs.push_back({s2, -1});
      for (int s2 = 1; s2 < HK; ++s2)
        for (int t2 = s2 + 1; t2 < HK; ++t2) dirs.push_back({s2, t2});
      int uid_base = 600000;
      for (size_t d = 0; d < dirs.size(); ++d) {
        for (int sign = 0; sign < 2; ++sign) {
          Gen fd(P, pv);
          fd.uid = uid_base;
          uid_base += 20000;
          fd.hoisted_names = ev.hoisted_names;
          fd.lds_constants = ev.lds_constants;
          fd.scalars_in_regs = ev.scalars_in_regs;
          fd.lanes_in_regs = ev.lanes_in_regs;
          for (int idx = 0; idx < P.n_active; ++idx) (void)fd.derived_op(P.active_op[idx], false);
          (void)fd.emit_rows_residual_only();
          g.f("    {");
          for (int F = 0; F < nf; ++F) {
            if (dirs[d].t < 0)
              g.f("    p%d = fma(%s, hQ%d_%d, hx0_%d);", ev.fp(F), sign ? "-0.25" : "0.25", dirs[d].s, F, F);
            else
              g.f("    p%d = fma(%s, hQ%d_%d + hQ%d_%d, hx0_%d);", ev.fp(F), sign ? "-0.25" : "0.25", dirs[d].s, F, dirs[d].t, F, F);
          }
          g.out += fd.out;
          for (int i = 0; i < P.m; ++i) g.f("    hDl[%d + hdq] += r%d;", (int)(d * fd_rows + i) * 16, i);
          if (pv) {  // the joining row at the displaced halves (each half moved its own joined point)
            g.out += couple_light;
            for (int j = 0; j < NK; ++j) g.f("    hDl[%d + hdq] += %s;", (int)(d * fd_rows + P.m + j) * 16, rcn(j).c_str());
          }
          g.f("    }");
        }
      }
      (void)kFdStep;
      // r''(Q_s, Q_t) per pair = the right-hand sides hR{pair}_{row}, formed from the LDS accumulators where the row is
      auto dir_of = [&](int s2, int t2) {
        for (size_t d = 0; d < dirs.size(); ++d)
          if (dirs[d].s == s2 && dirs[d].t == t2) return (int)d;
        return -1;
      };
      auto hd = [&](int d, int i) { return "hDl[" + std::to_string((d * fd_rows + i) * 16) + " + hdq]"; };
      auto pair_value = [&](int pi, int i) {
        const int s2 = head_pairs[pi].first, t2 = head_pairs[pi].second;
        char buf[256];
        if (s2 == t2)
          std::snprintf(buf, sizeof(buf), "%s * %.17g", hd(dir_of(s2, -1), i).c_str(), 1.0 / (0.25 * 0.25));
        else
          std::snprintf(buf, sizeof(buf), "(%s - %s - %s) * %.17g", hd(dir_of(s2, t2), i).c_str(), hd(dir_of(s2, -1), i).c_str(),
                        hd(dir_of(t2, -1), i).c_str(), 0.5 / (0.25 * 0.25));
        return std::string(buf);
      };
      // J^T r'' at the design state needs the rows' gradients once more - and the substitutions the factor.  Keeping the
      // factor alive across the six residual evaluations above costs more registers than the file has (564 B of scratch);
      // instead the design state is evaluated and factored a second time here (one pass of ~2 k instructions, once per
      // geometry), in a scope of its own, with J^T r'' accumulated beside J^T r.
      g.f("    {");
      for (int F = 0; F < nf; ++F) g.f("    p%d = hx0_%d;", ev.fp(F), F);
      {
        Gen jt(P, pv);
        jt.uid = uid_base;
        jt.hoisted_names = ev.hoisted_names;
        jt.lds_constants = ev.lds_constants;
        jt.scalars_in_regs = ev.scalars_in_regs;
        jt.lanes_in_regs = ev.lanes_in_regs;
        jt.pin_ata = ev.pin_ata;
        jt.pin_atr = ev.pin_atr;
        for (int pi = 0; pi < NPAIR; ++pi) jt.jtv_rhs.push_back("hR" + std::to_string(pi) + "_");
        jt.jtv_value = pair_value;
        jt.jtv_only = false;
        for (int idx = 0; idx < P.n_active; ++idx) (void)jt.derived_op(P.active_op[idx], true);
        (void)jt.emit_rows();
        g.out += jt.out;
        if (pv) {  // the joining row: its gradient lives in the joined point's block, its curvature term with it
          g.out += couple_eval;
          for (int j = 0; j < NK; ++j)
            for (int pi = 0; pi < NPAIR; ++pi)
              g.f("    hR%d_g%d = fma(%s, %s, hR%d_g%d);", pi, FUj[j], cun(j).c_str(), pair_value(pi, P.m + j).c_str(), pi, FUj[j]);
        }
        for (int F = 0; F < nf; ++F)
          for (int G = 0; G <= F; ++G)
            if (ev.fillf[F][G]) {
              for (int k = 0; k < 3; ++k) {
                if (!(F == G && k == 2)) g.f("    double %s;", Gen::Ln(F, G, k).c_str());
                if (!ev.nz[F][G]) g.f("    double %s = 0.0;", Gen::A(F, G, k).c_str());
              }
            }
        if (pv && NK > 1) g.out += join_rank_one_src();
        else
        if (pv)
          for (int k = 0; k < 3; ++k)
            g.f("    %s = fma(cu, QB%d(cu), %s);", Gen::A(FU, FU, k).c_str(), k, Gen::A(FU, FU, k).c_str());
        g.out += factor_src;
        if (pv && NK > 1) g.out += join_z_src();
        else
        if (pv) {  // D~ z = w once more (the first scope's z is not kept alive across the residual passes above)
          std::vector<std::string> rhs_w;
          for (int F = 0; F < nf; ++F) rhs_w.push_back(F == FU ? "cu" : "0.0");
          for (int F = 0; F < nf; ++F) g.f("    double nz%d;", F);
          ev.out.clear();
          ev.emit_substitute(rhs_w, "sz");
          g.f("    {");
          g.out += ev.out;
          for (int F = 0; F < nf; ++F) g.f("    nz%d = sz%d;", F, F);
          g.f("    }");
          g.f("    const double sm_g = qsum(cu * nz%d), sm_gp = xq(sm_g);", FU);
          g.f("    const double sm_det = 1.0 - sm_g * sm_gp;");
        }
        for (int pi = 0; pi < NPAIR; ++pi) {
          std::vector<std::string> rhs;
          for (int F = 0; F < nf; ++F) rhs.push_back("hR" + std::to_string(pi) + "_g" + std::to_string(F));
          ev.out.clear();
          const std::string outn = "hS" + std::to_string(pi) + "_";
          ev.emit_substitute(rhs, outn.c_str());
          g.f("    {");
          g.out += ev.out;
          if (pv && NK > 1) {
            g.f("    double* hso = hs + (q1 ? %d : 0);  // this half's S block", head_s_side);
            g.out += join_correct_src([&](int F) { return outn + std::to_string(F); }, [&](int F, const std::string& e) {
              return sfmt("    if (valid) hso[%d + c] = c < 3 ? %s : 0.0;\n", head_s_off + 4 * (pi * nf + F), e.c_str());
            });
          } else if (pv) {  // the coupling between the halves, as for the columns
            g.f("    const double sm_s = qsum(cu * %s%d);", outn.c_str(), FU);
            g.f("    const double sm_c = (xq(sm_s) - sm_gp * sm_s) / sm_det;");
            g.f("    double* hso = hs + (q1 ? %d : 0);  // this half's S block", head_s_side);
            g.f("    if (valid) {");
            for (int F = 0; F < nf; ++F)
              g.f("      hso[%d + c] = c < 3 ? fma(-nz%d, sm_c, %s%d) : 0.0;", head_s_off + 4 * (pi * nf + F), F, outn.c_str(), F);
            g.f("    }");
          } else {
          g.f("    if (valid) {");
          for (int F = 0; F < nf; ++F) g.f("      ho[%d + c] = c < 3 ? %s%d : 0.0;", head_s_off + 4 * (pi * nf + F), outn.c_str(), F);
          g.f("    }");
          }
          g.f("    }");
        }
      }
      g.f("    }");
    }
    g.f("    if (valid && c == 0%s) {", pv ? " && !q1" : "");
    g.f("      hs[%d] = diag; hs[%d] = pmin; hs[%d] = ss; hs[%d] = mres_new; hs[%d] = ok ? 1.0 : 0.0; hs[%d] = pmax; hs[%d] = %d.0; hs[%d] = 0.0;",
        head_off, head_off + 1, head_off + 2, head_off + 3, head_off + 4, head_off + 5, head_off + 6, NPAIR, head_off + 7);
    g.f("    }");
    g.f("  }");
    g.f("}");
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_head_u(QHeadArgs a) { okx_quad_head_body<false>(a); }", waves_per_simd);
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_head_g(QHeadArgs a) { okx_quad_head_body<true>(a); }", waves_per_simd);
    g.f("");
  }
  g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_solve_u(QArgs a) { okx_quad_body<false>(a); }",
      waves_per_simd);
  g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_solve_g(QArgs a) { okx_quad_body<true>(a); }",
      waves_per_simd);
  if (cold_body) {
    g.f("extern \"C\" __global__ void __launch_bounds__(64, %d) okx_quad_cold_u(QArgs a) { okx_quad_cold_body(a); }", waves_per_simd);
  }
  *src = g.out;
  return true;
}

}  // namespace okx
