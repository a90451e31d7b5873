// okx_pairview.cpp — recognises constraint programs made of TWO STRUCTURALLY IDENTICAL HALVES joined
// by one distance row (the reference's composed axle: two corners, side-qualified by
// suspensions/axle/suspension.py:146-211, joined by the rack length row :196-209) and splits off
// the half ("side") program the quad generator is run on.
//
// Why: J^T J of such a program is blockdiag(A_left, A_right) plus the rank-one term of the joining
// row, so the damped normal equations are solved by two independent, identical block LDL^T
// factorisations and a low-rank (2 x 2 Woodbury) correction.  One quad per side runs the same generated
// instruction stream on its own half of the data; 8 problems per wavefront.
//
// The match is positional, which is how the reference emits such programs: free points sorted by
// (Side, PointID) = first half / second half, derived ops and rows of the two sides in the same
// relative order.  Anything that does not match exactly keeps the generic interpreter kernels.
#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "okx_plan.hpp"
#include "okx_quad.hpp"

namespace okx {

namespace {

int row_slots(int type) {
  switch (type) {
    case OKX_ROW_DISTANCE: case OKX_ROW_SPHERICAL: case OKX_ROW_MIDPOINT_ON_PLANE: return 2;
    case OKX_ROW_THREE_POINT_ANGLE: return 3;
    case OKX_ROW_ANGLE: case OKX_ROW_VECTORS_PARALLEL: case OKX_ROW_VECTORS_PERPENDICULAR:
    case OKX_ROW_EQUAL_DISTANCE: case OKX_ROW_COPLANAR: case OKX_ROW_SCALAR_TRIPLE: return 4;
    default: return 1;
  }
}

}  // namespace

bool build_pair_view(const DevProgram& P, PairView* pv, std::string* why) {
  const int nf = P.n_free, D = P.n_derived, NP = P.n_points, Mc = P.n_crows, T = P.n_targets;
  if (nf < 4 || nf % 2 || D % 2) {
    *why = "not two equal halves";
    return false;
  }
  const int h = nf / 2, hd = D / 2;
  std::vector<int> side(NP, -1), mirror(NP, -1);
  std::vector<bool> is_free(NP, false);
  for (int k = 0; k < h; ++k) {
    const int a = P.free_point[k], b = P.free_point[k + h];
    side[a] = 0, side[b] = 1, mirror[a] = b, mirror[b] = a;
    is_free[a] = is_free[b] = true;
  }
  for (int e = 0; e < hd; ++e) {
    if (P.dop_type[e] != P.dop_type[e + hd]) {
      *why = "derived ops of the two halves differ";
      return false;
    }
    const int a = P.dop_out[e], b = P.dop_out[e + hd];
    side[a] = 0, side[b] = 1, mirror[a] = b, mirror[b] = a;
  }
  std::map<int, int> fmap;  // fixed point of side 0 -> its counterpart on side 1 (itself when shared)
  auto match = [&](int p0, int p1) -> bool {
    if (p0 < 0 || p1 < 0) return p0 < 0 && p1 < 0;
    if (side[p0] == 0) return mirror[p0] == p1;
    if (side[p0] == 1 || side[p1] != -1) return false;
    auto it = fmap.find(p0);
    if (it == fmap.end()) {
      fmap[p0] = p1;
      return true;
    }
    return it->second == p1;
  };
  for (int e = 0; e < hd; ++e)
    for (int s = 0; s < 4; ++s)
      if (!match(P.dop_pts[e][s], P.dop_pts[e + hd][s])) {
        *why = "derived-op inputs of the two halves differ";
        return false;
      }
  std::vector<int> r0, r1, cross;
  for (int i = 0; i < Mc; ++i) {
    bool s0 = false, s1 = false;
    for (int s = 0; s < row_slots(P.row_type[i]); ++s) {
      const int p = P.row_pts[i][s];
      if (p >= 0 && side[p] == 0) s0 = true;
      if (p >= 0 && side[p] == 1) s1 = true;
    }
    if (s0 && s1) cross.push_back(i);
    else if (s0) r0.push_back(i);
    else if (s1) r1.push_back(i);
    else {
      *why = "a row touches fixed points only";
      return false;
    }
  }
  if (r0.size() != r1.size() || cross.empty() || cross.size() > (size_t)kQuadMaxJoins) {
    *why = "need equal halves joined by one to " + std::to_string(kQuadMaxJoins) + " rows";
    return false;
  }
  for (size_t j = 0; j < r0.size(); ++j) {
    const int a = r0[j], b = r1[j];
    bool ok = P.row_type[a] == P.row_type[b];
    for (int s = 0; ok && s < 4; ++s) ok = match(P.row_pts[a][s], P.row_pts[b][s]);
    if (ok && P.row_type[a] == OKX_ROW_LINE_PIN) ok = P.row_param[a][6] == P.row_param[b][6];
    if (ok && P.row_type[a] == OKX_ROW_FIXED_AXIS) ok = P.row_param[a][0] == P.row_param[b][0];
    if (!ok) {
      *why = "rows of the two halves differ";
      return false;
    }
  }
  pv->joins.clear();
  for (int ci : cross) {
    const int a = P.row_pts[ci][0], b = P.row_pts[ci][1];
    const int type = P.row_type[ci];
    if ((type != OKX_ROW_DISTANCE && type != OKX_ROW_MIDPOINT_ON_PLANE) || a < 0 || b < 0 || mirror[a] != b || !is_free[a]) {
      *why = "a joining row is not a distance between mirrored free points (or their midpoint on a plane)";
      return false;
    }
    if (type == OKX_ROW_MIDPOINT_ON_PLANE && side[a] != 0) {
      *why = "a joining midpoint row lists the second half's point first";
      return false;
    }
    pv->joins.push_back({ci, side[a] == 0 ? a : b, type});  // program point index for now
  }
  if (pv->joins[0].type != OKX_ROW_DISTANCE) {
    *why = "the first joining row is not a distance";
    return false;
  }
  pv->couple_row = pv->joins[0].row;
  pv->couple_point = pv->joins[0].point;
  // side-0 point set -> V numbering (ascending program index)
  std::set<int> vset;
  for (int p = 0; p < NP; ++p)
    if (side[p] == 0) vset.insert(p);
  for (auto& kv : fmap) vset.insert(kv.first);
  std::vector<int> vpts(vset.begin(), vset.end());
  std::map<int, int> vidx;
  for (size_t k = 0; k < vpts.size(); ++k) vidx[vpts[k]] = (int)k;
  pv->pt[0] = vpts;
  pv->pt[1].clear();
  for (int p : vpts) pv->pt[1].push_back(side[p] == 0 ? mirror[p] : fmap[p]);
  pv->row[0] = r0;
  pv->row[1] = r1;
  pv->dop[0].clear(), pv->dop[1].clear();
  for (int e = 0; e < hd; ++e) pv->dop[0].push_back(e), pv->dop[1].push_back(e + hd);
  // targets: union over the two sides, expressed on side-0 points
  std::vector<int> tpoint(T);
  for (int t = 0; t < T; ++t) {
    tpoint[t] = P.row_pts[Mc + t][0];
    if (side[tpoint[t]] < 0) {
      *why = "a target sits on a fixed point";
      return false;
    }
  }
  pv->tgt[0].clear(), pv->tgt[1].clear();
  std::vector<int> vt_point;
  std::vector<bool> taken(T, false);
  for (int t = 0; t < T; ++t) {
    if (side[tpoint[t]] != 0) continue;
    int partner = -1;
    for (int u = 0; u < T; ++u)
      if (!taken[u] && side[tpoint[u]] == 1 && mirror[tpoint[t]] == tpoint[u] &&
          std::memcmp(P.row_param[Mc + t], P.row_param[Mc + u], 3 * sizeof(double)) == 0)
        partner = u;
    if (partner >= 0) taken[partner] = true;
    pv->tgt[0].push_back(t), pv->tgt[1].push_back(partner);
    vt_point.push_back(tpoint[t]);
  }
  for (int u = 0; u < T; ++u)
    if (side[tpoint[u]] == 1 && !taken[u]) {
      pv->tgt[0].push_back(-1), pv->tgt[1].push_back(u);
      vt_point.push_back(mirror[tpoint[u]]);
    }
  // outputs
  std::vector<int> out_index(NP, -1);
  for (int k = 0; k < P.n_out; ++k) out_index[P.out_point[k]] = k;
  std::vector<int> v_out;
  std::vector<bool> covered(P.n_out, false);
  pv->out[0].clear(), pv->out[1].clear();
  for (int k = 0; k < P.n_out; ++k) {
    const int p = P.out_point[k];
    if (!vidx.count(p)) continue;
    const int m = pv->pt[1][vidx[p]];
    const int k1 = m != p ? out_index[m] : -1;
    v_out.push_back(vidx[p]);
    pv->out[0].push_back(k), pv->out[1].push_back(k1);
    covered[k] = true;
    if (k1 >= 0) covered[k1] = true;
  }
  pv->shared_out.clear(), pv->shared_pt.clear();
  for (int k = 0; k < P.n_out; ++k)
    if (!covered[k]) {
      if (side[P.out_point[k]] >= 0) {
        *why = "an output point of side 1 has no counterpart among side 0's outputs";
        return false;
      }
      pv->shared_out.push_back(k), pv->shared_pt.push_back(P.out_point[k]);
    }
  // the side program as a descriptor -> DevProgram (plans, activity, validation)
  const int PV = (int)vpts.size(), MV = (int)r0.size(), TV = (int)pv->tgt[0].size();
  std::vector<int32_t> free_point, dop_type, dop_out, dop_pts, row_type, row_pts, tgt_point, out_point;
  std::vector<double> dop_param, row_param, tgt_dir, design_pos;
  for (int k = 0; k < h; ++k) free_point.push_back(vidx[P.free_point[k]]);
  for (int e = 0; e < hd; ++e) {
    dop_type.push_back(P.dop_type[e]);
    dop_out.push_back(vidx[P.dop_out[e]]);
    for (int s = 0; s < 4; ++s) dop_pts.push_back(P.dop_pts[e][s] >= 0 ? vidx[P.dop_pts[e][s]] : -1);
    dop_param.push_back(P.dop_param[e]);
  }
  for (int i : r0) {
    row_type.push_back(P.row_type[i]);
    for (int s = 0; s < 4; ++s) row_pts.push_back(P.row_pts[i][s] >= 0 ? vidx[P.row_pts[i][s]] : -1);
    for (int k = 0; k < OKX_ROW_PARAMS; ++k) row_param.push_back(P.row_param[i][k]);
  }
  for (int t = 0; t < TV; ++t) {
    tgt_point.push_back(vidx[vt_point[t]]);
    const int src = pv->tgt[0][t] >= 0 ? pv->tgt[0][t] : pv->tgt[1][t];
    for (int k = 0; k < 3; ++k) tgt_dir.push_back(P.row_param[Mc + src][k]);
  }
  for (int v : v_out) out_point.push_back(v);
  for (int p : vpts)
    for (int k = 0; k < 3; ++k) design_pos.push_back(P.design_pos[p][k]);
  const int32_t none32 = 0;
  const double none64 = 0.0;
  okx_program_desc d;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = OKX_ABI_VERSION;
  d.n_points = PV, d.n_free = h, d.n_derived = hd, d.n_rows = MV, d.n_targets = TV, d.n_out = (int)out_point.size();
  d.free_point = free_point.data();
  d.dop_type = hd ? dop_type.data() : &none32;
  d.dop_out = hd ? dop_out.data() : &none32;
  d.dop_pts = hd ? dop_pts.data() : &none32;
  d.dop_param = hd ? dop_param.data() : &none64;
  d.row_type = row_type.data();
  d.row_pts = row_pts.data();
  d.row_param = row_param.data();
  d.tgt_point = TV ? tgt_point.data() : &none32;
  d.tgt_dir = TV ? tgt_dir.data() : &none64;
  d.out_point = out_point.empty() ? &none32 : out_point.data();
  d.design_pos = design_pos.data();
  char err[256] = "";
  if (build_dev_program(&d, &pv->side, err, (int)sizeof(err)) != OKX_OK) {
    *why = std::string("side program: ") + err;
    return false;
  }
  pv->couple_point = vidx[pv->couple_point];
  for (auto& join : pv->joins) join.point = vidx[join.point];
  pv->n_prog_points = NP, pv->n_prog_crows = Mc, pv->n_prog_targets = T, pv->n_prog_out = P.n_out;
  return true;
}

}  // namespace okx
