// CPU driver for the camber-shim setup solve: runs okx_shim.hip's host + device function `solve_one` on
// the host, so the exact kernel source can be checked (and run under ASan / UBSan) without a GPU.
// TEST INFRASTRUCTURE ONLY — the library never takes this path.
//
//   shim_host <in.bin> <out.bin>
//   in : okx_shim_roles | int64 G | int32 P | int32 pad | double points[G][P][3] | double shim[G][11]
//   out: double points[G][P][3] | okx_shim_info info[G]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../open_kinematics_amd/csrc/okx_shim.hip"

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 3;
  okx_shim_roles roles;
  long long n = 0;
  int n_points = 0, pad = 0;
  bool ok = fread(&roles, sizeof roles, 1, in) == 1 && fread(&n, sizeof n, 1, in) == 1 &&
            fread(&n_points, sizeof n_points, 1, in) == 1 && fread(&pad, sizeof pad, 1, in) == 1;
  if (!ok || n < 0 || n_points <= 0) return 4;
  std::vector<double> points((size_t)n * n_points * 3), shim((size_t)n * OKX_SHIM_PARAMS);
  std::vector<okx_shim_info> info((size_t)n);
  ok = fread(points.data(), sizeof(double), points.size(), in) == points.size() &&
       fread(shim.data(), sizeof(double), shim.size(), in) == shim.size();
  fclose(in);
  if (!ok) return 5;
  okx::shim::ShimArgs a;
  a.roles = roles;
  a.points = points.data();
  a.shim = shim.data();
  a.info = info.data();
  a.n_geometries = n;
  a.n_points = n_points;
  for (long long g = 0; g < n; ++g) okx::shim::solve_one(a, g);
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 6;
  fwrite(points.data(), sizeof(double), points.size(), out);
  fwrite(info.data(), sizeof(okx_shim_info), info.size(), out);
  fclose(out);
  return 0;
}
