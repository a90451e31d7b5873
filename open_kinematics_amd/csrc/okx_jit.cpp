// okx_jit.cpp — hiprtc compilation and on-disk caching of the generated quad kernels.
//
// hiprtc needs no device (it is comgr + the bundled device headers), so the cache can be filled
// by `__graft_entry__.build()` on a machine without a GPU; a miss at run time compiles in place.
// Cache key = FNV-1a of (source text, compile options, hiprtc version).
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "okx_quad.hpp"

namespace okx {
bool dev_switch(const char* name) {
  const char* env = getenv("OKX_DEV");
  if (!env || !name) return false;
  const size_t n = std::strlen(name);
  for (const char* at = env; *at;) {
    const char* end = std::strchr(at, ',');
    const size_t len = end ? (size_t)(end - at) : std::strlen(at);
    if (len >= n && std::strncmp(at, name, n) == 0 && (len == n || at[n] == '=')) {
      // `name` or `name=1` switch it on, `name=0` leaves it off
      return !(len == n + 2 && at[n + 1] == '0');
    }
    if (!end) break;
    at = end + 1;
  }
  return false;
}

namespace {

// fp contraction stays at HIP's default (fast-honor-pragmas): the generated source switches it
// off inside its quad reductions, which a global -ffp-contract=fast would override.
const char* kOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
constexpr int kNumOptions = 3;

unsigned long long fnv1a(const std::string& s, unsigned long long h = 1469598103934665603ull) {
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ull;
  }
  return h;
}

std::string cache_dir() {
  if (const char* env = getenv("OKX_KERNEL_CACHE")) return env;
  Dl_info info;
  if (dladdr((const void*)&fnv1a, &info) && info.dli_fname) {
    std::string path = info.dli_fname;
    const size_t slash = path.rfind('/');
    path = slash == std::string::npos ? "." : path.substr(0, slash);
    return path + "/_kcache";
  }
  return "/tmp/okx_kcache";
}

bool read_file(const std::string& path, std::string* data) {
  FILE* fh = std::fopen(path.c_str(), "rb");
  if (!fh) return false;
  std::fseek(fh, 0, SEEK_END);
  const long size = std::ftell(fh);
  std::fseek(fh, 0, SEEK_SET);
  if (size <= 0) {
    std::fclose(fh);
    return false;
  }
  data->resize((size_t)size);
  const size_t got = std::fread(&(*data)[0], 1, (size_t)size, fh);
  std::fclose(fh);
  return got == (size_t)size;
}

// Cache entries carry a header (magic, payload size, FNV-1a of the payload): the HIP runtime aborts
// the process on a malformed code object, so a truncated or foreign file must be caught here.
constexpr unsigned long long kCacheMagic = 0x314b43584b4fULL;  // "OKXCK1"

std::string wrap_entry(const std::string& code) {
  unsigned long long header[3] = {kCacheMagic, (unsigned long long)code.size(), fnv1a(code)};
  std::string out(reinterpret_cast<const char*>(header), sizeof(header));
  out += code;
  return out;
}

bool unwrap_entry(const std::string& file, std::string* code) {
  unsigned long long header[3];
  if (file.size() < sizeof(header)) return false;
  std::memcpy(header, file.data(), sizeof(header));
  if (header[0] != kCacheMagic || header[1] != file.size() - sizeof(header)) return false;
  code->assign(file, sizeof(header), std::string::npos);
  return fnv1a(*code) == header[2];
}

void write_file_atomic(const std::string& path, const std::string& data) {
  static std::atomic<unsigned> serial{0};  // (two compile threads of one process may write the same entry)
  const std::string part = path + ".tmp" + std::to_string((long)getpid()) + "_" + std::to_string(serial.fetch_add(1));
  FILE* fh = std::fopen(part.c_str(), "wb");
  if (!fh) return;  // a read-only cache directory only costs recompilation
  const size_t put = std::fwrite(data.data(), 1, data.size(), fh);
  std::fclose(fh);
  if (put == data.size())
    std::rename(part.c_str(), path.c_str());
  else
    std::remove(part.c_str());
}

}  // namespace

const char* const kNotCached = "not in the kernel cache";

bool quad_compile(const std::string& src, std::string* code, std::string* err, bool ignore_cached, bool cache_only) {
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  unsigned long long h = fnv1a(src);
  for (int k = 0; k < kNumOptions; ++k) h = fnv1a(kOptions[k], h);
  h = fnv1a(std::to_string(major) + "." + std::to_string(minor), h);
  char name[64];
  std::snprintf(name, sizeof(name), "/okxq_%016llx", h);
  const std::string dir = cache_dir();
  const std::string path = dir + name + ".okxc";  // header + gfx950 code object
  if (!ignore_cached) {
    std::string file;
    if (read_file(path, &file) && unwrap_entry(file, code)) return true;
  }

  if (cache_only) {
    *err = kNotCached;
    return false;
  }
  hiprtcProgram prog;
  hiprtcResult rc = hiprtcCreateProgram(&prog, src.c_str(), "okx_quad.hip", 0, nullptr, nullptr);
  if (rc != HIPRTC_SUCCESS) {
    *err = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(rc);
    return false;
  }
  std::vector<const char*> options(kOptions, kOptions + kNumOptions);
  rc = hiprtcCompileProgram(prog, (int)options.size(), options.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t log_size = 0;
    (void)hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) (void)hiprtcGetProgramLog(prog, &log[0]);
    if (log.size() > 3000) log.resize(3000);
    *err = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(rc) + "\n" + log;
    (void)hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t size = 0;
  rc = hiprtcGetCodeSize(prog, &size);
  if (rc != HIPRTC_SUCCESS || size == 0) {
    *err = "hiprtcGetCodeSize failed";
    (void)hiprtcDestroyProgram(&prog);
    return false;
  }
  code->resize(size);
  rc = hiprtcGetCode(prog, &(*code)[0]);
  (void)hiprtcDestroyProgram(&prog);
  if (rc != HIPRTC_SUCCESS) {
    *err = "hiprtcGetCode failed";
    return false;
  }
  (void)mkdir(dir.c_str(), 0777);
  write_file_atomic(path, wrap_entry(*code));
  if (dev_switch("keep_source")) write_file_atomic(dir + name + ".hip", src);
  return true;
}

// Minimal reader for the two msgpack fields needed from the AMDGPU metadata note: kernel-level maps list their keys in
// alphabetical order, so the ".name" string nearest before a ".private_segment_fixed_size" key is that kernel's name
// (argument names sit under ".args", earlier in the map).
static int code_kernel_field(const std::string& code, const char* prefix, const char* key, bool name_follows, bool exact = false) {
  static const char kName[] = ".name";
  const size_t klen = std::strlen(key), nlen = sizeof(kName) - 1, plen = std::strlen(prefix);
  int best = -1;
  for (size_t at = code.find(key); at != std::string::npos; at = code.find(key, at + klen)) {
    const unsigned char* v = reinterpret_cast<const unsigned char*>(code.data()) + at + klen;
    const size_t left = code.size() - (at + klen);
    long long value = -1;
    if (left >= 1 && v[0] <= 0x7f) value = v[0];
    else if (left >= 2 && v[0] == 0xcc) value = v[1];
    else if (left >= 3 && v[0] == 0xcd) value = (v[1] << 8) | v[2];
    else if (left >= 5 && v[0] == 0xce) value = ((long long)v[1] << 24) | (v[2] << 16) | (v[3] << 8) | v[4];
    if (value < 0) continue;
    const size_t nm = name_follows ? code.find(kName, at) : code.rfind(kName, at);
    if (nm == std::string::npos) continue;
    if (name_follows && nm - at > 512) continue;  // the same kernel's map: a few short keys further on
    const unsigned char* s = reinterpret_cast<const unsigned char*>(code.data()) + nm + nlen;
    size_t len = 0, skip = 0;
    if ((s[0] & 0xe0) == 0xa0) len = s[0] & 0x1f, skip = 1;
    else if (s[0] == 0xd9) len = s[1], skip = 2;
    else if (s[0] == 0xda) len = (s[1] << 8) | s[2], skip = 3;
    else continue;
    if ((!name_follows && nm + nlen + skip + len > at) || len < plen || (exact && len != plen)) continue;
    if (std::memcmp(s + skip, prefix, plen) != 0) continue;
    if (value > best) best = (int)value;
  }
  return best;
}

int quad_code_scratch_bytes(const std::string& code, const char* prefix) {
  return code_kernel_field(code, prefix, ".private_segment_fixed_size", false);
}

int quad_code_kernel_scratch_bytes(const std::string& code, const char* name) {
  return code_kernel_field(code, name, ".private_segment_fixed_size", false, true);
}

// Static LDS bytes (largest over the kernels whose name starts with `prefix`): ".group_segment_fixed_size" sorts before
// ".name" in the kernel's metadata map.
int quad_code_lds_bytes(const std::string& code, const char* prefix) {
  return code_kernel_field(code, prefix, ".group_segment_fixed_size", true);
}

static const char* const kLaneKernels[8] = {"okx_lane_solve_u", "okx_lane_solve_u_c", "okx_lane_solve_g", "okx_lane_solve_g_c",
                                            "okx_lane_chain_u", "okx_lane_chain_u_c", "okx_lane_chain_g", "okx_lane_chain_g_c"};

bool lane_build(const DevProgram& P, std::string* src, std::string* code, std::string* why, bool ignore_cached, int* variant_out,
                int good_enough_scratch, bool cache_only, std::vector<LaneOverride>* overrides) {
  std::string src0, err;
  if (!lane_generate(P, &src0, why, 0)) return false;
  if (overrides) overrides->clear();
  char name[64];
  std::snprintf(name, sizeof(name), "/okxl_%016llx.lanevar", fnv1a(src0));
  const std::string memo = cache_dir() + name;
  int first = 0, last = lane_variant_count() - 1;
  std::vector<std::pair<std::string, int>> remembered;  // kernel -> the variant to take it from
  {
    // "<variant>" or "<variant> <kernel>=<variant> ..."
    std::string text;
    if (read_file(memo, &text)) {
      const int v = atoi(text.c_str());
      if (v >= 0 && v < lane_variant_count()) {
        first = last = v;
        for (size_t at = text.find(' '); at != std::string::npos; at = text.find(' ', at + 1)) {
          const size_t eq = text.find('=', at);
          if (eq == std::string::npos) break;
          const int ov = atoi(text.c_str() + eq + 1);
          if (ov >= 0 && ov < lane_variant_count() && ov != v) remembered.push_back({text.substr(at + 1, eq - at - 1), ov});
        }
      }
    }
  }
  int best = -1, best_scratch = 1 << 30;
  bool searched_all = true;  // every variant compiled and looked at (not cut short by `good_enough_scratch` or a failure)
  std::string best_src, best_code;
  std::vector<std::pair<int, std::string>> seen;  // (variant, code) of a search, for the per-kernel choice
  // Search order: 0, 12, 1, 13, ... - a small program's register layout (variants 0 - 11, okx_lanegen.cpp) alternates with
  // the LDS layout of the same hints (12 - 23), so that a program whose register form spills reaches the layout every
  // larger program has at its second compile (variants 12 - 23 do not exist for larger programs: skipped).
  std::vector<int> order;
  if (first == last) order.push_back(first);
  else
    for (int k = 0; k < lane_variant_count() / 2; ++k) {
      order.push_back(k);
      order.push_back(k + lane_variant_count() / 2);
    }
  for (size_t at = 0; at < order.size(); ++at) {
    const int v = order[at];
    std::string s1, c1, w1;
    if (v == 0) s1 = src0;
    else if (!lane_generate(P, &s1, &w1, v)) continue;
    if (!quad_compile(s1, &c1, &err, ignore_cached, cache_only)) {
      if (best < 0) *why = err == kNotCached ? err : "compile failed: " + err;
      searched_all = false;
      if (cache_only) break;  // (a variant search is a compile job)
      continue;
    }
    const int scratch = quad_code_scratch_bytes(c1, "okx_lane_solve");
    if (overrides && first != last) seen.push_back({v, c1});
    if (scratch >= 0 && scratch < best_scratch) {
      best = v;
      best_scratch = scratch;
      best_src.swap(s1);
      best_code.swap(c1);
    }
    if (best_scratch <= good_enough_scratch) {
      searched_all = searched_all && (at + 1 == order.size() || best_scratch == 0);
      break;
    }
  }
  if (best < 0) return false;
  std::string memo_text = std::to_string(best);
  if (overrides && first == last) {
    // a remembered choice: the other modules come from the cache (one that is not there any more is simply not used)
    for (const auto& rk : remembered) {
      std::string s1, c1, w1;
      if (!lane_generate(P, &s1, &w1, rk.second) || !quad_compile(s1, &c1, &err, false, true)) continue;
      overrides->push_back({rk.first, c1, rk.second, quad_code_kernel_scratch_bytes(c1, rk.first.c_str())});
    }
  } else if (overrides && searched_all && best_scratch > 0) {
    for (const char* kernel : kLaneKernels) {
      int have = quad_code_kernel_scratch_bytes(best_code, kernel), from = -1;
      if (have <= 0) continue;
      for (size_t k = 0; k < seen.size(); ++k) {
        if (!lane_variants_same_arithmetic(seen[k].first, best)) continue;
        const int sc = quad_code_kernel_scratch_bytes(seen[k].second, kernel);
        if (sc >= 0 && sc < have) have = sc, from = (int)k;
      }
      if (from < 0) continue;
      overrides->push_back({kernel, seen[from].second, seen[from].first, have});
      memo_text += std::string(" ") + kernel + "=" + std::to_string(seen[from].first);
    }
  }
  // (only a search that ran to its goal - no scratch, or the least of all variants - is remembered: a program whose every
  //  variant spills a little would otherwise be given the FIRST variant under the create-time bound at every start)
  if (first != last && (best_scratch == 0 || searched_all)) {
    (void)mkdir(cache_dir().c_str(), 0777);
    write_file_atomic(memo, memo_text + "\n");
  }
  *src = best_src;
  *code = best_code;
  if (variant_out) *variant_out = best;
  return true;
}

bool quad_eval_build(const DevProgram& P, const EvalSpec& spec, int waves_per_simd, std::string* code, std::string* why, bool cache_only) {
  std::string src, err;
  if (!quad_generate(P, waves_per_simd, &src, why, false, &spec)) return false;
  if (!quad_compile(src, code, &err, false, cache_only)) {
    *why = err == kNotCached ? err : "compile failed: " + err;
    return false;
  }
  return true;
}

bool quad_axle_eval_build(const DevProgram& P, const AxleEvalSpec& spec, int waves_per_simd, std::string* code, std::string* why, bool cache_only) {
  std::string src, err;
  if (!quad_generate(P, waves_per_simd, &src, why, false, nullptr, &spec)) return false;
  if (!quad_compile(src, code, &err, false, cache_only)) {
    *why = err == kNotCached ? err : "compile failed: " + err;
    return false;
  }
  return true;
}

// The lane form of the evaluated module: the emission variant whose okx_lane_evsolve_* kernels spill least (the search stops
// at the first one without scratch); the choice is remembered next to the code objects like lane_build's.
bool lane_eval_build(const DevProgram& P, const EvalSpec& spec, std::string* code, std::string* why, bool cache_only, int* scratch_out) {
  std::string src0, err;
  if (!lane_generate(P, &src0, why, 0, &spec)) return false;
  char name[64];
  std::snprintf(name, sizeof(name), "/okxe_%016llx.lanevar", fnv1a(src0));
  const std::string memo = cache_dir() + name;
  int first = 0, last = lane_variant_count() - 1;
  {
    std::string text;
    if (read_file(memo, &text)) {
      const int v = atoi(text.c_str());
      if (v >= 0 && v < lane_variant_count()) first = last = v;
    }
  }
  // only variants with the arithmetic of the program's plain lane module (lane_build's remembered choice): an evaluated
  // solve returns the plain solve's positions bit for bit
  int plain = 0;
  {
    std::string a1, a2, a3;
    int v = -1;
    if (lane_build(P, &a1, &a2, &a3, false, &v, 256, true) && v >= 0) plain = v;
  }
  int best = -1, best_scratch = 1 << 30;
  std::string best_code;
  bool searched_all = true;
  for (int v = first; v <= last; ++v) {
    if (!lane_variants_same_arithmetic(v, plain)) continue;
    std::string s1, c1, w1;
    if (v == 0) s1 = src0;
    else if (!lane_generate(P, &s1, &w1, v, &spec)) continue;
    if (!quad_compile(s1, &c1, &err, false, cache_only)) {
      if (best < 0) *why = err == kNotCached ? err : "compile failed: " + err;
      searched_all = false;
      if (cache_only) break;
      continue;
    }
    const int scratch = quad_code_scratch_bytes(c1, "okx_lane_evsolve");
    if (scratch >= 0 && scratch < best_scratch) {
      best = v;
      best_scratch = scratch;
      best_code.swap(c1);
    }
    if (best_scratch == 0) break;
  }
  if (best < 0) return false;
  if (first != last && (best_scratch == 0 || searched_all)) {
    (void)mkdir(cache_dir().c_str(), 0777);
    write_file_atomic(memo, std::to_string(best) + "\n");
  }
  *code = best_code;
  if (scratch_out) *scratch_out = best_scratch;
  return true;
}

constexpr int kPairScratchOk = 256;

bool quad_build(const DevProgram& P, int waves_per_simd, std::string* src, std::string* code, std::string* why,
                bool ignore_cached, bool cache_only) {
  std::string err;
  const bool force_lds = dev_switch("pair_lds_homes") && P.n_free > kQuadMaxFree;  // (tests: the LDS-homes layout of a pair program)
  if (!quad_generate(P, waves_per_simd, src, why, force_lds)) return false;
  if (!quad_compile(*src, code, &err, ignore_cached, cache_only)) {
    *why = err == kNotCached ? err : "compile failed: " + err;
    return false;
  }
  if (force_lds) return true;
  // Pair mode: up to kPairScratchOk bytes of scratch the register-resident layout stays - measured on the axle grid with
  // the second-order first step (164 B, almost all of it prologue temporaries): 0.403 ms per cold grid against 0.796 ms
  // for the LDS-homes layout, whose 42 KB of LDS leave three wavefronts per CU instead of four.
  if (P.n_free <= kQuadMaxFree || quad_code_scratch_bytes(*code, "okx_quad_solve") <= kPairScratchOk) return true;
  // beyond that (round 1's 636 B, re-read in every pass): constants and fixed points back to LDS, if that fits
  std::string src2, code2, why2;
  if (!quad_generate(P, waves_per_simd, &src2, &why2, true) || !quad_compile(src2, &code2, &err, ignore_cached, cache_only)) {
    if (cache_only && err == kNotCached) {  // the fallback layout was never compiled: the compile job decides
      *why = kNotCached;
      return false;
    }
    return true;
  }
  if (quad_code_lds_bytes(code2, "okx_quad_solve") > 40 * 1024) return true;  // four wavefronts per CU need <= 40 KB each
  if (quad_code_scratch_bytes(code2, "okx_quad_solve") < quad_code_scratch_bytes(*code, "okx_quad_solve")) {
    *src = src2;
    *code = code2;
  }
  return true;
}

}  // namespace okx
