// okx_plan.hpp — device-resident form of a constraint program plus the static sparsity plans.
//
// The reference rebuilds, for every Jacobian call, a dense m x n matrix through Python
// closures (ResidualComputer.build_jac_plan / compute_jacobian, reference
// core/solver.py:281-581).  Here the structure is resolved ONCE on the host:
//   * every row knows which free-point blocks (3 columns each) it touches and how each of
//     its point slots maps onto them (directly, or through a derived point's chain blocks);
//   * every pair of blocks that shares a row gets the list of rows contributing to that
//     3x3 block of J^T J, so the normal equations are formed without ever touching zeros.
#pragma once

#include <stdint.h>

#include "../../include/okx.h"

#if defined(__HIPCC__)
#define OKX_HD __host__ __device__
#else
#define OKX_HD
#endif

namespace okx {

constexpr int kWave = 64;
constexpr int kMaxPoints = OKX_MAX_POINTS;   // 96
constexpr int kMaxVars = OKX_MAX_VARS;       // 126: one thread per variable (one wavefront up to 63, two beyond)
constexpr int kMaxFree = OKX_MAX_VARS / 3;   // 42
constexpr int kMaxDerived = 32;
constexpr int kMaxRows = OKX_MAX_ROWS;       // 128 (constraint rows + target rows)
constexpr int kMaxTargets = OKX_MAX_TARGETS;
constexpr int kDepMax = 4;                   // free blocks one derived point may depend on
constexpr int kRowBlkMax = 6;                // free blocks one row may touch
constexpr int kMaxPairs = kMaxFree * (kMaxFree + 1) / 2;
constexpr int kMaxContrib = kMaxRows * (kRowBlkMax * (kRowBlkMax + 1) / 2);
constexpr int kMaxGContrib = kMaxRows * kRowBlkMax;
constexpr int kRowTarget = OKX_ROW_TYPE_COUNT;  // internal row type of a target row
constexpr int kItemTermsMax = 16;
constexpr int kGradTermsMax = 16;

enum : int { kRefFixed = 0, kRefFree = 1, kRefDerived = 2 };

// How one input point of a row / derived op reaches the solver variables.  Packed into two
// words and decoded with shifts so that a lane never indexes a register-held array.
struct PointRef {
  uint32_t w0;  // kind | slot << 8 | src << 16 | nsrc << 24
  uint32_t w1;  // map[0] | map[1] << 8 | map[2] << 16 | map[3] << 24  (producer slot -> owner slot)
  OKX_HD int kind() const { return (int)(w0 & 0xff); }
  OKX_HD int slot() const { return (int)((w0 >> 8) & 0xff); }
  OKX_HD int src() const { return (int)((w0 >> 16) & 0xff); }
  OKX_HD int nsrc() const { return (int)(w0 >> 24); }
  OKX_HD int map(int j) const { return (int)((w1 >> (8 * j)) & 0xff); }
};
static_assert(sizeof(PointRef) == 8, "PointRef packs into 8 bytes");

struct DevProgram {
  int32_t n_points, n_free, n, n_derived, n_crows, n_targets, m, n_out;
  int32_t lda;          // (unused since the packed-triangle layout; kept for ABI stability of tools)
  int32_t js_stride;    // doubles per row of the block-sparse Jacobian
  int32_t n_active;     // derived ops needed while iterating
  int32_t n_pairs, n_items;
  int32_t lds_doubles;  // dynamic LDS size of the solve kernel, in doubles
  int32_t pad0, pad1;

  int32_t free_point[kMaxFree];
  int32_t out_point[kMaxPoints];

  int32_t dop_type[kMaxDerived];
  int32_t dop_out[kMaxDerived];
  int32_t dop_pts[kMaxDerived][4];
  int32_t dop_active[kMaxDerived];  // index into the active list or -1
  int32_t dop_nblk[kMaxDerived];
  int32_t dop_blk[kMaxDerived][kDepMax];
  int32_t active_op[kMaxDerived];   // active list -> op index (program order)
  PointRef dop_in[kMaxDerived][3];
  double dop_param[kMaxDerived];

  int32_t row_type[kMaxRows];
  int32_t row_pts[kMaxRows][4];
  int32_t row_nblk[kMaxRows];
  int32_t row_blk[kMaxRows][kRowBlkMax];
  PointRef row_in[kMaxRows][4];
  double row_param[kMaxRows][OKX_ROW_PARAMS];  // targets: dir in q0..2, target index in q3

  int32_t pair_p[kMaxPairs];
  int32_t pair_q[kMaxPairs];
  int32_t pair_start[kMaxPairs + 1];
  uint16_t contrib[kMaxContrib];    // row | slot_p << 7 | slot_q << 10
  int32_t g_start[kMaxFree + 1];
  uint16_t g_contrib[kMaxGContrib]; // row | slot << 7

  // Flattened J^T J work items (one per structurally non-zero scalar entry of the upper
  // triangle incl. diagonal); staged into LDS by the solve kernel.
  int32_t n_work;                       // number of valid items
  int32_t n_contrib;                    // entries used in contrib[]
  int32_t n_gcontrib;                   // entries used in g_contrib[]
  int32_t pad2;
  int32_t item_dst[kMaxPairs * 9];      // >= 0: packed-triangle offset col*(col-1)/2 + row (row < col); < 0: -(1 + diag index)
  uint32_t item_desc[kMaxPairs * 9];    // start | count << 12 | a << 20 | b << 22

  // Batched-load form of the same plans: every product term is a pair of absolute offsets
  // into one Jacobian buffer, padded per item to `kc` terms with the always-zero slot, so a
  // lane can issue all loads of an item at once instead of walking a dependent list.
  int32_t kc;                            // terms per J^T J item (multiple of 4, <= kItemTermsMax)
  int32_t kg;                            // terms per J^T r variable (multiple of 4, <= kGradTermsMax)
  int32_t zero_off;                      // offset of the zero slot inside a Jacobian buffer (= m * js_stride)
  int32_t pad3;
  uint32_t row_first[kMaxRows];          // scatter: bit (4 s + j) set = that write is the first to its slot
  alignas(16) uint32_t item_terms[kMaxPairs * 9 * kItemTermsMax];  // offA | offB << 16
  alignas(16) uint32_t grad_terms[kMaxVars * kGradTermsMax];       // offJ | row << 16  (row = m: zero residual slot)

  double design_pos[kMaxPoints][3];
};

// Builds the device program and all plans from the C-ABI descriptor.  Returns OKX_OK or a
// negative okx_status and fills `err` (at most errlen bytes).
int build_dev_program(const okx_program_desc* desc, DevProgram* out, char* err, int errlen);

}  // namespace okx
