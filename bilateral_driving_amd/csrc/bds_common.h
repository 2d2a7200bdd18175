// Shared launch / error helpers for libbds.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bds.h"

#define BDS_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return BDS_EINVAL; \
  } while (0)

#define BDS_LAUNCH_CHECK()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return BDS_ELAUNCH; \
  } while (0)

namespace bds {

constexpr int kWave = 64;  // CDNA wavefront

static inline hipStream_t as_stream(bds_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Length of a work list whose element count may live in DEVICE memory (a compaction result the host never reads: the `_dev` entry
// points of include/bds.h).  The launch is sized for the host-side capacity n_cap; surplus workgroups see no elements.
__device__ __forceinline__ int64_t list_length(int64_t n_cap, const uint64_t *__restrict__ n_dev) {
  if (n_dev == nullptr) return n_cap;
  const int64_t n = (int64_t)*n_dev;
  return n < n_cap ? n : n_cap;
}

// Row form of a view's projection outputs.  The five per-Gaussian arrays the projection leaves (means2d [N,2], depths [N], radii [N],
// conics [N,3], activated opacities [N]) are gathered by the tile stage and the record pack for the VISIBLE Gaussians only -- one
// partly used cache line per array and Gaussian.  A caller may hand them over as the COLUMNS of one [N,8] block of 32-byte rows
// {m2d.x, m2d.y, depth, radius (int bits) | conic a, b, c, opacity}: no signature changes, the layout is recognised by the addresses
// (depths == means2d + 2 can not hold for separate arrays of more than one row: they would overlap).  Strides in floats.
struct ProjLayout {
  int s2, sd, sc, so;   // means2d, depths, conics, opacities
  int row_radius;       // gather the radius from the row (word 3) instead of the dense radii array
};
static inline ProjLayout proj_layout(const float *means2d, const float *depths, const float *conics, const float *opacities) {
  ProjLayout p;
  const bool rows = means2d != nullptr && depths == means2d + 2;
  p.s2 = rows ? 8 : 2;
  p.sd = rows ? 8 : 1;
  p.row_radius = rows ? 1 : 0;
  p.sc = (rows && conics == means2d + 4) ? 8 : 3;
  p.so = (rows && opacities == means2d + 7) ? 8 : 1;      // (a caller may composite other opacities over the same rows: render_classes)
  return p;
}
__device__ __forceinline__ int proj_radius(const ProjLayout &pl, const float *__restrict__ means2d, const int32_t *__restrict__ radii, int64_t o) {
  return pl.row_radius ? __float_as_int(means2d[o * 8 + 3]) : radii[o];
}

// Row form of the per-Gaussian parameter GRADIENTS: v_means [N,3], v_quats [N,4], v_log_scales [N,3], v_logits [N] may be the columns
// of one [N,16] block of 64-byte rows {v_mean 3, v_logit | v_quat 4 | v_log_scale 3, -, | - - - -} (the list-driven backward then
// updates ONE line per visible Gaussian instead of four partly used ones); recognised by the addresses, as ProjLayout.
struct GradLayout { int sm, sq, ss, sl; };
static inline GradLayout grad_layout(const float *v_means, const float *v_quats, const float *v_log_scales, const float *v_logits) {
  const bool rows = v_means != nullptr && v_quats == v_means + 4 && v_log_scales == v_means + 8 && v_logits == v_means + 3;
  return rows ? GradLayout{16, 16, 16, 16} : GradLayout{3, 4, 3, 1};
}

// MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8 (observed; used for L2 locality only).
// Map a linear block id to a work item so that each XCD owns one contiguous range of items.
__device__ __forceinline__ int xcd_contiguous(int bid, int total) {
  constexpr int kXcd = 8;
  int per = total / kXcd, rem = total % kXcd;
  int xcd = bid % kXcd, slot = bid / kXcd;
  // XCD x owns items [start(x), start(x)+cnt(x)), cnt = per + (x < rem)
  int cnt = per + (xcd < rem ? 1 : 0);
  if (slot < cnt) return xcd * per + (xcd < rem ? xcd : rem) + slot;
  // tail blocks (slot >= cnt) cannot occur when grid == total, but stay safe:
  return bid;
}


// ---- wave64 sum that leaves the total in lane 63 (VALU-only: DPP row shifts + row broadcasts) --
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f<0x111, 0xf, true>(v);   // row_shr:1
  v += dpp_f<0x112, 0xf, true>(v);   // row_shr:2
  v += dpp_f<0x114, 0xf, true>(v);   // row_shr:4
  v += dpp_f<0x118, 0xf, true>(v);   // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_f<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1,3
  v += dpp_f<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
}

// ---- 16-value wave64 transpose-reduce ----------------------------------------------------------
// Sums 16 per-lane values over the 64 lanes in 35 VALU instructions (vs 16 x 6 for one-at-a-time
// DPP reductions): every step halves the number of live registers while it halves the lane distance
// -- v_permlane32_swap / v_permlane16_swap (gfx950) across rows, DPP row_ror / half_mirror / quad_perm
// inside a row.  On return lane l holds the wave total of value  k(l) = b2 + 2*b3 + 4*(l >> 4)
// (b2, b3 = bits 2, 3 of l), identical in the four lanes of each quad.
__device__ __forceinline__ int butterfly_slot(int lane) { return ((lane >> 2) & 1) + 2 * ((lane >> 3) & 1) + 4 * (lane >> 4); }

__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float butterfly_sum16(const float *v, int lane) {
  float s[8], t[4];
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = swap_add32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; i++) t[i] = swap_add16(s[i], s[i + 4]);
  const bool b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
  float u[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const float sa = t[j] + dpp_f<0x128, 0xf, false>(t[j]);          // row_ror:8
    const float sb = t[j + 2] + dpp_f<0x128, 0xf, false>(t[j + 2]);
    u[j] = b3 ? sb : sa;
  }
  const float sa = u[0] + dpp_f<0x141, 0xf, false>(u[0]);            // row_half_mirror
  const float sb = u[1] + dpp_f<0x141, 0xf, false>(u[1]);
  float w = b2 ? sb : sa;
  w += dpp_f<0xB1, 0xf, false>(w);                                   // quad_perm [1,0,3,2]
  w += dpp_f<0x4E, 0xf, false>(w);                                   // quad_perm [2,3,0,1]
  return w;
}

// Slots of the tile stage's prepare workspace that the ONE-VIEW projection fills when it takes over the stage's first launch (the
// count of visible Gaussians per 256-Gaussian workgroup, the tables the later launches add into): csrc/tiles.hip prep_reduce_slots.
struct PrepReduceSlots {
  uint32_t *sums256;     // [cdiv(N, 256)] visible Gaussians per projection workgroup
  uint32_t *zero_me;     // tables to clear
  int64_t zero_elems;
  uint64_t *m_total;     // cleared: the counting kernels accumulate into it
};
int prep_reduce_slots(void *ws, size_t ws_bytes, int64_t CN, PrepReduceSlots *out);   // BDS_OK, or why the short path does not apply

// test hooks (bds_set_option): force the large-input fallback paths of the tile stage; see include/bds.h
enum Option { kOptCapLaunch = 0 /* device-count tile stage: launches sized by the visible-entry capacity instead of C*N */, 
              /* (1, 2: retired -- LDS padding of the compositors, measured in rounds 3 and 5, profiles/NOTES.md) */ kOptDebug = 3 /* profiling only: ablation mask */, kOptShortSort = 4, kOptPacked = 6,
              kOptCells = 7 /* bilateral transform, bit 0: cell-aligned kernels where a level qualifies, bit 1: one-pass pyramid forward; 0 = general kernels */,
              kOptSchedBins = 8 /* device-count form: the forward compositor bins the backward's schedule itself (no sort launch) */, kOptCount = 9 };
int option_get(int which);

}  // namespace bds
