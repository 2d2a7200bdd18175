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

// run-time selectable kernel variants (bds_set_option): A/B measurement and bisecting
enum Option { kOptRasterBwd = 0, kOptRadix = 1, kOptCount = 8 };
int option_get(int which);

}  // namespace bds
