// Multi-GPU exchange of the per-Gaussian gradient rows (no reference counterpart: the reference trains one view per step on one
// GPU, /root/reference/project/models/trainers/base.py:411).  dist.FrameExchange all-reduces, per view, only the rows of the
// Gaussians that SOME rank sees: the union of the ranks' visibility masks gives every such Gaussian a slot of a compact buffer
// [capacity, 3 + 4 + 3 + 1 + K * 3].  This file turns the reduced mask into everything the exchange needs, on the device and in
// two launches (it was ten framework operators on N-element tensors, in a loop that is host-bound as it is):
//   row_map [N] i32   slot of Gaussian g (clamped to capacity - 1 beyond the capacity: the overflow is reported, the frame repeated)
//   ids [capacity] i32 Gaussian at slot s, -1 beyond the union
//   the compact buffer's rows [0, count) zeroed (the list-driven backward kernels STORE the rows of the Gaussians this rank sees;
//   rows of Gaussians only other ranks see must read as zero)
//   count -> device word + page-locked host word (the host looks at it after the fact)
// HBM-bound byte work: N mask bytes in, 4 N + 4 capacity out, 4 * row_floats per union member zeroed.
#include "bds_common.h"

namespace bds {

constexpr int kUnionBlock = 256;
constexpr int kUnionItems = 16;                       // mask bytes per thread: one 16-byte load
constexpr int kUnionTile = kUnionBlock * kUnionItems;  // 4096 Gaussians per workgroup

__device__ __forceinline__ uint32_t union_block_scan(uint32_t v, uint32_t &total, uint32_t *lw) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == kWave - 1) lw[wv] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kUnionBlock / kWave; w++) {
    const uint32_t s = lw[w];
    if (w < wv) base += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(kUnionBlock) void union_count_kernel(int64_t N, const uint8_t *__restrict__ mask,
                                                                 uint32_t *__restrict__ tile_sums) {
  __shared__ uint32_t lw[kUnionBlock / kWave + 1];
  const int64_t base = (int64_t)blockIdx.x * kUnionTile + (int64_t)threadIdx.x * kUnionItems;
  uint32_t s = 0;
  if (base + kUnionItems <= N && (reinterpret_cast<uintptr_t>(mask + base) & 15u) == 0) {
    const uint4 m = *reinterpret_cast<const uint4 *>(mask + base);
    const uint32_t w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int b = 0; b < 4; b++) s += ((w[q] >> (8 * b)) & 0xffu) ? 1u : 0u;
  } else {
    for (int i = 0; i < kUnionItems; i++)
      if (base + i < N) s += mask[base + i] ? 1u : 0u;
  }
  uint32_t tot;
  union_block_scan(s, tot, lw);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kUnionBlock) void union_slots_kernel(
    int64_t N, const uint8_t *__restrict__ mask, const uint32_t *__restrict__ tile_sums, int64_t cap, int K,
    int32_t *__restrict__ row_map, int32_t *__restrict__ ids, float *__restrict__ b_means, float *__restrict__ b_quats,
    float *__restrict__ b_log_scales, float *__restrict__ b_logits, float *__restrict__ b_sh, uint64_t *__restrict__ count_dev,
    volatile int64_t *__restrict__ count_host) {
  __shared__ uint32_t lw[kUnionBlock / kWave + 1];
  __shared__ uint32_t s_off, s_cnt;
  // this tile's first slot = union members in the tiles in front of it (a few hundred L2-resident sums)
  uint32_t part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kUnionBlock) part += tile_sums[b];
  uint32_t my_offset;
  union_block_scan(part, my_offset, lw);
  const int64_t base = (int64_t)blockIdx.x * kUnionTile + (int64_t)threadIdx.x * kUnionItems;
  bool in[kUnionItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kUnionItems; i++) {
    in[i] = base + i < N && mask[base + i] != 0;
    s += in[i] ? 1u : 0u;
  }
  uint32_t tot;
  uint32_t slot = union_block_scan(s, tot, lw) + my_offset;
  if (threadIdx.x == 0) { s_off = my_offset; s_cnt = tot; }
  // running slot of every Gaussian (cumsum(mask) - 1 of the framework formulation: a non-member carries the slot of the last member
  // in front of it, -1 -> 0 clamped), members' ids into the list
#pragma unroll
  for (int i = 0; i < kUnionItems; i++) {
    if (base + i < N) {
      if (in[i]) {
        if ((int64_t)slot < cap) ids[slot] = (int32_t)(base + i);
        row_map[base + i] = (int32_t)((int64_t)slot < cap ? slot : cap - 1);
        slot++;
      } else {
        const int64_t sl = (int64_t)slot - 1;
        row_map[base + i] = (int32_t)(sl < 0 ? 0 : (sl < cap ? sl : cap - 1));
      }
    }
  }
  __syncthreads();
  // zero the compact rows of this tile's slots (stored into by this rank's visible Gaussians, read as zero for the others')
  const int64_t s0 = s_off, s1 = (int64_t)s_off + s_cnt < cap ? (int64_t)s_off + s_cnt : cap;
  if (s1 > s0) {
    const int64_t n = s1 - s0;
    for (int64_t e = threadIdx.x; e < n * 3; e += kUnionBlock) { b_means[s0 * 3 + e] = 0.f; b_log_scales[s0 * 3 + e] = 0.f; }
    for (int64_t e = threadIdx.x; e < n * 4; e += kUnionBlock) b_quats[s0 * 4 + e] = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += kUnionBlock) b_logits[s0 + e] = 0.f;
    const int64_t row = (int64_t)K * 3;
    for (int64_t e = threadIdx.x; e < n * row; e += kUnionBlock) b_sh[s0 * row + e] = 0.f;
  }
  if (blockIdx.x == gridDim.x - 1) {   // the last tile knows the union's size: pad the id list, publish the count
    const int64_t count = (int64_t)my_offset + tot;
    for (int64_t e = count + threadIdx.x; e < cap; e += kUnionBlock) ids[e] = -1;
    if (threadIdx.x == 0) {
      if (count_dev) *count_dev = (uint64_t)count;
      if (count_host) { count_host[0] = count; __threadfence_system(); }
    }
  }
}

}  // namespace bds

using namespace bds;

extern "C" size_t bds_union_slots_workspace_bytes(int64_t N) {
  return N < 0 ? 0 : sizeof(uint32_t) * (size_t)(cdiv(N > 0 ? N : 1, kUnionTile) + 1);
}

extern "C" int bds_union_slots(int64_t N, const uint8_t *mask, int64_t capacity, int K, int32_t *row_map, int32_t *ids,
                               float *b_means, float *b_quats, float *b_log_scales, float *b_logits, float *b_sh, void *ws,
                               size_t ws_bytes, uint64_t *count_dev, int64_t *count_pinned, bds_stream_t stream) {
  BDS_REQUIRE(N > 0 && N < ((int64_t)1 << 31) && capacity > 0 && capacity <= ((int64_t)1 << 31) - 1 && K >= 1 && K <= 16);
  BDS_REQUIRE(mask && row_map && ids && b_means && b_quats && b_log_scales && b_logits && b_sh && ws);
  if (ws_bytes < bds_union_slots_workspace_bytes(N)) return BDS_EWORKSPACE;
  const unsigned tiles = (unsigned)cdiv(N, kUnionTile);
  BDS_REQUIRE(tiles <= 65535u * 8u);
  void *mapped = nullptr;
  if (count_pinned && hipHostGetDevicePointer(&mapped, count_pinned, 0) != hipSuccess) { (void)hipGetLastError(); return BDS_EINVAL; }
  hipStream_t st = as_stream(stream);
  uint32_t *sums = static_cast<uint32_t *>(ws);
  hipLaunchKernelGGL(union_count_kernel, dim3(tiles), dim3(kUnionBlock), 0, st, N, mask, sums);
  hipLaunchKernelGGL(union_slots_kernel, dim3(tiles), dim3(kUnionBlock), 0, st, N, mask, sums, capacity, K, row_map, ids, b_means,
                     b_quats, b_log_scales, b_logits, b_sh, count_dev, static_cast<volatile int64_t *>(mapped));
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
