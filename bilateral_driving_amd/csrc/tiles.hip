// K4-K6: tile intersection, (tile|depth) ordering, per-tile offsets.
// isect_tiles / radix sort / isect_offset_encode stages of gsplat.rendering.rasterization
// (/root/reference/project/models/trainers/base.py:393-408).  Integer work; the outputs
// (flatten_ids, isect_ids, isect_offsets) are bit-identical to a stable sort of the 64-bit
// keys  camera|tile << 32 | fp32-depth-bits  emitted in Gaussian order.
//
// MI355X-first restructuring (HBM-bound integer work; no 64-bit key sort of M duplicated keys):
//   1. depth-order the C*N Gaussians ONCE (32-bit keys, 4 radix passes over C*N, not over M);
//   2. emit (camera*tiles + tile, id) pairs in that order (M pairs, 32-bit keys);
//   3. stable-sort the pairs by the tile key only: ceil(log2(C*tiles)) bits -> 2 passes at 1080p
//      instead of 6 passes over 12-byte records.
// A stable sort by tile of a depth-ordered sequence is exactly the (tile, depth, emission) order.
#include "bds_common.h"
#include "gs_math.h"

namespace bds {

// ------------------------------------------------------------------------------------------
// exclusive scan of uint32 (three-phase, recursive over block sums)
// ------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;  // 2048

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// returns the exclusive prefix of `v` within the block and the block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t &total, uint32_t *lds_w /*>= waves+1*/) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const int nw = blockDim.x / kWave;
  uint32_t inc = wave_incl_scan(v);
  if (lane == kWave - 1) lds_w[wv] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < nw; w++) {
    uint32_t s = lds_w[w];
    if (w < wv) base += s;
    tot += s;
  }
  total = tot;
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_reduce_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                                uint32_t *__restrict__ block_sums) {
  __shared__ uint32_t lw[kScanBlock / kWave + 1];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++)
    if (base + i < n) s += in[base + i];
  uint32_t tot;
  block_excl_scan(s, tot, lw);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single-block in-place exclusive scan of a short array (n <= kScanTile)
__global__ __launch_bounds__(kScanBlock) void scan_small_kernel(uint32_t *__restrict__ data, int64_t n,
                                                               uint64_t *__restrict__ total_out) {
  __shared__ uint32_t lw[kScanBlock / kWave + 1];
  const int64_t base = (int64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    v[i] = base + i < n ? data[base + i] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, tot, lw);
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) data[base + i] = ex;
    ex += v[i];
  }
  if (total_out && threadIdx.x == 0) *total_out = tot;
}

// kSelfOffset: block_offsets holds the raw tile TOTALS and every workgroup sums the ones in front of it itself
// (at most a few hundred values) -- saves the separate scan-of-totals launch for all but huge inputs.
template <bool kSelfOffset>
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                               const uint32_t *__restrict__ block_offsets,
                                                               uint32_t *__restrict__ out, uint64_t *__restrict__ total_out) {
  __shared__ uint32_t lw[kScanBlock / kWave + 1];
  uint32_t my_offset;
  if (kSelfOffset) {
    uint32_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kScanBlock) part += block_offsets[b];
    uint32_t tot;
    block_excl_scan(part, tot, lw);
    my_offset = tot;
  } else {
    my_offset = block_offsets[blockIdx.x];
  }
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    v[i] = base + i < n ? in[base + i] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, tot, lw) + my_offset;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
  if (kSelfOffset && total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = (uint64_t)my_offset + tot;
}

// uint32 elements of temp needed to scan n elements
static size_t scan_temp_elems(int64_t n) {
  size_t tot = 0;
  while (n > kScanTile) {
    n = cdiv(n, kScanTile);
    tot += align_up((size_t)n, 4);
  }
  return tot + 4;
}

// out may alias in.  total_out (device uint64, may be null) receives the grand total
// (valid as long as it fits 32 bits, which the callers guarantee).
static int exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *temp, uint64_t *total_out,
                              hipStream_t st) {
  if (n <= kScanTile) {
    if (in != out) {
      if (hipMemcpyAsync(out, in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st) != hipSuccess) return BDS_ELAUNCH;
    }
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kScanBlock), 0, st, out, n, total_out);
    BDS_LAUNCH_CHECK();
    return BDS_OK;
  }
  const int64_t nb = cdiv(n, kScanTile);
  uint32_t *sums = temp;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, st, in, n, sums);
  BDS_LAUNCH_CHECK();
  if (nb <= 4096) {  // two launches: every workgroup derives its own offset from the tile totals
    hipLaunchKernelGGL((scan_apply_kernel<true>), dim3((unsigned)nb), dim3(kScanBlock), 0, st, in, n, sums, out, total_out);
    BDS_LAUNCH_CHECK();
    return BDS_OK;
  }
  int rc = exclusive_scan_u32(sums, sums, nb, temp + align_up((size_t)nb, 4), total_out, st);
  if (rc != BDS_OK) return rc;
  hipLaunchKernelGGL((scan_apply_kernel<false>), dim3((unsigned)nb), dim3(kScanBlock), 0, st, in, n, sums, out, nullptr);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

// ------------------------------------------------------------------------------------------
// stable LSD radix pass on (uint32 key, uint32 value), digit of <= 8 bits
// ------------------------------------------------------------------------------------------
constexpr int kSortBlock = 256;
constexpr int kSortRounds = 16;
constexpr int kSortChunk = kSortBlock * kSortRounds;  // 4096 pairs per workgroup
constexpr int kSortWaves = kSortBlock / kWave;

// n_dev (optional): the element count lives in device memory (a compaction result the host never
// reads); the launch is sized for the host-side upper bound n_host and surplus workgroups see no elements.
__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(const uint32_t *__restrict__ keys, int64_t n_host,
                                                               const uint64_t *__restrict__ n_dev, int shift,
                                                               uint32_t mask, int nblocks,
                                                               uint32_t *__restrict__ hist /*[256][nblocks]*/) {
  __shared__ uint32_t h[256];
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortChunk;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; r++) {
    int64_t i = base + r * kSortBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  if (threadIdx.x <= mask) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Scatter with wave-private ranking: each wave owns a contiguous quarter of the workgroup's chunk and ranks it without
// cross-wave traffic (one LDS histogram + prefix over (wave, digit), then 16 rounds with NO barriers: ballot match).
// Stable: position = scanned block base + elements of earlier waves + earlier rounds of this wave + rank in the round.
// The workgroup's 4096 pairs are first ordered by digit in LDS and then
// written out by consecutive threads: a digit's run leaves as whole cache lines (4096 / 2^bits pairs at a time)
// instead of the 8-pair fragments of one wave round.  Pays on the long tile passes, where the scatter is write-bound.
__global__ __launch_bounds__(kSortBlock) void radix_scatter_lds_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, int64_t n_host,
    const uint64_t *__restrict__ n_dev, int shift, uint32_t mask, int bits, int nblocks,
    const uint32_t *__restrict__ hist_scanned, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
  const int64_t n = n_dev ? (int64_t)*n_dev : n_host;
  const int64_t bbase = (int64_t)blockIdx.x * kSortChunk;
  if (bbase >= n) return;
  __shared__ uint32_t wrun[kSortWaves][256];  // next LOCAL position per (wave, digit)
  __shared__ uint32_t dstart[256], gbase[256];
  __shared__ uint32_t lw[kSortBlock / kWave + 1];
  __shared__ uint32_t lk[kSortChunk], lv[kSortChunk];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
#pragma unroll
  for (int w = 0; w < kSortWaves; w++) wrun[w][tid] = 0;
  __syncthreads();
  constexpr int kPerWave = kSortChunk / kSortWaves;
  const int64_t wbase = bbase + (int64_t)wv * kPerWave;
  uint32_t k[kSortRounds], v[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    k[r] = 0xFFFFFFFFu; v[r] = 0;
    if (i < n) {
      k[r] = keys_in[i]; v[r] = vals_in[i];
      atomicAdd(&wrun[wv][(k[r] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) tot += wrun[w][tid];
    uint32_t all;
    uint32_t base = block_excl_scan(tot, all, lw);
    dstart[tid] = base;
    gbase[tid] = tid <= (int)mask ? hist_scanned[(int64_t)tid * nblocks + blockIdx.x] : 0u;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
      const uint32_t c = wrun[w][tid];
      wrun[w][tid] = base;
      base += c;
    }
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    const bool on = i < n;
    const uint32_t d = (k[r] >> shift) & mask;
    unsigned long long peers = __ballot(on);
    for (int b = 0; b < bits; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(peers & lt);
    uint32_t pos = 0;
    if (on) pos = wrun[wv][d];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (on && rank == 0) wrun[wv][d] = pos + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (on) { lk[pos + rank] = k[r]; lv[pos + rank] = v[r]; }
  }
  __syncthreads();
  const int cnt = (int)((n - bbase) < (int64_t)kSortChunk ? (n - bbase) : (int64_t)kSortChunk);
  for (int i = tid; i < cnt; i += kSortBlock) {
    const uint32_t kk = lk[i];
    const uint32_t d = (kk >> shift) & mask;
    const uint32_t g = gbase[d] + ((uint32_t)i - dstart[d]);
    keys_out[g] = kk;
    vals_out[g] = lv[i];
  }
}

// Keys-only form of the digit-ordered scatter, for the PACKED tile lists: an entry is one 32-bit word
// (tile << rank_bits | depth rank of the Gaussian among the visible ones), so a pass moves 4 bytes per entry instead
// of 8.  On the last pass (`unpack` != null) every entry's list value (Gaussian id, or compact position) is looked up from its
// rank and written to vals_out next to the packed word.
__global__ __launch_bounds__(kSortBlock) void radix_scatter_keys_kernel(
    const uint32_t *__restrict__ keys_in, int64_t n_host, const uint64_t *__restrict__ n_dev, int shift, uint32_t mask, int bits,
    int nblocks, const uint32_t *__restrict__ hist_scanned, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ unpack,
    uint32_t rank_mask, uint32_t *__restrict__ vals_out) {
  const int64_t n = list_length(n_host, n_dev);
  const int64_t bbase = (int64_t)blockIdx.x * kSortChunk;
  if (bbase >= n) return;
  __shared__ uint32_t wrun[kSortWaves][256];
  __shared__ uint32_t dstart[256], gbase[256];
  __shared__ uint32_t lw[kSortBlock / kWave + 1];
  __shared__ uint32_t lk[kSortChunk];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
#pragma unroll
  for (int w = 0; w < kSortWaves; w++) wrun[w][tid] = 0;
  __syncthreads();
  constexpr int kPerWave = kSortChunk / kSortWaves;
  const int64_t wbase = bbase + (int64_t)wv * kPerWave;
  uint32_t k[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    k[r] = 0xFFFFFFFFu;
    if (i < n) {
      k[r] = keys_in[i];
      atomicAdd(&wrun[wv][(k[r] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) tot += wrun[w][tid];
    uint32_t all;
    uint32_t base = block_excl_scan(tot, all, lw);
    dstart[tid] = base;
    gbase[tid] = tid <= (int)mask ? hist_scanned[(int64_t)tid * nblocks + blockIdx.x] : 0u;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
      const uint32_t c = wrun[w][tid];
      wrun[w][tid] = base;
      base += c;
    }
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    const bool on = i < n;
    const uint32_t d = (k[r] >> shift) & mask;
    unsigned long long peers = __ballot(on);
    for (int b = 0; b < bits; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(peers & lt);
    uint32_t pos = 0;
    if (on) pos = wrun[wv][d];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (on && rank == 0) wrun[wv][d] = pos + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (on) lk[pos + rank] = k[r];
  }
  __syncthreads();
  const int cnt = (int)((n - bbase) < (int64_t)kSortChunk ? (n - bbase) : (int64_t)kSortChunk);
  for (int i = tid; i < cnt; i += kSortBlock) {
    const uint32_t kk = lk[i];
    const uint32_t d = (kk >> shift) & mask;
    const uint32_t g = gbase[d] + ((uint32_t)i - dstart[d]);
    keys_out[g] = kk;
    if (unpack) vals_out[g] = unpack[kk & rank_mask];
  }
}


// ---- wide-digit form of the two kernels above, for the packed tile pass: digits of up to 10 bits (kBins = 512 / 1024) --------
// With coarse list tiles the whole tile key is 9-10 bits (1080p, 64-px list tiles: 510 lists): ONE stable pass orders the
// pairs instead of two (a pass over the list costs a histogram, a scan of [bins][workgroups] and a scatter launch).
// No scan launches either: as in the short sort below, the histogram is stored workgroup-major, every 64 workgroups also add
// theirs to a group row (zeroed by the emission kernel in front), and a scatter workgroup derives its bases from <= ng group rows +
// <= 63 workgroup rows (L2-resident) -- two launches per pass instead of four, which matters most where it runs: next to another
// stream's compositor every launch of this latency-bound stage waits for wave slots (profiles/NOTES.md, round 4).
constexpr int kWideGroupShift = 6;
// (the kernels of the device-count tile stage are thin wrappers around *_block bodies that take their workgroup index and element
//  count as arguments: the launch shape is not baked into the bodies.  Round 5 ran the same bodies phase by phase inside ONE
//  persistent launch with device-wide barriers; it lost -- every barrier is an L2 write-back + invalidate per workgroup on this
//  multi-XCD part -- and was removed again: profiles/NOTES.md)
template <int kBins>
struct HistWideSh { uint32_t h[kBins]; };
template <int kBins>
__device__ __forceinline__ void radix_hist_wide_block(HistWideSh<kBins> &sh, int vb, const uint32_t *keys, int64_t n, int shift, uint32_t mask,
                                                      uint32_t *hist /*[nblocks][kBins]*/, uint32_t *ghist /*[ng][kBins], zeroed*/) {
  const int64_t base = (int64_t)vb * kSortChunk;
  if (base >= n) return;   // (rows of surplus workgroups are never read)
  for (int d = threadIdx.x; d < kBins; d += kSortBlock) sh.h[d] = 0;
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = base + r * kSortBlock + threadIdx.x;
    if (i < n) atomicAdd(&sh.h[(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kBins; d += kSortBlock) {
    const uint32_t c = sh.h[d];
    hist[(int64_t)vb * kBins + d] = c;
    if (c) atomicAdd(&ghist[(int64_t)(vb >> kWideGroupShift) * kBins + d], c);
  }
}
template <int kBins>
__global__ __launch_bounds__(kSortBlock) void radix_hist_wide_kernel(const uint32_t *__restrict__ keys, int64_t n_host,
                                                                    const uint64_t *__restrict__ n_dev, int shift, uint32_t mask,
                                                                    uint32_t *__restrict__ hist /*[nblocks][kBins]*/,
                                                                    uint32_t *__restrict__ ghist /*[ng][kBins], zeroed*/) {
  __shared__ HistWideSh<kBins> sh;
  radix_hist_wide_block<kBins>(sh, (int)blockIdx.x, keys, list_length(n_host, n_dev), shift, mask, hist, ghist);
}

template <int kBins>
struct ScatterWideSh {
  uint32_t wrun[kSortWaves][kBins];
  uint32_t dstart[kBins], gbase[kBins];
  uint32_t lw[kSortBlock / kWave + 1];
  uint32_t lk[kSortChunk];
};
template <int kBins>
__device__ __forceinline__ void radix_scatter_keys_wide_block(
    ScatterWideSh<kBins> &sh, int vb, const uint32_t *keys_in, int64_t n, int shift, uint32_t mask, int bits, const uint32_t *hist,
    const uint32_t *ghist, uint32_t *keys_out, const uint32_t *unpack, uint32_t rank_mask, uint32_t *vals_out, int32_t *offsets_out,
    int n_offsets) {
  constexpr int kPer = kBins / kSortBlock;   // digits per thread (consecutive: thread t owns [t * kPer, (t + 1) * kPer))
  const int64_t bbase = (int64_t)vb * kSortChunk;
  if (bbase >= n && !(offsets_out && vb == 0)) return;   // (an empty list still owes its per-tile offsets: all zero)
  auto &wrun = sh.wrun;
  auto &dstart = sh.dstart;
  auto &gbase = sh.gbase;
  auto &lk = sh.lk;
  uint32_t *lw = sh.lw;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  for (int d = tid; d < kBins; d += kSortBlock) {
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) wrun[w][d] = 0;
  }
  __syncthreads();
  constexpr int kPerWave = kSortChunk / kSortWaves;
  const int64_t wbase = bbase + (int64_t)wv * kPerWave;
  uint32_t k[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    k[r] = 0xFFFFFFFFu;
    if (i < n) {
      k[r] = keys_in[i];
      atomicAdd(&wrun[wv][(k[r] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  {
    uint32_t tot[kPer], sum = 0;
#pragma unroll
    for (int u = 0; u < kPer; u++) {
      const int d = tid * kPer + u;
      tot[u] = 0;
#pragma unroll
      for (int w = 0; w < kSortWaves; w++) tot[u] += wrun[w][d];
      sum += tot[u];
    }
    // this thread's digits in the whole input (gt) and in front of this workgroup (below): group rows, then the <= 63 workgroup rows
    // of its own group (independent partial sums: several loads in flight)
    uint32_t gt[kPer], below[kPer];
#pragma unroll
    for (int u = 0; u < kPer; u++) { gt[u] = 0; below[u] = 0; }
    {
      const int nb = (int)((n + kSortChunk - 1) / kSortChunk), ng = (nb + (1 << kWideGroupShift) - 1) >> kWideGroupShift;
      const int gb = vb >> kWideGroupShift;
      for (int g = 0; g < ng; g++) {
        const uint32_t *q = ghist + (int64_t)g * kBins + tid * kPer;
#pragma unroll
        for (int u = 0; u < kPer; u++) {
          const uint32_t c = q[u];
          gt[u] += c;
          if (g < gb) below[u] += c;
        }
      }
      const uint32_t *hp = hist + ((int64_t)gb << kWideGroupShift) * kBins + tid * kPer;
      const int nrows = vb - (gb << kWideGroupShift);
      uint32_t p0[kPer], p1[kPer], p2[kPer], p3[kPer];
#pragma unroll
      for (int u = 0; u < kPer; u++) { p0[u] = 0; p1[u] = 0; p2[u] = 0; p3[u] = 0; }
      int r = 0;
      for (; r + 4 <= nrows; r += 4) {
        const uint32_t *q = hp + (int64_t)r * kBins;
#pragma unroll
        for (int u = 0; u < kPer; u++) { p0[u] += q[u]; p1[u] += q[kBins + u]; p2[u] += q[2 * kBins + u]; p3[u] += q[3 * kBins + u]; }
      }
      for (; r < nrows; r++) {
#pragma unroll
        for (int u = 0; u < kPer; u++) p0[u] += hp[(int64_t)r * kBins + u];
      }
#pragma unroll
      for (int u = 0; u < kPer; u++) below[u] += (p0[u] + p1[u]) + (p2[u] + p3[u]);
    }
    uint32_t gsum = 0;
#pragma unroll
    for (int u = 0; u < kPer; u++) gsum += gt[u];
    uint32_t all;
    uint32_t dbase = block_excl_scan(gsum, all, lw);   // entries of smaller digits in the whole input
    uint32_t base = block_excl_scan(sum, all, lw);
#pragma unroll
    for (int u = 0; u < kPer; u++) {
      const int d = tid * kPer + u;
      dstart[d] = base;
      gbase[d] = dbase + below[u];
      dbase += gt[u];
      // when this pass covers the whole tile key, the first workgroup's bases ARE the per-tile offsets (entries with a smaller key)
      if (offsets_out && vb == 0 && d < n_offsets) offsets_out[d] = (int32_t)gbase[d];
      uint32_t b2 = base;
#pragma unroll
      for (int w = 0; w < kSortWaves; w++) {
        const uint32_t c = wrun[w][d];
        wrun[w][d] = b2;
        b2 += c;
      }
      base += tot[u];
    }
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    const bool on = i < n;
    const uint32_t d = (k[r] >> shift) & mask;
    unsigned long long peers = __ballot(on);
    for (int b = 0; b < bits; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(peers & lt);
    uint32_t pos = 0;
    if (on) pos = wrun[wv][d];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (on && rank == 0) wrun[wv][d] = pos + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (on) lk[pos + rank] = k[r];
  }
  __syncthreads();
  const int cnt = (int)((n - bbase) < (int64_t)kSortChunk ? (n - bbase) : (int64_t)kSortChunk);
  for (int i = tid; i < cnt; i += kSortBlock) {
    const uint32_t kk = lk[i];
    const uint32_t d = (kk >> shift) & mask;
    const uint32_t g = gbase[d] + ((uint32_t)i - dstart[d]);
    keys_out[g] = kk;
    if (unpack) vals_out[g] = unpack[kk & rank_mask];
  }
}
template <int kBins>
__global__ __launch_bounds__(kSortBlock) void radix_scatter_keys_wide_kernel(
    const uint32_t *__restrict__ keys_in, int64_t n_host, const uint64_t *__restrict__ n_dev, int shift, uint32_t mask, int bits,
    const uint32_t *__restrict__ hist, const uint32_t *__restrict__ ghist, uint32_t *__restrict__ keys_out,
    const uint32_t *__restrict__ unpack, uint32_t rank_mask, uint32_t *__restrict__ vals_out, int32_t *__restrict__ offsets_out,
    int n_offsets) {
  __shared__ ScatterWideSh<kBins> sh;
  radix_scatter_keys_wide_block<kBins>(sh, (int)blockIdx.x, keys_in, list_length(n_host, n_dev), shift, mask, bits, hist, ghist, keys_out,
                                       unpack, rank_mask, vals_out, offsets_out, n_offsets);
}

constexpr int kWideBits = 10;   // widest digit of the packed tile pass

static size_t radix_temp_elems(int64_t n, int max_bins = 256) {
  const int64_t nblocks = cdiv(n > 0 ? n : 1, kSortChunk);
  const size_t h = align_up((size_t)max_bins * nblocks, 4);
  const size_t group_rows = max_bins > 256 ? (size_t)max_bins * (size_t)cdiv(nblocks, 1 << kWideGroupShift) : 0;   // wide passes
  const size_t behind = scan_temp_elems((int64_t)max_bins * nblocks);
  return h + (behind > group_rows ? behind : group_rows);
}

// uint32 words behind the histogram that the wide pass expects ZERO when it starts (its group rows)
static size_t radix_wide_group_elems(int64_t n, int bits) {
  return (size_t)(1 << bits) * (size_t)cdiv(cdiv(n > 0 ? n : 1, kSortChunk), 1 << kWideGroupShift);
}

static int radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, int shift,
                      int bits, uint32_t *temp, hipStream_t st, const uint64_t *n_dev = nullptr) {
  if (n == 0) return BDS_OK;
  const int nblocks = (int)cdiv(n, kSortChunk);
  const uint32_t mask = (1u << bits) - 1u;
  const int64_t hn = (int64_t)(mask + 1) * nblocks;
  uint32_t *hist = temp;
  uint32_t *stemp = temp + align_up((size_t)256 * nblocks, 4);
  hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, nblocks, hist);
  BDS_LAUNCH_CHECK();
  int rc = exclusive_scan_u32(hist, hist, hn, stemp, nullptr, st);
  if (rc != BDS_OK) return rc;
  hipLaunchKernelGGL(radix_scatter_lds_kernel, dim3(nblocks), dim3(kSortBlock), 0, st, kin, vin, n, n_dev, shift, mask, bits,
                     nblocks, hist, kout, vout);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}


static int radix_pass_keys(const uint32_t *kin, uint32_t *kout, int64_t n, int shift, int bits, uint32_t *temp, hipStream_t st,
                           const uint32_t *unpack, uint32_t rank_mask, uint32_t *vout, int32_t *offsets_out = nullptr,
                           int n_offsets = 0, const uint64_t *n_dev = nullptr) {
  if (n == 0) return BDS_OK;
  const int nblocks = (int)cdiv(n, kSortChunk);
  const uint32_t mask = (1u << bits) - 1u;
  const int64_t hn = (int64_t)(mask + 1) * nblocks;
  uint32_t *hist = temp;
  uint32_t *stemp = temp + align_up((size_t)(bits > 8 ? (1 << kWideBits) : 256) * nblocks, 4);
  if (bits > 8) {
    uint32_t *ghist = stemp;   // zeroed by the caller's preceding launch (radix_wide_group_elems words)
    if (bits == 9) {
      hipLaunchKernelGGL((radix_hist_wide_kernel<512>), dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, hist, ghist);
      hipLaunchKernelGGL((radix_scatter_keys_wide_kernel<512>), dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, bits,
                         hist, ghist, kout, unpack, rank_mask, vout, offsets_out, n_offsets);
    } else {
      hipLaunchKernelGGL((radix_hist_wide_kernel<1024>), dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, hist, ghist);
      hipLaunchKernelGGL((radix_scatter_keys_wide_kernel<1024>), dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, bits,
                         hist, ghist, kout, unpack, rank_mask, vout, offsets_out, n_offsets);
    }
    BDS_LAUNCH_CHECK();
    return BDS_OK;
  }
  hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, nblocks, hist);
  BDS_LAUNCH_CHECK();
  int rc = exclusive_scan_u32(hist, hist, hn, stemp, nullptr, st);
  if (rc != BDS_OK) return rc;
  hipLaunchKernelGGL(radix_scatter_keys_kernel, dim3(nblocks), dim3(kSortBlock), 0, st, kin, n, n_dev, shift, mask, bits, nblocks,
                     hist, kout, unpack, rank_mask, vout);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

// ------------------------------------------------------------------------------------------
// short sort: the same stable LSD pass in TWO launches, for inputs whose length lives on the device
// ------------------------------------------------------------------------------------------
// The depth sort of the visible entries is latency-bound (a few hundred thousand keys: every launch costs more
// than the bytes it moves), so the scan of the [digit][workgroup] histogram is folded away: histograms are stored
// workgroup-major, every 32 workgroups also add theirs to a group row (256 atomics per workgroup), and a scatter
// workgroup derives its own bases from <= ng group rows + <= 31 workgroup rows (all L2-resident).
// Chunks of 1024 pairs (4 rounds per wave): a few hundred thousand keys then spread over > 256 workgroups.
// (Measured and dropped: accumulating the NEXT digit's histogram inside the scatter, per output chunk, with global atomics -- the
// upper bytes of fp32 depths take a handful of values, so a whole chunk hits ONE counter: 14 -> 500 us for the third pass.)
constexpr int kGroupShift = 6;                      // 64 workgroups per group row
constexpr int kShortRounds = 4;
constexpr int kShortChunk = kSortBlock * kShortRounds;   // 1024
constexpr int kChunkShift = 10;
static_assert(kShortChunk == (1 << kChunkShift), "chunk shift");
constexpr int64_t kShortSortMax = (int64_t)kShortChunk * 8192;   // <= 128 group rows

struct ShortHistSh { uint32_t h[256]; };
__device__ __forceinline__ void short_hist_block(ShortHistSh &sh, int vb, const uint32_t *keys, int64_t n, int shift,
                                                 uint32_t *hist /*[nblocks][256]*/, uint32_t *ghist /*[ng][256], zeroed*/) {
  const int64_t base = (int64_t)vb * kShortChunk;
  if (base >= n) return;  // the launch is sized for the host-side bound
  sh.h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < kShortRounds; r++) {
    const int64_t i = base + r * kSortBlock + threadIdx.x;
    if (i < n) atomicAdd(&sh.h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t c = sh.h[threadIdx.x];
  hist[(int64_t)vb * 256 + threadIdx.x] = c;
  if (c) atomicAdd(&ghist[(int64_t)(vb >> kGroupShift) * 256 + threadIdx.x], c);
}
// n_cap >= 0: the launch covers n_cap entries only (device-count form: the caller's capacity); a count beyond it is an overflow the
// host will hear about -- the entries are then left alone altogether (a partial sort would hand garbage positions downstream)
__device__ __forceinline__ int64_t bounded_count(const uint64_t *n_dev, int64_t n_cap) {
  const int64_t n = (int64_t)*n_dev;
  return (n_cap >= 0 && n > n_cap) ? 0 : n;
}
__global__ __launch_bounds__(kSortBlock) void short_hist_kernel(const uint32_t *__restrict__ keys,
                                                               const uint64_t *__restrict__ n_dev, int64_t n_cap, int shift,
                                                               uint32_t *__restrict__ hist /*[nblocks][256]*/,
                                                               uint32_t *__restrict__ ghist /*[ng][256], zeroed*/) {
  __shared__ ShortHistSh sh;
  short_hist_block(sh, (int)blockIdx.x, keys, bounded_count(n_dev, n_cap), shift, hist, ghist);
}

// wave-private ranking as in radix_scatter_lds_kernel; bases from the group / workgroup rows
struct ShortScatterSh {
  uint32_t wrun[kSortWaves][256];
  uint32_t lw[kSortBlock / kWave + 1];
};
__device__ __forceinline__ void short_scatter_block(ShortScatterSh &sh, int vb, const uint32_t *keys_in, const uint32_t *vals_in, int64_t n,
                                                    int shift, const uint32_t *hist, const uint32_t *ghist, uint32_t *keys_out,
                                                    uint32_t *vals_out) {
  if ((int64_t)vb * kShortChunk >= n) return;
  auto &wrun = sh.wrun;
  uint32_t *lw = sh.lw;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
#pragma unroll
  for (int w = 0; w < kSortWaves; w++) wrun[w][tid] = 0;
  // digit `tid`: elements of this digit in front of this workgroup, and in the whole input
  const int nb = (int)((n + kShortChunk - 1) >> kChunkShift), ng = (nb + (1 << kGroupShift) - 1) >> kGroupShift;
  const int gb = vb >> kGroupShift;
  uint32_t below = 0, total = 0;
  {
    int g = 0;
    for (; g + 4 <= ng; g += 4) {   // (independent loads first, then the sums: see the row loop below)
      const uint32_t *q = ghist + (int64_t)g * 256 + tid;
      const uint32_t c0 = q[0], c1 = q[256], c2 = q[512], c3 = q[768];
      total += (c0 + c1) + (c2 + c3);
      below += (g < gb ? c0 : 0u) + (g + 1 < gb ? c1 : 0u) + (g + 2 < gb ? c2 : 0u) + (g + 3 < gb ? c3 : 0u);
    }
    for (; g < ng; g++) {
      const uint32_t c = ghist[(int64_t)g * 256 + tid];
      total += c;
      if (g < gb) below += c;
    }
  }
  {
    // up to 63 rows of the workgroup-major histogram: EIGHT independent partial sums, so that eight loads are in flight at a time
    // (a single running sum serialises the L2 latency of every row: that chain was most of this kernel's 14 us)
    const uint32_t *hp = hist + ((int64_t)gb << kGroupShift) * 256 + tid;
    const int nrows = vb - (gb << kGroupShift);
    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
      const uint32_t *q = hp + (int64_t)r * 256;
      const uint32_t a0 = q[0], a1 = q[256], a2 = q[512], a3 = q[768], a4 = q[1024], a5 = q[1280], a6 = q[1536], a7 = q[1792];
      p0 += a0; p1 += a1; p2 += a2; p3 += a3; p4 += a4; p5 += a5; p6 += a6; p7 += a7;
    }
    for (; r < nrows; r++) p0 += hp[(int64_t)r * 256];
    below += ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
  }
  uint32_t all;
  const uint32_t digit_base = block_excl_scan(total, all, lw);   // (contains the barrier that publishes wrun = 0)
  constexpr int kPerWave = kShortChunk / kSortWaves;
  const int64_t wbase = (int64_t)vb * kShortChunk + (int64_t)wv * kPerWave;
  uint32_t k[kShortRounds], v[kShortRounds];
#pragma unroll
  for (int r = 0; r < kShortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    k[r] = 0xFFFFFFFFu; v[r] = 0;
    if (i < n) {
      k[r] = keys_in[i]; v[r] = vals_in[i];
      atomicAdd(&wrun[wv][(k[r] >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  {
    uint32_t base = digit_base + below;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
      const uint32_t c = wrun[w][tid];
      wrun[w][tid] = base;
      base += c;
    }
  }
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kShortRounds; r++) {
    const int64_t i = wbase + r * kWave + lane;
    const bool on = i < n;
    const uint32_t d = (k[r] >> shift) & 255u;
    unsigned long long peers = __ballot(on);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = __popcll(peers & lt);
    uint32_t pos = 0;
    if (on) pos = wrun[wv][d];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (on && rank == 0) wrun[wv][d] = pos + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (on) {
      keys_out[pos + rank] = k[r];
      vals_out[pos + rank] = v[r];
    }
  }
}
__global__ __launch_bounds__(kSortBlock) void short_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, const uint64_t *__restrict__ n_dev, int64_t n_cap,
    int shift, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ ghist, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out) {
  __shared__ ShortScatterSh sh;
  short_scatter_block(sh, (int)blockIdx.x, keys_in, vals_in, bounded_count(n_dev, n_cap), shift, hist, ghist, keys_out, vals_out);
}

// uint32 elements: one workgroup-major histogram + four zero-initialised group tables
static size_t short_sort_elems(int64_t n_bound) {
  const int64_t nb = cdiv(n_bound > 0 ? n_bound : 1, kShortChunk), ng = cdiv(nb, 1 << kGroupShift);
  return (size_t)(nb + 4 * ng) * 256;
}

// ------------------------------------------------------------------------------------------
// intersection kernels
// ------------------------------------------------------------------------------------------
constexpr int kIsectBlock = 256;

// Visible (camera, Gaussian) entries are compacted first (typically ~15 % of C*N): the row loops of
// the counting / emission kernels then run with dense lanes (they were 2-9 % lane-utilised on the
// un-compacted list) and the depth sort shrinks with them.  The visible count stays on the device.
__global__ __launch_bounds__(kIsectBlock) void isect_flag_kernel(int64_t CN, const int32_t *__restrict__ radii,
                                                                uint32_t *__restrict__ flags) {
  const int64_t o = (int64_t)blockIdx.x * kIsectBlock + threadIdx.x;
  if (o < CN) flags[o] = radii[o] > 0 ? 1u : 0u;
}

// keys = fp32 depth bits (depth > 0 for every visible Gaussian: the bits are monotone), vals = cam*N+g (or, in compact mode, the
// position j in the ascending list asc[] of the visible entries, which is written in both modes)
__global__ __launch_bounds__(kIsectBlock) void isect_compact_kernel(int64_t CN, const int32_t *__restrict__ radii,
                                                                   const uint32_t *__restrict__ pos,
                                                                   const float *__restrict__ depths,
                                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                   uint32_t *__restrict__ asc, int compact, int sd) {
  const int64_t o = (int64_t)blockIdx.x * kIsectBlock + threadIdx.x;
  if (o >= CN || radii[o] <= 0) return;
  const uint32_t j = pos[o];
  keys[j] = __float_as_uint(depths[o * sd]);
  vals[j] = compact ? j : (uint32_t)o;   // compact mode: the sort carries the entry's position in the ascending visible list
  asc[j] = (uint32_t)o;
}

// ---- short path (C*N <= kShortSortMax): compaction fused with its scan and with the first histogram --------
// visible_reduce: visible entries per 2048-entry tile; also clears the tables the later launches add into.
__global__ __launch_bounds__(kScanBlock) void visible_reduce_kernel(int64_t CN, const int32_t *__restrict__ radii,
                                                                   uint32_t *__restrict__ tile_sums,
                                                                   uint32_t *__restrict__ zero_me, int64_t zero_elems,
                                                                   uint64_t *__restrict__ m_total, int32_t *__restrict__ zero_cn) {
  __shared__ uint32_t lw[kScanBlock / kWave + 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) *m_total = 0;   // M is accumulated by the counting kernels
  for (int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x; i < zero_elems; i += (int64_t)gridDim.x * kScanBlock)
    zero_me[i] = 0u;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < CN) {
      s += radii[base + i] > 0 ? 1u : 0u;
      if (zero_cn) zero_cn[base + i] = 0;   // tiles_per_gauss: the counting kernel writes the visible entries only
    }
  }
  uint32_t tot;
  block_excl_scan(s, tot, lw);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// visible_compact: (depth bits, cam*N+g) of the visible entries in index order; the histogram of the first
// sort digit is accumulated on the way (a tile's outputs fall into at most kSpan sort chunks).
constexpr int kCompactSpan = kScanTile / kShortChunk + 1;   // sort chunks a tile's outputs can straddle
struct VisCompactSh {
  uint32_t lw[kScanBlock / kWave + 1];
  uint32_t h[kCompactSpan][256];
};
__device__ __forceinline__ void visible_compact_block(VisCompactSh &sh, int vb, int nvb, int64_t CN, const int32_t *radii, const float *depths,
                                                      const uint32_t *tile_sums, uint32_t *keys, uint32_t *vals, uint32_t *hist,
                                                      uint32_t *ghist, uint64_t *n_vis_out, uint32_t *asc, int compact, int sums_per_tile,
                                                      int sd) {
  constexpr int kSpan = kCompactSpan;
  uint32_t *lw = sh.lw;
  auto &h = sh.h;
#pragma unroll
  for (int t = 0; t < kSpan; t++) h[t][threadIdx.x] = 0;
  uint32_t part = 0;
  // (sums_per_tile = 8: the counts were left per 256-Gaussian workgroup by the one-view projection, bds_project_view_prepare_fwd)
  for (int b = threadIdx.x; b < vb * sums_per_tile; b += kScanBlock) part += tile_sums[b];
  uint32_t my_offset;
  block_excl_scan(part, my_offset, lw);   // total of the partial sums = this tile's offset
  const int64_t base = (int64_t)vb * kScanTile + (int64_t)threadIdx.x * kScanItems;
  bool vis[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    vis[i] = base + i < CN && radii[base + i] > 0;
    s += vis[i] ? 1u : 0u;
  }
  uint32_t tot;
  uint32_t j = block_excl_scan(s, tot, lw) + my_offset;
  const uint32_t chunk0 = my_offset >> kChunkShift;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (vis[i]) {
      const uint32_t key = __float_as_uint(depths[(base + i) * sd]);
      keys[j] = key;
      vals[j] = compact ? j : (uint32_t)(base + i);
      asc[j] = (uint32_t)(base + i);
      atomicAdd(&h[(j >> kChunkShift) - chunk0][key & 255u], 1u);
      j++;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kSpan; t++) {
    const uint32_t c = h[t][threadIdx.x];
    if (c) {
      atomicAdd(&hist[(int64_t)(chunk0 + t) * 256 + threadIdx.x], c);
      atomicAdd(&ghist[(int64_t)((chunk0 + t) >> kGroupShift) * 256 + threadIdx.x], c);
    }
  }
  if (vb == nvb - 1 && threadIdx.x == 0) *n_vis_out = (uint64_t)my_offset + tot;
}
__global__ __launch_bounds__(kScanBlock) void visible_compact_kernel(int64_t CN, const int32_t *__restrict__ radii,
                                                                    const float *__restrict__ depths,
                                                                    const uint32_t *__restrict__ tile_sums,
                                                                    uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                    uint32_t *__restrict__ hist, uint32_t *__restrict__ ghist,
                                                                    uint64_t *__restrict__ n_vis_out, uint32_t *__restrict__ asc,
                                                                    int compact, int sums_per_tile, int sd) {
  __shared__ VisCompactSh sh;
  visible_compact_block(sh, (int)blockIdx.x, (int)gridDim.x, CN, radii, depths, tile_sums, keys, vals, hist, ghist, n_vis_out, asc, compact,
                        sums_per_tile, sd);
}

// ---- row-parallel counting / emission ---------------------------------------------------------------------
// One thread per Gaussian leaves most lanes idle: the tile rectangles of 64 depth-neighbours differ by an order of
// magnitude in size and the wave runs as long as its largest member (rows x tiles, serially).  Here a workgroup still
// owns 256 consecutive (depth-ordered) Gaussians -- so its output range is contiguous and starts at the scanned
// offset of its first member -- but the unit of work is one tile ROW of one Gaussian: the members' records are parked
// in LDS, a prefix over their row counts maps a row item back to its owner (8-step search in LDS), and for the
// emission a workgroup-wide running prefix over the rows' tile counts gives every row its output offset.
struct RowStage {
  float mx[kIsectBlock], my[kIsectBlock], a[kIsectBlock], b[kIsectBlock], c[kIsectBlock], qmax[kIsectBlock];
  int x0[kIsectBlock], x1[kIsectBlock], y0[kIsectBlock];
  uint32_t id[kIsectBlock];          // list value of the member: cam*N + gaussian, or its compact position
  uint32_t cam_base[kIsectBlock];    // camera * tiles-per-camera
  uint32_t rowoff[kIsectBlock + 1];  // exclusive prefix of the members' row counts
  uint32_t lw[kIsectBlock / kWave + 1];
};

// Per-member record in depth order (12 words): the counting kernel gathers a member's attributes from the four
// per-Gaussian arrays ONCE (4 scattered cache lines per member) and leaves the derived rectangle here; the emission
// reads it back coalesced instead of gathering again, and is guaranteed to take the same decisions.
constexpr int kRecWords = 12;

// fills the stage for members j0 .. j0+255 of the depth order; returns the workgroup's number of row items
// rec_out != null: compute from the attribute arrays and store the records; rec_in != null: load the records.
__device__ __forceinline__ uint32_t stage_rows(RowStage &S, int64_t j0, int64_t n_vis, int64_t N, const uint32_t *__restrict__ sorted_idx,
                                               const uint32_t *__restrict__ asc /* compact mode: sorted_idx holds positions in asc[] */,
                                               const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                                               const float *__restrict__ conics, const float *__restrict__ opacities,
                                               int tile_size, int tile_w, int tile_h, float4 *__restrict__ rec_out,
                                               const float4 *__restrict__ rec_in, const ProjLayout pl) {
  const int t = threadIdx.x;
  const int64_t j = j0 + t;
  uint32_t nrows = 0;
  if (j < n_vis) {
    if (rec_in != nullptr) {
      const float4 r0 = rec_in[j * 3], r1 = rec_in[j * 3 + 1], r2 = rec_in[j * 3 + 2];
      nrows = __float_as_uint(r2.y);
      S.mx[t] = r0.x; S.my[t] = r0.y; S.a[t] = r0.z; S.b[t] = r0.w;
      S.c[t] = r1.x; S.qmax[t] = r1.y; S.x0[t] = __float_as_int(r1.z); S.x1[t] = __float_as_int(r1.w);
      S.y0[t] = __float_as_int(r2.x); S.id[t] = __float_as_uint(r2.z); S.cam_base[t] = __float_as_uint(r2.w);
    } else {
      const uint32_t val = sorted_idx[j];            // what the lists will carry for this entry: its id, or its compact position
      const uint32_t o = asc ? asc[val] : val;
      // every gather of this member is issued before the first one is consumed (the entries of the list ARE visible: testing the
      // radius first would only put one more memory latency in front of the others)
      const int r = proj_radius(pl, means2d, radii, (int64_t)o);
      float mx = means2d[(int64_t)o * pl.s2], my = means2d[(int64_t)o * pl.s2 + 1];
      float a = 0.f, b = 0.f, c = 0.f, q_max = 0.f, op = 0.f;
      if (conics != nullptr) {
        a = conics[(int64_t)o * pl.sc]; b = conics[(int64_t)o * pl.sc + 1]; c = conics[(int64_t)o * pl.sc + 2];
        op = opacities[(int64_t)o * pl.so];
      }
      int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
      if (r > 0) {
        bool ok;
        if (conics == nullptr) {
          tile_rect(mx, my, r, tile_size, tile_w, tile_h, x0, y0, x1, y1);
          ok = x1 > x0 && y1 > y0;
        } else {
          ok = tile_rect_tight(mx, my, r, a, b, c, op, tile_size, tile_w, tile_h, x0, y0, x1, y1, q_max);
        }
        if (ok) nrows = (uint32_t)(y1 - y0);
      } else {
        mx = my = a = b = c = 0.f;
      }
      // camera of entry o = o / N: entries of the first camera (the only one on the per-view path) need no division
      const uint32_t cam_base = (int64_t)o < N ? 0u : (uint32_t)((uint32_t)o / (uint32_t)N) * (uint32_t)(tile_w * tile_h);
      S.mx[t] = mx; S.my[t] = my; S.a[t] = a; S.b[t] = b; S.c[t] = c; S.qmax[t] = q_max;
      S.x0[t] = x0; S.x1[t] = x1; S.y0[t] = y0; S.id[t] = val; S.cam_base[t] = cam_base;
      if (rec_out != nullptr) {
        rec_out[j * 3] = make_float4(mx, my, a, b);
        rec_out[j * 3 + 1] = make_float4(c, q_max, __int_as_float(x0), __int_as_float(x1));
        rec_out[j * 3 + 2] = make_float4(__int_as_float(y0), __uint_as_float(nrows), __uint_as_float(val), __uint_as_float(cam_base));
      }
    }
  }
  uint32_t total;
  const uint32_t ex = block_excl_scan(nrows, total, S.lw);
  S.rowoff[t] = ex;
  if (t == 0) S.rowoff[kIsectBlock] = total;
  __syncthreads();
  return total;
}

// owner of row item r (largest g with rowoff[g] <= r; members without rows are skipped by construction)
__device__ __forceinline__ int row_owner(const RowStage &S, uint32_t r) {
  int lo = 0, hi = kIsectBlock;   // invariant: rowoff[lo] <= r < rowoff[hi]
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const int mid = (lo + hi) >> 1;
    if (S.rowoff[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void row_span_of(const RowStage &S, int g, int ty, bool cull, int tile_size, int &lo, int &hi) {
  lo = S.x0[g]; hi = S.x1[g];
  if (cull) row_tile_span(S.mx[g], S.my[g], S.a[g], S.b[g], S.c[g], S.qmax[g], ty, tile_size, S.x0[g], S.x1[g], lo, hi);
}

struct CountRowsSh {
  RowStage S;
  uint32_t cnt[kIsectBlock];
};
__device__ __forceinline__ void isect_count_rows_block(CountRowsSh &sh, int vb, int64_t n_vis, const uint32_t *sorted_idx, const float *means2d,
                                                       const int32_t *radii, const float *conics, const float *opacities, int tile_size,
                                                       int tile_w, int tile_h, int32_t *tiles_per_gauss, int64_t N, float4 *rec,
                                                       uint32_t *btot, const uint32_t *asc, const ProjLayout pl) {
  RowStage &S = sh.S;
  uint32_t *cnt = sh.cnt;
  const int64_t j0 = (int64_t)vb * kIsectBlock, j = j0 + threadIdx.x;
  if (j0 >= n_vis) {   // (groups beyond the visible count: the launch is sized for a host-side bound)
    if (threadIdx.x == 0) btot[vb] = 0u;
    return;
  }
  cnt[threadIdx.x] = 0u;
  const uint32_t R = stage_rows(S, j0, n_vis, N, sorted_idx, asc, means2d, radii, conics, opacities, tile_size, tile_w, tile_h, rec, nullptr, pl);
  const bool cull = conics != nullptr;
  for (uint32_t r = threadIdx.x; r < R; r += kIsectBlock) {
    const int g = row_owner(S, r);
    int lo, hi;
    row_span_of(S, g, S.y0[g] + (int)(r - S.rowoff[g]), cull, tile_size, lo, hi);
    if (hi > lo) atomicAdd(&cnt[g], (uint32_t)(hi - lo));
  }
  __syncthreads();
  if (tiles_per_gauss && j < n_vis) tiles_per_gauss[asc ? asc[sorted_idx[j]] : sorted_idx[j]] = (int32_t)cnt[threadIdx.x];
  uint32_t total;
  block_excl_scan(cnt[threadIdx.x], total, S.lw);
  // (the grand total M is formed from btot[] by finish_counts_kernel: one atomic per workgroup on ONE address serialises in L2,
  //  ~7 ns each, at the tail of a kernel that runs a single round of workgroups)
  if (threadIdx.x == 0) btot[vb] = total;
}
__global__ __launch_bounds__(kIsectBlock) void isect_count_rows_kernel(
    const uint64_t *__restrict__ n_vis_dev, int64_t n_cap, const uint32_t *__restrict__ sorted_idx, const float *__restrict__ means2d,
    const int32_t *__restrict__ radii, const float *__restrict__ conics, const float *__restrict__ opacities, int tile_size,
    int tile_w, int tile_h, int32_t *__restrict__ tiles_per_gauss, int64_t N, float4 *__restrict__ rec, uint32_t *__restrict__ btot,
    const uint32_t *__restrict__ asc, const ProjLayout pl) {
  __shared__ CountRowsSh sh;
  isect_count_rows_block(sh, (int)blockIdx.x, bounded_count(n_vis_dev, n_cap), sorted_idx, means2d, radii, conics, opacities, tile_size, tile_w, tile_h,
                         tiles_per_gauss, N, rec, btot, asc, pl);
}

struct EmitRowsSh { RowStage S; };
__device__ __forceinline__ void isect_emit_rows_block(EmitRowsSh &sh, int vb, int64_t n_vis, int64_t N, const uint32_t *sorted_idx,
                                                      const uint32_t *btot, const float *means2d, const int32_t *radii, const float *conics,
                                                      const float *opacities, int tile_size, int tile_w, int tile_h, uint32_t *keys,
                                                      uint32_t *vals, int pack_shift, const float4 *rec) {
  RowStage &S = sh.S;
  const int64_t j0 = (int64_t)vb * kIsectBlock;
  if (j0 >= n_vis) return;
  const uint32_t R = stage_rows(S, j0, n_vis, N, sorted_idx, nullptr, means2d, radii, conics, opacities, tile_size, tile_w, tile_h, nullptr, rec,
                                ProjLayout{2, 1, 3, 1, 0});   // (records only: nothing is gathered here)
  const bool cull = conics != nullptr;
  // output offset of the workgroup's first row = intersections of all groups in front of it (no scan launch: <= a few
  // thousand L2-resident totals are summed here)
  uint32_t carry;
  {
    uint32_t part = 0;
    for (int b = threadIdx.x; b < vb; b += kIsectBlock) part += btot[b];
    block_excl_scan(part, carry, S.lw);
  }
  for (uint32_t base = 0; base < R; base += kIsectBlock) {   // uniform trip count: the scan below has barriers
    const uint32_t r = base + threadIdx.x;
    int lo = 0, hi = 0;
    uint32_t id = 0, key0 = 0;
    if (r < R) {
      const int g = row_owner(S, r);
      const int ty = S.y0[g] + (int)(r - S.rowoff[g]);
      row_span_of(S, g, ty, cull, tile_size, lo, hi);
      id = S.id[g];
      key0 = S.cam_base[g] + (uint32_t)(ty * tile_w);
      if (pack_shift) id = (uint32_t)(j0 + g);   // packed lists carry the depth rank; the list value is looked up when the last pass writes out
    }
    const uint32_t c = hi > lo ? (uint32_t)(hi - lo) : 0u;
    uint32_t total;
    uint32_t off = carry + block_excl_scan(c, total, S.lw);
    carry += total;
    if (pack_shift) {
      for (int tx = lo; tx < hi; tx++) keys[off++] = ((key0 + (uint32_t)tx) << pack_shift) | id;
    } else {
      for (int tx = lo; tx < hi; tx++) {
        keys[off] = key0 + (uint32_t)tx;
        vals[off] = id;
        off++;
      }
    }
  }
}
__global__ __launch_bounds__(kIsectBlock) void isect_emit_rows_kernel(
    const uint64_t *__restrict__ n_vis_dev, int64_t N, const uint32_t *__restrict__ sorted_idx, const uint32_t *__restrict__ btot,
    const float *__restrict__ means2d, const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ opacities, int tile_size, int tile_w, int tile_h, uint32_t *__restrict__ keys,
    uint32_t *__restrict__ vals, int pack_shift, const float4 *__restrict__ rec, uint32_t *__restrict__ zero_words, int n_zero) {
  __shared__ EmitRowsSh sh;
  // (the group rows of the tile pass that follows: cleared here, by every workgroup of the launch a word each, before anything returns)
  for (int64_t i = (int64_t)blockIdx.x * kIsectBlock + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * kIsectBlock) zero_words[i] = 0u;
  isect_emit_rows_block(sh, (int)blockIdx.x, (int64_t)*n_vis_dev, N, sorted_idx, btot, means2d, radii, conics, opacities, tile_size, tile_w,
                        tile_h, keys, vals, pack_shift, rec);
}


// offsets[t] = first index whose key >= t  (lower bound; empty tiles point at the next run)
// (key_shift: the tile key sits above the rank bits of a packed entry)
__global__ __launch_bounds__(kIsectBlock) void isect_offsets_kernel(int64_t M_host, const uint64_t *__restrict__ M_dev,
                                                                   const uint32_t *__restrict__ keys, int key_shift,
                                                                   int n_tiles_total, int32_t *__restrict__ offsets) {
  const int64_t M = list_length(M_host, M_dev);
  const int64_t i = (int64_t)blockIdx.x * kIsectBlock + threadIdx.x;
  if (i > M) return;
  // thread i < M closes the gap (key[i-1], key[i]]; thread M closes (key[M-1], n_tiles)
  const int64_t lo = i == 0 ? 0 : (int64_t)(keys[i - 1] >> key_shift) + 1;
  const int64_t hi = i == M ? (int64_t)n_tiles_total - 1 : (int64_t)(keys[i] >> key_shift);
  for (int64_t t = lo; t <= hi; t++) offsets[t] = (int32_t)i;
}

__global__ __launch_bounds__(kIsectBlock) void isect_ids_kernel(int64_t M, const uint32_t *__restrict__ keys, int key_shift,
                                                               const int32_t *__restrict__ flatten_ids,
                                                               const float *__restrict__ depths,
                                                               int64_t *__restrict__ isect_ids, int sd) {
  const int64_t i = (int64_t)blockIdx.x * kIsectBlock + threadIdx.x;
  if (i >= M) return;
  const uint32_t d = __float_as_uint(depths[(int64_t)flatten_ids[i] * sd]);
  isect_ids[i] = (int64_t)(((uint64_t)(keys[i] >> key_shift) << 32) | (uint64_t)d);
}

// ------------------------------------------------------------------------------------------
// workspace layouts
// ------------------------------------------------------------------------------------------
struct PrepWs {
  uint64_t *total;      // [0] = M (intersections), [1] = visible (camera, Gaussian) entries
  uint32_t *ka, *va, *kb, *vb;  // [CN] each; after prepare: sorted ids live in `sorted`
  uint32_t *cum;        // [CN] exclusive scan of counts in depth order
  uint32_t *asc;        // [CN] ids cam*N+g of the visible entries in ascending order (compact position -> id)
  uint32_t *temp;       // radix / scan temp
  uint32_t *tables;     // short path: workgroup-major histogram + 4 group tables
  float4 *rec;          // [CN][3] per-member records in depth order, written by EVERY counting kernel, read by the row emission
  uint32_t *btot;       // [cdiv(CN,256)] intersections of each 256-member group of the depth order (every counting kernel)
  size_t bytes;
};

static PrepWs prep_layout(void *ws, int64_t CN) {
  PrepWs L;
  char *p = static_cast<char *>(ws);
  size_t off = 0;
  auto take = [&](size_t elems, size_t esz) {
    char *q = p ? p + off : nullptr;
    off += align_up(elems * esz, 256);
    return q;
  };
  L.total = reinterpret_cast<uint64_t *>(take(8, 8));   // M, visible, and the effective / overflow words of the device-count form
  L.ka = reinterpret_cast<uint32_t *>(take(CN, 4));
  L.va = reinterpret_cast<uint32_t *>(take(CN, 4));
  L.kb = reinterpret_cast<uint32_t *>(take(CN, 4));
  L.vb = reinterpret_cast<uint32_t *>(take(CN, 4));
  L.cum = reinterpret_cast<uint32_t *>(take(CN, 4));
  L.asc = reinterpret_cast<uint32_t *>(take(CN, 4));
  size_t t = radix_temp_elems(CN);
  size_t t2 = scan_temp_elems(CN);
  L.temp = reinterpret_cast<uint32_t *>(take(t > t2 ? t : t2, 4));
  L.tables = reinterpret_cast<uint32_t *>(take(CN <= kShortSortMax ? short_sort_elems(CN) : 0, 4));
  L.rec = reinterpret_cast<float4 *>(take((size_t)CN * kRecWords, 4));
  L.btot = reinterpret_cast<uint32_t *>(take((size_t)cdiv(CN > 0 ? CN : 1, kIsectBlock) + 1, 4));
  L.bytes = off;
  return L;
}

struct BuildWs {
  uint32_t *ka, *va, *kb;  // [M] each (the 4th buffer is flatten_ids itself)
  uint32_t *temp;
  size_t bytes;
};

static BuildWs build_layout(void *ws, int64_t M) {
  BuildWs L;
  char *p = static_cast<char *>(ws);
  size_t off = 0;
  auto take = [&](size_t elems, size_t esz) {
    char *q = p ? p + off : nullptr;
    off += align_up(elems * esz, 256);
    return q;
  };
  L.ka = reinterpret_cast<uint32_t *>(take(M, 4));
  L.va = reinterpret_cast<uint32_t *>(take(M, 4));
  L.kb = reinterpret_cast<uint32_t *>(take(M, 4));
  L.temp = reinterpret_cast<uint32_t *>(take(radix_temp_elems(M, 1 << kWideBits), 4));
  L.bytes = off + 256;
  return L;
}

}  // namespace bds

using namespace bds;

int bds::prep_reduce_slots(void *ws, size_t ws_bytes, int64_t CN, PrepReduceSlots *out) {
  BDS_REQUIRE(ws && out && CN > 0 && CN < (int64_t)1 << 31);
  if (!(CN <= kShortSortMax && option_get(kOptShortSort))) return BDS_ECAPACITY;
  PrepWs L = prep_layout(ws, CN);
  if (ws_bytes < L.bytes) return BDS_EWORKSPACE;
  out->sums256 = L.temp;
  out->zero_me = L.tables;
  out->zero_elems = (int64_t)short_sort_elems(CN);
  out->m_total = L.total;
  return BDS_OK;
}

extern "C" size_t bds_isect_visible_ids_offset(int C, int64_t N) {
  if (C < 1 || N < 0) return 0;
  char *const base = reinterpret_cast<char *>(static_cast<uintptr_t>(4096));   // layout arithmetic only, never dereferenced
  const PrepWs L = prep_layout(base, (int64_t)C * N);
  return static_cast<size_t>(reinterpret_cast<char *>(L.asc) - base);
}

extern "C" size_t bds_isect_prepare_workspace_bytes(int C, int64_t N) {
  if (C < 1 || N < 0) return 0;
  return prep_layout(nullptr, (int64_t)C * N).bytes;
}

extern "C" size_t bds_isect_build_workspace_bytes(int C, int64_t N, int64_t M) {
  (void)C; (void)N;
  if (M < 0) return 0;
  return build_layout(nullptr, M).bytes;
}

// M = sum of the counting kernel's per-workgroup totals -> counts_dev[0]; then, for the asynchronous form, both counts -> page-locked
// host memory, written by the GPU itself (a copy node between two kernels costs ~15 us of idle GPU: engine switch + barriers; this
// one-workgroup kernel ~4.5 us)
// cap_m / cap_vis >= 0 (the `_dev` forms): the "effective" counts downstream kernels size themselves by are published next to the raw
// ones -- both ZERO when either count outgrew its capacity (the view then renders nothing instead of overrunning a buffer; the host
// sees the raw counts and the overflow word one read-back later and provisions more).
constexpr int kCountMEff = 2, kCountVisEff = 3, kCountOverflow = 4;
__global__ __launch_bounds__(256) void finish_counts_kernel(const uint32_t *__restrict__ btot, int nblocks, uint64_t *__restrict__ counts_dev,
                                                           volatile int64_t *__restrict__ counts_host, int64_t cap_m, int64_t cap_vis) {
  __shared__ unsigned long long part[256];
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;   // (independent partial sums: four loads in flight per thread)
  int b = threadIdx.x;
  for (; b + 768 < nblocks; b += 1024) {
    const uint32_t a0 = btot[b], a1 = btot[b + 256], a2 = btot[b + 512], a3 = btot[b + 768];
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; b < nblocks; b += 256) s0 += btot[b];
  part[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts_dev[0] = part[0];
    const uint64_t nv = counts_dev[1];
    if (cap_m >= 0) {
      const bool over = (int64_t)part[0] > cap_m || (int64_t)nv > cap_vis;
      counts_dev[kCountMEff] = over ? 0ull : part[0];
      counts_dev[kCountVisEff] = over ? 0ull : nv;
      counts_dev[kCountOverflow] = over ? 1ull : 0ull;
      if (counts_host && over) counts_host[2] = 1;   // sticky: the host clears it when it has provisioned more
    }
    if (counts_host) { counts_host[0] = (int64_t)part[0]; counts_host[1] = (int64_t)nv; }
  }
  if (counts_host) __threadfence_system();
}

// enqueues the whole prepare stage; the counts (M, visible entries) end up in L.total on the device
static int prepare_enqueue(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                           const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                           int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, int compact, bds_stream_t stream,
                           uint64_t **counts_dev, const uint32_t **btot_out, int *nblocks_out, int64_t nvis_bound = -1) {
  BDS_REQUIRE(C >= 1 && N >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0);
  const int64_t CN = (int64_t)C * N;
  BDS_REQUIRE(CN < (int64_t)1 << 31);
  BDS_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)1 << 31);
  *counts_dev = nullptr;
  *btot_out = nullptr; *nblocks_out = 0;
  if (CN == 0) return BDS_OK;
  BDS_REQUIRE(means2d && radii && depths && ws);
  BDS_REQUIRE((conics == nullptr) == (opacities == nullptr));
  const ProjLayout pl = proj_layout(means2d, depths, conics, opacities);   // (the [N,8] row form is recognised by the addresses)
  PrepWs L = prep_layout(ws, CN);
  if (ws_bytes < L.bytes) return BDS_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  // nvis_bound (device-count form: the caller's capacity for the visible entries): everything behind the compaction is launched for
  // that many entries instead of for all C*N -- at 15 % visibility four of five workgroups of those launches had nothing to do, and
  // next to another stream's compositor every workgroup, idle or not, waits for wave slots (a count beyond the bound is an overflow:
  // the view renders nothing, whatever these launches leave behind)
  if (nvis_bound < 0 || nvis_bound > CN || !option_get(kOptCapLaunch)) nvis_bound = CN;
  const int64_t n_cap = nvis_bound < CN ? nvis_bound : (int64_t)-1;
  const unsigned grid = (unsigned)cdiv(nvis_bound > 0 ? nvis_bound : 1, kIsectBlock);
  uint64_t *n_vis = L.total + 1;
  int rc;
  if (CN <= kShortSortMax && option_get(kOptShortSort)) {
    // 12 launches instead of 27: the whole stage is launch-latency bound at this size
    const int64_t nb = cdiv(CN, kShortChunk), ng = cdiv(nb, 1 << kGroupShift);
    const unsigned nb_launch = (unsigned)cdiv(nvis_bound > 0 ? nvis_bound : 1, kShortChunk);
    uint32_t *hist = L.tables, *ghist = L.tables + nb * 256;   // ghist[p] = ghist + p * ng * 256
    const unsigned tiles = (unsigned)cdiv(CN, kScanTile);
    // 1. visible entries -> (depth key, id) pairs in index order + histogram of the first digit
    // (compact & 2: the one-view projection has already left the visible counts -- per 256 Gaussians -- and cleared the tables)
    if (!(compact & 2))
      hipLaunchKernelGGL(visible_reduce_kernel, dim3(tiles), dim3(kScanBlock), 0, st, CN, radii, L.temp, L.tables,
                         (int64_t)short_sort_elems(CN), L.total, tiles_per_gauss);
    hipLaunchKernelGGL(visible_compact_kernel, dim3(tiles), dim3(kScanBlock), 0, st, CN, radii, depths, L.temp, L.ka, L.va, hist,
                       ghist, n_vis, L.asc, compact & 1, (compact & 2) ? kScanTile / 256 : 1, pl.sd);
    BDS_LAUNCH_CHECK();
    // 2. depth order: 4 stable passes of 8 bits; ends in (ka, va)
    uint32_t *kin = L.ka, *vin = L.va, *kout = L.kb, *vout = L.vb;
    for (int p = 0; p < 4; p++) {
      uint32_t *gh = ghist + (int64_t)p * ng * 256;
      if (p > 0) hipLaunchKernelGGL(short_hist_kernel, dim3(nb_launch), dim3(kSortBlock), 0, st, kin, n_vis, n_cap, 8 * p, hist, gh);
      hipLaunchKernelGGL(short_scatter_kernel, dim3(nb_launch), dim3(kSortBlock), 0, st, kin, vin, n_vis, n_cap, 8 * p, hist, gh, kout,
                         vout);
      uint32_t *t;
      t = kin; kin = kout; kout = t;
      t = vin; vin = vout; vout = t;
    }
    BDS_LAUNCH_CHECK();
    // 3. tiles per entry, in depth order (tiles_per_gauss was zeroed by visible_reduce_kernel)
    hipLaunchKernelGGL(isect_count_rows_kernel, dim3(grid), dim3(kIsectBlock), 0, st, n_vis, n_cap, L.va, means2d, radii, conics,
                       opacities, tile_size, tile_w, tile_h, tiles_per_gauss, N, L.rec, L.btot,
                       (compact & 1) ? L.asc : (const uint32_t *)nullptr, pl);
    BDS_LAUNCH_CHECK();
  } else {
    BDS_REQUIRE(!(compact & 2));   // (the pre-reduced form exists for the short path only)
    if (hipMemsetAsync(L.total, 0, sizeof(uint64_t), st) != hipSuccess) return BDS_ELAUNCH;   // M is accumulated by the counting kernel
    // 1. compact the visible entries: flags -> exclusive scan -> (depth key, id) pairs, count stays on the device
    hipLaunchKernelGGL(isect_flag_kernel, dim3(grid), dim3(kIsectBlock), 0, st, CN, radii, L.kb);
    BDS_LAUNCH_CHECK();
    rc = exclusive_scan_u32(L.kb, L.cum, CN, L.temp, n_vis, st);
    if (rc != BDS_OK) return rc;
    hipLaunchKernelGGL(isect_compact_kernel, dim3(grid), dim3(kIsectBlock), 0, st, CN, radii, L.cum, depths, L.ka, L.va, L.asc, compact, pl.sd);
    BDS_LAUNCH_CHECK();
    // 2. depth order: 4 stable passes of 8 bits over the visible entries; ends in (ka, va)
    uint32_t *kin = L.ka, *vin = L.va, *kout = L.kb, *vout = L.vb;
    for (int p = 0; p < 4; p++) {
      rc = radix_pass(kin, vin, kout, vout, CN, 8 * p, 8, L.temp, st, n_vis);
      if (rc != BDS_OK) return rc;
      uint32_t *t;
      t = kin; kin = kout; kout = t;
      t = vin; vin = vout; vout = t;
    }
    // 3. tiles per entry, in depth order
    if (tiles_per_gauss && hipMemsetAsync(tiles_per_gauss, 0, sizeof(int32_t) * CN, st) != hipSuccess) return BDS_ELAUNCH;
    hipLaunchKernelGGL(isect_count_rows_kernel, dim3(grid), dim3(kIsectBlock), 0, st, n_vis, (int64_t)-1, L.va, means2d, radii, conics,
                       opacities, tile_size, tile_w, tile_h, tiles_per_gauss, N, L.rec, L.btot,
                       compact ? L.asc : (const uint32_t *)nullptr, pl);
    BDS_LAUNCH_CHECK();
  }
  // 4. (no scan: every counting kernel leaves the per-256-member totals and adds them to M; the emission derives its offsets)
  (void)rc;
  *counts_dev = L.total;
  *btot_out = L.btot; *nblocks_out = (int)grid;
  return BDS_OK;
}

extern "C" int bds_isect_prepare(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                                 const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                                 int32_t *tiles_per_gauss, void *ws,
                                 size_t ws_bytes, int64_t *n_isects, int64_t *n_visible, int compact, bds_stream_t stream) {
  BDS_REQUIRE(n_isects);
  *n_isects = 0;
  if (n_visible) *n_visible = 0;
  uint64_t *counts_dev = nullptr;
  const uint32_t *btot = nullptr;
  int nblocks = 0;
  int rc = prepare_enqueue(C, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, tiles_per_gauss, ws, ws_bytes,
                           compact, stream, &counts_dev, &btot, &nblocks);
  if (rc != BDS_OK || counts_dev == nullptr) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(finish_counts_kernel, dim3(1), dim3(256), 0, st, btot, nblocks, counts_dev, static_cast<volatile int64_t *>(nullptr),
                     (int64_t)-1, (int64_t)-1);
  BDS_LAUNCH_CHECK();
  uint64_t total[2] = {0, 0};   // M, visible entries
  if (hipMemcpyAsync(total, counts_dev, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return BDS_ELAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return BDS_ELAUNCH;
  *n_isects = (int64_t)total[0];
  if (n_visible) *n_visible = (int64_t)total[1];
  return BDS_OK;
}

extern "C" int bds_isect_prepare_async(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                                       const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                                       int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, int64_t *counts_pinned,
                                       void *event, int compact, bds_stream_t stream) {
  BDS_REQUIRE(counts_pinned && event);
  uint64_t *counts_dev = nullptr;
  const uint32_t *btot = nullptr;
  int nblocks = 0;
  int rc = prepare_enqueue(C, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, tiles_per_gauss, ws, ws_bytes,
                           compact, stream, &counts_dev, &btot, &nblocks);
  if (rc != BDS_OK) return rc;
  hipStream_t st = as_stream(stream);
  if (counts_dev == nullptr) {
    counts_pinned[0] = 0; counts_pinned[1] = 0;
  } else {
    void *mapped = nullptr;   // device view of the caller's page-locked buffer (the same address under unified addressing)
    if (hipHostGetDevicePointer(&mapped, counts_pinned, 0) != hipSuccess) { (void)hipGetLastError(); mapped = nullptr; }
    hipLaunchKernelGGL(finish_counts_kernel, dim3(1), dim3(256), 0, st, btot, nblocks, counts_dev, static_cast<volatile int64_t *>(mapped),
                       (int64_t)-1, (int64_t)-1);
    BDS_LAUNCH_CHECK();
    if (mapped == nullptr &&   // not mapped: the copy engine
        hipMemcpyAsync(counts_pinned, counts_dev, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess)
      return BDS_ELAUNCH;
  }
  if (hipEventRecord(static_cast<hipEvent_t>(event), st) != hipSuccess) return BDS_ELAUNCH;
  return BDS_OK;
}

// Device-count form of the prepare stage: nothing is read back.  The counts stay in the first words of the workspace --
// {M, visible, M effective, visible effective, overflow} as uint64 -- where the `_dev` entry points of the later stages read them
// (bds_isect_counts_offset(which)); `counts_pinned` (optional, page-locked int64[3]: M, visible, overflow) is written by the GPU
// itself for the host to look at whenever it likes (e.g. one frame later).  Capturable in a hipGraph.
extern "C" int bds_isect_prepare_dev(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                                     const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                                     int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, int64_t M_capacity,
                                     int64_t n_visible_capacity, int64_t *counts_pinned, int compact, bds_stream_t stream) {
  BDS_REQUIRE(M_capacity >= 0 && n_visible_capacity >= 0 && (int64_t)C * N > 0);
  uint64_t *counts_dev = nullptr;
  const uint32_t *btot = nullptr;
  int nblocks = 0;
  int rc = prepare_enqueue(C, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, tiles_per_gauss, ws, ws_bytes,
                           compact, stream, &counts_dev, &btot, &nblocks, n_visible_capacity);
  if (rc != BDS_OK) return rc;
  void *mapped = nullptr;
  if (counts_pinned && hipHostGetDevicePointer(&mapped, counts_pinned, 0) != hipSuccess) { (void)hipGetLastError(); return BDS_EINVAL; }
  hipLaunchKernelGGL(finish_counts_kernel, dim3(1), dim3(256), 0, as_stream(stream), btot, nblocks, counts_dev,
                     static_cast<volatile int64_t *>(mapped), M_capacity, n_visible_capacity);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" size_t bds_isect_counts_offset(int which) {
  // byte offset, inside the prepare workspace, of: 0 = M, 1 = visible entries, 2 = M effective, 3 = visible effective, 4 = overflow
  return (which >= 0 && which <= kCountOverflow) ? (size_t)which * sizeof(uint64_t) : (size_t)0;
}

// dev = true: M / n_visible are CAPACITIES (launch sizes, buffer sizes); the actual counts are read on the device from the
// "effective" slots of the prepare workspace (kCountMEff / kCountVisEff: zero when a count outgrew its capacity)
static int isect_build_impl(int C, int64_t N, int64_t M, int64_t n_visible, const float *means2d, const int32_t *radii,
                            const float *depths, const float *conics, const float *opacities, int tile_size,
                            int tile_w, int tile_h, const void *ws,
                            size_t ws_bytes, void *ws2, size_t ws2_bytes, int64_t *isect_ids, int32_t *flatten_ids,
                            int32_t *isect_offsets, int32_t *visible_ids, int compact, bds_stream_t stream, bool dev) {
  BDS_REQUIRE(C >= 1 && N >= 0 && M >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0 && isect_offsets);
  BDS_REQUIRE(!(compact && isect_ids));          // the 64-bit keys need the Gaussian ids
  BDS_REQUIRE(!visible_ids || n_visible >= 0);
  if (visible_ids && n_visible > 0) {   // ascending ids of the visible entries (compact position -> cam*N + g) for the caller
    BDS_REQUIRE(ws);
    const PrepWs P0 = prep_layout(const_cast<void *>(ws), (int64_t)C * N);
    if (ws_bytes < P0.bytes) return BDS_EWORKSPACE;
    if (hipMemcpyAsync(visible_ids, P0.asc, sizeof(int32_t) * n_visible, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess)
      return BDS_ELAUNCH;
  }
  BDS_REQUIRE(M < (int64_t)1 << 31);
  const int64_t CN = (int64_t)C * N;
  const int n_tiles_total = C * tile_w * tile_h;
  hipStream_t st = as_stream(stream);
  if (M == 0) {
    if (hipMemsetAsync(isect_offsets, 0, sizeof(int32_t) * n_tiles_total, st) != hipSuccess) return BDS_ELAUNCH;
    return BDS_OK;
  }
  BDS_REQUIRE(means2d && radii && depths && ws && ws2 && flatten_ids);
  const ProjLayout pl = proj_layout(means2d, depths, conics, opacities);
  PrepWs P = prep_layout(const_cast<void *>(ws), CN);
  if (ws_bytes < P.bytes) return BDS_EWORKSPACE;
  const uint64_t *const M_dev = dev ? P.total + kCountMEff : nullptr;
  const uint64_t *const nvis_dev = dev ? P.total + kCountVisEff : P.total + 1;
  BuildWs B = build_layout(ws2, M);
  if (ws2_bytes < B.bytes) return BDS_EWORKSPACE;
  // number of radix passes over the tile key
  int nbits = 1;
  while (((int64_t)1 << nbits) < n_tiles_total) nbits++;
  BDS_REQUIRE((conics == nullptr) == (opacities == nullptr));
  uint32_t *fl = reinterpret_cast<uint32_t *>(flatten_ids);
  // Packed lists: when the depth rank of every visible entry fits next to the tile key in ONE 32-bit word
  // (tile << rank_bits | rank), the sort moves 4 bytes per entry instead of 8 and the Gaussian ids are looked up from
  // the ranks while the last pass writes out.  Same order: entries are emitted by increasing rank and the passes are stable.
  const int rank_bits = 32 - nbits;
  const bool packed = n_visible >= 0 && rank_bits >= 1 && n_visible <= ((int64_t)1 << rank_bits) && option_get(kOptPacked);
  if (dev && !packed) return BDS_ECAPACITY;   // the device-count form exists for the packed lists only (every fused-view shape)
  // digits: 8 bits, or -- packed lists whose whole key fits -- ONE pass of 9-10 bits (radix_scatter_keys_wide_kernel)
  const int max_digit = (packed && nbits > 8 && nbits <= kWideBits) ? kWideBits : 8;
  const int npass = (nbits + max_digit - 1) / max_digit;
  const int bits_per = (nbits + npass - 1) / npass;
  int key_shift = 0;
  bool offsets_fused = false;
  uint32_t *kin;
  if (packed) {
    key_shift = rank_bits;
    uint32_t *k_emit = (npass % 2 == 1) ? B.ka : B.kb;   // the last pass lands in B.kb
    // a wide pass (radix_pass_keys, bits > 8) finds its group rows behind the histogram; the emission clears them on its way
    const bool wide = bits_per > 8;
    uint32_t *wide_zero = wide ? B.temp + align_up((size_t)(1 << kWideBits) * (size_t)cdiv(M, kSortChunk), 4) : nullptr;
    const int wide_zero_n = wide ? (int)radix_wide_group_elems(M, bits_per) : 0;
    // (n_visible: the exact count, or the capacity of the device-count form -- no workgroups for entries that cannot exist)
    const int64_t emit_bound = (n_visible > 0 && n_visible < CN && option_get(kOptCapLaunch)) ? n_visible : CN;
    hipLaunchKernelGGL(isect_emit_rows_kernel, dim3((unsigned)cdiv(emit_bound, kIsectBlock)), dim3(kIsectBlock), 0, st, nvis_dev, N, P.va,
                       P.btot, means2d, radii, conics, opacities, tile_size, tile_w, tile_h, k_emit, (uint32_t *)nullptr, rank_bits, P.rec,
                       wide_zero, wide_zero_n);
    BDS_LAUNCH_CHECK();
    kin = k_emit;
    const uint32_t rank_mask = (1u << rank_bits) - 1u;
    for (int p = 0; p < npass; p++) {
      uint32_t *kout = (kin == B.ka) ? B.kb : B.ka;
      int bits = bits_per;
      if (bits_per * (p + 1) > nbits) bits = nbits - bits_per * p;
      const bool last = p == npass - 1;
      // a single wide pass over the whole tile key also yields the per-tile offsets (no separate launch)
      offsets_fused = npass == 1 && bits > 8 && !isect_ids;
      int rc = radix_pass_keys(kin, kout, M, rank_bits + bits_per * p, bits, B.temp, st, last ? P.va : nullptr, rank_mask,
                               last ? fl : nullptr, offsets_fused ? isect_offsets : nullptr, n_tiles_total, M_dev);
      if (rc != BDS_OK) return rc;
      kin = kout;
    }
  } else {
    // buffers: pass outputs alternate so that the LAST pass writes values into flatten_ids
    uint32_t *k_emit, *v_emit;
    if (npass % 2 == 1) { k_emit = B.ka; v_emit = B.va; }   // A -> (kb, fl)
    else { k_emit = B.kb; v_emit = fl; }                      // (kb, fl) -> A -> (kb, fl)
    hipLaunchKernelGGL(isect_emit_rows_kernel, dim3((unsigned)cdiv(CN, kIsectBlock)), dim3(kIsectBlock), 0, st, P.total + 1, N,
                       P.va, P.btot, means2d, radii, conics, opacities, tile_size, tile_w, tile_h, k_emit, v_emit, 0, P.rec,
                       (uint32_t *)nullptr, 0);
    BDS_LAUNCH_CHECK();
    kin = k_emit;
    uint32_t *vin = v_emit;
    for (int p = 0; p < npass; p++) {
      uint32_t *kout = (kin == B.ka) ? B.kb : B.ka;
      uint32_t *vout = (vin == B.va) ? fl : B.va;
      int bits = bits_per;
      if (bits_per * (p + 1) > nbits) bits = nbits - bits_per * p;
      int rc = radix_pass(kin, vin, kout, vout, M, bits_per * p, bits, B.temp, st);
      if (rc != BDS_OK) return rc;
      kin = kout; vin = vout;
    }
  }
  // now kin == B.kb (sorted tile keys, or packed words), flatten_ids holds the Gaussian ids
  if (!offsets_fused) {
    hipLaunchKernelGGL(isect_offsets_kernel, dim3((unsigned)cdiv(M + 1, kIsectBlock)), dim3(kIsectBlock), 0, st, M, M_dev, kin, key_shift,
                       n_tiles_total, isect_offsets);
    BDS_LAUNCH_CHECK();
  }
  if (isect_ids) {
    hipLaunchKernelGGL(isect_ids_kernel, dim3((unsigned)cdiv(M, kIsectBlock)), dim3(kIsectBlock), 0, st, M, kin, key_shift,
                       flatten_ids, depths, isect_ids, pl.sd);
    BDS_LAUNCH_CHECK();
  }
  return BDS_OK;
}

extern "C" int bds_isect_build(int C, int64_t N, int64_t M, int64_t n_visible, const float *means2d, const int32_t *radii,
                               const float *depths, const float *conics, const float *opacities, int tile_size,
                               int tile_w, int tile_h, const void *ws,
                               size_t ws_bytes, void *ws2, size_t ws2_bytes, int64_t *isect_ids, int32_t *flatten_ids,
                               int32_t *isect_offsets, int32_t *visible_ids, int compact, bds_stream_t stream) {
  return isect_build_impl(C, N, M, n_visible, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, ws, ws_bytes, ws2,
                          ws2_bytes, isect_ids, flatten_ids, isect_offsets, visible_ids, compact, stream, false);
}

extern "C" int bds_isect_build_dev(int C, int64_t N, int64_t M_capacity, int64_t n_visible_capacity, const float *means2d,
                                   const int32_t *radii, const float *depths, const float *conics, const float *opacities,
                                   int tile_size, int tile_w, int tile_h, const void *ws, size_t ws_bytes, void *ws2,
                                   size_t ws2_bytes, int32_t *flatten_ids, int32_t *isect_offsets, int compact, bds_stream_t stream) {
  BDS_REQUIRE(M_capacity > 0 && n_visible_capacity > 0);
  return isect_build_impl(C, N, M_capacity, n_visible_capacity, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, ws,
                          ws_bytes, ws2, ws2_bytes, nullptr, flatten_ids, isect_offsets, nullptr, compact, stream, true);
}

extern "C" int bds_isect_tiles(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                               const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                               int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, void *ws2, size_t ws2_bytes,
                               int64_t flatten_capacity, int64_t *isect_ids, int32_t *flatten_ids, int32_t *isect_offsets,
                               int64_t *n_isects, int64_t *n_visible, bds_stream_t stream) {
  BDS_REQUIRE(flatten_capacity >= 0 && n_isects);
  int64_t nvis_local = 0;
  if (!n_visible) n_visible = &nvis_local;
  int rc = bds_isect_prepare(C, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, tiles_per_gauss, ws,
                             ws_bytes, n_isects, n_visible, 0, stream);
  if (rc != BDS_OK) return rc;
  const int64_t M = *n_isects;
  if (M > flatten_capacity || (M > 0 && ws2_bytes < build_layout(nullptr, M).bytes)) return BDS_ECAPACITY;
  return bds_isect_build(C, N, M, *n_visible, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, ws, ws_bytes, ws2,
                         ws2_bytes, isect_ids, flatten_ids, isect_offsets, nullptr, 0, stream);
}
