// Adaptive density control: split / duplicate / cull of one class of Gaussians together with its Adam state, and the periodic
// opacity reset -- VanillaGaussians.refinement_after (/root/reference/project/models/gaussians/vanilla.py:205-304),
// split_gaussians (:336-363), dup_gaussians (:365-376), cull_gaussians (:306-334) and the optimiser surgery
// dup_in_optim / remove_from_optim (models/gaussians/basics.py:162-206).  SURVEY.md 8f rank 2, second slice.
//
// The reference builds the new set in three rounds of boolean-mask indexing + torch.cat per tensor (6 parameters + 12 state
// tensors, each round a nonzero() with a host sync) and then compacts everything again for the cull.  Here the whole topology
// change is PLANNED once per Gaussian (5 flag bits + 4 exclusive ranks) and every array is then written exactly once, straight
// into its final, culled layout:
//
//   reference order after cat  : [ originals 0..N ) [ split children, sample-major: N + s*S + j ) [ dup children )
//   after the cull (order kept): [ kept originals ) [ kept split children of sample 0 | sample 1 | .. ) [ kept dup children )
//
// The children of one parent share its opacity and (shrunk) scale and have max_2Dsize = 0, so their cull decision is a function
// of the parent alone -> the destination of every row follows from per-parent ranks:
//   original g            -> rank_keepO[g]
//   split child s of g    -> KO + s*KS + rank_keepS[g]        (its noise sample is row s*S + rank_split[g] of `samples`)
//   dup child of g        -> KO + samps*KS + rank_keepD[g]
// HBM-bound byte movement: every parameter / state row is read once and written once (+ once per kept child).
// PINNED by tests/golden/refine_*.npz (the reference's own refinement_after, oracle/gen_golden_refine.py).
#include "bds_common.h"

namespace bds {

constexpr int kRefBlock = 256;
constexpr int kRefWaves = kRefBlock / kWave;
enum RefineFlag : uint8_t { kSplit = 1, kDup = 2, kKeepO = 4, kKeepS = 8, kKeepD = 16 };
constexpr int kChan = 5;  // scanned channels: split, dup, keepO, keepS, keepD   (totals[] in this order)

struct RefineCfg {
  int do_densify;
  float grad_thresh, size_thresh;
  int split_by_screen;
  float split_screen_size;
  int do_cull;
  float cull_alpha;
  int cull_by_scale;
  float cull_scale;
  int cull_by_screen;
  float cull_screen;
};

__device__ __forceinline__ float shrink_log_scale(float ls) {
#pragma clang fp contract(off)
  return logf(expf(ls) / 1.6f);  // vanilla.py:358-359  log(exp(s) / size_fac)
}

__device__ __forceinline__ uint8_t refine_classify(int64_t g, const RefineCfg &c, const float *__restrict__ xys_grad_norm,
                                                   const float *__restrict__ vis_counts, const float *__restrict__ max_2Dsize,
                                                   const float *__restrict__ log_scales, const float *__restrict__ logits,
                                                   const uint8_t *__restrict__ extra_cull) {
#pragma clang fp contract(off)
  const float lmax = fmaxf(fmaxf(log_scales[g * 3], log_scales[g * 3 + 1]), log_scales[g * 3 + 2]);
  const float smax = expf(lmax);  // exp is monotone: max_i exp(s_i) = exp(max_i s_i)
  const float m2d = max_2Dsize ? max_2Dsize[g] : 0.f;
  bool split = false, dup = false;
  float s_post = smax;  // largest world-space scale after split_gaussians has shrunk the split parents in place
  if (c.do_densify) {  // vanilla.py:219-250
    const bool high = (xys_grad_norm[g] / vis_counts[g]) > c.grad_thresh;
    split = smax > c.size_thresh;
    if (c.split_by_screen) split = split || (m2d > c.split_screen_size);
    split = split && high;
    if (split) s_post = expf(shrink_log_scale(lmax));
    // the dup mask is evaluated AFTER the split (:246-250): a small Gaussian that is large on screen, or one that drops below
    // the size threshold by the 1/1.6 shrink, is split AND duplicated
    dup = (s_post <= c.size_thresh) && high;
  }
  bool cull_o = false, cull_child = false;
  if (c.do_cull) {  // vanilla.py:312-325, evaluated on the post-split scales
    const float op = 1.f / (1.f + expf(-logits[g]));
    cull_o = cull_child = op < c.cull_alpha;
    if (c.cull_by_scale) {
      const bool toobig = s_post > c.cull_scale;
      cull_o = cull_o || toobig;
      cull_child = cull_child || toobig;
      if (c.cull_by_screen) cull_o = cull_o || (m2d > c.cull_screen);  // children enter with max_2Dsize = 0
    }
  }
  // a caller-computed mask over the INPUT rows (the node classes' box test, see bds_refine_out_of_bound): the children are judged
  // by their own rows in a second plan, not by their parent
  if (c.do_cull && extra_cull && extra_cull[g]) cull_o = true;
  uint8_t f = 0;
  if (split) f |= kSplit;
  if (dup) f |= kDup;
  if (!cull_o) f |= kKeepO;
  if (split && !cull_child) f |= kKeepS;
  if (dup && !cull_child) f |= kKeepD;
  return f;
}

// flags + per-workgroup counts of the five channels
__global__ __launch_bounds__(kRefBlock) void refine_flags_kernel(int64_t N, RefineCfg c, const float *__restrict__ xys_grad_norm,
                                                                const float *__restrict__ vis_counts,
                                                                const float *__restrict__ max_2Dsize,
                                                                const float *__restrict__ log_scales,
                                                                const float *__restrict__ logits,
                                                                const uint8_t *__restrict__ extra_cull, uint8_t *__restrict__ flags,
                                                                uint32_t *__restrict__ blk) {
  __shared__ uint32_t wsum[kRefWaves][kChan];
  const int64_t g = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  uint8_t f = 0;
  if (g < N) {
    f = refine_classify(g, c, xys_grad_norm, vis_counts, max_2Dsize, log_scales, logits, extra_cull);
    flags[g] = f;
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int ch = 0; ch < kChan; ch++) {
    const uint64_t b = __ballot((f >> ch) & 1);
    if (lane == 0) wsum[wave][ch] = (uint32_t)__popcll(b);
  }
  __syncthreads();
  if (threadIdx.x < kChan) {
    uint32_t s = 0;
    for (int w = 0; w < kRefWaves; w++) s += wsum[w][threadIdx.x];
    blk[(int64_t)blockIdx.x * kChan + threadIdx.x] = s;
  }
}

// exclusive scan of the per-workgroup counts (in place) + totals; ONE workgroup of 1024 threads, each owning a contiguous
// segment of the counts
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void refine_blockscan_kernel(int64_t nb, uint32_t *__restrict__ blk,
                                                                       int64_t *__restrict__ totals) {
  __shared__ uint32_t part[kScanThreads][kChan];
  const int t = threadIdx.x;
  const int64_t seg = (nb + kScanThreads - 1) / kScanThreads;
  const int64_t lo = t * seg, hi = (lo + seg < nb) ? lo + seg : nb;
  uint32_t s[kChan] = {0, 0, 0, 0, 0};
  for (int64_t i = lo; i < hi; i++)
#pragma unroll
    for (int ch = 0; ch < kChan; ch++) s[ch] += blk[i * kChan + ch];
#pragma unroll
  for (int ch = 0; ch < kChan; ch++) part[t][ch] = s[ch];
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partial sums
  for (int d = 1; d < kScanThreads; d <<= 1) {
    uint32_t add[kChan];
#pragma unroll
    for (int ch = 0; ch < kChan; ch++) add[ch] = t >= d ? part[t - d][ch] : 0u;
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < kChan; ch++) part[t][ch] += add[ch];
    __syncthreads();
  }
  uint32_t run[kChan];
#pragma unroll
  for (int ch = 0; ch < kChan; ch++) run[ch] = part[t][ch] - s[ch];  // exclusive prefix of this segment
  for (int64_t i = lo; i < hi; i++)
#pragma unroll
    for (int ch = 0; ch < kChan; ch++) {
      const uint32_t v = blk[i * kChan + ch];
      blk[i * kChan + ch] = run[ch];
      run[ch] += v;
    }
  if (t == kScanThreads - 1)
#pragma unroll
    for (int ch = 0; ch < kChan; ch++) totals[ch] = (int64_t)part[t][ch];
}

// exclusive ranks per Gaussian: [split, keepO, keepS, keepD]
__global__ __launch_bounds__(kRefBlock) void refine_ranks_kernel(int64_t N, const uint8_t *__restrict__ flags,
                                                                const uint32_t *__restrict__ blk, uint32_t *__restrict__ ranks) {
  __shared__ uint32_t wsum[kRefWaves][4];
  const int64_t g = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  const uint8_t f = g < N ? flags[g] : 0;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int chan[4] = {0, 2, 3, 4};
  uint32_t local[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t b = __ballot((f >> chan[k]) & 1);
    local[k] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave][k] = (uint32_t)__popcll(b);
  }
  __syncthreads();
  if (g >= N) return;
  uint4 r;
  uint32_t out[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t o = blk[(int64_t)blockIdx.x * kChan + chan[k]] + local[k];
    for (int w = 0; w < wave; w++) o += wsum[w][k];
    out[k] = o;
  }
  r.x = out[0]; r.y = out[1]; r.z = out[2]; r.w = out[3];
  reinterpret_cast<uint4 *>(ranks)[g] = r;
}

// means + log-scales of the new set (the only two parameters a split changes)
__global__ __launch_bounds__(kRefBlock) void refine_geometry_kernel(int64_t N, int samps, const uint8_t *__restrict__ flags,
                                                                   const uint32_t *__restrict__ ranks,
                                                                   const int64_t *__restrict__ totals,
                                                                   const float *__restrict__ samples,
                                                                   const float *__restrict__ means, const float *__restrict__ quats,
                                                                   const float *__restrict__ log_scales,
                                                                   float *__restrict__ new_means, float *__restrict__ new_log_scales) {
#pragma clang fp contract(off)
  const int64_t g = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  if (g >= N) return;
  const uint8_t f = flags[g];
  if (!(f & (kKeepO | kKeepS | kKeepD))) return;
  const uint4 r = reinterpret_cast<const uint4 *>(ranks)[g];
  const int64_t S = totals[0], KO = totals[2], KS = totals[3];
  const float m[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
  const float ls[3] = {log_scales[g * 3], log_scales[g * 3 + 1], log_scales[g * 3 + 2]};
  float lp[3] = {ls[0], ls[1], ls[2]};
  if (f & kSplit)
    for (int i = 0; i < 3; i++) lp[i] = shrink_log_scale(ls[i]);  // parent and children alike (vanilla.py:358-359)
  auto put = [&](int64_t dst, const float *mm) {
    for (int i = 0; i < 3; i++) { new_means[dst * 3 + i] = mm[i]; new_log_scales[dst * 3 + i] = lp[i]; }
  };
  if (f & kKeepO) put((int64_t)r.y, m);
  if (f & kKeepS) {
    // vanilla.py:343-348: rotate (scale * noise) by the normalised quaternion, add the mean
    float q[4] = {quats[g * 4], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
    const float n1 = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] = q[i] / n1;                                  // quat_act
    const float n2 = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    for (int i = 0; i < 4; i++) q[i] = q[i] / n2;                                  // F.normalize inside quat_to_rotmat
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                        2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                        2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    const float sc[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};  // the scale BEFORE the shrink
    for (int s = 0; s < samps; s++) {
      const float *smp = samples + ((int64_t)s * S + r.x) * 3;
      const float v[3] = {sc[0] * smp[0], sc[1] * smp[1], sc[2] * smp[2]};
      float nm[3];
      for (int i = 0; i < 3; i++) nm[i] = (R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2]) + m[i];
      put(KO + (int64_t)s * KS + r.z, nm);
    }
  }
  if (f & kKeepD) put(KO + (int64_t)samps * KS + r.w, m);
}

// any other per-Gaussian array, one thread per element: rows of kept originals move to their new place; the children receive a
// copy of the parent's row (parameters) or zeros (Adam moments: basics.py:191-201 appends zeros_like)
__global__ __launch_bounds__(kRefBlock) void refine_rows_kernel(int64_t N, int width, int samps, const uint8_t *__restrict__ flags,
                                                               const uint32_t *__restrict__ ranks,
                                                               const int64_t *__restrict__ totals, const float *__restrict__ src,
                                                               float *__restrict__ dst, int zero_children) {
  const int64_t i = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  if (i >= N * width) return;
  const int64_t g = i / width;
  const int c = (int)(i - g * width);
  const uint8_t f = flags[g];
  if (!(f & (kKeepO | kKeepS | kKeepD))) return;
  const uint4 r = reinterpret_cast<const uint4 *>(ranks)[g];
  const int64_t KO = totals[2], KS = totals[3];
  const float v = src[i];
  const float cv = zero_children ? 0.f : v;
  if (f & kKeepO) dst[(int64_t)r.y * width + c] = v;
  if (f & kKeepS)
    for (int s = 0; s < samps; s++) dst[(KO + (int64_t)s * KS + r.z) * width + c] = cv;
  if (f & kKeepD) dst[(KO + (int64_t)samps * KS + r.w) * width + c] = cv;
}

// nodes/rigid.py:374-383 (get_out_of_bound_mask): |mean| > instances_size[point_id] / 2 in any axis, means in the object frame
__global__ __launch_bounds__(kRefBlock) void refine_out_of_bound_kernel(int64_t N, const float *__restrict__ means,
                                                                       const int64_t *__restrict__ point_ids, int64_t n_instances,
                                                                       const float *__restrict__ instances_size,
                                                                       uint8_t *__restrict__ mask) {
#pragma clang fp contract(off)
  const int64_t g = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  if (g >= N) return;
  const int64_t id = point_ids[g];
  bool out = true;  // an id outside the table cannot be inside any box
  if (id >= 0 && id < n_instances) {
    out = false;
    for (int i = 0; i < 3; i++) out = out || (fabsf(means[g * 3 + i]) > instances_size[id * 3 + i] / 2.f);
  }
  mask[g] = out ? 1 : 0;
}

// vanilla.py:286-299: opacity = logit(min(sigmoid(opacity), reset_value)); the opacity group's Adam moments start again from zero
__global__ __launch_bounds__(kRefBlock) void opacity_reset_kernel(int64_t N, float *__restrict__ logits, float reset_value,
                                                                 float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kRefBlock + threadIdx.x;
  if (i >= N) return;
  const float op = 1.f / (1.f + expf(-logits[i]));
  const float x = fminf(op, reset_value);
  logits[i] = logf(x / (1.f - x));  // torch.logit
  if (exp_avg) exp_avg[i] = 0.f;
  if (exp_avg_sq) exp_avg_sq[i] = 0.f;
}

}  // namespace bds

using namespace bds;

extern "C" size_t bds_refine_plan_temp_bytes(int64_t N) {
  if (N < 0) return 0;
  return (size_t)(cdiv(N, kRefBlock) + 1) * kChan * sizeof(uint32_t);
}

extern "C" int bds_refine_plan(int64_t N, const float *xys_grad_norm, const float *vis_counts, const float *max_2Dsize,
                               const float *log_scales, const float *logits, const uint8_t *extra_cull, int do_densify,
                               float grad_thresh,
                               float size_thresh, int split_by_screen, float split_screen_size, int do_cull,
                               float cull_alpha_thresh, int cull_by_scale, float cull_scale_thresh, int cull_by_screen,
                               float cull_screen_size, uint8_t *flags, uint32_t *ranks, int64_t *totals, void *temp,
                               size_t temp_bytes, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && N < ((int64_t)1 << 31) && totals);
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    if (hipMemsetAsync(totals, 0, kChan * sizeof(int64_t), st) != hipSuccess) return BDS_ELAUNCH;
    return BDS_OK;
  }
  BDS_REQUIRE(log_scales && logits && flags && ranks && temp && temp_bytes >= bds_refine_plan_temp_bytes(N));
  BDS_REQUIRE(!do_densify || (xys_grad_norm && vis_counts));
  BDS_REQUIRE(aligned16(ranks));
  RefineCfg c{do_densify, grad_thresh, size_thresh, split_by_screen, split_screen_size, do_cull, cull_alpha_thresh,
              cull_by_scale, cull_scale_thresh, cull_by_screen, cull_screen_size};
  const int64_t nb = cdiv(N, kRefBlock);
  uint32_t *blk = static_cast<uint32_t *>(temp);
  hipLaunchKernelGGL(refine_flags_kernel, dim3((unsigned)nb), dim3(kRefBlock), 0, st, N, c, xys_grad_norm, vis_counts, max_2Dsize,
                     log_scales, logits, extra_cull, flags, blk);
  hipLaunchKernelGGL(refine_blockscan_kernel, dim3(1), dim3(kScanThreads), 0, st, nb, blk, totals);
  hipLaunchKernelGGL(refine_ranks_kernel, dim3((unsigned)nb), dim3(kRefBlock), 0, st, N, flags, blk, ranks);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_refine_geometry(int64_t N, int samps, const uint8_t *flags, const uint32_t *ranks, const int64_t *totals,
                                   const float *samples, const float *means, const float *quats, const float *log_scales,
                                   float *new_means, float *new_log_scales, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && samps >= 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(flags && ranks && totals && means && quats && log_scales && new_means && new_log_scales);
  hipLaunchKernelGGL(refine_geometry_kernel, dim3((unsigned)cdiv(N, kRefBlock)), dim3(kRefBlock), 0, as_stream(stream), N, samps,
                     flags, ranks, totals, samples, means, quats, log_scales, new_means, new_log_scales);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_refine_rows(int64_t N, int width, int samps, const uint8_t *flags, const uint32_t *ranks,
                               const int64_t *totals, const float *src, float *dst, int zero_children, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && width >= 1 && samps >= 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(flags && ranks && totals && src && dst);
  hipLaunchKernelGGL(refine_rows_kernel, dim3((unsigned)cdiv(N * width, kRefBlock)), dim3(kRefBlock), 0, as_stream(stream), N, width,
                     samps, flags, ranks, totals, src, dst, zero_children);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_refine_out_of_bound(int64_t N, const float *means, const int64_t *point_ids, int64_t n_instances,
                                       const float *instances_size, uint8_t *mask, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && n_instances >= 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(means && point_ids && instances_size && mask);
  hipLaunchKernelGGL(refine_out_of_bound_kernel, dim3((unsigned)cdiv(N, kRefBlock)), dim3(kRefBlock), 0, as_stream(stream), N, means,
                     point_ids, n_instances, instances_size, mask);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_opacity_reset(int64_t N, float *logits, float reset_value, float *exp_avg, float *exp_avg_sq,
                                 bds_stream_t stream) {
  BDS_REQUIRE(N >= 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(logits);
  hipLaunchKernelGGL(opacity_reset_kernel, dim3((unsigned)cdiv(N, kRefBlock)), dim3(kRefBlock), 0, as_stream(stream), N, logits,
                     reset_value, exp_avg, exp_avg_sq);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
