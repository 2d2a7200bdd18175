// K7/K8: per-pixel front-to-back alpha compositing, forward and backward.
// rasterize_to_pixels stage of gsplat.rendering.rasterization as called at
// /root/reference/project/models/trainers/base.py:393-408 (render_mode "RGB+ED" -> CH = 4,
// viewer "RGB" -> CH = 3).
//
// Mapping (gfx950): ONE wave64 per 16x16 tile, four pixels per lane (lane l: column l % 16, rows
// l / 16 + 4q): the tile's depth-ordered Gaussians are staged through LDS in chunks of 64 (one gathered
// Gaussian per lane, next chunk prefetched into registers), then every lane walks the chunk reading LDS
// at a wave-uniform address (broadcast reads); dx and the x-terms of the quadratic form are shared by the
// lane's four pixels.  Early termination: a pixel stops at T*(1-a) <= 1e-4, the wave leaves once all its
// pixels are done (ballot).  Workgroup ids are remapped so that each XCD rasterises one contiguous band of
// the image (its private L2 then serves the re-reads of Gaussians shared by neighbouring tiles).
#include "bds_common.h"
#include "gs_math.h"

namespace bds {

constexpr int kTile = 16;
constexpr int kRastBlock = kTile * kTile;  // 256

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// Work item of a workgroup: the XCD-contiguous position, optionally redirected through a schedule
// (bds_rasterize_bwd_schedule: each XCD's range re-ordered longest tile first, see below).
__device__ __forceinline__ int pick_item(const int32_t *__restrict__ order, int bid, int total) {
  const int p = xcd_contiguous(bid, total);
  return order ? order[p] : p;
}

// value slots of the per-Gaussian gradient record that is reduced over a tile's pixels
//   0-3 colour, 4-6 conic, 7-8 mean2d, 9-10 |mean2d| (absgrad), 11 opacity
struct GradTarget {
  float *ptr;   // nullptr: this lane does not commit anything
  int stride;
};
template <int CH, bool ABS>
__device__ __forceinline__ GradTarget grad_target(int lane, float *v_means2d, float *v_means2d_abs, float *v_conics,
                                                  float *v_colors, float *v_opacities) {
  GradTarget g{nullptr, 0};
  if (lane & 3) return g;  // one committing lane per quad
  const int k = butterfly_slot(lane);
  if (k < 4) { if (k < CH) { g.ptr = v_colors + k; g.stride = CH; } }
  else if (k < 7) { g.ptr = v_conics + (k - 4); g.stride = 3; }
  else if (k < 9) { g.ptr = v_means2d + (k - 7); g.stride = 2; }
  else if (k < 11) { if (ABS) { g.ptr = v_means2d_abs + (k - 9); g.stride = 2; } }
  else if (k == 11) { g.ptr = v_opacities; g.stride = 1; }
  return g;
}

template <int CH>
__device__ __forceinline__ void stage_gaussian(int32_t g, const float *__restrict__ means2d,
                                               const float *__restrict__ conics, const float *__restrict__ colors,
                                               const float *__restrict__ opacities, float4 &A, float4 &B, float4 &Cc) {
  const float2 xy = *reinterpret_cast<const float2 *>(means2d + (int64_t)g * 2);
  const float *cn = conics + (int64_t)g * 3;
  const float *cl = colors + (int64_t)g * CH;
  A = make_float4(xy.x, xy.y, cn[0], cn[1]);
  B = make_float4(cn[2], opacities[g], cl[0], CH > 1 ? cl[CH > 1 ? 1 : 0] : 0.f);
  Cc = make_float4(CH > 2 ? cl[CH > 2 ? 2 : 0] : 0.f, CH > 3 ? cl[CH > 3 ? 3 : 0] : 0.f, 0.f, 0.f);
}

// ---- forward, one wave64 per 16x16 tile, four pixels per lane ------------------------------------
// Same pixel ownership as the backward wave kernel (lane l: column l % 16, rows (l / 16) + 4q): the
// Gaussian record is read from LDS once per four pixels and dx / a*dx^2 / b*dx are shared.
template <int CH>
__global__ __launch_bounds__(kWave) void rasterize_fwd_wave_kernel(
    int C, int64_t N, int64_t M, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int W,
    int H, int tile_w, int tile_h, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids,
    float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids) {
  __shared__ float4 sA[kWave], sB[kWave], sC[kWave];
  const int n_tiles = tile_w * tile_h;
  const int item = xcd_contiguous(blockIdx.x, C * n_tiles);
  const int cam = item / n_tiles, tile = item - cam * n_tiles;
  const int ty = tile / tile_w, tx = tile - ty * tile_w;
  const int lane = threadIdx.x;
  const int j = tx * kTile + (lane & 15);
  const int i0 = ty * kTile + (lane >> 4);
  const float px = (float)j + 0.5f;
  const int start = offsets[item];
  const int end = (item == C * n_tiles - 1) ? (int)M : offsets[item + 1];
  // T[q] > 0: running transmittance; T[q] < 0: the pixel is finished and |T[q]| is its final transmittance
  // (pixels outside the image start finished).  One register instead of a flag + a value per pixel.
  float T[4];
  int cur[4] = {0, 0, 0, 0};
  float out[4][4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    T[q] = ((i0 + 4 * q) < H && j < W) ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[q][k] = 0.f;
  }
  const int nbatch = (end - start + kWave - 1) / kWave;
  // the next chunk's records are fetched into registers while the current chunk is blended, so the
  // two dependent gather latencies (id -> attributes) are off the critical path of this single-wave workgroup
  float4 pA = make_float4(0.f, 0.f, 0.f, 0.f), pB = pA, pC = pA;
  if (start + lane < end) stage_gaussian<CH>(flatten_ids[start + lane], means2d, conics, colors, opacities, pA, pB, pC);
  for (int b = 0; b < nbatch; b++) {
    if (__all(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])) < 0.f)) break;
    __syncthreads();
    const int bstart = start + b * kWave;
    if (bstart + lane < end) {
      sA[lane] = pA; sB[lane] = pB;
      if (CH > 2) sC[lane] = pC;
    }
    __syncthreads();
    if (bstart + kWave + lane < end)
      stage_gaussian<CH>(flatten_ids[bstart + kWave + lane], means2d, conics, colors, opacities, pA, pB, pC);
    const int bs = min(kWave, end - bstart);
    for (int t = 0; t < bs; t++) {
      if (__all(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])) < 0.f)) break;
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px;
      const float hax2 = 0.5f * A.z * dx * dx, bdx = A.w * dx;
      float4 Cc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (CH > 2) Cc = sC[t];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float dy = A.y - ((float)(i0 + 4 * q) + 0.5f);
        const float sigma = hax2 + (0.5f * B.x * dy + bdx) * dy;
        const float alpha = fminf(kAlphaMax, B.y * __expf(-sigma));
        const bool hit = T[q] > 0.f && !(sigma < 0.f || alpha < kAlphaMin);
        const float nT = T[q] * (1.f - alpha);
        if (hit && nT <= kTStop) T[q] = -T[q];
        else if (hit) {
          const float vis = alpha * T[q];
          out[q][0] += B.z * vis;
          if (CH > 1) out[q][1] += B.w * vis;
          if (CH > 2) out[q][2] += Cc.x * vis;
          if (CH > 3) out[q][3] += Cc.y * vis;
          cur[q] = bstart + t;
          T[q] = nT;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + 4 * q;
    if (i < H && j < W) {
      const int64_t pix = ((int64_t)cam * H + i) * W + j;
      const float Tf = fabsf(T[q]);
      alphas[pix] = 1.f - Tf;
      last_ids[pix] = cur[q];
      float *r = render + pix * CH;
#pragma unroll
      for (int k = 0; k < CH; k++) r[k] = backgrounds ? out[q][k] + Tf * backgrounds[cam * CH + k] : out[q][k];
    }
  }
}

// ---- backward, one wave64 per 16x16 tile, four pixels per lane ------------------------------------
// Lane l owns column l % 16 and rows (l / 16) + 4q, q = 0..3: the four pixels share dx, and their
// partial gradients are summed in registers before the single 16-value transpose-reduce, so the
// cross-lane work and the atomics are paid once per (Gaussian, tile) instead of once per wave.
// Strip q (rows 4q..4q+3) is skipped as a whole when none of its 64 pixels needs the Gaussian.
template <int CH, bool ABS>
__global__ __launch_bounds__(kWave) void rasterize_bwd_wave_kernel(
    int C, int64_t N, int64_t M, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int W,
    int H, int tile_w, int tile_h, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids, const float *__restrict__ v_render,
    const float *__restrict__ v_alphas, float *__restrict__ v_means2d, float *__restrict__ v_means2d_abs,
    float *__restrict__ v_conics, float *__restrict__ v_colors, float *__restrict__ v_opacities,
    const int32_t *__restrict__ tile_order) {
  __shared__ float4 sA[kWave], sB[kWave], sC[kWave];
  __shared__ int32_t sId[kWave];
  const int n_tiles = tile_w * tile_h;
  const int item = pick_item(tile_order, blockIdx.x, C * n_tiles);
  const int cam = item / n_tiles, tile = item - cam * n_tiles;
  const int ty = tile / tile_w, tx = tile - ty * tile_w;
  const int lane = threadIdx.x;
  const int start = offsets[item];
  const int end = (item == C * n_tiles - 1) ? (int)M : offsets[item + 1];
  if (end <= start) return;
  const int j = tx * kTile + (lane & 15);
  const float px = (float)j + 0.5f;
  const int i0 = ty * kTile + (lane >> 4);
  bool inside[4];
  // Bd[q] = sum_k buffer_k * v_render_k - T_final * (v_alpha - bg . v_render): the only combination of the accumulated
  // colour `buffer` that the gradient needs, kept as ONE scalar per pixel (saves 16 VGPRs and 5 VALU ops per pixel-pair)
  float T[4], Bd[4], vr[4][4];
  int bin_final[4];
  int max_bin = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + 4 * q;
    inside[q] = i < H && j < W;
    const int64_t pix = ((int64_t)cam * H + (inside[q] ? i : 0)) * W + (inside[q] ? j : 0);
    const float T_final = inside[q] ? 1.f - alphas[pix] : 1.f;
    T[q] = T_final;
    bin_final[q] = inside[q] ? last_ids[pix] : -1;   // -1: never valid (pixel outside the image)
    max_bin = max(max_bin, bin_final[q]);
    const float vra = inside[q] ? v_alphas[pix] : 0.f;
    float bgdot = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      vr[q][k] = (k < CH && inside[q]) ? v_render[pix * CH + (k < CH ? k : 0)] : 0.f;
      if (backgrounds && k < CH) bgdot += backgrounds[cam * CH + k] * vr[q][k];
    }
    Bd[q] = -T_final * (vra - bgdot);
  }
  const int tile_bin_final = wave_max_i32(max_bin);
  const GradTarget tgt = grad_target<CH, ABS>(lane, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
  const int py0 = i0;
  const int nbatch = (end - start + kWave - 1) / kWave;
  const int b0 = (end - 1 - tile_bin_final) / kWave;  // chunks in front of it lie behind every pixel's last Gaussian
  // register prefetch of the next chunk (see the forward wave kernel)
  int32_t pg = 0;
  float4 pA = make_float4(0.f, 0.f, 0.f, 0.f), pB = pA, pC = pA;
  {
    const int idx = end - 1 - kWave * b0 - lane;
    if (idx >= start) {
      pg = flatten_ids[idx];
      stage_gaussian<CH>(pg, means2d, conics, colors, opacities, pA, pB, pC);
    }
  }
  for (int b = b0; b < nbatch; b++) {
    const int batch_end = end - 1 - kWave * b;
    __syncthreads();
    const int bs = min(kWave, batch_end + 1 - start);
    if (batch_end - lane >= start) {
      sId[lane] = pg; sA[lane] = pA; sB[lane] = pB;
      if (CH > 2) sC[lane] = pC;
    }
    __syncthreads();
    {
      const int idx = batch_end - kWave - lane;
      if (idx >= start) {
        pg = flatten_ids[idx];
        stage_gaussian<CH>(pg, means2d, conics, colors, opacities, pA, pB, pC);
      }
    }
    for (int t = max(0, batch_end - tile_bin_final); t < bs; t++) {
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px;
      const float opac = B.y;
      const float hax2 = 0.5f * A.z * dx * dx, bdx = A.w * dx;
      float dy[4], vis[4], alpha[4];
      bool valid[4];
      bool any = false;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        dy[q] = A.y - ((float)(py0 + 4 * q) + 0.5f);
        const float sigma = hax2 + (0.5f * B.x * dy[q] + bdx) * dy[q];
        vis[q] = __expf(-sigma);
        alpha[q] = fminf(kAlphaMax, opac * vis[q]);
        valid[q] = (batch_end - t <= bin_final[q]) && !(sigma < 0.f || alpha[q] < kAlphaMin);
        any |= valid[q];
      }
      if (!__any(any)) continue;
      float col[4] = {B.z, B.w, 0.f, 0.f};
      if (CH > 2) { const float4 Cc = sC[t]; col[2] = Cc.x; col[3] = Cc.y; }
      float acc[16];
#pragma unroll
      for (int k = 0; k < 16; k++) acc[k] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!__any(valid[q])) continue;  // strip-uniform skip
        if (valid[q]) {
          const float ra = __builtin_amdgcn_rcpf(1.f - alpha[q]);
          T[q] *= ra;
          const float fac = alpha[q] * T[q];
          float cdot = 0.f;  // colour . v_render
#pragma unroll
          for (int k = 0; k < CH; k++) {
            acc[k] += fac * vr[q][k];
            cdot += col[k] * vr[q][k];
          }
          const float v_alpha = T[q] * cdot - ra * Bd[q];
          Bd[q] += fac * cdot;
          if (opac * vis[q] <= kAlphaMax) {
            const float v_sigma = -opac * vis[q] * v_alpha;
            const float t1 = v_sigma * dx, t2 = v_sigma * dy[q];
            acc[4] += t1 * dx;      // x 0.5 after the strip loop
            acc[5] += t1 * dy[q];
            acc[6] += t2 * dy[q];   // x 0.5 after the strip loop
            const float gx = A.z * t1 + A.w * t2;
            const float gy = A.w * t1 + B.x * t2;
            acc[7] += gx; acc[8] += gy;
            if (ABS) { acc[9] += fabsf(gx); acc[10] += fabsf(gy); }
            acc[11] += vis[q] * v_alpha;
          }
        }
      }
      acc[4] *= 0.5f; acc[6] *= 0.5f;
      const float tot = butterfly_sum16(acc, lane);
      if (tgt.ptr != nullptr) atomicAdd(tgt.ptr + (int64_t)sId[t] * tgt.stride, tot);
    }
  }
}

// ---- backward schedule: longest tile first inside each XCD's range ------------------------------------
// One wave per tile finishes when its LAST pixel does, and the chip holds only ~2 rounds of tiles
// (8160 tiles at 1080p over 1024 SIMDs x 4 resident waves), so tiles dispatched late that happen to be long
// leave most SIMDs idle at the end of the launch.  The visited length of every tile is known exactly after the
// forward pass (max last_id - list start); dispatching each XCD's contiguous range longest-first (LPT rule)
// removes that tail while keeping the range -> XCD assignment (L2 locality) unchanged.
__global__ __launch_bounds__(kRastBlock) void tile_work_kernel(int C, int W, int H, int tile_w, int tile_h,
                                                               const int32_t *__restrict__ offsets,
                                                               const int32_t *__restrict__ last_ids,
                                                               int32_t *__restrict__ work) {
  const int n_tiles = tile_w * tile_h, total = C * n_tiles;
  const int item = blockIdx.x * (kRastBlock / kWave) + (threadIdx.x >> 6);
  if (item >= total) return;
  const int lane = threadIdx.x & 63;
  const int cam = item / n_tiles, tile = item - cam * n_tiles;
  const int ty = tile / tile_w, tx = tile - ty * tile_w;
  const int j = tx * kTile + (lane & 15);
  int m = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = ty * kTile + (lane >> 4) + 4 * q;
    if (i < H && j < W) m = max(m, last_ids[((int64_t)cam * H + i) * W + j]);
  }
  m = wave_max_i32(m);
  // (an empty list leaves last_ids at 0: the estimate is then 0, or 1 for the very first tile)
  if (lane == 0) work[item] = max(0, m - offsets[item] + 1);
}

constexpr int kSchedThreads = 1024, kSchedBins = 1024;
__global__ __launch_bounds__(kSchedThreads) void tile_order_kernel(int total, const int32_t *__restrict__ work,
                                                                   int32_t *__restrict__ order) {
  __shared__ int hist[kSchedBins];
  __shared__ int s_max;
  constexpr int kXcd = 8;
  const int per = total / kXcd, rem = total % kXcd, x = blockIdx.x;
  const int cnt = per + (x < rem ? 1 : 0), first = x * per + (x < rem ? x : rem);
  const int tid = threadIdx.x;
  if (tid == 0) s_max = 0;
  for (int b = tid; b < kSchedBins; b += kSchedThreads) hist[b] = 0;
  __syncthreads();
  int lmax = 0;
  for (int i = tid; i < cnt; i += kSchedThreads) lmax = max(lmax, work[first + i]);
  lmax = wave_max_i32(lmax);
  if ((tid & 63) == 0) atomicMax(&s_max, lmax);
  __syncthreads();
  int shift = 0;
  while ((s_max >> shift) >= kSchedBins) shift++;
  // bin 0 = longest
  for (int i = tid; i < cnt; i += kSchedThreads) atomicAdd(&hist[kSchedBins - 1 - (work[first + i] >> shift)], 1);
  __syncthreads();
  // exclusive scan of the 1024 bins (one bin per thread, Hillis-Steele in LDS)
  const int mine = hist[tid];
  int incl = mine;
  for (int o = 1; o < kSchedBins; o <<= 1) {
    __syncthreads();
    hist[tid] = incl;
    __syncthreads();
    if (tid >= o) incl += hist[tid - o];
  }
  __syncthreads();
  hist[tid] = incl - mine;   // becomes the bin's running cursor
  __syncthreads();
  for (int i = tid; i < cnt; i += kSchedThreads) {
    const int pos = atomicAdd(&hist[kSchedBins - 1 - (work[first + i] >> shift)], 1);
    order[first + pos] = first + i;
  }
}

}  // namespace bds

using namespace bds;

extern "C" int bds_rasterize_fwd(int C, int64_t N, int64_t M, int CH, const float *means2d, const float *conics,
                                 const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                                 int tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                                 const int32_t *flatten_ids, float *render, float *alphas, int32_t *last_ids,
                                 bds_stream_t stream) {
  BDS_REQUIRE(C >= 1 && N >= 0 && M >= 0 && W > 0 && H > 0);
  BDS_REQUIRE(tile_size == kTile);
  BDS_REQUIRE(tile_w == (W + kTile - 1) / kTile && tile_h == (H + kTile - 1) / kTile);
  BDS_REQUIRE(CH == 1 || CH == 3 || CH == 4);
  BDS_REQUIRE(isect_offsets && render && alphas && last_ids);
  BDS_REQUIRE(M == 0 || (means2d && conics && colors && opacities && flatten_ids));
  BDS_REQUIRE((reinterpret_cast<uintptr_t>(means2d) & 7u) == 0);
  const dim3 grid((unsigned)(C * tile_w * tile_h));
  hipStream_t st = as_stream(stream);
#define BDS_FWD_ARGS                                                                                                   \
  C, N, M, means2d, conics, colors, opacities, backgrounds, W, H, tile_w, tile_h, isect_offsets, flatten_ids, render,  \
      alphas, last_ids
#define BDS_FWD(ch) hipLaunchKernelGGL((rasterize_fwd_wave_kernel<ch>), grid, dim3(kWave), 0, st, BDS_FWD_ARGS)
  if (CH == 1) BDS_FWD(1);
  else if (CH == 3) BDS_FWD(3);
  else BDS_FWD(4);
#undef BDS_FWD
#undef BDS_FWD_ARGS
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_rasterize_bwd(int C, int64_t N, int64_t M, int CH, const float *means2d, const float *conics,
                                 const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                                 int tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                                 const int32_t *flatten_ids, const float *alphas, const int32_t *last_ids,
                                 const float *v_render, const float *v_alphas, float *v_means2d, float *v_means2d_abs,
                                 float *v_conics, float *v_colors, float *v_opacities, const int32_t *tile_order,
                                 bds_stream_t stream) {
  BDS_REQUIRE(C >= 1 && N >= 0 && M >= 0 && W > 0 && H > 0);
  BDS_REQUIRE(tile_size == kTile);
  BDS_REQUIRE(tile_w == (W + kTile - 1) / kTile && tile_h == (H + kTile - 1) / kTile);
  BDS_REQUIRE(CH == 1 || CH == 3 || CH == 4);
  if (M == 0) return BDS_OK;
  BDS_REQUIRE(means2d && conics && colors && opacities && isect_offsets && flatten_ids && alphas && last_ids &&
              v_render && v_alphas && v_means2d && v_conics && v_colors && v_opacities);
  BDS_REQUIRE((reinterpret_cast<uintptr_t>(means2d) & 7u) == 0);
  const dim3 grid((unsigned)(C * tile_w * tile_h));
  hipStream_t st = as_stream(stream);
#define BDS_BWD_ARGS                                                                                                   \
  C, N, M, means2d, conics, colors, opacities, backgrounds, W, H, tile_w, tile_h, isect_offsets, flatten_ids, alphas,  \
      last_ids, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, tile_order
#define BDS_BWD(ch, ab) hipLaunchKernelGGL((rasterize_bwd_wave_kernel<ch, ab>), grid, dim3(kWave), 0, st, BDS_BWD_ARGS)
  if (v_means2d_abs) {
    if (CH == 1) BDS_BWD(1, true);
    else if (CH == 3) BDS_BWD(3, true);
    else BDS_BWD(4, true);
  } else {
    if (CH == 1) BDS_BWD(1, false);
    else if (CH == 3) BDS_BWD(3, false);
    else BDS_BWD(4, false);
  }
#undef BDS_BWD
#undef BDS_BWD_ARGS
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_rasterize_bwd_schedule(int C, int W, int H, int tile_size, int tile_w, int tile_h,
                                          const int32_t *isect_offsets, const int32_t *last_ids, int32_t *tile_order,
                                          bds_stream_t stream) {
  BDS_REQUIRE(C >= 1 && W > 0 && H > 0);
  BDS_REQUIRE(tile_size == kTile);
  BDS_REQUIRE(tile_w == (W + kTile - 1) / kTile && tile_h == (H + kTile - 1) / kTile);
  BDS_REQUIRE(isect_offsets && last_ids && tile_order);
  const int total = C * tile_w * tile_h;
  hipStream_t st = as_stream(stream);
  int32_t *work = tile_order + total;   // second half of the caller's buffer
  constexpr int per_block = kRastBlock / kWave;
  hipLaunchKernelGGL(tile_work_kernel, dim3((unsigned)((total + per_block - 1) / per_block)), dim3(kRastBlock), 0, st, C, W,
                     H, tile_w, tile_h, isect_offsets, last_ids, work);
  hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(kSchedThreads), 0, st, total, work, tile_order);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
