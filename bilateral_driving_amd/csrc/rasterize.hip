// K7/K8: per-pixel front-to-back alpha compositing, forward and backward.
// rasterize_to_pixels stage of gsplat.rendering.rasterization as called at
// /root/reference/project/models/trainers/base.py:393-408 (render_mode "RGB+ED" -> CH = 4,
// viewer "RGB" -> CH = 3).
//
// Mapping (gfx950): one workgroup of 256 threads = four wave64 per 16x16 tile; wave w owns the
// 16x4 pixel strip of rows 4w..4w+3, so image rows are written as 64-byte coalesced segments.
// The tile's depth-ordered Gaussians are staged through LDS in chunks of 256 (one gathered
// Gaussian per thread), then every lane walks the chunk reading LDS at a wave-uniform address
// (broadcast reads).  Early termination: a lane stops at T*(1-a) <= 1e-4, a wave skips the rest
// of a chunk once all 64 lanes are done (ballot), the workgroup stops fetching once all four
// waves are done.  Workgroup ids are remapped so that each XCD rasterises one contiguous band of
// the image (its private L2 then serves the re-reads of Gaussians shared by neighbouring tiles).
#include "bds_common.h"
#include "gs_math.h"

namespace bds {

constexpr int kTile = 16;
constexpr int kRastBlock = kTile * kTile;  // 256

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
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <int CH>
struct Staged {
  float4 a;  // x, y, conic.a, conic.b
  float4 b;  // conic.c, opacity, col0, col1
  float4 c;  // col2, col3, -, -
};

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

template <int CH>
__global__ __launch_bounds__(kRastBlock) void rasterize_fwd_kernel(
    int C, int64_t N, int64_t M, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int W,
    int H, int tile_w, int tile_h, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids,
    float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids) {
  __shared__ float4 sA[kRastBlock], sB[kRastBlock], sC[kRastBlock];
  const int n_tiles = tile_w * tile_h;
  const int item = xcd_contiguous(blockIdx.x, C * n_tiles);
  const int cam = item / n_tiles, tile = item - cam * n_tiles;
  const int ty = tile / tile_w, tx = tile - ty * tile_w;
  const int tid = threadIdx.x;
  const int i = ty * kTile + tid / kTile, j = tx * kTile + tid % kTile;
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const bool inside = i < H && j < W;
  bool done = !inside;
  const int start = offsets[item];
  const int end = (item == C * n_tiles - 1) ? (int)M : offsets[item + 1];
  const int nbatch = (end - start + kRastBlock - 1) / kRastBlock;
  float T = 1.f;
  int cur = 0;
  float out[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nbatch; b++) {
    if (__syncthreads_and(done)) break;
    const int bstart = start + b * kRastBlock;
    if (bstart + tid < end) {
      float4 A, B, Cc;
      stage_gaussian<CH>(flatten_ids[bstart + tid], means2d, conics, colors, opacities, A, B, Cc);
      sA[tid] = A; sB[tid] = B;
      if (CH > 2) sC[tid] = Cc;
    }
    __syncthreads();
    if (__all(done)) continue;  // this wave is finished; it only keeps helping with the staging
    const int bs = min(kRastBlock, end - bstart);
    for (int t = 0; t < bs && !done; t++) {
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px, dy = A.y - py;
      const float sigma = 0.5f * (A.z * dx * dx + B.x * dy * dy) + A.w * dx * dy;
      const float alpha = fminf(kAlphaMax, B.y * __expf(-sigma));
      if (sigma < 0.f || alpha < kAlphaMin) continue;
      const float nT = T * (1.f - alpha);
      if (nT <= kTStop) { done = true; break; }
      const float vis = alpha * T;
      out[0] += B.z * vis;
      if (CH > 1) out[1] += B.w * vis;
      if (CH > 2) {
        const float4 Cc = sC[t];
        out[2] += Cc.x * vis;
        if (CH > 3) out[3] += Cc.y * vis;
      }
      cur = bstart + t;
      T = nT;
    }
  }
  if (inside) {
    const int64_t pix = ((int64_t)cam * H + i) * W + j;
    alphas[pix] = 1.f - T;
    last_ids[pix] = cur;
    float *r = render + pix * CH;
#pragma unroll
    for (int k = 0; k < CH; k++) r[k] = backgrounds ? out[k] + T * backgrounds[cam * CH + k] : out[k];
  }
}

template <int CH, bool ABS>
__global__ __launch_bounds__(kRastBlock) void rasterize_bwd_kernel(
    int C, int64_t N, int64_t M, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int W,
    int H, int tile_w, int tile_h, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids, const float *__restrict__ v_render,
    const float *__restrict__ v_alphas, float *__restrict__ v_means2d, float *__restrict__ v_means2d_abs,
    float *__restrict__ v_conics, float *__restrict__ v_colors, float *__restrict__ v_opacities) {
  __shared__ float4 sA[kRastBlock], sB[kRastBlock], sC[kRastBlock];
  __shared__ int32_t sId[kRastBlock];
  const int n_tiles = tile_w * tile_h;
  const int item = xcd_contiguous(blockIdx.x, C * n_tiles);
  const int cam = item / n_tiles, tile = item - cam * n_tiles;
  const int ty = tile / tile_w, tx = tile - ty * tile_w;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int i = ty * kTile + tid / kTile, j = tx * kTile + tid % kTile;
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const bool inside = i < H && j < W;
  const int start = offsets[item];
  const int end = (item == C * n_tiles - 1) ? (int)M : offsets[item + 1];
  if (end <= start) return;  // uniform
  const int nbatch = (end - start + kRastBlock - 1) / kRastBlock;
  const int64_t pix = ((int64_t)cam * H + (inside ? i : 0)) * W + (inside ? j : 0);
  const float T_final = inside ? 1.f - alphas[pix] : 1.f;
  float T = T_final;
  float buffer[4] = {0.f, 0.f, 0.f, 0.f};
  const int bin_final = inside ? last_ids[pix] : 0;
  float vr[4] = {0.f, 0.f, 0.f, 0.f};
  float vra = 0.f;
  if (inside) {
#pragma unroll
    for (int k = 0; k < CH; k++) vr[k] = v_render[pix * CH + k];
    vra = v_alphas[pix];
  }
  float bgdot = 0.f;  // sum_k bg[k] * v_render[k]
  if (backgrounds) {
#pragma unroll
    for (int k = 0; k < CH; k++) bgdot += backgrounds[cam * CH + k] * vr[k];
  }
  const int wave_bin_final = wave_max_i32(bin_final);
  for (int b = 0; b < nbatch; b++) {
    __syncthreads();
    const int batch_end = end - 1 - kRastBlock * b;
    const int bs = min(kRastBlock, batch_end + 1 - start);
    const int idx = batch_end - tid;
    if (idx >= start) {
      const int32_t g = flatten_ids[idx];
      float4 A, B, Cc;
      stage_gaussian<CH>(g, means2d, conics, colors, opacities, A, B, Cc);
      sId[tid] = g; sA[tid] = A; sB[tid] = B;
      if (CH > 2) sC[tid] = Cc;
    }
    __syncthreads();
    for (int t = max(0, batch_end - wave_bin_final); t < bs; t++) {
      bool valid = inside && (batch_end - t <= bin_final);
      const float4 A = sA[t], B = sB[t];
      const float dx = A.x - px, dy = A.y - py;
      const float sigma = 0.5f * (A.z * dx * dx + B.x * dy * dy) + A.w * dx * dy;
      const float vis = __expf(-sigma);
      const float opac = B.y;
      const float alpha = fminf(kAlphaMax, opac * vis);
      if (sigma < 0.f || alpha < kAlphaMin) valid = false;
      if (!__any(valid)) continue;
      float col[4] = {B.z, B.w, 0.f, 0.f};
      if (CH > 2) { const float4 Cc = sC[t]; col[2] = Cc.x; col[3] = Cc.y; }
      float g_col[4] = {0.f, 0.f, 0.f, 0.f};
      float g_conic[3] = {0.f, 0.f, 0.f};
      float g_xy[2] = {0.f, 0.f}, g_xy_abs[2] = {0.f, 0.f};
      float g_opac = 0.f;
      if (valid) {
        const float ra = 1.f / (1.f - alpha);
        T *= ra;
        const float fac = alpha * T;
        float v_alpha = 0.f;
#pragma unroll
        for (int k = 0; k < CH; k++) {
          g_col[k] = fac * vr[k];
          v_alpha += (col[k] * T - buffer[k] * ra) * vr[k];
        }
        v_alpha += T_final * ra * vra;
        if (backgrounds) v_alpha += -T_final * ra * bgdot;
        if (opac * vis <= kAlphaMax) {
          const float v_sigma = -opac * vis * v_alpha;
          g_conic[0] = 0.5f * v_sigma * dx * dx;
          g_conic[1] = v_sigma * dx * dy;
          g_conic[2] = 0.5f * v_sigma * dy * dy;
          g_xy[0] = v_sigma * (A.z * dx + A.w * dy);
          g_xy[1] = v_sigma * (A.w * dx + B.x * dy);
          if (ABS) { g_xy_abs[0] = fabsf(g_xy[0]); g_xy_abs[1] = fabsf(g_xy[1]); }
          g_opac = vis * v_alpha;
        }
#pragma unroll
        for (int k = 0; k < CH; k++) buffer[k] += col[k] * fac;
      }
      // wave64 reduction (DPP), then one atomic per value per wave
#pragma unroll
      for (int k = 0; k < CH; k++) g_col[k] = wave_sum_to_lane63(g_col[k]);
      g_conic[0] = wave_sum_to_lane63(g_conic[0]);
      g_conic[1] = wave_sum_to_lane63(g_conic[1]);
      g_conic[2] = wave_sum_to_lane63(g_conic[2]);
      g_xy[0] = wave_sum_to_lane63(g_xy[0]);
      g_xy[1] = wave_sum_to_lane63(g_xy[1]);
      if (ABS) { g_xy_abs[0] = wave_sum_to_lane63(g_xy_abs[0]); g_xy_abs[1] = wave_sum_to_lane63(g_xy_abs[1]); }
      g_opac = wave_sum_to_lane63(g_opac);
      if (lane == kWave - 1) {
        const int64_t g = sId[t];
#pragma unroll
        for (int k = 0; k < CH; k++) atomicAdd(v_colors + g * CH + k, g_col[k]);
        atomicAdd(v_conics + g * 3, g_conic[0]);
        atomicAdd(v_conics + g * 3 + 1, g_conic[1]);
        atomicAdd(v_conics + g * 3 + 2, g_conic[2]);
        atomicAdd(v_means2d + g * 2, g_xy[0]);
        atomicAdd(v_means2d + g * 2 + 1, g_xy[1]);
        if (ABS) {
          atomicAdd(v_means2d_abs + g * 2, g_xy_abs[0]);
          atomicAdd(v_means2d_abs + g * 2 + 1, g_xy_abs[1]);
        }
        atomicAdd(v_opacities + g, g_opac);
      }
    }
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
  const dim3 grid((unsigned)(C * tile_w * tile_h)), block(kRastBlock);
  hipStream_t st = as_stream(stream);
#define BDS_FWD(ch)                                                                                                    \
  hipLaunchKernelGGL((rasterize_fwd_kernel<ch>), grid, block, 0, st, C, N, M, means2d, conics, colors, opacities,      \
                     backgrounds, W, H, tile_w, tile_h, isect_offsets, flatten_ids, render, alphas, last_ids)
  if (CH == 1) BDS_FWD(1);
  else if (CH == 3) BDS_FWD(3);
  else BDS_FWD(4);
#undef BDS_FWD
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_rasterize_bwd(int C, int64_t N, int64_t M, int CH, const float *means2d, const float *conics,
                                 const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                                 int tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                                 const int32_t *flatten_ids, const float *alphas, const int32_t *last_ids,
                                 const float *v_render, const float *v_alphas, float *v_means2d, float *v_means2d_abs,
                                 float *v_conics, float *v_colors, float *v_opacities, bds_stream_t stream) {
  BDS_REQUIRE(C >= 1 && N >= 0 && M >= 0 && W > 0 && H > 0);
  BDS_REQUIRE(tile_size == kTile);
  BDS_REQUIRE(tile_w == (W + kTile - 1) / kTile && tile_h == (H + kTile - 1) / kTile);
  BDS_REQUIRE(CH == 1 || CH == 3 || CH == 4);
  if (M == 0) return BDS_OK;
  BDS_REQUIRE(means2d && conics && colors && opacities && isect_offsets && flatten_ids && alphas && last_ids &&
              v_render && v_alphas && v_means2d && v_conics && v_colors && v_opacities);
  BDS_REQUIRE((reinterpret_cast<uintptr_t>(means2d) & 7u) == 0);
  const dim3 grid((unsigned)(C * tile_w * tile_h)), block(kRastBlock);
  hipStream_t st = as_stream(stream);
#define BDS_BWD(ch, ab)                                                                                                \
  hipLaunchKernelGGL((rasterize_bwd_kernel<ch, ab>), grid, block, 0, st, C, N, M, means2d, conics, colors, opacities,  \
                     backgrounds, W, H, tile_w, tile_h, isect_offsets, flatten_ids, alphas, last_ids, v_render,        \
                     v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities)
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
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
