// Photometric L1 of the training step: mean |a - b| over an image, forward and backward.
// models/trainers/base.py:518-529 (rgb loss, losses.rgb.w = 0.8 L1 part) -- the step right after the hot path
// (SURVEY.md 8f rank 1).  One streaming pass each way instead of the ~8 framework kernels of
// (a - b).abs().mean() and its autograd graph; HBM-bound (8 B/element forward, 12 B/element backward).
#include "bds_common.h"

namespace bds {

constexpr int kLossBlock = 256;

__global__ __launch_bounds__(kLossBlock) void l1_mean_fwd_kernel(int64_t n4, int64_t n, const float4 *__restrict__ a4,
                                                                const float4 *__restrict__ b4, const float *__restrict__ a,
                                                                const float *__restrict__ b, float scale,
                                                                float *__restrict__ out) {
  __shared__ float red[kLossBlock / kWave];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * kLossBlock;
  int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // 8 independent 16-byte loads in flight per lane
    float4 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { x[u] = a4[i + u * stride]; y[u] = b4[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; u++)
      s += fabsf(x[u].x - y[u].x) + fabsf(x[u].y - y[u].y) + fabsf(x[u].z - y[u].z) + fabsf(x[u].w - y[u].w);
  }
  for (; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    s += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {  // tail (n not a multiple of 4)
    const int64_t i = n4 * 4 + threadIdx.x;
    s += fabsf(a[i] - b[i]);
  }
  s = wave_sum_to_lane63(s);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1) red[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kLossBlock / kWave; w++) t += red[w];
    atomicAdd(out, t * scale);
  }
}

__global__ __launch_bounds__(kLossBlock) void l1_mean_bwd_kernel(int64_t n, const float *__restrict__ a,
                                                                const float *__restrict__ b, float scale,
                                                                const float *__restrict__ v_out, float *__restrict__ v_a) {
  const float g = *v_out * scale;
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLossBlock) {
    const float d = a[i] - b[i];
    v_a[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);   // torch: sign(0) = 0
  }
}

// ---- SSIM (11x11 Gaussian window, sigma 1.5, VALID region, K = (0.01, 0.03), data range 1) -----------------------
// pytorch_msssim 1.0.0 `SSIM(data_range=1.0, size_average=True, channel=3)` as used at models/trainers/base.py:114,541
// (external package, parity unpinned: oracle/loss_oracle.py).  Images are [H,W,CH] as the path produces them (the
// reference permutes to [1,CH,H,W] first).  One workgroup = one 16x16 tile of OUTPUT pixels of one channel: the
// 26x26 input patch is staged in LDS, the five windowed moments are taken separably (row pass into LDS, column pass in
// registers), and next to the SSIM value the three partial derivatives with respect to the moments of `pred` that the
// backward needs are written out, so that the backward is ONE more separable pass (the adjoint of the valid filter).
constexpr int kSsimWin = 11, kSsimTile = 16, kSsimIn = kSsimTile + kSsimWin - 1;  // 26
struct SsimWindow { float w[kSsimWin]; };

__global__ __launch_bounds__(kSsimTile * kSsimTile) void ssim_fwd_kernel(int H, int W, int CH, const float *__restrict__ x,
                                                                        const float *__restrict__ y, SsimWindow win, float C1,
                                                                        float C2, float scale, float *__restrict__ out,
                                                                        float *__restrict__ dmaps) {
  __shared__ float sx[kSsimIn][kSsimIn + 1], sy[kSsimIn][kSsimIn + 1];
  __shared__ float hb[5][kSsimIn][kSsimTile + 1];
  __shared__ float red[kSsimTile * kSsimTile / kWave];
  const int Ho = H - (kSsimWin - 1), Wo = W - (kSsimWin - 1);
  const int c = blockIdx.z, oi0 = blockIdx.y * kSsimTile, oj0 = blockIdx.x * kSsimTile, tid = threadIdx.x;
  for (int e = tid; e < kSsimIn * kSsimIn; e += kSsimTile * kSsimTile) {
    const int r = e / kSsimIn, q = e - r * kSsimIn, i = oi0 + r, j = oj0 + q;
    const bool in = i < H && j < W;
    const int64_t o = ((int64_t)i * W + j) * CH + c;
    sx[r][q] = in ? x[o] : 0.f;
    sy[r][q] = in ? y[o] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSsimIn * kSsimTile; e += kSsimTile * kSsimTile) {   // row pass
    const int r = e / kSsimTile, j = e - r * kSsimTile;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int b = 0; b < kSsimWin; b++) {
      const float xv = sx[r][j + b], yv = sy[r][j + b], w = win.w[b];
      a0 += w * xv; a1 += w * yv; a2 += w * xv * xv; a3 += w * yv * yv; a4 += w * xv * yv;
    }
    hb[0][r][j] = a0; hb[1][r][j] = a1; hb[2][r][j] = a2; hb[3][r][j] = a3; hb[4][r][j] = a4;
  }
  __syncthreads();
  const int ti = tid / kSsimTile, tj = tid - ti * kSsimTile;
  float m1 = 0.f, m2 = 0.f, e1 = 0.f, e2 = 0.f, e12 = 0.f;
#pragma unroll
  for (int a = 0; a < kSsimWin; a++) {   // column pass
    const float w = win.w[a];
    m1 += w * hb[0][ti + a][tj]; m2 += w * hb[1][ti + a][tj];
    e1 += w * hb[2][ti + a][tj]; e2 += w * hb[3][ti + a][tj]; e12 += w * hb[4][ti + a][tj];
  }
  const int oi = oi0 + ti, oj = oj0 + tj;
  const bool valid = oi < Ho && oj < Wo;
  const float s1 = e1 - m1 * m1, s2 = e2 - m2 * m2, s12 = e12 - m1 * m2;
  const float A = 2.f * m1 * m2 + C1, B = m1 * m1 + m2 * m2 + C1, Cn = 2.f * s12 + C2, D = s1 + s2 + C2;
  const float lum = A / B, cs = Cn / D;
  float S = valid ? lum * cs : 0.f;
  if (valid && dmaps != nullptr) {
    // S as a function of (m2, e2, e12) with s2 = e2 - m2^2, s12 = e12 - m1 m2
    const float d_lum = (2.f * m1 * B - A * 2.f * m2) / (B * B);
    const float d_cs = (-2.f * m1 * D + 2.f * m2 * Cn) / (D * D);
    const int64_t plane = (int64_t)Ho * Wo, o = ((int64_t)c * Ho + oi) * Wo + oj;
    dmaps[o] = d_lum * cs + lum * d_cs;                       // dS / d mu_pred
    dmaps[(int64_t)CH * plane + o] = -lum * Cn / (D * D);     // dS / d E[pred^2]
    dmaps[2 * (int64_t)CH * plane + o] = lum * 2.f / D;       // dS / d E[gt * pred]
  }
  S = wave_sum_to_lane63(S);
  if ((tid & (kWave - 1)) == kWave - 1) red[tid / kWave] = S;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kSsimTile * kSsimTile / kWave; w++) t += red[w];
    atomicAdd(out, t * scale);
  }
}

__global__ __launch_bounds__(kSsimTile * kSsimTile) void ssim_bwd_kernel(int H, int W, int CH, const float *__restrict__ x,
                                                                        const float *__restrict__ y, SsimWindow win, float scale,
                                                                        const float *__restrict__ v_out,
                                                                        const float *__restrict__ dmaps, float *__restrict__ v_y) {
  __shared__ float sd[3][kSsimIn][kSsimIn + 1];
  __shared__ float hb[3][kSsimIn][kSsimTile + 1];
  const int Ho = H - (kSsimWin - 1), Wo = W - (kSsimWin - 1);
  const int c = blockIdx.z, i0 = blockIdx.y * kSsimTile, j0 = blockIdx.x * kSsimTile, tid = threadIdx.x;
  const int64_t plane = (int64_t)Ho * Wo;
  // output pixels whose window covers input pixel (i,j): (i-a, j-b), a, b = 0..10
  for (int e = tid; e < kSsimIn * kSsimIn; e += kSsimTile * kSsimTile) {
    const int r = e / kSsimIn, q = e - r * kSsimIn, oi = i0 - (kSsimWin - 1) + r, oj = j0 - (kSsimWin - 1) + q;
    const bool in = oi >= 0 && oi < Ho && oj >= 0 && oj < Wo;
    const int64_t o = ((int64_t)c * Ho + (in ? oi : 0)) * Wo + (in ? oj : 0);
#pragma unroll
    for (int k = 0; k < 3; k++) sd[k][r][q] = in ? dmaps[(int64_t)k * CH * plane + o] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSsimIn * kSsimTile; e += kSsimTile * kSsimTile) {
    const int r = e / kSsimTile, j = e - r * kSsimTile;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int b = 0; b < kSsimWin; b++) {
      const float w = win.w[b];
      const int q = j + (kSsimWin - 1) - b;
      a0 += w * sd[0][r][q]; a1 += w * sd[1][r][q]; a2 += w * sd[2][r][q];
    }
    hb[0][r][j] = a0; hb[1][r][j] = a1; hb[2][r][j] = a2;
  }
  __syncthreads();
  const int ti = tid / kSsimTile, tj = tid - ti * kSsimTile, i = i0 + ti, j = j0 + tj;
  if (i >= H || j >= W) return;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int a = 0; a < kSsimWin; a++) {
    const float w = win.w[a];
    const int r = ti + (kSsimWin - 1) - a;
    c0 += w * hb[0][r][tj]; c1 += w * hb[1][r][tj]; c2 += w * hb[2][r][tj];
  }
  const int64_t o = ((int64_t)i * W + j) * CH + c;
  v_y[o] = v_out[0] * scale * (c0 + 2.f * y[o] * c1 + x[o] * c2);
}

// ---- per-pixel terms of the reference's image loss ---------------------------------------------------------------
// BasicTrainer.compute_losses, models/trainers/base.py:518-565 with the loss functions of :230-250 (models/losses.py:82-84
// binary_cross_entropy, :92-178 DepthLoss(normalize=False, use_inverse_depth=False, reduction="mean_on_hit")):
//   valid = 1 - egocar (or 1);  rgb: |pixels*valid - rgb*valid|.mean();  mask: BCE(opacity*valid, (1-sky)*valid).mean()
//   depth: hit = (lidar > 0)*valid, pred = depth*hit, gt = lidar*hit, over {gt > 0.01, gt < max_depth, pred > 1e-4}: L1 or L2 mean
// One pass computes the four sums (three numerators + the depth count); the backward is one more pass.  PINNED by
// tests/golden/pixel_loss_*.npz (generated with the reference's own functions).
struct PixelLossArgs {
  const float *rgb, *pixels, *opacity, *sky, *depth, *lidar, *egocar;
  int depth_l2;
  float max_depth;
};

__device__ __forceinline__ float bce_log(float v) { return fmaxf(logf(v), -100.f); }   // torch clamps the logs at -100

__global__ __launch_bounds__(kLossBlock) void pixel_loss_fwd_kernel(int64_t P, PixelLossArgs a, float *__restrict__ sums) {
  __shared__ float red[4][kLossBlock / kWave];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < P; i += (int64_t)gridDim.x * kLossBlock) {
    const float valid = a.egocar ? 1.f - a.egocar[i] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; c++) s[0] += fabsf(a.pixels[i * 3 + c] * valid - a.rgb[i * 3 + c] * valid);
    if (a.opacity) {
      const float x = a.opacity[i] * valid, t = (1.f - a.sky[i]) * valid;
      s[1] += (t - 1.f) * bce_log(1.f - x) - t * bce_log(x);
    }
    if (a.depth) {
      const float l = a.lidar[i], hit = (l > 0.f ? 1.f : 0.f) * valid;
      const float pred = a.depth[i] * hit, gt = l * hit;
      if (gt > 0.01f && gt < a.max_depth && pred > 0.0001f) {
        const float e = pred - gt;
        s[2] += a.depth_l2 ? e * e : fabsf(e);
        s[3] += 1.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float t = wave_sum_to_lane63(s[k]);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) red[k][threadIdx.x / kWave] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
    for (int w = 0; w < kLossBlock / kWave; w++) t += red[threadIdx.x][w];
    if (t != 0.f) atomicAdd(sums + threadIdx.x, t);
  }
}

// terms[0..2] = weighted rgb / mask / depth losses (depth: 0/0 = NaN when no lidar return is valid, as torch's empty mean)
__global__ void pixel_loss_finalize_kernel(int64_t P, const float *__restrict__ sums, float w_rgb, float w_mask, float w_depth,
                                           int has_mask, int has_depth, float *__restrict__ terms) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  terms[0] = w_rgb * (sums[0] / (3.f * (float)P));
  terms[1] = has_mask ? w_mask * (sums[1] / (float)P) : 0.f;
  terms[2] = has_depth ? w_depth * (sums[2] / sums[3]) : 0.f;
}

__global__ __launch_bounds__(kLossBlock) void pixel_loss_bwd_kernel(int64_t P, PixelLossArgs a, const float *__restrict__ sums,
                                                                   const float *__restrict__ v_terms, float w_rgb, float w_mask,
                                                                   float w_depth, float *__restrict__ v_rgb,
                                                                   float *__restrict__ v_opacity, float *__restrict__ v_depth) {
  const float g_rgb = v_terms[0] * w_rgb / (3.f * (float)P), g_mask = v_terms[1] * w_mask / (float)P;
  const float g_depth = a.depth ? v_terms[2] * w_depth / sums[3] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < P; i += (int64_t)gridDim.x * kLossBlock) {
    const float valid = a.egocar ? 1.f - a.egocar[i] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float d = a.rgb[i * 3 + c] * valid - a.pixels[i * 3 + c] * valid;
      v_rgb[i * 3 + c] = (d > 0.f ? g_rgb : (d < 0.f ? -g_rgb : 0.f)) * valid;
    }
    if (v_opacity) {
      float v = 0.f;
      if (a.opacity) {
        const float x = a.opacity[i] * valid, t = (1.f - a.sky[i]) * valid;
        v = g_mask * (x - t) / fmaxf((1.f - x) * x, 1e-12f) * valid;   // torch's binary_cross_entropy backward
      }
      v_opacity[i] = v;
    }
    if (v_depth) {
      float v = 0.f;
      if (a.depth) {
        const float l = a.lidar[i], hit = (l > 0.f ? 1.f : 0.f) * valid;
        const float pred = a.depth[i] * hit, gt = l * hit;
        if (gt > 0.01f && gt < a.max_depth && pred > 0.0001f) {
          const float e = pred - gt;
          v = g_depth * (a.depth_l2 ? 2.f * e : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f))) * hit;
        }
      }
      v_depth[i] = v;
    }
  }
}

// ---- regularisers of the reference's image loss (models/trainers/base.py:566-585, 638-659) ---------------------------
//   opacity entropy     p = clamp(opacity, 1e-6, 1 - 1e-6);  (-p log p).mean()                                        (:566-572)
//   inverse-depth smoothness (kornia.losses.inverse_depth_smoothness_loss -- external package, PARITY UNPINNED, restated from its
//     published definition):  id = 1 / (depth + 1e-5);  mean |(id[x] - id[x+1]) * exp(-mean_c |img[x] - img[x+1]|)| over (H, W-1)
//     + the same along y over (H-1, W); the reference repeats id to three equal channels (:575-582), which leaves the mean unchanged
//   dynamic-region L1   over the pixels with Dynamic_opacity > 0.2 (detached) and valid: |pixels*valid - rgb*valid|.mean()  (:638-651)
// One pass computes the sums (sums[0] entropy, [1] x-smoothness, [2] y-smoothness, [3] masked L1, [4] masked pixel count);
// the backward is one more (gather-style: a pixel's depth gradient collects its four neighbour differences).
struct RegLossArgs {
  const float *opacity, *depth, *pixels, *rgb, *dyn_opacity, *egocar;
  int H, W;
  float dyn_threshold;
};

__device__ __forceinline__ float inv_depth(const RegLossArgs &a, int y, int x) { return 1.f / (a.depth[(int64_t)y * a.W + x] + 1e-5f); }
// exp(-mean_c |img[p] - img[q]|)
__device__ __forceinline__ float edge_weight(const RegLossArgs &a, int64_t p, int64_t q) {
  const float d = fabsf(a.pixels[p * 3] - a.pixels[q * 3]) + fabsf(a.pixels[p * 3 + 1] - a.pixels[q * 3 + 1]) +
                  fabsf(a.pixels[p * 3 + 2] - a.pixels[q * 3 + 2]);
  return expf(-(d / 3.f));
}

__global__ __launch_bounds__(kLossBlock) void reg_loss_fwd_kernel(RegLossArgs a, float *__restrict__ sums) {
  __shared__ float red[5][kLossBlock / kWave];
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int64_t P = (int64_t)a.H * a.W;
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < P; i += (int64_t)gridDim.x * kLossBlock) {
    const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
    if (a.opacity) {
      const float p = fminf(fmaxf(a.opacity[i], 1e-6f), 1.f - 1e-6f);
      s[0] += -p * logf(p);
    }
    if (a.depth) {
      const float id = inv_depth(a, y, x);
      if (x + 1 < a.W) s[1] += fabsf((id - inv_depth(a, y, x + 1)) * edge_weight(a, i, i + 1));
      if (y + 1 < a.H) s[2] += fabsf((id - inv_depth(a, y + 1, x)) * edge_weight(a, i, i + a.W));
    }
    if (a.dyn_opacity && a.dyn_opacity[i] > a.dyn_threshold) {
      const float valid = a.egocar ? 1.f - a.egocar[i] : 1.f;
      if (valid != 0.f) {   // dynamic_pred_mask & valid_loss_mask.bool()
#pragma unroll
        for (int c = 0; c < 3; c++) s[3] += fabsf(a.pixels[i * 3 + c] * valid - a.rgb[i * 3 + c] * valid);
        s[4] += 1.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const float t = wave_sum_to_lane63(s[k]);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) red[k][threadIdx.x / kWave] = t;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
    for (int w = 0; w < kLossBlock / kWave; w++) t += red[threadIdx.x][w];
    if (t != 0.f) atomicAdd(sums + threadIdx.x, t);
  }
}

// terms[0..2] = entropy, smoothness, dynamic-region L1 (0 when the mask is empty: the reference then adds no term)
__global__ void reg_loss_finalize_kernel(int H, int W, const float *__restrict__ sums, int has_entropy, int has_smooth, int has_dyn,
                                         float *__restrict__ terms) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float P = (float)H * (float)W;
  terms[0] = has_entropy ? sums[0] / P : 0.f;
  float sm = 0.f;
  if (has_smooth) {   // a mean over an empty tensor (W = 1 or H = 1) is NaN in the reference, too
    sm = sums[1] / ((float)H * (float)(W - 1)) + sums[2] / ((float)(H - 1) * (float)W);
  }
  terms[1] = sm;
  terms[2] = (has_dyn && sums[4] > 0.f) ? sums[3] / (3.f * sums[4]) : 0.f;
}

__global__ __launch_bounds__(kLossBlock) void reg_loss_bwd_kernel(RegLossArgs a, const float *__restrict__ sums,
                                                                 const float *__restrict__ v_terms, float *__restrict__ v_opacity,
                                                                 float *__restrict__ v_depth, float *__restrict__ v_rgb) {
  const int64_t P = (int64_t)a.H * a.W;
  const float g_ent = v_terms[0] / (float)P;
  const float gx = v_terms[1] / ((float)a.H * (float)(a.W - 1)), gy = v_terms[1] / ((float)(a.H - 1) * (float)a.W);
  const float g_dyn = sums[4] > 0.f ? v_terms[2] / (3.f * sums[4]) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kLossBlock + threadIdx.x; i < P; i += (int64_t)gridDim.x * kLossBlock) {
    const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
    if (v_opacity) {
      float v = 0.f;
      if (a.opacity) {
        const float o = a.opacity[i];
        if (o >= 1e-6f && o <= 1.f - 1e-6f) v = -g_ent * (logf(o) + 1.f);   // torch.clamp passes the gradient on the closed interval
      }
      v_opacity[i] = v;
    }
    if (v_depth) {
      float v = 0.f;
      if (a.depth) {
        const float id = inv_depth(a, y, x);
        float vid = 0.f;   // d / d id of the four differences this pixel takes part in; |t| has derivative sign(t), sign(0) = 0
        auto sgn = [](float t) { return t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); };
        if (x + 1 < a.W) { const float w = edge_weight(a, i, i + 1); vid += gx * sgn((id - inv_depth(a, y, x + 1)) * w) * w; }
        if (x > 0) { const float w = edge_weight(a, i - 1, i); vid -= gx * sgn((inv_depth(a, y, x - 1) - id) * w) * w; }
        if (y + 1 < a.H) { const float w = edge_weight(a, i, i + a.W); vid += gy * sgn((id - inv_depth(a, y + 1, x)) * w) * w; }
        if (y > 0) { const float w = edge_weight(a, i - a.W, i); vid -= gy * sgn((inv_depth(a, y - 1, x) - id) * w) * w; }
        v = -vid * id * id;
      }
      v_depth[i] = v;
    }
    if (v_rgb) {
      float v[3] = {0.f, 0.f, 0.f};
      if (a.dyn_opacity && a.dyn_opacity[i] > a.dyn_threshold) {
        const float valid = a.egocar ? 1.f - a.egocar[i] : 1.f;
        if (valid != 0.f) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float d = a.rgb[i * 3 + c] * valid - a.pixels[i * 3 + c] * valid;
            v[c] = (d > 0.f ? g_dyn : (d < 0.f ? -g_dyn : 0.f)) * valid;
          }
        }
      }
      v_rgb[i * 3] = v[0]; v_rgb[i * 3 + 1] = v[1]; v_rgb[i * 3 + 2] = v[2];
    }
  }
}

}  // namespace bds

using namespace bds;

extern "C" int bds_l1_mean_fwd(int64_t n, const float *a, const float *b, float *out, bds_stream_t stream) {
  BDS_REQUIRE(n >= 0 && out);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(a && b);
  const bool vec = aligned16(a) && aligned16(b);
  const int64_t n4 = vec ? n / 4 : 0;
  int64_t blocks = cdiv(n4 > 0 ? n4 : 1, kLossBlock * 4);
  if (blocks > 512) blocks = 512;   // one atomic per workgroup on ONE address: keep them few
  BDS_REQUIRE(vec);  // 16-byte aligned buffers (torch allocations are)
  hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, as_stream(stream), n4, n,
                     reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), a, b, 1.0f / (float)n, out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_l1_mean_bwd(int64_t n, const float *a, const float *b, const float *v_out, float *v_a,
                               bds_stream_t stream) {
  BDS_REQUIRE(n >= 0);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(a && b && v_out && v_a);
  int64_t blocks = cdiv(n, kLossBlock * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, as_stream(stream), n, a, b,
                     1.0f / (float)n, v_out, v_a);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static SsimWindow ssim_window() {   // _fspecial_gauss_1d(11, 1.5) in float arithmetic
  SsimWindow w;
  float sum = 0.f;
  for (int i = 0; i < kSsimWin; i++) {
    const float cidx = (float)(i - kSsimWin / 2);
    w.w[i] = expf(-(cidx * cidx) / (2.f * 1.5f * 1.5f));
    sum += w.w[i];
  }
  for (int i = 0; i < kSsimWin; i++) w.w[i] /= sum;
  return w;
}

extern "C" size_t bds_ssim_workspace_bytes(int H, int W, int CH) {
  if (H < kSsimWin || W < kSsimWin || CH < 1) return 0;
  return sizeof(float) * 3 * (size_t)CH * (size_t)(H - kSsimWin + 1) * (size_t)(W - kSsimWin + 1);
}

extern "C" int bds_ssim_fwd(int H, int W, int CH, const float *target, const float *pred, float *ssim_out, void *ws,
                            size_t ws_bytes, bds_stream_t stream) {
  BDS_REQUIRE(H >= kSsimWin && W >= kSsimWin && CH >= 1 && CH <= 65535 && target && pred && ssim_out);
  if (ws != nullptr && ws_bytes < bds_ssim_workspace_bytes(H, W, CH)) return BDS_EWORKSPACE;
  const int Ho = H - kSsimWin + 1, Wo = W - kSsimWin + 1;
  const dim3 grid((unsigned)cdiv(Wo, kSsimTile), (unsigned)cdiv(Ho, kSsimTile), (unsigned)CH);
  const float scale = 1.0f / ((float)Ho * (float)Wo * (float)CH);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(kSsimTile * kSsimTile), 0, as_stream(stream), H, W, CH, target, pred,
                     ssim_window(), 0.01f * 0.01f, 0.03f * 0.03f, scale, ssim_out, static_cast<float *>(ws));
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_ssim_bwd(int H, int W, int CH, const float *target, const float *pred, const void *ws, size_t ws_bytes,
                            const float *v_ssim, float *v_pred, bds_stream_t stream) {
  BDS_REQUIRE(H >= kSsimWin && W >= kSsimWin && CH >= 1 && CH <= 65535 && target && pred && ws && v_ssim && v_pred);
  if (ws_bytes < bds_ssim_workspace_bytes(H, W, CH)) return BDS_EWORKSPACE;
  const int Ho = H - kSsimWin + 1, Wo = W - kSsimWin + 1;
  const dim3 grid((unsigned)cdiv(W, kSsimTile), (unsigned)cdiv(H, kSsimTile), (unsigned)CH);
  const float scale = 1.0f / ((float)Ho * (float)Wo * (float)CH);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kSsimTile * kSsimTile), 0, as_stream(stream), H, W, CH, target, pred,
                     ssim_window(), scale, v_ssim, static_cast<const float *>(ws), v_pred);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int pixel_loss_args(PixelLossArgs &a, int64_t P, const float *rgb, const float *pixels, const float *opacity,
                           const float *sky_masks, const float *depth, const float *lidar, const float *egocar, int depth_l2,
                           float max_depth) {
  BDS_REQUIRE(P >= 1 && rgb && pixels);
  BDS_REQUIRE((opacity == nullptr) == (sky_masks == nullptr));
  BDS_REQUIRE((depth == nullptr) == (lidar == nullptr));
  a.rgb = rgb; a.pixels = pixels; a.opacity = opacity; a.sky = sky_masks; a.depth = depth; a.lidar = lidar; a.egocar = egocar;
  a.depth_l2 = depth_l2 ? 1 : 0; a.max_depth = max_depth;
  return BDS_OK;
}

extern "C" int bds_pixel_loss_fwd(int64_t P, const float *rgb, const float *pixels, const float *opacity, const float *sky_masks,
                                  const float *depth, const float *lidar, const float *egocar, float w_rgb, float w_mask,
                                  float w_depth, int depth_l2, float max_depth, float *sums, float *terms, bds_stream_t stream) {
  PixelLossArgs a;
  int rc = pixel_loss_args(a, P, rgb, pixels, opacity, sky_masks, depth, lidar, egocar, depth_l2, max_depth);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(sums && terms);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(sums, 0, 4 * sizeof(float), st) != hipSuccess) return BDS_ELAUNCH;
  int64_t blocks = cdiv(P, kLossBlock * 4);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(pixel_loss_fwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, st, P, a, sums);
  hipLaunchKernelGGL(pixel_loss_finalize_kernel, dim3(1), dim3(kWave), 0, st, P, sums, w_rgb, w_mask, w_depth, opacity != nullptr,
                     depth != nullptr, terms);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_pixel_loss_bwd(int64_t P, const float *rgb, const float *pixels, const float *opacity, const float *sky_masks,
                                  const float *depth, const float *lidar, const float *egocar, float w_rgb, float w_mask,
                                  float w_depth, int depth_l2, float max_depth, const float *sums, const float *v_terms,
                                  float *v_rgb, float *v_opacity, float *v_depth, bds_stream_t stream) {
  PixelLossArgs a;
  int rc = pixel_loss_args(a, P, rgb, pixels, opacity, sky_masks, depth, lidar, egocar, depth_l2, max_depth);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(sums && v_terms && v_rgb);
  int64_t blocks = cdiv(P, kLossBlock * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pixel_loss_bwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, as_stream(stream), P, a, sums, v_terms,
                     w_rgb, w_mask, w_depth, v_rgb, v_opacity, v_depth);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int reg_loss_args(RegLossArgs &a, int H, int W, const float *opacity, const float *depth, const float *pixels, const float *rgb,
                         const float *dyn_opacity, const float *egocar, float dyn_threshold) {
  BDS_REQUIRE(H >= 1 && W >= 1);
  BDS_REQUIRE(depth == nullptr || pixels != nullptr);
  BDS_REQUIRE(dyn_opacity == nullptr || (pixels != nullptr && rgb != nullptr));
  a.opacity = opacity; a.depth = depth; a.pixels = pixels; a.rgb = rgb; a.dyn_opacity = dyn_opacity; a.egocar = egocar;
  a.H = H; a.W = W; a.dyn_threshold = dyn_threshold;
  return BDS_OK;
}

extern "C" int bds_reg_loss_fwd(int H, int W, const float *opacity, const float *depth, const float *pixels, const float *rgb,
                                const float *dyn_opacity, const float *egocar, float dyn_threshold, float *sums, float *terms,
                                bds_stream_t stream) {
  RegLossArgs a;
  int rc = reg_loss_args(a, H, W, opacity, depth, pixels, rgb, dyn_opacity, egocar, dyn_threshold);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(sums && terms);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(sums, 0, 5 * sizeof(float), st) != hipSuccess) return BDS_ELAUNCH;
  int64_t blocks = cdiv((int64_t)H * W, kLossBlock * 4);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(reg_loss_fwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, st, a, sums);
  hipLaunchKernelGGL(reg_loss_finalize_kernel, dim3(1), dim3(kWave), 0, st, H, W, sums, opacity != nullptr, depth != nullptr,
                     dyn_opacity != nullptr, terms);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_reg_loss_bwd(int H, int W, const float *opacity, const float *depth, const float *pixels, const float *rgb,
                                const float *dyn_opacity, const float *egocar, float dyn_threshold, const float *sums,
                                const float *v_terms, float *v_opacity, float *v_depth, float *v_rgb, bds_stream_t stream) {
  RegLossArgs a;
  int rc = reg_loss_args(a, H, W, opacity, depth, pixels, rgb, dyn_opacity, egocar, dyn_threshold);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(sums && v_terms);
  int64_t blocks = cdiv((int64_t)H * W, kLossBlock * 2);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reg_loss_bwd_kernel, dim3((unsigned)blocks), dim3(kLossBlock), 0, as_stream(stream), a, sums, v_terms,
                     v_opacity, v_depth, v_rgb);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
