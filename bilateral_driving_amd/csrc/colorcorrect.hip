// Colour-correct post-process of the evaluation path: bilateral/lib_bilagrid.py:56-120 `color_correct(img, ref, num_iters, eps)`,
// called per rendered frame at models/video_utils_color_correction.py:201 (eval-time path).
// Per iteration and per output channel c the reference solves  min_w || M_c (A w - ref_c) ||  with torch.linalg.lstsq, where a row of A is
// the quadratic expansion of a pixel  [rr, rg, rb, gg, gb, bb, r, g, b, 1]  (:98-104) and M_c masks the pixels that are unclipped in the
// input, in the current estimate and in the reference (:110); then img <- clip(A W, 0, 1) (:117).  That is 3 QR factorisations of a
// [pixels x 10] matrix per iteration.  Here one streaming pass per iteration applies the previous warp and accumulates the masked normal
// equations  G_c = sum a a^T (55 unique entries), h_c = sum a ref_c (10)  in double; the three 10x10 systems are solved by the caller
// (float64).  HBM-bound: 24 B/pixel read (+1 B mask) and 12 B/pixel written per iteration, three times the read for the channel split.
// PINNED by tests/golden/color_correct_*.npz (the reference's own function, oracle/gen_golden_color_correct.py).
#include "bds_common.h"

namespace bds {

constexpr int kCcBlock = 256;
constexpr int kCcTerms = 10, kCcAcc = 65;  // 55 (upper triangle of a a^T, row-major) + 10 (a * ref_c)

__device__ __forceinline__ void expand(float r, float g, float b, float *a) {
  a[0] = r * r; a[1] = r * g; a[2] = r * b; a[3] = g * g; a[4] = g * b; a[5] = b * b; a[6] = r; a[7] = g; a[8] = b; a[9] = 1.f;
}
__device__ __forceinline__ bool unclipped(float z, float eps) { return z >= eps && z <= 1.f - eps; }

__global__ __launch_bounds__(kCcBlock) void color_correct_step_kernel(int64_t P, const float *__restrict__ cur_in,
                                                                     const float *__restrict__ ref, const float *__restrict__ warp,
                                                                     float eps, uint8_t *__restrict__ mask0,
                                                                     float *__restrict__ cur_out, double *__restrict__ acc) {
#pragma clang fp contract(off)
  __shared__ double part[kCcBlock / kWave][kCcAcc];
  const int c = blockIdx.y;  // output channel whose normal equations this block accumulates
  float w[kCcTerms][3];
  if (warp)
    for (int k = 0; k < kCcTerms; k++)
      for (int j = 0; j < 3; j++) w[k][j] = warp[k * 3 + j];
  float s[kCcAcc];
#pragma unroll
  for (int k = 0; k < kCcAcc; k++) s[k] = 0.f;
  double d[kCcAcc];
#pragma unroll
  for (int k = 0; k < kCcAcc; k++) d[k] = 0.0;
  int pending = 0;
  for (int64_t i = (int64_t)blockIdx.x * kCcBlock + threadIdx.x; i < P; i += (int64_t)gridDim.x * kCcBlock) {
    float r = cur_in[i * 3], g = cur_in[i * 3 + 1], b = cur_in[i * 3 + 2];
    float a[kCcTerms];
    uint8_t m0;
    if (warp) {
      expand(r, g, b, a);
      float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < kCcTerms; k++) { o[0] += a[k] * w[k][0]; o[1] += a[k] * w[k][1]; o[2] += a[k] * w[k][2]; }
      r = fminf(fmaxf(o[0], 0.f), 1.f); g = fminf(fmaxf(o[1], 0.f), 1.f); b = fminf(fmaxf(o[2], 0.f), 1.f);
      m0 = mask0[i];
    } else {  // first pass: the input itself; remember which of its channels are unclipped
      m0 = (uint8_t)((unclipped(r, eps) ? 1 : 0) | (unclipped(g, eps) ? 2 : 0) | (unclipped(b, eps) ? 4 : 0));
      if (c == 0) mask0[i] = m0;
    }
    if (c == 0 && cur_out) { cur_out[i * 3] = r; cur_out[i * 3 + 1] = g; cur_out[i * 3 + 2] = b; }
    if (!acc) continue;
    const float x = c == 0 ? r : (c == 1 ? g : b);
    const float y = ref[i * 3 + c];
    if (((m0 >> c) & 1) && unclipped(x, eps) && unclipped(y, eps)) {
      expand(r, g, b, a);
      int t = 0;
#pragma unroll
      for (int p = 0; p < kCcTerms; p++)
#pragma unroll
        for (int q = p; q < kCcTerms; q++) s[t++] += a[p] * a[q];
#pragma unroll
      for (int p = 0; p < kCcTerms; p++) s[55 + p] += a[p] * y;
    }
    if (++pending == 32) {  // bound the length of a float running sum
#pragma unroll
      for (int k = 0; k < kCcAcc; k++) { d[k] += (double)s[k]; s[k] = 0.f; }
      pending = 0;
    }
  }
  if (!acc) return;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int k = 0; k < kCcAcc; k++) {
    double v = d[k] + (double)s[k];
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if (lane == 0) part[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kCcAcc) {
    double v = 0.0;
    for (int wv = 0; wv < kCcBlock / kWave; wv++) v += part[wv][threadIdx.x];
    if (v != 0.0) atomicAdd(acc + c * kCcAcc + threadIdx.x, v);
  }
}

}  // namespace bds

using namespace bds;

extern "C" int bds_color_correct_step(int64_t P, const float *cur_in, const float *ref, const float *warp, float eps, uint8_t *mask0,
                                      float *cur_out, double *acc, bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && eps >= 0.f);
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(cur_in && mask0 && (cur_out || acc) && (!acc || ref));
  int64_t blocks = cdiv(P, (int64_t)kCcBlock * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(color_correct_step_kernel, dim3((unsigned)blocks, acc ? 3 : 1), dim3(kCcBlock), 0, as_stream(stream), P, cur_in, ref,
                     warp, eps, mask0, cur_out, acc);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
