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
