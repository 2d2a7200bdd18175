// Image-side glue of the training-step call sequence, one launch each (the per-Gaussian glue lives in the
// one-view kernels of project.hip / sh.hip).
// Reference arithmetic (file:line under /root/reference/project):
//   gsplat rasterization(render_mode="RGB+ED") depth = D / clamp(alpha, 1e-10)
//   models/trainers/base.py:414-419           split [3,1]
#include "bds_common.h"

namespace bds {

constexpr int kGlueBlock = 256;

// render[P,4], alpha[P] -> rgb[P,3], depth[P] = render.w / max(alpha, 1e-10)
__global__ __launch_bounds__(kGlueBlock) void render_unpack_fwd_kernel(int64_t P, const float4 *__restrict__ render,
                                                                      const float *__restrict__ alphas, float *__restrict__ rgb,
                                                                      float *__restrict__ depth) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= P) return;
  const float4 r = render[i];
  rgb[i * 3] = r.x; rgb[i * 3 + 1] = r.y; rgb[i * 3 + 2] = r.z;
  depth[i] = r.w / fmaxf(alphas[i], 1e-10f);
}

__global__ __launch_bounds__(kGlueBlock) void render_unpack_bwd_kernel(int64_t P, const float4 *__restrict__ render,
                                                                      const float *__restrict__ alphas, const float *__restrict__ v_rgb,
                                                                      const float *__restrict__ v_depth,
                                                                      const float *__restrict__ v_alpha_a,
                                                                      const float *__restrict__ v_alpha_b, float4 *__restrict__ v_render,
                                                                      float *__restrict__ v_alphas) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= P) return;
  const float a = alphas[i];
  const float ac = fmaxf(a, 1e-10f);
  const float vd = v_depth ? v_depth[i] : 0.f;
  v_render[i] = make_float4(v_rgb[i * 3], v_rgb[i * 3 + 1], v_rgb[i * 3 + 2], vd / ac);
  float va = (v_alpha_a ? v_alpha_a[i] : 0.f) + (v_alpha_b ? v_alpha_b[i] : 0.f);
  if (a >= 1e-10f) va -= render[i].w * vd / (ac * ac);  // clamp(min) passes the gradient where alpha >= 1e-10
  v_alphas[i] = va;
}

}  // namespace bds

using namespace bds;

#define BDS_GLUE_LAUNCH(kernel, n, ...)                                                                              \
  do {                                                                                                               \
    if ((n) > 0) {                                                                                                   \
      hipLaunchKernelGGL(kernel, dim3((unsigned)cdiv((n), kGlueBlock)), dim3(kGlueBlock), 0, as_stream(stream), (n), \
                         __VA_ARGS__);                                                                               \
      BDS_LAUNCH_CHECK();                                                                                            \
    }                                                                                                                \
  } while (0)

extern "C" int bds_render_unpack_fwd(int64_t P, const float *render, const float *alphas, float *rgb, float *depth,
                                     bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && (P == 0 || (render && alphas && rgb && depth && aligned16(render))));
  BDS_GLUE_LAUNCH(render_unpack_fwd_kernel, P, reinterpret_cast<const float4 *>(render), alphas, rgb, depth);
  return BDS_OK;
}

extern "C" int bds_render_unpack_bwd(int64_t P, const float *render, const float *alphas, const float *v_rgb,
                                     const float *v_depth, const float *v_alpha_a, const float *v_alpha_b, float *v_render,
                                     float *v_alphas, bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && (P == 0 || (render && alphas && v_rgb && v_render && v_alphas && aligned16(render) && aligned16(v_render))));
  BDS_GLUE_LAUNCH(render_unpack_bwd_kernel, P, reinterpret_cast<const float4 *>(render), alphas, v_rgb, v_depth, v_alpha_a,
                  v_alpha_b, reinterpret_cast<float4 *>(v_render), v_alphas);
  return BDS_OK;
}
