// Element-wise glue of the training-step call sequence, fused into single launches so that the harness's
// fast path does not spend its time in ~25 tiny framework kernels between the big ones.
// Reference arithmetic (file:line under /root/reference/project):
//   models/gaussians/vanilla.py:389,393-394   clamp(sh + 0.5, 0, 1); sigmoid(opacity); exp(scale)
//   gsplat rasterization(render_mode="RGB+ED") colours = cat(rgb, depth); depth = D / clamp(alpha, 1e-10)
//   models/trainers/base.py:414-419           split [3,1]
#include "bds_common.h"

namespace bds {

constexpr int kGlueBlock = 256;

__global__ __launch_bounds__(kGlueBlock) void activate_fwd_kernel(int64_t N, const float *__restrict__ log_scales,
                                                                 const float *__restrict__ logits, float *__restrict__ scales,
                                                                 float *__restrict__ opac) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int k = 0; k < 3; k++) scales[i * 3 + k] = expf(log_scales[i * 3 + k]);
  opac[i] = 1.f / (1.f + expf(-logits[i]));
}

__global__ __launch_bounds__(kGlueBlock) void activate_bwd_kernel(int64_t N, const float *__restrict__ scales,
                                                                 const float *__restrict__ opac, const float *__restrict__ v_scales,
                                                                 const float *__restrict__ v_opac, float *__restrict__ v_log_scales,
                                                                 float *__restrict__ v_logits) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int k = 0; k < 3; k++) v_log_scales[i * 3 + k] = v_scales[i * 3 + k] * scales[i * 3 + k];
  const float o = opac[i];
  v_logits[i] = v_opac[i] * o * (1.f - o);
}

// colors[N,4] = (clamp(sh_rgb + 0.5, 0, 1), depth)
__global__ __launch_bounds__(kGlueBlock) void colors_pack_fwd_kernel(int64_t N, const float *__restrict__ sh_rgb,
                                                                    const float *__restrict__ depths, float4 *__restrict__ colors) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= N) return;
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) c[k] = fminf(fmaxf(sh_rgb[i * 3 + k] + 0.5f, 0.f), 1.f);
  colors[i] = make_float4(c[0], c[1], c[2], depths[i]);
}

__global__ __launch_bounds__(kGlueBlock) void colors_pack_bwd_kernel(int64_t N, const float *__restrict__ sh_rgb,
                                                                    const float4 *__restrict__ v_colors, float *__restrict__ v_sh_rgb,
                                                                    float *__restrict__ v_depths) {
  const int64_t i = (int64_t)blockIdx.x * kGlueBlock + threadIdx.x;
  if (i >= N) return;
  const float4 v = v_colors[i];
  const float g[3] = {v.x, v.y, v.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float x = sh_rgb[i * 3 + k] + 0.5f;
    v_sh_rgb[i * 3 + k] = (x >= 0.f && x <= 1.f) ? g[k] : 0.f;  // torch.clamp passes the gradient on the closed interval
  }
  v_depths[i] = v.w;
}

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

extern "C" int bds_activate_fwd(int64_t N, const float *log_scales, const float *logits, float *scales, float *opacities,
                                bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && (N == 0 || (log_scales && logits && scales && opacities)));
  BDS_GLUE_LAUNCH(activate_fwd_kernel, N, log_scales, logits, scales, opacities);
  return BDS_OK;
}

extern "C" int bds_activate_bwd(int64_t N, const float *scales, const float *opacities, const float *v_scales,
                                const float *v_opacities, float *v_log_scales, float *v_logits, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && (N == 0 || (scales && opacities && v_scales && v_opacities && v_log_scales && v_logits)));
  BDS_GLUE_LAUNCH(activate_bwd_kernel, N, scales, opacities, v_scales, v_opacities, v_log_scales, v_logits);
  return BDS_OK;
}

extern "C" int bds_colors_pack_fwd(int64_t N, const float *sh_rgb, const float *depths, float *colors, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && (N == 0 || (sh_rgb && depths && colors && aligned16(colors))));
  BDS_GLUE_LAUNCH(colors_pack_fwd_kernel, N, sh_rgb, depths, reinterpret_cast<float4 *>(colors));
  return BDS_OK;
}

extern "C" int bds_colors_pack_bwd(int64_t N, const float *sh_rgb, const float *v_colors, float *v_sh_rgb, float *v_depths,
                                   bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && (N == 0 || (sh_rgb && v_colors && v_sh_rgb && v_depths && aligned16(v_colors))));
  BDS_GLUE_LAUNCH(colors_pack_bwd_kernel, N, sh_rgb, reinterpret_cast<const float4 *>(v_colors), v_sh_rgb, v_depths);
  return BDS_OK;
}

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
