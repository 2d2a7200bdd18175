// Adam step on one parameter tensor: the optimiser of the reference trainer
// (torch.optim.Adam(groups, lr=0.0, eps=1e-15), /root/reference/project/models/trainers/base.py:222; per-group lr / eps /
// weight_decay :201-207; betas (0.9, 0.999), amsgrad off) as ONE streaming pass -- SURVEY.md 8f rank 2, first slice.
// The arithmetic follows torch/optim/adam.py `_single_tensor_adam` (non-capturable branch) operation by operation:
//   g = grad + weight_decay * p;  m = m + (g - m) * (1 - b1)  [lerp];  v = v * b2 + (1 - b2) * g * g  [mul, addcmul]
//   denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p = p - (lr / (1 - b1^t)) * m / denom  [addcdiv]
// HBM-bound: 28 B per element (read p, g, m, v; write p, m, v).  59 floats per Gaussian -> 1.65 KB per Gaussian per step.
#include "bds_common.h"

namespace bds {

constexpr int kOptBlock = 256;

// kClear: the gradient is cleared as it is consumed ("consume and clear": the next step's backward accumulates into zeros without a
// clearing pass of its own -- 4 more bytes written per element here against a row-wise clear launch per view there)
template <bool kVec, bool kClear>
__global__ __launch_bounds__(kOptBlock) void adam_step_kernel(int64_t n, float *__restrict__ p, float *__restrict__ g,
                                                             float *__restrict__ m, float *__restrict__ v, float step_size,
                                                             float one_minus_b1, float b2, float one_minus_b2,
                                                             float bc2_sqrt, float eps, float weight_decay) {
#pragma clang fp contract(off)  // torch evaluates these as separate element-wise operations
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {
    if (weight_decay != 0.f) gg = gg + weight_decay * pp;
    // torch's lerp_(g, w): w < 0.5 ? m + w * (g - m) : g - (g - m) * (1 - w)   (w = 1 - beta1; the second branch is taken for beta1 <= 0.5)
    mm = one_minus_b1 < 0.5f ? mm + one_minus_b1 * (gg - mm) : gg - (gg - mm) * (1.f - one_minus_b1);
    vv = vv * b2 + (one_minus_b2 * gg) * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp = pp + (-step_size) * (mm / denom);
  };
  const int64_t stride = (int64_t)gridDim.x * kOptBlock;
  if (kVec) {
    const int64_t n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    float4 *g4 = reinterpret_cast<float4 *>(g);
    for (int64_t i = (int64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += stride) {
      float4 pp = p4[i], mm = m4[i], vv = v4[i];
      const float4 gg = g4[i];
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      p4[i] = pp; m4[i] = mm; v4[i] = vv;
      if (kClear) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
      const int64_t i = n4 * 4 + threadIdx.x;
      upd(p[i], g[i], m[i], v[i]);
      if (kClear) g[i] = 0.f;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n; i += stride) {
      upd(p[i], g[i], m[i], v[i]);
      if (kClear) g[i] = 0.f;
    }
  }
}

// The same update for a parameter [N, width] whose GRADIENT is a column range of a wider row block (element (r, c) at
// grad[r * grad_stride + c]: dist.FlatGradients(row_block=True) keeps the four small per-Gaussian gradients as one [N,16] block)
template <bool kClear>
__global__ __launch_bounds__(kOptBlock) void adam_step_rows_kernel(int64_t n, int width, int64_t grad_stride, float *__restrict__ p,
                                                                  float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                                  float step_size, float one_minus_b1, float b2, float one_minus_b2,
                                                                  float bc2_sqrt, float eps, float weight_decay) {
#pragma clang fp contract(off)
  for (int64_t i = (int64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kOptBlock) {
    const int64_t r = i / width;
    float *gp = g + r * grad_stride + (i - r * width);
    float gg = *gp, pp = p[i], mm = m[i], vv = v[i];
    if (weight_decay != 0.f) gg = gg + weight_decay * pp;
    mm = one_minus_b1 < 0.5f ? mm + one_minus_b1 * (gg - mm) : gg - (gg - mm) * (1.f - one_minus_b1);
    vv = vv * b2 + (one_minus_b2 * gg) * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pp + (-step_size) * (mm / denom);
    m[i] = mm; v[i] = vv;
    if (kClear) *gp = 0.f;
  }
}

}  // namespace bds

using namespace bds;

extern "C" int bds_adam_step_rows(int64_t n_rows, int width, int64_t grad_stride, float *param, float *grad, float *exp_avg,
                                  float *exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                                  int consume, bds_stream_t stream) {
  BDS_REQUIRE(n_rows >= 0 && width >= 1 && grad_stride >= width && step >= 1);
  if (n_rows == 0) return BDS_OK;
  BDS_REQUIRE(param && grad && exp_avg && exp_avg_sq);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const int64_t n = n_rows * width;
  int64_t blocks = cdiv(n, kOptBlock * 2);
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = as_stream(stream);
#define BDS_ADAM_ROWS(C)                                                                                                              \
  hipLaunchKernelGGL((adam_step_rows_kernel<C>), dim3((unsigned)blocks), dim3(kOptBlock), 0, st, n, width, grad_stride, param, grad,   \
                     exp_avg, exp_avg_sq, step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), bc2_sqrt, (float)eps,   \
                     (float)weight_decay)
  if (consume) BDS_ADAM_ROWS(true); else BDS_ADAM_ROWS(false);
#undef BDS_ADAM_ROWS
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int adam_step_impl(int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1, double beta2,
                          double eps, double weight_decay, int64_t step, bool clear, bds_stream_t stream) {
  BDS_REQUIRE(n >= 0 && step >= 1);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(param && grad && exp_avg && exp_avg_sq);
  // scalars in double like Python's float arithmetic in torch/optim/adam.py (1 - beta, bias corrections), each rounded
  // to fp32 once where torch hands it to an fp32 element-wise kernel
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const bool vec = aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq);
  int64_t blocks = cdiv(vec ? cdiv(n, 4) : n, kOptBlock * 2);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipStream_t st = as_stream(stream);
#define BDS_ADAM(V, C)                                                                                                                   \
  hipLaunchKernelGGL((adam_step_kernel<V, C>), dim3((unsigned)blocks), dim3(kOptBlock), 0, st, n, param, grad, exp_avg, exp_avg_sq, step_size, \
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), bc2_sqrt, (float)eps, (float)weight_decay)
  if (vec) { if (clear) BDS_ADAM(true, true); else BDS_ADAM(true, false); }
  else     { if (clear) BDS_ADAM(false, true); else BDS_ADAM(false, false); }
#undef BDS_ADAM
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int64_t step, bds_stream_t stream) {
  return adam_step_impl(n, param, const_cast<float *>(grad), exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, false, stream);
}
extern "C" int bds_adam_step_consume(int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1,
                                     double beta2, double eps, double weight_decay, int64_t step, bds_stream_t stream) {
  return adam_step_impl(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, true, stream);
}

// ---- per-step densification statistics --------------------------------------------------------------------------------
// BasicTrainer.postprocess_per_train_step (models/trainers/base.py:279-297) + VanillaGaussians.after_train
// (models/gaussians/vanilla.py:163-191) for one set of Gaussians, in one launch and without the boolean-mask indexing of the
// reference (each `x[mask] = ...` there is a nonzero() with a host sync):
//   g = |absgrad| scaled by (width/2, height/2) * batch_size;  n = ||g||_2
//   first call : xys_grad_norm = n for EVERY Gaussian, vis_counts = 1 for EVERY Gaussian (the reference's initialisation)
//   later calls: visible (radii > 0): xys_grad_norm += n, vis_counts += 1
//   always     : visible: max_2Dsize = max(max_2Dsize, radii / last_size)   (max_2Dsize starts at 0)
// PINNED by tests/golden/densify_stats.npz (the reference's own method, oracle/gen_golden_densify.py).
namespace bds {
__global__ __launch_bounds__(kOptBlock) void densify_stats_kernel(int64_t N, const float *__restrict__ grad2d,
                                                                 const int32_t *__restrict__ radii, float sx, float sy,
                                                                 float last_size, int first, float *__restrict__ xys_grad_norm,
                                                                 float *__restrict__ vis_counts, float *__restrict__ max_2Dsize) {
#pragma clang fp contract(off)  // torch: separate multiply, square, add, sqrt
  const int64_t i = (int64_t)blockIdx.x * kOptBlock + threadIdx.x;
  if (i >= N) return;
  const int r = radii[i];
  const bool vis = r > 0;
  float n = 0.f;
  if (first || vis) {
    const float gx = grad2d[i * 2] * sx, gy = grad2d[i * 2 + 1] * sy;
    n = sqrtf(gx * gx + gy * gy);
  }
  if (first) {
    xys_grad_norm[i] = n;
    vis_counts[i] = 1.f;
    max_2Dsize[i] = vis ? fmaxf(0.f, (float)r / last_size) : 0.f;
  } else if (vis) {
    xys_grad_norm[i] = n + xys_grad_norm[i];
    vis_counts[i] = vis_counts[i] + 1.f;
    max_2Dsize[i] = fmaxf(max_2Dsize[i], (float)r / last_size);
  }
}
}  // namespace bds

extern "C" int bds_densify_stats(int64_t N, const float *grad2d, const int32_t *radii, int width, int height, int batch_size,
                                 int last_size, int first, float *xys_grad_norm, float *vis_counts, float *max_2Dsize,
                                 bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && width > 0 && height > 0 && batch_size >= 1 && last_size > 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(grad2d && radii && xys_grad_norm && vis_counts && max_2Dsize);
  // the reference multiplies by the Python float (width / 2.0 * batch_size): formed in double, rounded once
  const float sx = (float)((double)width / 2.0 * (double)batch_size), sy = (float)((double)height / 2.0 * (double)batch_size);
  hipLaunchKernelGGL(bds::densify_stats_kernel, dim3((unsigned)cdiv(N, bds::kOptBlock)), dim3(bds::kOptBlock), 0, as_stream(stream),
                     N, grad2d, radii, sx, sy, (float)last_size, first, xys_grad_norm, vis_counts, max_2Dsize);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
