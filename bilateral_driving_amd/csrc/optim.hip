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
      // streamed once: non-temporal loads and stores (no reuse to keep in L2; measured 0.664 -> 0.622 ms on the 2 M-Gaussian parameter set)
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f *pn = reinterpret_cast<v4f *>(p4 + i), *mn = reinterpret_cast<v4f *>(m4 + i), *vn = reinterpret_cast<v4f *>(v4 + i), *gn = reinterpret_cast<v4f *>(g4 + i);
      v4f P = __builtin_nontemporal_load(pn), M = __builtin_nontemporal_load(mn), V = __builtin_nontemporal_load(vn);
      const v4f Gv = __builtin_nontemporal_load(gn);
      float4 pp = make_float4(P.x, P.y, P.z, P.w), mm = make_float4(M.x, M.y, M.z, M.w), vv = make_float4(V.x, V.y, V.z, V.w);
      const float4 gg = make_float4(Gv.x, Gv.y, Gv.z, Gv.w);
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      P = (v4f){pp.x, pp.y, pp.z, pp.w}; M = (v4f){mm.x, mm.y, mm.z, mm.w}; V = (v4f){vv.x, vv.y, vv.z, vv.w};
      __builtin_nontemporal_store(P, pn); __builtin_nontemporal_store(M, mn); __builtin_nontemporal_store(V, vn);
      if (kClear) __builtin_nontemporal_store((v4f){0.f, 0.f, 0.f, 0.f}, gn);
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

// ---- deferred ("row-lazy") Adam: the SAME numbers as the dense pass, bytes proportional to the rows a step touches ---------------
// The reference steps ONE dense torch.optim.Adam after every single-view iteration (tools/train.py:252-283 -> trainers/base.py:502-516,
// optimizer at :222-226; gsplat's gradients are dense tensors whose rows are exact zeros for the ~85 % of the Gaussians the view does
// not see).  A row with zero gradient still moves (m <- b1 m, v <- b2 v, p <- p - lr_t m^/(sqrt(v^)+eps)), so skipping it changes the
// training; but its recurrence needs nothing except the per-step scalars.  For a parameter that only the VISIBLE rows of a view read
// (the SH coefficients: evaluated by the record pack over the view's visible-id list) the missed steps can therefore be replayed, in
// the same fp32 order, right before the row is next read:
//   * last_step[row]  = the optimizer step this row's (p, m, v) are current for;
//   * table[s % T]    = (lr_s / (1 - b1^s) for the columns < split, the same for the columns >= split, sqrt(1 - b2^s), -) written by
//                       the step launch of step s (block 0), read by every later replay of step s;
//   * advance(list, t): rows of the list with last < t replay the zero-gradient steps last+1 .. t-1 (or .. t without a gradient) from
//     the table and, with a gradient, take step t from it; last = t.  Called (a) inside the view's forward, after the visible list is
//     known and before the record pack reads the coefficients (target = the device-side clock, so that a captured hipGraph stays
//     valid), (b) by the optimizer's step over the frame's lists, (c) densely (ids = NULL) before anything else reads the tensor
//     (densification, checkpoints, evaluation) and at least every T - 1 steps.
// The arithmetic is `adam_upd` below for both forms: bit-equal to bds_adam_step by construction (tests/test_gpu_10_optim.py).
// A block owns floor(256 / row_floats) whole rows: the row's step word is read by all of its threads, a barrier, then written by one.
namespace bds {

__device__ __forceinline__ void adam_upd(float &pp, float gg, float &mm, float &vv, float step_size, float bc2_sqrt, float one_minus_b1,
                                         float b2, float one_minus_b2, float eps, float weight_decay) {
#pragma clang fp contract(off)
  if (weight_decay != 0.f) gg = gg + weight_decay * pp;
  mm = one_minus_b1 < 0.5f ? mm + one_minus_b1 * (gg - mm) : gg - (gg - mm) * (1.f - one_minus_b1);
  vv = vv * b2 + (one_minus_b2 * gg) * gg;
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  pp = pp + (-step_size) * (mm / denom);
}

// ---- several tensors, one launch ------------------------------------------------------------------------------------------------
// The reference's optimizer holds ~10 small groups next to the two SH tensors (xyz, rotation, scaling, opacity, the bilateral grids of
// every level, sky, poses: models/trainers/base.py:201-226); one launch per tensor is ~8 us of launch gap each for microseconds of
// work.  bds_adam_step_multi takes up to kAdamMulti tensors with their own hyper-parameters, steps and gradient layouts (contiguous,
// or a column range of a row block as bds_adam_step_rows) and gives every tensor a contiguous range of the workgroups.
constexpr int kAdamMulti = 12;
struct AdamMultiArgs {
  float *p[kAdamMulti], *g[kAdamMulti], *m[kAdamMulti], *v[kAdamMulti];
  int64_t n[kAdamMulti], grad_stride[kAdamMulti];
  int width[kAdamMulti], block0[kAdamMulti + 1];
  float step_size[kAdamMulti], bc2_sqrt[kAdamMulti], eps[kAdamMulti], wd[kAdamMulti], one_minus_b1[kAdamMulti], b2[kAdamMulti],
      one_minus_b2[kAdamMulti];
  int count, consume;
};
__global__ __launch_bounds__(kOptBlock) void adam_step_multi_kernel(AdamMultiArgs A) {
  int t = 0;
#pragma unroll 1
  while (t + 1 < A.count && (int)blockIdx.x >= A.block0[t + 1]) t++;
  const int nb = A.block0[t + 1] - A.block0[t], lb = (int)blockIdx.x - A.block0[t];
  const int64_t n = A.n[t], gs = A.grad_stride[t];
  const int width = A.width[t];
  float *__restrict__ p = A.p[t], *__restrict__ g = A.g[t], *__restrict__ m = A.m[t], *__restrict__ v = A.v[t];
  const float ss = A.step_size[t], bc2s = A.bc2_sqrt[t], eps = A.eps[t], wd = A.wd[t], omb1 = A.one_minus_b1[t], b2 = A.b2[t],
              omb2 = A.one_minus_b2[t];
  for (int64_t i = (int64_t)lb * kOptBlock + threadIdx.x; i < n; i += (int64_t)nb * kOptBlock) {
    float *gp = g + i;
    if (width > 0) { const int64_t r = i / width; gp = g + r * gs + (i - r * width); }
    float pp = p[i], mm = m[i], vv = v[i];
    adam_upd(pp, *gp, mm, vv, ss, bc2s, omb1, b2, omb2, eps, wd);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (A.consume) *gp = 0.f;
  }
}

// ---- the four small per-Gaussian tensors through their gradient ROW BLOCK, once ------------------------------------------------------
// dist.FlatGradients(row_block=True) keeps the gradients of means / opacity logit / quats / log-scales as the columns of one [N,16] block
// of 64-byte rows (bds_common.h GradLayout).  Stepping the four tensors one after the other reads every 64-byte row FOUR times for 3 / 1
// / 4 / 3 of its floats (512 MB of lines for 88 MB of gradients at 2 M Gaussians: the step was 230 us, 2.7 TB/s nominal); here sixteen
// lanes take one row, each lane the element of its column's tensor: the block is read (and, consuming, cleared) once.
constexpr int kRowParts = 4;
struct AdamRowBlockArgs {
  float *p[kRowParts], *m[kRowParts], *v[kRowParts];
  int col0[kRowParts], width[kRowParts];
  float step_size[kRowParts], bc2_sqrt[kRowParts], eps[kRowParts], wd[kRowParts], one_minus_b1[kRowParts], b2[kRowParts], one_minus_b2[kRowParts];
  int count, consume;
};
// one thread per ROW: the 64-byte gradient row as four 16-byte loads (a wave reads 4 KB contiguous), then the row's elements of each
// tensor (consecutive threads, consecutive rows: every tensor's accesses of a wave cover one contiguous span)
__global__ __launch_bounds__(kOptBlock) void adam_step_rowblock_kernel(int64_t N, float *__restrict__ block, AdamRowBlockArgs A) {
  for (int64_t row = (int64_t)blockIdx.x * kOptBlock + threadIdx.x; row < N; row += (int64_t)gridDim.x * kOptBlock) {
    float4 *g4 = reinterpret_cast<float4 *>(block + row * 16);
    const float4 q0 = g4[0], q1 = g4[1], q2 = g4[2], q3 = g4[3];
    const float g[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
    for (int t = 0; t < kRowParts; t++) {
      if (t >= A.count) break;
      const int w = A.width[t], c0 = A.col0[t];
      float *__restrict__ p = A.p[t] + row * w, *__restrict__ m = A.m[t] + row * w, *__restrict__ v = A.v[t] + row * w;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (k >= w) break;
        float gg = 0.f;
#pragma unroll
        for (int c = 0; c < 16; c++) gg = (c == c0 + k) ? g[c] : gg;      // (register select: g[] stays in registers)
        float pp = p[k], mm = m[k], vv = v[k];
        adam_upd(pp, gg, mm, vv, A.step_size[t], A.bc2_sqrt[t], A.one_minus_b1[t], A.b2[t], A.one_minus_b2[t], A.eps[t], A.wd[t]);
        p[k] = pp; m[k] = mm; v[k] = vv;
      }
    }
    if (A.consume) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      g4[0] = z; g4[1] = z; g4[2] = z; g4[3] = z;
    }
  }
}

// kVec = 4: row_floats % 4 == 0 and 16-byte aligned arrays: a thread owns four consecutive columns of a row (one float4 of p / m / v /
// g each way; the replay loop reads its table entry once for the four)
template <int kVec>
__global__ __launch_bounds__(kOptBlock) void adam_rows_advance_kernel(
    int64_t n_cap, const uint64_t *__restrict__ n_dev, const int32_t *__restrict__ ids, int64_t N, int row_floats, int split_col,
    float *__restrict__ p, float *__restrict__ g, int consume, float *__restrict__ m, float *__restrict__ v, int32_t *__restrict__ last,
    int32_t *__restrict__ clock, int32_t t_host, int with_step, float4 *__restrict__ table, int T, float ss_a, float ss_b, float bc2s,
    float one_minus_b1, float b2, float one_minus_b2, float eps, float weight_decay) {
  const int tpr = row_floats / kVec;                   // threads per row
  const int rpb = kOptBlock / tpr;                     // whole rows per block
  const int lr = threadIdx.x / tpr, c = (threadIdx.x - lr * tpr) * kVec;
  const int32_t t = with_step ? t_host : (clock != nullptr ? *clock : t_host);
  int64_t n = n_cap;
  if (n_dev != nullptr) { const int64_t nd = (int64_t)*n_dev; n = nd < n_cap ? nd : n_cap; }
  if (ids == nullptr && n > N) n = N;
  const int64_t groups = (n + rpb - 1) / rpb;
  for (int64_t gi = blockIdx.x; gi < groups; gi += gridDim.x) {
    const int64_t r = gi * rpb + lr;
    int64_t row = -1;
    if (lr < rpb && r < n) row = ids != nullptr ? (int64_t)ids[r] : r;
    if (row >= N) row = -1;
    const int32_t k0 = row >= 0 ? last[row] : t;
    __syncthreads();                                     // every thread of the row has read its step word
    if (row >= 0 && k0 < t) {
      const int64_t e = row * row_floats + c;
      float pp[kVec], mm[kVec], vv[kVec];
      if (kVec == 4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(p + e), b4 = *reinterpret_cast<const float4 *>(m + e),
                     c4 = *reinterpret_cast<const float4 *>(v + e);
        pp[0] = a4.x; pp[1 % kVec] = a4.y; pp[2 % kVec] = a4.z; pp[3 % kVec] = a4.w;
        mm[0] = b4.x; mm[1 % kVec] = b4.y; mm[2 % kVec] = b4.z; mm[3 % kVec] = b4.w;
        vv[0] = c4.x; vv[1 % kVec] = c4.y; vv[2 % kVec] = c4.z; vv[3 % kVec] = c4.w;
      } else {
        pp[0] = p[e]; mm[0] = m[e]; vv[0] = v[e];
      }
      const int32_t stop = with_step ? t - 1 : t;
      for (int32_t s = k0 + 1; s <= stop; ++s) {
        const float4 h = table[s % T];
#pragma unroll
        for (int j = 0; j < kVec; ++j)
          adam_upd(pp[j], 0.f, mm[j], vv[j], c + j < split_col ? h.x : h.y, h.z, one_minus_b1, b2, one_minus_b2, eps, weight_decay);
      }
      if (with_step) {
        float gg[kVec];
        if (kVec == 4) {
          const float4 g4 = *reinterpret_cast<const float4 *>(g + e);
          gg[0] = g4.x; gg[1 % kVec] = g4.y; gg[2 % kVec] = g4.z; gg[3 % kVec] = g4.w;
          if (consume) *reinterpret_cast<float4 *>(g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          gg[0] = g[e];
          if (consume) g[e] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < kVec; ++j)
          adam_upd(pp[j], gg[j], mm[j], vv[j], c + j < split_col ? ss_a : ss_b, bc2s, one_minus_b1, b2, one_minus_b2, eps, weight_decay);
      }
      if (kVec == 4) {
        *reinterpret_cast<float4 *>(p + e) = make_float4(pp[0], pp[1 % kVec], pp[2 % kVec], pp[3 % kVec]);
        *reinterpret_cast<float4 *>(m + e) = make_float4(mm[0], mm[1 % kVec], mm[2 % kVec], mm[3 % kVec]);
        *reinterpret_cast<float4 *>(v + e) = make_float4(vv[0], vv[1 % kVec], vv[2 % kVec], vv[3 % kVec]);
      } else {
        p[e] = pp[0]; m[e] = mm[0]; v[e] = vv[0];
      }
      if (c == 0) last[row] = t;
    }
    __syncthreads();                                     // (the next group's rows are other rows, but keep the phases apart)
  }
  if (with_step && blockIdx.x == 0 && threadIdx.x == 0) {
    table[t % T] = make_float4(ss_a, ss_b, bc2s, 0.f);   // (no replay of this launch reads slot t % T: every gap is < T)
    if (clock != nullptr) *clock = t;
  }
}

}  // namespace bds

extern "C" int bds_adam_rows_advance(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int64_t N, int row_floats, int split_col,
                                     float *param, float *grad, int consume, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                                     int32_t *clock_dev, int64_t step, int with_step, float *table, int table_steps, double lr_a,
                                     double lr_b, double beta1, double beta2, double eps, double weight_decay, bds_stream_t stream) {
  BDS_REQUIRE(n_capacity >= 0 && N >= 0 && row_floats >= 1 && row_floats <= kOptBlock && split_col >= 0 && table_steps >= 2);
  BDS_REQUIRE(step >= 0 && step < (int64_t)1 << 31 && (with_step == 0 || (step >= 1 && grad != nullptr)));
  BDS_REQUIRE(param && exp_avg && exp_avg_sq && last_step && table && aligned16(table));
  BDS_REQUIRE(ids != nullptr || n_capacity <= N || n_dev != nullptr);
  float ss_a = 0.f, ss_b = 0.f, bc2s = 1.f;
  if (with_step) {      // scalars in double, rounded once: as bds_adam_step
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    ss_a = (float)(lr_a / bc1); ss_b = (float)(lr_b / bc1); bc2s = (float)sqrt(bc2);
  }
  const bool vec = row_floats % 4 == 0 && aligned16(param) && aligned16(exp_avg) && aligned16(exp_avg_sq) && (grad == nullptr || aligned16(grad));
  const int rpb = kOptBlock / (vec ? row_floats / 4 : row_floats);
  int64_t blocks = cdiv(n_capacity, (int64_t)rpb);
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
#define BDS_ADAM_ADVANCE(V)                                                                                                              \
  hipLaunchKernelGGL((bds::adam_rows_advance_kernel<V>), dim3((unsigned)blocks), dim3(kOptBlock), 0, as_stream(stream), n_capacity, n_dev, \
                     ids, N, row_floats, split_col, param, grad, consume, exp_avg, exp_avg_sq, last_step, clock_dev, (int32_t)step,          \
                     with_step, reinterpret_cast<float4 *>(table), table_steps, ss_a, ss_b, bc2s, (float)(1.0 - beta1), (float)beta2,       \
                     (float)(1.0 - beta2), (float)eps, (float)weight_decay)
  if (vec) BDS_ADAM_ADVANCE(4); else BDS_ADAM_ADVANCE(1);
#undef BDS_ADAM_ADVANCE
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_adam_step_rowblock(int64_t N, float *grad_block, int n_parts, float *const *params, float *const *exp_avgs,
                                      float *const *exp_avg_sqs, const int *col0, const int *widths, const double *lrs,
                                      const double *beta1s, const double *beta2s, const double *eps, const double *weight_decays,
                                      const int64_t *steps, int consume, bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && n_parts >= 1 && n_parts <= bds::kRowParts);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(grad_block && aligned16(grad_block) && params && exp_avgs && exp_avg_sqs && col0 && widths && lrs && beta1s && beta2s && eps &&
              weight_decays && steps);
  bds::AdamRowBlockArgs A;
  A.count = n_parts; A.consume = consume;
  unsigned used = 0;
  for (int t = 0; t < n_parts; t++) {
    BDS_REQUIRE(params[t] && exp_avgs[t] && exp_avg_sqs[t] && steps[t] >= 1 && widths[t] >= 1 && widths[t] <= 4 && col0[t] >= 0 &&
                col0[t] + widths[t] <= 16);
    const unsigned mask = ((1u << widths[t]) - 1u) << col0[t];
    BDS_REQUIRE((used & mask) == 0);      // column ranges do not overlap
    used |= mask;
    A.p[t] = params[t]; A.m[t] = exp_avgs[t]; A.v[t] = exp_avg_sqs[t]; A.col0[t] = col0[t]; A.width[t] = widths[t];
    const double bc1 = 1.0 - pow(beta1s[t], (double)steps[t]), bc2 = 1.0 - pow(beta2s[t], (double)steps[t]);   // (as bds_adam_step)
    A.step_size[t] = (float)(lrs[t] / bc1); A.bc2_sqrt[t] = (float)sqrt(bc2);
    A.eps[t] = (float)eps[t]; A.wd[t] = (float)weight_decays[t];
    A.one_minus_b1[t] = (float)(1.0 - beta1s[t]); A.b2[t] = (float)beta2s[t]; A.one_minus_b2[t] = (float)(1.0 - beta2s[t]);
  }
  int64_t blocks = cdiv(N, kOptBlock);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bds::adam_step_rowblock_kernel, dim3((unsigned)blocks), dim3(kOptBlock), 0, as_stream(stream), N, grad_block, A);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_adam_step_multi(int n_tensors, float *const *params, float *const *grads, float *const *exp_avgs,
                                   float *const *exp_avg_sqs, const int64_t *counts, const int *widths, const int64_t *grad_strides,
                                   const double *lrs, const double *beta1s, const double *beta2s, const double *eps,
                                   const double *weight_decays, const int64_t *steps, int consume, bds_stream_t stream) {
  BDS_REQUIRE(n_tensors >= 0 && n_tensors <= bds::kAdamMulti);
  if (n_tensors == 0) return BDS_OK;
  BDS_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && counts && widths && grad_strides && lrs && beta1s && beta2s && eps &&
              weight_decays && steps);
  bds::AdamMultiArgs A;
  A.count = 0; A.consume = consume;
  int blocks = 0;
  for (int t = 0; t < n_tensors; t++) {
    BDS_REQUIRE(counts[t] >= 0 && steps[t] >= 1 && widths[t] >= 0 && (widths[t] == 0 || grad_strides[t] >= widths[t]));
    if (counts[t] == 0) continue;
    BDS_REQUIRE(params[t] && grads[t] && exp_avgs[t] && exp_avg_sqs[t]);
    const int k = A.count++;
    A.p[k] = params[t]; A.g[k] = grads[t]; A.m[k] = exp_avgs[t]; A.v[k] = exp_avg_sqs[t];
    A.n[k] = counts[t]; A.width[k] = widths[t]; A.grad_stride[k] = grad_strides[t];
    const double bc1 = 1.0 - pow(beta1s[t], (double)steps[t]), bc2 = 1.0 - pow(beta2s[t], (double)steps[t]);   // (as bds_adam_step)
    A.step_size[k] = (float)(lrs[t] / bc1); A.bc2_sqrt[k] = (float)sqrt(bc2);
    A.eps[k] = (float)eps[t]; A.wd[k] = (float)weight_decays[t];
    A.one_minus_b1[k] = (float)(1.0 - beta1s[t]); A.b2[k] = (float)beta2s[t]; A.one_minus_b2[k] = (float)(1.0 - beta2s[t]);
    int64_t nb = cdiv(counts[t], kOptBlock * 2);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    A.block0[k] = blocks;
    blocks += (int)nb;
  }
  if (A.count == 0) return BDS_OK;
  A.block0[A.count] = blocks;
  hipLaunchKernelGGL(bds::adam_step_multi_kernel, dim3((unsigned)blocks), dim3(kOptBlock), 0, as_stream(stream), A);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
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
