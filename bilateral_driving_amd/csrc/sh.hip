// K1: spherical-harmonics colour evaluation, forward and backward.
// Serves gsplat.cuda._wrapper.spherical_harmonics (/root/reference/project/models/gaussians/
// basics.py:15, vanilla.py:388).  HBM-bound streaming kernel: 216 B/Gaussian at degree 3.
//
// Layout: coeffs [n, K, 3] AoS (192 B/Gaussian at K=16).  A workgroup of 256 threads owns 256
// consecutive Gaussians; their coefficient rows are one contiguous span that is moved HBM<->LDS with
// fully coalesced 16-byte accesses, and each thread then reads / writes its own row in LDS
// (row stride padded by one dword so that the per-thread walk is bank-conflict free).
#include "bds_common.h"
#include "gs_math.h"

namespace bds {

constexpr int kShBlock = 256;

// kFull: nb == K (every coefficient is used) -> 16-byte coalesced path.  DEG is a template
// parameter so that the basis array is indexed statically and stays in registers.
template <int DEG, bool kFull>
__global__ __launch_bounds__(kShBlock) void sh_fwd_kernel(int64_t n, int K, const float *__restrict__ dirs,
                                                         const float *__restrict__ coeffs,
                                                         const uint8_t *__restrict__ masks, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int deg = DEG, nb = (DEG + 1) * (DEG + 1);
  const int row = nb * 3;       // floats used per Gaussian
  const int ldr = kFull ? row + 4 : row + 1;   // padded LDS row (kFull: 16-byte aligned rows; 52 floats put 16 rows on 16 bank quads)
  const int64_t g0 = (int64_t)blockIdx.x * kShBlock;
  const int cnt = (int)min((int64_t)kShBlock, n - g0);
  const int tid = threadIdx.x;
  if (kFull) {
    // span of cnt*row floats starting at coeffs + g0*row (16-byte aligned: row*4*256 % 16 == 0)
    const float4 *src = reinterpret_cast<const float4 *>(coeffs + g0 * row);
    const int n4 = cnt * row / 4;  // row = 48, 27.. ; kFull only used when row % 4 == 0
    for (int i = tid; i < n4; i += kShBlock) {
      int e = i * 4;
      int r = e / row, c = e - r * row;  // row % 4 == 0 -> the 4 floats stay in one row
      if (masks != nullptr && !masks[g0 + r]) continue;  // culled Gaussian: its 192-byte row is never fetched
      *reinterpret_cast<float4 *>(lds + r * ldr + c) = src[i];
    }
  } else {
    const int tot = cnt * row;
    for (int e = tid; e < tot; e += kShBlock) {
      int r = e / row, c = e - r * row;
      if (masks != nullptr && !masks[g0 + r]) continue;
      lds[r * ldr + c] = coeffs[(g0 + r) * (int64_t)K * 3 + c];
    }
  }
  __syncthreads();
  if (tid >= cnt) return;
  const int64_t g = g0 + tid;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  if (masks == nullptr || masks[g]) {
    float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
    float inorm = 1.0f / sqrtf(x * x + y * y + z * z);
    float B[16];
    sh_bases(deg, x * inorm, y * inorm, z * inorm, B);
    const float *c = lds + tid * ldr;
    if (kFull) {   // 16-byte LDS reads of the row (same products, same order)
      float cf[nb * 3 + 3];
#pragma unroll
      for (int i = 0; i < (nb * 3) / 4; i++) {
        const float4 v = reinterpret_cast<const float4 *>(c)[i];
        cf[i * 4] = v.x; cf[i * 4 + 1] = v.y; cf[i * 4 + 2] = v.z; cf[i * 4 + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < nb; k++) {
        o0 += B[k] * cf[k * 3];
        o1 += B[k] * cf[k * 3 + 1];
        o2 += B[k] * cf[k * 3 + 2];
      }
    } else {
#pragma unroll
      for (int k = 0; k < nb; k++) {
        o0 += B[k] * c[k * 3];
        o1 += B[k] * c[k * 3 + 1];
        o2 += B[k] * c[k * 3 + 2];
      }
    }
  }
  out[g * 3] = o0; out[g * 3 + 1] = o1; out[g * 3 + 2] = o2;
}

// Masked forward (the hot path evaluates SH after projection, only for the ~15 % of Gaussians that are
// on screen): no LDS staging -- a visible Gaussian reads its own row with 16-byte loads, a culled one
// touches nothing but its mask byte, so HBM traffic scales with the visible count.
template <int DEG, bool kVec>
__global__ __launch_bounds__(kShBlock) void sh_fwd_masked_kernel(int64_t n, int K, const float *__restrict__ dirs,
                                                                const float *__restrict__ coeffs,
                                                                const uint8_t *__restrict__ masks, float *__restrict__ out) {
  constexpr int nb = (DEG + 1) * (DEG + 1);
  const int64_t g = (int64_t)blockIdx.x * kShBlock + threadIdx.x;
  if (g >= n) return;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  if (masks[g]) {
    float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
    float inorm = 1.0f / sqrtf(x * x + y * y + z * z);
    float B[16];
    sh_bases(DEG, x * inorm, y * inorm, z * inorm, B);
    const float *c = coeffs + g * (int64_t)K * 3;
    float cf[nb * 3 + 3];
    if (kVec) {
      constexpr int n4 = (nb * 3 + 3) / 4;
#pragma unroll
      for (int i = 0; i < n4; i++) {
        if (i * 4 < K * 3) {  // stay inside the row
          const float4 v = reinterpret_cast<const float4 *>(c)[i];
          cf[i * 4] = v.x; cf[i * 4 + 1] = v.y; cf[i * 4 + 2] = v.z; cf[i * 4 + 3] = v.w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < nb * 3; i++) cf[i] = c[i];
    }
#pragma unroll
    for (int k = 0; k < nb; k++) {
      o0 += B[k] * cf[k * 3];
      o1 += B[k] * cf[k * 3 + 1];
      o2 += B[k] * cf[k * 3 + 2];
    }
  }
  out[g * 3] = o0; out[g * 3 + 1] = o1; out[g * 3 + 2] = o2;
}

template <int DEG, bool kFull>
__global__ __launch_bounds__(kShBlock) void sh_bwd_kernel(int64_t n, int K, const float *__restrict__ dirs,
                                                         const float *__restrict__ coeffs,
                                                         const uint8_t *__restrict__ masks,
                                                         const float *__restrict__ v_out, float *__restrict__ v_coeffs,
                                                         float *__restrict__ v_dirs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int deg = DEG, nb = (DEG + 1) * (DEG + 1);
  const int row = K * 3;   // the whole output row is written (bases >= nb get zero)
  const int ldr = row + 1;
  const int64_t g0 = (int64_t)blockIdx.x * kShBlock;
  const int cnt = (int)min((int64_t)kShBlock, n - g0);
  const int tid = threadIdx.x;
  if (tid < cnt) {
    const int64_t g = g0 + tid;
    float *c = lds + tid * ldr;
    bool on = (masks == nullptr || masks[g]);
    float vo0 = v_out[g * 3], vo1 = v_out[g * 3 + 1], vo2 = v_out[g * 3 + 2];
    float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
    float inorm = 1.0f / sqrtf(x * x + y * y + z * z);
    float ux = x * inorm, uy = y * inorm, uz = z * inorm;
    float B[16];
    sh_bases(deg, ux, uy, uz, B);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < K) {
        float b = (on && k < nb) ? B[k < nb ? k : 0] : 0.f;
        c[k * 3] = b * vo0; c[k * 3 + 1] = b * vo1; c[k * 3 + 2] = b * vo2;
      }
    }
    if (v_dirs != nullptr) {
      float vx = 0.f, vy = 0.f, vz = 0.f;
      if (on) {
        float gk[16];
        const float *cf = coeffs + g * (int64_t)K * 3;
#pragma unroll
        for (int k = 0; k < 16; k++)
          gk[k] = k < nb ? cf[k * 3] * vo0 + cf[k * 3 + 1] * vo1 + cf[k * 3 + 2] * vo2 : 0.f;
        float ax, ay, az;
        sh_bases_vjp(deg, ux, uy, uz, gk, ax, ay, az);
        // through the normalisation: v = (a - (a.u) u) / |d|
        float dot = ax * ux + ay * uy + az * uz;
        vx = (ax - dot * ux) * inorm; vy = (ay - dot * uy) * inorm; vz = (az - dot * uz) * inorm;
      }
      v_dirs[g * 3] = vx; v_dirs[g * 3 + 1] = vy; v_dirs[g * 3 + 2] = vz;
    }
  }
  __syncthreads();
  if (kFull) {
    float4 *dst = reinterpret_cast<float4 *>(v_coeffs + g0 * row);
    const int n4 = cnt * row / 4;
    for (int i = tid; i < n4; i += kShBlock) {
      int e = i * 4;
      int r = e / row, c = e - r * row;
      const float *s = lds + r * ldr + c;
      dst[i] = make_float4(s[0], s[1], s[2], s[3]);
    }
  } else {
    const int tot = cnt * row;
    for (int e = tid; e < tot; e += kShBlock) {
      int r = e / row, c = e - r * row;
      v_coeffs[g0 * row + e] = lds[r * ldr + c];
    }
  }
}

// ---- one-view variants for the fused training step -----------------------------------------------------
// Same arithmetic as above with the glue folded in: the view direction is means - cam_pos (vanilla.py:384,
// detached), visibility is radii > 0, and the result leaves already packed for the compositor as
// (clamp(rgb + 0.5, 0, 1), depth)  (vanilla.py:389 + the RGB+ED channel layout of base.py:393-408).
// sh_rgb keeps the un-clamped value: the backward needs to know where the clamp was active.
template <int DEG, bool kVec>
__global__ __launch_bounds__(kShBlock) void sh_view_fwd_kernel(int64_t n, int K, const float *__restrict__ means,
                                                              const float *__restrict__ cam_pos,
                                                              const float *__restrict__ coeffs,
                                                              const int32_t *__restrict__ radii,
                                                              const float *__restrict__ depths, float *__restrict__ sh_rgb,
                                                              float4 *__restrict__ colors) {
  constexpr int nb = (DEG + 1) * (DEG + 1);
  const int64_t g = (int64_t)blockIdx.x * kShBlock + threadIdx.x;
  if (g >= n) return;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  if (radii[g] > 0) {
    const float x = means[g * 3] - cam_pos[0], y = means[g * 3 + 1] - cam_pos[1], z = means[g * 3 + 2] - cam_pos[2];
    const float inorm = 1.0f / sqrtf(x * x + y * y + z * z);
    float B[16];
    sh_bases(DEG, x * inorm, y * inorm, z * inorm, B);
    const float *c = coeffs + g * (int64_t)K * 3;
    float cf[nb * 3 + 3];
    if (kVec) {
      constexpr int n4 = (nb * 3 + 3) / 4;
#pragma unroll
      for (int i = 0; i < n4; i++) {
        if (i * 4 < K * 3) {
          const float4 v = reinterpret_cast<const float4 *>(c)[i];
          cf[i * 4] = v.x; cf[i * 4 + 1] = v.y; cf[i * 4 + 2] = v.z; cf[i * 4 + 3] = v.w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < nb * 3; i++) cf[i] = c[i];
    }
#pragma unroll
    for (int k = 0; k < nb; k++) {
      o0 += B[k] * cf[k * 3];
      o1 += B[k] * cf[k * 3 + 1];
      o2 += B[k] * cf[k * 3 + 2];
    }
  }
  sh_rgb[g * 3] = o0; sh_rgb[g * 3 + 1] = o1; sh_rgb[g * 3 + 2] = o2;
  colors[g] = make_float4(fminf(fmaxf(o0 + 0.5f, 0.f), 1.f), fminf(fmaxf(o1 + 0.5f, 0.f), 1.f),
                          fminf(fmaxf(o2 + 0.5f, 0.f), 1.f), depths[g]);
}

// ---- backward over the VISIBLE Gaussians, list-driven -----------------------------------------------------------------------
// A view sees ~15 % of the Gaussians; the dense form streams 192 B of zeros for every culled one.  Here the work list is the
// depth-ordered id list of the visible entries (bds_isect_build: visible_ids) and the input is the compositor's gradient record of
// the same rank (colour gradient = floats 0-2 of the 64-byte record, read fully coalesced).  A workgroup stages its 256 rows in LDS
// and writes each as whole 16-byte pieces, 12 lanes per 192-byte row.  kAcc = false stores the rows (every other row of v_coeffs is
// the caller's business: zero-filled, or kept zero by bds_view_grads_clear_list), kAcc = true adds to them (several views summed
// into one buffer before one exchange).
template <int DEG, bool kVec, bool kAcc>
__global__ __launch_bounds__(kShBlock) void sh_view_bwd_list_kernel(int64_t n_cap, const uint64_t *__restrict__ n_dev,
                                                                   const int32_t *__restrict__ ids, int K,
                                                                   const float *__restrict__ means,
                                                                   const float *__restrict__ cam_pos,
                                                                   const float *__restrict__ sh_rgb,
                                                                   const float4 *__restrict__ v_rec, float *__restrict__ v_coeffs,
                                                                   const int32_t *__restrict__ row_map, int sh_rgb_by_rank,
                                                                   float *__restrict__ v_rest) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int32_t s_g[kShBlock];   // destination row of each staged entry
  constexpr int nb = (DEG + 1) * (DEG + 1);
  const int row = K * 3;
  const int ldr = row + 1;
  const int64_t n_list = list_length(n_cap, n_dev);
  const int64_t r0 = (int64_t)blockIdx.x * kShBlock;
  if (r0 >= n_list) return;
  const int cnt = (int)min((int64_t)kShBlock, n_list - r0);
  const int tid = threadIdx.x;
  if (tid < cnt) {
    const int64_t g = ids[r0 + tid];
    s_g[tid] = row_map ? row_map[g] : (int32_t)g;
    const float4 v = v_rec[(r0 + tid) * (BDS_GRAD_RECORD_FLOATS / 4)];
    float vo[3] = {v.x, v.y, v.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float x = sh_rgb[(sh_rgb_by_rank ? r0 + tid : g) * 3 + k] + 0.5f;   // [N,3] by Gaussian, or [n_list,3] in list order
      if (!(x >= 0.f && x <= 1.f)) vo[k] = 0.f;   // torch.clamp passes the gradient on the closed interval
    }
    const float x = means[g * 3] - cam_pos[0], y = means[g * 3 + 1] - cam_pos[1], z = means[g * 3 + 2] - cam_pos[2];
    const float inorm = 1.0f / sqrtf(x * x + y * y + z * z);
    float B[16];
    sh_bases(DEG, x * inorm, y * inorm, z * inorm, B);
    float *c = lds + tid * ldr;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < K) {
        const float b = k < nb ? B[k < nb ? k : 0] : 0.f;
        c[k * 3] = b * vo[0]; c[k * 3 + 1] = b * vo[1]; c[k * 3 + 2] = b * vo[2];
      }
    }
  }
  __syncthreads();
  if (kVec) {
    const int q4 = row / 4;   // 16-byte pieces per row
    const int total = cnt * q4;
    // four pieces per step: in accumulate mode their old values are loaded together BEFORE the first store (a load of v_coeffs may
    // not be moved above a store to it, so piece-by-piece every piece would wait out a full memory latency behind the previous one)
    for (int e0 = tid; e0 < total; e0 += 4 * kShBlock) {
      float4 *dst[4];
      float4 o[4], p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kShBlock;
        dst[u] = nullptr;
        if (e < total) {
          const int r = e / q4, cc = (e - r * q4) * 4;
          const float *sp = lds + r * ldr + cc;
          dst[u] = reinterpret_cast<float4 *>(v_coeffs + (int64_t)s_g[r] * row + cc);
          o[u] = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
      }
      if (kAcc) {
#pragma unroll
        for (int u = 0; u < 4; u++) if (dst[u]) p[u] = *dst[u];
#pragma unroll
        for (int u = 0; u < 4; u++) if (dst[u]) { o[u].x += p[u].x; o[u].y += p[u].y; o[u].z += p[u].z; o[u].w += p[u].w; }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (dst[u]) *dst[u] = o[u];
    }
  } else if (v_rest != nullptr && row % 4 == 0) {
    // split storage (the reference's two parameters): band 0 -> v_coeffs [N,3], bands 1.. -> v_rest [N,K-1,3], rows of 4-byte
    // alignment.  q4 lanes per row: piece 0 = the three floats of band 0 + the first of the rest, piece c >= 1 = floats 4c-3 .. 4c of
    // the row's rest as ONE 16-byte access at 4-byte alignment (unaligned vector access is on for gfx950 compute); the last float of
    // a row's last piece is the rest's last (row - 3 = 4 q4 - 3 floats: 4 (q4-1) + 1)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int q4 = row / 4, total = cnt * q4;
    for (int e = tid; e < total; e += kShBlock) {
      const int r = e / q4, c = e - r * q4;
      const float *sp = lds + r * ldr + c * 4;
      const int64_t g = s_g[r];
      float *rest = v_rest + g * (row - 3);
      if (c == 0) {
        float *dc = v_coeffs + g * 3;
        if (kAcc) { dc[0] += sp[0]; dc[1] += sp[1]; dc[2] += sp[2]; rest[0] += sp[3]; }
        else { dc[0] = sp[0]; dc[1] = sp[1]; dc[2] = sp[2]; rest[0] = sp[3]; }
      } else {
        f4u *dst = reinterpret_cast<f4u *>(rest + 4 * c - 3);
        f4u o;
        o.x = sp[0]; o.y = sp[1]; o.z = sp[2]; o.w = sp[3];
        if (kAcc) { const f4u p = *dst; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        *dst = o;
      }
    }
  } else {
    for (int e = tid; e < cnt * row; e += kShBlock) {
      const int r = e / row, cc = e - r * row;
      // split storage (v_rest != null): band 0 -> v_coeffs [N,3], bands 1.. -> v_rest [N,K-1,3]
      float *dst = v_rest == nullptr ? v_coeffs + (int64_t)s_g[r] * row + cc
                   : (cc < 3 ? v_coeffs + (int64_t)s_g[r] * 3 + cc : v_rest + (int64_t)s_g[r] * (row - 3) + (cc - 3));
      *dst = kAcc ? *dst + lds[r * ldr + cc] : lds[r * ldr + cc];
    }
  }
}

// zero the rows ids[0..n_list) of the five per-Gaussian gradient arrays (row-wise clear of a persistent buffer: the rows a
// previous view wrote); v_sh rows leave as whole 16-byte pieces, 12 lanes per 192-byte row
template <bool kVec>
__global__ __launch_bounds__(kShBlock) void view_grads_clear_list_kernel(int64_t n_cap, const uint64_t *__restrict__ n_dev,
                                                                        const int32_t *__restrict__ ids, int K,
                                                                        float *__restrict__ v_means, float *__restrict__ v_quats,
                                                                        float *__restrict__ v_log_scales, float *__restrict__ v_logits,
                                                                        float *__restrict__ v_sh, float2 *__restrict__ grad2d,
                                                                        float2 *__restrict__ absgrad2d, const GradLayout gl) {
  __shared__ int32_t s_g[kShBlock];
  const int64_t n_list = list_length(n_cap, n_dev);
  const int64_t r0 = (int64_t)blockIdx.x * kShBlock;
  if (r0 >= n_list) return;
  const int cnt = (int)min((int64_t)kShBlock, n_list - r0);
  const int tid = threadIdx.x;
  if (tid < cnt) {
    const int64_t g = ids[r0 + tid];
    s_g[tid] = (int32_t)g;
    if (g >= 0) {   // negative ids: padding of a fixed-capacity list
      if (v_means) {   // (null: only the screen-space arrays -- the parameter gradients are cleared by their consumer, bds_adam_step_consume)
        if (gl.sm == 16) {      // row form (bds_common.h GradLayout): the whole 64-byte row
          float4 *row = reinterpret_cast<float4 *>(v_means + g * 16);
          row[0] = row[1] = row[2] = row[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          for (int i = 0; i < 3; i++) { v_means[g * 3 + i] = 0.f; v_log_scales[g * 3 + i] = 0.f; }
          for (int i = 0; i < 4; i++) v_quats[g * 4 + i] = 0.f;
          v_logits[g] = 0.f;
        }
      }
      // (the view's persistent screen-space gradient arrays: the list-driven projection backward STORES the visible rows)
      if (grad2d) grad2d[g] = make_float2(0.f, 0.f);
      if (absgrad2d) absgrad2d[g] = make_float2(0.f, 0.f);
    }
  }
  if (!v_sh) return;
  __syncthreads();
  const int row = K * 3;
  if (kVec) {
    const int q4 = row / 4;
    for (int e = tid; e < cnt * q4; e += kShBlock) {
      const int r = e / q4, cc = (e - r * q4) * 4;
      if (s_g[r] >= 0) *reinterpret_cast<float4 *>(v_sh + (int64_t)s_g[r] * row + cc) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int e = tid; e < cnt * row; e += kShBlock) {
      const int r = e / row;
      if (s_g[r] >= 0) v_sh[(int64_t)s_g[r] * row + (e - r * row)] = 0.f;
    }
  }
}

// dst[ids[s]] += src[s] for the five per-Gaussian gradient arrays: compact rows (an exchange buffer in the order of `ids`) added to
// the dense arrays; negative ids (padding of a fixed-capacity list) are skipped.  v_sh rows move as 16-byte pieces, 12 lanes per row.
template <bool kVec>
__global__ __launch_bounds__(kShBlock) void view_grads_add_list_kernel(
    int64_t n_list, const int32_t *__restrict__ ids, int K, const float *__restrict__ s_means, const float *__restrict__ s_quats,
    const float *__restrict__ s_log_scales, const float *__restrict__ s_logits, const float *__restrict__ s_sh,
    float *__restrict__ v_means, float *__restrict__ v_quats, float *__restrict__ v_log_scales, float *__restrict__ v_logits,
    float *__restrict__ v_sh) {
  __shared__ int32_t s_g[kShBlock];
  const int64_t r0 = (int64_t)blockIdx.x * kShBlock;
  const int cnt = (int)min((int64_t)kShBlock, n_list - r0);
  const int tid = threadIdx.x;
  if (tid < cnt) {
    const int64_t r = r0 + tid;
    const int64_t g = ids[r];
    s_g[tid] = (int32_t)g;
    if (g >= 0) {   // all loads first, then the stores (a load of an array may not be moved above a store to it: see sh_view_bwd_list)
      float om[3], os[3], oq[4], ol = v_logits[g] + s_logits[r];
#pragma unroll
      for (int i = 0; i < 3; i++) { om[i] = v_means[g * 3 + i] + s_means[r * 3 + i]; os[i] = v_log_scales[g * 3 + i] + s_log_scales[r * 3 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) oq[i] = v_quats[g * 4 + i] + s_quats[r * 4 + i];
#pragma unroll
      for (int i = 0; i < 3; i++) { v_means[g * 3 + i] = om[i]; v_log_scales[g * 3 + i] = os[i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) v_quats[g * 4 + i] = oq[i];
      v_logits[g] = ol;
    }
  }
  __syncthreads();
  const int row = K * 3;
  if (kVec) {
    const int q4 = row / 4;
    const int total = cnt * q4;
    for (int e0 = tid; e0 < total; e0 += 4 * kShBlock) {   // four pieces per step, their loads in front of the first store
      float4 *d[4];
      float4 o[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * kShBlock;
        d[u] = nullptr;
        if (e < total) {
          const int r = e / q4, cc = (e - r * q4) * 4;
          if (s_g[r] >= 0) {
            const float4 a = *reinterpret_cast<const float4 *>(s_sh + (r0 + r) * row + cc);
            d[u] = reinterpret_cast<float4 *>(v_sh + (int64_t)s_g[r] * row + cc);
            const float4 p = *d[u];
            o[u] = make_float4(p.x + a.x, p.y + a.y, p.z + a.z, p.w + a.w);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (d[u]) *d[u] = o[u];
    }
  } else {
    for (int e = tid; e < cnt * row; e += kShBlock) {
      const int r = e / row;
      if (s_g[r] < 0) continue;
      v_sh[(int64_t)s_g[r] * row + (e - r * row)] += s_sh[(r0 + r) * row + (e - r * row)];
    }
  }
}

}  // namespace bds

using namespace bds;

template <int DEG>
static void launch_fwd(bool full, int grid, size_t lds, hipStream_t st, int64_t n, int K, const float *dirs,
                       const float *coeffs, const uint8_t *masks, float *out) {
  if (masks != nullptr) {
    const bool vec = ((K * 3) % 4 == 0) && aligned16(coeffs);  // every row starts on a 16-byte boundary
    if (vec)
      hipLaunchKernelGGL((sh_fwd_masked_kernel<DEG, true>), dim3(grid), dim3(kShBlock), 0, st, n, K, dirs, coeffs, masks, out);
    else
      hipLaunchKernelGGL((sh_fwd_masked_kernel<DEG, false>), dim3(grid), dim3(kShBlock), 0, st, n, K, dirs, coeffs, masks, out);
    return;
  }
  if (full)
    hipLaunchKernelGGL((sh_fwd_kernel<DEG, true>), dim3(grid), dim3(kShBlock), lds, st, n, K, dirs, coeffs, masks, out);
  else
    hipLaunchKernelGGL((sh_fwd_kernel<DEG, false>), dim3(grid), dim3(kShBlock), lds, st, n, K, dirs, coeffs, masks, out);
}

template <int DEG>
static void launch_bwd(bool full, int grid, size_t lds, hipStream_t st, int64_t n, int K, const float *dirs,
                       const float *coeffs, const uint8_t *masks, const float *v_out, float *v_coeffs, float *v_dirs) {
  if (full)
    hipLaunchKernelGGL((sh_bwd_kernel<DEG, true>), dim3(grid), dim3(kShBlock), lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs);
  else
    hipLaunchKernelGGL((sh_bwd_kernel<DEG, false>), dim3(grid), dim3(kShBlock), lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs);
}

extern "C" int bds_sh_fwd(int64_t n, int K, int deg, const float *dirs, const float *coeffs, const uint8_t *masks,
                          float *out, bds_stream_t stream) {
  BDS_REQUIRE(n >= 0 && deg >= 0 && deg <= 3 && K >= (deg + 1) * (deg + 1) && K <= 16);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(dirs && coeffs && out);
  const int nb = (deg + 1) * (deg + 1);
  const int grid = (int)cdiv(n, kShBlock);
  const bool full = (nb == K) && ((nb * 3) % 4 == 0) && aligned16(coeffs);
  const size_t lds = (size_t)kShBlock * (nb * 3 + (full ? 4 : 1)) * sizeof(float);
  hipStream_t st = as_stream(stream);
  switch (deg) {
    case 0: launch_fwd<0>(full, grid, lds, st, n, K, dirs, coeffs, masks, out); break;
    case 1: launch_fwd<1>(full, grid, lds, st, n, K, dirs, coeffs, masks, out); break;
    case 2: launch_fwd<2>(full, grid, lds, st, n, K, dirs, coeffs, masks, out); break;
    default: launch_fwd<3>(full, grid, lds, st, n, K, dirs, coeffs, masks, out); break;
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_sh_bwd(int64_t n, int K, int deg, const float *dirs, const float *coeffs, const uint8_t *masks,
                          const float *v_out, float *v_coeffs, float *v_dirs, bds_stream_t stream) {
  BDS_REQUIRE(n >= 0 && deg >= 0 && deg <= 3 && K >= (deg + 1) * (deg + 1) && K <= 16);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(dirs && coeffs && v_out && v_coeffs);
  const int grid = (int)cdiv(n, kShBlock);
  const size_t lds = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
  const bool full = ((K * 3) % 4 == 0) && aligned16(v_coeffs);
  hipStream_t st = as_stream(stream);
  switch (deg) {
    case 0: launch_bwd<0>(full, grid, lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs); break;
    case 1: launch_bwd<1>(full, grid, lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs); break;
    case 2: launch_bwd<2>(full, grid, lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs); break;
    default: launch_bwd<3>(full, grid, lds, st, n, K, dirs, coeffs, masks, v_out, v_coeffs, v_dirs); break;
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

template <int DEG>
static void launch_view_fwd(bool vec, int grid, hipStream_t st, int64_t n, int K, const float *means, const float *cam_pos,
                            const float *coeffs, const int32_t *radii, const float *depths, float *sh_rgb, float4 *colors) {
  if (vec)
    hipLaunchKernelGGL((sh_view_fwd_kernel<DEG, true>), dim3(grid), dim3(kShBlock), 0, st, n, K, means, cam_pos, coeffs, radii,
                       depths, sh_rgb, colors);
  else
    hipLaunchKernelGGL((sh_view_fwd_kernel<DEG, false>), dim3(grid), dim3(kShBlock), 0, st, n, K, means, cam_pos, coeffs, radii,
                       depths, sh_rgb, colors);
}

extern "C" int bds_sh_view_fwd(int64_t n, int K, int deg, const float *means, const float *cam_pos, const float *coeffs,
                               const int32_t *radii, const float *depths, float *sh_rgb, float *colors, bds_stream_t stream) {
  BDS_REQUIRE(n >= 0 && deg >= 0 && deg <= 3 && K >= (deg + 1) * (deg + 1) && K <= 16);
  if (n == 0) return BDS_OK;
  BDS_REQUIRE(means && cam_pos && coeffs && radii && depths && sh_rgb && colors && aligned16(colors));
  const int grid = (int)cdiv(n, kShBlock);
  const bool vec = ((K * 3) % 4 == 0) && aligned16(coeffs);
  hipStream_t st = as_stream(stream);
  float4 *c4 = reinterpret_cast<float4 *>(colors);
  switch (deg) {
    case 0: launch_view_fwd<0>(vec, grid, st, n, K, means, cam_pos, coeffs, radii, depths, sh_rgb, c4); break;
    case 1: launch_view_fwd<1>(vec, grid, st, n, K, means, cam_pos, coeffs, radii, depths, sh_rgb, c4); break;
    case 2: launch_view_fwd<2>(vec, grid, st, n, K, means, cam_pos, coeffs, radii, depths, sh_rgb, c4); break;
    default: launch_view_fwd<3>(vec, grid, st, n, K, means, cam_pos, coeffs, radii, depths, sh_rgb, c4); break;
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

template <int DEG>
static void launch_view_bwd_list(bool vec, bool acc, int grid, size_t lds, hipStream_t st, int64_t n_list, const uint64_t *n_dev,
                                 const int32_t *ids, int K,
                                 const float *means, const float *cam_pos, const float *sh_rgb, const float4 *v_rec, float *v_coeffs,
                                 const int32_t *row_map, int by_rank, float *v_rest) {
#define BDS_LIST(V, A)                                                                                                           \
  hipLaunchKernelGGL((sh_view_bwd_list_kernel<DEG, V, A>), dim3(grid), dim3(kShBlock), lds, st, n_list, n_dev, ids, K, means, cam_pos, \
                     sh_rgb, v_rec, v_coeffs, row_map, by_rank, v_rest)
  if (vec) { if (acc) BDS_LIST(true, true); else BDS_LIST(true, false); }
  else     { if (acc) BDS_LIST(false, true); else BDS_LIST(false, false); }
#undef BDS_LIST
}

static int sh_view_bwd_list_impl(int64_t n_list, const uint64_t *n_dev, const int32_t *ids, int K, int deg, const float *means,
                                 const float *cam_pos, const float *sh_rgb, int sh_rgb_by_rank, const float *v_records, float *v_coeffs,
                                 const int32_t *row_map, int accumulate, bds_stream_t stream, float *v_rest = nullptr) {
  BDS_REQUIRE(n_list >= 0 && deg >= 0 && deg <= 3 && K >= (deg + 1) * (deg + 1) && K <= 16);
  if (n_list == 0) return BDS_OK;
  BDS_REQUIRE(ids && means && cam_pos && sh_rgb && v_records && v_coeffs && aligned16(v_records));
  const int grid = (int)cdiv(n_list, kShBlock);
  const size_t lds = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
  const bool vec = ((K * 3) % 4 == 0) && aligned16(v_coeffs) && v_rest == nullptr;
  hipStream_t st = as_stream(stream);
  const float4 *v4 = reinterpret_cast<const float4 *>(v_records);
  switch (deg) {
    case 0: launch_view_bwd_list<0>(vec, accumulate != 0, grid, lds, st, n_list, n_dev, ids, K, means, cam_pos, sh_rgb, v4, v_coeffs, row_map, sh_rgb_by_rank, v_rest); break;
    case 1: launch_view_bwd_list<1>(vec, accumulate != 0, grid, lds, st, n_list, n_dev, ids, K, means, cam_pos, sh_rgb, v4, v_coeffs, row_map, sh_rgb_by_rank, v_rest); break;
    case 2: launch_view_bwd_list<2>(vec, accumulate != 0, grid, lds, st, n_list, n_dev, ids, K, means, cam_pos, sh_rgb, v4, v_coeffs, row_map, sh_rgb_by_rank, v_rest); break;
    default: launch_view_bwd_list<3>(vec, accumulate != 0, grid, lds, st, n_list, n_dev, ids, K, means, cam_pos, sh_rgb, v4, v_coeffs, row_map, sh_rgb_by_rank, v_rest); break;
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_sh_view_bwd_list(int64_t n_list, const int32_t *ids, int K, int deg, const float *means, const float *cam_pos,
                                    const float *sh_rgb, int sh_rgb_by_rank, const float *v_records, float *v_coeffs,
                                    const int32_t *row_map, int accumulate, bds_stream_t stream) {
  return sh_view_bwd_list_impl(n_list, nullptr, ids, K, deg, means, cam_pos, sh_rgb, sh_rgb_by_rank, v_records, v_coeffs, row_map,
                               accumulate, stream);
}

extern "C" int bds_sh_view_bwd_list_split(int64_t n_list, const int32_t *ids, int K, int deg, const float *means, const float *cam_pos,
                                          const float *sh_rgb, int sh_rgb_by_rank, const float *v_records, float *v_coeffs_dc,
                                          float *v_coeffs_rest, int accumulate, bds_stream_t stream) {
  BDS_REQUIRE(n_list == 0 || (v_coeffs_dc && (v_coeffs_rest || K == 1)));
  return sh_view_bwd_list_impl(n_list, nullptr, ids, K, deg, means, cam_pos, sh_rgb, sh_rgb_by_rank, v_records, v_coeffs_dc, nullptr,
                               accumulate, stream, K == 1 ? nullptr : v_coeffs_rest);
}

extern "C" int bds_sh_view_bwd_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int K, int deg, const float *means,
                                        const float *cam_pos, const float *sh_rgb, int sh_rgb_by_rank, const float *v_records,
                                        float *v_coeffs, const int32_t *row_map, int accumulate, bds_stream_t stream) {
  BDS_REQUIRE(n_dev);
  return sh_view_bwd_list_impl(n_capacity, n_dev, ids, K, deg, means, cam_pos, sh_rgb, sh_rgb_by_rank, v_records, v_coeffs, row_map,
                               accumulate, stream);
}

static int view_grads_clear_list_impl(int64_t n_list, const uint64_t *n_dev, const int32_t *ids, int K, float *v_means, float *v_quats,
                                      float *v_log_scales, float *v_logits, float *v_sh, bds_stream_t stream, float *grad2d = nullptr,
                                      float *absgrad2d = nullptr) {
  BDS_REQUIRE(n_list >= 0 && K >= 1 && K <= 16);
  if (n_list == 0) return BDS_OK;
  const bool params = v_means || v_quats || v_log_scales || v_logits || v_sh;    // all five, or none (then the screen-space arrays only)
  BDS_REQUIRE(ids && (params ? (v_means && v_quats && v_log_scales && v_logits && v_sh) : (grad2d || absgrad2d)));
  BDS_REQUIRE((reinterpret_cast<uintptr_t>(grad2d) & 7u) == 0 && (reinterpret_cast<uintptr_t>(absgrad2d) & 7u) == 0);
  float2 *g2 = reinterpret_cast<float2 *>(grad2d), *a2 = reinterpret_cast<float2 *>(absgrad2d);
  const dim3 grid((unsigned)cdiv(n_list, kShBlock)), block(kShBlock);
  const GradLayout gl = grad_layout(v_means, v_quats, v_log_scales, v_logits);
  BDS_REQUIRE(gl.sm != 16 || aligned16(v_means));
  if (((K * 3) % 4 == 0) && aligned16(v_sh))
    hipLaunchKernelGGL((view_grads_clear_list_kernel<true>), grid, block, 0, as_stream(stream), n_list, n_dev, ids, K, v_means, v_quats,
                       v_log_scales, v_logits, v_sh, g2, a2, gl);
  else
    hipLaunchKernelGGL((view_grads_clear_list_kernel<false>), grid, block, 0, as_stream(stream), n_list, n_dev, ids, K, v_means, v_quats,
                       v_log_scales, v_logits, v_sh, g2, a2, gl);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_view_grads_clear_list(int64_t n_list, const int32_t *ids, int K, float *v_means, float *v_quats,
                                         float *v_log_scales, float *v_logits, float *v_sh, bds_stream_t stream) {
  return view_grads_clear_list_impl(n_list, nullptr, ids, K, v_means, v_quats, v_log_scales, v_logits, v_sh, stream);
}

extern "C" int bds_view_grads_clear_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int K, float *v_means,
                                             float *v_quats, float *v_log_scales, float *v_logits, float *v_sh, float *grad2d,
                                             float *absgrad2d, bds_stream_t stream) {
  BDS_REQUIRE(n_dev);
  return view_grads_clear_list_impl(n_capacity, n_dev, ids, K, v_means, v_quats, v_log_scales, v_logits, v_sh, stream, grad2d, absgrad2d);
}

extern "C" int bds_view_grads_add_list(int64_t n_list, const int32_t *ids, int K, const float *s_means, const float *s_quats,
                                       const float *s_log_scales, const float *s_logits, const float *s_sh, float *v_means,
                                       float *v_quats, float *v_log_scales, float *v_logits, float *v_sh, bds_stream_t stream) {
  BDS_REQUIRE(n_list >= 0 && K >= 1 && K <= 16);
  if (n_list == 0) return BDS_OK;
  BDS_REQUIRE(ids && s_means && s_quats && s_log_scales && s_logits && s_sh && v_means && v_quats && v_log_scales && v_logits && v_sh);
  const dim3 grid((unsigned)cdiv(n_list, kShBlock)), block(kShBlock);
  if (((K * 3) % 4 == 0) && aligned16(v_sh) && aligned16(s_sh))
    hipLaunchKernelGGL((view_grads_add_list_kernel<true>), grid, block, 0, as_stream(stream), n_list, ids, K, s_means, s_quats,
                       s_log_scales, s_logits, s_sh, v_means, v_quats, v_log_scales, v_logits, v_sh);
  else
    hipLaunchKernelGGL((view_grads_add_list_kernel<false>), grid, block, 0, as_stream(stream), n_list, ids, K, s_means, s_quats,
                       s_log_scales, s_logits, s_sh, v_means, v_quats, v_log_scales, v_logits, v_sh);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
