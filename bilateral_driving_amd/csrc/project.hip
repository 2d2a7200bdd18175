// K2/K3: per-Gaussian 3D->2D projection, forward and backward.
// First stage of gsplat.rendering.rasterization as called at
// /root/reference/project/models/trainers/base.py:393-408.  HBM-bound streaming kernels
// (68 B/Gaussian forward, 108 B/Gaussian backward).  One thread per Gaussian; the camera loop
// runs inside the thread so that parameter gradients are summed in registers (no atomics,
// deterministic); only the 12-float camera-pose gradient is block-reduced.
#include "bds_common.h"
#include "gs_math.h"

namespace bds {

constexpr int kProjBlock = 256;

__global__ __launch_bounds__(kProjBlock) void project_fwd_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane,
    float far_plane, float radius_clip, int32_t *__restrict__ radii, float *__restrict__ means2d,
    float *__restrict__ depths, float *__restrict__ conics, float *__restrict__ comps) {
  const int64_t g = (int64_t)blockIdx.x * kProjBlock + threadIdx.x;
  if (g >= N) return;
  float m[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
  float q[4] = {quats[g * 4], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
  float s[3] = {scales[g * 3], scales[g * 3 + 1], scales[g * 3 + 2]};
  for (int c = 0; c < C; c++) {
    Camera cam = load_camera(viewmats + c * 16, Ks + c * 9);  // wave-uniform -> scalar loads
    Proj p = project_one(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip);
    const int64_t o = (int64_t)c * N + g;
    radii[o] = p.radius;
    means2d[o * 2] = p.mx; means2d[o * 2 + 1] = p.my;
    depths[o] = p.depth;
    conics[o * 3] = p.ca; conics[o * 3 + 1] = p.cb; conics[o * 3 + 2] = p.cc;
    if (comps) comps[o] = p.comp;
  }
}

__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(kProjBlock) void project_bwd_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d,
    const int32_t *__restrict__ radii, const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
    const float *__restrict__ v_conics, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_viewmats) {
  __shared__ float red[kProjBlock / kWave][12];
  const int64_t g = (int64_t)blockIdx.x * kProjBlock + threadIdx.x;
  const bool live = g < N;
  float m[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, s[3] = {1, 1, 1};
  if (live) {
    m[0] = means[g * 3]; m[1] = means[g * 3 + 1]; m[2] = means[g * 3 + 2];
    q[0] = quats[g * 4]; q[1] = quats[g * 4 + 1]; q[2] = quats[g * 4 + 2]; q[3] = quats[g * 4 + 3];
    s[0] = scales[g * 3]; s[1] = scales[g * 3 + 1]; s[2] = scales[g * 3 + 2];
  }
  float am[3] = {0, 0, 0}, aq[4] = {0, 0, 0, 0}, as[3] = {0, 0, 0};
  for (int c = 0; c < C; c++) {
    Camera cam = load_camera(viewmats + c * 16, Ks + c * 9);
    ProjGrad pg;
    for (int i = 0; i < 9; i++) pg.v_R[i] = 0.f;
    for (int i = 0; i < 3; i++) pg.v_t[i] = 0.f;
    const int64_t o = (int64_t)c * N + g;
    if (live && radii[o] > 0) {
      project_one_vjp(m, q, s, cam, W, H, eps2d, v_means2d[o * 2], v_means2d[o * 2 + 1], v_depths[o], v_conics[o * 3],
                      v_conics[o * 3 + 1], v_conics[o * 3 + 2], pg);
      for (int i = 0; i < 3; i++) { am[i] += pg.v_mean[i]; as[i] += pg.v_scale[i]; }
      for (int i = 0; i < 4; i++) aq[i] += pg.v_quat[i];
    }
    if (v_viewmats != nullptr) {  // block reduction of the pose gradient, one atomic set per block
      const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
      float r[12];
      for (int i = 0; i < 9; i++) r[i] = wave_sum_shfl(pg.v_R[i]);
      for (int i = 0; i < 3; i++) r[9 + i] = wave_sum_shfl(pg.v_t[i]);
      __syncthreads();
      if (lane == 0)
        for (int i = 0; i < 12; i++) red[wv][i] = r[i];
      __syncthreads();
      if (threadIdx.x < 12) {
        float t = 0.f;
        for (int w = 0; w < kProjBlock / kWave; w++) t += red[w][threadIdx.x];
        const int i = threadIdx.x;
        float *dst = v_viewmats + c * 16 + (i < 9 ? (i / 3) * 4 + (i % 3) : (i - 9) * 4 + 3);
        if (t != 0.f) atomicAdd(dst, t);
      }
    }
  }
  if (live) {
    for (int i = 0; i < 3; i++) { v_means[g * 3 + i] = am[i]; v_scales[g * 3 + i] = as[i]; }
    for (int i = 0; i < 4; i++) v_quats[g * 4 + i] = aq[i];
  }
}

// ---- one-view variants for the fused training step: activations folded in --------------------------------
// scales = exp(log_scales), opacities = sigmoid(logits) (models/gaussians/vanilla.py:393-394) are produced by the
// projection itself (they are also outputs: the compositor and the backward need them), and the backward returns
// the gradients of the RAW parameters.  A culled Gaussian reads only its radius and writes zeros.
// kReduce: the launch also does the tile stage's first one (visible_reduce_kernel, csrc/tiles.hip): visible Gaussians per workgroup,
// the sort tables and tiles_per_gauss cleared -- it reads every radius anyway.
template <bool kReduce>
__global__ __launch_bounds__(kProjBlock) void project_view_fwd_kernel(
    int64_t N, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ log_scales,
    const float *__restrict__ logits, const float *__restrict__ viewmat, const float *__restrict__ K, int W, int H,
    float eps2d, float near_plane, float far_plane, float radius_clip, float *__restrict__ scales,
    float *__restrict__ opacities, int32_t *__restrict__ radii, float *__restrict__ means2d, float *__restrict__ depths,
    float *__restrict__ conics, PrepReduceSlots rs, int32_t *__restrict__ tiles_per_gauss, int rows,
    const float *__restrict__ block_bounds) {
  const int64_t g = (int64_t)blockIdx.x * kProjBlock + threadIdx.x;
  if (kReduce) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *rs.m_total = 0;
    for (int64_t i = g; i < rs.zero_elems; i += (int64_t)gridDim.x * kProjBlock) rs.zero_me[i] = 0u;
  }
  // Block bound (bds_gaussian_block_bounds: this workgroup's 256 rows lie in a box, none is larger than smax): when no centre in the
  // box can come out visible, the rows are not even read -- a culled Gaussian's outputs are zeros.  (Rows kept in spatial order,
  // densify.spatial_order, make the boxes small: a camera then rejects most of the ~85 % it does not see block-wise.)
  if (block_bounds != nullptr) {
    const float *bb = block_bounds + (int64_t)blockIdx.x * 8;
    const float lo[3] = {bb[0], bb[1], bb[2]}, hi[3] = {bb[4], bb[5], bb[6]};
    if (!box_may_be_visible(lo, hi, bb[3], load_camera(viewmat, K), W, H, eps2d, near_plane, far_plane)) {
      if (g < N) {
        radii[g] = 0;
        // (dense [N] outputs a later dense consumer may read -- render_classes' opacity x mask: never uninitialised memory)
        scales[g * 3] = 0.f; scales[g * 3 + 1] = 0.f; scales[g * 3 + 2] = 0.f; opacities[g] = 0.f;
        if (rows) {
          float4 *row = reinterpret_cast<float4 *>(means2d + g * 8);
          row[0] = row[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          means2d[g * 2] = 0.f; means2d[g * 2 + 1] = 0.f;
          depths[g] = 0.f;
          conics[g * 3] = 0.f; conics[g * 3 + 1] = 0.f; conics[g * 3 + 2] = 0.f;
        }
        if (kReduce && tiles_per_gauss) tiles_per_gauss[g] = 0;
      }
      if (kReduce && threadIdx.x == 0) rs.sums256[blockIdx.x] = 0u;
      return;
    }
  }
  if (kReduce) {
    int radius = 0;
    if (g < N) {
      float m[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
      float q[4] = {quats[g * 4], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
      float s[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        s[k] = expf(log_scales[g * 3 + k]);
        scales[g * 3 + k] = s[k];
      }
      const float op = 1.f / (1.f + expf(-logits[g]));
      opacities[g] = op;
      Camera cam = load_camera(viewmat, K);
      Proj p = project_one(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip);
      radius = p.radius;
      radii[g] = p.radius;
      if (rows) {      // (bds_common.h ProjLayout: one 32-byte row per Gaussian; radii / opacities stay dense arrays as well)
        float4 *row = reinterpret_cast<float4 *>(means2d + g * 8);
        row[0] = make_float4(p.mx, p.my, p.depth, __int_as_float(p.radius));
        row[1] = make_float4(p.ca, p.cb, p.cc, op);
      } else {
        means2d[g * 2] = p.mx; means2d[g * 2 + 1] = p.my;
        depths[g] = p.depth;
        conics[g * 3] = p.ca; conics[g * 3 + 1] = p.cb; conics[g * 3 + 2] = p.cc;
      }
      if (tiles_per_gauss) tiles_per_gauss[g] = 0;   // (the counting kernel writes the visible entries only)
    }
    const int cnt = __syncthreads_count(radius > 0);
    if (threadIdx.x == 0) rs.sums256[blockIdx.x] = (uint32_t)cnt;
    return;
  }
  if (g >= N) return;
  float m[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
  float q[4] = {quats[g * 4], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
  float s[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    s[k] = expf(log_scales[g * 3 + k]);
    scales[g * 3 + k] = s[k];
  }
  const float op = 1.f / (1.f + expf(-logits[g]));
  opacities[g] = op;
  Camera cam = load_camera(viewmat, K);
  Proj p = project_one(m, q, s, cam, W, H, eps2d, near_plane, far_plane, radius_clip);
  radii[g] = p.radius;
  if (rows) {
    float4 *row = reinterpret_cast<float4 *>(means2d + g * 8);
    row[0] = make_float4(p.mx, p.my, p.depth, __int_as_float(p.radius));
    row[1] = make_float4(p.ca, p.cb, p.cc, op);
  } else {
    means2d[g * 2] = p.mx; means2d[g * 2 + 1] = p.my;
    depths[g] = p.depth;
    conics[g * 3] = p.ca; conics[g * 3 + 1] = p.cb; conics[g * 3 + 2] = p.cc;
  }
}

// Block reduction of the camera-pose gradient (9 rotation + 3 translation partials per Gaussian): one 16-value
// transpose-reduce per wave, the workgroup's waves summed through LDS, one atomic set per workgroup into one of
// kPoseSlots replicated [4,4] accumulators (every workgroup on the same 12 addresses serialises in L2: measured +58 us
// at 7800 workgroups); the caller sums the slots.
constexpr int kPoseSlots = BDS_POSE_GRAD_SLOTS;
__device__ __forceinline__ void pose_grad_reduce(const ProjGrad &pg, float (*red)[12], float *__restrict__ v_viewmat_slots) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  float v[16];
#pragma unroll
  for (int i = 0; i < 9; i++) v[i] = pg.v_R[i];
#pragma unroll
  for (int i = 0; i < 3; i++) v[9 + i] = pg.v_t[i];
  v[12] = v[13] = v[14] = v[15] = 0.f;
  const float tot = butterfly_sum16(v, lane);
  const int k = butterfly_slot(lane);
  if ((lane & 3) == 0 && k < 12) red[wv][k] = tot;
  __syncthreads();
  if (threadIdx.x < 12) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kProjBlock / kWave; w++) t += red[w][threadIdx.x];
    const int i = threadIdx.x;
    float *dst = v_viewmat_slots + (blockIdx.x % kPoseSlots) * 16;
    if (t != 0.f) atomicAdd(dst + (i < 9 ? (i / 3) * 4 + (i % 3) : (i - 9) * 4 + 3), t);
  }
}

// ---- one-view backward over the VISIBLE Gaussians, list-driven ----------------------------------------------------------------
// Work list = the depth-ordered ids of the visible entries (bds_isect_build: visible_ids); input = the compositor's gradient
// record of the same rank (conic 4-6, mean2d 7-8, |mean2d| 9-10, opacity 11, depth channel 3), read coalesced.  Returns the
// gradients of the RAW parameters (log-scales, opacity logits: models/gaussians/vanilla.py:393-394), stored (kAcc = false) or added
// (kAcc = true) to the rows of the visible Gaussians only -- rows of culled Gaussians are the caller's business.  Optionally
// scatters the screen-space gradient and its absolute sum to the dense [N,2] arrays the densification statistics read
// (models/trainers/base.py:280-297), and reduces the camera-pose gradient (models/trainers/base.py:328-329,399).
// kRaw = false (the gsplat-shaped operator, rendering.rasterization): the caller passed ACTIVATED scales / opacities -- their gradients
// are returned as they are -- and post-activation colours, whose gradient (record channels 0-2) is scattered to v_colors [N,3].
template <bool kAcc, bool kPose, bool kRaw = true>
__global__ __launch_bounds__(kProjBlock) void project_view_bwd_list_kernel(
    int64_t n_cap, const uint64_t *__restrict__ n_dev, const int32_t *__restrict__ ids, const float *__restrict__ means,
    const float *__restrict__ quats, const float *__restrict__ scales, const float *__restrict__ opacities, const float *__restrict__ viewmat,
    const float *__restrict__ K, int W, int H, float eps2d, const float4 *__restrict__ v_rec, float *__restrict__ v_means,
    float *__restrict__ v_quats, float *__restrict__ v_log_scales, float *__restrict__ v_logits, float *__restrict__ v_viewmat_slots,
    float *__restrict__ grad2d, float *__restrict__ absgrad2d, const int32_t *__restrict__ row_map, float *__restrict__ v_colors = nullptr,
    const GradLayout gl = GradLayout{3, 4, 3, 1}) {
  __shared__ float red[kProjBlock / kWave][12];
  const int64_t n_list = list_length(n_cap, n_dev);
  if ((int64_t)blockIdx.x * kProjBlock >= n_list) return;   // (whole workgroup: nothing to add to the pose slots either)
  const int64_t r = (int64_t)blockIdx.x * kProjBlock + threadIdx.x;
  ProjGrad pg;
  for (int i = 0; i < 9; i++) pg.v_R[i] = 0.f;
  for (int i = 0; i < 3; i++) pg.v_t[i] = 0.f;
  if (r < n_list) {
    const int64_t g = ids[r];
    const float4 r0 = v_rec[r * 4], r1 = v_rec[r * 4 + 1], r2 = v_rec[r * 4 + 2];
    // every gather that depends on g is issued here, in front of the arithmetic -- including, in accumulate mode, the eleven old
    // gradient values: read next to the stores that update them, each would wait out a full memory latency behind the previous
    // store (the compiler may not move a load of an array above a store to it)
    const float m[3] = {means[g * 3], means[g * 3 + 1], means[g * 3 + 2]};
    const float q[4] = {quats[g * 4], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
    const float s[3] = {scales[g * 3], scales[g * 3 + 1], scales[g * 3 + 2]};
    const float o = opacities[g];
    const int64_t d = row_map ? (int64_t)row_map[g] : g;   // destination row of the parameter gradients
    float old_m[3] = {0.f, 0.f, 0.f}, old_s[3] = {0.f, 0.f, 0.f}, old_q[4] = {0.f, 0.f, 0.f, 0.f}, old_l = 0.f;
    if (kAcc) {
#pragma unroll
      for (int i = 0; i < 3; i++) { old_m[i] = v_means[d * gl.sm + i]; old_s[i] = v_log_scales[d * gl.ss + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) old_q[i] = v_quats[d * gl.sq + i];
      old_l = v_logits[d * gl.sl];
    }
    Camera cam = load_camera(viewmat, K);
    project_one_vjp(m, q, s, cam, W, H, eps2d, r1.w, r2.x, /*v_depth*/ r0.w, r1.x, r1.y, r1.z, pg);
    const float al = kRaw ? r2.w * o * (1.f - o) : r2.w;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      v_means[d * gl.sm + i] = old_m[i] + pg.v_mean[i];
      v_log_scales[d * gl.ss + i] = old_s[i] + (kRaw ? pg.v_scale[i] * s[i] : pg.v_scale[i]);
    }
    if (!kRaw && v_colors) { v_colors[g * 3] = r0.x; v_colors[g * 3 + 1] = r0.y; v_colors[g * 3 + 2] = r0.z; }
#pragma unroll
    for (int i = 0; i < 4; i++) v_quats[d * gl.sq + i] = old_q[i] + pg.v_quat[i];
    v_logits[d * gl.sl] = old_l + al;
    if (grad2d) { grad2d[g * 2] = r1.w; grad2d[g * 2 + 1] = r2.x; }
    if (absgrad2d) { absgrad2d[g * 2] = r2.y; absgrad2d[g * 2 + 1] = r2.z; }
  }
  // (A "last workgroup sums the slots" epilogue was measured and dropped: the __threadfence() it needs in every workgroup costs an
  // L2 write-back each on gfx950 -- 52 -> 295 us for this kernel; the caller sums the 64 slots with one small reduction instead.)
  if (kPose) pose_grad_reduce(pg, red, v_viewmat_slots);
}

// ---- NaN / Inf check of a Gaussian class's tensors (models/gaussians/vanilla.py:407-412 raises per tensor: 2 reductions + 2 host
// waits each; here one streaming launch over up to 8 tensors that ORs bit t into a flag word when tensor t holds a non-finite value)
struct FiniteArgs {
  const uint32_t *p[8];
  int64_t n[8];
  int kind[8];   // BDS_FINITE_*: what makes the ACTIVATED value of an element non-finite (bds_nonfinite_flags_kinds)
  int count;
};
constexpr int kFiniteBlock = 256;
// kind 0 (plain): NaN or +-Inf.  kind 1 (argument of exp -- log-scales): NaN, +Inf, or a finite value whose exp overflows
// (x >= 88.72284, the first float with exp(x) = Inf); -Inf is fine (exp = 0).  kind 2 (quaternions, rows of four, q / |q|): NaN or
// +-Inf components (Inf / Inf), or an all-zero row (0 / 0).  kind 3 (argument of sigmoid -- logits): NaN only.
template <int kKind>
__device__ __forceinline__ bool nonfinite_bits(uint32_t x) {
  if (kKind == 1) return ((x & 0x7f800000u) == 0x7f800000u && x != 0xff800000u) || (x < 0x80000000u && x >= 0x42b17218u);
  if (kKind == 3) return (x & 0x7fffffffu) > 0x7f800000u;
  return (x & 0x7f800000u) == 0x7f800000u;
}
template <int kKind>
__device__ __forceinline__ uint32_t nonfinite4(uint4 v) {
  uint32_t b = (uint32_t)nonfinite_bits<kKind>(v.x) | (uint32_t)nonfinite_bits<kKind>(v.y) | (uint32_t)nonfinite_bits<kKind>(v.z) |
               (uint32_t)nonfinite_bits<kKind>(v.w);
  if (kKind == 2) {   // (an aligned 16-byte piece of a [N,4] array is a row.)  |q|^2 == 0 in fp32 -- all zeros, or components so small that
    //                     their squares underflow, as torch's norm does: q / 0 is NaN / Inf
    const float x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z), w = __uint_as_float(v.w);
    const float n2 = x * x + y * y + z * z + w * w;
    b |= (uint32_t)(!(n2 > 0.f));
  }
  return b;
}
template <int kKind>
__device__ __forceinline__ uint32_t nonfinite_tensor(const uint32_t *q, int64_t n, int64_t gtid) {
  constexpr int64_t kPiece = 4 * kFiniteBlock;
  const int64_t head = min(n, (int64_t)((16 - (reinterpret_cast<uintptr_t>(q) & 15u)) & 15u) / 4);   // floats in front of a 16-byte boundary
  const int64_t n4 = (n - head) / 4;
  const uint4 *q4 = reinterpret_cast<const uint4 *>(q + head);
  uint32_t b = 0;
  for (int64_t base = (int64_t)blockIdx.x * kPiece; base < n4; base += (int64_t)gridDim.x * kPiece) {
    const int64_t i = base + threadIdx.x;
    if (base + kPiece <= n4) {
      // (read once: non-temporal loads -- 84 -> 76 us over the 472 MB of a 2 M-Gaussian parameter set)
      typedef uint32_t v4u __attribute__((ext_vector_type(4)));
      const v4u *qn = reinterpret_cast<const v4u *>(q4);
      const v4u a0 = __builtin_nontemporal_load(qn + i), a1 = __builtin_nontemporal_load(qn + i + kFiniteBlock),
                a2 = __builtin_nontemporal_load(qn + i + 2 * kFiniteBlock), a3 = __builtin_nontemporal_load(qn + i + 3 * kFiniteBlock);
      const uint4 v0 = make_uint4(a0.x, a0.y, a0.z, a0.w), v1 = make_uint4(a1.x, a1.y, a1.z, a1.w), v2 = make_uint4(a2.x, a2.y, a2.z, a2.w),
                  v3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
      b |= (nonfinite4<kKind>(v0) | nonfinite4<kKind>(v1)) | (nonfinite4<kKind>(v2) | nonfinite4<kKind>(v3));
    } else {
      for (int64_t k = i; k < n4; k += kFiniteBlock) b |= nonfinite4<kKind>(q4[k]);
    }
  }
  if (gtid < head) b |= (uint32_t)nonfinite_bits<kKind>(q[gtid]);
  const int64_t tail0 = head + n4 * 4;
  if (gtid < n - tail0) b |= (uint32_t)nonfinite_bits<kKind>(q[tail0 + gtid]);
  return b;
}
__global__ __launch_bounds__(kFiniteBlock) void nonfinite_flags_kernel(FiniteArgs A, uint32_t *__restrict__ flags) {
  // a workgroup reads CONTIGUOUS 16 KB pieces (four 16-byte loads per thread in flight), grid-strided piece by piece
  const int64_t gtid = (int64_t)blockIdx.x * kFiniteBlock + threadIdx.x;
  uint32_t bad = 0;
  for (int t = 0; t < A.count; t++) {
    uint32_t b;
    switch (A.kind[t]) {
      case 1: b = nonfinite_tensor<1>(A.p[t], A.n[t], gtid); break;
      case 2: b = nonfinite_tensor<2>(A.p[t], A.n[t], gtid); break;
      case 3: b = nonfinite_tensor<3>(A.p[t], A.n[t], gtid); break;
      default: b = nonfinite_tensor<0>(A.p[t], A.n[t], gtid); break;
    }
    if (b) bad |= 1u << t;
  }
  if (bad) atomicOr(flags, bad);
}

}  // namespace bds

using namespace bds;

extern "C" int bds_project_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                               const float *viewmats, const float *Ks, int W, int H, float eps2d, float near_plane,
                               float far_plane, float radius_clip, int32_t *radii, float *means2d, float *depths,
                               float *conics, float *compensations, bds_stream_t stream) {
  BDS_REQUIRE(C >= 1 && N >= 0 && W > 0 && H > 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(means && quats && scales && viewmats && Ks && radii && means2d && depths && conics);
  hipLaunchKernelGGL(project_fwd_kernel, dim3((unsigned)cdiv(N, kProjBlock)), dim3(kProjBlock), 0, as_stream(stream), C, N,
                     means, quats, scales, viewmats, Ks, W, H, eps2d, near_plane, far_plane, radius_clip, radii,
                     means2d, depths, conics, compensations);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_project_bwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                               const float *viewmats, const float *Ks, int W, int H, float eps2d, const int32_t *radii,
                               const float *conics, const float *compensations, const float *v_means2d,
                               const float *v_depths, const float *v_conics, const float *v_compensations,
                               float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                               bds_stream_t stream) {
  (void)conics; (void)compensations;
  BDS_REQUIRE(C >= 1 && N >= 0 && W > 0 && H > 0);
  BDS_REQUIRE(v_compensations == nullptr);  // "antialiased" backward is not on the reference's path
  if (v_viewmats) {
    if (hipMemsetAsync(v_viewmats, 0, sizeof(float) * 16 * C, as_stream(stream)) != hipSuccess) return BDS_ELAUNCH;
  }
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(means && quats && scales && viewmats && Ks && radii && v_means2d && v_depths && v_conics && v_means &&
              v_quats && v_scales);
  hipLaunchKernelGGL(project_bwd_kernel, dim3((unsigned)cdiv(N, kProjBlock)), dim3(kProjBlock), 0, as_stream(stream), C, N,
                     means, quats, scales, viewmats, Ks, W, H, eps2d, radii, v_means2d, v_depths, v_conics, v_means,
                     v_quats, v_scales, v_viewmats);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

// the [N,8] row form of the projection outputs (bds_common.h ProjLayout): depths and conics are then columns of means2d's block
static int view_rows(const float *means2d, const float *depths, const float *conics, int *rows) {
  const bool a = depths == means2d + 2, b = conics == means2d + 4;
  if (a != b || (a && !aligned16(means2d))) return BDS_EINVAL;
  *rows = a ? 1 : 0;
  return BDS_OK;
}

static int project_view_fwd_impl(int64_t N, const float *means, const float *quats, const float *log_scales,
                                 const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                 float near_plane, float far_plane, float radius_clip, float *scales, float *opacities,
                                 int32_t *radii, float *means2d, float *depths, float *conics, const float *block_bounds,
                                 bds_stream_t stream) {
  BDS_REQUIRE(N >= 0 && W > 0 && H > 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(means && quats && log_scales && logits && viewmat && K && scales && opacities && radii && means2d && depths &&
              conics);
  int rows = 0;
  if (view_rows(means2d, depths, conics, &rows) != BDS_OK) return BDS_EINVAL;
  hipLaunchKernelGGL(project_view_fwd_kernel<false>, dim3((unsigned)cdiv(N, kProjBlock)), dim3(kProjBlock), 0, as_stream(stream), N,
                     means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip, scales,
                     opacities, radii, means2d, depths, conics, PrepReduceSlots{}, static_cast<int32_t *>(nullptr), rows, block_bounds);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_project_view_fwd(int64_t N, const float *means, const float *quats, const float *log_scales,
                                    const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                    float near_plane, float far_plane, float radius_clip, float *scales, float *opacities,
                                    int32_t *radii, float *means2d, float *depths, float *conics, bds_stream_t stream) {
  return project_view_fwd_impl(N, means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip, scales,
                               opacities, radii, means2d, depths, conics, nullptr, stream);
}

static int project_view_prepare_fwd_impl(int64_t N, const float *means, const float *quats, const float *log_scales,
                                         const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                         float near_plane, float far_plane, float radius_clip, float *scales, float *opacities,
                                         int32_t *radii, float *means2d, float *depths, float *conics, int32_t *tiles_per_gauss,
                                         void *prep_ws, size_t prep_ws_bytes, const float *block_bounds, bds_stream_t stream) {
  BDS_REQUIRE(N > 0 && W > 0 && H > 0);
  BDS_REQUIRE(means && quats && log_scales && logits && viewmat && K && scales && opacities && radii && means2d && depths &&
              conics);
  static_assert(kProjBlock == 256, "the tile stage reads the visible counts per 256 Gaussians");
  PrepReduceSlots rs;
  int rc = prep_reduce_slots(prep_ws, prep_ws_bytes, N, &rs);
  if (rc != BDS_OK) return rc;
  int rows = 0;
  if (view_rows(means2d, depths, conics, &rows) != BDS_OK) return BDS_EINVAL;
  hipLaunchKernelGGL(project_view_fwd_kernel<true>, dim3((unsigned)cdiv(N, kProjBlock)), dim3(kProjBlock), 0, as_stream(stream), N,
                     means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip, scales,
                     opacities, radii, means2d, depths, conics, rs, tiles_per_gauss, rows, block_bounds);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_project_view_prepare_fwd(int64_t N, const float *means, const float *quats, const float *log_scales,
                                            const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                            float near_plane, float far_plane, float radius_clip, float *scales, float *opacities,
                                            int32_t *radii, float *means2d, float *depths, float *conics, int32_t *tiles_per_gauss,
                                            void *prep_ws, size_t prep_ws_bytes, bds_stream_t stream) {
  return project_view_prepare_fwd_impl(N, means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip,
                                       scales, opacities, radii, means2d, depths, conics, tiles_per_gauss, prep_ws, prep_ws_bytes, nullptr,
                                       stream);
}

// ... with a bound per 256-row block (bds_gaussian_block_bounds over the SAME means / log_scales): blocks no centre of which can come
// out visible are not read; their rows get the outputs of a culled Gaussian (radius 0, zeros).  scales / opacities of such rows are
// NOT written (nothing reads them for a culled Gaussian).
extern "C" int bds_project_view_fwd_blocks(int64_t N, const float *means, const float *quats, const float *log_scales,
                                           const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                           float near_plane, float far_plane, float radius_clip, float *scales, float *opacities,
                                           int32_t *radii, float *means2d, float *depths, float *conics, const float *block_bounds,
                                           bds_stream_t stream) {
  BDS_REQUIRE(block_bounds);
  return project_view_fwd_impl(N, means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip, scales,
                               opacities, radii, means2d, depths, conics, block_bounds, stream);
}
extern "C" int bds_project_view_prepare_fwd_blocks(int64_t N, const float *means, const float *quats, const float *log_scales,
                                                   const float *logits, const float *viewmat, const float *K, int W, int H, float eps2d,
                                                   float near_plane, float far_plane, float radius_clip, float *scales,
                                                   float *opacities, int32_t *radii, float *means2d, float *depths, float *conics,
                                                   int32_t *tiles_per_gauss, void *prep_ws, size_t prep_ws_bytes,
                                                   const float *block_bounds, bds_stream_t stream) {
  BDS_REQUIRE(block_bounds);
  return project_view_prepare_fwd_impl(N, means, quats, log_scales, logits, viewmat, K, W, H, eps2d, near_plane, far_plane, radius_clip,
                                       scales, opacities, radii, means2d, depths, conics, tiles_per_gauss, prep_ws, prep_ws_bytes,
                                       block_bounds, stream);
}

namespace bds {
// bounds[b] = {lo.x, lo.y, lo.z, smax | hi.x, hi.y, hi.z, -} of rows [256 b, 256 (b + 1)): box of the centres, largest activated scale
__global__ __launch_bounds__(kProjBlock) void block_bounds_kernel(int64_t N, const float *__restrict__ means,
                                                                 const float *__restrict__ log_scales, float *__restrict__ bounds) {
  __shared__ float red[kProjBlock / kWave][8];
  const int64_t g = (int64_t)blockIdx.x * kProjBlock + threadIdx.x;
  float v[7] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};   // lo xyz | hi xyz | max log-scale
  if (g < N) {
    for (int k = 0; k < 3; k++) { const float m = means[g * 3 + k]; v[k] = m; v[3 + k] = m; }
    v[6] = fmaxf(fmaxf(log_scales[g * 3], log_scales[g * 3 + 1]), log_scales[g * 3 + 2]);
    // (a NaN centre or scale: the comparisons below would drop it -- make the box unbounded instead, the projection then decides)
    if (!(v[0] == v[0]) || !(v[1] == v[1]) || !(v[2] == v[2]) || !(v[6] == v[6])) {
      for (int k = 0; k < 3; k++) { v[k] = -3.0e38f; v[3 + k] = 3.0e38f; }
      v[6] = 80.f;
    }
  }
  for (int o = kWave / 2; o > 0; o >>= 1) {
    for (int k = 0; k < 3; k++) v[k] = fminf(v[k], __shfl_xor(v[k], o));
    for (int k = 3; k < 7; k++) v[k] = fmaxf(v[k], __shfl_xor(v[k], o));
  }
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0)
    for (int k = 0; k < 7; k++) red[wv][k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kProjBlock / kWave; w++) {
      for (int k = 0; k < 3; k++) v[k] = fminf(v[k], red[w][k]);
      for (int k = 3; k < 7; k++) v[k] = fmaxf(v[k], red[w][k]);
    }
    float *b = bounds + (int64_t)blockIdx.x * 8;
    b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = expf(v[6]);
    b[4] = v[3]; b[5] = v[4]; b[6] = v[5]; b[7] = 0.f;
  }
}
}  // namespace bds

// block_bounds [cdiv(N, 256), 8] for bds_project_view_*_fwd_blocks; recompute whenever means / log_scales changed (once per frame)
extern "C" int bds_gaussian_block_bounds(int64_t N, const float *means, const float *log_scales, float *block_bounds,
                                         bds_stream_t stream) {
  BDS_REQUIRE(N >= 0);
  if (N == 0) return BDS_OK;
  BDS_REQUIRE(means && log_scales && block_bounds);
  hipLaunchKernelGGL(block_bounds_kernel, dim3((unsigned)cdiv(N, kProjBlock)), dim3(kProjBlock), 0, as_stream(stream), N, means,
                     log_scales, block_bounds);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int project_view_bwd_list_impl(int64_t n_list, const uint64_t *n_dev, const int32_t *ids, const float *means, const float *quats,
                                      const float *scales, const float *opacities, const float *viewmat, const float *K, int W,
                                      int H, float eps2d, const float *v_records, float *v_means, float *v_quats,
                                      float *v_log_scales, float *v_logits, float *v_viewmat_slots, float *grad2d,
                                      float *absgrad2d, const int32_t *row_map, int accumulate, bds_stream_t stream) {
  BDS_REQUIRE(n_list >= 0 && W > 0 && H > 0);
  // (v_viewmat_slots is ADDED to: the caller zero-fills it -- a memset node of 4 KB between two kernels costs ~15 us of idle GPU)
  if (n_list == 0) return BDS_OK;
  BDS_REQUIRE(ids && means && quats && scales && opacities && viewmat && K && v_records && aligned16(v_records) && v_means &&
              v_quats && v_log_scales && v_logits);
  const dim3 grid((unsigned)cdiv(n_list, kProjBlock)), block(kProjBlock);
  const float4 *v4 = reinterpret_cast<const float4 *>(v_records);
  const GradLayout gl = grad_layout(v_means, v_quats, v_log_scales, v_logits);   // (the [N,16] row form is recognised by the addresses)
#define BDS_LIST(A, P)                                                                                                                \
  hipLaunchKernelGGL((project_view_bwd_list_kernel<A, P>), grid, block, 0, as_stream(stream), n_list, n_dev, ids, means, quats, scales, \
                     opacities, viewmat, K, W, H, eps2d, v4, v_means, v_quats, v_log_scales, v_logits, v_viewmat_slots, grad2d,        \
                     absgrad2d, row_map, static_cast<float *>(nullptr), gl)
  if (accumulate) { if (v_viewmat_slots) BDS_LIST(true, true); else BDS_LIST(true, false); }
  else            { if (v_viewmat_slots) BDS_LIST(false, true); else BDS_LIST(false, false); }
#undef BDS_LIST
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_project_view_bwd_list(int64_t n_list, const int32_t *ids, const float *means, const float *quats,
                                         const float *scales, const float *opacities, const float *viewmat, const float *K, int W,
                                         int H, float eps2d, const float *v_records, float *v_means, float *v_quats,
                                         float *v_log_scales, float *v_logits, float *v_viewmat_slots, float *grad2d,
                                         float *absgrad2d, const int32_t *row_map, int accumulate, bds_stream_t stream) {
  return project_view_bwd_list_impl(n_list, nullptr, ids, means, quats, scales, opacities, viewmat, K, W, H, eps2d, v_records, v_means,
                                    v_quats, v_log_scales, v_logits, v_viewmat_slots, grad2d, absgrad2d, row_map, accumulate, stream);
}

// The gsplat-shaped operator's backward over the visible entries (rendering.rasterization, C = 1): gradients of the ACTIVATED scales
// and opacities and of the post-activation colours, stored to the visible rows of dense, caller-zeroed arrays.
extern "C" int bds_project_bwd_list(int64_t n_list, const int32_t *ids, const float *means, const float *quats, const float *scales,
                                    const float *opacities, const float *viewmat, const float *K, int W, int H, float eps2d,
                                    const float *v_records, float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                                    float *v_colors, float *v_viewmat_slots, float *grad2d, float *absgrad2d, bds_stream_t stream) {
  BDS_REQUIRE(n_list >= 0 && W > 0 && H > 0);
  if (n_list == 0) return BDS_OK;
  BDS_REQUIRE(ids && means && quats && scales && opacities && viewmat && K && v_records && aligned16(v_records) && v_means &&
              v_quats && v_scales && v_opacities);
  const dim3 grid((unsigned)cdiv(n_list, kProjBlock)), block(kProjBlock);
  const float4 *v4 = reinterpret_cast<const float4 *>(v_records);
  if (v_viewmat_slots)
    hipLaunchKernelGGL((project_view_bwd_list_kernel<false, true, false>), grid, block, 0, as_stream(stream), n_list, nullptr, ids, means,
                       quats, scales, opacities, viewmat, K, W, H, eps2d, v4, v_means, v_quats, v_scales, v_opacities, v_viewmat_slots,
                       grad2d, absgrad2d, nullptr, v_colors);
  else
    hipLaunchKernelGGL((project_view_bwd_list_kernel<false, false, false>), grid, block, 0, as_stream(stream), n_list, nullptr, ids, means,
                       quats, scales, opacities, viewmat, K, W, H, eps2d, v4, v_means, v_quats, v_scales, v_opacities, v_viewmat_slots,
                       grad2d, absgrad2d, nullptr, v_colors);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_project_view_bwd_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, const float *means,
                                             const float *quats, const float *scales, const float *opacities, const float *viewmat,
                                             const float *K, int W, int H, float eps2d, const float *v_records, float *v_means,
                                             float *v_quats, float *v_log_scales, float *v_logits, float *v_viewmat_slots, float *grad2d,
                                             float *absgrad2d, const int32_t *row_map, int accumulate, bds_stream_t stream) {
  BDS_REQUIRE(n_dev);
  return project_view_bwd_list_impl(n_capacity, n_dev, ids, means, quats, scales, opacities, viewmat, K, W, H, eps2d, v_records,
                                    v_means, v_quats, v_log_scales, v_logits, v_viewmat_slots, grad2d, absgrad2d, row_map, accumulate,
                                    stream);
}

// Bit t of *flags_dev is set when tensors[t] (counts[t] floats) holds a NaN or an Inf (vanilla.py:407-412); the word is cleared first.
// flags_pinned (optional, page-locked): receives a copy behind the launch -- the host reads it after its next wait on the stream.
extern "C" int bds_nonfinite_flags_kinds(int n_tensors, const float *const *tensors, const int64_t *counts, const int *kinds,
                                         uint32_t *flags_dev, uint32_t *flags_pinned, bds_stream_t stream);
extern "C" int bds_nonfinite_flags(int n_tensors, const float *const *tensors, const int64_t *counts, uint32_t *flags_dev,
                                   uint32_t *flags_pinned, bds_stream_t stream) {
  return bds_nonfinite_flags_kinds(n_tensors, tensors, counts, nullptr, flags_dev, flags_pinned, stream);
}
// kinds (optional, [n_tensors]): 0 plain | 1 argument of exp | 2 quaternion rows [n/4, 4] (16-byte aligned) | 3 argument of sigmoid:
// the bit then says "the ACTIVATED tensor would hold a NaN / Inf" (vanilla.py:393-395 activations, :407-412 check)
extern "C" int bds_nonfinite_flags_kinds(int n_tensors, const float *const *tensors, const int64_t *counts, const int *kinds,
                                         uint32_t *flags_dev, uint32_t *flags_pinned, bds_stream_t stream) {
  BDS_REQUIRE(n_tensors >= 0 && n_tensors <= 8 && flags_dev && (n_tensors == 0 || (tensors && counts)));
  FiniteArgs A;
  A.count = n_tensors;
  int64_t total = 0;
  for (int t = 0; t < n_tensors; t++) {
    A.kind[t] = kinds ? kinds[t] : 0;
    BDS_REQUIRE(A.kind[t] >= 0 && A.kind[t] <= 3);
    BDS_REQUIRE(A.kind[t] != 2 || (counts[t] % 4 == 0 && (reinterpret_cast<uintptr_t>(tensors[t]) & 15u) == 0));
    BDS_REQUIRE(counts[t] >= 0 && (counts[t] == 0 || tensors[t]) && (reinterpret_cast<uintptr_t>(tensors[t]) & 3u) == 0);
    A.p[t] = reinterpret_cast<const uint32_t *>(tensors[t]);
    A.n[t] = counts[t];
    total += counts[t];
  }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(flags_dev, 0, sizeof(uint32_t), st) != hipSuccess) return BDS_ELAUNCH;
  if (total > 0) {
    const int64_t want = cdiv(cdiv(total, 16), kFiniteBlock);     // four 16-byte pieces per thread and step
    const unsigned grid = (unsigned)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(nonfinite_flags_kernel, dim3(grid), dim3(kFiniteBlock), 0, st, A, flags_dev);
    BDS_LAUNCH_CHECK();
  }
  if (flags_pinned && hipMemcpyAsync(flags_pinned, flags_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return BDS_ELAUNCH;
  return BDS_OK;
}
