// Fused head of the neural bilateral variants: per-pixel sliced features -> Linear(F,64) tanh Linear(64,64) tanh Linear(64,12), no
// biases (/root/reference/project/models/modules.py:621-627, 700-706: `affine_network`) -> the 3x4 map applied to the pixel with the
// trainer's residual (models/trainers/scene_graph.py:99-106: rgb' = A[:, :3] rgb + A[:, 3] + rgb), forward and backward.
// SURVEY.md 8f rank 3.
//
// The reference runs it as three cuBLAS GEMMs over [H*W, 64] activations plus tanh / matmul passes: ~2 KB of activation traffic per
// pixel each way.  Here a wave owns 32 pixels at a time and the whole chain stays in its registers, on the FP32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, an fmaf chain per output -- no reduced-precision shortcut, parity with the f32 reference):
//
//   * activations are kept TRANSPOSED, [neurons x 32 pixels], as MFMA D tiles (lane l: column = pixel l & 31, register r = row
//     d_row(r, l >> 5)).  The B operand of the instruction is laid out the same way (lane l: B[k = l >> 5][j = l & 31]), and the
//     order of the k summation is free, so the D registers of one layer ARE the B operands of the next: step s consumes register
//     s of the tile, the weights (A operand, read from an LDS image of the [out, in] matrices) follow the same k order.  Nothing
//     moves between lanes from the feature load to the affine entries.
//   * the 12 affine entries land as rows 0-3 / 8-11 (half-0 lane of a pixel) and 4-7 (half-1 lane): output channels 0 and 2 are
//     finished by one lane, channel 1 by the other; the input gradient needs one exchange with lane ^ 32.
//   * backward recomputes the two hidden tiles (cheaper than 512 B of stored activations per pixel), runs the transposed chain
//     (A operand = W^T, same LDS image read the other way) and takes the three weight gradients as products over the PIXELS, which
//     needs the tiles with lane = neuron: they go through per-wave LDS tiles (row stride 36 floats) once per layer.  The weight
//     gradients accumulate in 128 registers per lane over the whole persistent loop; every wave then writes its partial and a
//     second kernel sums the partials (deterministic, no atomics).
//
// Register dataflow modelled lane by lane and checked against autograd on the CPU: oracle/mfma_dataflow_model.py,
// tests/test_mlp_head_dataflow.py.  Bound: the FP32 MFMA rate (157 TFLOP/s): 2 * (64 F + 4096 + 2048) flop per pixel forward (the
// 12-row layer is padded to 32), three times that backward (recompute + data path + weight gradients).
#include "bds_common.h"

namespace bds {

typedef float acc16 __attribute__((ext_vector_type(16)));

constexpr int kMhBlock = 256;  // four waves, one per SIMD
constexpr int kMhWaves = kMhBlock / kWave;
constexpr int kMhTile = 32;    // pixels per wave step
constexpr int kMhHid = 64;
constexpr int kMhAff = 12;
constexpr int kMhWStride = 68;  // row stride of the W2 / W3 images in LDS: 16-byte rows, 16 consecutive rows hit 16 distinct bank groups
constexpr int kMhTStride = 36;  // row stride of the transposition tiles
constexpr int kMhBufBig = kMhHid * kMhTStride, kMhBufSmall = 32 * kMhTStride;

__device__ __forceinline__ constexpr int d_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ acc16 zero16() {
  acc16 z;
#pragma unroll
  for (int r = 0; r < 16; r++) z[r] = 0.f;
  return z;
}

__device__ __forceinline__ acc16 mfma(float a, float b, acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// tanh(x) = 1 - 2 / (exp(2x) + 1): saturates correctly at both ends (exp -> inf: 1, exp -> 0: -1), |error| ~ 1e-7
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }

// LDS image of the three weight matrices ([out, in] as torch.nn.Linear stores them); W3 padded to 32 zero rows
template <int F>
struct MhImage {
  static constexpr int S1 = F + 4;
  static constexpr int off1 = 0, off2 = off1 + kMhHid * S1, off3 = off2 + kMhHid * kMhWStride, floats = off3 + 32 * kMhWStride;
};

template <int F>
__device__ __forceinline__ void mh_load_weights(float *__restrict__ lds, const float *__restrict__ w1, const float *__restrict__ w2,
                                                const float *__restrict__ w3) {
  using Im = MhImage<F>;
  for (int i = threadIdx.x; i < kMhHid * F; i += kMhBlock) lds[Im::off1 + (i / F) * Im::S1 + (i % F)] = w1[i];
  for (int i = threadIdx.x; i < kMhHid * kMhHid; i += kMhBlock) lds[Im::off2 + (i >> 6) * kMhWStride + (i & 63)] = w2[i];
  for (int i = threadIdx.x; i < 32 * kMhHid; i += kMhBlock) lds[Im::off3 + (i >> 6) * kMhWStride + (i & 63)] = i < kMhAff * kMhHid ? w3[i] : 0.f;
}

// B operands of the first layer: lane (pixel, half) holds features half * F/2 + s, s < F/2 -- consecutive floats of its row
template <int F>
__device__ __forceinline__ void mh_load_features(const float *__restrict__ feats, int64_t px, int half, float (&x)[F / 2]) {
  const float4 *src = reinterpret_cast<const float4 *>(feats + px * F + (F / 2) * half);
#pragma unroll
  for (int q = 0; q < F / 8; q++) {
    const float4 v = src[q];
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
}

// features -> hidden 1 -> hidden 2 -> the affine tile, all as D tiles
template <int F>
__device__ __forceinline__ void mh_forward_tile(const float *__restrict__ lds, int col, int half, const float (&x)[F / 2], acc16 (&h1)[2],
                                                acc16 (&h2)[2], acc16 &aff) {
  using Im = MhImage<F>;
  constexpr int KS1 = F / 2;
#pragma unroll
  for (int o = 0; o < 2; o++) {
    acc16 acc = zero16();
    const float *wr = lds + Im::off1 + (32 * o + col) * Im::S1 + KS1 * half;
#pragma unroll
    for (int s = 0; s < KS1; s++) acc = mfma(wr[s], x[s], acc);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = tanh_fast(acc[r]);
    h1[o] = acc;
  }
#pragma unroll
  for (int o = 0; o < 2; o++) {
    acc16 acc = zero16();
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const float *wr = lds + Im::off2 + (32 * o + col) * kMhWStride + 32 * b + 4 * half;
#pragma unroll
      for (int s = 0; s < 16; s++) acc = mfma(wr[(s & 3) + 8 * (s >> 2)], h1[b][s], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = tanh_fast(acc[r]);
    h2[o] = acc;
  }
  acc16 acc = zero16();
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const float *wr = lds + Im::off3 + col * kMhWStride + 32 * b + 4 * half;
#pragma unroll
    for (int s = 0; s < 16; s++) acc = mfma(wr[(s & 3) + 8 * (s >> 2)], h2[b][s], acc);
  }
  aff = acc;
}

template <int F>
__global__ __launch_bounds__(kMhBlock) void mlp_head_fwd_kernel(int64_t P, const float *__restrict__ feats, const float *__restrict__ rgb,
                                                               const float *__restrict__ w1, const float *__restrict__ w2,
                                                               const float *__restrict__ w3, int residual, float *__restrict__ out,
                                                               float *__restrict__ affine) {
  extern __shared__ float lds[];
  mh_load_weights<F>(lds, w1, w2, w3);
  __syncthreads();
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, col = lane & 31, half = lane >> 5;
  const int64_t n_tiles = (P + kMhTile - 1) / kMhTile;
  for (int64_t tile = (int64_t)blockIdx.x * kMhWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kMhWaves) {
    const int64_t px = tile * kMhTile + col;
    const bool on = px < P;
    const int64_t pc = on ? px : P - 1;
    float x[F / 2];
    mh_load_features<F>(feats, pc, half, x);
    acc16 h1[2], h2[2], aff;
    mh_forward_tile<F>(lds, col, half, x, h1, h2, aff);
    if (!on) continue;
    if (affine) {  // rows 0-3 | 8-11 from the half-0 lane, 4-7 from the half-1 lane
      float4 *dst = reinterpret_cast<float4 *>(affine + px * kMhAff);
      dst[half] = make_float4(aff[0], aff[1], aff[2], aff[3]);
      if (half == 0) dst[2] = make_float4(aff[4], aff[5], aff[6], aff[7]);
    }
    if (out) {
      const float c0 = rgb[pc * 3], c1 = rgb[pc * 3 + 1], c2 = rgb[pc * 3 + 2];
      float lo = aff[0] * c0 + aff[1] * c1 + aff[2] * c2 + aff[3];
      float hi = aff[4] * c0 + aff[5] * c1 + aff[6] * c2 + aff[7];
      if (residual) { lo += half ? c1 : c0; hi += c2; }
      out[px * 3 + half] = lo;                 // channel 0 (half 0) / 1 (half 1)
      if (half == 0) out[px * 3 + 2] = hi;
    }
  }
}

// LDS ordering inside one wave: its LDS operations complete in order; the fence keeps the compiler from moving them and waits for
// the writes before other lanes' reads
__device__ __forceinline__ void mh_wave_fence() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// D tiles -> T[row][pixel]
template <int NB>
__device__ __forceinline__ void mh_store_tiles(float *__restrict__ T, int col, int half, const acc16 (&t)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) T[(32 * b + d_row(r, half)) * kMhTStride + col] = t[b][r];
}

// G[ob][vb] += U V^T over the 32 pixels: step s takes pixel 16 * half + s on both operands (16 consecutive floats of a lane's row)
template <int NU, int NV>
__device__ __forceinline__ void mh_outer(const float *__restrict__ TU, const float *__restrict__ TV, int col, int half,
                                         acc16 (&g)[NU * NV]) {
#pragma unroll
  for (int ob = 0; ob < NU; ob++) {
    float a[16];
    const float4 *ua = reinterpret_cast<const float4 *>(TU + (32 * ob + col) * kMhTStride + 16 * half);
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 v = ua[q]; a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
#pragma unroll
    for (int vb = 0; vb < NV; vb++) {
      float b[16];
      const float4 *vbp = reinterpret_cast<const float4 *>(TV + (32 * vb + col) * kMhTStride + 16 * half);
#pragma unroll
      for (int q = 0; q < 4; q++) { const float4 v = vbp[q]; b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
      acc16 acc = g[ob * NV + vb];
#pragma unroll
      for (int s = 0; s < 16; s++) acc = mfma(a[s], b[s], acc);
      g[ob * NV + vb] = acc;
    }
  }
}

template <int F>
__global__ __launch_bounds__(kMhBlock) void mlp_head_bwd_kernel(int64_t P, const float *__restrict__ feats, const float *__restrict__ rgb,
                                                               const float *__restrict__ w1, const float *__restrict__ w2,
                                                               const float *__restrict__ w3, int residual,
                                                               const float *__restrict__ v_out, const float *__restrict__ v_affine,
                                                               float *__restrict__ v_feats, float *__restrict__ v_rgb,
                                                               float *__restrict__ partials) {
  using Im = MhImage<F>;
  constexpr int KS1 = F / 2;
  extern __shared__ float lds[];
  mh_load_weights<F>(lds, w1, w2, w3);
  __syncthreads();
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, col = lane & 31, half = lane >> 5;
  float *bufA = lds + Im::floats + wave * (2 * kMhBufBig + kMhBufSmall);
  float *bufB = bufA + kMhBufBig;
  float *bufS = bufB + kMhBufBig;
  acc16 g1[2], g2[4], g3[2];
#pragma unroll
  for (int i = 0; i < 2; i++) { g1[i] = zero16(); g3[i] = zero16(); }
#pragma unroll
  for (int i = 0; i < 4; i++) g2[i] = zero16();

  const int64_t n_tiles = (P + kMhTile - 1) / kMhTile;
  for (int64_t tile = (int64_t)blockIdx.x * kMhWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kMhWaves) {
    const int64_t px = tile * kMhTile + col;
    const bool on = px < P;
    const int64_t pc = on ? px : P - 1;
    float x[KS1];
    mh_load_features<F>(feats, pc, half, x);
    acc16 h1[2], h2[2], aff;
    mh_forward_tile<F>(lds, col, half, x, h1, h2, aff);

    // gradient of the 12 entries in the affine tile's layout (registers 0-7; rows 12-31 are zero); lanes past the end carry zeros,
    // which silences every weight-gradient contribution of theirs
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = 0.f;
    if (v_affine && on) {
      const float4 *src = reinterpret_cast<const float4 *>(v_affine + px * kMhAff);
      const float4 lo = src[half];
      t[0] = lo.x; t[1] = lo.y; t[2] = lo.z; t[3] = lo.w;
      if (half == 0) { const float4 hi = src[2]; t[4] = hi.x; t[5] = hi.y; t[6] = hi.z; t[7] = hi.w; }
    }
    if (v_out) {
      const float c0 = rgb[pc * 3], c1 = rgb[pc * 3 + 1], c2 = rgb[pc * 3 + 2];
      const float g0 = on ? v_out[pc * 3] : 0.f, gg1 = on ? v_out[pc * 3 + 1] : 0.f, gg2 = on ? v_out[pc * 3 + 2] : 0.f;
      const float r_lo = half ? gg1 : g0, r_hi = half ? 0.f : gg2;
      t[0] += r_lo * c0; t[1] += r_lo * c1; t[2] += r_lo * c2; t[3] += r_lo;
      t[4] += r_hi * c0; t[5] += r_hi * c1; t[6] += r_hi * c2; t[7] += r_hi;
      if (v_rgb) {  // sum_r A[r][c] v_out[r]: each half holds part of the rows
        float p0 = aff[0] * r_lo + aff[4] * r_hi, p1 = aff[1] * r_lo + aff[5] * r_hi, p2 = aff[2] * r_lo + aff[6] * r_hi;
        p0 += __shfl_xor(p0, 32); p1 += __shfl_xor(p1, 32); p2 += __shfl_xor(p2, 32);
        if (on && half == 0) {
          v_rgb[px * 3] = p0 + (residual ? g0 : 0.f);
          v_rgb[px * 3 + 1] = p1 + (residual ? gg1 : 0.f);
          v_rgb[px * 3 + 2] = p2 + (residual ? gg2 : 0.f);
        }
      }
    }

    // ---- layer 3: weight gradient (d_aff x h2 over the pixels), then d_h2 = W3^T d_aff and through the tanh
#pragma unroll
    for (int r = 0; r < 8; r++) bufS[d_row(r, half) * kMhTStride + col] = t[r];
    mh_store_tiles<2>(bufA, col, half, h2);
    mh_wave_fence();
    mh_outer<1, 2>(bufS, bufA, col, half, g3);
    mh_wave_fence();
#pragma unroll
    for (int o = 0; o < 2; o++) {
      acc16 acc = zero16();
#pragma unroll
      for (int s = 0; s < 8; s++) acc = mfma(lds[Im::off3 + d_row(s, half) * kMhWStride + 32 * o + col], t[s], acc);
#pragma unroll
      for (int r = 0; r < 16; r++) h2[o][r] = acc[r] * (1.f - h2[o][r] * h2[o][r]);  // h2 now holds d_z2
    }
    // ---- layer 2
    mh_store_tiles<2>(bufA, col, half, h2);
    mh_store_tiles<2>(bufB, col, half, h1);
    mh_wave_fence();
    mh_outer<2, 2>(bufA, bufB, col, half, g2);
    mh_wave_fence();
    {
      acc16 d[2];
#pragma unroll
      for (int o = 0; o < 2; o++) {
        acc16 acc = zero16();
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int s = 0; s < 16; s++)
            acc = mfma(lds[Im::off2 + (32 * b + d_row(s, half)) * kMhWStride + 32 * o + col], h2[b][s], acc);
        d[o] = acc;
      }
#pragma unroll
      for (int o = 0; o < 2; o++)
#pragma unroll
        for (int r = 0; r < 16; r++) h1[o][r] = d[o][r] * (1.f - h1[o][r] * h1[o][r]);  // h1 now holds d_z1
    }
    // ---- layer 1
    mh_store_tiles<2>(bufA, col, half, h1);
#pragma unroll
    for (int s = 0; s < KS1; s++) bufS[(s + KS1 * half) * kMhTStride + col] = x[s];
    mh_wave_fence();
    mh_outer<2, 1>(bufA, bufS, col, half, g1);
    mh_wave_fence();
    if (v_feats) {
      acc16 acc = zero16();
      const bool live = col < F;
      const int cc = live ? col : 0;
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int s = 0; s < 16; s++) {
          const float w = lds[Im::off1 + (32 * b + d_row(s, half)) * Im::S1 + cc];
          acc = mfma(live ? w : 0.f, h1[b][s], acc);
        }
      if (on) {  // rows = features d_row(r, half): four consecutive features per register group
        float *dst = v_feats + px * F;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (8 * q + 4 * half < F)
            *reinterpret_cast<float4 *>(dst + 8 * q + 4 * half) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      }
    }
  }

  // this wave's weight-gradient partial, in the matrices' own [out, in] layout: register r of tile (ob, vb) is
  // G[32 ob + d_row(r, half)][32 vb + col]
  float *part = partials + ((int64_t)blockIdx.x * kMhWaves + wave) * (kMhHid * F + kMhHid * kMhHid + kMhAff * kMhHid);
#pragma unroll
  for (int ob = 0; ob < 2; ob++)
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (col < F) part[(32 * ob + d_row(r, half)) * F + col] = g1[ob][r];
  part += kMhHid * F;
#pragma unroll
  for (int ob = 0; ob < 2; ob++)
#pragma unroll
    for (int vb = 0; vb < 2; vb++)
#pragma unroll
      for (int r = 0; r < 16; r++) part[(32 * ob + d_row(r, half)) * kMhHid + 32 * vb + col] = g2[ob * 2 + vb][r];
  part += kMhHid * kMhHid;
#pragma unroll
  for (int vb = 0; vb < 2; vb++)
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (d_row(r, half) < kMhAff) part[d_row(r, half) * kMhHid + 32 * vb + col] = g3[vb][r];
}

// sums the waves' partials in a fixed order; the three matrices are contiguous in a partial: [64, F] | [64, 64] | [12, 64].
// Workgroup = 32 entries x 8 partial-lanes: coalesced 128-byte rows, 8 independent loads in flight per thread
__global__ __launch_bounds__(256) void mlp_head_reduce_kernel(int n_parts, int F, const float *__restrict__ partials,
                                                             float *__restrict__ v_w1, float *__restrict__ v_w2,
                                                             float *__restrict__ v_w3, int accumulate) {
  __shared__ float sred[8][33];
  const int n1 = kMhHid * F, n2 = kMhHid * kMhHid, n3 = kMhAff * kMhHid, len = n1 + n2 + n3;
  const int ex = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + ex;
  float s = 0.f;
  if (e < len) {
    int p = py;
    for (; p + 56 < n_parts; p += 64) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = partials[(int64_t)(p + 8 * u) * len + e];
#pragma unroll
      for (int u = 0; u < 8; u++) s += t[u];
    }
    for (; p < n_parts; p += 8) s += partials[(int64_t)p * len + e];
  }
  sred[py][ex] = s;
  __syncthreads();
  if (py != 0 || e >= len) return;
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) t += sred[q][ex];
  float *base = e < n1 ? v_w1 : (e < n1 + n2 ? v_w2 : v_w3);
  if (!base) return;
  float *dst = base + (e < n1 ? e : (e < n1 + n2 ? e - n1 : e - n1 - n2));
  *dst = accumulate ? *dst + t : t;
}

inline int mh_grid_fwd(int64_t P) {
  const int64_t groups = cdiv(cdiv(P, (int64_t)kMhTile), (int64_t)kMhWaves);
  return (int)(groups < 768 ? (groups > 0 ? groups : 1) : 768);   // three resident workgroups per CU
}
inline int mh_grid_bwd(int64_t P) {
  const int64_t groups = cdiv(cdiv(P, (int64_t)kMhTile), (int64_t)kMhWaves);
  return (int)(groups < 256 ? (groups > 0 ? groups : 1) : 256);   // one per CU: 125 KB of LDS each
}
inline bool mh_supported(int F) { return F == 8 || F == 16 || F == 24 || F == 32; }

template <int F>
int mh_launch_fwd(int64_t P, const float *feats, const float *rgb, const float *w1, const float *w2, const float *w3, int residual,
                  float *out, float *affine, hipStream_t st) {
  const size_t lds_bytes = MhImage<F>::floats * sizeof(float);
  hipLaunchKernelGGL((mlp_head_fwd_kernel<F>), dim3(mh_grid_fwd(P)), dim3(kMhBlock), lds_bytes, st, P, feats, rgb, w1, w2, w3, residual, out,
                     affine);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

template <int F>
int mh_launch_bwd(int64_t P, const float *feats, const float *rgb, const float *w1, const float *w2, const float *w3, int residual,
                  const float *v_out, const float *v_affine, float *v_feats, float *v_rgb, float *partials, hipStream_t st) {
  const size_t lds_bytes = (MhImage<F>::floats + kMhWaves * (2 * kMhBufBig + kMhBufSmall)) * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_head_bwd_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return BDS_ELAUNCH;
  hipLaunchKernelGGL((mlp_head_bwd_kernel<F>), dim3(mh_grid_bwd(P)), dim3(kMhBlock), lds_bytes, st, P, feats, rgb, w1, w2, w3, residual,
                     v_out, v_affine, v_feats, v_rgb, partials);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

}  // namespace bds

using namespace bds;

extern "C" size_t bds_mlp_head_bwd_temp_bytes(int64_t P, int F) {
  if (P <= 0 || !mh_supported(F)) return 0;
  return (size_t)mh_grid_bwd(P) * kMhWaves * (size_t)(kMhHid * F + kMhHid * kMhHid + kMhAff * kMhHid) * sizeof(float);
}

extern "C" int bds_mlp_head_fwd(int64_t P, int F, int hidden, const float *feats, const float *rgb, const float *w1, const float *w2,
                                const float *w3, int residual, float *out, float *affine, bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && hidden == kMhHid && mh_supported(F));
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(feats && w1 && w2 && w3 && (out || affine) && (!out || rgb));
  BDS_REQUIRE(aligned16(feats) && (!affine || aligned16(affine)));
  hipStream_t st = as_stream(stream);
  switch (F) {
    case 8: return mh_launch_fwd<8>(P, feats, rgb, w1, w2, w3, residual, out, affine, st);
    case 16: return mh_launch_fwd<16>(P, feats, rgb, w1, w2, w3, residual, out, affine, st);
    case 24: return mh_launch_fwd<24>(P, feats, rgb, w1, w2, w3, residual, out, affine, st);
    default: return mh_launch_fwd<32>(P, feats, rgb, w1, w2, w3, residual, out, affine, st);
  }
}

extern "C" int bds_mlp_head_bwd(int64_t P, int F, int hidden, const float *feats, const float *rgb, const float *w1, const float *w2,
                                const float *w3, int residual, const float *v_out, const float *v_affine, float *v_feats,
                                float *v_rgb, float *v_w1, float *v_w2, float *v_w3, int accumulate_w, void *temp, size_t temp_bytes,
                                bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && hidden == kMhHid && mh_supported(F));
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(feats && w1 && w2 && w3 && (v_out || v_affine) && (!v_out || rgb) && (!v_rgb || v_out));
  BDS_REQUIRE(temp && temp_bytes >= bds_mlp_head_bwd_temp_bytes(P, F));
  BDS_REQUIRE(aligned16(feats) && (!v_feats || aligned16(v_feats)) && (!v_affine || aligned16(v_affine)));
  hipStream_t st = as_stream(stream);
  float *partials = static_cast<float *>(temp);
  int rc;
  switch (F) {
    case 8: rc = mh_launch_bwd<8>(P, feats, rgb, w1, w2, w3, residual, v_out, v_affine, v_feats, v_rgb, partials, st); break;
    case 16: rc = mh_launch_bwd<16>(P, feats, rgb, w1, w2, w3, residual, v_out, v_affine, v_feats, v_rgb, partials, st); break;
    case 24: rc = mh_launch_bwd<24>(P, feats, rgb, w1, w2, w3, residual, v_out, v_affine, v_feats, v_rgb, partials, st); break;
    default: rc = mh_launch_bwd<32>(P, feats, rgb, w1, w2, w3, residual, v_out, v_affine, v_feats, v_rgb, partials, st); break;
  }
  if (rc != BDS_OK) return rc;
  if (v_w1 || v_w2 || v_w3) {
    const int len = kMhHid * F + kMhHid * kMhHid + kMhAff * kMhHid;
    hipLaunchKernelGGL(mlp_head_reduce_kernel, dim3((unsigned)cdiv(len, 32)), dim3(256), 0, st, mh_grid_bwd(P) * kMhWaves, F, partials,
                       v_w1, v_w2, v_w3, accumulate_w);
    BDS_LAUNCH_CHECK();
  }
  return BDS_OK;
}
