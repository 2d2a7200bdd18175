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
#include "bilagrid_math.h"

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

// layer 1 from feature registers (k order: feature half * F/2 + s)
template <int F>
__device__ __forceinline__ void mh_layer1_feats(const float *__restrict__ lds, int col, int half, const float (&x)[F / 2], acc16 (&h1)[2]) {
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
}

// layer 1 from a D tile of features (rows = channels d_row(r, half)): the chained k order, only the register groups that hold
// real channels (F is a multiple of 8 = one group of both halves)
template <int F>
__device__ __forceinline__ void mh_layer1_tile(const float *__restrict__ lds, int col, int half, const acc16 &xt, acc16 (&h1)[2]) {
  using Im = MhImage<F>;
#pragma unroll
  for (int o = 0; o < 2; o++) {
    acc16 acc = zero16();
    const float *wr = lds + Im::off1 + (32 * o + col) * Im::S1 + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (8 * (r >> 2) < F) acc = mfma(wr[(r & 3) + 8 * (r >> 2)], xt[r], acc);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = tanh_fast(acc[r]);
    h1[o] = acc;
  }
}

// hidden 1 -> hidden 2 -> the affine tile
template <int F>
__device__ __forceinline__ void mh_layers23(const float *__restrict__ lds, int col, int half, const acc16 (&h1)[2], acc16 (&h2)[2],
                                            acc16 &aff) {
  using Im = MhImage<F>;
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

// features -> hidden 1 -> hidden 2 -> the affine tile, all as D tiles
template <int F>
__device__ __forceinline__ void mh_forward_tile(const float *__restrict__ lds, int col, int half, const float (&x)[F / 2], acc16 (&h1)[2],
                                                acc16 (&h2)[2], acc16 &aff) {
  mh_layer1_feats<F>(lds, col, half, x, h1);
  mh_layers23<F>(lds, col, half, h1, h2, aff);
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

// ==================================================================================================================================
// The slice folded in: the whole `transform` of the neural variants for one image (features never reach HBM).
//   models/modules.py:643-670 / 728-790 (feature slice at the pixel grid, one grid per level) + the head + scene_graph.py:99-106.
// With tiles cut at the grids' cell boundaries the 32 pixels of a tile share their (x0, y0) cell on every level and touch
// K = sum_l 4 gl_l grid "slots" (z, y corner, x corner):
//   features^T [F x 32] = G^T [F x K] . Wt [K x 32]              one more chained D tile in front of layer 1
//   d(slots) [K x F]   += Wt [K x px] . dF [px x F]              one more product over the pixels (the LDS transposes of the weight gradients)
//   d(gray) [px]        = sum_ch dF^T[ch][px] (G^T . dWt/dz)[ch][px]   row-wise dot of two D tiles + one exchange with lane ^ 32
// Slot q = off_l + (z * 2 + yb) * 2 + xb is consumed as k = 2 s + half: the lane half IS the x corner.  A pixel ON the last grid
// column / row (x0 = gx - 1, f = 0) is moved to the cell before it with f = 1 -- the same sample, the same scatter -- so that it does
// not form a cell of its own.  Dataflow modelled lane by lane: oracle/mfma_dataflow_model.py (fused_forward / fused_backward).
//
// Work: jobs = (row band, x segment, chunk of rows); a wave takes whole jobs, loads the region's 2 x 2 x gl nodes once (its A
// operands, [step][lane] in LDS), accumulates the region's slot gradient in registers and flushes it with one atomic per entry.
// Segment / band boundaries are found ON THE DEVICE with the same coordinate functions the pixels use (a host copy could disagree
// in the last bit and put a pixel in the wrong cell).
constexpr int kNiMaxSeg = 64, kNiMaxTile = 192;

struct NiLevel {
  const float *grid;
  float *v_grid;
  int gx, gy, gl;
};
struct NiParams {
  int H, W, rows_per_job;
  float lin_x, lin_y;
  NiLevel lv[2];
};
struct NiTables {
  int nseg, nband, cpb, cnt;
  int seg_start[kNiMaxSeg + 1], band_start[kNiMaxSeg + 1], seg_tile[kNiMaxSeg + 1], tmp[kNiMaxSeg + 1];
  short tile_x[kNiMaxTile], tile_n[kNiMaxTile];
};

template <int NL>
__device__ __forceinline__ bool ni_new_cell(const NiParams &p, int i, int n, float lin, bool x_axis) {
  if (i == 0) return true;
  bool nw = false;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const int g = x_axis ? p.lv[l].gx : p.lv[l].gy;
    int a, b;
    float f;
    axis_cell(linspace01_s(i, n, lin), g, a, f);
    axis_cell(linspace01_s(i - 1, n, lin), g, b, f);
    nw = nw || (a != b);
  }
  return nw;
}

// boundaries along one axis: the indices where some level's cell changes, ascending, closed by n
template <int NL>
__device__ void ni_boundaries(const NiParams &p, NiTables &T, int n, float lin, bool x_axis, int *out, int &count) {
  if (threadIdx.x == 0) T.cnt = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kMhBlock)
    if (ni_new_cell<NL>(p, i, n, lin, x_axis)) {
      const int k = atomicAdd(&T.cnt, 1);
      if (k < kNiMaxSeg) T.tmp[k] = i;
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = T.cnt < kNiMaxSeg ? T.cnt : kNiMaxSeg;
    for (int a = 1; a < c; a++) {   // insertion sort of <= 64 entries
      const int v = T.tmp[a];
      int b = a - 1;
      for (; b >= 0 && T.tmp[b] > v; b--) T.tmp[b + 1] = T.tmp[b];
      T.tmp[b + 1] = v;
    }
    for (int a = 0; a < c; a++) out[a] = T.tmp[a];
    out[c] = n;
    count = c;
  }
  __syncthreads();
}

template <int NL>
__device__ void ni_build_tables(const NiParams &p, NiTables &T) {
  ni_boundaries<NL>(p, T, p.W, p.lin_x, true, T.seg_start, T.nseg);
  ni_boundaries<NL>(p, T, p.H, p.lin_y, false, T.band_start, T.nband);
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < T.nseg; i++) {
      T.seg_tile[i] = t;
      for (int x = T.seg_start[i]; x < T.seg_start[i + 1] && t < kNiMaxTile; x += kMhTile, t++) {
        T.tile_x[t] = (short)x;
        const int rest = T.seg_start[i + 1] - x;
        T.tile_n[t] = (short)(rest < kMhTile ? rest : kMhTile);
      }
    }
    T.seg_tile[T.nseg] = t;
    int tallest = 1;
    for (int b = 0; b < T.nband; b++) {
      const int h = T.band_start[b + 1] - T.band_start[b];
      tallest = h > tallest ? h : tallest;
    }
    T.cpb = (tallest + p.rows_per_job - 1) / p.rows_per_job;
  }
  __syncthreads();
}

// per-pixel state of the slice on one level
struct NiPix {
  float fx, fz, isc;   // isc = (gl - 1) inside the guidance range, 0 on / outside its border (grid_sample's border clip)
  int z0, z1;
};

template <int NCH0, int GL0, int NCH1, int GL1>
struct NiShape {
  static constexpr int NL = GL1 > 0 ? 2 : 1;
  static constexpr int F = NCH0 + NCH1;
  static constexpr int KS = 2 * (GL0 + GL1);   // steps = slots / 2
  static constexpr int K = 2 * KS;
  static constexpr int NU = (K + 31) / 32;
  __device__ static constexpr int gl(int l) { return l == 0 ? GL0 : GL1; }
  __device__ static constexpr int nch(int l) { return l == 0 ? NCH0 : NCH1; }
  __device__ static constexpr int choff(int l) { return l == 0 ? 0 : NCH0; }
  __device__ static constexpr int stepoff(int l) { return l == 0 ? 0 : 2 * GL0; }
};

// Wt[slot][px] of this lane's pixel for step (l, z, yb) and x corner = half; deriv: d/d(gray)
__device__ __forceinline__ float ni_weight(const NiPix &c, float fy, int z, int yb, int half, bool deriv) {
  const float wx = half ? c.fx : 1.f - c.fx;
  const float wy = yb ? fy : 1.f - fy;
  float wz;
  if (!deriv) wz = (c.z0 == z ? 1.f - c.fz : 0.f) + (c.z1 == z ? c.fz : 0.f);
  else wz = ((c.z1 == z ? 1.f : 0.f) - (c.z0 == z ? 1.f : 0.f)) * c.isc;
  return wz * wy * wx;
}

template <class S>
__device__ __forceinline__ acc16 ni_slice(const float *__restrict__ gimg, int lane, int half, const NiPix (&c)[2], const float (&fy)[2],
                                          bool deriv) {
  acc16 acc = zero16();
#pragma unroll
  for (int l = 0; l < S::NL; l++)
#pragma unroll
    for (int z = 0; z < S::gl(l); z++)
#pragma unroll
      for (int yb = 0; yb < 2; yb++) {
        const int s = S::stepoff(l) + z * 2 + yb;
        acc = mfma(gimg[s * kWave + lane], ni_weight(c[l], fy[l], z, yb, half, deriv), acc);
      }
  return acc;
}

// the region's nodes as A operands: lane (channel = col, x corner = half), step (l, z, yb)
template <class S>
__device__ __forceinline__ void ni_load_region(const NiParams &p, float *__restrict__ gimg, int lane, int col, int half, const int (&x0)[2],
                                               const int (&y0)[2]) {
#pragma unroll
  for (int l = 0; l < S::NL; l++) {
    const NiLevel &L = p.lv[l];
    const int ch = col - S::choff(l);
    const bool mine = ch >= 0 && ch < S::nch(l);
    const int x = x0[l] + half < L.gx ? x0[l] + half : L.gx - 1;
#pragma unroll
    for (int z = 0; z < S::gl(l); z++)
#pragma unroll
      for (int yb = 0; yb < 2; yb++) {
        const int y = y0[l] + yb < L.gy ? y0[l] + yb : L.gy - 1;
        const int s = S::stepoff(l) + z * 2 + yb;
        gimg[s * kWave + lane] = mine ? L.grid[(((int64_t)ch * L.gl + z) * L.gy + y) * L.gx + x] : 0.f;
      }
  }
}

template <class S>
__device__ __forceinline__ void ni_pixel(const NiParams &p, int x, float r, float g, float b, NiPix (&c)[2]) {
  const float x01 = linspace01_s(x, p.W, p.lin_x);
  const float gray = rgb2gray(r, g, b);
#pragma unroll
  for (int l = 0; l < S::NL; l++) {
    int xi;
    axis_cell(x01, p.lv[l].gx, xi, c[l].fx);
    bool interior;
    const float iz = guide_coord(gray, S::gl(l), interior);
    const float zf = floorf(iz);
    c[l].z0 = (int)zf;
    c[l].z1 = c[l].z0 + 1 < S::gl(l) ? c[l].z0 + 1 : S::gl(l) - 1;
    c[l].fz = iz - zf;
    c[l].isc = interior ? (float)(S::gl(l) - 1) : 0.f;
  }
}

struct NiJob {
  int r0, r1, t0, t1, xs, ys;
};
__device__ __forceinline__ NiJob ni_job(const NiTables &T, int j, int rows_per_job) {
  const int per_band = T.nseg * T.cpb;
  const int b = j / per_band, rem = j - b * per_band, i = rem / T.cpb, k = rem - i * T.cpb;
  NiJob J;
  J.r0 = T.band_start[b] + k * rows_per_job;
  const int r1 = J.r0 + rows_per_job;
  J.r1 = r1 < T.band_start[b + 1] ? r1 : T.band_start[b + 1];
  J.t0 = T.seg_tile[i]; J.t1 = T.seg_tile[i + 1];
  J.xs = T.seg_start[i]; J.ys = T.band_start[b];
  return J;
}

template <int NCH0, int GL0, int NCH1, int GL1>
__global__ __launch_bounds__(kMhBlock, 2) void neural_image_fwd_kernel(NiParams p, const float *__restrict__ rgb, const float *__restrict__ w1,
                                                                   const float *__restrict__ w2, const float *__restrict__ w3,
                                                                   int residual, float *__restrict__ out) {
  using S = NiShape<NCH0, GL0, NCH1, GL1>;
  constexpr int F = S::F;
  extern __shared__ float lds[];
  __shared__ NiTables T;
  mh_load_weights<F>(lds, w1, w2, w3);
  ni_build_tables<S::NL>(p, T);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, col = lane & 31, half = lane >> 5;
  float *gimg = lds + MhImage<F>::floats + wave * (S::KS * kWave);
  const int njobs = T.nband * T.nseg * T.cpb;
  for (int j = blockIdx.x * kMhWaves + wave; j < njobs; j += gridDim.x * kMhWaves) {
    const NiJob J = ni_job(T, j, p.rows_per_job);
    if (J.r0 >= J.r1) continue;
    int x0[2] = {0, 0}, y0[2] = {0, 0};
    float ftmp;
#pragma unroll
    for (int l = 0; l < S::NL; l++) {
      axis_cell(linspace01_s(J.xs, p.W, p.lin_x), p.lv[l].gx, x0[l], ftmp);
      axis_cell(linspace01_s(J.ys, p.H, p.lin_y), p.lv[l].gy, y0[l], ftmp);
    }
    ni_load_region<S>(p, gimg, lane, col, half, x0, y0);
    for (int y = J.r0; y < J.r1; y++) {
      float fy[2] = {0.f, 0.f};
      int ytmp;
#pragma unroll
      for (int l = 0; l < S::NL; l++) axis_cell(linspace01_s(y, p.H, p.lin_y), p.lv[l].gy, ytmp, fy[l]);
      for (int t = J.t0; t < J.t1; t++) {
        const int n = T.tile_n[t];
        const bool on = col < n;
        const int x = T.tile_x[t] + (on ? col : n - 1);
        const int64_t i = (int64_t)y * p.W + x;
        const float c0 = rgb[i * 3], c1 = rgb[i * 3 + 1], c2 = rgb[i * 3 + 2];
        NiPix c[2];
        ni_pixel<S>(p, x, c0, c1, c2, c);
        const acc16 xt = ni_slice<S>(gimg, lane, half, c, fy, false);
        acc16 h1[2], h2[2], aff;
        mh_layer1_tile<F>(lds, col, half, xt, h1);
        mh_layers23<F>(lds, col, half, h1, h2, aff);
        if (!on) continue;
        float lo = aff[0] * c0 + aff[1] * c1 + aff[2] * c2 + aff[3];
        float hi = aff[4] * c0 + aff[5] * c1 + aff[6] * c2 + aff[7];
        if (residual) { lo += half ? c1 : c0; hi += c2; }
        out[i * 3 + half] = lo;
        if (half == 0) out[i * 3 + 2] = hi;
      }
    }
  }
}

template <int NCH0, int GL0, int NCH1, int GL1>
__global__ __launch_bounds__(kMhBlock) void neural_image_bwd_kernel(NiParams p, const float *__restrict__ rgb, const float *__restrict__ w1,
                                                                   const float *__restrict__ w2, const float *__restrict__ w3,
                                                                   int residual, const float *__restrict__ v_out,
                                                                   float *__restrict__ v_rgb, float *__restrict__ partials) {
  using S = NiShape<NCH0, GL0, NCH1, GL1>;
  constexpr int F = S::F;
  using Im = MhImage<F>;
  extern __shared__ float lds[];
  __shared__ NiTables T;
  mh_load_weights<F>(lds, w1, w2, w3);
  ni_build_tables<S::NL>(p, T);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, col = lane & 31, half = lane >> 5;
  float *bufA = lds + Im::floats + wave * (2 * kMhBufBig + kMhBufSmall + S::KS * kWave);
  float *bufB = bufA + kMhBufBig;
  float *bufS = bufB + kMhBufBig;
  float *gimg = bufS + kMhBufSmall;
  acc16 g1[2], g2[4], g3[2], gs[S::NU];
#pragma unroll
  for (int i = 0; i < 2; i++) { g1[i] = zero16(); g3[i] = zero16(); }
#pragma unroll
  for (int i = 0; i < 4; i++) g2[i] = zero16();

  const int njobs = T.nband * T.nseg * T.cpb;
  for (int j = blockIdx.x * kMhWaves + wave; j < njobs; j += gridDim.x * kMhWaves) {
    const NiJob J = ni_job(T, j, p.rows_per_job);
    if (J.r0 >= J.r1) continue;
    int x0[2] = {0, 0}, y0[2] = {0, 0};
    float ftmp;
#pragma unroll
    for (int l = 0; l < S::NL; l++) {
      axis_cell(linspace01_s(J.xs, p.W, p.lin_x), p.lv[l].gx, x0[l], ftmp);
      axis_cell(linspace01_s(J.ys, p.H, p.lin_y), p.lv[l].gy, y0[l], ftmp);
    }
    ni_load_region<S>(p, gimg, lane, col, half, x0, y0);
#pragma unroll
    for (int u = 0; u < S::NU; u++) gs[u] = zero16();
    for (int y = J.r0; y < J.r1; y++) {
      float fy[2] = {0.f, 0.f};
      int ytmp;
#pragma unroll
      for (int l = 0; l < S::NL; l++) axis_cell(linspace01_s(y, p.H, p.lin_y), p.lv[l].gy, ytmp, fy[l]);
      for (int t = J.t0; t < J.t1; t++) {
        const int n = T.tile_n[t];
        const bool on = col < n;
        const int x = T.tile_x[t] + (on ? col : n - 1);
        const int64_t i = (int64_t)y * p.W + x;
        const float c0 = rgb[i * 3], c1 = rgb[i * 3 + 1], c2 = rgb[i * 3 + 2];
        NiPix c[2];
        ni_pixel<S>(p, x, c0, c1, c2, c);
        const acc16 xt = ni_slice<S>(gimg, lane, half, c, fy, false);
        acc16 h1[2], h2[2], aff;
        mh_layer1_tile<F>(lds, col, half, xt, h1);
        mh_layers23<F>(lds, col, half, h1, h2, aff);

        const float g0 = on ? v_out[i * 3] : 0.f, gg1 = on ? v_out[i * 3 + 1] : 0.f, gg2 = on ? v_out[i * 3 + 2] : 0.f;
        const float r_lo = half ? gg1 : g0, r_hi = half ? 0.f : gg2;
        float t8[8];
        t8[0] = r_lo * c0; t8[1] = r_lo * c1; t8[2] = r_lo * c2; t8[3] = r_lo;
        t8[4] = r_hi * c0; t8[5] = r_hi * c1; t8[6] = r_hi * c2; t8[7] = r_hi;
        float p0 = aff[0] * r_lo + aff[4] * r_hi, p1 = aff[1] * r_lo + aff[5] * r_hi, p2 = aff[2] * r_lo + aff[6] * r_hi;
        p0 += __shfl_xor(p0, 32); p1 += __shfl_xor(p1, 32); p2 += __shfl_xor(p2, 32);

        // ---- layer 3
#pragma unroll
        for (int r = 0; r < 8; r++) bufS[d_row(r, half) * kMhTStride + col] = t8[r];
        mh_store_tiles<2>(bufA, col, half, h2);
        mh_wave_fence();
        mh_outer<1, 2>(bufS, bufA, col, half, g3);
        mh_wave_fence();
#pragma unroll
        for (int o = 0; o < 2; o++) {
          acc16 acc = zero16();
#pragma unroll
          for (int s = 0; s < 8; s++) acc = mfma(lds[Im::off3 + d_row(s, half) * kMhWStride + 32 * o + col], t8[s], acc);
#pragma unroll
          for (int r = 0; r < 16; r++) h2[o][r] = acc[r] * (1.f - h2[o][r] * h2[o][r]);
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
            for (int r = 0; r < 16; r++) h1[o][r] = d[o][r] * (1.f - h1[o][r] * h1[o][r]);
        }
        // ---- layer 1: the feature tile is already in D layout
        mh_store_tiles<2>(bufA, col, half, h1);
        {
          const acc16 xs[1] = {xt};
          mh_store_tiles<1>(bufS, col, half, xs);
        }
        mh_wave_fence();
        mh_outer<2, 1>(bufA, bufS, col, half, g1);
        mh_wave_fence();
        acc16 dx = zero16();
        {
          const bool live = col < F;
          const int cc = live ? col : 0;
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int s = 0; s < 16; s++) {
              const float w = lds[Im::off1 + (32 * b + d_row(s, half)) * Im::S1 + cc];
              dx = mfma(live ? w : 0.f, h1[b][s], dx);
            }
        }
        // ---- the slice: guidance gradient and the region's slot gradient
        const acc16 dz = ni_slice<S>(gimg, lane, half, c, fy, true);
        float vg = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) vg += dx[r] * dz[r];
        vg += __shfl_xor(vg, 32);
        if (v_rgb && on && half == 0) {
          v_rgb[i * 3] = p0 + (residual ? g0 : 0.f) + vg * kGrayR;
          v_rgb[i * 3 + 1] = p1 + (residual ? gg1 : 0.f) + vg * kGrayG;
          v_rgb[i * 3 + 2] = p2 + (residual ? gg2 : 0.f) + vg * kGrayB;
        }
#pragma unroll
        for (int l = 0; l < S::NL; l++)
#pragma unroll
          for (int z = 0; z < S::gl(l); z++)
#pragma unroll
            for (int yb = 0; yb < 2; yb++) {
              const int s = S::stepoff(l) + z * 2 + yb;
              bufB[(2 * s + half) * kMhTStride + col] = ni_weight(c[l], fy[l], z, yb, half, false);
            }
        {
          const acc16 ds[1] = {dx};
          mh_store_tiles<1>(bufA, col, half, ds);
        }
        mh_wave_fence();
        mh_outer<S::NU, 1>(bufB, bufA, col, half, gs);
        mh_wave_fence();
      }
    }
    // flush the region: register r of block u is slot q = 32 u + d_row(r, half), column = channel
#pragma unroll
    for (int u = 0; u < S::NU; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int q = 32 * u + d_row(r, half);
        if (q >= S::K) continue;
        const int l = (S::NL == 2 && q >= 4 * GL0) ? 1 : 0;
        const int ql = q - (l ? 4 * GL0 : 0);
        const int xb = ql & 1, yb = (ql >> 1) & 1, z = ql >> 2;
        const NiLevel &L = p.lv[l];
        const int ch = col - (l ? NCH0 : 0), nch = l ? NCH1 : NCH0;
        if (!L.v_grid || ch < 0 || ch >= nch) continue;
        const int xx = x0[l] + xb < L.gx ? x0[l] + xb : L.gx - 1, yy = y0[l] + yb < L.gy ? y0[l] + yb : L.gy - 1;
        const float v = gs[u][r];
        if (v != 0.f) atomicAdd(L.v_grid + (((int64_t)ch * L.gl + z) * L.gy + yy) * L.gx + xx, v);
      }
  }

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

// ---- host side of the fused image transform --------------------------------------------------------------------------------------
namespace bds {

inline int ni_shape_id(int n_levels, const bds_feat_level *lv) {
  if (n_levels == 1 && lv[0].gl == 8) {
    switch (lv[0].nch) { case 8: return 0; case 16: return 1; case 24: return 2; case 32: return 3; default: return -1; }
  }
  if (n_levels == 1 && lv[0].gl == 4 && lv[0].nch == 24) return 5;
  if (n_levels == 2 && lv[0].gl == 1 && lv[0].nch == 8 && lv[1].nch == 8) {
    if (lv[1].gl == 8) return 4;
    if (lv[1].gl == 4) return 6;
  }
  return -1;
}

inline bool ni_fill(NiParams &p, int H, int W, int n_levels, const bds_feat_level *lv, int rows_per_job) {
  p.H = H; p.W = W; p.rows_per_job = rows_per_job;
  p.lin_x = W > 1 ? 1.0f / (float)(W - 1) : 0.f;
  p.lin_y = H > 1 ? 1.0f / (float)(H - 1) : 0.f;
  int segs = 1, bands = 1;
  for (int l = 0; l < 2; l++) {
    if (l < n_levels) {
      if (!lv[l].grid || lv[l].gx < 1 || lv[l].gy < 1) return false;
      p.lv[l] = NiLevel{lv[l].grid, lv[l].v_grid, lv[l].gx, lv[l].gy, lv[l].gl};
      segs += lv[l].gx > 1 ? lv[l].gx - 2 : 0;
      bands += lv[l].gy > 1 ? lv[l].gy - 2 : 0;
    } else {
      p.lv[l] = NiLevel{nullptr, nullptr, 1, 1, 1};
    }
  }
  // table capacities: cell boundaries per axis, 32-pixel tiles per row
  return segs <= kNiMaxSeg && bands <= kNiMaxSeg && (W + kMhTile - 1) / kMhTile + segs <= kNiMaxTile;
}

inline int ni_jobs_upper(const NiParams &p, int n_levels, const bds_feat_level *lv) {
  int segs = 1, bands = 1;
  for (int l = 0; l < n_levels; l++) { segs += lv[l].gx > 1 ? lv[l].gx - 2 : 0; bands += lv[l].gy > 1 ? lv[l].gy - 2 : 0; }
  const int tallest = (p.H + bands - 1) / bands + 1;
  return bands * segs * ((tallest + p.rows_per_job - 1) / p.rows_per_job);
}

template <int NCH0, int GL0, int NCH1, int GL1>
int ni_launch_fwd(const NiParams &p, int grid, const float *rgb, const float *w1, const float *w2, const float *w3, int residual, float *out,
                  hipStream_t st) {
  using S = NiShape<NCH0, GL0, NCH1, GL1>;
  const size_t lds_bytes = (MhImage<S::F>::floats + kMhWaves * S::KS * kWave) * sizeof(float);
  hipLaunchKernelGGL((neural_image_fwd_kernel<NCH0, GL0, NCH1, GL1>), dim3(grid), dim3(kMhBlock), lds_bytes, st, p, rgb, w1, w2, w3, residual,
                     out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

template <int NCH0, int GL0, int NCH1, int GL1>
int ni_launch_bwd(const NiParams &p, int grid, const float *rgb, const float *w1, const float *w2, const float *w3, int residual,
                  const float *v_out, float *v_rgb, float *partials, hipStream_t st) {
  using S = NiShape<NCH0, GL0, NCH1, GL1>;
  const size_t lds_bytes = (MhImage<S::F>::floats + kMhWaves * (2 * kMhBufBig + kMhBufSmall + S::KS * kWave)) * sizeof(float);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&neural_image_bwd_kernel<NCH0, GL0, NCH1, GL1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
    return BDS_ELAUNCH;
  hipLaunchKernelGGL((neural_image_bwd_kernel<NCH0, GL0, NCH1, GL1>), dim3(grid), dim3(kMhBlock), lds_bytes, st, p, rgb, w1, w2, w3, residual,
                     v_out, v_rgb, partials);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

constexpr int kNiGridBwd = 256;   // one workgroup per CU (142 KB of LDS each)

}  // namespace bds

extern "C" int bds_neural_image_ok(int H, int W, int n_levels, const bds_feat_level *levels, int hidden) {
  if (H < 1 || W < 1 || n_levels < 1 || n_levels > 2 || !levels || hidden != kMhHid) return 0;
  if (ni_shape_id(n_levels, levels) < 0) return 0;
  NiParams p;
  bds_feat_level probe[2];
  for (int l = 0; l < n_levels; l++) { probe[l] = levels[l]; if (!probe[l].grid) probe[l].grid = reinterpret_cast<const float *>(16); }
  return ni_fill(p, H, W, n_levels, probe, 4) ? 1 : 0;
}

extern "C" size_t bds_neural_image_bwd_temp_bytes(int F) {
  if (!mh_supported(F)) return 0;
  return (size_t)kNiGridBwd * kMhWaves * (size_t)(kMhHid * F + kMhHid * kMhHid + kMhAff * kMhHid) * sizeof(float);
}

extern "C" int bds_neural_image_fwd(int H, int W, int n_levels, const bds_feat_level *levels, int hidden, const float *rgb, const float *w1,
                                    const float *w2, const float *w3, int residual, float *out, bds_stream_t stream) {
  BDS_REQUIRE(H >= 0 && W >= 0 && n_levels >= 1 && n_levels <= 2 && levels && hidden == kMhHid);
  if (H == 0 || W == 0) return BDS_OK;
  const int id = ni_shape_id(n_levels, levels);
  NiParams p;
  BDS_REQUIRE(id >= 0 && ni_fill(p, H, W, n_levels, levels, 4) && rgb && w1 && w2 && w3 && out);
  const int jobs = ni_jobs_upper(p, n_levels, levels);
  int grid = (jobs + kMhWaves - 1) / kMhWaves;
  grid = grid < 768 ? (grid > 0 ? grid : 1) : 768;
  hipStream_t st = as_stream(stream);
  switch (id) {
    case 0: return ni_launch_fwd<8, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, out, st);
    case 1: return ni_launch_fwd<16, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, out, st);
    case 2: return ni_launch_fwd<24, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, out, st);
    case 3: return ni_launch_fwd<32, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, out, st);
    case 4: return ni_launch_fwd<8, 1, 8, 8>(p, grid, rgb, w1, w2, w3, residual, out, st);
    case 5: return ni_launch_fwd<24, 4, 0, 0>(p, grid, rgb, w1, w2, w3, residual, out, st);
    default: return ni_launch_fwd<8, 1, 8, 4>(p, grid, rgb, w1, w2, w3, residual, out, st);
  }
}

extern "C" int bds_neural_image_bwd(int H, int W, int n_levels, const bds_feat_level *levels, int hidden, const float *rgb, const float *w1,
                                    const float *w2, const float *w3, int residual, const float *v_out, float *v_rgb, float *v_w1,
                                    float *v_w2, float *v_w3, int accumulate_w, void *temp, size_t temp_bytes, bds_stream_t stream) {
  BDS_REQUIRE(H >= 0 && W >= 0 && n_levels >= 1 && n_levels <= 2 && levels && hidden == kMhHid);
  if (H == 0 || W == 0) return BDS_OK;
  const int id = ni_shape_id(n_levels, levels);
  NiParams p;
  BDS_REQUIRE(id >= 0 && ni_fill(p, H, W, n_levels, levels, 8) && rgb && w1 && w2 && w3 && v_out);
  int F = 0;
  for (int l = 0; l < n_levels; l++) F += levels[l].nch;
  BDS_REQUIRE(temp && temp_bytes >= bds_neural_image_bwd_temp_bytes(F));
  hipStream_t st = as_stream(stream);
  float *partials = static_cast<float *>(temp);
  const int grid = kNiGridBwd;
  int rc;
  switch (id) {
    case 0: rc = ni_launch_bwd<8, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    case 1: rc = ni_launch_bwd<16, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    case 2: rc = ni_launch_bwd<24, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    case 3: rc = ni_launch_bwd<32, 8, 0, 0>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    case 4: rc = ni_launch_bwd<8, 1, 8, 8>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    case 5: rc = ni_launch_bwd<24, 4, 0, 0>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
    default: rc = ni_launch_bwd<8, 1, 8, 4>(p, grid, rgb, w1, w2, w3, residual, v_out, v_rgb, partials, st); break;
  }
  if (rc != BDS_OK) return rc;
  if (v_w1 || v_w2 || v_w3) {
    const int len = kMhHid * F + kMhHid * kMhHid + kMhAff * kMhHid;
    hipLaunchKernelGGL(mlp_head_reduce_kernel, dim3((unsigned)cdiv(len, 32)), dim3(256), 0, st, grid * kMhWaves, F, partials, v_w1, v_w2,
                       v_w3, accumulate_w);
    BDS_LAUNCH_CHECK();
  }
  return BDS_OK;
}

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
