// K9-K13: sky blend + multi-scale bilateral-grid slice + 3x4 affine colour transform, TV.
// Restates (file:line under /root/reference/project):
//   bilateral/lib_bilagrid.py:171-230,317-368   slice(), BilateralGrid.forward
//   models/modules.py:317-335,494-522,409-420   Bilateral / MultiScaleBilateral AffineTransform.forward
//   models/trainers/scene_graph.py:95-98,112-117,292-294   application, composition, sky blend
//   bilateral/lib_bilagrid.py:152-168           total_variation_loss
//
// HBM-bound pixel work.  The reference materialises one [H,W,3,4] affine map per level
// (>= 96 B/pixel/level); here only the LOW-resolution maps exist in memory (0.375 x 48 B per
// full-res pixel for the shipped 3-level config) and the full-resolution kernel re-derives each
// level's 3x4 matrix from four cached taps while the pixel stays in registers: 24-40 B/pixel of
// HBM traffic forward.  Backward is gather-based (deterministic, no atomics on image-sized arrays);
// grid gradients are accumulated per workgroup in LDS and flushed once.
#include <stdio.h>
#include <string.h>

#include "bilagrid_ms.h"
#include "ed_epilogue.h"

namespace bds {

// ---- A: per-level low-resolution slice -------------------------------------------------------
// trilinear sample of the 12 channels from a CELL-MAJOR copy of the grid ([cell][12] as three float4): the arithmetic (and its
// order) is slice_sample's, the eight taps are three 16-byte reads each instead of twelve scalar gathers
__device__ __forceinline__ void slice_sample_cells(const float4 *__restrict__ cells, int gx, int gy, const Cell &c, float *out12) {
  const int plane = gy * gx;
  const int o00 = c.y0 * gx + c.x0, o01 = c.y0 * gx + c.x1, o10 = c.y1 * gx + c.x0, o11 = c.y1 * gx + c.x1;
  const float w00 = (1.f - c.fy) * (1.f - c.fx), w01 = (1.f - c.fy) * c.fx, w10 = c.fy * (1.f - c.fx), w11 = c.fy * c.fx;
  const float4 *g0 = cells + (int64_t)c.z0 * plane * 3, *g1 = cells + (int64_t)c.z1 * plane * 3;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a00 = g0[o00 * 3 + q], a01 = g0[o01 * 3 + q], a10 = g0[o10 * 3 + q], a11 = g0[o11 * 3 + q];
    const float4 b00 = g1[o00 * 3 + q], b01 = g1[o01 * 3 + q], b10 = g1[o10 * 3 + q], b11 = g1[o11 * 3 + q];
    const float ax = a00.x * w00 + a01.x * w01 + a10.x * w10 + a11.x * w11, bx = b00.x * w00 + b01.x * w01 + b10.x * w10 + b11.x * w11;
    const float ay = a00.y * w00 + a01.y * w01 + a10.y * w10 + a11.y * w11, by = b00.y * w00 + b01.y * w01 + b10.y * w10 + b11.y * w11;
    const float az = a00.z * w00 + a01.z * w01 + a10.z * w10 + a11.z * w11, bz = b00.z * w00 + b01.z * w01 + b10.z * w10 + b11.z * w11;
    const float aw = a00.w * w00 + a01.w * w01 + a10.w * w10 + a11.w * w11, bw = b00.w * w00 + b01.w * w01 + b10.w * w10 + b11.w * w11;
    out12[q * 4 + 0] = ax * (1.f - c.fz) + bx * c.fz;
    out12[q * 4 + 1] = ay * (1.f - c.fz) + by * c.fz;
    out12[q * 4 + 2] = az * (1.f - c.fz) + bz * c.fz;
    out12[q * 4 + 3] = aw * (1.f - c.fz) + bw * c.fz;
  }
}

// d(slice)/d(iz) of the 12 channels from the cell-major copy (slice_sample's dz: upper plane minus lower plane)
__device__ __forceinline__ void slice_dz_cells(const float4 *__restrict__ cells, int gx, int gy, const Cell &c, float *dz12) {
  const int plane = gy * gx;
  const int o00 = c.y0 * gx + c.x0, o01 = c.y0 * gx + c.x1, o10 = c.y1 * gx + c.x0, o11 = c.y1 * gx + c.x1;
  const float w00 = (1.f - c.fy) * (1.f - c.fx), w01 = (1.f - c.fy) * c.fx, w10 = c.fy * (1.f - c.fx), w11 = c.fy * c.fx;
  const float4 *g0 = cells + (int64_t)c.z0 * plane * 3, *g1 = cells + (int64_t)c.z1 * plane * 3;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a00 = g0[o00 * 3 + q], a01 = g0[o01 * 3 + q], a10 = g0[o10 * 3 + q], a11 = g0[o11 * 3 + q];
    const float4 b00 = g1[o00 * 3 + q], b01 = g1[o01 * 3 + q], b10 = g1[o10 * 3 + q], b11 = g1[o11 * 3 + q];
    dz12[q * 4 + 0] = (b00.x * w00 + b01.x * w01 + b10.x * w10 + b11.x * w11) - (a00.x * w00 + a01.x * w01 + a10.x * w10 + a11.x * w11);
    dz12[q * 4 + 1] = (b00.y * w00 + b01.y * w01 + b10.y * w10 + b11.y * w11) - (a00.y * w00 + a01.y * w01 + a10.y * w10 + a11.y * w11);
    dz12[q * 4 + 2] = (b00.z * w00 + b01.z * w01 + b10.z * w10 + b11.z * w11) - (a00.z * w00 + a01.z * w01 + a10.z * w10 + a11.z * w11);
    dz12[q * 4 + 3] = (b00.w * w00 + b01.w * w01 + b10.w * w10 + b11.w * w11) - (a00.w * w00 + a01.w * w01 + a10.w * w10 + a11.w * w11);
  }
}

// kLds: the level's grid(s) (n_avg * 12*gl*gy*gx floats) are staged cell-major in the workgroup's LDS once, and the workgroup then
// walks a grid-stride range of low-res pixels (persistent workgroups: the staging is amortised over several chunks).  The
// global-memory form gathered 96 scalars per pixel from L1/L2 and was bound by the latency of those chains.
template <bool kLds>
__global__ __launch_bounds__(kBgBlock) void ms_lowres_fwd_kernel(MsParams p, LevelSched sc) {
  extern __shared__ __attribute__((aligned(16))) float lds_grid[];
  int local;
  const int k_entry = sched_find(sc, blockIdx.x, local);
  const int l = sc.level[k_entry];
  const LevelDev &L = p.lv[l];
  const int vol = L.gl * L.gy * L.gx, gsz = 12 * vol;
  if (kLds) {   // cell-major copy of the grid; (image, channel) outermost: no integer division per element
    for (int nc = 0; nc < 12 * L.n_avg; nc++) {
      const int n = nc / 12, ch = nc - n * 12;
      for (int cell = threadIdx.x; cell < vol; cell += kBgBlock) lds_grid[(n * vol + cell) * 12 + ch] = L.grid[(int64_t)nc * vol + cell];
    }
    __syncthreads();
  }
  const int64_t n_low = (int64_t)L.Hd * L.Wd;
  const int64_t n_chunks = (n_low + kBgBlock - 1) / kBgBlock, per_wg = (n_chunks + sc.nblk[k_entry] - 1) / sc.nblk[k_entry];
  const int64_t c_first = (int64_t)xcd_contiguous(local, sc.nblk[k_entry]) * per_wg;   // contiguous range per workgroup, per XCD
  const int64_t c_last = c_first + per_wg < n_chunks ? c_first + per_wg : n_chunks;
  for (int64_t idx = c_first * kBgBlock + threadIdx.x; idx < c_last * kBgBlock && idx < n_low; idx += kBgBlock) {
    int i, j;
    fast_divmod((uint32_t)idx, (uint32_t)L.Wd, L.magic_wd, i, j);
    const Tap ty = resample_tap_s(i, L.Hd, p.H, L.dn_y), tx = resample_tap_s(j, L.Wd, p.W, L.dn_x);
    float r, g, b;
    lowres_colour(p, ty, tx, r, g, b);
    const float gray = rgb2gray(r, g, b);
    L.lg[idx] = gray;   // (kept for the cell-aligned backward, csrc/bilagrid_cells.hip)
    const Cell c = slice_cell(linspace01_s(j, L.Wd, L.lin_x), linspace01_s(i, L.Hd, L.lin_y), gray, L.gx, L.gy, L.gl);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0.f;
    for (int n = 0; n < L.n_avg; n++) {
      float a[12];
      if (kLds) slice_sample_cells(reinterpret_cast<const float4 *>(lds_grid) + (int64_t)n * vol * 3, L.gx, L.gy, c, a);
      else slice_sample(L.grid + (int64_t)n * gsz, L.gx, L.gy, L.gl, c, a, nullptr);
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] += a[k];
    }
    float4 *dst = reinterpret_cast<float4 *>(L.lo + idx * 12);
    if (L.n_avg > 1) {
      const float inv = (float)L.n_avg;
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = acc[k] / inv;
    }
    dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    dst[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
  }
}

// bilinear up-sample of one level's low-res map at full-res pixel (i,j)
__device__ __forceinline__ void upsample_affine(const LevelDev &L, int H, int W, int i, int j, float *A) {
  if (L.Hd == H && L.Wd == W) {
    const float4 *s = reinterpret_cast<const float4 *>(L.lo) + times3(row_major(i, W, j));
    const float4 a = s[0], b = s[1], c = s[2];
    A[0] = a.x; A[1] = a.y; A[2] = a.z; A[3] = a.w; A[4] = b.x; A[5] = b.y; A[6] = b.z; A[7] = b.w;
    A[8] = c.x; A[9] = c.y; A[10] = c.z; A[11] = c.w;
    return;
  }
  const Tap ty = resample_tap_s(i, H, L.Hd, L.up_y), tx = resample_tap_s(j, W, L.Wd, L.up_x);
  const float4 *lo4 = reinterpret_cast<const float4 *>(L.lo);
  const int r0 = __mul24(ty.i0, L.Wd), r1 = __mul24(ty.i1, L.Wd);
  const float4 *s00 = lo4 + times3(r0 + tx.i0);
  const float4 *s01 = lo4 + times3(r0 + tx.i1);
  const float4 *s10 = lo4 + times3(r1 + tx.i0);
  const float4 *s11 = lo4 + times3(r1 + tx.i1);
  const float wx = tx.w1, wy = ty.w1;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a = s00[q], b = s01[q], c = s10[q], d = s11[q];
    A[q * 4 + 0] = (a.x * (1.f - wx) + b.x * wx) * (1.f - wy) + (c.x * (1.f - wx) + d.x * wx) * wy;
    A[q * 4 + 1] = (a.y * (1.f - wx) + b.y * wx) * (1.f - wy) + (c.y * (1.f - wx) + d.y * wx) * wy;
    A[q * 4 + 2] = (a.z * (1.f - wx) + b.z * wx) * (1.f - wy) + (c.z * (1.f - wx) + d.z * wx) * wy;
    A[q * 4 + 3] = (a.w * (1.f - wx) + b.w * wx) * (1.f - wy) + (c.w * (1.f - wx) + d.w * wx) * wy;
  }
}

// ---- B: full-resolution compose --------------------------------------------------------------
// (Measured and dropped, round 2: staging the y-interpolated low-res rows of a 256-pixel run in LDS so that a pixel reads 6 LDS
// entries per level instead of 12 float4 gathers -- the staging prologue + barrier cost more than the gathers it removed:
// forward 44 -> 48 us, backward x kernel 88 -> 111 us at 1080p.  The gathers overlap across waves; a per-workgroup prologue does not.)
template <int NL, bool kTrain>  // NL >= p.nlevels: bounds the static unrolling (registers) of the level loop
__global__ __launch_bounds__(kBgBlock) void ms_apply_fwd_kernel(MsParams p, float *__restrict__ out, TrainLoss tl) {
  __shared__ float red[kBgBlock / kWave];
  if (kTrain && (int)blockIdx.x >= tl.pix_blocks) {   // the TV term of the loss: one grid element per thread
    const float t = block_sum_to_thread0(tv_train_element(tl.T, (int)blockIdx.x - tl.pix_blocks, tl.v_loss), red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(tl.loss + (size_t)(blockIdx.x & (tl.loss_slots - 1)) * kLossSlotStride, t);
    return;
  }
  // each XCD works on one contiguous band of the image: the rows of the low-res maps that 2f consecutive pixel rows share are then
  // fetched into ONE private L2 (measured before: 236 MB of fabric traffic for 136 MB of distinct data)
  const int pix = xcd_contiguous((int)blockIdx.x, kTrain ? tl.pix_blocks : (int)gridDim.x) * kBgBlock + (int)threadIdx.x;
  float l1 = 0.f;
  if (pix < p.H * p.W) {
    int i, j;
    fast_divmod((uint32_t)pix, (uint32_t)p.W, p.magic_w, i, j);
    float r, g, b;
    load_input(p, i, j, r, g, b);
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l < p.nlevels) {
        float A[12];
        upsample_affine(p.lv[l], p.H, p.W, i, j, A);
        if (p.lv[l].aff_out) {
          float4 *d = reinterpret_cast<float4 *>(p.lv[l].aff_out) + times3(pix);
          d[0] = make_float4(A[0], A[1], A[2], A[3]);
          d[1] = make_float4(A[4], A[5], A[6], A[7]);
          d[2] = make_float4(A[8], A[9], A[10], A[11]);
        }
        apply_affine(A, r, g, b);
      }
    }
    const int p3 = times3(pix);
    out[p3] = r; out[p3 + 1] = g; out[p3 + 2] = b;
    if (p.depth_out) p.depth_out[pix] = p.rgb[(pix << 2) + 3] / fmaxf(p.alpha[pix], 1e-10f);
    if (kTrain) {   // photometric L1 of the pixel just produced + its gradient (torch: sign(0) = 0)
      const float gs = tl.v_loss * tl.inv_n;
      const float d0 = r - tl.target[p3], d1 = g - tl.target[p3 + 1], d2 = b - tl.target[p3 + 2];
      l1 = fabsf(d0) + fabsf(d1) + fabsf(d2);
      tl.v_out[p3] = d0 > 0.f ? gs : (d0 < 0.f ? -gs : 0.f);
      tl.v_out[p3 + 1] = d1 > 0.f ? gs : (d1 < 0.f ? -gs : 0.f);
      tl.v_out[p3 + 2] = d2 > 0.f ? gs : (d2 < 0.f ? -gs : 0.f);
    }
  }
  if (kTrain) {
    const float t = block_sum_to_thread0(l1, red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(tl.loss + (size_t)(blockIdx.x & (tl.loss_slots - 1)) * kLossSlotStride, t * tl.inv_n);
  }
}

// ---- C: full-resolution backward: direct route + per-level (P, Q) ------------------------------------
// d(loss)/d(A_l) at a pixel is the outer product Q (x) [P;1] of the gradient arriving at level l's output
// and the colour entering it; only those 6 floats per level are stored.  The adjoint of the bilinear
// up-sampler is then applied SEPARABLY and gather-style (deterministic, no atomics): an x pass
// (kernel D1) reduces each image row onto the low-res columns, a y pass (fused in kernel D2) reduces
// those onto the low-res rows -- 2(2f+2) taps per output instead of (2f+2)^2.
template <int NL>
__global__ __launch_bounds__(kBgBlock) void ms_apply_bwd_kernel(MsParams p, const float *__restrict__ v_out,
                                                               float *__restrict__ v_in) {
  const int64_t pix = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  if (pix >= (int64_t)p.H * p.W) return;
  int i, j;
  fast_divmod((uint32_t)pix, (uint32_t)p.W, p.magic_w, i, j);
  float r, g, b;
  load_input(p, i, j, r, g, b);
  float A[NL][12];   // each level's up-sampled 3x4 map, kept for the way back (one gather of the low-res taps, not two)
#pragma unroll
  for (int l = 0; l < NL; l++) {
    if (l < p.nlevels) {
      float *P = p.lv[l].P + pix * 3;
      P[0] = r; P[1] = g; P[2] = b;
      upsample_affine(p.lv[l], p.H, p.W, i, j, A[l]);
      apply_affine(A[l], r, g, b);
    }
  }
  float v0 = v_out[pix * 3], v1 = v_out[pix * 3 + 1], v2 = v_out[pix * 3 + 2];
#pragma unroll
  for (int l = NL - 1; l >= 0; l--) {
    if (l < p.nlevels) {
      float *Q = p.lv[l].Q + pix * 3;
      Q[0] = v0; Q[1] = v1; Q[2] = v2;
      const float n0 = A[l][0] * v0 + A[l][4] * v1 + A[l][8] * v2;
      const float n1 = A[l][1] * v0 + A[l][5] * v1 + A[l][9] * v2;
      const float n2 = A[l][2] * v0 + A[l][6] * v1 + A[l][10] * v2;
      v0 = n0; v1 = n1; v2 = n2;
    }
  }
  v_in[pix * p.cs] = v0; v_in[pix * p.cs + 1] = v1; v_in[pix * p.cs + 2] = v2;
}

// destination indices of a bilinear up-sample (size `full` from `low`) whose taps touch source cell c:
// conservative [lo, hi] (every candidate is re-checked with resample_tap)
__device__ __forceinline__ void adjoint_range(int c, int full, float s /* = (float)full / (float)low */, int &lo, int &hi) {
  lo = (int)floorf(((float)c - 0.5f) * s - 0.5f);
  hi = (int)ceilf(((float)c + 1.5f) * s - 0.5f);
  lo = lo < 0 ? 0 : lo;
  hi = hi > full - 1 ? full - 1 : hi;
}

// ---- D1: x pass of the up-sampler adjoint: R[y, cx, :] = sum_x wx(x -> cx) * Q[y,x] (x) [P[y,x]; 1] -------
__global__ __launch_bounds__(kBgBlock) void ms_adjoint_x_kernel(MsParams p, LevelSched sc) {
  int local;
  const int l = sc.level[sched_find(sc, blockIdx.x, local)];
  const LevelDev &L = p.lv[l];
  const int64_t idx = (int64_t)local * kBgBlock + threadIdx.x;
  if (idx >= (int64_t)p.H * L.Wd) return;
  int y, cx;
  fast_divmod((uint32_t)idx, (uint32_t)L.Wd, L.magic_wd, y, cx);
  int xlo, xhi;
  adjoint_range(cx, p.W, L.dn_x, xlo, xhi);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.f;
#pragma unroll 4
  for (int x = xlo; x <= xhi; x++) {
    const Tap tx = resample_tap_s(x, p.W, L.Wd, L.up_x);
    const float w = (tx.i0 == cx ? 1.f - tx.w1 : 0.f) + (tx.i1 == cx ? tx.w1 : 0.f);
    const int64_t o = ((int64_t)y * p.W + x) * 3;
    const float p0 = L.P[o], p1 = L.P[o + 1], p2 = L.P[o + 2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float q = L.Q[o + r] * w;
      acc[r * 4 + 0] += q * p0; acc[r * 4 + 1] += q * p1; acc[r * 4 + 2] += q * p2; acc[r * 4 + 3] += q;
    }
  }
  float4 *d = reinterpret_cast<float4 *>(L.R + idx * 12);
  d[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  d[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  d[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
}


// ---- C+D1 fused: full-resolution backward with the x pass of the up-sampler adjoint done in LDS -------------
// A workgroup owns `stride` = 256 - 2*halo consecutive pixels of ONE image row and also evaluates `halo` pixels on
// either side, so that every low-res column anchored in its range finds all of its taps (a column's support is its
// anchor +- (scale + 2)) among the workgroup's own pixels: P and Q (6 floats per pixel and level) never travel
// through global memory, and the x reduction reads them from LDS.  Arithmetic and summation order are those of
// ms_apply_bwd_kernel + ms_adjoint_x_kernel (the halo pixels are recomputed, ~7 % extra work at factor 4).
template <int NL>
__global__ __launch_bounds__(kBgBlock) void ms_apply_bwd_x_kernel(MsParams p, const float *__restrict__ v_out,
                                                                 float *__restrict__ v_in, int halo, int nbx, int dbg) {
  // pixel k of the window sits at k + k / 32: the x pass reads with a stride of `factor` pixels between neighbouring threads, which on
  // the plain layout put every 8th (factor 4) thread of a 32-lane group on the same bank (5.7 M conflict cycles of 9.4 M LDS cycles)
  constexpr int kRow = kBgBlock + kBgBlock / 32;
  __shared__ float sP[NL][3][kRow], sQ[NL][3][kRow];
  const int tsk = (int)threadIdx.x + ((int)threadIdx.x >> 5);
  const int bid = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // one contiguous band of rows per XCD (L2 locality of the maps)
  const int y = bid / nbx, bx = bid - y * nbx;
  const int stride = kBgBlock - 2 * halo;
  const int own0 = bx * stride, own1 = min(p.W, own0 + stride);
  const int xs = own0 - halo;
  const int x = xs + (int)threadIdx.x;
  if (x >= 0 && x < p.W) {
    const int pix = row_major(y, p.W, x);
    const bool owner = x >= own0 && x < own1;
    float r, g, b;
    load_input(p, y, x, r, g, b);
    float A[NL][12];
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l < p.nlevels) {
        sP[l][0][tsk] = r; sP[l][1][tsk] = g; sP[l][2][tsk] = b;
        if (owner && p.lv[l].Wd == p.W && p.lv[l].Hd == p.H) {  // level without up-sampling: the low-res kernel reads P, Q
          float *P = p.lv[l].P + times3(pix);
          P[0] = r; P[1] = g; P[2] = b;
        }
        upsample_affine(p.lv[l], p.H, p.W, y, x, A[l]);
        apply_affine(A[l], r, g, b);
      }
    }
    float v0 = v_out[times3(pix)], v1 = v_out[times3(pix) + 1], v2 = v_out[times3(pix) + 2];
#pragma unroll
    for (int l = NL - 1; l >= 0; l--) {
      if (l < p.nlevels) {
        sQ[l][0][tsk] = v0; sQ[l][1][tsk] = v1; sQ[l][2][tsk] = v2;
        if (owner && p.lv[l].Wd == p.W && p.lv[l].Hd == p.H) {
          float *Q = p.lv[l].Q + times3(pix);
          Q[0] = v0; Q[1] = v1; Q[2] = v2;
        }
        const float n0 = A[l][0] * v0 + A[l][4] * v1 + A[l][8] * v2;
        const float n1 = A[l][1] * v0 + A[l][5] * v1 + A[l][9] * v2;
        const float n2 = A[l][2] * v0 + A[l][6] * v1 + A[l][10] * v2;
        v0 = n0; v1 = n1; v2 = n2;
      }
    }
    if (owner) {
      const int oc = p.cs == 4 ? pix << 2 : times3(pix);
      v_in[oc] = v0; v_in[oc + 1] = v1; v_in[oc + 2] = v2;
    }
  }
  __syncthreads();
  // x pass: candidate columns of every up-sampled level, one item per thread and level.  (One flat item list over all levels --
  // every thread busy once instead of the first 64 threads three times -- was measured slower, 211 -> 232 us for the bilateral
  // backward: the per-item level lookup and the mixed loop lengths inside a wave cost more than the idle lanes.)
#pragma unroll
  for (int l = 0; l < NL; l++) {
    if (l >= p.nlevels) break;
    const LevelDev &L = p.lv[l];
    if (L.Wd == p.W && L.Hd == p.H) continue;
    const float sc = L.dn_x;
    const int cx0 = max(0, (int)((float)own0 * L.up_x) - 2);      // conservative window: every candidate's anchor is re-checked
    const int ncand = (int)((float)stride * L.up_x) + 6;
    for (int t = threadIdx.x; t < ncand; t += kBgBlock) {
      const int cx = cx0 + t;
      if (cx >= L.Wd) continue;
      const int anchor = min(p.W - 1, (int)(((float)cx + 0.5f) * sc));
      if (anchor < own0 || anchor >= own0 + stride) continue;   // owned by a neighbour
      int xlo, xhi;
      // power-of-two factor f dividing the image, interior column: its taps are the 2 f pixels f cx - f/2 .. f cx + 3f/2 - 1 with the
      // tent weights 1 - |src - cx| (exact binary fractions: the same values, in the same order, as the general form below, whose
      // conservative window also visits 2-4 pixels of weight zero and re-derives both taps of every pixel)
      const bool tent = L.dn_shift > 0 && cx >= 1 && cx <= L.Wd - 2 && !(dbg & 4096);
      if (tent) {
        const int f = 1 << L.dn_shift;
        xlo = f * cx - (f >> 1); xhi = xlo + 2 * f - 1;
      } else {
        adjoint_range(cx, p.W, L.dn_x, xlo, xhi);
      }
      float acc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = 0.f;
      for (int xx = xlo; xx <= xhi; xx++) {
        float w;
        if (tent) {
          w = 1.f - fabsf((L.up_x * ((float)xx + 0.5f) - 0.5f) - (float)cx);
        } else {
          const Tap tx = resample_tap_s(xx, p.W, L.Wd, L.up_x);
          w = (tx.i0 == cx ? 1.f - tx.w1 : 0.f) + (tx.i1 == cx ? tx.w1 : 0.f);
        }
        const int k0 = xx - xs, k = k0 + (k0 >> 5);   // inside the window by construction (halo >= scale + 2)
        const float p0 = sP[l][0][k], p1 = sP[l][1][k], p2 = sP[l][2][k];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const float q = sQ[l][r][k] * w;
          acc[r * 4 + 0] += q * p0; acc[r * 4 + 1] += q * p1; acc[r * 4 + 2] += q * p2; acc[r * 4 + 3] += q;
        }
      }
      float4 *d = reinterpret_cast<float4 *>(L.R) + times3(row_major(y, L.Wd, cx));
      d[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      d[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      d[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
    }
  }
}

// wave64 sum leaving the result in every lane (used only on wave-uniform-address grid updates)
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Scatter the slice gradient of a wave's 64 samples into the grid-gradient accumulator (LDS or global).
// Neighbouring samples fall into the same grid cell (coarse grids: the whole wave), so per-lane atomics
// would serialise 20-64 ways on one address.  Instead the wave walks its DISTINCT cells (typically 1-6):
// for each cell and each of its 8 trilinear corners the 12 channel contributions of the member lanes are
// summed with the 16-value transpose-reduce (35 VALU ops) and committed by 12 lanes to 12 different
// addresses -- conflict-free by construction.
__device__ __forceinline__ void slice_grid_scatter(float *acc, const Cell &c, int gx, int gy, int gl, float scale,
                                                   const float *va, bool active, int nch = 12) {
  const int lane = threadIdx.x & (kWave - 1);
  const int plane = gy * gx, vol = gl * plane;
  const int key = active ? (c.z0 * gy + c.y0) * gx + c.x0 : -1;
  const int slot = butterfly_slot(lane);
  const bool committer = ((lane & 3) == 0) && slot < nch;
  unsigned long long remaining = __ballot(active);
  while (remaining) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int k = __builtin_amdgcn_readlane(key, leader);
    const bool member = active && key == k;
    remaining &= ~__ballot(member);
    // the cell's corner indices are the same for every member: take the leader's
    const int z0 = __builtin_amdgcn_readlane(c.z0, leader), z1 = __builtin_amdgcn_readlane(c.z1, leader);
    const int y0 = __builtin_amdgcn_readlane(c.y0, leader), y1 = __builtin_amdgcn_readlane(c.y1, leader);
    const int x0 = __builtin_amdgcn_readlane(c.x0, leader), x1 = __builtin_amdgcn_readlane(c.x1, leader);
#pragma unroll 1
    for (int q = 0; q < 8; q++) {
      const int base = ((q & 4) ? z1 : z0) * plane + ((q & 2) ? y1 : y0) * gx + ((q & 1) ? x1 : x0);
      const float w = member ? ((q & 4) ? c.fz : 1.f - c.fz) * ((q & 2) ? c.fy : 1.f - c.fy) * ((q & 1) ? c.fx : 1.f - c.fx) * scale : 0.f;
      float v[16];
#pragma unroll
      for (int ch = 0; ch < 12; ch++) v[ch] = w * va[ch];
      v[12] = v[13] = v[14] = v[15] = 0.f;
      const float tot = butterfly_sum16(v, lane);
      if (committer && tot != 0.f) atomicAdd(acc + base + slot * vol, tot);
    }
  }
}

// ---- D2: per-level low-resolution backward (y pass, slice vjp into the grid, guidance route) -----------
// kLds: the level's grid gradient (n_avg * 12*gl*gy*gx floats) fits the workgroup's LDS accumulator.
// Persistent workgroups (grid-stride over the low-res pixels): each accumulates the whole level's grid
// gradient in LDS and writes ONE partial grid; a tiny second kernel sums the partials.  (Flushing
// with atomics from thousands of short workgroups serialises on the few-thousand grid addresses in L2:
// that was 3x the cost of everything else in this kernel.)  Deterministic for the LDS path.
template <bool kLds, int kUnroll = 4>
__global__ __launch_bounds__(kBgBlock) void ms_lowres_bwd_kernel(MsParams p, LevelSched sc, float *__restrict__ v_in,
                                                                float *__restrict__ partials, int dbg, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds_acc[];
  int local;
  const int k_entry = sched_find(sc, blockIdx.x, local);
  const int l = sc.level[k_entry];
  const LevelDev &L = p.lv[l];
  const int vol = L.gl * L.gy * L.gx, gsz = 12 * vol;
  const int gtot = gsz * L.n_avg;
  // small grids: a cell-major copy of the grid itself sits behind the accumulator (the guidance route samples it per pixel)
  const bool cells = kLds && 2 * gtot <= lds_floats;
  float *lds_cells = lds_acc + gtot;
  if (kLds) {
    for (int e = threadIdx.x; e < gtot; e += kBgBlock) lds_acc[e] = 0.f;
    if (cells) {   // (image, channel) outermost: no integer division per element
      for (int nc = 0; nc < 12 * L.n_avg; nc++) {
        const int n = nc / 12, ch = nc - n * 12;
        for (int cell = threadIdx.x; cell < vol; cell += kBgBlock) lds_cells[(n * vol + cell) * 12 + ch] = L.grid[(int64_t)nc * vol + cell];
      }
    }
    __syncthreads();
  }
  float *acc = kLds ? lds_acc : L.v_grid;
  const int64_t n_low = (int64_t)L.Hd * L.Wd;
  // a workgroup walks ONE contiguous range of chunks (and the workgroups of an XCD neighbouring ranges): consecutive low-res rows
  // share most of the rows of R / of the image they read
  const int64_t n_chunks = (n_low + kBgBlock - 1) / kBgBlock, per_wg = (n_chunks + sc.nblk[k_entry] - 1) / sc.nblk[k_entry];
  const int64_t c_first = (int64_t)xcd_contiguous(local, sc.nblk[k_entry]) * per_wg;
  const int64_t c_last = c_first + per_wg < n_chunks ? c_first + per_wg : n_chunks;
  for (int64_t chunk = c_first; chunk < c_last; chunk++) {
  const int64_t idx = chunk * kBgBlock + threadIdx.x;
  const bool active = idx < n_low;
  int i = 0, j = 0;
  if (active) fast_divmod((uint32_t)idx, (uint32_t)L.Wd, L.magic_wd, i, j);
  float va[12];
#pragma unroll
  for (int k = 0; k < 12; k++) va[k] = 0.f;
  if (active) {
    if (L.Hd == p.H && L.Wd == p.W) {
      const float *P = L.P + idx * 3, *Q = L.Q + idx * 3;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        va[r * 4 + 0] = Q[r] * P[0]; va[r * 4 + 1] = Q[r] * P[1]; va[r * 4 + 2] = Q[r] * P[2]; va[r * 4 + 3] = Q[r];
      }
    } else if (dbg & 1) {
      for (int k = 0; k < 12; k++) va[k] = 1.f;
    } else {  // y pass of the up-sampler adjoint over the x-reduced rows
      int ylo, yhi;
      // (power-of-two factor, interior row: exactly the 2 f rows with the tent weights, see the x pass of ms_apply_bwd_x_kernel)
      const bool tent = L.dn_shift > 0 && i >= 1 && i <= L.Hd - 2 && !(dbg & 4096);
      if (tent) {
        const int f = 1 << L.dn_shift;
        ylo = f * i - (f >> 1); yhi = ylo + 2 * f - 1;
      } else {
        adjoint_range(i, p.H, L.dn_y, ylo, yhi);
      }
      // no early-out on zero weights: the few extra rows of the conservative window cost less than the
      // serialised load -> test -> load chain they would otherwise create (the loop is latency-bound)
#pragma unroll kUnroll
      for (int y = ylo; y <= yhi; y++) {
        float w;
        if (tent) {
          w = 1.f - fabsf((L.up_y * ((float)y + 0.5f) - 0.5f) - (float)i);
        } else {
          const Tap ty = resample_tap_s(y, p.H, L.Hd, L.up_y);
          w = (ty.i0 == i ? 1.f - ty.w1 : 0.f) + (ty.i1 == i ? ty.w1 : 0.f);
        }
        const float4 *sv = reinterpret_cast<const float4 *>(L.R) + times3(row_major(y, L.Wd, j));
        const float4 a = sv[0], b = sv[1], c = sv[2];
        va[0] += w * a.x; va[1] += w * a.y; va[2] += w * a.z; va[3] += w * a.w;
        va[4] += w * b.x; va[5] += w * b.y; va[6] += w * b.z; va[7] += w * b.w;
        va[8] += w * c.x; va[9] += w * c.y; va[10] += w * c.z; va[11] += w * c.w;
      }
    }
  }
  // slice backward
  const Tap ty = resample_tap_s(i, L.Hd, p.H, L.dn_y), tx = resample_tap_s(j, L.Wd, p.W, L.dn_x);
  float r = 0.f, g = 0.f, b = 0.f;
  if (active) lowres_colour(p, ty, tx, r, g, b);
  const Cell c = slice_cell(linspace01_s(j, L.Wd, L.lin_x), linspace01_s(i, L.Hd, L.lin_y), rgb2gray(r, g, b), L.gx, L.gy, L.gl);
  const float inv_n = 1.f / (float)L.n_avg;
  float v_iz = 0.f;
  for (int n = 0; n < L.n_avg; n++) {
    if (L.v_grid && !(dbg & 2)) slice_grid_scatter(acc + n * gsz, c, L.gx, L.gy, L.gl, inv_n, va, active);
    if (active && c.z_interior && !(dbg & 4)) {
      float a12[12], dz[12];
      if (cells) slice_dz_cells(reinterpret_cast<const float4 *>(lds_cells) + (int64_t)n * vol * 3, L.gx, L.gy, c, dz);
      else slice_sample(L.grid + (int64_t)n * gsz, L.gx, L.gy, L.gl, c, a12, dz);
#pragma unroll
      for (int ch = 0; ch < 12; ch++) v_iz += va[ch] * dz[ch] * inv_n;
    }
  }
  // guidance route: d(loss)/d(gray) of this low-res pixel; the full-resolution kernel E gathers it through the
  // adjoint of the bilinear down-sampler (no atomics on the image)
  if (active) L.vg[idx] = c.z_interior ? v_iz * (float)(L.gl - 1) : 0.f;
  }  // grid-stride loop
  if (kLds && L.v_grid) {
    __syncthreads();
    float *dst = partials + sc.part_off[k_entry] + (int64_t)local * gtot;
    for (int e = threadIdx.x; e < gtot; e += kBgBlock) dst[e] = lds_acc[e];
  }
}

// v_grid[e] += sum_b partials[b][e].  Workgroup = 32 grid entries x 8 partial-lanes: coalesced 128-byte
// rows, 8 independent loads in flight per thread, fixed summation order (deterministic).
struct PartialsJob {   // the reduction of the low-res backward's per-workgroup partial grids, riding on the launch behind it
  LevelSched sc, red;
  const float *partials;
  int pix_blocks;      // workgroups [pix_blocks, pix_blocks + red.blk_off[red.n]) of the host launch run it (0 extra: nothing to do)
};

__device__ __forceinline__ void grid_partials_reduce_block(const MsParams &p, const LevelSched &sc, const LevelSched &red,
                                                           const float *__restrict__ partials_all, int bid, float (*sred)[33]) {
  int local;
  const int k = sched_find(red, bid, local);  // `red` mirrors `sc` entry by entry, with its own block ranges
  const LevelDev &L = p.lv[sc.level[k]];
  const int gtot = 12 * L.gl * L.gy * L.gx * L.n_avg, nparts = sc.nblk[k];
  const float *partials = partials_all + sc.part_off[k];
  const int ex = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int e = local * 32 + ex;
  float s = 0.f;
  if (e < gtot) {
    int b = py;
    for (; b + 56 < nparts; b += 64) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = partials[(int64_t)(b + 8 * u) * gtot + e];
#pragma unroll
      for (int u = 0; u < 8; u++) s += t[u];
    }
    for (; b < nparts; b += 8) s += partials[(int64_t)b * gtot + e];
  }
  sred[py][ex] = s;
  __syncthreads();
  if (py == 0 && e < gtot) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) t += sred[q][ex];
    atomicAdd(L.v_grid + e, t);   // (one add per entry and launch; atomic so that another view's backward may run next to this one)
  }
}


// ---- E: guidance route gather + clamp / sky blend backward (in place on v_in) -----------------------
// v_in already holds the direct-route gradient.  Every level adds  gray_weights * sum_{low-res pixels whose
// down-sample taps include this pixel} w * v_gray  (adjoint of the bilinear down-sampler, gather form), then the
// clamp(max=1) + sky blend in front of the transform is back-propagated.
template <int NL>
__global__ __launch_bounds__(kBgBlock) void ms_guidance_blend_bwd_kernel(MsParams p, float *__restrict__ v_in,
                                                                        float *__restrict__ v_alpha, float *__restrict__ v_sky, int dbg,
                                                                        PartialsJob pj) {
  if ((int)blockIdx.x >= pj.pix_blocks) {   // (both only need the low-res kernel's results: one launch instead of two)
    __shared__ float sred[8][33];
    grid_partials_reduce_block(p, pj.sc, pj.red, pj.partials, (int)blockIdx.x - pj.pix_blocks, sred);
    return;
  }
  const int pix = (int)blockIdx.x * kBgBlock + (int)threadIdx.x;
  if (pix >= p.H * p.W) return;
  int y, x;
  fast_divmod((uint32_t)pix, (uint32_t)p.W, p.magic_w, y, x);
  float vg = 0.f;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    if (l >= p.nlevels) break;
    const LevelDev &L = p.lv[l];
    if (L.Hd == p.H && L.Wd == p.W) { vg += L.vg[pix]; continue; }
    if (L.dn_shift > 0 && !(dbg & 2048)) {
      // power-of-two factor f dividing the image: low-res pixel (i, j) was sampled at f (i + 1/2) - 1/2, i.e. from the two central
      // rows / columns of its f x f block with weights 1/2, 1/2 -- a pixel receives 1/4 of ONE low-res value or nothing (the same
      // single term, the same product, as the general gather below: bit-identical, ~10 instructions instead of ~200)
      const int f = 1 << L.dn_shift, h = f >> 1;
      const int fy = y & (f - 1), fx = x & (f - 1);
      if ((fy == h - 1 || fy == h) && (fx == h - 1 || fx == h))
        vg += (0.5f * 0.5f) * L.vg[row_major(y >> L.dn_shift, L.Wd, x >> L.dn_shift)];
      continue;
    }
    int ilo, ihi, jlo, jhi;
    adjoint_range(y, L.Hd, L.up_y, ilo, ihi);
    adjoint_range(x, L.Wd, L.up_x, jlo, jhi);
    if (ihi - ilo < 3 && jhi - jlo < 3) {
      // the usual case (factor >= 2: at most three candidates per axis): weights first, then ALL loads, then the sum in the order
      // of the general loops below -- those wait out one memory latency per candidate, the dominant cost of this kernel
      float wy[3], wx[3], val[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const int i = ilo + a, j = jlo + a;
        const Tap ty = resample_tap_s(i <= ihi ? i : ihi, L.Hd, p.H, L.dn_y), tx = resample_tap_s(j <= jhi ? j : jhi, L.Wd, p.W, L.dn_x);
        wy[a] = i <= ihi ? (ty.i0 == y ? 1.f - ty.w1 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f) : 0.f;
        wx[a] = j <= jhi ? (tx.i0 == x ? 1.f - tx.w1 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f) : 0.f;
      }
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
          val[a][b] = (wy[a] != 0.f && wx[b] != 0.f) ? L.vg[(int64_t)(ilo + a) * L.Wd + (jlo + b)] : 0.f;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
          if (wy[a] != 0.f && wx[b] != 0.f) vg += wy[a] * wx[b] * val[a][b];
    } else {
      for (int i = ilo; i <= ihi; i++) {
        const Tap ty = resample_tap_s(i, L.Hd, p.H, L.dn_y);
        const float wy = (ty.i0 == y ? 1.f - ty.w1 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f);
        if (wy == 0.f) continue;
        for (int j = jlo; j <= jhi; j++) {
          const Tap tx = resample_tap_s(j, L.Wd, p.W, L.dn_x);
          const float wx = (tx.i0 == x ? 1.f - tx.w1 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f);
          if (wx != 0.f) vg += wy * wx * L.vg[(int64_t)i * L.Wd + j];
        }
      }
    }
  }
  const int cs = p.cs;
  const int oc = cs == 4 ? pix << 2 : times3(pix), p3 = times3(pix);
  float v[3] = {v_in[oc] + vg * kGrayR, v_in[oc + 1] + vg * kGrayG, v_in[oc + 2] + vg * kGrayB};
  float va = 0.f;
  if (p.sky) {
    const float k = 1.f - p.alpha[pix];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      va -= v[c] * p.sky[p3 + c];
      if (v_sky) v_sky[p3 + c] = v[c] * k;
      v[c] = p.rgb[oc + c] <= 1.f ? v[c] : 0.f;  // torch.clamp(max=1) passes gradient at x <= 1
    }
  }
  v_in[oc] = v[0]; v_in[oc + 1] = v[1]; v_in[oc + 2] = v[2];
  if (cs == 4) {  // RGB+ED form: depth = D / clamp(alpha, min=1e-10); plus the caller's own alpha gradient
    const float a = p.alpha[pix], ac = fmaxf(a, 1e-10f);
    const float vd = p.v_depth ? p.v_depth[pix] : 0.f;
    v_in[(pix << 2) + 3] = vd / ac;
    if (p.v_alpha_in) va += p.v_alpha_in[pix];
    if (a >= 1e-10f) va -= p.rgb[(pix << 2) + 3] * vd / (ac * ac);  // clamp(min) passes the gradient where alpha >= 1e-10
    if (v_alpha) v_alpha[pix] = va;
  } else if (p.sky && v_alpha) {
    v_alpha[pix] = va;
  }
}

// ---- generic point slice (BilateralGrid.forward on arbitrary points) ------------------------------
__global__ __launch_bounds__(kBgBlock) void slice_fwd_kernel(int64_t P, const float *__restrict__ grid, int gx, int gy, int gl,
                                                            const float *__restrict__ xy, const float *__restrict__ rgb,
                                                            float *__restrict__ affine) {
  const int64_t i = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  if (i >= P) return;
  const Cell c = slice_cell(xy[i * 2], xy[i * 2 + 1], rgb2gray(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]), gx, gy, gl);
  float a[12];
  slice_sample(grid, gx, gy, gl, c, a, nullptr);
#pragma unroll
  for (int k = 0; k < 12; k++) affine[i * 12 + k] = a[k];
}

__global__ __launch_bounds__(kBgBlock) void slice_bwd_kernel(int64_t P, const float *__restrict__ grid, int gx, int gy, int gl,
                                                            const float *__restrict__ xy, const float *__restrict__ rgb,
                                                            const float *__restrict__ v_affine, float *__restrict__ v_grid,
                                                            float *__restrict__ v_rgb) {
  const int64_t i = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  const bool active = i < P;
  const int64_t ii = active ? i : 0;
  const Cell c = slice_cell(xy[ii * 2], xy[ii * 2 + 1], rgb2gray(rgb[ii * 3], rgb[ii * 3 + 1], rgb[ii * 3 + 2]), gx, gy, gl);
  float va[12];
#pragma unroll
  for (int k = 0; k < 12; k++) va[k] = active ? v_affine[ii * 12 + k] : 0.f;
  if (v_grid) slice_grid_scatter(v_grid, c, gx, gy, gl, 1.f, va, active);
  if (active && v_rgb) {
    float vg = 0.f;
    if (c.z_interior) {
      float a12[12], dz[12];
      slice_sample(grid, gx, gy, gl, c, a12, dz);
      float v_iz = 0.f;
#pragma unroll
      for (int ch = 0; ch < 12; ch++) v_iz += va[ch] * dz[ch];
      vg = v_iz * (float)(gl - 1);
    }
    v_rgb[i * 3] = vg * kGrayR; v_rgb[i * 3 + 1] = vg * kGrayG; v_rgb[i * 3 + 2] = vg * kGrayB;
  }
}

// ---- point slice of a FEATURE grid with any channel count (NeuralBilateralGrid.forward, lib_bilagrid.py:370-461) ---------
__global__ __launch_bounds__(kBgBlock) void slice_feat_fwd_kernel(int64_t P, int NC, const float *__restrict__ grid, int gx, int gy,
                                                                 int gl, const float *__restrict__ xy, const float *__restrict__ rgb,
                                                                 float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  if (i >= P) return;
  const Cell c = slice_cell(xy[i * 2], xy[i * 2 + 1], rgb2gray(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]), gx, gy, gl);
  const int vol = gl * gy * gx;
  for (int c0 = 0; c0 < NC; c0 += 12) {
    const int nch = NC - c0 < 12 ? NC - c0 : 12;
    float a[12];
    slice_sample_n(grid + (int64_t)c0 * vol, gx, gy, gl, c, nch, a, nullptr);
    for (int k = 0; k < nch; k++) out[i * NC + c0 + k] = a[k];
  }
}

__global__ __launch_bounds__(kBgBlock) void slice_feat_bwd_kernel(int64_t P, int NC, const float *__restrict__ grid, int gx, int gy,
                                                                 int gl, const float *__restrict__ xy, const float *__restrict__ rgb,
                                                                 const float *__restrict__ v_out, float *__restrict__ v_grid,
                                                                 float *__restrict__ v_rgb) {
  const int64_t i = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  const bool active = i < P;
  const int64_t ii = active ? i : 0;
  const Cell c = slice_cell(xy[ii * 2], xy[ii * 2 + 1], rgb2gray(rgb[ii * 3], rgb[ii * 3 + 1], rgb[ii * 3 + 2]), gx, gy, gl);
  const int vol = gl * gy * gx;
  float v_iz = 0.f;
  for (int c0 = 0; c0 < NC; c0 += 12) {   // uniform trip count: the scatter is a wave-collective
    const int nch = NC - c0 < 12 ? NC - c0 : 12;
    float va[12];
#pragma unroll
    for (int k = 0; k < 12; k++) va[k] = (active && k < nch) ? v_out[ii * NC + c0 + k] : 0.f;
    if (v_grid) slice_grid_scatter(v_grid + (int64_t)c0 * vol, c, gx, gy, gl, 1.f, va, active, nch);
    if (active && v_rgb && c.z_interior) {
      float a12[12], dz[12];
      slice_sample_n(grid + (int64_t)c0 * vol, gx, gy, gl, c, nch, a12, dz);
      for (int k = 0; k < nch; k++) v_iz += va[k] * dz[k];
    }
  }
  if (active && v_rgb) {
    const float vg = c.z_interior ? v_iz * (float)(gl - 1) : 0.f;
    v_rgb[i * 3] = vg * kGrayR; v_rgb[i * 3 + 1] = vg * kGrayG; v_rgb[i * 3 + 2] = vg * kGrayB;
  }
}

// ---- feature slice of a whole IMAGE (xy = the pixel grid, linspace(0,1) both ways: what the neural modules slice at,
// models/modules.py:643-650, 728-760) ------------------------------------------------------------------------------------------
// A pixel row touches two grid rows (y0, y1) only: that band of the grid -- [NC][gl][2][gx] floats, 24 KB for the shipped
// 16x16x8 / 24-feature grid whose full 196 KB does not fit the LDS -- is staged per workgroup, the per-pixel gathers and, backward,
// the scatter (wave-level butterfly pre-reduction + LDS atomics, slice_grid_scatter) run against the LDS band through the SAME
// sampling / scatter functions with a band-local cell (gy = 2).  A workgroup owns a block of consecutive rows and flushes its
// band accumulator with one atomic per non-zero entry when the band changes: ~2 M global atomics per image instead of the
// point kernel's ~25 M same-address ones (1.77 ms -> see profiles/).
constexpr int kBandMaxFloats = 6 * 1024;   // values (forward) or values + accumulator (backward): <= 48 KB per workgroup

__device__ __forceinline__ void band_load(float *__restrict__ vals, const float *__restrict__ grid, int NC, int gx, int gy, int gl,
                                          int y0, int y1) {
  const int n = NC * gl * 2 * gx;
  for (int e = threadIdx.x; e < n; e += kBgBlock) {
    const int x = e % gx, yb = (e / gx) & 1, cz = e / (2 * gx);
    vals[e] = grid[((int64_t)cz * gy + (yb ? y1 : y0)) * gx + x];
  }
}

__device__ __forceinline__ Cell band_cell(int x, int W, float lin_x, float y01, float r, float g, float b, int gx, int gy, int gl) {
  Cell c = slice_cell(linspace01_s(x, W, lin_x), y01, rgb2gray(r, g, b), gx, gy, gl);
  c.y1 = c.y1 != c.y0 ? 1 : 0;   // band-local rows
  c.y0 = 0;
  return c;
}

__global__ __launch_bounds__(kBgBlock) void slice_feat_image_fwd_kernel(int H, int W, int NC, const float *__restrict__ grid, int gx,
                                                                       int gy, int gl, float lin_x, float lin_y, int rows_per_wg,
                                                                       const float *__restrict__ rgb, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float band[];
  const int volb = gl * 2 * gx;
  int cur = -1;
  const int r0 = blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;
  for (int y = r0; y < r1; y++) {
    const float y01 = linspace01_s(y, H, lin_y);
    const int y0 = (int)floorf(grid_coord(y01, gy)), y1 = y0 + 1 < gy ? y0 + 1 : gy - 1;
    if (y0 != cur) {
      __syncthreads();
      band_load(band, grid, NC, gx, gy, gl, y0, y1);
      __syncthreads();
      cur = y0;
    }
    for (int x = threadIdx.x; x < W; x += kBgBlock) {
      const int64_t i = (int64_t)y * W + x;
      const Cell c = band_cell(x, W, lin_x, y01, rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], gx, gy, gl);
      for (int c0 = 0; c0 < NC; c0 += 12) {
        const int nch = NC - c0 < 12 ? NC - c0 : 12;
        float a[12];
        slice_sample_n(band + c0 * volb, gx, 2, gl, c, nch, a, nullptr);
        float *dst = out + i * NC + c0;
        if ((NC & 3) == 0) {   // rows and chunks are 16-byte aligned
          for (int k = 0; k + 4 <= nch; k += 4) *reinterpret_cast<float4 *>(dst + k) = make_float4(a[k], a[k + 1], a[k + 2], a[k + 3]);
        } else {
          for (int k = 0; k < nch; k++) dst[k] = a[k];
        }
      }
    }
  }
}

__global__ __launch_bounds__(kBgBlock) void slice_feat_image_bwd_kernel(int H, int W, int NC, const float *__restrict__ grid, int gx,
                                                                       int gy, int gl, float lin_x, float lin_y, int rows_per_wg,
                                                                       const float *__restrict__ rgb, const float *__restrict__ v_out,
                                                                       float *__restrict__ v_grid, float *__restrict__ v_rgb) {
  extern __shared__ __attribute__((aligned(16))) float band[];
  const int volb = gl * 2 * gx, nband = NC * volb;
  float *vals = band, *acc = band + nband;
  int cur = -1, cur_y1 = 0;
  auto flush = [&]() {   // one atomic per touched entry of the band; leaves the accumulator zeroed
    for (int e = threadIdx.x; e < nband; e += kBgBlock) {
      const float v = acc[e];
      if (v != 0.f) {
        const int x = e % gx, yb = (e / gx) & 1, cz = e / (2 * gx);
        atomicAdd(v_grid + ((int64_t)cz * gy + (yb ? cur_y1 : cur)) * gx + x, v);
        acc[e] = 0.f;
      }
    }
  };
  for (int e = threadIdx.x; e < nband; e += kBgBlock) acc[e] = 0.f;
  const int r0 = blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;
  const int w_pad = (W + kBgBlock - 1) / kBgBlock * kBgBlock;   // the scatter is a wave collective: uniform trip count
  for (int y = r0; y < r1; y++) {
    const float y01 = linspace01_s(y, H, lin_y);
    const int y0 = (int)floorf(grid_coord(y01, gy)), y1 = y0 + 1 < gy ? y0 + 1 : gy - 1;
    if (y0 != cur) {
      __syncthreads();
      if (cur >= 0 && v_grid) flush();
      band_load(vals, grid, NC, gx, gy, gl, y0, y1);
      cur = y0; cur_y1 = y1;
      __syncthreads();
    }
    for (int x = threadIdx.x; x < w_pad; x += kBgBlock) {
      const bool active = x < W;
      const int64_t i = (int64_t)y * W + (active ? x : 0);
      const Cell c = band_cell(active ? x : 0, W, lin_x, y01, rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], gx, gy, gl);
      float v_iz = 0.f;
      for (int c0 = 0; c0 < NC; c0 += 12) {
        const int nch = NC - c0 < 12 ? NC - c0 : 12;
        float va[12];
#pragma unroll
        for (int k = 0; k < 12; k++) va[k] = (active && k < nch) ? v_out[i * NC + c0 + k] : 0.f;
        if (v_grid) slice_grid_scatter(acc + c0 * volb, c, gx, 2, gl, 1.f, va, active, nch);
        if (active && v_rgb && c.z_interior) {
          float a12[12], dz[12];
          slice_sample_n(vals + c0 * volb, gx, 2, gl, c, nch, a12, dz);
          for (int k = 0; k < nch; k++) v_iz += va[k] * dz[k];
        }
      }
      if (active && v_rgb) {
        const float vg = c.z_interior ? v_iz * (float)(gl - 1) : 0.f;
        v_rgb[i * 3] = vg * kGrayR; v_rgb[i * 3 + 1] = vg * kGrayG; v_rgb[i * 3 + 2] = vg * kGrayB;
      }
    }
  }
  __syncthreads();
  if (cur >= 0 && v_grid) flush();
}

// ---- TV regulariser ----------------------------------------------------------------------------
__global__ __launch_bounds__(kBgBlock) void tv_fwd_kernel(int64_t total, int gx, int gy, int gl, const float *__restrict__ x,
                                                         float scale_l, float scale_y, float scale_x,
                                                         float *__restrict__ tv_out) {
  __shared__ float red[kBgBlock / kWave];
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * kBgBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBgBlock) {
    const int ix = (int)(e % gx), iy = (int)((e / gx) % gy), il = (int)((e / ((int64_t)gx * gy)) % gl);
    const float v = x[e];
    if (ix > 0) { const float d = v - x[e - 1]; acc += d * d * scale_x; }
    if (iy > 0) { const float d = v - x[e - gx]; acc += d * d * scale_y; }
    if (il > 0) { const float d = v - x[e - (int64_t)gx * gy]; acc += d * d * scale_l; }
  }
  acc = wave_sum_all(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kBgBlock / kWave; w++) s += red[w];
    atomicAdd(tv_out, s);
  }
}

__global__ __launch_bounds__(kBgBlock) void tv_bwd_kernel(int64_t total, int gx, int gy, int gl, const float *__restrict__ x,
                                                         float scale_l, float scale_y, float scale_x,
                                                         const float *__restrict__ v_tv, float *__restrict__ v_x) {
  const int64_t e = (int64_t)blockIdx.x * kBgBlock + threadIdx.x;
  if (e >= total) return;
  const int ix = (int)(e % gx), iy = (int)((e / gx) % gy), il = (int)((e / ((int64_t)gx * gy)) % gl);
  const int64_t sl = (int64_t)gx * gy;
  const float v = x[e];
  float g = 0.f;
  if (ix > 0) g += 2.f * (v - x[e - 1]) * scale_x;
  if (ix < gx - 1) g -= 2.f * (x[e + 1] - v) * scale_x;
  if (iy > 0) g += 2.f * (v - x[e - gx]) * scale_y;
  if (iy < gy - 1) g -= 2.f * (x[e + gx] - v) * scale_y;
  if (il > 0) g += 2.f * (v - x[e - sl]) * scale_l;
  if (il < gl - 1) g -= 2.f * (x[e + sl] - v) * scale_l;
  atomicAdd(v_x + e, g * v_tv[0]);
}


__global__ __launch_bounds__(kBgBlock) void tv_ms_fwd_kernel(TvLevels L, float *__restrict__ tv_out) {
  __shared__ float red[kBgBlock / kWave];
  int k = 0;
  while (k + 1 < L.n && (int)blockIdx.x >= L.blk_off[k + 1]) k++;
  const int nb = L.blk_off[k + 1] - L.blk_off[k], local = (int)blockIdx.x - L.blk_off[k];
  const int gx = L.gx[k], gy = L.gy[k], gl = L.gl[k];
  const float *x = L.x[k];
  float acc = 0.f;
  for (int64_t e = (int64_t)local * kBgBlock + threadIdx.x; e < L.total[k]; e += (int64_t)nb * kBgBlock) {
    const int ix = (int)(e % gx), iy = (int)((e / gx) % gy), il = (int)((e / ((int64_t)gx * gy)) % gl);
    const float v = x[e];
    if (ix > 0) { const float d = v - x[e - 1]; acc += d * d * L.sx[k]; }
    if (iy > 0) { const float d = v - x[e - gx]; acc += d * d * L.sy[k]; }
    if (il > 0) { const float d = v - x[e - (int64_t)gx * gy]; acc += d * d * L.sl[k]; }
  }
  acc = wave_sum_all(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBgBlock / kWave; w++) t += red[w];
    atomicAdd(tv_out, t);
  }
}

__global__ __launch_bounds__(kBgBlock) void tv_ms_bwd_kernel(TvLevels L, const float *__restrict__ v_tv) {
  int k = 0;
  while (k + 1 < L.n && (int)blockIdx.x >= L.blk_off[k + 1]) k++;
  const int64_t e = (int64_t)((int)blockIdx.x - L.blk_off[k]) * kBgBlock + threadIdx.x;
  if (e >= L.total[k]) return;
  const int gx = L.gx[k], gy = L.gy[k], gl = L.gl[k];
  const float *x = L.x[k];
  const int ix = (int)(e % gx), iy = (int)((e / gx) % gy), il = (int)((e / ((int64_t)gx * gy)) % gl);
  const int64_t sl = (int64_t)gx * gy;
  const float v = x[e];
  float g = 0.f;
  if (ix > 0) g += 2.f * (v - x[e - 1]) * L.sx[k];
  if (ix < gx - 1) g -= 2.f * (x[e + 1] - v) * L.sx[k];
  if (iy > 0) g += 2.f * (v - x[e - gx]) * L.sy[k];
  if (iy < gy - 1) g -= 2.f * (x[e + gx] - v) * L.sy[k];
  if (il > 0) g += 2.f * (v - x[e - sl]) * L.sl[k];
  if (il < gl - 1) g -= 2.f * (x[e + sl] - v) * L.sl[k];
  atomicAdd(L.v_x[k] + e, g * v_tv[0]);   // (concurrent views add their TV terms to the same gradient slices)
}

// ---- host side ---------------------------------------------------------------------------------
struct MsLayout {
  size_t lo_off[BDS_MAX_LEVELS], lg_off[BDS_MAX_LEVELS], p_off[BDS_MAX_LEVELS], q_off[BDS_MAX_LEVELS], r_off[BDS_MAX_LEVELS], vg_off[BDS_MAX_LEVELS];
  size_t part_off;  // per-workgroup partial grid gradients (shared by the levels, which run one after the other)
  size_t bytes;
};
constexpr int kPartBlocks = 1024;              // upper bound of persistent workgroups of the low-res backward (4 per CU)
constexpr size_t kPartBytes = 64u << 20;       // bound of the partial-grid buffer
static int part_blocks(size_t gbytes) {
  size_t n = gbytes ? kPartBytes / gbytes : kPartBlocks;
  if (n > (size_t)kPartBlocks) n = kPartBlocks;
  if (n < 256) n = 256;
  return (int)n;
}
constexpr size_t kMaxGridLds = 150 * 1024;     // a level's grid gradient must fit the 160 KiB LDS to use the LDS path
static MsLayout ms_layout(int nlevels, const bds_bilagrid_level_t *lv, int H, int W) {
  MsLayout L;
  size_t off = 0;
  for (int l = 0; l < nlevels; l++) {
    const int Hd = H / lv[l].factor, Wd = W / lv[l].factor;
    L.lo_off[l] = off;
    off += align_up((size_t)Hd * Wd * 12 * sizeof(float), 256);
    L.lg_off[l] = off;
    off += align_up((size_t)Hd * Wd * sizeof(float), 256);
  }
  for (int l = 0; l < nlevels; l++) {
    const int Hd = H / lv[l].factor, Wd = W / lv[l].factor;
    L.p_off[l] = off; off += align_up((size_t)H * W * 3 * sizeof(float), 256);
    L.q_off[l] = off; off += align_up((size_t)H * W * 3 * sizeof(float), 256);
    L.r_off[l] = off;
    if (!(Hd == H && Wd == W)) off += align_up((size_t)H * Wd * 12 * sizeof(float), 256);
    L.vg_off[l] = off; off += align_up((size_t)Hd * Wd * sizeof(float), 256);
  }
  L.part_off = off;
  size_t gmax = 0;
  for (int l = 0; l < nlevels; l++) {
    const size_t g = sizeof(float) * 12 * lv[l].gl * lv[l].gy * lv[l].gx * lv[l].n_avg;
    if (g <= kMaxGridLds && g > gmax) gmax = g;
  }
  size_t psum = 0;  // the levels run concurrently: every level has its own partial region
  for (int l = 0; l < nlevels; l++) {
    const size_t g = sizeof(float) * 12 * lv[l].gl * lv[l].gy * lv[l].gx * lv[l].n_avg;
    if (g <= kMaxGridLds) psum += align_up(g * part_blocks(g), 256);
  }
  (void)gmax;
  off += psum;
  L.bytes = off;
  return L;
}

static int ms_fill(MsParams &p, int nlevels, const bds_bilagrid_level_t *lv, int H, int W, const float *rgb,
                   const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *const *affine_out) {
  BDS_REQUIRE(nlevels >= 1 && nlevels <= BDS_MAX_LEVELS && lv && H > 0 && W > 0 && rgb && ws);
  BDS_REQUIRE((sky == nullptr) || (alpha != nullptr));
  // element offsets into the image-sized arrays (up to 12 floats per pixel) are formed in 32 bits; rows / columns below 2^23 (24-bit multiplies)
  BDS_REQUIRE((int64_t)H * W * 12 < ((int64_t)1 << 31) && H < (1 << 23) && W < (1 << 23));
  const MsLayout L = ms_layout(nlevels, lv, H, W);
  if (ws_bytes < L.bytes) return BDS_EWORKSPACE;
  BDS_REQUIRE(aligned16(ws));
  p.nlevels = nlevels; p.H = H; p.W = W; p.rgb = rgb; p.alpha = alpha; p.sky = sky;
  p.cs = 3; p.depth_out = nullptr; p.v_depth = nullptr; p.v_alpha_in = nullptr;
  char *base = static_cast<char *>(ws);
  for (int l = 0; l < nlevels; l++) {
    BDS_REQUIRE(lv[l].grid && lv[l].gx >= 1 && lv[l].gy >= 1 && lv[l].gl >= 1 && lv[l].factor >= 1 && lv[l].n_avg >= 1);
    BDS_REQUIRE(H / lv[l].factor >= 1 && W / lv[l].factor >= 1);
    LevelDev &d = p.lv[l];
    d.grid = lv[l].grid; d.v_grid = lv[l].v_grid;
    d.lo = reinterpret_cast<float *>(base + L.lo_off[l]);
    d.lg = reinterpret_cast<float *>(base + L.lg_off[l]);
    d.P = reinterpret_cast<float *>(base + L.p_off[l]);
    d.Q = reinterpret_cast<float *>(base + L.q_off[l]);
    d.R = reinterpret_cast<float *>(base + L.r_off[l]);
    d.vg = reinterpret_cast<float *>(base + L.vg_off[l]);
    d.aff_out = affine_out ? affine_out[l] : nullptr;
    BDS_REQUIRE(d.aff_out == nullptr || aligned16(d.aff_out));
    d.gx = lv[l].gx; d.gy = lv[l].gy; d.gl = lv[l].gl; d.factor = lv[l].factor; d.n_avg = lv[l].n_avg;
    d.Hd = H / lv[l].factor; d.Wd = W / lv[l].factor;
    d.up_x = (float)d.Wd / (float)W; d.up_y = (float)d.Hd / (float)H;
    d.dn_x = (float)W / (float)d.Wd; d.dn_y = (float)H / (float)d.Hd;
    d.lin_x = d.Wd > 1 ? 1.0f / (float)(d.Wd - 1) : 0.f; d.lin_y = d.Hd > 1 ? 1.0f / (float)(d.Hd - 1) : 0.f;
    d.magic_wd = divmod_magic(d.Wd);
    d.dn_shift = 0;
    {
      const int f = lv[l].factor;
      if (f >= 2 && (f & (f - 1)) == 0 && d.Hd * f == H && d.Wd * f == W) {
        int sh = 0;
        while ((1 << sh) < f) sh++;
        d.dn_shift = sh;
      }
    }
  }
  p.magic_w = divmod_magic(W);
  return BDS_OK;
}

}  // namespace bds

using namespace bds;

extern "C" size_t bds_bilagrid_ms_workspace_bytes(int nlevels, const bds_bilagrid_level_t *levels, int H, int W) {
  if (nlevels < 1 || nlevels > BDS_MAX_LEVELS || !levels || H <= 0 || W <= 0) return 0;
  for (int l = 0; l < nlevels; l++)
    if (levels[l].factor < 1) return 0;
  return ms_layout(nlevels, levels, H, W).bytes;
}

static int l1_tv_train_launch(int64_t n, const float *a, const float *b, const TvLevels &T, int tv_blocks, float v_loss, float *loss_out,
                              int loss_slots, float *v_a, hipStream_t st);

static int ms_fwd_impl(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb, int cs,
                       const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out, float *depth_out,
                       float *const *affine_out, bds_stream_t stream, const TrainLoss *train = nullptr) {
  MsParams p;
  int rc = ms_fill(p, nlevels, levels, H, W, rgb, alpha, sky, ws, ws_bytes, affine_out);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(rgb_out);
  p.cs = cs; p.depth_out = depth_out;
  hipStream_t st = as_stream(stream);
  const bool cells = (option_get(kOptCells) & 1) != 0;
  if (cells && cells_fused_ok(p))   // one level at full resolution: slice + application (+ loss) in one launch, no maps in memory
    return cells_fused_fwd(p, rgb_out, train, st);
  if ((option_get(kOptCells) & 2) && tile_fwd_ok(p))   // pyramid of dividing power-of-two factors: one pass over the image
    return tile_fwd(p, rgb_out, train, st);
  {
    // low-resolution slice of every level: cell-aligned jobs (csrc/bilagrid_cells.hip) for the levels with one grid; levels averaged
    // over several grids (the test branch) take the general kernels -- grids up to 32 KB staged whole in LDS by persistent workgroups
    // (one launch for all of them), larger ones gathered from global memory, one workgroup per chunk
    constexpr size_t kStageGridBytes = 32 * 1024;
    LevelSched sl{}, sg{};
    size_t lds_max = 0;
    unsigned cell_mask = 0;
    for (int l = 0; l < nlevels; l++) {
      if (cells && cells_level_ok(p.lv[l])) { cell_mask |= 1u << l; continue; }
      const size_t gbytes = sizeof(float) * 12 * p.lv[l].gl * p.lv[l].gy * p.lv[l].gx * p.lv[l].n_avg;
      const int64_t chunks = cdiv((int64_t)p.lv[l].Hd * p.lv[l].Wd, kBgBlock);
      LevelSched &s = gbytes <= kStageGridBytes ? sl : sg;
      const int k = s.n++;
      s.level[k] = l;
      s.nblk[k] = (int)(gbytes <= kStageGridBytes ? (chunks + 3) / 4 : chunks);
      if (s.nblk[k] < 1) s.nblk[k] = 1;
      s.blk_off[k + 1] = s.blk_off[k] + s.nblk[k];
      if (gbytes <= kStageGridBytes && gbytes > lds_max) lds_max = gbytes;
    }
    rc = cells_lowres_fwd(p, cell_mask, st);
    if (rc != BDS_OK) return rc;
    if (sl.n > 0) {
      hipLaunchKernelGGL((ms_lowres_fwd_kernel<true>), dim3((unsigned)sl.blk_off[sl.n]), dim3(kBgBlock), lds_max, st, p, sl);
      BDS_LAUNCH_CHECK();
    }
    if (sg.n > 0) {
      hipLaunchKernelGGL((ms_lowres_fwd_kernel<false>), dim3((unsigned)sg.blk_off[sg.n]), dim3(kBgBlock), 0, st, p, sg);
      BDS_LAUNCH_CHECK();
    }
  }
  {
    const int pix_blocks = (int)cdiv((int64_t)H * W, kBgBlock);
    const dim3 block(kBgBlock);
    if (train) {   // the loss rides on the launch: L1 in the pixel workgroups' epilogue, TV in extra workgroups behind them
      TrainLoss tl = *train;
      tl.pix_blocks = pix_blocks;
      const dim3 grid((unsigned)(pix_blocks + tl.tv_blocks));
      switch (nlevels) {
        case 1: hipLaunchKernelGGL((ms_apply_fwd_kernel<1, true>), grid, block, 0, st, p, rgb_out, tl); break;
        case 2: hipLaunchKernelGGL((ms_apply_fwd_kernel<2, true>), grid, block, 0, st, p, rgb_out, tl); break;
        case 3: hipLaunchKernelGGL((ms_apply_fwd_kernel<3, true>), grid, block, 0, st, p, rgb_out, tl); break;
        case 4: hipLaunchKernelGGL((ms_apply_fwd_kernel<4, true>), grid, block, 0, st, p, rgb_out, tl); break;
        default: hipLaunchKernelGGL((ms_apply_fwd_kernel<BDS_MAX_LEVELS, true>), grid, block, 0, st, p, rgb_out, tl); break;
      }
      BDS_LAUNCH_CHECK();
      return BDS_OK;
    }
    const dim3 grid((unsigned)pix_blocks);
    TrainLoss none{};
    switch (nlevels) {
      case 1: hipLaunchKernelGGL((ms_apply_fwd_kernel<1, false>), grid, block, 0, st, p, rgb_out, none); break;
      case 2: hipLaunchKernelGGL((ms_apply_fwd_kernel<2, false>), grid, block, 0, st, p, rgb_out, none); break;
      case 3: hipLaunchKernelGGL((ms_apply_fwd_kernel<3, false>), grid, block, 0, st, p, rgb_out, none); break;
      case 4: hipLaunchKernelGGL((ms_apply_fwd_kernel<4, false>), grid, block, 0, st, p, rgb_out, none); break;
      default: hipLaunchKernelGGL((ms_apply_fwd_kernel<BDS_MAX_LEVELS, false>), grid, block, 0, st, p, rgb_out, none); break;
    }
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_ms_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb,
                                   const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out,
                                   float *const *affine_out, bds_stream_t stream) {
  return ms_fwd_impl(nlevels, levels, H, W, rgb, 3, alpha, sky, ws, ws_bytes, rgb_out, nullptr, affine_out, stream);
}

extern "C" int bds_bilagrid_ms_ed_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                                      const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out,
                                      float *depth_out, bds_stream_t stream) {
  BDS_REQUIRE(alpha && depth_out);
  return ms_fwd_impl(nlevels, levels, H, W, render, 4, alpha, sky, ws, ws_bytes, rgb_out, depth_out, nullptr, stream);
}

// The backward's last stage can move into the compositor's backward (ed_epilogue.h) when every level takes the cell-aligned low-res
// stage (nothing rides on the epilogue's launch) and its guidance taps are the closed form: a dividing power-of-two factor, or 1.
static bool epilogue_deferrable(const MsParams &p) {
  if (!(option_get(kOptCells) & 1) || (option_get(kOptCells) & 4) || cells_fused_ok(p)) return false;
  for (int l = 0; l < p.nlevels; l++) {
    const LevelDev &L = p.lv[l];
    if (!cells_level_ok(L)) return false;
    if (!(L.dn_shift > 0 || (L.Hd == p.H && L.Wd == p.W))) return false;
  }
  return true;
}
namespace bds {
int ed_epilogue_fill(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, void *ws, size_t ws_bytes, EdEpilogue *e) {
  MsParams p;
  float dummy = 0.f;
  int rc = ms_fill(p, nlevels, levels, H, W, &dummy, nullptr, nullptr, ws, ws_bytes, nullptr);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(e && epilogue_deferrable(p) && (int64_t)H * W * 4 < ((int64_t)1 << 31));
  *e = EdEpilogue{};
  e->nlevels = nlevels; e->W = W;
  for (int l = 0; l < nlevels; l++) { e->vg[l] = p.lv[l].vg; e->Wd[l] = p.lv[l].Wd; e->shift[l] = p.lv[l].dn_shift; }
  return BDS_OK;
}
}  // namespace bds

static int ms_bwd_impl(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb, int cs,
                       const float *alpha, const float *sky, void *ws, size_t ws_bytes, const float *v_rgb_out,
                       const float *v_depth, const float *v_alpha_in, float *v_rgb, float *v_alpha, float *v_sky,
                       bds_stream_t stream, bool defer_epilogue = false) {
  MsParams p;
  int rc = ms_fill(p, nlevels, levels, H, W, rgb, alpha, sky, ws, ws_bytes, nullptr);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(v_rgb_out && v_rgb);
  if (defer_epilogue) BDS_REQUIRE(epilogue_deferrable(p));
  p.cs = cs; p.v_depth = v_depth; p.v_alpha_in = v_alpha_in;
  hipStream_t st = as_stream(stream);
  const int64_t HW = (int64_t)H * W;
  const bool cells = (option_get(kOptCells) & 1) != 0;
  if (cells && cells_fused_ok(p))   // one level at full resolution: the whole backward in one launch, no scratch
    return cells_fused_bwd(p, v_rgb_out, v_rgb, v_alpha, v_sky, st);
  // fused x pass when some level is up-sampled and the widest support fits the workgroup's halo
  float smax = 1.f;
  bool any_up = false;
  for (int l = 0; l < nlevels; l++) {
    if (p.lv[l].Hd == H && p.lv[l].Wd == W) continue;
    any_up = true;
    const float sc = (float)W / (float)p.lv[l].Wd;
    if (sc > smax) smax = sc;
  }
  const int halo = (int)ceilf(smax) + 2;
  if (any_up && halo <= 48 && !(option_get(kOptDebug) & 8)) {
    const int stride = kBgBlock - 2 * halo;
    const int nbx = (int)cdiv(W, stride);
    const dim3 grid((unsigned)((int64_t)H * nbx)), block(kBgBlock);
    switch (nlevels) {
      case 1: hipLaunchKernelGGL((ms_apply_bwd_x_kernel<1>), grid, block, 0, st, p, v_rgb_out, v_rgb, halo, nbx, option_get(kOptDebug)); break;
      case 2: hipLaunchKernelGGL((ms_apply_bwd_x_kernel<2>), grid, block, 0, st, p, v_rgb_out, v_rgb, halo, nbx, option_get(kOptDebug)); break;
      case 3: hipLaunchKernelGGL((ms_apply_bwd_x_kernel<3>), grid, block, 0, st, p, v_rgb_out, v_rgb, halo, nbx, option_get(kOptDebug)); break;
      case 4: hipLaunchKernelGGL((ms_apply_bwd_x_kernel<4>), grid, block, 0, st, p, v_rgb_out, v_rgb, halo, nbx, option_get(kOptDebug)); break;
      default: hipLaunchKernelGGL((ms_apply_bwd_x_kernel<BDS_MAX_LEVELS>), grid, block, 0, st, p, v_rgb_out, v_rgb, halo, nbx, option_get(kOptDebug)); break;
    }
    BDS_LAUNCH_CHECK();
  } else {
    {
      const dim3 grid((unsigned)cdiv(HW, kBgBlock)), block(kBgBlock);
      switch (nlevels) {
        case 1: hipLaunchKernelGGL((ms_apply_bwd_kernel<1>), grid, block, 0, st, p, v_rgb_out, v_rgb); break;
        case 2: hipLaunchKernelGGL((ms_apply_bwd_kernel<2>), grid, block, 0, st, p, v_rgb_out, v_rgb); break;
        case 3: hipLaunchKernelGGL((ms_apply_bwd_kernel<3>), grid, block, 0, st, p, v_rgb_out, v_rgb); break;
        case 4: hipLaunchKernelGGL((ms_apply_bwd_kernel<4>), grid, block, 0, st, p, v_rgb_out, v_rgb); break;
        default: hipLaunchKernelGGL((ms_apply_bwd_kernel<BDS_MAX_LEVELS>), grid, block, 0, st, p, v_rgb_out, v_rgb); break;
      }
    }
    BDS_LAUNCH_CHECK();
    {  // x pass of the up-sampler adjoint, all up-sampled levels in one launch
      LevelSched sc{};
      for (int l = 0; l < nlevels; l++) {
        if (p.lv[l].Hd == H && p.lv[l].Wd == W) continue;
        const int k = sc.n++;
        sc.level[k] = l;
        sc.nblk[k] = (int)cdiv((int64_t)H * p.lv[l].Wd, kBgBlock);
        sc.blk_off[k + 1] = sc.blk_off[k] + sc.nblk[k];
      }
      if (sc.n > 0) {
        hipLaunchKernelGGL(ms_adjoint_x_kernel, dim3((unsigned)sc.blk_off[sc.n]), dim3(kBgBlock), 0, st, p, sc);
        BDS_LAUNCH_CHECK();
      }
    }
  }
  const MsLayout ML = ms_layout(nlevels, levels, H, W);
  float *partials = reinterpret_cast<float *>(static_cast<char *>(ws) + ML.part_off);
  PartialsJob pj{};
  int extra_blocks = 0;
  {  // low-res backward.  Levels with one grid: cell-aligned jobs (csrc/bilagrid_cells.hip), all of them in one launch.  The others
     // (averaged grids): LDS-path levels together in one persistent launch; their partial grids are reduced by the next launch
    LevelSched sc{}, red{};
    size_t lds_max = 0;
    long long poff = 0;
    unsigned cell_mask = 0;
    for (int l = 0; l < nlevels; l++)
      if (cells && cells_level_ok(p.lv[l])) cell_mask |= 1u << l;
    rc = cells_lowres_bwd(p, cell_mask, st);
    if (rc != BDS_OK) return rc;
    for (int l = 0; l < nlevels; l++) {
      if (cell_mask & (1u << l)) continue;
      const int gtot = 12 * p.lv[l].gl * p.lv[l].gy * p.lv[l].gx * p.lv[l].n_avg;
      const size_t gbytes = sizeof(float) * gtot;
      const int64_t need = cdiv((int64_t)p.lv[l].Hd * p.lv[l].Wd, kBgBlock);
      if (gbytes > kMaxGridLds) {  // grid gradient too large for LDS: direct global atomics, own launch
        LevelSched one{};
        one.n = 1; one.level[0] = l; one.nblk[0] = (int)need; one.blk_off[1] = (int)need;
        hipLaunchKernelGGL((ms_lowres_bwd_kernel<false>), dim3((unsigned)need), dim3(kBgBlock), 0, st, p, one, v_rgb, partials,
                           option_get(kOptDebug), 0);
        BDS_LAUNCH_CHECK();
        continue;
      }
      const int cap = part_blocks(gbytes);
      const int k = sc.n++;
      sc.level[k] = l;
      sc.nblk[k] = (int)(need < cap ? need : cap);
      sc.blk_off[k + 1] = sc.blk_off[k] + sc.nblk[k];
      sc.part_off[k] = poff;
      poff += (long long)(align_up(gbytes * part_blocks(gbytes), 256) / sizeof(float));
      if (gbytes > lds_max) lds_max = gbytes;
      red.level[k] = l;
      red.nblk[k] = p.lv[l].v_grid ? (int)cdiv(gtot, 32) : 0;
      red.blk_off[k + 1] = red.blk_off[k] + red.nblk[k];
      red.n = sc.n;
    }
    if (sc.n > 0) {
      // room for the cell-major grid copies of the small levels (accumulator + copy <= 48 KiB) next to the largest accumulator
      size_t lds_bytes = lds_max;
      for (int k = 0; k < sc.n; k++) {
        const LevelDev &Lk = p.lv[sc.level[k]];
        const size_t gb = sizeof(float) * 12 * Lk.gl * Lk.gy * Lk.gx * Lk.n_avg;
        if (2 * gb <= 48 * 1024 && 2 * gb > lds_bytes) lds_bytes = 2 * gb;
      }
      if (lds_bytes > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ms_lowres_bwd_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
          return BDS_ELAUNCH;
      }
      // (unroll 2 of the y pass -- 128 instead of 147 VGPRs, a fourth resident wave -- measured slower: the loop wants the loads in flight)
      hipLaunchKernelGGL((ms_lowres_bwd_kernel<true>), dim3((unsigned)sc.blk_off[sc.n]), dim3(kBgBlock), lds_bytes, st, p, sc, v_rgb,
                         partials, option_get(kOptDebug), (int)(lds_bytes / sizeof(float)));
      BDS_LAUNCH_CHECK();
      if (red.blk_off[red.n] > 0) { pj.sc = sc; pj.red = red; pj.partials = partials; extra_blocks = red.blk_off[red.n]; }
    }
  }
  if (defer_epilogue) return BDS_OK;   // (the compositor's backward finishes the pixel: bds_rasterize_bwd_ms)
  {
    // (the reduction of the partial grids rides on this launch as extra workgroups)
    pj.pix_blocks = (int)cdiv(HW, kBgBlock);
    const dim3 grid((unsigned)(pj.pix_blocks + extra_blocks)), block(kBgBlock);
    switch (nlevels) {
      case 1: hipLaunchKernelGGL((ms_guidance_blend_bwd_kernel<1>), grid, block, 0, st, p, v_rgb, v_alpha, v_sky, option_get(kOptDebug), pj); break;
      case 2: hipLaunchKernelGGL((ms_guidance_blend_bwd_kernel<2>), grid, block, 0, st, p, v_rgb, v_alpha, v_sky, option_get(kOptDebug), pj); break;
      case 3: hipLaunchKernelGGL((ms_guidance_blend_bwd_kernel<3>), grid, block, 0, st, p, v_rgb, v_alpha, v_sky, option_get(kOptDebug), pj); break;
      case 4: hipLaunchKernelGGL((ms_guidance_blend_bwd_kernel<4>), grid, block, 0, st, p, v_rgb, v_alpha, v_sky, option_get(kOptDebug), pj); break;
      default: hipLaunchKernelGGL((ms_guidance_blend_bwd_kernel<BDS_MAX_LEVELS>), grid, block, 0, st, p, v_rgb, v_alpha, v_sky, option_get(kOptDebug), pj); break;
    }
    BDS_LAUNCH_CHECK();
  }
  return BDS_OK;
}

// ---- the image's grids by a DEVICE-side image index (a captured view whose image changes from replay to replay) -------------------
namespace bds {
struct GridSelect {
  int n;
  const float *full[BDS_MAX_LEVELS];   // [n_img, count]
  float *v_full[BDS_MAX_LEVELS];
  float *sel[BDS_MAX_LEVELS];          // [count]
  int count[BDS_MAX_LEVELS], n_img[BDS_MAX_LEVELS];
  int blk_off[BDS_MAX_LEVELS + 1];
  volatile int32_t *err;
};
// kBwd = false: sel = full[idx];  kBwd = true: v_full[idx] += sel, sel = 0 (ready for the next replay's backward)
template <bool kBwd>
__global__ __launch_bounds__(kBgBlock) void grid_select_kernel(GridSelect S, const int32_t *__restrict__ idx_dev) {
  int k = 0;
  while (k + 1 < S.n && (int)blockIdx.x >= S.blk_off[k + 1]) k++;
  const int e = ((int)blockIdx.x - S.blk_off[k]) * kBgBlock + (int)threadIdx.x;
  const int idx = *idx_dev;
  // an index outside [0, n_img) selects / adds nothing -- and says so: the host cannot see a device-side index, so the sticky error
  // word (page-locked, may be NULL) is what graph_view.FrameGraph.valid() looks at.  The backward clears the staging gradients
  // either way: a stale gradient must not leak into the next replay.
  const bool ok = idx >= 0 && idx < S.n_img[k];
  if (!ok && S.err && blockIdx.x == 0 && threadIdx.x == 0) *S.err = 1;
  if (e >= S.count[k]) return;
  const int64_t o = (int64_t)(ok ? idx : 0) * S.count[k] + e;
  if (kBwd) {
    const float t = S.sel[k][e];
    if (ok && t != 0.f) atomicAdd(S.v_full[k] + o, t);   // (atomic: another view's TV term may add to the same slice next to this launch)
    S.sel[k][e] = 0.f;
  } else if (ok) {
    S.sel[k][e] = S.full[k][o];
  }
}
}  // namespace bds

static int grid_select_launch(int nlevels, const bds_bilagrid_level_t *levels, const int32_t *img_idx_dev, float *const *sel, bool bwd,
                              int32_t *error_pinned, bds_stream_t stream) {
  BDS_REQUIRE(nlevels >= 1 && nlevels <= BDS_MAX_LEVELS && levels && img_idx_dev && sel);
  GridSelect S{};
  S.n = nlevels;
  if (error_pinned) {
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, error_pinned, 0) != hipSuccess) { (void)hipGetLastError(); return BDS_EINVAL; }
    S.err = static_cast<volatile int32_t *>(mapped);
  }
  for (int l = 0; l < nlevels; l++) {
    BDS_REQUIRE(sel[l] && levels[l].gx >= 1 && levels[l].gy >= 1 && levels[l].gl >= 1 && levels[l].n_avg >= 1);
    BDS_REQUIRE(bwd ? levels[l].v_grid != nullptr : levels[l].grid != nullptr);
    S.full[l] = levels[l].grid; S.v_full[l] = levels[l].v_grid; S.sel[l] = sel[l];
    S.count[l] = 12 * levels[l].gl * levels[l].gy * levels[l].gx;
    S.n_img[l] = levels[l].n_avg;
    S.blk_off[l + 1] = S.blk_off[l] + (int)cdiv(S.count[l], kBgBlock);
  }
  if (bwd) hipLaunchKernelGGL((grid_select_kernel<true>), dim3((unsigned)S.blk_off[nlevels]), dim3(kBgBlock), 0, as_stream(stream), S, img_idx_dev);
  else hipLaunchKernelGGL((grid_select_kernel<false>), dim3((unsigned)S.blk_off[nlevels]), dim3(kBgBlock), 0, as_stream(stream), S, img_idx_dev);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
extern "C" int bds_bilagrid_select(int nlevels, const bds_bilagrid_level_t *levels, const int32_t *img_idx_dev, float *const *sel,
                                   int32_t *error_pinned, bds_stream_t stream) {
  return grid_select_launch(nlevels, levels, img_idx_dev, sel, false, error_pinned, stream);
}
extern "C" int bds_bilagrid_select_bwd(int nlevels, const bds_bilagrid_level_t *levels, const int32_t *img_idx_dev, float *const *v_sel,
                                       int32_t *error_pinned, bds_stream_t stream) {
  return grid_select_launch(nlevels, levels, img_idx_dev, v_sel, true, error_pinned, stream);
}

// Names (as rocprofv3 prints them, without "bds::" and the argument list; comma-separated) of the kernels that the forward / backward
// of this configuration launches under the current options, in launch order; measurement plumbing for bench.py's counter look-up.
extern "C" int bds_bilagrid_kernel_names(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, int backward, int train,
                                         char *buf, int buf_len) {
  BDS_REQUIRE(buf && buf_len > 0 && nlevels >= 1 && nlevels <= BDS_MAX_LEVELS && levels && H > 0 && W > 0);
  MsParams p{};
  p.nlevels = nlevels; p.H = H; p.W = W;
  bool any_cell = false, any_lds = false, any_gather = false, any_up = false;
  for (int l = 0; l < nlevels; l++) {
    BDS_REQUIRE(levels[l].factor >= 1);
    LevelDev &d = p.lv[l];
    d.gx = levels[l].gx; d.gy = levels[l].gy; d.gl = levels[l].gl; d.factor = levels[l].factor; d.n_avg = levels[l].n_avg;
    d.Hd = H / levels[l].factor; d.Wd = W / levels[l].factor;
    d.aff_out = nullptr;
    const size_t gbytes = sizeof(float) * 12 * d.gl * d.gy * d.gx * d.n_avg;
    {
      const int f = d.factor;
      d.dn_shift = (f >= 2 && (f & (f - 1)) == 0 && d.Hd * f == H && d.Wd * f == W) ? 1 : 0;   // (non-zero is all tile_fwd_ok asks)
    }
    if ((option_get(kOptCells) & 1) && cells_level_ok(d)) any_cell = true;
    else if (gbytes <= (backward ? kMaxGridLds : (size_t)32 * 1024)) any_lds = true;
    else any_gather = true;
    if (!(d.Hd == H && d.Wd == W)) any_up = true;
  }
  const int nl = nlevels <= 4 ? nlevels : BDS_MAX_LEVELS;
  char tmp[512];
  int n = 0;
  if (!backward && !((option_get(kOptCells) & 1) && cells_fused_ok(p)) && (option_get(kOptCells) & 2) && tile_fwd_ok(p)) {
    n = snprintf(tmp, sizeof(tmp), "ms_tile_fwd_kernel<%d, %s>", nlevels, train ? "true" : "false");
  } else if ((option_get(kOptCells) & 1) && cells_fused_ok(p)) {
    n = backward ? snprintf(tmp, sizeof(tmp), "cell_bwd_kernel<true>") : snprintf(tmp, sizeof(tmp), "cell_fwd_kernel<true, %s>", train ? "true" : "false");
  } else if (!backward) {
    n = snprintf(tmp, sizeof(tmp), "%s%s%sms_apply_fwd_kernel<%d, %s>", any_cell ? "cell_fwd_kernel<false, false>," : "",
                 any_lds ? "ms_lowres_fwd_kernel<true>," : "", any_gather ? "ms_lowres_fwd_kernel<false>," : "", nl, train ? "true" : "false");
  } else {
    char first[96];
    if (any_up) snprintf(first, sizeof(first), "ms_apply_bwd_x_kernel<%d>", nl);
    else snprintf(first, sizeof(first), "ms_apply_bwd_kernel<%d>", nl);
    n = snprintf(tmp, sizeof(tmp), "%s,%s%s%sms_guidance_blend_bwd_kernel<%d>", first, any_cell ? "cell_bwd_kernel<false>," : "",
                 any_gather ? "ms_lowres_bwd_kernel<false, 4>," : "", any_lds ? "ms_lowres_bwd_kernel<true, 4>," : "", nl);
  }
  if (n <= 0 || n >= buf_len || n >= (int)sizeof(tmp)) return BDS_EINVAL;
  memcpy(buf, tmp, (size_t)n + 1);
  return BDS_OK;
}

extern "C" int bds_bilagrid_ms_bwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb,
                                   const float *alpha, const float *sky, void *ws, size_t ws_bytes,
                                   const float *v_rgb_out, float *v_rgb, float *v_alpha, float *v_sky,
                                   bds_stream_t stream) {
  return ms_bwd_impl(nlevels, levels, H, W, rgb, 3, alpha, sky, ws, ws_bytes, v_rgb_out, nullptr, nullptr, v_rgb, v_alpha, v_sky,
                     stream);
}

extern "C" int bds_bilagrid_ms_ed_bwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                                      const float *alpha, const float *sky, void *ws, size_t ws_bytes,
                                      const float *v_rgb_out, const float *v_depth, const float *v_opacity, float *v_render,
                                      float *v_alpha, float *v_sky, bds_stream_t stream) {
  BDS_REQUIRE(alpha && v_alpha);
  return ms_bwd_impl(nlevels, levels, H, W, render, 4, alpha, sky, ws, ws_bytes, v_rgb_out, v_depth, v_opacity, v_render, v_alpha,
                     v_sky, stream);
}

// 1 when this configuration's backward can leave its last stage to the compositor's backward (bds_rasterize_bwd_ms), else 0
extern "C" int bds_bilagrid_ms_ed_bwd_deferrable(int nlevels, const bds_bilagrid_level_t *levels, int H, int W) {
  if (nlevels < 1 || nlevels > BDS_MAX_LEVELS || !levels || H <= 0 || W <= 0) return 0;
  MsParams p{};
  p.nlevels = nlevels; p.H = H; p.W = W;
  for (int l = 0; l < nlevels; l++) {
    if (levels[l].factor < 1) return 0;
    LevelDev &d = p.lv[l];
    const int f = levels[l].factor;
    d.gx = levels[l].gx; d.gy = levels[l].gy; d.gl = levels[l].gl; d.factor = f; d.n_avg = levels[l].n_avg;
    d.Hd = H / f; d.Wd = W / f;
    d.aff_out = nullptr;
    d.dn_shift = (f >= 2 && (f & (f - 1)) == 0 && d.Hd * f == H && d.Wd * f == W) ? 1 : 0;
  }
  return epilogue_deferrable(p) ? 1 : 0;
}

extern "C" int bds_bilagrid_ms_ed_bwd_deferred(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                                               const float *alpha, const float *sky, void *ws, size_t ws_bytes,
                                               const float *v_rgb_out, float *v_direct, bds_stream_t stream) {
  BDS_REQUIRE(alpha && v_direct);
  return ms_bwd_impl(nlevels, levels, H, W, render, 4, alpha, sky, ws, ws_bytes, v_rgb_out, nullptr, nullptr, v_direct, nullptr, nullptr,
                     stream, true);
}

extern "C" int bds_bilagrid_slice_fwd(int64_t P, const float *grid, int gx, int gy, int gl, const float *xy,
                                      const float *rgb, float *affine, bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && gx >= 1 && gy >= 1 && gl >= 1);
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(grid && xy && rgb && affine);
  hipLaunchKernelGGL(slice_fwd_kernel, dim3((unsigned)cdiv(P, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), P, grid, gx, gy,
                     gl, xy, rgb, affine);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_slice_bwd(int64_t P, const float *grid, int gx, int gy, int gl, const float *xy,
                                      const float *rgb, const float *v_affine, float *v_grid, float *v_rgb,
                                      bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && gx >= 1 && gy >= 1 && gl >= 1);
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(grid && xy && rgb && v_affine);
  hipLaunchKernelGGL(slice_bwd_kernel, dim3((unsigned)cdiv(P, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), P, grid, gx, gy,
                     gl, xy, rgb, v_affine, v_grid, v_rgb);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static void tv_scales(int64_t n, int gx, int gy, int gl, float weight, float &sl, float &sy, float &sx, int channels = 12) {
  // lib_bilagrid.py:147-168: each axis' squared differences are divided by the element count of
  // the differenced tensor (per batch item, floor 1); the sum is divided by the batch size.
  auto cnt = [channels](int64_t a, int64_t b, int64_t c) { double v = (double)channels * a * b * c; return v < 1.0 ? 1.0 : v; };
  sl = (float)(weight / (cnt(gl - 1, gy, gx) * (double)n));
  sy = (float)(weight / (cnt(gl, gy - 1, gx) * (double)n));
  sx = (float)(weight / (cnt(gl, gy, gx - 1) * (double)n));
}

extern "C" int bds_bilagrid_tv_fwd(int64_t n, int gx, int gy, int gl, const float *grids, float weight, float *tv_out,
                                   bds_stream_t stream) {
  BDS_REQUIRE(n >= 1 && gx >= 1 && gy >= 1 && gl >= 1 && grids && tv_out);
  float sl, sy, sx;
  tv_scales(n, gx, gy, gl, weight, sl, sy, sx);
  const int64_t total = n * 12 * gl * gy * gx;
  const int64_t blocks = cdiv(total, kBgBlock);
  hipLaunchKernelGGL(tv_fwd_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(kBgBlock), 0, as_stream(stream), total,
                     gx, gy, gl, grids, sl, sy, sx, tv_out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_tv_bwd(int64_t n, int gx, int gy, int gl, const float *grids, float weight, const float *v_tv,
                                   float *v_grids, bds_stream_t stream) {
  BDS_REQUIRE(n >= 1 && gx >= 1 && gy >= 1 && gl >= 1 && grids && v_tv && v_grids);
  float sl, sy, sx;
  tv_scales(n, gx, gy, gl, weight, sl, sy, sx);
  const int64_t total = n * 12 * gl * gy * gx;
  hipLaunchKernelGGL(tv_bwd_kernel, dim3((unsigned)cdiv(total, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), total, gx, gy,
                     gl, grids, sl, sy, sx, v_tv, v_grids);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

// ---- L1 + TV training loss, value AND gradient in one launch (the direct step: d(loss) is known to be `v_loss` before the value is) ----
// loss = mean |a - b| + sum_l w_l TV(grid_l)  (models/trainers/base.py:518-565 rgb term with losses.affine, modules.py:445,466-472);
// workgroups [0, tv_blocks) own the grids' elements (one element per thread: its forward differences for the value, its full stencil
// for the gradient, added to v_grid with one atomic), the rest stream the image (|a - b| summed, sign(a - b) * v_loss / n written).
// Replaces four launches (L1 forward, TV forward, L1 backward, TV backward) that read the image twice.
__global__ __launch_bounds__(kBgBlock) void l1_tv_train_kernel(TvLevels L, int tv_blocks, int64_t n4, int64_t n,
                                                              const float4 *__restrict__ a4, const float4 *__restrict__ b4,
                                                              const float *__restrict__ a, const float *__restrict__ b, float inv_n,
                                                              float v_loss, float *__restrict__ loss_out, int loss_slots,
                                                              float4 *__restrict__ v_a4, float *__restrict__ v_a) {
  __shared__ float red[kBgBlock / kWave];
  float acc = 0.f;
  if ((int)blockIdx.x < tv_blocks) {
    acc = tv_train_element(L, (int)blockIdx.x, v_loss);
  } else {
    const int64_t nb = (int64_t)gridDim.x - tv_blocks, bid = (int64_t)blockIdx.x - tv_blocks;
    const float gs = v_loss * inv_n;
    float s = 0.f;
    for (int64_t i = bid * kBgBlock + threadIdx.x; i < n4; i += nb * kBgBlock) {
      const float4 p = a4[i], q = b4[i];
      const float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
      s += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
      v_a4[i] = make_float4(d0 > 0.f ? gs : (d0 < 0.f ? -gs : 0.f), d1 > 0.f ? gs : (d1 < 0.f ? -gs : 0.f),
                            d2 > 0.f ? gs : (d2 < 0.f ? -gs : 0.f), d3 > 0.f ? gs : (d3 < 0.f ? -gs : 0.f));   // torch: sign(0) = 0
    }
    if (bid == 0 && threadIdx.x < n - n4 * 4) {   // tail (n not a multiple of 4)
      const int64_t i = n4 * 4 + threadIdx.x;
      const float d = a[i] - b[i];
      s += fabsf(d);
      v_a[i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
    acc = s * inv_n;
  }
  acc = wave_sum_all(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBgBlock / kWave; w++) t += red[w];
    if (t != 0.f) atomicAdd(loss_out + (size_t)(blockIdx.x & (loss_slots - 1)) * kLossSlotStride, t);
  }
}

static int tv_levels_fill(TvLevels &T, int nlevels, const bds_bilagrid_level_t *lv, const float *weights, bool bwd, int cap) {
  BDS_REQUIRE(nlevels >= 1 && nlevels <= BDS_MAX_LEVELS && lv && weights);
  T.n = nlevels;
  T.blk_off[0] = 0;
  for (int l = 0; l < nlevels; l++) {
    BDS_REQUIRE(lv[l].grid && lv[l].gx >= 1 && lv[l].gy >= 1 && lv[l].gl >= 1 && lv[l].n_avg >= 1);
    BDS_REQUIRE(!bwd || lv[l].v_grid);
    T.x[l] = lv[l].grid; T.v_x[l] = lv[l].v_grid;
    T.gx[l] = lv[l].gx; T.gy[l] = lv[l].gy; T.gl[l] = lv[l].gl;
    T.total[l] = (long long)lv[l].n_avg * 12 * lv[l].gl * lv[l].gy * lv[l].gx;
    tv_scales(lv[l].n_avg, lv[l].gx, lv[l].gy, lv[l].gl, weights[l], T.sl[l], T.sy[l], T.sx[l]);
    int64_t nb = cdiv(T.total[l], kBgBlock);
    if (cap > 0 && nb > cap) nb = cap;
    T.blk_off[l + 1] = T.blk_off[l] + (int)nb;
  }
  return BDS_OK;
}

extern "C" int bds_bilagrid_tv_ms_fwd(int nlevels, const bds_bilagrid_level_t *levels, const float *weights, float *tv_out,
                                      bds_stream_t stream) {
  TvLevels T;
  int rc = tv_levels_fill(T, nlevels, levels, weights, false, 256);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(tv_out);
  hipLaunchKernelGGL(tv_ms_fwd_kernel, dim3((unsigned)T.blk_off[nlevels]), dim3(kBgBlock), 0, as_stream(stream), T, tv_out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_tv_ms_bwd(int nlevels, const bds_bilagrid_level_t *levels, const float *weights, const float *v_tv,
                                      bds_stream_t stream) {
  TvLevels T;
  int rc = tv_levels_fill(T, nlevels, levels, weights, true, 0);
  if (rc != BDS_OK) return rc;
  BDS_REQUIRE(v_tv);
  hipLaunchKernelGGL(tv_ms_bwd_kernel, dim3((unsigned)T.blk_off[nlevels]), dim3(kBgBlock), 0, as_stream(stream), T, v_tv);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int l1_tv_train_launch(int64_t n, const float *a, const float *b, const TvLevels &T, int tv_blocks, float v_loss, float *loss_out,
                              int loss_slots, float *v_a, hipStream_t st) {
  const int64_t n4 = n / 4;
  int64_t l1_blocks = cdiv(n4 > 0 ? n4 : 1, kBgBlock * 4);
  if (l1_blocks > 2048) l1_blocks = 2048;
  hipLaunchKernelGGL(l1_tv_train_kernel, dim3((unsigned)(tv_blocks + l1_blocks)), dim3(kBgBlock), 0, st, T, tv_blocks, n4, n,
                     reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), a, b, 1.0f / (float)n, v_loss, loss_out,
                     loss_slots, reinterpret_cast<float4 *>(v_a), v_a);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

static int tv_train_levels(TvLevels &T, int &tv_blocks, int nlevels, const bds_bilagrid_level_t *levels, const float *weights) {
  T.n = 0; T.blk_off[0] = 0;
  tv_blocks = 0;
  if (nlevels > 0) {
    int rc = tv_levels_fill(T, nlevels, levels, weights, false, 0);
    if (rc != BDS_OK) return rc;
    tv_blocks = T.blk_off[nlevels];
  }
  return BDS_OK;
}

extern "C" int bds_l1_tv_train(int64_t n, const float *a, const float *b, int nlevels, const bds_bilagrid_level_t *levels,
                               const float *weights, float v_loss, float *loss_out, int loss_slots, float *v_a, bds_stream_t stream) {
  BDS_REQUIRE(n > 0 && a && b && loss_out && v_a && aligned16(a) && aligned16(b) && aligned16(v_a));
  BDS_REQUIRE(loss_slots >= 1 && (loss_slots & (loss_slots - 1)) == 0);
  TvLevels T;
  int tv_blocks = 0;
  int rc = tv_train_levels(T, tv_blocks, nlevels, levels, weights);
  if (rc != BDS_OK) return rc;
  return l1_tv_train_launch(n, a, b, T, tv_blocks, v_loss, loss_out, loss_slots, v_a, as_stream(stream));
}

extern "C" int bds_bilagrid_ms_ed_train_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                                            const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out,
                                            float *depth_out, const float *target, int tv_nlevels,
                                            const bds_bilagrid_level_t *tv_levels, const float *tv_weights, float v_loss,
                                            float *loss_out, int loss_slots, float *v_rgb_out, bds_stream_t stream) {
  BDS_REQUIRE(alpha && depth_out && target && loss_out && v_rgb_out && H > 0 && W > 0);
  BDS_REQUIRE(loss_slots >= 1 && (loss_slots & (loss_slots - 1)) == 0);
  BDS_REQUIRE(aligned16(target) && aligned16(v_rgb_out) && aligned16(rgb_out));
  TrainLoss tl{};
  int rc = tv_train_levels(tl.T, tl.tv_blocks, tv_nlevels, tv_levels, tv_weights);
  if (rc != BDS_OK) return rc;
  tl.target = target; tl.v_out = v_rgb_out; tl.loss = loss_out; tl.loss_slots = loss_slots;
  tl.inv_n = 1.0f / (float)((int64_t)H * W * 3); tl.v_loss = v_loss;
  return ms_fwd_impl(nlevels, levels, H, W, render, 4, alpha, sky, ws, ws_bytes, rgb_out, depth_out, nullptr, stream, &tl);
}

extern "C" int bds_bilagrid_slice_feat_fwd(int64_t P, int NC, const float *grid, int gx, int gy, int gl, const float *xy,
                                           const float *rgb, float *out, bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && NC >= 1 && gx >= 1 && gy >= 1 && gl >= 1);
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(grid && xy && rgb && out);
  hipLaunchKernelGGL(slice_feat_fwd_kernel, dim3((unsigned)cdiv(P, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), P, NC, grid, gx,
                     gy, gl, xy, rgb, out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_slice_feat_bwd(int64_t P, int NC, const float *grid, int gx, int gy, int gl, const float *xy,
                                           const float *rgb, const float *v_out, float *v_grid, float *v_rgb,
                                           bds_stream_t stream) {
  BDS_REQUIRE(P >= 0 && NC >= 1 && gx >= 1 && gy >= 1 && gl >= 1);
  if (P == 0) return BDS_OK;
  BDS_REQUIRE(grid && xy && rgb && v_out);
  hipLaunchKernelGGL(slice_feat_bwd_kernel, dim3((unsigned)cdiv(P, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), P, NC, grid, gx,
                     gy, gl, xy, rgb, v_out, v_grid, v_rgb);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_slice_feat_image_ok(int NC, int gx, int gy, int gl) {
  return NC >= 1 && gx >= 1 && gy >= 1 && gl >= 1 && (int64_t)NC * gl * 2 * gx <= kBandMaxFloats;
}

extern "C" int bds_bilagrid_slice_feat_image_fwd(int H, int W, int NC, const float *grid, int gx, int gy, int gl, const float *rgb,
                                                 float *out, bds_stream_t stream) {
  BDS_REQUIRE(H >= 0 && W >= 0 && bds_bilagrid_slice_feat_image_ok(NC, gx, gy, gl));
  if (H == 0 || W == 0) return BDS_OK;
  BDS_REQUIRE(grid && rgb && out && ((NC & 3) || aligned16(out)));
  const float lin_x = W > 1 ? 1.0f / (float)(W - 1) : 0.f, lin_y = H > 1 ? 1.0f / (float)(H - 1) : 0.f;
  const size_t lds = (size_t)NC * gl * 2 * gx * sizeof(float);
  hipLaunchKernelGGL(slice_feat_image_fwd_kernel, dim3((unsigned)H), dim3(kBgBlock), lds, as_stream(stream), H, W, NC, grid, gx, gy, gl,
                     lin_x, lin_y, 1, rgb, out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_bilagrid_slice_feat_image_bwd(int H, int W, int NC, const float *grid, int gx, int gy, int gl, const float *rgb,
                                                 const float *v_out, float *v_grid, float *v_rgb, bds_stream_t stream) {
  BDS_REQUIRE(H >= 0 && W >= 0 && bds_bilagrid_slice_feat_image_ok(NC, gx, gy, gl));
  if (H == 0 || W == 0) return BDS_OK;
  BDS_REQUIRE(grid && rgb && v_out);
  const float lin_x = W > 1 ? 1.0f / (float)(W - 1) : 0.f, lin_y = H > 1 ? 1.0f / (float)(H - 1) : 0.f;
  const size_t lds = 2 * (size_t)NC * gl * 2 * gx * sizeof(float);
  // consecutive rows per workgroup: fewer band flushes; still >= 256 workgroups on a 1080-row image
  const int rows = H >= 1024 ? 4 : (H >= 512 ? 2 : 1);
  hipLaunchKernelGGL(slice_feat_image_bwd_kernel, dim3((unsigned)cdiv(H, rows)), dim3(kBgBlock), lds, as_stream(stream), H, W, NC, grid,
                     gx, gy, gl, lin_x, lin_y, rows, rgb, v_out, v_grid, v_rgb);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

// TV of grids with any channel count: x [n, channels, gl, gy, gx]
extern "C" int bds_grid_tv_fwd(int64_t n, int channels, int gx, int gy, int gl, const float *grids, float weight, float *tv_out,
                               bds_stream_t stream) {
  BDS_REQUIRE(n >= 1 && channels >= 1 && gx >= 1 && gy >= 1 && gl >= 1 && grids && tv_out);
  float sl, sy, sx;
  tv_scales(n, gx, gy, gl, weight, sl, sy, sx, channels);
  const int64_t total = n * channels * gl * gy * gx;
  const int64_t blocks = cdiv(total, kBgBlock);
  hipLaunchKernelGGL(tv_fwd_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(kBgBlock), 0, as_stream(stream), total,
                     gx, gy, gl, grids, sl, sy, sx, tv_out);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

extern "C" int bds_grid_tv_bwd(int64_t n, int channels, int gx, int gy, int gl, const float *grids, float weight, const float *v_tv,
                               float *v_grids, bds_stream_t stream) {
  BDS_REQUIRE(n >= 1 && channels >= 1 && gx >= 1 && gy >= 1 && gl >= 1 && grids && v_tv && v_grids);
  float sl, sy, sx;
  tv_scales(n, gx, gy, gl, weight, sl, sy, sx, channels);
  const int64_t total = n * channels * gl * gy * gx;
  hipLaunchKernelGGL(tv_bwd_kernel, dim3((unsigned)cdiv(total, kBgBlock)), dim3(kBgBlock), 0, as_stream(stream), total, gx, gy,
                     gl, grids, sl, sy, sx, v_tv, v_grids);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}
