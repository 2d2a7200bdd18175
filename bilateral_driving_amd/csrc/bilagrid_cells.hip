// Cell-aligned kernels of the bilateral-grid colour transform (K9-K12).
// Restates (file:line under /root/reference/project):
//   models/modules.py:317-346             BilateralAffineTransform.forward: one 16x16x8 grid sliced at full resolution
//   models/modules.py:494-522             MultiScaleBilateralAffineTransform.forward: per-level low-resolution slice
//   bilateral/lib_bilagrid.py:317-368     BilateralGrid.forward (F.grid_sample, trilinear, align_corners=True, border)
//   models/trainers/scene_graph.py:95-98  application of the 3x4 matrix
//
// The x / y coordinates of the slice are the pixel's own position (torch.linspace over the low-res image), so all pixels of a
// rectangle of the image lie in ONE (y, x) cell of the grid and touch only its 2 x 2 x gl nodes (<= 384 floats for the largest
// shipped grid, against 96 KB for the grid as a whole).  A job = up to kCellJobPix pixels of one cell (its full width x a few
// rows); the rectangle is found ON THE DEVICE with the pixels' own coordinate functions (bilagrid_math.h), so a pixel can never be
// attributed to a cell its arithmetic does not put it in.  Per job:
//   * the cell's nodes are staged once in LDS ([z][corner][12], plane stride 52 floats: eight planes on eight distinct groups
//     of four banks) and every pixel samples them with 24 16-byte LDS reads -- no global gathers, no 96 KB staging;
//   * backward: the grid gradient of the job is 2 x 2 x gl x 12 sums over its pixels.  The pixels of a batch are counting-sorted
//     by their guidance plane z0 into an LDS staging array ([field][slot]); then lane (corner, channel) of each wave walks a quarter
//     of the slots, four at a time (three ds_read_b128), with TWO fixed accumulator registers per plane bucket -- the weights
//     (1 - fz, fz) of planes z0, z0 + 1.  ~5 instructions per pixel and wave whatever the image looks like (the wave-wide
//     transpose-reduce per distinct cell it replaces cost 50 per pixel on a noisy image), one float atomic per touched node
//     entry and job into the caller's gradient.
// Single-scale form (`fused`): slice + 3x4 application (+ L1 / TV loss) in one launch, and the whole backward -- direct route,
// guidance route, grid gradient, clamp / sky blend / expected-depth backward -- in one launch, no scratch arrays at all.
#include <algorithm>

#include "bilagrid_ms.h"

namespace bds {

constexpr int kCellJobPix = 512;
constexpr int kNodeStride = 52;
constexpr int kStgRows = 20;                     // 12 va | 4 corner weights x (1 - fz) | 4 corner weights x fz
constexpr int kStgStride = kWave + 3 * kCellMaxGl + 4;      // 92: a wave's 64 slots + every bucket's start rounded up to a multiple of 4;
                                                 // 92 mod 64 = 28: sixteen rows start on sixteen distinct groups of four banks
constexpr int kCellWaves = kBgBlock / kWave;

struct CellSched {
  int n;
  int level[BDS_MAX_LEVELS];
  int blk_off[BDS_MAX_LEVELS + 1];
  int nblk[BDS_MAX_LEVELS];                      // jobs of the entry, rounded up to a multiple of 8 (one range per XCD)
  int ncx[BDS_MAX_LEVELS], ncy[BDS_MAX_LEVELS];  // cells per axis
  int rpj[BDS_MAX_LEVELS], nsb[BDS_MAX_LEVELS];  // rows per job, jobs per cell (the last one takes whatever rows remain)
};

// cell of index i of an axis of n pixels over g nodes (the last node folded into the cell before it: axis_cell)
BDS_HD int axis_cell_of(int i, int n, float step, int g) {
  int c;
  float f;
  axis_cell(linspace01_s(i, n, step), g, c, f);
  return c;
}
// first index in [0, n] whose cell is >= c (the cell index is monotone in the pixel index)
BDS_HD int cell_first(int c, int g, int n, float step) {
  const int nc = g > 1 ? g - 1 : 1;
  if (c <= 0) return 0;
  if (c >= nc) return n;
  int i = (int)ceilf((float)c * (float)(n - 1) / (float)(g - 1));
  i = i < 0 ? 0 : (i > n ? n : i);
  while (i > 0 && axis_cell_of(i - 1, n, step, g) >= c) i--;
  while (i < n && axis_cell_of(i, n, step, g) < c) i++;
  return i;
}

struct CellJob {
  int l, cx, cy, r0, r1, c0, c1;
};
// The job of workgroup `bid`; false (for the whole workgroup) when it has no pixels.  Contains a barrier: the four cell boundaries are
// found by four threads of the first wave (a few dozen instructions each) and shared through `bounds`; callers stage the cell's
// nodes BEFORE calling the pixel loop and need no further barrier for them (stage_nodes sits in front of this one).
__device__ __forceinline__ bool cell_job(const MsParams &p, const CellSched &sc, int bid, CellJob &J, int *bounds /* [4] shared */,
                                         float *nodes) {
  int k = 0;
  while (k + 1 < sc.n && bid >= sc.blk_off[k + 1]) k++;
  const int local = xcd_contiguous(bid - sc.blk_off[k], sc.nblk[k]);
  J.l = sc.level[k];
  const LevelDev &L = p.lv[J.l];
  const int cell = local / sc.nsb[k], sb = local - cell * sc.nsb[k];
  if (cell >= sc.ncx[k] * sc.ncy[k]) return false;
  J.cy = cell / sc.ncx[k];
  J.cx = cell - J.cy * sc.ncx[k];
  if (threadIdx.x < 4) {
    const bool xaxis = threadIdx.x < 2;
    bounds[threadIdx.x] = cell_first((xaxis ? J.cx : J.cy) + (int)(threadIdx.x & 1), xaxis ? L.gx : L.gy, xaxis ? L.Wd : L.Hd,
                                     xaxis ? L.lin_x : L.lin_y);
  }
  for (int e = threadIdx.x; e < L.gl * 48; e += kBgBlock) {   // the cell's nodes [z][corner][12]
    const int z = e / 48, r = e - z * 48, q = r / 12, ch = r - q * 12;
    const int y = min(J.cy + (q >> 1), L.gy - 1), x = min(J.cx + (q & 1), L.gx - 1);
    nodes[z * kNodeStride + r] = L.grid[((ch * L.gl + z) * L.gy + y) * L.gx + x];
  }
  __syncthreads();
  J.c0 = bounds[0]; J.c1 = bounds[1];
  const int R0 = bounds[2], R1 = bounds[3];
  J.r0 = R0 + sb * sc.rpj[k];
  J.r1 = sb == sc.nsb[k] - 1 ? R1 : min(R1, J.r0 + sc.rpj[k]);
  return J.r0 < J.r1 && J.c0 < J.c1;
}

// node (z, corner q = 2 yq + xq, channel) of the job's cell <-> element of the grid [12, gl, gy, gx]
__device__ __forceinline__ int node_element(const LevelDev &L, const CellJob &J, int z, int q, int ch) {
  const int y = min(J.cy + (q >> 1), L.gy - 1), x = min(J.cx + (q & 1), L.gx - 1);
  return ((ch * L.gl + z) * L.gy + y) * L.gx + x;
}

// a pixel's place in its cell: corner weights in x / y (its own position), plane pair and fraction in z (its guidance)
struct PixCell {
  int z0, z1;
  float fz, w00, w01, w10, w11;
  bool z_interior;
};
__device__ __forceinline__ PixCell pix_cell(const LevelDev &L, int i, int j, float gray) {
  PixCell c;
  int x0, y0;
  float fx, fy;
  axis_cell(linspace01_s(j, L.Wd, L.lin_x), L.gx, x0, fx);
  axis_cell(linspace01_s(i, L.Hd, L.lin_y), L.gy, y0, fy);
  c.w00 = (1.f - fy) * (1.f - fx); c.w01 = (1.f - fy) * fx; c.w10 = fy * (1.f - fx); c.w11 = fy * fx;
  const float iz = guide_coord(gray, L.gl, c.z_interior);
  const float z0 = floorf(iz);
  c.fz = iz - z0;
  c.z0 = (int)z0;
  c.z1 = c.z0 + 1 < L.gl ? c.z0 + 1 : L.gl - 1;
  return c;
}
// a 16-byte LDS read the compiler may not narrow: where a component of the result is unused it splits the access into ds_read_b96 +
// ds_read2_b32 (8 + 4 LDS cycles instead of 4: measured as the first cost of the backward, which is bound by the LDS pipeline)
__device__ __forceinline__ float4 lds_read4(const float4 *p) {
  float4 v = *p;
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
  return v;
}
// trilinear sample of the 12 channels from the staged nodes (slice_sample's arithmetic and order); dz12: upper minus lower plane
template <bool kDz>
__device__ __forceinline__ void slice_nodes(const float *__restrict__ nodes, const PixCell &c, float *out12, float *dz12) {
  const float4 *n0 = reinterpret_cast<const float4 *>(nodes + c.z0 * kNodeStride);
  const float4 *n1 = reinterpret_cast<const float4 *>(nodes + c.z1 * kNodeStride);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a00 = lds_read4(n0 + q), a01 = lds_read4(n0 + 3 + q), a10 = lds_read4(n0 + 6 + q), a11 = lds_read4(n0 + 9 + q);
    const float4 b00 = lds_read4(n1 + q), b01 = lds_read4(n1 + 3 + q), b10 = lds_read4(n1 + 6 + q), b11 = lds_read4(n1 + 9 + q);
    const float ax = a00.x * c.w00 + a01.x * c.w01 + a10.x * c.w10 + a11.x * c.w11, bx = b00.x * c.w00 + b01.x * c.w01 + b10.x * c.w10 + b11.x * c.w11;
    const float ay = a00.y * c.w00 + a01.y * c.w01 + a10.y * c.w10 + a11.y * c.w11, by = b00.y * c.w00 + b01.y * c.w01 + b10.y * c.w10 + b11.y * c.w11;
    const float az = a00.z * c.w00 + a01.z * c.w01 + a10.z * c.w10 + a11.z * c.w11, bz = b00.z * c.w00 + b01.z * c.w01 + b10.z * c.w10 + b11.z * c.w11;
    const float aw = a00.w * c.w00 + a01.w * c.w01 + a10.w * c.w10 + a11.w * c.w11, bw = b00.w * c.w00 + b01.w * c.w01 + b10.w * c.w10 + b11.w * c.w11;
    out12[q * 4 + 0] = ax * (1.f - c.fz) + bx * c.fz;
    out12[q * 4 + 1] = ay * (1.f - c.fz) + by * c.fz;
    out12[q * 4 + 2] = az * (1.f - c.fz) + bz * c.fz;
    out12[q * 4 + 3] = aw * (1.f - c.fz) + bw * c.fz;
    if (kDz) { dz12[q * 4 + 0] = bx - ax; dz12[q * 4 + 1] = by - ay; dz12[q * 4 + 2] = bz - az; dz12[q * 4 + 3] = bw - aw; }
  }
}

// the job's pixels, 256 at a time: thread t takes pixel base + t of the row-major rectangle
struct PixWalk {
  int row, col, ncols, nrows, q256, r256;
};
__device__ __forceinline__ PixWalk walk_begin(const CellJob &J) {
  PixWalk w;
  w.ncols = J.c1 - J.c0; w.nrows = J.r1 - J.r0;
  w.row = (int)threadIdx.x / w.ncols; w.col = (int)threadIdx.x - w.row * w.ncols;
  w.q256 = kBgBlock / w.ncols; w.r256 = kBgBlock - w.q256 * w.ncols;
  return w;
}
__device__ __forceinline__ void walk_next(PixWalk &w) {
  w.row += w.q256; w.col += w.r256;
  if (w.col >= w.ncols) { w.col -= w.ncols; w.row++; }
}

// ---- forward --------------------------------------------------------------------------------------------------------------------
// kFused = false: lo[i', j'] = slice(grid_l, low-res pixel), lg = its guidance, for the levels of the schedule.
// kFused = true : out = A(pixel) * in + b at full resolution (one level, factor 1); kTrain adds the L1 / TV loss epilogue.
template <bool kFused, bool kTrain>
__global__ __launch_bounds__(kBgBlock) void cell_fwd_kernel(MsParams p, CellSched sc, float *__restrict__ out, TrainLoss tl) {
  __shared__ __attribute__((aligned(16))) float nodes[kCellMaxGl * kNodeStride];
  __shared__ float red[kCellWaves];
  __shared__ int bounds[4];
  if (kTrain && (int)blockIdx.x >= tl.pix_blocks) {   // the TV term of the loss: one grid element per thread
    const float t = block_sum_to_thread0(tv_train_element(tl.T, (int)blockIdx.x - tl.pix_blocks, tl.v_loss), red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(tl.loss + (size_t)(blockIdx.x & (tl.loss_slots - 1)) * kLossSlotStride, t);
    return;
  }
  CellJob J;
  const bool any = cell_job(p, sc, (int)blockIdx.x, J, bounds, nodes);
  if (!any && !kTrain) return;
  const LevelDev &L = p.lv[J.l];
  float l1 = 0.f;
  if (any)
  for (PixWalk w = walk_begin(J); w.row < w.nrows; walk_next(w)) {   // (uniform trip count is not needed: no barrier inside)
    const int i = J.r0 + w.row, j = J.c0 + w.col;
    float r, g, b;
    if (kFused) {
      load_input(p, i, j, r, g, b);
    } else {
      const Tap ty = resample_tap_s(i, L.Hd, p.H, L.dn_y), tx = resample_tap_s(j, L.Wd, p.W, L.dn_x);
      lowres_colour(p, ty, tx, r, g, b);
    }
    const float gray = rgb2gray(r, g, b);
    const PixCell c = pix_cell(L, i, j, gray);
    float A[12];
    slice_nodes<false>(nodes, c, A, nullptr);
    const int idx = row_major(i, L.Wd, j);
    if (!kFused) {
      float4 *dst = reinterpret_cast<float4 *>(L.lo) + times3(idx);
      dst[0] = make_float4(A[0], A[1], A[2], A[3]);
      dst[1] = make_float4(A[4], A[5], A[6], A[7]);
      dst[2] = make_float4(A[8], A[9], A[10], A[11]);
      L.lg[idx] = gray;
    } else {
      apply_affine(A, r, g, b);
      const int p3 = times3(idx);
      out[p3] = r; out[p3 + 1] = g; out[p3 + 2] = b;
      if (p.depth_out) p.depth_out[idx] = p.rgb[(idx << 2) + 3] / fmaxf(p.alpha[idx], 1e-10f);
      if (kTrain) {   // photometric L1 of the pixel just produced + its gradient (torch: sign(0) = 0)
        const float gs = tl.v_loss * tl.inv_n;
        const float d0 = r - tl.target[p3], d1 = g - tl.target[p3 + 1], d2 = b - tl.target[p3 + 2];
        l1 += fabsf(d0) + fabsf(d1) + fabsf(d2);
        tl.v_out[p3] = d0 > 0.f ? gs : (d0 < 0.f ? -gs : 0.f);
        tl.v_out[p3 + 1] = d1 > 0.f ? gs : (d1 < 0.f ? -gs : 0.f);
        tl.v_out[p3 + 2] = d2 > 0.f ? gs : (d2 < 0.f ? -gs : 0.f);
      }
    }
  }
  if (kTrain) {
    const float t = block_sum_to_thread0(l1, red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(tl.loss + (size_t)(blockIdx.x & (tl.loss_slots - 1)) * kLossSlotStride, t * tl.inv_n);
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------------------
// d(loss)/d(lo) of a low-res pixel of an up-sampled level: y pass of the up-sampler's adjoint over the x-reduced rows R (the x pass
// ran in the full-resolution kernel, csrc/bilagrid.hip ms_apply_bwd_x_kernel); arithmetic and order of ms_lowres_bwd_kernel
__device__ __forceinline__ void adjoint_y(const MsParams &p, const LevelDev &L, int i, int j, float *va) {
  int ylo, yhi;
  // power-of-two factor f dividing the image, interior row: exactly the 2 f rows f i - f/2 .. f i + 3f/2 - 1 with the tent weights
  const bool tent = L.dn_shift > 0 && i >= 1 && i <= L.Hd - 2;
  if (tent) {
    const int f = 1 << L.dn_shift;
    ylo = f * i - (f >> 1); yhi = ylo + 2 * f - 1;
  } else {
    ylo = (int)floorf(((float)i - 0.5f) * L.dn_y - 0.5f);
    yhi = (int)ceilf(((float)i + 1.5f) * L.dn_y - 0.5f);
    ylo = ylo < 0 ? 0 : ylo;
    yhi = yhi > p.H - 1 ? p.H - 1 : yhi;
  }
#pragma unroll 4
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

// kFused = false: per low-res pixel of the schedule's levels: va = d(loss)/d(lo) (adjoint_y), grid gradient, guidance gradient -> vg.
// kFused = true : per full-res pixel of the single level: everything (see the file header); v_in / v_alpha / v_sky are final.
template <bool kFused>
__global__ __launch_bounds__(kBgBlock) void cell_bwd_kernel(MsParams p, CellSched sc, const float *__restrict__ v_out,
                                                           float *__restrict__ v_in, float *__restrict__ v_alpha,
                                                           float *__restrict__ v_sky) {
  __shared__ __attribute__((aligned(16))) float nodes[kCellMaxGl * kNodeStride];
  __shared__ __attribute__((aligned(16))) float stg_all[kCellWaves][kStgRows * kStgStride];   // per wave: [field][slot]
  __shared__ int bounds[4];
  CellJob J;
  if (!cell_job(p, sc, (int)blockIdx.x, J, bounds, nodes)) return;
  const LevelDev &L = p.lv[J.l];
  const int gl = L.gl;
  const bool want_grid = L.v_grid != nullptr;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  float *stg = stg_all[wave];
  const int lq = lane / 12, lch = lane - lq * 12;   // lanes 48..63 carry no entry (they read valid rows, their sums are dropped)
  const float4 *sva = reinterpret_cast<const float4 *>(stg + lch * kStgStride);
  const float4 *sw0 = reinterpret_cast<const float4 *>(stg + (12 + (lq & 3)) * kStgStride);   // corner weight x (1 - fz): plane z0
  const float4 *sw1 = reinterpret_cast<const float4 *>(stg + (16 + (lq & 3)) * kStgStride);   // corner weight x fz: plane z0 + 1
  float acc[kCellMaxGl][2];
#pragma unroll
  for (int b = 0; b < kCellMaxGl; b++) acc[b][0] = acc[b][1] = 0.f;
  PixWalk w = walk_begin(J);
  const int npix = w.nrows * w.ncols;
  for (int base = 0; base < npix; base += kBgBlock, walk_next(w)) {
    const bool active = w.row < w.nrows;
    float va[12];
#pragma unroll
    for (int k = 0; k < 12; k++) va[k] = 0.f;
    PixCell c;
    c.z0 = 0; c.z1 = 0; c.fz = 0.f; c.w00 = c.w01 = c.w10 = c.w11 = 0.f; c.z_interior = false;
    if (active) {
      const int i = J.r0 + w.row, j = J.c0 + w.col;
      const int idx = row_major(i, L.Wd, j);
      if (kFused) {
        float r, g, b;
        load_input(p, i, j, r, g, b);
        c = pix_cell(L, i, j, rgb2gray(r, g, b));
        float A[12], dz[12];
        slice_nodes<true>(nodes, c, A, dz);
        const int p3 = times3(idx);
        const float q0 = v_out[p3], q1 = v_out[p3 + 1], q2 = v_out[p3 + 2];
        va[0] = q0 * r; va[1] = q0 * g; va[2] = q0 * b; va[3] = q0;
        va[4] = q1 * r; va[5] = q1 * g; va[6] = q1 * b; va[7] = q1;
        va[8] = q2 * r; va[9] = q2 * g; va[10] = q2 * b; va[11] = q2;
        float v_iz = 0.f;
        if (c.z_interior) {
#pragma unroll
          for (int ch = 0; ch < 12; ch++) v_iz += va[ch] * dz[ch];
        }
        const float vgr = c.z_interior ? v_iz * (float)(gl - 1) : 0.f;
        // direct route (A^T v_out) + guidance route, then the clamp / sky blend / expected-depth backward in front of the transform
        float v[3] = {A[0] * q0 + A[4] * q1 + A[8] * q2 + vgr * kGrayR, A[1] * q0 + A[5] * q1 + A[9] * q2 + vgr * kGrayG,
                      A[2] * q0 + A[6] * q1 + A[10] * q2 + vgr * kGrayB};
        const int cs = p.cs, oc = cs == 4 ? idx << 2 : p3;
        float vaa = 0.f;
        if (p.sky) {
          const float k = 1.f - p.alpha[idx];
#pragma unroll
          for (int cc = 0; cc < 3; cc++) {
            vaa -= v[cc] * p.sky[p3 + cc];
            if (v_sky) v_sky[p3 + cc] = v[cc] * k;
            v[cc] = p.rgb[oc + cc] <= 1.f ? v[cc] : 0.f;   // torch.clamp(max=1) passes gradient at x <= 1
          }
        }
        v_in[oc] = v[0]; v_in[oc + 1] = v[1]; v_in[oc + 2] = v[2];
        if (cs == 4) {   // RGB+ED form: depth = D / clamp(alpha, min=1e-10); plus the caller's own alpha gradient
          const float a = p.alpha[idx], ac = fmaxf(a, 1e-10f);
          const float vd = p.v_depth ? p.v_depth[idx] : 0.f;
          v_in[(idx << 2) + 3] = vd / ac;
          if (p.v_alpha_in) vaa += p.v_alpha_in[idx];
          if (a >= 1e-10f) vaa -= p.rgb[(idx << 2) + 3] * vd / (ac * ac);
          if (v_alpha) v_alpha[idx] = vaa;
        } else if (p.sky && v_alpha) {
          v_alpha[idx] = vaa;
        }
      } else {
        if (L.Hd == p.H && L.Wd == p.W) {   // a level without up-sampling inside a pyramid: the full-resolution kernel left P, Q
          const float *P = L.P + times3(idx), *Q = L.Q + times3(idx);
#pragma unroll
          for (int r = 0; r < 3; r++) {
            va[r * 4 + 0] = Q[r] * P[0]; va[r * 4 + 1] = Q[r] * P[1]; va[r * 4 + 2] = Q[r] * P[2]; va[r * 4 + 3] = Q[r];
          }
        } else {
          adjoint_y(p, L, i, j, va);
        }
        c = pix_cell(L, i, j, L.lg[idx]);
        float v_iz = 0.f;
        if (c.z_interior) {
          float a12[12], dz[12];
          slice_nodes<true>(nodes, c, a12, dz);
#pragma unroll
          for (int ch = 0; ch < 12; ch++) v_iz += va[ch] * dz[ch];
        }
        L.vg[idx] = c.z_interior ? v_iz * (float)(gl - 1) : 0.f;
      }
    }
    if (!want_grid) continue;   // (uniform)
    // ---- grid gradient.  Each wave on its own (no barrier in this loop; the LDS executes one wave's accesses in order): counting
    // sort of its 64 pixels by plane z0 into its staging array, every bucket starting on a multiple of four slots ... ----
    int cnt[kCellMaxGl], off[kCellMaxGl];   // wave-uniform (scalar registers)
    int slot = 0, o = 0;
#pragma unroll
    for (int b = 0; b < kCellMaxGl; b++) {
      cnt[b] = 0; off[b] = o;
      if (b < gl) {
        const unsigned long long m = __ballot(active && c.z0 == b);
        cnt[b] = __popcll(m);
        if (c.z0 == b) slot = o + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        o += (cnt[b] + 3) & ~3;
      }
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < 12; k++) stg[k * kStgStride + slot] = va[k];
      const float gz = 1.f - c.fz;
      stg[12 * kStgStride + slot] = c.w00 * gz; stg[13 * kStgStride + slot] = c.w01 * gz;
      stg[14 * kStgStride + slot] = c.w10 * gz; stg[15 * kStgStride + slot] = c.w11 * gz;
      stg[16 * kStgStride + slot] = c.w00 * c.fz; stg[17 * kStgStride + slot] = c.w01 * c.fz;
      stg[18 * kStgStride + slot] = c.w10 * c.fz; stg[19 * kStgStride + slot] = c.w11 * c.fz;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ... then lane (corner q, channel ch) adds its entry's share of every slot, four slots per step (three 16-byte reads), into the
    // bucket's two accumulators (planes z0 and z0 + 1): two multiply-adds per slot and entry
#pragma unroll
    for (int b = 0; b < kCellMaxGl; b++) {
      if (b >= gl) break;
      if (cnt[b] == 0) continue;
      const int q0 = off[b] >> 2, nfull = cnt[b] >> 2, rem = cnt[b] & 3;
      float a0 = acc[b][0], a1 = acc[b][1];
      for (int qd = q0; qd < q0 + nfull; qd++) {
        const float4 v4 = sva[qd], g4 = sw0[qd], f4 = sw1[qd];
        a0 += g4.x * v4.x; a1 += f4.x * v4.x;
        a0 += g4.y * v4.y; a1 += f4.y * v4.y;
        a0 += g4.z * v4.z; a1 += f4.z * v4.z;
        a0 += g4.w * v4.w; a1 += f4.w * v4.w;
      }
      if (rem) {   // the bucket's last, partly filled quad: the slots behind its end hold whatever was there (never written)
        const int qd = q0 + nfull;
        const float4 v4 = sva[qd], g4 = sw0[qd], f4 = sw1[qd];
        a0 += g4.x * v4.x; a1 += f4.x * v4.x;
        a0 += rem > 1 ? g4.y * v4.y : 0.f; a1 += rem > 1 ? f4.y * v4.y : 0.f;
        a0 += rem > 2 ? g4.z * v4.z : 0.f; a1 += rem > 2 ? f4.z * v4.z : 0.f;
      }
      acc[b][0] = a0; acc[b][1] = a1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (want_grid) {
    // node z of the cell collects bucket z's first accumulator and bucket z - 1's second one (the last plane also its own second one:
    // z1 is clamped there).  Per-wave partials through the staging arrays, summed in a fixed order, one atomic per entry and job.
    if (lane < 48) {
#pragma unroll
      for (int b = 0; b < kCellMaxGl; b++) {
        if (b >= gl) break;
        float t = acc[b][0];
        if (b > 0) t += acc[b - 1][1];
        if (b == gl - 1) t += acc[b][1];
        stg[b * 48 + lane] = t;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < gl * 48; e += kBgBlock) {
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < kCellWaves; ww++) t += stg_all[ww][e];
      if (t != 0.f) {
        const int z = e / 48, r = e - z * 48, q = r / 12, ch = r - q * 12;
        atomicAdd(L.v_grid + node_element(L, J, z, q, ch), t);
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static void sched_add(CellSched &sc, const MsParams &p, int l) {
  const LevelDev &L = p.lv[l];
  const int k = sc.n++;
  sc.level[k] = l;
  const int ncx = L.gx > 1 ? L.gx - 1 : 1, ncy = L.gy > 1 ? L.gy - 1 : 1;
  sc.ncx[k] = ncx; sc.ncy[k] = ncy;
  // widest / tallest cell, with the arithmetic the device uses (a last job per cell takes whatever rows remain, so an estimate
  // that were off would cost balance, not coverage)
  int maxc = 1, maxr = 1;
  for (int c = 0; c < ncx; c++) maxc = std::max(maxc, cell_first(c + 1, L.gx, L.Wd, L.lin_x) - cell_first(c, L.gx, L.Wd, L.lin_x));
  for (int c = 0; c < ncy; c++) maxr = std::max(maxr, cell_first(c + 1, L.gy, L.Hd, L.lin_y) - cell_first(c, L.gy, L.Hd, L.lin_y));
  const int rpj = std::max(1, (kCellJobPix + maxc / 2) / maxc);
  sc.rpj[k] = rpj;
  sc.nsb[k] = (int)cdiv(maxr, rpj);
  sc.nblk[k] = (int)cdiv((int64_t)ncx * ncy * sc.nsb[k], 8) * 8;
  sc.blk_off[k + 1] = sc.blk_off[k] + sc.nblk[k];
}

int cells_lowres_fwd(const MsParams &p, unsigned mask, hipStream_t st) {
  CellSched sc{};
  for (int l = 0; l < p.nlevels; l++)
    if (mask & (1u << l)) sched_add(sc, p, l);
  if (sc.n == 0) return BDS_OK;
  TrainLoss none{};
  hipLaunchKernelGGL((cell_fwd_kernel<false, false>), dim3((unsigned)sc.blk_off[sc.n]), dim3(kBgBlock), 0, st, p, sc, nullptr, none);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

int cells_lowres_bwd(const MsParams &p, unsigned mask, hipStream_t st) {
  CellSched sc{};
  for (int l = 0; l < p.nlevels; l++)
    if (mask & (1u << l)) sched_add(sc, p, l);
  if (sc.n == 0) return BDS_OK;
  hipLaunchKernelGGL((cell_bwd_kernel<false>), dim3((unsigned)sc.blk_off[sc.n]), dim3(kBgBlock), 0, st, p, sc, nullptr, nullptr, nullptr,
                     nullptr);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

int cells_fused_fwd(const MsParams &p, float *out, const TrainLoss *train, hipStream_t st) {
  CellSched sc{};
  sched_add(sc, p, 0);
  if (train) {
    TrainLoss tl = *train;
    tl.pix_blocks = sc.blk_off[1];
    hipLaunchKernelGGL((cell_fwd_kernel<true, true>), dim3((unsigned)(tl.pix_blocks + tl.tv_blocks)), dim3(kBgBlock), 0, st, p, sc, out, tl);
  } else {
    TrainLoss none{};
    hipLaunchKernelGGL((cell_fwd_kernel<true, false>), dim3((unsigned)sc.blk_off[1]), dim3(kBgBlock), 0, st, p, sc, out, none);
  }
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

int cells_fused_bwd(const MsParams &p, const float *v_out, float *v_in, float *v_alpha, float *v_sky, hipStream_t st) {
  CellSched sc{};
  sched_add(sc, p, 0);
  hipLaunchKernelGGL((cell_bwd_kernel<true>), dim3((unsigned)sc.blk_off[1]), dim3(kBgBlock), 0, st, p, sc, v_out, v_in, v_alpha, v_sky);
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

}  // namespace bds
