// Multi-scale bilateral transform, forward, as ONE pass over the image (K9-K11).
// Restates (file:line under /root/reference/project):
//   models/modules.py:494-522,409-420     MultiScaleBilateralAffineTransform.forward: per-level low-res slice, bilinear up-sampling
//   models/trainers/scene_graph.py:112-117,292-294   sequential composition, sky blend in front
//
// The two-stage form (low-resolution slice of every level -> maps in memory -> full-resolution compose) reads the input image once
// per stage and level (80 + 96 B/pixel moved for 44 algorithmic).  Here a workgroup owns a kTileW x kTileH tile of the image:
//   A. it evaluates, for every level, the low-resolution maps of the (TH/f + 2) x (TW/f + 2) low-res pixels its own pixels
//      interpolate between -- guidance from the four central input pixels of each f x f block (formed ONCE for all levels of the same
//      factor), trilinear sample from an LDS copy of the <= 3 x 3 x gl grid nodes the tile can touch -- into LDS (48 B per low-res
//      pixel; 26.7 KB for the shipped pyramid);
//   B. a thread composes FOUR consecutive rows of one column: the x pass of the up-sampler once per low-res row those pixels touch
//      (their row taps are wave-uniform), the y pass and the 3x4 apply per pixel, level after level; the expected-depth normalise /
//      clamp / sky blend / L1 + TV loss ride along as in the two-stage form.
// The ring of low-res pixels around a tile is evaluated by its neighbours too (+45 % low-res work, which is 0.375 of a pixel's), in
// exchange no map is ever read back from memory by the forward; the tile's OWN low-res pixels are still written out (maps + guidance)
// for the backward.  Taken when every level's factor is a power of two >= 2 that divides the image and the level has one grid with
// gl <= 8; everything else runs the two-stage form (csrc/bilagrid.hip, csrc/bilagrid_cells.hip).  Same values either way.
#include "bilagrid_ms.h"

namespace bds {

constexpr int kTileW = 64, kTileH = 16;
constexpr int kSubNodes = 3;                              // nodes per axis of the sub-grid a tile may touch (two cells)
constexpr int kSubPlane = kSubNodes * kSubNodes * 12;     // floats per guidance plane of a sub-grid copy

struct TileGeom {
  int tiles_x, tiles_y;
  int lo_off[BDS_MAX_LEVELS];      // float offset of the level's low-res block in the dynamic LDS
  int nrows[BDS_MAX_LEVELS], ncols[BDS_MAX_LEVELS];   // block extent (before clamping at the image border)
  int nodes_off[BDS_MAX_LEVELS];   // float offset of the level's sub-grid copy [gl][3][3][12]
  // levels of equal factor have the same low-res pixels: their down-sampled input colour / guidance is formed once per group
  int ngroups, g_nlev[BDS_MAX_LEVELS], g_lev[BDS_MAX_LEVELS][BDS_MAX_LEVELS];
  float g_inv_ncols[BDS_MAX_LEVELS];   // 1 / ncols of the group's block
};

// trilinear sample from the tile's sub-grid copy sub[z][ly][lx][12] (origin node (yn0, xn0)); slice_sample's arithmetic and order
__device__ __forceinline__ void slice_sub(const float *__restrict__ sub, int xn0, int yn0, const Cell &c, float *out12) {
  const int o00 = ((c.y0 - yn0) * kSubNodes + (c.x0 - xn0)) * 3, o01 = ((c.y0 - yn0) * kSubNodes + (c.x1 - xn0)) * 3;
  const int o10 = ((c.y1 - yn0) * kSubNodes + (c.x0 - xn0)) * 3, o11 = ((c.y1 - yn0) * kSubNodes + (c.x1 - xn0)) * 3;
  const float w00 = (1.f - c.fy) * (1.f - c.fx), w01 = (1.f - c.fy) * c.fx, w10 = c.fy * (1.f - c.fx), w11 = c.fy * c.fx;
  const float4 *g0 = reinterpret_cast<const float4 *>(sub) + c.z0 * (kSubNodes * kSubNodes * 3);
  const float4 *g1 = reinterpret_cast<const float4 *>(sub) + c.z1 * (kSubNodes * kSubNodes * 3);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a00 = g0[o00 + q], a01 = g0[o01 + q], a10 = g0[o10 + q], a11 = g0[o11 + q];
    const float4 b00 = g1[o00 + q], b01 = g1[o01 + q], b10 = g1[o10 + q], b11 = g1[o11 + q];
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

// x pass of the up-sampler for one low-res row of the tile's block: out[c] = lo[row][i0][c] (1 - wx) + lo[row][i1][c] wx
__device__ __forceinline__ void xlerp_row(const float4 *__restrict__ lo4, int row_base, int c0, int c1, float wx, float *out12) {
  const float4 *s0 = lo4 + times3(row_base + c0), *s1 = lo4 + times3(row_base + c1);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float4 a = s0[q], b = s1[q];
    out12[q * 4 + 0] = a.x * (1.f - wx) + b.x * wx;
    out12[q * 4 + 1] = a.y * (1.f - wx) + b.y * wx;
    out12[q * 4 + 2] = a.z * (1.f - wx) + b.z * wx;
    out12[q * 4 + 3] = a.w * (1.f - wx) + b.w * wx;
  }
}

template <int NL, bool kTrain>
__global__ __launch_bounds__(kBgBlock) void ms_tile_fwd_kernel(MsParams p, TileGeom G, float *__restrict__ out, TrainLoss tl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[kBgBlock / kWave];
  __shared__ int sub_org[BDS_MAX_LEVELS][4];   // per level: xn0, yn0, nodes in x, nodes in y (0 = sample the grid in global memory)
  if (kTrain && (int)blockIdx.x >= tl.pix_blocks) {   // the TV term of the loss: one grid element per thread
    const float t = block_sum_to_thread0(tv_train_element(tl.T, (int)blockIdx.x - tl.pix_blocks, tl.v_loss), red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(tl.loss + (size_t)(blockIdx.x & (tl.loss_slots - 1)) * kLossSlotStride, t);
    return;
  }
  const int tile = xcd_contiguous((int)blockIdx.x, G.tiles_x * G.tiles_y);   // one band of tile rows per XCD
  const int ty_ = tile / G.tiles_x, tx_ = tile - ty_ * G.tiles_x;
  const int Y0 = ty_ * kTileH, X0 = tx_ * kTileW;
  // ---- the sub-grid of every level: nodes from the first low-res pixel's lower node to the last one's upper node ----------------
  if ((int)threadIdx.x < p.nlevels) {
    const LevelDev &L = p.lv[threadIdx.x];
    const int f = L.factor;
    const int r0 = max(Y0 / f - 1, 0), r1 = min((Y0 + kTileH) / f, L.Hd - 1), c0 = max(X0 / f - 1, 0), c1 = min((X0 + kTileW) / f, L.Wd - 1);
    const Cell a = slice_cell(linspace01_s(c0, L.Wd, L.lin_x), linspace01_s(r0, L.Hd, L.lin_y), 0.f, L.gx, L.gy, L.gl);
    const Cell b = slice_cell(linspace01_s(c1, L.Wd, L.lin_x), linspace01_s(r1, L.Hd, L.lin_y), 0.f, L.gx, L.gy, L.gl);
    const int nx = b.x1 - a.x0 + 1, ny = b.y1 - a.y0 + 1;
    const bool fits = nx <= kSubNodes && ny <= kSubNodes;
    sub_org[threadIdx.x][0] = a.x0; sub_org[threadIdx.x][1] = a.y0;
    sub_org[threadIdx.x][2] = fits ? nx : 0; sub_org[threadIdx.x][3] = fits ? ny : 0;
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < NL; l++) {
    if (l >= p.nlevels) break;
    const LevelDev &L = p.lv[l];
    const int xn0 = sub_org[l][0], yn0 = sub_org[l][1], nx = sub_org[l][2], ny = sub_org[l][3];
    const int cnt = L.gl * ny * nx * 12;
    for (int e = threadIdx.x; e < cnt; e += kBgBlock) {   // (channel innermost in LDS, x innermost in memory: 12 strided streams)
      const int ch = e % 12, r = e / 12, lx = r % nx, r2 = r / nx, ly = r2 % ny, z = r2 / ny;
      lds[G.nodes_off[l] + ((z * kSubNodes + ly) * kSubNodes + lx) * 12 + ch] = L.grid[((ch * L.gl + z) * L.gy + (yn0 + ly)) * L.gx + (xn0 + lx)];
    }
  }
  __syncthreads();
  // ---- A: low-res maps of the tile's block of every level; group by group (one factor each: the loop and every level parameter
  //         inside it are wave-uniform), the down-sampled colour and the guidance of a low-res pixel formed once per group ------------
  for (int gi = 0; gi < G.ngroups; gi++) {
    const int l0 = G.g_lev[gi][0];
    const LevelDev &L0 = p.lv[l0];
    const int f = L0.factor, nc = G.ncols[l0], nitems = G.nrows[l0] * nc;
    const float inv_nc = G.g_inv_ncols[gi];
    for (int k = threadIdx.x; k < nitems; k += kBgBlock) {
      const int lr = (int)(((float)k + 0.5f) * inv_nc), lc = k - lr * nc;
      const int i = Y0 / f - 1 + lr, j = X0 / f - 1 + lc;
      if (i < 0 || j < 0 || i >= L0.Hd || j >= L0.Wd) continue;
      const Tap ty = resample_tap_s(i, L0.Hd, p.H, L0.dn_y), tx = resample_tap_s(j, L0.Wd, p.W, L0.dn_x);
      float r, g, b;
      lowres_colour(p, ty, tx, r, g, b);
      const float gray = rgb2gray(r, g, b);
      const float x01 = linspace01_s(j, L0.Wd, L0.lin_x), y01 = linspace01_s(i, L0.Hd, L0.lin_y);
      const bool own = lr >= 1 && lr <= kTileH / f && lc >= 1 && lc <= kTileW / f;   // the tile's own low-res pixels: kept for the backward
      const int idx = row_major(i, L0.Wd, j);
      for (int u = 0; u < G.g_nlev[gi]; u++) {
        const int l = G.g_lev[gi][u];
        const LevelDev &L = p.lv[l];
        const Cell c = slice_cell(x01, y01, gray, L.gx, L.gy, L.gl);
        float A[12];
        if (sub_org[l][2] > 0) slice_sub(lds + G.nodes_off[l], sub_org[l][0], sub_org[l][1], c, A);
        else slice_sample(L.grid, L.gx, L.gy, L.gl, c, A, nullptr);
        float4 *dst = reinterpret_cast<float4 *>(lds + G.lo_off[l]) + times3(k);
        const float4 a0 = make_float4(A[0], A[1], A[2], A[3]), a1 = make_float4(A[4], A[5], A[6], A[7]), a2 = make_float4(A[8], A[9], A[10], A[11]);
        dst[0] = a0; dst[1] = a1; dst[2] = a2;
        if (own) {
          float4 *gdst = reinterpret_cast<float4 *>(L.lo) + times3(idx);
          gdst[0] = a0; gdst[1] = a1; gdst[2] = a2;
          L.lg[idx] = gray;
        }
      }
    }
  }
  __syncthreads();
  // ---- B: compose + apply.  A thread owns FOUR consecutive rows of one column (a wave: 64 columns x 4 rows): the rows of a wave and
  //         hence their low-res row taps are wave-uniform, so the x pass of the up-sampler is done once per low-res ROW the four pixels
  //         touch (3 at factor 4, 4 at factor 2, 2 at factor 8) instead of twice per pixel, and every pixel adds its y pass -------------
  const int lx = (int)threadIdx.x & (kTileW - 1);
  const int rg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int j = X0 + lx, ib = Y0 + 4 * rg;
  float l1 = 0.f;
  if (j < p.W && ib < p.H) {
    float r[4], g[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      r[q] = g[q] = b[q] = 0.f;
      if (ib + q < p.H) load_input(p, ib + q, j, r[q], g[q], b[q]);
    }
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l >= p.nlevels) break;
      const LevelDev &L = p.lv[l];
      const int f = L.factor, nc = G.ncols[l];
      const Tap tx = resample_tap_s(j, p.W, L.Wd, L.up_x);
      const float4 *lo4 = reinterpret_cast<const float4 *>(lds + G.lo_off[l]);
      const int rb = Y0 / f - 1, cb = X0 / f - 1;
      const int c0 = tx.i0 - cb, c1 = tx.i1 - cb;
      float R0[12], R1[12];
      int t0 = -1, t1 = -1;   // low-res rows (block-local) held in R0 / R1
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (ib + q >= p.H) break;
        const Tap ty = resample_tap_s(ib + q, p.H, L.Hd, L.up_y);   // (wave-uniform)
        const int a = ty.i0 - rb, bb = ty.i1 - rb;
        if (a != t0) {
          if (a == t1) {
#pragma unroll
            for (int c = 0; c < 12; c++) R0[c] = R1[c];
          } else {
            xlerp_row(lo4, a * nc, c0, c1, tx.w1, R0);
          }
          t0 = a;
        }
        if (bb != t1) {
          if (bb == t0) {
#pragma unroll
            for (int c = 0; c < 12; c++) R1[c] = R0[c];
          } else {
            xlerp_row(lo4, bb * nc, c0, c1, tx.w1, R1);
          }
          t1 = bb;
        }
        const float wy = ty.w1;
        float A[12];
#pragma unroll
        for (int c = 0; c < 12; c++) A[c] = R0[c] * (1.f - wy) + R1[c] * wy;
        apply_affine(A, r[q], g[q], b[q]);
      }
    }
    const float gs = kTrain ? tl.v_loss * tl.inv_n : 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (ib + q >= p.H) break;
      const int pix = row_major(ib + q, p.W, j), p3 = times3(pix);
      out[p3] = r[q]; out[p3 + 1] = g[q]; out[p3 + 2] = b[q];
      if (p.depth_out) p.depth_out[pix] = p.rgb[(pix << 2) + 3] / fmaxf(p.alpha[pix], 1e-10f);
      if (kTrain) {   // photometric L1 of the pixel just produced + its gradient (torch: sign(0) = 0)
        const float d0 = r[q] - tl.target[p3], d1 = g[q] - tl.target[p3 + 1], d2 = b[q] - tl.target[p3 + 2];
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

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool tile_fwd_ok(const MsParams &p) {
  if (p.nlevels < 1 || p.nlevels > 4) return false;
  for (int l = 0; l < p.nlevels; l++) {
    const LevelDev &L = p.lv[l];
    if (L.dn_shift < 1 || L.factor > 8 || !cells_level_ok(L) || L.aff_out) return false;
  }
  return true;
}

int tile_fwd(const MsParams &p, float *out, const TrainLoss *train, hipStream_t st) {
  TileGeom G{};
  G.tiles_x = (int)cdiv(p.W, kTileW); G.tiles_y = (int)cdiv(p.H, kTileH);
  int off = 0;
  for (int l = 0; l < p.nlevels; l++) {
    const int f = p.lv[l].factor;
    G.nrows[l] = kTileH / f + 2; G.ncols[l] = kTileW / f + 2;
    G.lo_off[l] = off;
    off += G.nrows[l] * G.ncols[l] * 12;
    int gi = 0;
    while (gi < G.ngroups && p.lv[G.g_lev[gi][0]].factor != f) gi++;
    if (gi == G.ngroups) { G.ngroups++; G.g_nlev[gi] = 0; G.g_inv_ncols[gi] = 1.f / (float)G.ncols[l]; }
    G.g_lev[gi][G.g_nlev[gi]++] = l;
  }
  for (int l = 0; l < p.nlevels; l++) {
    G.nodes_off[l] = off;
    off += p.lv[l].gl * kSubPlane;
  }
  const size_t lds_bytes = sizeof(float) * (size_t)off;
  const int tiles = G.tiles_x * G.tiles_y;
  TrainLoss tl{};
  if (train) { tl = *train; tl.pix_blocks = tiles; }
  const dim3 grid((unsigned)(tiles + (train ? tl.tv_blocks : 0))), block(kBgBlock);
#define BDS_TILE_FWD(n, t)                                                                                                       \
  do {                                                                                                                           \
    if (lds_bytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(&ms_tile_fwd_kernel<n, t>),                 \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) \
      return BDS_ELAUNCH;                                                                                                        \
    hipLaunchKernelGGL((ms_tile_fwd_kernel<n, t>), grid, block, lds_bytes, st, p, G, out, tl);                                   \
  } while (0)
  if (train) {
    switch (p.nlevels) {
      case 1: BDS_TILE_FWD(1, true); break;
      case 2: BDS_TILE_FWD(2, true); break;
      case 3: BDS_TILE_FWD(3, true); break;
      default: BDS_TILE_FWD(4, true); break;
    }
  } else {
    switch (p.nlevels) {
      case 1: BDS_TILE_FWD(1, false); break;
      case 2: BDS_TILE_FWD(2, false); break;
      case 3: BDS_TILE_FWD(3, false); break;
      default: BDS_TILE_FWD(4, false); break;
    }
  }
#undef BDS_TILE_FWD
  BDS_LAUNCH_CHECK();
  return BDS_OK;
}

}  // namespace bds
