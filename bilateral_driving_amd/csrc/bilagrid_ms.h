// Shared by the bilateral-grid translation units (csrc/bilagrid.hip: general kernels + host side; csrc/bilagrid_cells.hip: the
// cell-aligned kernels): launch parameter blocks, index helpers, the input-colour prologue (clamp + sky blend) and the loss epilogue.
#pragma once
#include "bds_common.h"
#include "bilagrid_math.h"

namespace bds {

constexpr int kBgBlock = 256;

struct LevelDev {
  const float *grid;   // [n_avg,12,gl,gy,gx]
  float *v_grid;
  float *lo;           // [Hd*Wd,12] low-res affine maps
  float *lg;           // [Hd*Wd] guidance (gray of the down-sampled input colour) of the low-res pixels, kept by the forward
  float *P;            // [H*W,3] input colour of this level                  (bwd scratch)
  float *Q;            // [H*W,3] gradient w.r.t. this level's output          (bwd scratch)
  float *R;            // [H*Wd,12] x-reduced adjoint of the up-sampler        (bwd scratch)
  float *vg;           // [Hd*Wd] gradient w.r.t. the low-res guidance (gray)  (bwd scratch)
  float *aff_out;      // optional [H*W,12]
  int gx, gy, gl, factor, n_avg, Hd, Wd;
  // resampling scales as torch forms them, divided ONCE on the host: up = low / full (taps of the up-sampler), dn = full / low
  float up_x, up_y, dn_x, dn_y;
  float lin_x, lin_y;  // 1 / (Wd - 1), 1 / (Hd - 1): step of torch.linspace(0, 1, n) over the low-res columns / rows
  uint32_t magic_wd;   // ceil(2^32 / Wd): row / column of a low-res index without an integer division (fast_divmod)
  int dn_shift;        // log2(factor) when the factor is a power of two >= 2 that divides H and W (else 0): the bilinear down-sampler
                       // then reads exactly the central 2 x 2 pixels of every factor x factor block with weight 1/4 each
};
struct MsParams {
  int nlevels, H, W;
  uint32_t magic_w;        // ceil(2^32 / W)
  int cs;                  // floats per pixel of `rgb` and of the returned colour gradient: 3, or 4 in the RGB+ED form
  const float *rgb, *alpha, *sky;
  float *depth_out;        // RGB+ED form: [H*W] expected depth = rgb[.,3] / max(alpha, 1e-10)
  const float *v_depth;    // RGB+ED form, backward: gradient of that depth (may be null)
  const float *v_alpha_in; // RGB+ED form, backward: gradient arriving at alpha from the caller (may be null)
  LevelDev lv[BDS_MAX_LEVELS];
};

// n / d and n % d for n < 2^31, d < 2^31 from magic = ceil(2^32 / d): the estimate is exact or one too large (a 64-bit division
// by a launch constant costs ~60 vector instructions per pixel in kernels that are bound by instruction issue)
__device__ __forceinline__ void fast_divmod(uint32_t n, uint32_t d, uint32_t magic, int &q, int &r) {
  uint32_t qq = d == 1u ? n : __umulhi(n, magic);
  int rr = (int)(n - qq * d);
  if (rr < 0) { qq--; rr += (int)d; }
  q = (int)qq; r = rr;
}
// 32-bit element offsets for the image-sized arrays (ms_fill requires 12 H W < 2^31): a 64-bit multiply-add per tap address
// (v_mad_u64_u32, quarter rate) was ~30 % of the full-resolution kernels' issue time.  Rows / columns are < 2^23, so row * width is
// one full-rate v_mul_i32_i24; the small constant factors are shift-adds.
__device__ __forceinline__ int row_major(int row, int width, int col) { return __mul24(row, width) + col; }
__device__ __forceinline__ int times3(int v) { return v + (v << 1); }
static inline uint32_t divmod_magic(int d) { return d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + (uint64_t)d - 1) / (uint64_t)d); }

// Several levels in ONE launch: workgroup ranges per level (the levels are independent, each alone
// under-fills the chip, and a launch boundary costs ~1.5-2 us).
struct LevelSched {
  int n;                               // entries
  int level[BDS_MAX_LEVELS];           // level index of entry k
  int blk_off[BDS_MAX_LEVELS + 1];     // workgroups [blk_off[k], blk_off[k+1]) belong to entry k
  int nblk[BDS_MAX_LEVELS];            // = blk_off[k+1] - blk_off[k]
  long long part_off[BDS_MAX_LEVELS];  // float offset of the entry's partial-grid region
};
__device__ __forceinline__ int sched_find(const LevelSched &s, int bid, int &local) {
  int k = 0;
  while (k + 1 < s.n && bid >= s.blk_off[k + 1]) k++;
  local = bid - s.blk_off[k];
  return k;
}

// input colour of the transform at pixel (y,x): clamp + sky blend fused when sky != null
__device__ __forceinline__ void load_input(const MsParams &p, int y, int x, float &r, float &g, float &b) {
  const int o = row_major(y, p.W, x);
  const int oc = p.cs == 4 ? o << 2 : times3(o), o3 = times3(o);
  r = p.rgb[oc]; g = p.rgb[oc + 1]; b = p.rgb[oc + 2];
  if (p.sky) {
    const float k = 1.f - p.alpha[o];
    r = fminf(r, 1.f) + p.sky[o3] * k;
    g = fminf(g, 1.f) + p.sky[o3 + 1] * k;
    b = fminf(b, 1.f) + p.sky[o3 + 2] * k;
  }
}

__device__ __forceinline__ void lowres_colour(const MsParams &p, const Tap &ty, const Tap &tx, float &r, float &g, float &b) {
  float r00, g00, b00, r01, g01, b01, r10, g10, b10, r11, g11, b11;
  load_input(p, ty.i0, tx.i0, r00, g00, b00);
  load_input(p, ty.i0, tx.i1, r01, g01, b01);
  load_input(p, ty.i1, tx.i0, r10, g10, b10);
  load_input(p, ty.i1, tx.i1, r11, g11, b11);
  const float wx = tx.w1, wy = ty.w1;
  r = (r00 * (1.f - wx) + r01 * wx) * (1.f - wy) + (r10 * (1.f - wx) + r11 * wx) * wy;
  g = (g00 * (1.f - wx) + g01 * wx) * (1.f - wy) + (g10 * (1.f - wx) + g11 * wx) * wy;
  b = (b00 * (1.f - wx) + b01 * wx) * (1.f - wy) + (b10 * (1.f - wx) + b11 * wx) * wy;
}

// ---- TV of several grid pyramids' levels in ONE launch each way (the levels are tiny: a launch costs more than a level) ----
struct TvLevels {
  int n;
  const float *x[BDS_MAX_LEVELS];
  float *v_x[BDS_MAX_LEVELS];
  long long total[BDS_MAX_LEVELS];
  int gx[BDS_MAX_LEVELS], gy[BDS_MAX_LEVELS], gl[BDS_MAX_LEVELS];
  float sl[BDS_MAX_LEVELS], sy[BDS_MAX_LEVELS], sx[BDS_MAX_LEVELS];
  int blk_off[BDS_MAX_LEVELS + 1];
};

// value and gradient of the TV term at the element (level by workgroup, element by thread) that workgroup `bid` of a
// T.blk_off[T.n]-workgroup range owns: returns its share of the value, ADDS v_loss * d(TV)/d(element) to the level's gradient slice
__device__ __forceinline__ float tv_train_element(const TvLevels &L, int bid, float v_loss) {
  int k = 0;
  while (k + 1 < L.n && bid >= L.blk_off[k + 1]) k++;
  const int64_t e = (int64_t)(bid - L.blk_off[k]) * kBgBlock + threadIdx.x;
  float acc = 0.f;
  if (e < L.total[k]) {
    const int gx = L.gx[k], gy = L.gy[k], gl = L.gl[k];
    const float *x = L.x[k];
    const int ix = (int)(e % gx), iy = (int)((e / gx) % gy), il = (int)((e / ((int64_t)gx * gy)) % gl);
    const int64_t sl = (int64_t)gx * gy;
    const float v = x[e];
    float g = 0.f;
    if (ix > 0) { const float d = v - x[e - 1]; acc += d * d * L.sx[k]; g += 2.f * d * L.sx[k]; }
    if (ix < gx - 1) g -= 2.f * (x[e + 1] - v) * L.sx[k];
    if (iy > 0) { const float d = v - x[e - gx]; acc += d * d * L.sy[k]; g += 2.f * d * L.sy[k]; }
    if (iy < gy - 1) g -= 2.f * (x[e + gx] - v) * L.sy[k];
    if (il > 0) { const float d = v - x[e - sl]; acc += d * d * L.sl[k]; g += 2.f * d * L.sl[k]; }
    if (il < gl - 1) g -= 2.f * (x[e + sl] - v) * L.sl[k];
    if (L.v_x[k]) atomicAdd(L.v_x[k] + e, g * v_loss);
  }
  return acc;
}

// the training loss folded into the full-resolution forward (bds_bilagrid_ms_ed_train_fwd): L1 against `target` over the pixels this
// launch produces, TV of the grids by tv_blocks extra workgroups behind the pix_blocks pixel workgroups
constexpr int kLossSlotStride = BDS_LOSS_SLOT_STRIDE;   // floats between two slots: every slot in a 256-byte segment of its own
struct TrainLoss {
  const float *target;   // [H,W,3]
  float *v_out;          // [H,W,3]  sign(out - target) * v_loss / (3 H W)
  float *loss;           // [loss_slots * kLossSlotStride], zeroed by the caller: slot (workgroup % loss_slots) += its share of
                         // mean|out - target| + TV terms.  (8100 float atomics on ONE address cost 65 us at the end of the launch.)
  int loss_slots;        // power of two
  float inv_n, v_loss;
  int pix_blocks, tv_blocks;
  TvLevels T;
};

__device__ __forceinline__ float block_sum_to_thread0(float acc, float *red /* [kBgBlock / kWave] shared */) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = acc;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < kBgBlock / kWave; w++) t += red[w];
  return t;
}

// ---- cell-aligned kernels (csrc/bilagrid_cells.hip) -------------------------------------------------------------------------
// A level qualifies when it has ONE grid (n_avg == 1) with gl <= kCellMaxGl planes; its low-resolution pixels are then processed
// in jobs that lie inside ONE (y, x) cell of the grid.
constexpr int kCellMaxGl = 8;
inline bool cells_level_ok(const LevelDev &L) { return L.n_avg == 1 && L.gl >= 1 && L.gl <= kCellMaxGl; }
// whole transform in one launch each way: one level at full resolution (BilateralAffineTransform, models/modules.py:317-346)
inline bool cells_fused_ok(const MsParams &p) {
  return p.nlevels == 1 && p.lv[0].Hd == p.H && p.lv[0].Wd == p.W && cells_level_ok(p.lv[0]) && p.lv[0].aff_out == nullptr;
}
// low-resolution stage of the levels in `mask` (bit l): slice -> lo, lg | y pass of the up-sampler adjoint + slice vjp -> v_grid, vg
int cells_lowres_fwd(const MsParams &p, unsigned mask, hipStream_t st);
int cells_lowres_bwd(const MsParams &p, unsigned mask, hipStream_t st);
// single full-resolution level: slice + apply (+ L1 / TV loss) | the whole backward
int cells_fused_fwd(const MsParams &p, float *out, const TrainLoss *train, hipStream_t st);
int cells_fused_bwd(const MsParams &p, const float *v_out, float *v_in, float *v_alpha, float *v_sky, hipStream_t st);
// pyramid forward in one pass over the image (csrc/bilagrid_tile.hip): every factor a power of two >= 2 dividing the image
bool tile_fwd_ok(const MsParams &p);
int tile_fwd(const MsParams &p, float *out, const TrainLoss *train, hipStream_t st);

}  // namespace bds
