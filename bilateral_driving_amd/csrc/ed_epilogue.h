// The last stage of the colour transform's backward -- guidance route added to the direct route, clamp(max=1) / sky blend /
// expected-depth backward (csrc/bilagrid.hip ms_guidance_blend_bwd_kernel; models/trainers/base.py:414-419,
// trainers/scene_graph.py:292-294) -- in a form the compositor's backward can run per pixel while it loads its tile: the per-pixel
// result (v_render [4], v_alpha) is then never written to memory and read back, and one launch over the image disappears.
#pragma once
#include "bds_common.h"

namespace bds {

struct EdEpilogue {
  int nlevels, W;
  const float *vg[BDS_MAX_LEVELS];   // [Hd*Wd] gradient w.r.t. the low-res guidance of every level (csrc/bilagrid_cells.hip)
  int Wd[BDS_MAX_LEVELS];
  int shift[BDS_MAX_LEVELS];         // log2(factor) of a dividing power-of-two factor; 0: the level works at full resolution
  const float *v_direct;             // [H*W,4] direct-route gradient of the transform's input colour (channels 0-2)
  const float *render;               // [H*W,4] the compositor's forward output (colour before the clamp | D)
  const float *sky;                  // [H*W,3] or null
  const float *v_depth;              // [H*W] or null: gradient of the expected depth D / max(alpha, 1e-10)
  const float *v_alpha_in;           // [H*W] or null: the caller's own gradient of alpha
  float *v_sky;                      // [H*W,3] or null (out)
};

// v_render[0..3] and v_alpha of pixel (y, x) with opacity `alpha`; writes v_sky.  Arithmetic and order of ms_guidance_blend_bwd_kernel.
__device__ __forceinline__ void ed_epilogue_pixel(const EdEpilogue &e, int y, int x, int pix, float alpha, float *vr4, float &v_alpha) {
  constexpr float kGrayR = 0.299f, kGrayG = 0.587f, kGrayB = 0.114f;
  float vg = 0.f;
#pragma unroll
  for (int l = 0; l < BDS_MAX_LEVELS; l++) {
    if (l >= e.nlevels) break;
    const int sh = e.shift[l];
    if (sh == 0) { vg += e.vg[l][pix]; continue; }
    // low-res pixel (i, j) was sampled from the two central rows / columns of its f x f block with weights 1/2, 1/2
    const int f = 1 << sh, h = f >> 1, fy = y & (f - 1), fx = x & (f - 1);
    if ((fy == h - 1 || fy == h) && (fx == h - 1 || fx == h)) vg += (0.5f * 0.5f) * e.vg[l][(y >> sh) * e.Wd[l] + (x >> sh)];
  }
  const float4 d = reinterpret_cast<const float4 *>(e.v_direct)[pix];
  const float4 r = reinterpret_cast<const float4 *>(e.render)[pix];
  float v[3] = {d.x + vg * kGrayR, d.y + vg * kGrayG, d.z + vg * kGrayB};
  const float rgb[3] = {r.x, r.y, r.z};
  float va = 0.f;
  if (e.sky) {
    const float k = 1.f - alpha;
    const int p3 = pix * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      va -= v[c] * e.sky[p3 + c];
      if (e.v_sky) e.v_sky[p3 + c] = v[c] * k;
      v[c] = rgb[c] <= 1.f ? v[c] : 0.f;   // torch.clamp(max=1) passes gradient at x <= 1
    }
  }
  const float ac = fmaxf(alpha, 1e-10f);
  const float vd = e.v_depth ? e.v_depth[pix] : 0.f;
  if (e.v_alpha_in) va += e.v_alpha_in[pix];
  if (alpha >= 1e-10f) va -= r.w * vd / (ac * ac);   // clamp(min) passes the gradient where alpha >= 1e-10
  vr4[0] = v[0]; vr4[1] = v[1]; vr4[2] = v[2]; vr4[3] = vd / ac;
  v_alpha = va;
}

// fills `e` from the transform's level list and workspace (csrc/bilagrid.hip); BDS_EINVAL when the configuration cannot defer
int ed_epilogue_fill(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, void *ws, size_t ws_bytes, EdEpilogue *e);

}  // namespace bds
