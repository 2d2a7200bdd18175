// Per-element math of the bilateral-grid half (resampler taps, slice coordinates, trilinear
// weights).  __host__ __device__ so that tests/ can check the device formulas on the host.
//
// Restates (file:line under /root/reference/project):
//   bilateral/lib_bilagrid.py:346-363   BilateralGrid.forward: (xy-0.5)*2, gray*2-1, F.grid_sample
//                                       (bilinear = trilinear on 5-D, align_corners=True, border)
//   models/modules.py:494-504           get_sample_grid: bilinear down-sample, linspace(0,1,.) coords
//   models/modules.py:409-420           fill_matrix_res: bilinear up-sample (align_corners=False)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#ifndef BDS_HD
#define BDS_HD __host__ __device__ __forceinline__
#endif

namespace bds {

constexpr float kGrayR = 0.299f, kGrayG = 0.587f, kGrayB = 0.114f;  // lib_bilagrid.py:287

// One axis of F.interpolate(mode="bilinear", align_corners=False): destination index -> two source
// taps and the weight of the second one.
struct Tap {
  int i0, i1;
  float w1;
};
// `scale` = (float)in_size / (float)out_size, evaluated once by the caller (the host, for the kernels: an IEEE division costs a dozen
// vector instructions per call and the sizes are launch constants; same value, same taps)
BDS_HD Tap resample_tap_s(int dst, int out_size, int in_size, float scale) {
#pragma clang fp contract(off)  // keep torch's (unfused) source-index arithmetic: floor() decisions depend on it
  Tap t;
  if (out_size == in_size) { t.i0 = t.i1 = dst; t.w1 = 0.f; return t; }
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  float w = src - (float)t.i0;
  t.w1 = w < 0.f ? 0.f : (w > 1.f ? 1.f : w);
  return t;
}
BDS_HD Tap resample_tap(int dst, int out_size, int in_size) {
  return resample_tap_s(dst, out_size, in_size, (float)in_size / (float)out_size);
}

// torch.linspace(0, 1, n)[i] in float32 (symmetric evaluation around the midpoint)
BDS_HD float linspace01(int i, int n) {
  // The upper half is a fused multiply-add, as in torch's vectorised CPU kernel and its CUDA kernel
  // (torch's scalar loop tail rounds twice: <= 1 ulp apart, only in the non-differentiated x/y).
  if (n <= 1) return 0.f;
  const float step = 1.0f / (float)(n - 1);
  return i < n / 2 ? step * (float)i : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
// the same with step = 1.0f / (float)(n - 1) divided once by the caller (0 for n <= 1)
BDS_HD float linspace01_s(int i, int n, float step) {
  if (n <= 1) return 0.f;
  return i < n / 2 ? step * (float)i : fmaf(-step, (float)(n - 1 - i), 1.0f);
}

// [0,1] coordinate -> clipped continuous grid index (reference op order kept)
BDS_HD float grid_coord(float c01, int size) {
  const float t = (c01 - 0.5f) * 2.f;
  float v = ((t + 1.f) / 2.f) * (float)(size - 1);
  v = v < 0.f ? 0.f : v;
  const float hi = (float)(size - 1);
  return v > hi ? hi : v;
}
// One axis of the slice as (first node, fraction towards the next one), with a coordinate ON the last node (i0 = g - 1, f = 0: the
// last pixel of a linspace) expressed from the cell before it (i0 = g - 2, f = 1): the same sample and the same scatter weights
// (slice_cell's clamped second node carries weight f = 0), but every coordinate now lies in one of the g - 1 cells -- what the
// cell-aligned tiles of the fused image transform need (csrc/mlp_head.hip).
BDS_HD void axis_cell(float c01, int g, int &i0, float &f) {
  const float v = grid_coord(c01, g);
  const float fl = floorf(v);
  i0 = (int)fl;
  f = v - fl;
  if (g > 1 && i0 == g - 1) { i0 = g - 2; f = 1.f; }
}

BDS_HD float guide_coord(float gray, int L, bool &interior) {
  const float z = gray * 2.f - 1.f;
  const float v = ((z + 1.f) / 2.f) * (float)(L - 1);
  const float hi = (float)(L - 1);
  interior = (v > 0.f) && (v < hi);  // grid_sample's border clip has zero gradient ON the boundary too
  return v < 0.f ? 0.f : (v > hi ? hi : v);
}

BDS_HD float rgb2gray(float r, float g, float b) { return r * kGrayR + g * kGrayG + b * kGrayB; }

struct Cell {
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  bool z_interior;
};
BDS_HD Cell slice_cell(float x01, float y01, float gray, int gx, int gy, int gl) {
  Cell c;
  const float ix = grid_coord(x01, gx), iy = grid_coord(y01, gy);
  const float iz = guide_coord(gray, gl, c.z_interior);
  const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
  c.fx = ix - x0; c.fy = iy - y0; c.fz = iz - z0;
  c.x0 = (int)x0; c.y0 = (int)y0; c.z0 = (int)z0;
  c.x1 = c.x0 + 1 < gx ? c.x0 + 1 : gx - 1;
  c.y1 = c.y0 + 1 < gy ? c.y0 + 1 : gy - 1;
  c.z1 = c.z0 + 1 < gl ? c.z0 + 1 : gl - 1;
  return c;
}

// trilinear sample of the 12 channels of grid [12, L, gy, gx]; optionally d(out)/d(iz)
BDS_HD void slice_sample(const float *grid, int gx, int gy, int gl, const Cell &c, float *out12, float *dz12) {
  const int plane = gy * gx, vol = gl * plane;
  const int o00 = c.y0 * gx + c.x0, o01 = c.y0 * gx + c.x1, o10 = c.y1 * gx + c.x0, o11 = c.y1 * gx + c.x1;
  const float w00 = (1.f - c.fy) * (1.f - c.fx), w01 = (1.f - c.fy) * c.fx, w10 = c.fy * (1.f - c.fx), w11 = c.fy * c.fx;
  for (int ch = 0; ch < 12; ch++) {
    const float *g0 = grid + ch * vol + c.z0 * plane;
    const float *g1 = grid + ch * vol + c.z1 * plane;
    const float a = g0[o00] * w00 + g0[o01] * w01 + g0[o10] * w10 + g0[o11] * w11;
    const float b = g1[o00] * w00 + g1[o01] * w01 + g1[o10] * w10 + g1[o11] * w11;
    out12[ch] = a * (1.f - c.fz) + b * c.fz;
    if (dz12) dz12[ch] = b - a;
  }
}

// the same for a run of `nch` <= 12 channels starting at `grid` (feature grids: any channel count, 12 at a time)
BDS_HD void slice_sample_n(const float *grid, int gx, int gy, int gl, const Cell &c, int nch, float *out, float *dz) {
  const int plane = gy * gx, vol = gl * plane;
  const int o00 = c.y0 * gx + c.x0, o01 = c.y0 * gx + c.x1, o10 = c.y1 * gx + c.x0, o11 = c.y1 * gx + c.x1;
  const float w00 = (1.f - c.fy) * (1.f - c.fx), w01 = (1.f - c.fy) * c.fx, w10 = c.fy * (1.f - c.fx), w11 = c.fy * c.fx;
  for (int ch = 0; ch < 12; ch++) {
    if (ch >= nch) { out[ch] = 0.f; if (dz) dz[ch] = 0.f; continue; }
    const float *g0 = grid + ch * vol + c.z0 * plane;
    const float *g1 = grid + ch * vol + c.z1 * plane;
    const float a = g0[o00] * w00 + g0[o01] * w01 + g0[o10] * w10 + g0[o11] * w11;
    const float b = g1[o00] * w00 + g1[o01] * w01 + g1[o10] * w10 + g1[o11] * w11;
    out[ch] = a * (1.f - c.fz) + b * c.fz;
    if (dz) dz[ch] = b - a;
  }
}

// p <- A[:, :3] p + A[:, 3]   (A row-major 3x4 in a12)
BDS_HD void apply_affine(const float *a12, float &r, float &g, float &b) {
  const float nr = a12[0] * r + a12[1] * g + a12[2] * b + a12[3];
  const float ng = a12[4] * r + a12[5] * g + a12[6] * b + a12[7];
  const float nb = a12[8] * r + a12[9] * g + a12[10] * b + a12[11];
  r = nr; g = ng; b = nb;
}

}  // namespace bds
