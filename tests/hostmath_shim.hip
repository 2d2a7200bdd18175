// TEST-ONLY host shim: runs the __host__ __device__ per-element formulas of the kernels on the CPU
// so that tests/ can compare them with oracle/ without a GPU.  Not part of libbds.so, never
// loaded by the product.
#include "../bilateral_driving_amd/csrc/gs_math.h"
#include "../bilateral_driving_amd/csrc/bilagrid_math.h"

using namespace bds;

extern "C" void hm_sh_bases(int n, int deg, const float *dirs, float *out16) {
  for (int i = 0; i < n; i++) {
    float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
    float inv = 1.f / sqrtf(x * x + y * y + z * z);
    float B[16] = {0};
    sh_bases(deg, x * inv, y * inv, z * inv, B);
    for (int k = 0; k < 16; k++) out16[i * 16 + k] = B[k];
  }
}

// v_dirs of  sum_k g[k] B_k(normalize(d))
extern "C" void hm_sh_bases_vjp(int n, int deg, const float *dirs, const float *g16, float *v_dirs) {
  for (int i = 0; i < n; i++) {
    float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
    float inv = 1.f / sqrtf(x * x + y * y + z * z);
    float ux = x * inv, uy = y * inv, uz = z * inv, ax, ay, az;
    sh_bases_vjp(deg, ux, uy, uz, g16 + i * 16, ax, ay, az);
    float dot = ax * ux + ay * uy + az * uz;
    v_dirs[i * 3] = (ax - dot * ux) * inv; v_dirs[i * 3 + 1] = (ay - dot * uy) * inv; v_dirs[i * 3 + 2] = (az - dot * uz) * inv;
  }
}

extern "C" void hm_project_fwd(int n, const float *means, const float *quats, const float *scales, const float *viewmat,
                               const float *K, int W, int H, float eps2d, float near_plane, float far_plane,
                               float radius_clip, int *radii, float *means2d, float *depths, float *conics, float *comps) {
  Camera cam = load_camera(viewmat, K);
  for (int i = 0; i < n; i++) {
    Proj p = project_one(means + i * 3, quats + i * 4, scales + i * 3, cam, W, H, eps2d, near_plane, far_plane, radius_clip);
    radii[i] = p.radius; means2d[i * 2] = p.mx; means2d[i * 2 + 1] = p.my; depths[i] = p.depth;
    conics[i * 3] = p.ca; conics[i * 3 + 1] = p.cb; conics[i * 3 + 2] = p.cc; comps[i] = p.comp;
  }
}

// the block bound of the one-view projection (bds_project_view_*_fwd_blocks): 1 = some centre of the box may come out visible
extern "C" int hm_box_may_be_visible(const float *lo, const float *hi, float smax, const float *viewmat, const float *K, int W, int H,
                                     float eps2d, float near_plane, float far_plane) {
  return box_may_be_visible(lo, hi, smax, load_camera(viewmat, K), W, H, eps2d, near_plane, far_plane) ? 1 : 0;
}

extern "C" void hm_project_bwd(int n, const float *means, const float *quats, const float *scales, const float *viewmat,
                               const float *K, int W, int H, float eps2d, const int *radii, const float *v_means2d,
                               const float *v_depths, const float *v_conics, float *v_means, float *v_quats,
                               float *v_scales, float *v_R /*9*/, float *v_t /*3*/) {
  Camera cam = load_camera(viewmat, K);
  for (int k = 0; k < 9; k++) v_R[k] = 0.f;
  for (int k = 0; k < 3; k++) v_t[k] = 0.f;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) { v_means[i * 3 + k] = 0.f; v_scales[i * 3 + k] = 0.f; }
    for (int k = 0; k < 4; k++) v_quats[i * 4 + k] = 0.f;
    if (radii[i] <= 0) continue;
    ProjGrad g;
    project_one_vjp(means + i * 3, quats + i * 4, scales + i * 3, cam, W, H, eps2d, v_means2d[i * 2], v_means2d[i * 2 + 1],
                    v_depths[i], v_conics[i * 3], v_conics[i * 3 + 1], v_conics[i * 3 + 2], g);
    for (int k = 0; k < 3; k++) { v_means[i * 3 + k] = g.v_mean[k]; v_scales[i * 3 + k] = g.v_scale[k]; v_t[k] += g.v_t[k]; }
    for (int k = 0; k < 4; k++) v_quats[i * 4 + k] = g.v_quat[k];
    for (int k = 0; k < 9; k++) v_R[k] += g.v_R[k];
  }
}

extern "C" void hm_tile_rect(int n, const float *means2d, const int *radii, int tile_size, int tw, int th, int *rect4) {
  for (int i = 0; i < n; i++)
    tile_rect(means2d[i * 2], means2d[i * 2 + 1], radii[i], tile_size, tw, th, rect4[i * 4], rect4[i * 4 + 1], rect4[i * 4 + 2],
              rect4[i * 4 + 3]);
}

extern "C" void hm_resample_taps(int out_size, int in_size, int *i0, int *i1, float *w1) {
  for (int d = 0; d < out_size; d++) {
    Tap t = resample_tap(d, out_size, in_size);
    i0[d] = t.i0; i1[d] = t.i1; w1[d] = t.w1;
  }
}

extern "C" void hm_linspace01(int n, float *out) {
  for (int i = 0; i < n; i++) out[i] = linspace01(i, n);
}

extern "C" void hm_slice(int P, const float *grid, int gx, int gy, int gl, const float *xy, const float *rgb, float *aff12,
                         float *dgray12) {
  for (int i = 0; i < P; i++) {
    Cell c = slice_cell(xy[i * 2], xy[i * 2 + 1], rgb2gray(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]), gx, gy, gl);
    float dz[12];
    slice_sample(grid, gx, gy, gl, c, aff12 + i * 12, dz);
    for (int k = 0; k < 12; k++) dgray12[i * 12 + k] = c.z_interior ? dz[k] * (float)(gl - 1) : 0.f;
  }
}

// culled tile set of one Gaussian as a tile_h x tile_w 0/1 mask (row-span method used by the kernels)
extern "C" int hm_culled_tiles(float mx, float my, int radius, float a, float b, float c, float opacity, int tile_size, int tw,
                               int th, unsigned char *mask) {
  for (int i = 0; i < tw * th; i++) mask[i] = 0;
  int x0, y0, x1, y1, n = 0;
  float q_max;
  if (!tile_rect_tight(mx, my, radius, a, b, c, opacity, tile_size, tw, th, x0, y0, x1, y1, q_max)) return 0;
  for (int ty = y0; ty < y1; ty++) {
    int lo, hi;
    row_tile_span(mx, my, a, b, c, q_max, ty, tile_size, x0, x1, lo, hi);
    for (int tx = lo; tx < hi; tx++) { mask[ty * tw + tx] = 1; n++; }
  }
  return n;
}

// axis_cell (the fused image transform's cell of a linspace coordinate) next to slice_cell's (x0, x1, fx) for every index of an axis
extern "C" void hm_axis_cells(int n, int g, int *i0, float *f, int *x0, int *x1, float *fx) {
  const float lin = n > 1 ? 1.0f / (float)(n - 1) : 0.f;
  for (int i = 0; i < n; i++) {
    const float c01 = linspace01_s(i, n, lin);
    axis_cell(c01, g, i0[i], f[i]);
    const Cell c = slice_cell(c01, 0.f, 0.5f, g, 1, 1);
    x0[i] = c.x0; x1[i] = c.x1; fx[i] = c.fx;
  }
}
