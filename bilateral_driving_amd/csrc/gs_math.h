// Per-element math of the rasterizer half (SH, projection, one pixel x one Gaussian blend).
// __host__ __device__ so that tests/ can exercise the exact device formulas on the host
// through a hipcc-built shim (tests/hostmath_shim.hip); the product only calls them from kernels.
//
// Algorithm: published 3DGS / gsplat v1.3.0; reference call sites:
// /root/reference/project/models/trainers/base.py:393-408, models/gaussians/vanilla.py:383-389.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define BDS_HD __host__ __device__ __forceinline__

namespace bds {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.999f;
constexpr float kTStop = 1e-4f;

// ------------------------------------------------------------------------------------------
// spherical harmonics
// ------------------------------------------------------------------------------------------
// Real SH bases up to degree 3 on the unit direction (x,y,z) (Sloan's recurrences).
BDS_HD void sh_bases(int deg, float x, float y, float z, float *B) {
  B[0] = 0.2820947917738781f;
  if (deg < 1) return;
  B[1] = -0.48860251190292f * y;
  B[2] = 0.48860251190292f * z;
  B[3] = -0.48860251190292f * x;
  if (deg < 2) return;
  float z2 = z * z;
  float fTmp0B = -1.092548430592079f * z;
  float fC1 = x * x - y * y;
  float fS1 = 2.f * x * y;
  B[4] = 0.5462742152960395f * fS1;
  B[5] = fTmp0B * y;
  B[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
  B[7] = fTmp0B * x;
  B[8] = 0.5462742152960395f * fC1;
  if (deg < 3) return;
  float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
  float fTmp1B = 1.445305721320277f * z;
  float fC2 = x * fC1 - y * fS1;
  float fS2 = x * fS1 + y * fC1;
  B[9] = -0.5900435899266435f * fS2;
  B[10] = fTmp1B * fS1;
  B[11] = fTmp0C * y;
  B[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
  B[13] = fTmp0C * x;
  B[14] = fTmp1B * fC1;
  B[15] = -0.5900435899266435f * fC2;
}

// d(sum_k g[k]*B[k]) / d(x,y,z) on the unit direction; g[k] = sum_c coeff[k][c]*v_out[c].
BDS_HD void sh_bases_vjp(int deg, float x, float y, float z, const float *g, float &vx, float &vy, float &vz) {
  vx = vy = vz = 0.f;
  if (deg < 1) return;
  vy += -0.48860251190292f * g[1];
  vz += 0.48860251190292f * g[2];
  vx += -0.48860251190292f * g[3];
  if (deg < 2) return;
  float z2 = z * z;
  float fTmp0B = -1.092548430592079f * z;
  float fC1 = x * x - y * y;
  float fS1 = 2.f * x * y;
  // B4 = c*fS1 ; B5 = fTmp0B*y ; B6 = a*z2 - b ; B7 = fTmp0B*x ; B8 = c*fC1
  float v_fS1 = 0.5462742152960395f * g[4];
  float v_fC1 = 0.5462742152960395f * g[8];
  float v_fTmp0B = y * g[5] + x * g[7];
  vy += fTmp0B * g[5];
  vx += fTmp0B * g[7];
  float v_z2 = 0.9461746957575601f * g[6];
  if (deg >= 3) {
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1;
    float fS2 = x * fS1 + y * fC1;
    (void)fC2; (void)fS2;
    float v_fS2 = -0.5900435899266435f * g[9];
    float v_fC2 = -0.5900435899266435f * g[15];
    float v_fTmp1B = fS1 * g[10] + fC1 * g[14];
    v_fS1 += fTmp1B * g[10];
    v_fC1 += fTmp1B * g[14];
    float v_fTmp0C = y * g[11] + x * g[13];
    vy += fTmp0C * g[11];
    vx += fTmp0C * g[13];
    // B12 = z*(a z2 - b)
    vz += (1.865881662950577f * z2 - 1.119528997770346f) * g[12];
    v_z2 += z * 1.865881662950577f * g[12];
    v_z2 += -2.285228997322329f * v_fTmp0C;
    vz += 1.445305721320277f * v_fTmp1B;
    // fC2 = x*fC1 - y*fS1 ; fS2 = x*fS1 + y*fC1
    vx += fC1 * v_fC2 + fS1 * v_fS2;
    vy += -fS1 * v_fC2 + fC1 * v_fS2;
    v_fC1 += x * v_fC2 + y * v_fS2;
    v_fS1 += -y * v_fC2 + x * v_fS2;
  }
  vz += -1.092548430592079f * v_fTmp0B;
  vz += 2.f * z * v_z2;
  vx += 2.f * x * v_fC1 + 2.f * y * v_fS1;
  vy += -2.f * y * v_fC1 + 2.f * x * v_fS1;
}

// ------------------------------------------------------------------------------------------
// tiny fixed-size linear algebra (row-major)
// ------------------------------------------------------------------------------------------
struct M3 { float m[9]; };

BDS_HD M3 mul33(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return r;
}
BDS_HD M3 mul33_nt(const M3 &a, const M3 &b) {  // a * b^T
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i * 3] * b.m[j * 3] + a.m[i * 3 + 1] * b.m[j * 3 + 1] + a.m[i * 3 + 2] * b.m[j * 3 + 2];
  return r;
}
BDS_HD M3 mul33_tn(const M3 &a, const M3 &b) {  // a^T * b
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i * 3 + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
  return r;
}

BDS_HD M3 quat_to_rotmat(float w, float x, float y, float z) {
  float inv = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
  w *= inv; x *= inv; y *= inv; z *= inv;
  M3 R;
  R.m[0] = 1.f - 2.f * (y * y + z * z); R.m[1] = 2.f * (x * y - w * z); R.m[2] = 2.f * (x * z + w * y);
  R.m[3] = 2.f * (x * y + w * z); R.m[4] = 1.f - 2.f * (x * x + z * z); R.m[5] = 2.f * (y * z - w * x);
  R.m[6] = 2.f * (x * z - w * y); R.m[7] = 2.f * (y * z + w * x); R.m[8] = 1.f - 2.f * (x * x + y * y);
  return R;
}

// gradient of the (normalising) quaternion -> rotation map
BDS_HD void quat_to_rotmat_vjp(float w0, float x0, float y0, float z0, const M3 &vR, float *vq) {
  float inv = 1.0f / sqrtf(w0 * w0 + x0 * x0 + y0 * y0 + z0 * z0);
  float w = w0 * inv, x = x0 * inv, y = y0 * inv, z = z0 * inv;
  const float *v = vR.m;
  float gw = 2.f * (x * (v[7] - v[5]) + y * (v[2] - v[6]) + z * (v[3] - v[1]));
  float gx = 2.f * (-2.f * x * (v[4] + v[8]) + y * (v[1] + v[3]) + z * (v[2] + v[6]) + w * (v[7] - v[5]));
  float gy = 2.f * (x * (v[1] + v[3]) - 2.f * y * (v[0] + v[8]) + z * (v[5] + v[7]) + w * (v[2] - v[6]));
  float gz = 2.f * (x * (v[2] + v[6]) + y * (v[5] + v[7]) - 2.f * z * (v[0] + v[4]) + w * (v[3] - v[1]));
  float dot = gw * w + gx * x + gy * y + gz * z;
  vq[0] = (gw - dot * w) * inv;
  vq[1] = (gx - dot * x) * inv;
  vq[2] = (gy - dot * y) * inv;
  vq[3] = (gz - dot * z) * inv;
}

// ------------------------------------------------------------------------------------------
// projection of one Gaussian into one camera
// ------------------------------------------------------------------------------------------
struct Camera {
  M3 R;          // world -> camera rotation
  float t[3];    // translation
  float fx, fy, cx, cy;
};

BDS_HD Camera load_camera(const float *viewmat /*4x4*/, const float *K /*3x3*/) {
  Camera c;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) c.R.m[i * 3 + j] = viewmat[i * 4 + j];
    c.t[i] = viewmat[i * 4 + 3];
  }
  c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
  return c;
}

struct Proj {
  int radius;        // 0 = culled
  float mx, my;      // means2d
  float depth;
  float ca, cb, cc;  // conic
  float comp;        // sqrt(det_orig / det_blur)
};

// 3D covariance in world space from quaternion (wxyz) + scale
BDS_HD M3 covar_world(const float *q, const float *s, M3 *Rq_out = nullptr) {
  M3 Rq = quat_to_rotmat(q[0], q[1], q[2], q[3]);
  M3 Ms;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ms.m[i * 3 + j] = Rq.m[i * 3 + j] * s[j];
  if (Rq_out) *Rq_out = Rq;
  return mul33_nt(Ms, Ms);
}

BDS_HD Proj project_one(const float *mean, const float *quat, const float *scale, const Camera &cam, int W, int H,
                        float eps2d, float near_plane, float far_plane, float radius_clip) {
  Proj o;
  o.radius = 0; o.mx = o.my = o.depth = o.ca = o.cb = o.cc = o.comp = 0.f;
  const float *R = cam.R.m;
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + cam.t[0];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + cam.t[1];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + cam.t[2];
  if (!(z > near_plane) || !(z < far_plane)) return o;
  M3 cov = covar_world(quat, scale);
  M3 covc = mul33_nt(mul33(cam.R, cov), cam.R);
  float lim_x = 1.3f * (0.5f * W / cam.fx), lim_y = 1.3f * (0.5f * H / cam.fy);
  float rz = 1.f / z, rz2 = rz * rz;
  float tx = z * fminf(lim_x, fmaxf(-lim_x, x * rz));
  float ty = z * fminf(lim_y, fmaxf(-lim_y, y * rz));
  // J = [[fx rz, 0, -fx tx rz2], [0, fy rz, -fy ty rz2]]
  float j00 = cam.fx * rz, j02 = -cam.fx * tx * rz2, j11 = cam.fy * rz, j12 = -cam.fy * ty * rz2;
  const float *c = covc.m;
  // T = J * covc  (2x3)
  float t00 = j00 * c[0] + j02 * c[6], t01 = j00 * c[1] + j02 * c[7], t02 = j00 * c[2] + j02 * c[8];
  float t10 = j11 * c[3] + j12 * c[6], t11 = j11 * c[4] + j12 * c[7], t12 = j11 * c[5] + j12 * c[8];
  float s00 = t00 * j00 + t02 * j02;
  float s01 = t01 * j11 + t02 * j12;
  float s10 = t10 * j00 + t12 * j02;
  float s11 = t11 * j11 + t12 * j12;
  float det_orig = s00 * s11 - s01 * s10;
  s00 += eps2d; s11 += eps2d;
  float det = s00 * s11 - s01 * s10;
  if (!(det > 0.f)) return o;
  float b = 0.5f * (s00 + s11);
  float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
  float radius = ceilf(3.f * sqrtf(v1));
  if (!(radius > radius_clip)) return o;
  float mx = cam.fx * x * rz + cam.cx, my = cam.fy * y * rz + cam.cy;
  if (mx + radius <= 0.f || mx - radius >= (float)W || my + radius <= 0.f || my - radius >= (float)H) return o;
  float idet = 1.f / det;
  o.radius = (int)radius;
  o.mx = mx; o.my = my; o.depth = z;
  o.ca = s11 * idet; o.cb = -0.5f * (s01 + s10) * idet; o.cc = s00 * idet;
  o.comp = sqrtf(fmaxf(0.f, det_orig * idet));
  return o;
}

// Can project_one return a visible Gaussian for ANY centre inside the box [lo, hi] when no Gaussian of the box has a scale above
// smax?  false = provably not (the caller may skip the whole block); true = maybe.  Conservative, from the tests of project_one:
//   near / far: z is affine in the centre, so its extremes over the box are at the 8 corners;
//   radius <= 3 sqrt(v1) + 1, v1 <= lambda_max(J covc J^T) + eps2d + 0.1 (the 0.01 floor under the root adds at most 0.1),
//     lambda_max <= ||J||_F^2 smax^2 <= F smax^2 / z^2 with F = fx^2 + fy^2 + (0.65 W)^2 + (0.65 H)^2 (|tx| <= 0.65 W z / fx by the
//     clamp) -- decreasing in z, so the value at the smallest depth any live centre of the box can have bounds them all;
//   image rectangle: mx + radius <= 0  <=  fx x + (cx + rho) z <= 0 for z > 0: linear in the camera-space centre, extremes at the
//     corners again (4 px of slack against the rounding of mx and of this test); likewise the other three sides.
BDS_HD bool box_may_be_visible(const float *lo, const float *hi, float smax, const Camera &cam, int W, int H, float eps2d,
                               float near_plane, float far_plane) {
  const float *R = cam.R.m;
  // an unbounded (or NaN) box -- a row of the block holds a non-finite centre: bds_gaussian_block_bounds opens the box for it -- would
  // run the tests below in Inf / NaN arithmetic and might cull the block's healthy rows: such a block is projected row by row
  for (int k = 0; k < 3; k++)
    if (!(fabsf(lo[k]) < 1.0e37f) || !(fabsf(hi[k]) < 1.0e37f)) return true;
  float zmin = 3.0e38f, zmax = -3.0e38f;
  float qx[8], qy[8], qz[8];
  for (int k = 0; k < 8; k++) {
    const float px = (k & 1) ? hi[0] : lo[0], py = (k & 2) ? hi[1] : lo[1], pz = (k & 4) ? hi[2] : lo[2];
    qx[k] = R[0] * px + R[1] * py + R[2] * pz + cam.t[0];
    qy[k] = R[3] * px + R[4] * py + R[5] * pz + cam.t[1];
    qz[k] = R[6] * px + R[7] * py + R[8] * pz + cam.t[2];
    zmin = fminf(zmin, qz[k]); zmax = fmaxf(zmax, qz[k]);
  }
  const float slack = 1e-4f * (fabsf(zmin) + fabsf(zmax)) + 1e-6f;     // rounding of the kernel's own z against this one's
  if (!(zmax + slack > near_plane) || !(zmin - slack < far_plane)) return false;
  const float zlo = fmaxf(zmin - slack, near_plane);
  if (!(zlo > 0.f) || !(smax < 3.0e38f) || !(smax == smax)) return true;
  const float F = cam.fx * cam.fx + cam.fy * cam.fy + 0.4225f * ((float)W * (float)W + (float)H * (float)H);
  const float rho = 3.f * sqrtf(F * smax * smax / (zlo * zlo) + eps2d + 0.1f) + 5.f;    // (+1 for the ceil, +4 px of slack)
  float left = -3.0e38f, right = 3.0e38f, top = -3.0e38f, bottom = 3.0e38f;
  for (int k = 0; k < 8; k++) {
    left = fmaxf(left, cam.fx * qx[k] + (cam.cx + rho) * qz[k]);
    right = fminf(right, cam.fx * qx[k] + (cam.cx - (float)W - rho) * qz[k]);
    top = fmaxf(top, cam.fy * qy[k] + (cam.cy + rho) * qz[k]);
    bottom = fminf(bottom, cam.fy * qy[k] + (cam.cy - (float)H - rho) * qz[k]);
  }
  if (left <= 0.f || right >= 0.f || top <= 0.f || bottom >= 0.f) return false;
  return true;
}

struct ProjGrad {
  float v_mean[3], v_quat[4], v_scale[3];
  float v_R[9], v_t[3];  // gradient w.r.t. the camera rotation / translation
};

// Backward of project_one for a Gaussian that was NOT culled.
BDS_HD void project_one_vjp(const float *mean, const float *quat, const float *scale, const Camera &cam, int W, int H,
                            float eps2d, float v_mx, float v_my, float v_depth, float v_ca, float v_cb, float v_cc,
                            ProjGrad &g) {
  const float *R = cam.R.m;
  float x = R[0] * mean[0] + R[1] * mean[1] + R[2] * mean[2] + cam.t[0];
  float y = R[3] * mean[0] + R[4] * mean[1] + R[5] * mean[2] + cam.t[1];
  float z = R[6] * mean[0] + R[7] * mean[1] + R[8] * mean[2] + cam.t[2];
  M3 Rq;
  M3 cov = covar_world(quat, scale, &Rq);
  M3 RC = mul33(cam.R, cov);
  M3 covc = mul33_nt(RC, cam.R);
  float fx = cam.fx, fy = cam.fy;
  float lim_x = 1.3f * (0.5f * W / fx), lim_y = 1.3f * (0.5f * H / fy);
  float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
  float tx = z * fminf(lim_x, fmaxf(-lim_x, x * rz));
  float ty = z * fminf(lim_y, fmaxf(-lim_y, y * rz));
  float j00 = fx * rz, j02 = -fx * tx * rz2, j11 = fy * rz, j12 = -fy * ty * rz2;
  const float *c = covc.m;
  float t00 = j00 * c[0] + j02 * c[6], t01 = j00 * c[1] + j02 * c[7], t02 = j00 * c[2] + j02 * c[8];
  float t10 = j11 * c[3] + j12 * c[6], t11 = j11 * c[4] + j12 * c[7], t12 = j11 * c[5] + j12 * c[8];
  float s00 = t00 * j00 + t02 * j02 + eps2d;
  float s01 = t01 * j11 + t02 * j12;
  float s10 = t10 * j00 + t12 * j02;
  float s11 = t11 * j11 + t12 * j12 + eps2d;
  float idet = 1.f / (s00 * s11 - s01 * s10);
  // conic (a, b, c) = (s11, -s01, s00) / det, det = s00 s11 - s01^2.  Through the ADJUGATE:
  //   v_S = adj(V) / det - tau conic,   tau = v_a a + v_b b + v_c c,   V = [[v_a, v_b / 2], [v_b / 2, v_c]]
  // and not as the closed form -conic V conic.  For a thin splat (conic eigenvalues 3 and 3e-4) seen along its length V is ~k u u^T
  // with u the long axis, so every entry of conic V conic is a difference of terms ~1e3 that leaves ~1: rounding of 1e-4 in ALL
  // directions of v_S, while the gradient of the quaternion rides on its mixed (long x short) component of ~1e-2 -- errors of 1-10 %
  // per such row (tests/test_gpu_25, seeds 16 / 20 / 35; gsplat 1.3.0's inverse_vjp is the closed form and shares them).  Here the
  // cancelling scalar tau multiplies the conic itself -- an error in it has no mixed component -- and adj(V) is exact.
  const float ca = s11 * idet, cb = -0.5f * (s01 + s10) * idet, cc = s00 * idet;
  const float tau = v_ca * ca + v_cb * cb + v_cc * cc;
  const float vs00 = v_cc * idet - tau * ca, vs11 = v_ca * idet - tau * cc;
  const float vs01 = -0.5f * v_cb * idet - tau * cb, vs10 = vs01;   // (each off-diagonal entry: half of d / d s01)
  // v_covc = J^T v_S J  (3x3)
  float J[6] = {j00, 0.f, j02, 0.f, j11, j12};
  float VS[4] = {vs00, vs01, vs10, vs11};
  M3 v_covc;
  float VJ[6];  // VS * J (2x3)
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 3; k++) VJ[i * 3 + k] = VS[i * 2] * J[k] + VS[i * 2 + 1] * J[3 + k];
  for (int a = 0; a < 3; a++)
    for (int k = 0; k < 3; k++) v_covc.m[a * 3 + k] = J[a] * VJ[k] + J[3 + a] * VJ[3 + k];
  // v_J = v_S J covc^T + v_S^T J covc   (2x3)
  float VJt[6];  // VS^T * J
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 3; k++) VJt[i * 3 + k] = VS[i] * J[k] + VS[2 + i] * J[3 + k];
  float vJ[6];
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 3; k++) {
      float acc = 0.f;
      for (int a = 0; a < 3; a++) acc += VJ[i * 3 + a] * c[k * 3 + a] + VJt[i * 3 + a] * c[a * 3 + k];
      vJ[i * 3 + k] = acc;
    }
  // mean in camera space
  float vmc[3];
  vmc[0] = fx * rz * v_mx;
  vmc[1] = fy * rz * v_my;
  vmc[2] = -(fx * x * v_mx + fy * y * v_my) * rz2 + v_depth;
  // J entries -> (x,y,z);  vJ[0]=dJ00, vJ[2]=dJ02, vJ[4]=dJ11, vJ[5]=dJ12
  if (x * rz <= lim_x && x * rz >= -lim_x) vmc[0] += -fx * rz2 * vJ[2];
  else vmc[2] += -fx * rz3 * vJ[2] * tx;
  if (y * rz <= lim_y && y * rz >= -lim_y) vmc[1] += -fy * rz2 * vJ[5];
  else vmc[2] += -fy * rz3 * vJ[5] * ty;
  vmc[2] += -fx * rz2 * vJ[0] - fy * rz2 * vJ[4] + 2.f * fx * tx * rz3 * vJ[2] + 2.f * fy * ty * rz3 * vJ[5];
  // world <- camera
  for (int i = 0; i < 3; i++) {
    g.v_mean[i] = R[i] * vmc[0] + R[3 + i] * vmc[1] + R[6 + i] * vmc[2];
    g.v_t[i] = vmc[i];
    for (int j = 0; j < 3; j++) g.v_R[i * 3 + j] = vmc[i] * mean[j];
  }
  // covc = R cov R^T :  v_R += v_covc R cov^T + v_covc^T R cov ; v_cov = R^T v_covc R
  {
    M3 A1 = mul33(v_covc, RC);                      // v_covc * (R cov)   [cov symmetric]
    M3 A2 = mul33_tn(v_covc, RC);                   // v_covc^T * (R cov)
    for (int i = 0; i < 9; i++) g.v_R[i] += A1.m[i] + A2.m[i];
  }
  M3 v_cov = mul33(mul33_tn(cam.R, v_covc), cam.R);
  // cov = M M^T, M = Rq diag(s):  v_M = (v_cov + v_cov^T) M
  M3 Ms;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ms.m[i * 3 + j] = Rq.m[i * 3 + j] * scale[j];
  M3 sym;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) sym.m[i * 3 + j] = v_cov.m[i * 3 + j] + v_cov.m[j * 3 + i];
  M3 vM = mul33(sym, Ms);
  M3 vRq;
  for (int j = 0; j < 3; j++) {
    float acc = 0.f;
    for (int i = 0; i < 3; i++) {
      vRq.m[i * 3 + j] = vM.m[i * 3 + j] * scale[j];
      acc += Rq.m[i * 3 + j] * vM.m[i * 3 + j];
    }
    g.v_scale[j] = acc;
  }
  quat_to_rotmat_vjp(quat[0], quat[1], quat[2], quat[3], vRq, g.v_quat);
}

// ------------------------------------------------------------------------------------------
// tile rectangle of a projected Gaussian (tile_min inclusive, tile_max exclusive)
// ------------------------------------------------------------------------------------------
BDS_HD void tile_rect(float mx, float my, int radius, int tile_size, int tile_w, int tile_h, int &x0, int &y0, int &x1,
                      int &y1) {
#pragma clang fp contract(off)  // the counting and the emitting kernel must take IDENTICAL integer decisions: no context-dependent fusing
  float ts = (float)tile_size;
  float tr = (float)radius / ts;
  float txf = mx / ts, tyf = my / ts;
  float fx0 = floorf(txf - tr), fy0 = floorf(tyf - tr), fx1 = ceilf(txf + tr), fy1 = ceilf(tyf + tr);
  x0 = (int)fminf(fmaxf(fx0, 0.f), (float)tile_w);
  y0 = (int)fminf(fmaxf(fy0, 0.f), (float)tile_h);
  x1 = (int)fminf(fmaxf(fx1, 0.f), (float)tile_w);
  y1 = (int)fminf(fmaxf(fy1, 0.f), (float)tile_h);
}


// ------------------------------------------------------------------------------------------
// exact (conservative) tile culling
// ------------------------------------------------------------------------------------------
// A pixel blends a Gaussian only if alpha = opacity * exp(-sigma) >= 1/255, i.e.
// sigma <= tau = ln(255 * opacity), sigma = 0.5*(a dx^2 + c dy^2) + b dx dy.  A (tile, Gaussian)
// pair none of whose 256 pixel centres can reach that is dropped from the intersection list: the
// rendered image and every gradient are unchanged (such a pair is skipped pixel by pixel anyway),
// only the lists the sort / compositing kernels walk get shorter.  kCullMargin keeps pairs whose best
// pixel is within rounding distance of the threshold.
constexpr float kCullMargin = 1e-3f;

BDS_HD float cull_tau(float opacity) { return logf(255.0f * opacity) + kCullMargin; }

// pixel-centre rectangle [x0,x1] x [y0,y1] (inclusive) vs the ellipse {q(d) <= q_max},
// q(d) = a dx^2 + 2 b dx dy + c dy^2 (a, c > 0), d = p - mean
BDS_HD bool rect_hits_ellipse(float mx, float my, float a, float b, float c, float q_max, float x0, float y0, float x1,
                              float y1) {
  const float dx0 = x0 - mx, dx1 = x1 - mx, dy0 = y0 - my, dy1 = y1 - my;
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;  // mean inside the rectangle
  float qmin = 3.4e38f;
  // vertical edges: dx fixed, dy in [dy0, dy1]; unconstrained minimiser dy* = -b dx / c
  for (int e = 0; e < 2; e++) {
    const float dx = e ? dx1 : dx0;
    float dy = -b * dx / c;
    dy = fminf(fmaxf(dy, dy0), dy1);
    qmin = fminf(qmin, a * dx * dx + 2.f * b * dx * dy + c * dy * dy);
  }
  for (int e = 0; e < 2; e++) {
    const float dy = e ? dy1 : dy0;
    float dx = -b * dy / a;
    dx = fminf(fmaxf(dx, dx0), dx1);
    qmin = fminf(qmin, a * dx * dx + 2.f * b * dx * dy + c * dy * dy);
  }
  return qmin <= q_max;
}

// Tile rectangle after culling: gsplat's bounding square (tile_rect) intersected with the axis-aligned
// bounding box of the tau-ellipse (pixel-centre convention).  Returns false if nothing is left.
BDS_HD bool tile_rect_tight(float mx, float my, int radius, float a, float b, float c, float opacity, int tile_size,
                            int tile_w, int tile_h, int &x0, int &y0, int &x1, int &y1, float &q_max) {
#pragma clang fp contract(off)  // the counting and the emitting kernel must take IDENTICAL integer decisions: no context-dependent fusing
  tile_rect(mx, my, radius, tile_size, tile_w, tile_h, x0, y0, x1, y1);
  const float tau = cull_tau(opacity);
  const float det = a * c - b * b;
  if (!(tau > 0.f) || !(det > 0.f) || !(a > 0.f) || !(c > 0.f)) { x1 = x0; y1 = y0; q_max = 0.f; return false; }
  q_max = 2.f * tau;
  const float hx = sqrtf(q_max * c / det) * 1.0001f + 0.01f, hy = sqrtf(q_max * a / det) * 1.0001f + 0.01f;
  const float ts = (float)tile_size;
  // tile t holds pixel centres t*ts + 0.5 ... t*ts + ts - 0.5
  const int tx0 = (int)ceilf((mx - hx - (ts - 0.5f)) / ts), tx1 = (int)floorf((mx + hx - 0.5f) / ts) + 1;
  const int ty0 = (int)ceilf((my - hy - (ts - 0.5f)) / ts), ty1 = (int)floorf((my + hy - 0.5f) / ts) + 1;
  x0 = x0 > tx0 ? x0 : tx0; x1 = x1 < tx1 ? x1 : tx1;
  y0 = y0 > ty0 ? y0 : ty0; y1 = y1 < ty1 ? y1 : ty1;
  if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; return false; }
  return true;
}

// Tiles of tile-row `ty` (inside [x0, x1)) that contain a pixel centre of the tau-ellipse: the ellipse
// cut by the row's band of pixel centres is convex, so the answer is ONE interval [tx_lo, tx_hi)
// obtained in O(1) from the x-extent of that cut (no per-tile test).  Conservative by kSpanSlack px.
constexpr float kSpanSlack = 0.02f;
BDS_HD void row_tile_span(float mx, float my, float a, float b, float c, float q_max, int ty, int tile_size, int x0,
                          int x1, int &tx_lo, int &tx_hi) {
#pragma clang fp contract(off)  // the counting and the emitting kernel must take IDENTICAL integer decisions: no context-dependent fusing
  tx_lo = tx_hi = x0;
  const float ts = (float)tile_size;
  const float det = a * c - b * b;
  const float hx = sqrtf(q_max * c / det), hy = sqrtf(q_max * a / det);
  // band of pixel-centre rows of this tile row, relative to the mean, clipped to the ellipse's y-extent
  float e0 = (ty * ts + 0.5f) - my, e1 = (ty * ts + ts - 0.5f) - my;
  if (e0 > hy + kSpanSlack || e1 < -hy - kSpanSlack) return;
  e0 = fminf(fmaxf(e0, -hy), hy);
  e1 = fminf(fmaxf(e1, -hy), hy);
  // x-extent of the ellipse at height dy: (-b dy -+ sqrt(a q - det dy^2)) / a
  const float r0 = sqrtf(fmaxf(a * q_max - det * e0 * e0, 0.f)), r1 = sqrtf(fmaxf(a * q_max - det * e1 * e1, 0.f));
  float xr = fmaxf((-b * e0 + r0) / a, (-b * e1 + r1) / a);
  float xl = fminf((-b * e0 - r0) / a, (-b * e1 - r1) / a);
  const float dyR = -b * hx / c;  // height of the right-most / left-most (-dyR) point of the ellipse
  if (dyR >= e0 && dyR <= e1) xr = hx;
  if (-dyR >= e0 && -dyR <= e1) xl = -hx;
  xl += mx - kSpanSlack;
  xr += mx + kSpanSlack;
  // tile t holds pixel centres t*ts + 0.5 ... t*ts + ts - 0.5
  int lo = (int)ceilf((xl - (ts - 0.5f)) / ts), hi = (int)floorf((xr - 0.5f) / ts) + 1;
  lo = lo > x0 ? lo : x0;
  hi = hi < x1 ? hi : x1;
  if (hi > lo) { tx_lo = lo; tx_hi = hi; }
}

}  // namespace bds
