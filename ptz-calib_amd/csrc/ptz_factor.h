// ptz_factor.h -- per-observation PTZ reprojection residuals and closed-form Jacobians (FP64).
//
// Device math for the HIP kernels.  Functions are PTZ_HD so the same inlines can be instantiated by a
// host-only unit harness (tests/cpu_harness) to check the algebra without a GPU; the product path only
// ever runs them inside kernels.
//
// Reference behaviour followed (file:line relative to the reference tree):
//   F1 PTZRayFactor::operator()        src/core/ptzray_optimizer.cc:20-56
//   F2 PTZRayDistFactor::operator()    src/core/ptzray_optimizer.cc:65-129
//   F3 Reproj2d3dFactor::operator()    src/core/ptzray_optimizer.cc:268-326
//   F4 Factor2d2d::operator()          src/core/krt_optimizer.cc:22-43
//   F5 Factor2d2dDist::operator()      src/core/krt_optimizer.cc:80-132
//   F6 Camera::FromVector/cv::Rodrigues src/core/types.cc:59-73
// The reference differentiates these functors numerically (ceres::NumericDiffCostFunction, CENTRAL);
// the kernels use the exact derivatives of the same functions.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PTZ_HD __host__ __device__ __forceinline__
#else
#define PTZ_HD inline
#endif

namespace ptz {

// ---- camera block staged in LDS: everything an observation needs from its camera ----------------
// [0..8]   R (row-major)         rotation of the current rvec (cv::Rodrigues formula)
// [9] f  [10] cx [11] cy [12] fy (fy is only read by the 2D-3D factor)
// [13..17] k1 k2 k3 p1 p2
// ---- the first CANDBLK entries are all a residual-only evaluation needs ----
// [18..26] Jl (row-major)        left Jacobian of SO(3) at rvec: d(R X)/d r_k = Jl[:,k] x (R X)
// [27..32] Jacobi column scales of the free camera parameters (NC <= 6 of them)
constexpr int CAMBLK = 34;
constexpr int CAMBLK_DISP = 40;  // PTZRayDistDisp: nine scale slots and the displacement block CB_D
constexpr int CANDBLK = 18;
constexpr int CB_R = 0, CB_F = 9, CB_CX = 10, CB_CY = 11, CB_FY = 12, CB_K = 13, CB_JL = 18, CB_S = 27, CB_D = 36;  // CB_S: up to 9 Jacobi scales; CB_D: displacement block (3)

constexpr double kDblEps = 2.220446049250313e-16;

// 1 / d for the throughput kernels: hardware seed + two Newton steps (5 instructions; the IEEE division is 12).  Within an ulp
// or so of the quotient, not correctly rounded.  Where it is used, exactly:
//   * Jacobian entries and the factored forms of k_schur (products that are summed anyway): always;
//   * the RESIDUALS of the bundle-adjustment functors on the device: as a * rcp_nr(z) (PTZ_PDIV below) -- an ulp or so from the
//     reference's quotient; the parity tests hold costs to 1e-12 and parameters to 1e-6;
//   * the residuals of the single-view LM (krt_eval) inside its iterations: likewise; but the FINAL cost, which decides
//     KRTOptimizer::CheckResults' accept test (krt_optimizer.cc:504-533), is evaluated once more at the final point with IEEE divisions
//     (krt_eval<.., EXACT>): the gate sees the reference functor's own arithmetic (up to the order of the sum over the matches);
//   * the host harness (tests/cpu_harness) divides everywhere.
// A zero denominator gives NaN here where the division gives +-inf (or NaN for 0 / 0): non-finite either way, and the only thing
// the LM loops ask of such a cost is isfinite() -- a non-finite INITIAL cost is FAILURE, a non-finite candidate a rejected step, as
// in Ceres 1.14 (tests: test_krt_nonfinite_input_fails_like_the_oracle).
PTZ_HD double rcp_nr(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
#else
  return 1.0 / d;
#endif
}

// The quotient of the perspective division, a / z with iz = rcp_nr(z).  HOST build: the division, as the reference's functors
// have it (tests/cpu_harness holds these functions to the oracle bit for bit).  DEVICE, PTZ_PDIV: a * iz -- an IEEE division is 12
// instructions on this chip and the functors had five of them per observation, a third of the instruction stream of the
// issue-bound k_eval / k_lin_ray / k_lin_cam loops.
#if defined(__HIP_DEVICE_COMPILE__)
#define PTZ_PDIV(a, z, iz) ((a) * (iz))
#else
#define PTZ_PDIV(a, z, iz) ((a) / (z))
#endif

// cv::Rodrigues vector -> matrix (OpenCV 4.5.3 cvRodrigues2): theta < DBL_EPSILON -> I
PTZ_HD void rodrigues(const double r[3], double R[9])
{
  const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < kDblEps) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
  const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
  R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
  R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
  R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

// cv::Rodrigues matrix -> vector (OpenCV 4.5.3 cvRodrigues2, 3x3 branch) for an orthonormal R.
// (OpenCV first re-orthonormalises R with an SVD; products of rotation matrices are orthonormal to
// round-off, so that projection is the identity to ~1e-16 and is omitted.)
PTZ_HD void rodrigues_inv(const double R[9], double r[3])
{
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { rx = ry = rz = 0; }
    else {
      double t;
      t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0.);
      t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0.) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0.) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      theta /= sqrt(rx * rx + ry * ry + rz * rz);
      rx *= theta; ry *= theta; rz *= theta;
    }
  }
  else {
    const double vth = theta / (2 * s);
    rx *= vth; ry *= vth; rz *= vth;
  }
  r[0] = rx; r[1] = ry; r[2] = rz;
}

PTZ_HD void mat3_mul(const double* A, const double* B, double* C)
{
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Left Jacobian of SO(3): R(r + d) = exp([Jl d]_x) R(r) + O(d^2), so d(R X)/d r_k = Jl[:,k] x (R X).
// Jl = I + a [r]_x + b [r]_x^2, a = (1 - cos t)/t^2, b = (t - sin t)/t^3 (series below t = 1e-2).
PTZ_HD void so3_left_jacobian(const double r[3], double Jl[9])
{
  const double t2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double a, b;
  if (t2 < 1e-4) {
    a = 0.5 - t2 * (1.0 / 24.0) + t2 * t2 * (1.0 / 720.0);
    b = (1.0 / 6.0) - t2 * (1.0 / 120.0) + t2 * t2 * (1.0 / 5040.0);
  }
  else {
    const double t = sqrt(t2);
    const double sh = sin(0.5 * t);
    a = 2.0 * sh * sh / t2;
    b = (t - sin(t)) / (t2 * t);
  }
  const double x = r[0], y = r[1], z = r[2];
  // [r]_x^2 = r r^T - t2 I
  Jl[0] = 1.0 + b * (x * x - t2); Jl[1] = -a * z + b * x * y;     Jl[2] = a * y + b * x * z;
  Jl[3] = a * z + b * x * y;      Jl[4] = 1.0 + b * (y * y - t2); Jl[5] = -a * x + b * y * z;
  Jl[6] = -a * y + b * x * z;     Jl[7] = a * x + b * y * z;      Jl[8] = 1.0 + b * (z * z - t2);
}

// Brown distortion exactly as written in ptzray_optimizer.cc:111-120
PTZ_HD void brown(double x, double y, const double* k, double& xd, double& yd)
{
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r2 * r2 * r2, xy = x * y, x2 = x * x, y2 = y * y;
  const double radial = 1.0 + k[0] * r2 + k[1] * r4 + k[2] * r6;
  xd = x * radial + 2.0 * k[3] * xy + k[4] * (r2 + 2.0 * x2);
  yd = y * radial + 2.0 * k[4] * xy + k[3] * (r2 + 2.0 * y2);
}
// d(xd,yd)/d(x,y) (row-major B[4]) and d(xd,yd)/dk1
PTZ_HD void brown_jac(double x, double y, const double* k, double B[4], double dk1[2])
{
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double rad = 1.0 + k[0] * r2 + k[1] * r4 + k[2] * r6;
  const double drad = k[0] + 2.0 * k[1] * r2 + 3.0 * k[2] * r4;
  B[0] = rad + 2.0 * x * x * drad + 2.0 * k[3] * y + 6.0 * k[4] * x;
  B[1] = 2.0 * x * y * drad + 2.0 * k[3] * x + 2.0 * k[4] * y;
  B[2] = 2.0 * x * y * drad + 2.0 * k[4] * y + 2.0 * k[3] * x;
  B[3] = rad + 2.0 * y * y * drad + 2.0 * k[4] * x + 6.0 * k[3] * y;
  dk1[0] = x * r2;
  dk1[1] = y * r2;
}

// ---- global-BA 2D-2D factors ---------------------------------------------------------------------
// TYPE 0 = PTZRay: free camera parameters [f, r1, r2, r3]      (NC = 4)
// TYPE 1 = PTZRayDist: free [f, k1, r1, r2, r3]                 (NC = 5)
// The reference also leaves intr[1] ("fy") free, but neither functor reads it (param[1] = intr[0],
// ptzray_optimizer.cc:24-25, 69-70): its Jacobian column is identically zero, its LM step is exactly
// zero, so the column is not materialised.
// TYPE 2 = PTZRayFxfyDist (ptzray_optimizer.cc:136-191): free [fx, fy, k1, r1, r2, r3] (NC = 6); the ray IS normalised (:161),
// there is no behind-the-camera branch, fy is read (:167,185).
// TYPE 3 = PTZRayDistDisp (ptzray_optimizer.cc:202-259): as PTZRayFxfyDist's geometry with fy := fx, plus a displacement of the
// camera-frame point along z, delta = d0 + d1 fx + d2 fx^2, by ONE 3-parameter block shared by every residual (disp_param_,
// :655).  Free [f, k1, r1, r2, r3, d0, d1, d2] (NC = 8); the three displacement columns are per-camera copies of the one block
// (made one parameter by the group machinery of ptz_ba_kernels.h, as the oracle's cmap does).
template <int TYPE> struct BaDims {
  static constexpr int NC = (TYPE == 0) ? 4 : (TYPE == 1 ? 5 : (TYPE == 2 ? 6 : 8));
  static constexpr int ROT0 = TYPE == 3 ? 2 : NC - 3;   // first rotation column
  static constexpr int NI = NC - 3;                      // columns that are not rotations (the "intrinsic" components of a step)
};  // TYPE = factor (0 / 1 / 2 / 3)

// the point the functor feeds to the rotation: X / |X| for PTZRay / PTZRayFxfyDist, X itself for PTZRayDist; inv_n = 1 / |X| (or 1).
// A ray-centric kernel computes it once per ray instead of once per observation (same operations, same bits).
template <int TYPE>
PTZ_HD void ba_ray_point(const double X[3], double Xn[3], double& inv_n)
{
  if (TYPE != 1) {
    const double n = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
    inv_n = 1.0 / n;
    Xn[0] = X[0] / n; Xn[1] = X[1] / n; Xn[2] = X[2] / n;
  }
  else {
    inv_n = 1.0;
    Xn[0] = X[0]; Xn[1] = X[1]; Xn[2] = X[2];
  }
}

// Residual only.  cb = camera block (R at [CB_R], intrinsics); X = ray parameter (3).
template <int TYPE>
PTZ_HD void ba_residual_unit(const double* cb, const double Xn[3], float u, float v, double res[2])
{
  const double* R = cb + CB_R;
  const double f = cb[CB_F], cx = cb[CB_CX], cy = cb[CB_CY];
  const double Px = R[0] * Xn[0] + R[1] * Xn[1] + R[2] * Xn[2];
  const double Py = R[3] * Xn[0] + R[4] * Xn[1] + R[5] * Xn[2];
  const double Pz = R[6] * Xn[0] + R[7] * Xn[1] + R[8] * Xn[2];
  if (TYPE == 0) {
    // uv_predict = K R ray; uv_predict /= uv_predict(2)   (ptzray_optimizer.cc:49-50)
    const double iz0 = rcp_nr(Pz);
    (void)iz0;
    res[0] = (double)u - PTZ_PDIV(f * Px + cx * Pz, Pz, iz0);
    res[1] = (double)v - PTZ_PDIV(f * Py + cy * Pz, Pz, iz0);
  }
  else {
    if (TYPE == 1 && Pz < 0) { res[0] = 1000000.0; res[1] = 1000000.0; return; }  // :97-102
    const double Pzd = TYPE == 3 ? Pz + (cb[CB_D] + cb[CB_D + 1] * f + cb[CB_D + 2] * f * f) : Pz;  // :233-234
    const double izd = rcp_nr(Pzd);
    (void)izd;
    const double x = PTZ_PDIV(Px, Pzd, izd), y = PTZ_PDIV(Py, Pzd, izd);
    double xd, yd;
    brown(x, y, cb + CB_K, xd, yd);
    res[0] = (double)u - (f * xd + cx);
    res[1] = (double)v - ((TYPE == 2 ? cb[CB_FY] : f) * yd + cy);
  }
}

template <int TYPE>
PTZ_HD void ba_residual(const double* cb, const double X[3], float u, float v, double res[2])
{
  double Xn[3], inv_n;
  ba_ray_point<TYPE>(X, Xn, inv_n);
  ba_residual_unit<TYPE>(cb, Xn, u, v, res);
}

// Residual + Jacobians.  Jc[2][NC] wrt the free camera parameters, Jr[2][3] wrt the ray.  Unweighted,
// unscaled.
template <int TYPE>
PTZ_HD void ba_linearize(const double* cb, const double X[3], float u, float v, double res[2],
                         double Jc[2][BaDims<TYPE>::NC], double Jr[2][3])
{
  constexpr int NC = BaDims<TYPE>::NC, ROT0 = BaDims<TYPE>::ROT0;
  const double* R = cb + CB_R;
  const double* Jl = cb + CB_JL;
  const double f = cb[CB_F], cx = cb[CB_CX], cy = cb[CB_CY];
  const double fy = TYPE == 2 ? cb[CB_FY] : f;
  double Xn[3], inv_n = 1.0;
  if (TYPE != 1) {
    const double n = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
    inv_n = 1.0 / n;
    Xn[0] = X[0] / n; Xn[1] = X[1] / n; Xn[2] = X[2] / n;
  }
  else {
    Xn[0] = X[0]; Xn[1] = X[1]; Xn[2] = X[2];
  }
  const double Px = R[0] * Xn[0] + R[1] * Xn[1] + R[2] * Xn[2];
  const double Py = R[3] * Xn[0] + R[4] * Xn[1] + R[5] * Xn[2];
  const double Pz = R[6] * Xn[0] + R[7] * Xn[1] + R[8] * Xn[2];
  if (TYPE == 1 && Pz < 0) {
    res[0] = 1000000.0; res[1] = 1000000.0;
    for (int k = 0; k < NC; ++k) { Jc[0][k] = 0; Jc[1][k] = 0; }
    for (int k = 0; k < 3; ++k) { Jr[0][k] = 0; Jr[1][k] = 0; }
    return;
  }
  // PTZRayDistDisp: the camera-frame point moves along z by delta(f) before the projection (:233-236)
  const double delta = TYPE == 3 ? cb[CB_D] + cb[CB_D + 1] * f + cb[CB_D + 2] * f * f : 0.0;
  const double Pzd = TYPE == 3 ? Pz + delta : Pz;
  const double iz = rcp_nr(Pzd);
  const double x = PTZ_PDIV(Px, Pzd, iz), y = PTZ_PDIV(Py, Pzd, iz);  // (host build: the reference functor's divisions, PTZ_PDIV)
  double xd = x, yd = y;
  double B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (TYPE == 0) {
    res[0] = (double)u - PTZ_PDIV(f * Px + cx * Pz, Pz, iz);
    res[1] = (double)v - PTZ_PDIV(f * Py + cy * Pz, Pz, iz);
  }
  else {
    brown(x, y, cb + CB_K, xd, yd);
    brown_jac(x, y, cb + CB_K, B, dk1);
    res[0] = (double)u - (f * xd + cx);
    res[1] = (double)v - (fy * yd + cy);
  }
  // M = d pred / dP = diag(f, fy) * B * dpi,  dpi = [[iz, 0, -x iz], [0, iz, -y iz]]
  double M[2][3];
  M[0][0] = f * (B[0] * iz);  M[0][1] = f * (B[1] * iz);  M[0][2] = f * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  if (TYPE == 2) {
    Jc[0][0] = -xd; Jc[1][0] = 0;
    Jc[0][1] = 0;   Jc[1][1] = -yd;
    Jc[0][2] = -f * dk1[0];
    Jc[1][2] = -fy * dk1[1];
  }
  else {
    Jc[0][0] = -xd;
    Jc[1][0] = -yd;
    if (TYPE != 0) {
      Jc[0][1] = -f * dk1[0];
      Jc[1][1] = -f * dk1[1];
    }
    if (TYPE == 3) {  // P.z also moves with f; displacement block: dP.z / d(d0, d1, d2) = (1, f, f^2)
      const double ddf = cb[CB_D + 1] + 2.0 * cb[CB_D + 2] * f;
      Jc[0][0] -= M[0][2] * ddf;
      Jc[1][0] -= M[1][2] * ddf;
      const double pw[3] = {1.0, f, f * f};
#pragma unroll
      for (int k = 0; k < 3; ++k) { Jc[0][ROT0 + 3 + k] = -M[0][2] * pw[k]; Jc[1][ROT0 + 3 + k] = -M[1][2] * pw[k]; }
    }
  }
  // rotation: dP/dr_k = Jl[:,k] x P   (P = R X, before the displacement)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double ax = Jl[k], ay = Jl[3 + k], az = Jl[6 + k];
    const double dx = ay * Pz - az * Py, dy = az * Px - ax * Pz, dz = ax * Py - ay * Px;
    Jc[0][ROT0 + k] = -(M[0][0] * dx + M[0][1] * dy + M[0][2] * dz);
    Jc[1][ROT0 + k] = -(M[1][0] * dx + M[1][1] * dy + M[1][2] * dz);
  }
  // ray: dP/dX = R (I - Xn Xn^T)/|X| for PTZRay / PTZRayFxfyDist; since M P = 0 (the projection is scale invariant)
  // the projector term vanishes identically: d res/dX = -(M R)/|X|.  PTZRayDist: dP/dX = R.
  // PTZRayDistDisp: M (P + delta e_z) = 0, so M P = -delta M[:,2] and d res/dX = -(M R + delta M[:,2] Xn^T)/|X|.
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Jr[0][k] = -(M[0][0] * R[k] + M[0][1] * R[3 + k] + M[0][2] * R[6 + k] + (TYPE == 3 ? delta * M[0][2] * Xn[k] : 0.0)) * inv_n;
    Jr[1][k] = -(M[1][0] * R[k] + M[1][1] * R[3 + k] + M[1][2] * R[6 + k] + (TYPE == 3 ? delta * M[1][2] * Xn[k] : 0.0)) * inv_n;
  }
}

// The Jacobians of one observation in the factored form the Schur complement's off-diagonal blocks use (k_schur, phase 2):
// they depend on the camera and on the ray only, not on the pixel, so a camera-pair entry (a, b) -- two observations of ONE
// ray -- rebuilds b's from the ray it shares with a instead of gathering a stored product row.  With Xn / inv_n from
// ba_ray_point and P = R Xn:
//   MR[2][3]:  d res / d X = -MR * inv_n                        (MR = M R, plus the displacement term of PTZRayDistDisp)
//   G[2][NC]:  the camera columns of ba_linearize, except that the three rotation columns stay in the CAMERA FRAME,
//              G[r][ROT0 + m] = (M[r] x P)[m], so that Jc[r][ROT0 + k] = sum_m G[r][ROT0 + m] Jl[3 m + k]
//              (-M[r] . (Jl[:,k] x P) = Jl[:,k] . (M[r] x P)): the 3 x 3 product with Jl is the same for every observation of
//              the camera and is applied once per camera pair, after the sum over the pair's entries.
// Returns false where ba_linearize returns zero Jacobians (PTZRayDist behind the camera).  x = Px * iz here (ba_linearize
// divides, as the reference functor does for its residual): the blocks differ from it in the last bit, nothing more.
template <int TYPE>
PTZ_HD bool ba_pair_side(const double* R, double f, double fy, const double* kd, const double* dsp, const double Xn[3],
                         double MR[2][3], double G[2][BaDims<TYPE>::NC])
{
  constexpr int ROT0 = BaDims<TYPE>::ROT0;
  const double Px = R[0] * Xn[0] + R[1] * Xn[1] + R[2] * Xn[2];
  const double Py = R[3] * Xn[0] + R[4] * Xn[1] + R[5] * Xn[2];
  const double Pz = R[6] * Xn[0] + R[7] * Xn[1] + R[8] * Xn[2];
  if (TYPE == 1 && Pz < 0) return false;
  const double delta = TYPE == 3 ? dsp[0] + dsp[1] * f + dsp[2] * f * f : 0.0;
  const double iz = rcp_nr(TYPE == 3 ? Pz + delta : Pz);
  const double x = Px * iz, y = Py * iz;
  double M[2][3];
  if (TYPE == 0) {
    const double fiz = f * iz;
    M[0][0] = fiz; M[0][1] = 0.0; M[0][2] = -(fiz * x);
    M[1][0] = 0.0; M[1][1] = fiz; M[1][2] = -(fiz * y);
    G[0][0] = -x; G[1][0] = -y;
    // M[r] x P with the zeros of M written out
    G[0][ROT0] = -(M[0][2] * Py);       G[0][ROT0 + 1] = M[0][2] * Px - fiz * Pz;  G[0][ROT0 + 2] = fiz * Py;
    G[1][ROT0] = fiz * Pz - M[1][2] * Py;  G[1][ROT0 + 1] = M[1][2] * Px;          G[1][ROT0 + 2] = -(fiz * Px);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      MR[0][k] = fiz * R[k] + M[0][2] * R[6 + k];
      MR[1][k] = fiz * R[3 + k] + M[1][2] * R[6 + k];
    }
    return true;
  }
  double xd, yd, B[4], dk1[2];
  brown(x, y, kd, xd, yd);
  brown_jac(x, y, kd, B, dk1);
  M[0][0] = f * (B[0] * iz);  M[0][1] = f * (B[1] * iz);  M[0][2] = f * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  if (TYPE == 2) {
    G[0][0] = -xd; G[1][0] = 0;
    G[0][1] = 0;   G[1][1] = -yd;
    G[0][2] = -f * dk1[0];
    G[1][2] = -fy * dk1[1];
  }
  else {
    G[0][0] = -xd;
    G[1][0] = -yd;
    G[0][1] = -f * dk1[0];
    G[1][1] = -f * dk1[1];
    if (TYPE == 3) {
      const double ddf = dsp[1] + 2.0 * dsp[2] * f;
      G[0][0] -= M[0][2] * ddf;
      G[1][0] -= M[1][2] * ddf;
      const double pw[3] = {1.0, f, f * f};
#pragma unroll
      for (int k = 0; k < 3; ++k) { G[0][ROT0 + 3 + k] = -M[0][2] * pw[k]; G[1][ROT0 + 3 + k] = -M[1][2] * pw[k]; }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    G[r][ROT0] = M[r][1] * Pz - M[r][2] * Py;
    G[r][ROT0 + 1] = M[r][2] * Px - M[r][0] * Pz;
    G[r][ROT0 + 2] = M[r][0] * Py - M[r][1] * Px;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      MR[r][k] = M[r][0] * R[k] + M[r][1] * R[3 + k] + M[r][2] * R[6 + k] + (TYPE == 3 ? delta * M[r][2] * Xn[k] : 0.0);
  }
  return true;
}

// Residual, ray Jacobian and the camera-side DIRECTIONAL derivative p = Jc v of one observation, for a camera step v given as
// its intrinsic components sv[NC - 3] (in the column order of ba_linearize) and om = Jl v_rot: the three rotation columns
// -M (Jl[:,k] x P) collapse into -M (om x P), one cross product instead of three.  Same projection arithmetic as ba_linearize.
template <int TYPE>
PTZ_HD void ba_step_dir_unit(const double* cb, const double Xn[3], double inv_n, float u, float v, const double* sv, const double om[3],
                             double res[2], double p[2], double Jr[2][3])
{
  const double* R = cb + CB_R;
  const double f = cb[CB_F], cx = cb[CB_CX], cy = cb[CB_CY];
  const double fy = TYPE == 2 ? cb[CB_FY] : f;
  const double Px = R[0] * Xn[0] + R[1] * Xn[1] + R[2] * Xn[2];
  const double Py = R[3] * Xn[0] + R[4] * Xn[1] + R[5] * Xn[2];
  const double Pz = R[6] * Xn[0] + R[7] * Xn[1] + R[8] * Xn[2];
  if (TYPE == 1 && Pz < 0) {
    res[0] = 1000000.0; res[1] = 1000000.0;
    p[0] = 0; p[1] = 0;
    for (int k = 0; k < 3; ++k) { Jr[0][k] = 0; Jr[1][k] = 0; }
    return;
  }
  const double delta = TYPE == 3 ? cb[CB_D] + cb[CB_D + 1] * f + cb[CB_D + 2] * f * f : 0.0;
  const double Pzd = TYPE == 3 ? Pz + delta : Pz;
  const double iz = rcp_nr(Pzd);
  const double x = PTZ_PDIV(Px, Pzd, iz), y = PTZ_PDIV(Py, Pzd, iz);
  double xd = x, yd = y;
  double B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (TYPE == 0) {
    res[0] = (double)u - PTZ_PDIV(f * Px + cx * Pz, Pz, iz);
    res[1] = (double)v - PTZ_PDIV(f * Py + cy * Pz, Pz, iz);
  }
  else {
    brown(x, y, cb + CB_K, xd, yd);
    brown_jac(x, y, cb + CB_K, B, dk1);
    res[0] = (double)u - (f * xd + cx);
    res[1] = (double)v - (fy * yd + cy);
  }
  double M[2][3];
  M[0][0] = f * (B[0] * iz);  M[0][1] = f * (B[1] * iz);  M[0][2] = f * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  if (TYPE == 2) {
    p[0] = -xd * sv[0] - f * dk1[0] * sv[2];
    p[1] = -yd * sv[1] - fy * dk1[1] * sv[2];
  }
  else {
    p[0] = -xd * sv[0];
    p[1] = -yd * sv[0];
    if (TYPE != 0) { p[0] -= f * dk1[0] * sv[1]; p[1] -= f * dk1[1] * sv[1]; }
    if (TYPE == 3) {  // sv = [f, k1, d0, d1, d2]: the z-displacement moves with f and with the displacement block
      const double ddf = cb[CB_D + 1] + 2.0 * cb[CB_D + 2] * f;
      const double dz_step = ddf * sv[0] + sv[2] + f * sv[3] + f * f * sv[4];
      p[0] -= M[0][2] * dz_step;
      p[1] -= M[1][2] * dz_step;
    }
  }
  const double dx = om[1] * Pz - om[2] * Py, dy = om[2] * Px - om[0] * Pz, dz = om[0] * Py - om[1] * Px;
  p[0] -= M[0][0] * dx + M[0][1] * dy + M[0][2] * dz;
  p[1] -= M[1][0] * dx + M[1][1] * dy + M[1][2] * dz;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Jr[0][k] = -(M[0][0] * R[k] + M[0][1] * R[3 + k] + M[0][2] * R[6 + k] + (TYPE == 3 ? delta * M[0][2] * Xn[k] : 0.0)) * inv_n;
    Jr[1][k] = -(M[1][0] * R[k] + M[1][1] * R[3 + k] + M[1][2] * R[6 + k] + (TYPE == 3 ? delta * M[1][2] * Xn[k] : 0.0)) * inv_n;
  }
}

// Residual and ray Jacobian only (no camera side): what k_eval's second pass needs to leave the ray blocks of the CANDIDATE point
// behind -- the linearisation an accepted step would otherwise ask k_lin_ray for.  Same projection arithmetic as ba_step_dir_unit;
// reads the camera block's candidate prefix (rotation, intrinsics) only.
template <int TYPE>
PTZ_HD void ba_res_jr_unit(const double* cb, const double Xn[3], double inv_n, float u, float v, double res[2], double Jr[2][3])
{
  const double* R = cb + CB_R;
  const double f = cb[CB_F], cx = cb[CB_CX], cy = cb[CB_CY];
  const double fy = TYPE == 2 ? cb[CB_FY] : f;
  const double Px = R[0] * Xn[0] + R[1] * Xn[1] + R[2] * Xn[2];
  const double Py = R[3] * Xn[0] + R[4] * Xn[1] + R[5] * Xn[2];
  const double Pz = R[6] * Xn[0] + R[7] * Xn[1] + R[8] * Xn[2];
  if (TYPE == 1 && Pz < 0) {
    res[0] = 1000000.0; res[1] = 1000000.0;
    for (int k = 0; k < 3; ++k) { Jr[0][k] = 0; Jr[1][k] = 0; }
    return;
  }
  const double delta = TYPE == 3 ? cb[CB_D] + cb[CB_D + 1] * f + cb[CB_D + 2] * f * f : 0.0;
  const double Pzd = TYPE == 3 ? Pz + delta : Pz;
  const double iz = rcp_nr(Pzd);
  const double x = PTZ_PDIV(Px, Pzd, iz), y = PTZ_PDIV(Py, Pzd, iz);
  double xd = x, yd = y;
  double B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (TYPE == 0) {
    res[0] = (double)u - PTZ_PDIV(f * Px + cx * Pz, Pz, iz);
    res[1] = (double)v - PTZ_PDIV(f * Py + cy * Pz, Pz, iz);
  }
  else {
    brown(x, y, cb + CB_K, xd, yd);
    brown_jac(x, y, cb + CB_K, B, dk1);
    res[0] = (double)u - (f * xd + cx);
    res[1] = (double)v - (fy * yd + cy);
  }
  double M[2][3];
  M[0][0] = f * (B[0] * iz);  M[0][1] = f * (B[1] * iz);  M[0][2] = f * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Jr[0][k] = -(M[0][0] * R[k] + M[0][1] * R[3 + k] + M[0][2] * R[6 + k] + (TYPE == 3 ? delta * M[0][2] * Xn[k] : 0.0)) * inv_n;
    Jr[1][k] = -(M[1][0] * R[k] + M[1][1] * R[3 + k] + M[1][2] * R[6 + k] + (TYPE == 3 ? delta * M[1][2] * Xn[k] : 0.0)) * inv_n;
  }
}

template <int TYPE>
PTZ_HD void ba_step_dir(const double* cb, const double X[3], float u, float v, const double* sv, const double om[3], double res[2],
                        double p[2], double Jr[2][3])
{
  double Xn[3], inv_n;
  ba_ray_point<TYPE>(X, Xn, inv_n);
  ba_step_dir_unit<TYPE>(cb, Xn, inv_n, u, v, sv, om, res, p, Jr);
}

// ---- 2D-3D annotation factor (F3, Reproj2d3dFactor, ptzray_optimizer.cc:268-326) ---------------------------
// X_l = R(tlw[0:3]) X_w + tlw[3:6];  P = R(rvec) X_l (the extrinsic translation is NOT applied, :300);
// r = uv - (fx xd + cx, fy yd + cy) with Brown distortion; fy IS read here (:273).
// Free camera columns (NC3 = 5 + FACTOR): [fx, fy, (k1), r1, r2, r3]; tlw columns: [rho1..3, t1..3].
// tl = {R_lw (9), Jl_lw (9), t_lw (3)}.
constexpr int TLWBLK = 21;
// DISP: Reproj2d3dDispFactor (:335-396): the same with delta(fx) added to the camera-frame z; camera columns
// [fx, fy, k1, r1, r2, r3, d0, d1, d2].
template <int FACTOR, bool JAC, bool DISP = false>
PTZ_HD void reproj2d3d_eval(const double* cb, const double* tl, const double xyz[3], float u, float v, double res[2],
                            double Jc[2][5 + FACTOR + 3 * DISP], double Jt[2][6])
{
  constexpr int ROT0 = 2 + FACTOR;
  const double* R = cb + CB_R;
  const double* Rl = tl;
  const double fx = cb[CB_F], fy = cb[CB_FY], cx = cb[CB_CX], cy = cb[CB_CY];
  const double Yx = Rl[0] * xyz[0] + Rl[1] * xyz[1] + Rl[2] * xyz[2];
  const double Yy = Rl[3] * xyz[0] + Rl[4] * xyz[1] + Rl[5] * xyz[2];
  const double Yz = Rl[6] * xyz[0] + Rl[7] * xyz[1] + Rl[8] * xyz[2];
  const double Xx = Yx + tl[18], Xy = Yy + tl[19], Xz = Yz + tl[20];
  const double Px = R[0] * Xx + R[1] * Xy + R[2] * Xz;
  const double Py = R[3] * Xx + R[4] * Xy + R[5] * Xz;
  const double Pz = R[6] * Xx + R[7] * Xy + R[8] * Xz;
  const double Pzd = DISP ? Pz + (cb[CB_D] + cb[CB_D + 1] * fx + cb[CB_D + 2] * fx * fx) : Pz;
  const double iz = rcp_nr(Pzd), x = PTZ_PDIV(Px, Pzd, iz), y = PTZ_PDIV(Py, Pzd, iz);
  double xd, yd;
  brown(x, y, cb + CB_K, xd, yd);
  res[0] = (double)u - (fx * xd + cx);
  res[1] = (double)v - (fy * yd + cy);
  if (!JAC) return;
  double B[4], dk1[2];
  brown_jac(x, y, cb + CB_K, B, dk1);
  double M[2][3];
  M[0][0] = fx * (B[0] * iz);  M[0][1] = fx * (B[1] * iz);  M[0][2] = fx * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  Jc[0][0] = -xd; Jc[1][0] = 0;
  Jc[0][1] = 0;   Jc[1][1] = -yd;
  if (FACTOR) { Jc[0][2] = -fx * dk1[0]; Jc[1][2] = -fy * dk1[1]; }
  if (DISP) {
    const double ddf = cb[CB_D + 1] + 2.0 * cb[CB_D + 2] * fx;
    Jc[0][0] -= M[0][2] * ddf;
    Jc[1][0] -= M[1][2] * ddf;
    const double pw[3] = {1.0, fx, fx * fx};
    for (int k = 0; k < 3; ++k) { Jc[0][ROT0 + 3 + k] = -M[0][2] * pw[k]; Jc[1][ROT0 + 3 + k] = -M[1][2] * pw[k]; }
  }
  const double* Jl = cb + CB_JL;
  const double* Jw = tl + 9;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // camera rotation: dP = Jl[:,k] x P
    double ax = Jl[k], ay = Jl[3 + k], az = Jl[6 + k];
    double dx = ay * Pz - az * Py, dy = az * Px - ax * Pz, dz = ax * Py - ay * Px;
    Jc[0][ROT0 + k] = -(M[0][0] * dx + M[0][1] * dy + M[0][2] * dz);
    Jc[1][ROT0 + k] = -(M[1][0] * dx + M[1][1] * dy + M[1][2] * dz);
    // tlw rotation: dX_l = Jw[:,k] x (R_lw X_w), dP = R dX_l
    ax = Jw[k]; ay = Jw[3 + k]; az = Jw[6 + k];
    const double ex = ay * Yz - az * Yy, ey = az * Yx - ax * Yz, ez = ax * Yy - ay * Yx;
    dx = R[0] * ex + R[1] * ey + R[2] * ez; dy = R[3] * ex + R[4] * ey + R[5] * ez; dz = R[6] * ex + R[7] * ey + R[8] * ez;
    Jt[0][k] = -(M[0][0] * dx + M[0][1] * dy + M[0][2] * dz);
    Jt[1][k] = -(M[1][0] * dx + M[1][1] * dy + M[1][2] * dz);
    // tlw translation: dP = R e_k
    Jt[0][3 + k] = -(M[0][0] * R[k] + M[0][1] * R[3 + k] + M[0][2] * R[6 + k]);
    Jt[1][3 + k] = -(M[1][0] * R[k] + M[1][1] * R[3 + k] + M[1][2] * R[6 + k]);
  }
}

// ---- KRT single-view factors (F4 / F5) ------------------------------------------------------------
// KTYPE = KRTOptimizer::FACTOR_TYPE (krt_optimizer.h:110): 0 = F, 1 = FDist, 2 = Fxfy, 3 = FxfyDist; bit 0 = Brown
// distortion with k1 free, bit 1 = fy free.  Free parameters in ascending 15-vector index (krt_optimizer.cc:321-340):
// [fx, (fy), r1, r2, r3, (k1)].  ray1 = normalise(K1^-1 [u1, v1, 1]) is constant per match (R1 = I in the local frame,
// krt_optimizer.cc:275) and is precomputed once per match by the caller.
template <int KTYPE> struct KrtDims {
  static constexpr int DIST = KTYPE & 1, FXFY = (KTYPE >> 1) & 1;
  static constexpr int NF = 4 + DIST + FXFY;
  static constexpr int ROT0 = 1 + FXFY;  // first rotation column
};

template <int KTYPE, bool JAC, bool EXACT = false>
PTZ_HD void krt_eval(const double* R, const double* Jl, double fx, double fy, double cx, double cy, const double* kd,
                     const double ray1[3], bool skip, float u2, float v2, double res[2],
                     double J[2][KrtDims<KTYPE>::NF])
{
  constexpr int NF = KrtDims<KTYPE>::NF, DIST = KrtDims<KTYPE>::DIST, FXFY = KrtDims<KTYPE>::FXFY, ROT0 = KrtDims<KTYPE>::ROT0;
  if (skip) {  // undistorted reference pixel outside the frame: residual 0 (krt_optimizer.cc:97-101, 156-162)
    res[0] = 0; res[1] = 0;
    if (JAC) for (int k = 0; k < NF; ++k) { J[0][k] = 0; J[1][k] = 0; }
    return;
  }
  const double Px = R[0] * ray1[0] + R[1] * ray1[1] + R[2] * ray1[2];
  const double Py = R[3] * ray1[0] + R[4] * ray1[1] + R[5] * ray1[2];
  const double Pz = R[6] * ray1[0] + R[7] * ray1[1] + R[8] * ray1[2];
  // (EXACT: IEEE divisions, as the reference functor has them -- the final cost's evaluation, which decides the accept test)
  const double iz = rcp_nr(Pz), x = EXACT ? Px / Pz : PTZ_PDIV(Px, Pz, iz), y = EXACT ? Py / Pz : PTZ_PDIV(Py, Pz, iz);
  double xd = x, yd = y, B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (!DIST) {
    res[0] = (double)u2 - (EXACT ? (fx * Px + cx * Pz) / Pz : PTZ_PDIV(fx * Px + cx * Pz, Pz, iz));
    res[1] = (double)v2 - (EXACT ? (fy * Py + cy * Pz) / Pz : PTZ_PDIV(fy * Py + cy * Pz, Pz, iz));
  }
  else {
    brown(x, y, kd, xd, yd);
    if (JAC) brown_jac(x, y, kd, B, dk1);
    res[0] = (double)u2 - (fx * xd + cx);
    res[1] = (double)v2 - (fy * yd + cy);
  }
  if (!JAC) return;
  double M[2][3];
  M[0][0] = fx * (B[0] * iz);  M[0][1] = fx * (B[1] * iz);  M[0][2] = fx * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  if (FXFY) {
    J[0][0] = -xd; J[1][0] = 0;
    J[0][1] = 0;   J[1][1] = -yd;
  }
  else {
    J[0][0] = -xd;
    J[1][0] = -yd;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double ax = Jl[k], ay = Jl[3 + k], az = Jl[6 + k];
    const double dx = ay * Pz - az * Py, dy = az * Px - ax * Pz, dz = ax * Py - ay * Px;
    J[0][ROT0 + k] = -(M[0][0] * dx + M[0][1] * dy + M[0][2] * dz);
    J[1][ROT0 + k] = -(M[1][0] * dx + M[1][1] * dy + M[1][2] * dz);
  }
  if (DIST) {
    J[0][ROT0 + 3] = -fx * dk1[0];
    J[1][ROT0 + 3] = -fy * dk1[1];
  }
}

// Factor2d3dDist / Factor2d3dFxfyDist (krt_optimizer.cc:200-249): cv::projectPoints of a point given in the local frame,
// P = R X_l + t with the (constant) local translation, z = P.z ? 1/P.z : 1, Brown distortion with OpenCV's reading of the
// stored coefficients -- (k1,k2,p1,p2,k3) = kd[0,1,2,3,4] while the reference's own functors read (k1,k2,k3,p1,p2): the
// permutation is the reference's behaviour and is kept.  F and FDist both use the Dist functor with fy := fx.
// Same free columns as krt_eval.
template <int KTYPE, bool JAC>
PTZ_HD void krt_eval_2d3d(const double* R, const double* Jl, double fx, double fy, double cx, double cy, const double* kd,
                          const double t[3], const double Xl[3], float u, float v, double res[2],
                          double J[2][KrtDims<KTYPE>::NF])
{
  constexpr int DIST = KrtDims<KTYPE>::DIST, FXFY = KrtDims<KTYPE>::FXFY, ROT0 = KrtDims<KTYPE>::ROT0;
  const double Qx = R[0] * Xl[0] + R[1] * Xl[1] + R[2] * Xl[2];
  const double Qy = R[3] * Xl[0] + R[4] * Xl[1] + R[5] * Xl[2];
  const double Qz = R[6] * Xl[0] + R[7] * Xl[1] + R[8] * Xl[2];
  const double Px = Qx + t[0], Py = Qy + t[1], Pz = Qz + t[2];
  const double iz = Pz != 0.0 ? 1.0 / Pz : 1.0;
  const double x = Px * iz, y = Py * iz;
  const double kcv[5] = {kd[0], kd[1], kd[4], kd[2], kd[3]};  // (k1,k2,k3,p1,p2) as cv::projectPoints reads the storage
  double xd, yd;
  brown(x, y, kcv, xd, yd);
  res[0] = (double)u - (xd * fx + cx);
  res[1] = (double)v - (yd * fy + cy);
  if (!JAC) return;
  double B[4], dk1[2];
  brown_jac(x, y, kcv, B, dk1);
  double M[2][3];
  M[0][0] = fx * (B[0] * iz);  M[0][1] = fx * (B[1] * iz);  M[0][2] = fx * (-(B[0] * x + B[1] * y) * iz);
  M[1][0] = fy * (B[2] * iz);  M[1][1] = fy * (B[3] * iz);  M[1][2] = fy * (-(B[2] * x + B[3] * y) * iz);
  if (FXFY) {
    J[0][0] = -xd; J[1][0] = 0;
    J[0][1] = 0;   J[1][1] = -yd;
  }
  else {
    J[0][0] = -xd;
    J[1][0] = -yd;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // the rotation acts on R X_l only: dP = Jl[:,k] x (R X_l)
    const double ax = Jl[k], ay = Jl[3 + k], az = Jl[6 + k];
    const double dx = ay * Qz - az * Qy, dy = az * Qx - ax * Qz, dz = ax * Qy - ay * Qx;
    J[0][ROT0 + k] = -(M[0][0] * dx + M[0][1] * dy + M[0][2] * dz);
    J[1][ROT0 + k] = -(M[1][0] * dx + M[1][1] * dy + M[1][2] * dz);
  }
  if (DIST) {
    J[0][ROT0 + 3] = -fx * dk1[0];
    J[1][ROT0 + 3] = -fy * dk1[1];
  }
}

// cv::undistortPoints(src, dst, K, dist, noArray(), K) for one point, 5 fixed-point iterations
// (OpenCV 4.5.3 default criteria), result rounded to float32 (cv::Point2f, krt_optimizer.cc:89-92).
// OpenCV reads dist as (k1,k2,p1,p2,k3) while the reference stores (k1,k2,k3,p1,p2): the permutation
// is part of the reference's behaviour and is kept.
PTZ_HD void undistort_point(double fx, double fy, double cx, double cy, const double* d, float u, float v,
                            float& ou, float& ov)
{
  const double ifx = 1.0 / fx, ify = 1.0 / fy;  // cvUndistortPointsInternal multiplies by the reciprocals (a last-bit matter: the
  double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;  // result is rounded to float32 and compared with the frame border)
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2);
    if (icdist < 0) { x = ((double)u - cx) * ifx; y = ((double)v - cy) * ify; break; }
    const double dX = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x);
    const double dY = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y;
    x = (x0 - dX) * icdist;
    y = (y0 - dY) * icdist;
  }
  ou = (float)(x * fx + cx);
  ov = (float)(y * fy + cy);
}

// Symmetric 3x3 inverse through LL^T (Ceres InvertPSDMatrix<3>: llt().solve(I)).  A, Ainv: 6 unique
// entries [a00 a10 a11 a20 a21 a22].  Returns false if A is not positive definite.
PTZ_HD bool inv3_spd(const double A[6], double Ai[6])
{
  if (!(A[0] > 0)) return false;
  const double l00 = sqrt(A[0]);
  const double l10 = A[1] / l00, l20 = A[3] / l00;
  const double d1 = A[2] - l10 * l10;
  if (!(d1 > 0)) return false;
  const double l11 = sqrt(d1);
  const double l21 = (A[4] - l20 * l10) / l11;
  const double d2 = A[5] - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return false;
  const double l22 = sqrt(d2);
  // Linv (lower)
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  // A^-1 = Linv^T Linv
  Ai[0] = i00 * i00 + i10 * i10 + i20 * i20;
  Ai[1] = i10 * i11 + i20 * i21;
  Ai[2] = i11 * i11 + i21 * i21;
  Ai[3] = i20 * i22;
  Ai[4] = i21 * i22;
  Ai[5] = i22 * i22;
  return true;
}

}  // namespace ptz
