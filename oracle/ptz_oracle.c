/*
 * ptz_oracle.c -- CPU restatement of the PTZ-Calib hot path.  TEST INFRASTRUCTURE ONLY
 * (see ptz_oracle.h for the rules and the parity status: **parity unpinned** for the
 * floating-point path; the union-find / track ids are pinned against oracle/_ref).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * [Ceres-1.14] / [OpenCV-4.5.3] mark restatements of the un-vendored dependencies
 * (install_deps.sh:44-72 OpenCV 4.5.3, :119-128 Ceres 1.14.0).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  No fast-math:
 * the restatement keeps IEEE-754 double semantics throughout.
 */
#include "ptz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static void mat3_mul_vec(const double* M, const double* v, double* out)
{
  out[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  out[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  out[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}

static void mat3_mul(const double* A, const double* B, double* C)
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static void mat3_transpose(const double* A, double* T)
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}

/* closed-form 3x3 inverse (what cv::Mat::inv() does for 3x3, DECOMP_LU special case) */
static int mat3_inv(const double* S, double* D)
{
  double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (det == 0.0) return 0;
  double d = 1.0 / det;
  D[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  D[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  D[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  D[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  D[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  D[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  D[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  D[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  D[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return 1;
}

static double vec3_norm(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

void orc_free(void* p) { free(p); }

void orc_lm_options_default(orc_lm_options* o)
{
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 200; /* run_ptz_ba.cc:52 */
  o->jacobian_mode = ORC_JAC_NUMERIC;
  o->num_threads = 1;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* F6  Rodrigues  (types.cc:41,68 -> cv::Rodrigues, [OpenCV-4.5.3] calib3d cvRodrigues2)       */
/* ------------------------------------------------------------------------------------------ */

void orc_rodrigues(const double r[3], double R[9])
{
  double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    return;
  }
  double c = cos(theta), s = sin(theta), c1 = 1.0 - c, itheta = 1.0 / theta;
  double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
  /* R = cos(theta) I + (1 - cos(theta)) r r^T + sin(theta) [r]_x */
  R[0] = c + c1 * x * x;
  R[1] = c1 * x * y - s * z;
  R[2] = c1 * x * z + s * y;
  R[3] = c1 * x * y + s * z;
  R[4] = c + c1 * y * y;
  R[5] = c1 * y * z - s * x;
  R[6] = c1 * x * z - s * y;
  R[7] = c1 * y * z + s * x;
  R[8] = c + c1 * z * z;
}

/* dR[9*k + e] = d R[e] / d r_k.  Closed form of [OpenCV-4.5.3] cvRodrigues2's jacobian:
 * dR/dr_i = a0 I + a1 rr^T + a2 (e_i r^T + r e_i^T) + a3 [r]_x + a4 [e_i]_x, r normalised;
 * theta -> 0 limit: dR/dr_i = [e_i]_x. */
void orc_rodrigues_jac(const double r[3], double R[9], double dR[27])
{
  orc_rodrigues(r, R);
  double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  static const double EX[3][9] = {{0, 0, 0, 0, 0, -1, 0, 1, 0}, {0, 0, 1, 0, 0, 0, -1, 0, 0}, {0, -1, 0, 1, 0, 0, 0, 0, 0}};
  if (theta < DBL_EPSILON) {
    for (int k = 0; k < 3; ++k)
      for (int e = 0; e < 9; ++e) dR[9 * k + e] = EX[k][e];
    return;
  }
  double c = cos(theta), s = sin(theta), itheta = 1.0 / theta;
  double sh = sin(0.5 * theta);
  double c1 = 2.0 * sh * sh; /* 1 - cos(theta) without cancellation */
  double rn[3] = {r[0] * itheta, r[1] * itheta, r[2] * itheta};
  double rrt[9], rx[9] = {0, -rn[2], rn[1], rn[2], 0, -rn[0], -rn[1], rn[0], 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) rrt[3 * i + j] = rn[i] * rn[j];
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 3; ++k) {
    double ri = rn[k];
    double a0 = -s * ri, a1 = (s - 2.0 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri, a4 = s * itheta;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        int e = 3 * i + j;
        double drrt = (i == k ? rn[j] : 0.0) + (j == k ? rn[i] : 0.0);
        dR[9 * k + e] = a0 * I3[e] + a1 * rrt[e] + a2 * drrt + a3 * rx[e] + a4 * EX[k][e];
      }
  }
}

/* matrix -> vector, [OpenCV-4.5.3] cvRodrigues2 (3x3 branch).  OpenCV first projects R onto SO(3)
 * with an SVD (R = U V^T); here the same projection is reached with Newton polar iterations. */
void orc_rodrigues_inv(const double Rin[9], double rv[3])
{
  double R[9];
  memcpy(R, Rin, sizeof(R));
  for (int it = 0; it < 8; ++it) {
    double Ri[9], Rit[9];
    if (!mat3_inv(R, Ri)) break;
    mat3_transpose(Ri, Rit);
    double diff = 0;
    for (int e = 0; e < 9; ++e) {
      double n = 0.5 * (R[e] + Rit[e]);
      diff += fabs(n - R[e]);
      R[e] = n;
    }
    if (diff < 1e-17) break;
  }
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) {
      rx = ry = rz = 0;
    }
    else {
      double t;
      t = (R[0] + 1) * 0.5;
      rx = sqrt(t > 0 ? t : 0.);
      t = (R[4] + 1) * 0.5;
      ry = sqrt(t > 0 ? t : 0.) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = sqrt(t > 0 ? t : 0.) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      double n = sqrt(rx * rx + ry * ry + rz * rz);
      theta /= n;
      rx *= theta; ry *= theta; rz *= theta;
    }
  }
  else {
    double vth = 1 / (2 * s);
    vth *= theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  rv[0] = rx; rv[1] = ry; rv[2] = rz;
}

/* ------------------------------------------------------------------------------------------ */
/* Brown distortion as written in ptzray_optimizer.cc:111-123 / krt_optimizer.cc:111-126      */
/* ------------------------------------------------------------------------------------------ */
static void brown(double x, double y, double k1, double k2, double k3, double p1, double p2, double* xd, double* yd)
{
  double r2 = x * x + y * y;
  double r4 = r2 * r2;
  double r6 = r2 * r2 * r2;
  double xy = x * y;
  double x2 = x * x;
  double y2 = y * y;
  double radial_dist = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  *xd = x * radial_dist + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2);
  *yd = y * radial_dist + 2.0 * p2 * xy + p1 * (r2 + 2.0 * y2);
}

/* d(xd,yd)/d(x,y) -> J[4] row-major, and d(xd,yd)/dk1 */
static void brown_jac(double x, double y, double k1, double k2, double k3, double p1, double p2, double* J, double* dk1)
{
  double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  double rad = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  double drad = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4; /* d rad / d r2 */
  J[0] = rad + 2.0 * x * x * drad + 2.0 * p1 * y + 6.0 * p2 * x;
  J[1] = 2.0 * x * y * drad + 2.0 * p1 * x + 2.0 * p2 * y;
  J[2] = 2.0 * x * y * drad + 2.0 * p2 * y + 2.0 * p1 * x;
  J[3] = rad + 2.0 * y * y * drad + 2.0 * p2 * x + 6.0 * p1 * y;
  dk1[0] = x * r2;
  dk1[1] = y * r2;
}

/* ------------------------------------------------------------------------------------------ */
/* F1  PTZRayFactor::operator()  ptzray_optimizer.cc:20-56                                     */
/* ------------------------------------------------------------------------------------------ */
void orc_res_ptzray(const double* intr, const double* extr, const double* ray, const float* uv, double* res)
{
  /* param[0] = param[1] = intrinsics[0]; cx, cy = intrinsics[2], [3]   (:24-27) */
  double f = intr[0], cx = intr[2], cy = intr[3];
  double R[9];
  orc_rodrigues(extr, R); /* Camera::FromVector, types.cc:67-68 */
  double n = vec3_norm(ray); /* cv_ray /= cv::norm(cv_ray)  (:45-46) */
  double X[3] = {ray[0] / n, ray[1] / n, ray[2] / n};
  double P[3];
  mat3_mul_vec(R, X, P);
  /* uv_predict = K * R * ray; uv_predict /= uv_predict(2)   (:49-50), K = [[f,0,cx],[0,f,cy],[0,0,1]] */
  double px = f * P[0] + cx * P[2], py = f * P[1] + cy * P[2], pz = P[2];
  res[0] = (double)uv[0] - px / pz;
  res[1] = (double)uv[1] - py / pz;
}

/* F2  PTZRayDistFactor::operator()  ptzray_optimizer.cc:65-129 */
void orc_res_ptzray_dist(const double* intr, const double* extr, const double* ray, const float* uv, double* res)
{
  double fx = intr[0], fy = intr[0], cx = intr[2], cy = intr[3]; /* param[1] = intrinsics[0]  (:70) */
  double R[9], P[3];
  orc_rodrigues(extr, R);
  mat3_mul_vec(R, ray, P); /* no normalisation (:91) */
  const double kPenalty = 1000000.0;
  if (P[2] < 0) { /* :98-102 */
    res[0] = kPenalty;
    res[1] = kPenalty;
    return;
  }
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, intr[4], intr[5], intr[6], intr[7], intr[8], &xd, &yd);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
}

/* PTZRayFxfyDistFactor::operator()  ptzray_optimizer.cc:138-193 (dead from the CLI) */
void orc_res_ptzray_fxfy_dist(const double* intr, const double* extr, const double* ray, const float* uv, double* res)
{
  double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  double R[9], P[3];
  orc_rodrigues(extr, R);
  double n = vec3_norm(ray);
  double X[3] = {ray[0] / n, ray[1] / n, ray[2] / n};
  mat3_mul_vec(R, X, P);
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, intr[4], intr[5], intr[6], intr[7], intr[8], &xd, &yd);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
}

/* PTZRayDistDispFactor::operator()  ptzray_optimizer.cc:195-259: unit ray, P = R X, P.z += d0 + d1 f + d2 f^2 (f = intr[0]),
 * perspective divide, Brown distortion, fy := fx.  No behind-the-camera branch. */
void orc_res_ptzray_dist_disp(const double* intr, const double* disp, const double* extr, const double* ray, const float* uv,
                              double* res)
{
  double fx = intr[0], fy = intr[0], cx = intr[2], cy = intr[3];
  double R[9], P[3];
  orc_rodrigues(extr, R);
  double n = vec3_norm(ray);
  double X[3] = {ray[0] / n, ray[1] / n, ray[2] / n};
  mat3_mul_vec(R, X, P);
  double displacement = disp[0] + disp[1] * fx + disp[2] * fx * fx;
  P[2] += displacement;
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, intr[4], intr[5], intr[6], intr[7], intr[8], &xd, &yd);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
}

/* Reproj2d3dDispFactor::operator()  ptzray_optimizer.cc:334-396: as Reproj2d3dFactor with the displacement added to the
 * camera-frame z; fy = intr[1] IS read (:340). */
void orc_res_reproj2d3d_disp(const double* intr, const double* disp, const double* extr, const double* tlw, const float* uv,
                             const double* xyz, double* res)
{
  double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  double R[9], Rlw[9], Xl[3], P[3];
  orc_rodrigues(extr, R);
  orc_rodrigues(tlw, Rlw);
  mat3_mul_vec(Rlw, xyz, Xl);
  Xl[0] += tlw[3]; Xl[1] += tlw[4]; Xl[2] += tlw[5];
  mat3_mul_vec(R, Xl, P);
  double displacement = disp[0] + disp[1] * fx + disp[2] * fx * fx;
  P[2] += displacement;
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, intr[4], intr[5], intr[6], intr[7], intr[8], &xd, &yd);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
}

/* F3  Reproj2d3dFactor::operator()  ptzray_optimizer.cc:268-326;  T_l_w :507-513 */
void orc_res_reproj2d3d(const double* intr, const double* extr, const double* tlw, const float* uv, const double* xyz,
                        double* res)
{
  double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3]; /* fy IS read here (:273) */
  double R[9], Rlw[9], Xl[3], P[3];
  orc_rodrigues(extr, R);
  orc_rodrigues(tlw, Rlw);
  mat3_mul_vec(Rlw, xyz, Xl);
  Xl[0] += tlw[3]; Xl[1] += tlw[4]; Xl[2] += tlw[5];
  mat3_mul_vec(R, Xl, P); /* extrinsic t is NOT applied (:300) */
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, intr[4], intr[5], intr[6], intr[7], intr[8], &xd, &yd);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
}

/* ray1 = normalise(R1^-1 K1^-1 [u,v,1]) with R1 = I  (krt_optimizer.cc:31-33, 275) */
static void krt_ray1(const double* k1, double u, double v, double* ray1)
{
  /* K1^-1 of [[fx,0,cx],[0,fy,cy],[0,0,1]] applied to (u,v,1) */
  double X[3] = {(u - k1[2]) / k1[0], (v - k1[3]) / k1[1], 1.0};
  double n = vec3_norm(X);
  ray1[0] = X[0] / n; ray1[1] = X[1] / n; ray1[2] = X[2] / n;
}

/* F4  Factor2d2d::operator()  krt_optimizer.cc:22-43 */
void orc_res_2d2d(const double* cam, const double* k1, const float* uv1, const float* uv2, double* res)
{
  double f = cam[0], cx = cam[2], cy = cam[3]; /* param[1] = param[0]  (:26) */
  double R[9], ray1[3], P[3];
  orc_rodrigues(cam + 4, R);
  krt_ray1(k1, (double)uv1[0], (double)uv1[1], ray1);
  mat3_mul_vec(R, ray1, P);
  double px = f * P[0] + cx * P[2], py = f * P[1] + cy * P[2], pz = P[2];
  res[0] = (double)uv2[0] - px / pz;
  res[1] = (double)uv2[1] - py / pz;
}

/* cv::undistortPoints(src, dst, K, dist, noArray(), K)  [OpenCV-4.5.3 cvUndistortPointsInternal,
 * default criteria = 5 fixed-point iterations].  OpenCV reads the 5 coefficients as
 * (k1,k2,p1,p2,k3); the reference hands it its own (k1,k2,k3,p1,p2) vector unchanged
 * (krt_optimizer.cc:91, types.cc:72), so k3/p1/p2 are permuted -- reproduced here on purpose. */
void orc_undistort_point(const double* k1, const double* dist1, const float* uv, float* out)
{
  double fx = k1[0], fy = k1[1], cx = k1[2], cy = k1[3];
  double k[5] = {dist1[0], dist1[1], dist1[2], dist1[3], dist1[4]}; /* OpenCV meaning: k1,k2,p1,p2,k3 */
  double ifx = 1.0 / fx, ify = 1.0 / fy; /* OpenCV: x = (x - cx)*ifx */
  double x = ((double)uv[0] - cx) * ifx, y = ((double)uv[1] - cy) * ify;
  double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    double r2 = x * x + y * y;
    double icdist = 1.0 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    if (icdist < 0) {
      x = ((double)uv[0] - cx) * ifx;
      y = ((double)uv[1] - cy) * ify;
      break;
    }
    double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
    double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  /* P = K: back to pixels, stored as cv::Point2f */
  out[0] = (float)(x * fx + cx);
  out[1] = (float)(y * fy + cy);
}

/* F5  Factor2d2dDist::operator()  krt_optimizer.cc:80-132 */
void orc_res_2d2d_dist(const double* cam, const double* k1, const double* dist1, const float* uv1, const float* uv2,
                       double* res)
{
  double fx = cam[0], fy = cam[0], cx = cam[2], cy = cam[3];
  float und[2];
  orc_undistort_point(k1, dist1, uv1, und);
  double width1 = k1[2] * 2, height1 = k1[3] * 2;
  if (und[0] < 0 || und[0] >= width1 || und[1] < 0 || und[1] >= height1) { /* :97-101 */
    res[0] = 0;
    res[1] = 0;
    return;
  }
  double R[9], ray1[3], P[3];
  orc_rodrigues(cam + 4, R);
  krt_ray1(k1, (double)und[0], (double)und[1], ray1);
  mat3_mul_vec(R, ray1, P);
  double x = P[0] / P[2], y = P[1] / P[2];
  double xd, yd;
  brown(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], &xd, &yd);
  res[0] = (double)uv2[0] - (fx * xd + cx);
  res[1] = (double)uv2[1] - (fy * yd + cy);
}

/* ------------------------------------------------------------------------------------------ */
/* P3  TracksBuilder  tracks.cc:19-113, union_find.h:33-106, flat_pair_map.h:22-52             */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t img, feat; } node_t;

static int node_cmp(const void* a, const void* b)
{
  const node_t* x = (const node_t*)a;
  const node_t* y = (const node_t*)b;
  if (x->img != y->img) return x->img < y->img ? -1 : 1;
  if (x->feat != y->feat) return x->feat < y->feat ? -1 : 1;
  return 0;
}

static int32_t node_find(const node_t* nodes, int32_t n, node_t key)
{ /* std::lower_bound on the sorted flat map (flat_pair_map.h:25-27) */
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    int32_t mid = lo + (hi - lo) / 2;
    if (node_cmp(&nodes[mid], &key) < 0) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

static int32_t uf_find(int32_t* parent, int32_t i)
{ /* recursive path compression, union_find.h:58-68 (iterative two-pass form, same result) */
  int32_t root = i;
  while (parent[root] != root) root = parent[root];
  while (parent[i] != root) {
    int32_t nx = parent[i];
    parent[i] = root;
    i = nx;
  }
  return root;
}

static int i32_cmp(const void* a, const void* b)
{
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : x > y;
}

int32_t orc_tracks_build(int32_t n_pairs, const int64_t* src, const int64_t* dst, const int64_t* match_ptr,
                         const int32_t* query_idx, const int32_t* train_idx, int32_t min_track_length,
                         int32_t** track_id_out, int64_t** track_ptr_out, int32_t** entry_image_out,
                         int32_t** entry_feature_out)
{
  int64_t n_match = match_ptr[n_pairs];
  /* 1.-2. all (image, feature) nodes, sorted unique -> flat index (tracks.cc:21-41) */
  node_t* nodes = (node_t*)malloc(sizeof(node_t) * (size_t)(2 * n_match + 1));
  int64_t m = 0;
  for (int32_t p = 0; p < n_pairs; ++p)
    for (int64_t k = match_ptr[p]; k < match_ptr[p + 1]; ++k) {
      nodes[m].img = (int32_t)src[p]; nodes[m].feat = query_idx[k]; ++m;
      nodes[m].img = (int32_t)dst[p]; nodes[m].feat = train_idx[k]; ++m;
    }
  qsort(nodes, (size_t)m, sizeof(node_t), node_cmp);
  int32_t n = 0;
  for (int64_t i = 0; i < m; ++i)
    if (n == 0 || node_cmp(&nodes[n - 1], &nodes[i]) != 0) nodes[n++] = nodes[i];

  /* 3. InitSets (union_find.h:43-52) */
  int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  int32_t* rank = (int32_t*)calloc((size_t)(n + 1), sizeof(int32_t));
  int32_t* size = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int32_t i = 0; i < n; ++i) { parent[i] = i; size[i] = 1; }

  /* 4. Union in file order (tracks.cc:47-60, union_find.h:71-95) */
  for (int32_t p = 0; p < n_pairs; ++p)
    for (int64_t k = match_ptr[p]; k < match_ptr[p + 1]; ++k) {
      node_t a = {(int32_t)src[p], query_idx[k]}, b = {(int32_t)dst[p], train_idx[k]};
      int32_t ri = uf_find(parent, node_find(nodes, n, a));
      int32_t rj = uf_find(parent, node_find(nodes, n, b));
      if (ri == rj) continue;
      if (rank[ri] < rank[rj]) {
        parent[ri] = rj;
        size[rj] += size[ri];
      }
      else {
        parent[rj] = ri;
        size[ri] += size[rj];
        if (rank[ri] == rank[rj]) ++rank[ri];
      }
    }

  /* Filter (tracks.cc:63-96): Find(k) for every node (compresses every path), collect distinct
   * images per root, mark roots with a repeated image or < min_track_length images. */
  int32_t* root_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int32_t k = 0; k < n; ++k) root_of[k] = uf_find(parent, k);
  /* sort node ids by (root, image) to count distinct images and detect duplicates */
  int64_t* key = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
  int32_t* n_img = (int32_t*)calloc((size_t)(n + 1), sizeof(int32_t));   /* distinct images per root */
  uint8_t* bad = (uint8_t*)calloc((size_t)(n + 1), 1);
  uint8_t* is_root_seen = (uint8_t*)calloc((size_t)(n + 1), 1);
  /* nodes are sorted by (img, feat); within one root, equal images are adjacent in node order only
   * per image, so walk nodes grouped by root using a per-root "last image" table. */
  int32_t* last_img = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int32_t i = 0; i < n; ++i) last_img[i] = -1;
  for (int32_t k = 0; k < n; ++k) {
    int32_t r = root_of[k];
    is_root_seen[r] = 1;
    /* node order is ascending image id, so a repeated image within a root shows up as
     * last_img[r] == img (std::set insert failing, tracks.cc:77) */
    if (last_img[r] == nodes[k].img) bad[r] = 1;
    else { last_img[r] = nodes[k].img; ++n_img[r]; }
  }
  for (int32_t r = 0; r < n; ++r)
    if (is_root_seen[r] && n_img[r] < min_track_length) bad[r] = 1; /* tracks.cc:83-87 */
  /* reset marked roots in the parent array (tracks.cc:90-96): after the Find sweep every
   * parent[k] is a root id, so every member of a bad track gets INT_MAX. */
  for (int32_t k = 0; k < n; ++k) {
    int32_t r = parent[k];
    if (bad[r]) { size[r] = 1; root_of[k] = INT32_MAX; }
    else root_of[k] = r;
  }

  /* ExportToSTL (tracks.cc:99-113): tracks[parent[k]].insert(node) if not rejected and size > 1 */
  int32_t n_tracks = 0;
  int64_t n_entries = 0;
  for (int32_t k = 0; k < n; ++k)
    if (root_of[k] != INT32_MAX && size[root_of[k]] > 1) ++n_entries;
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int32_t r = 0; r < n; ++r) is_root_seen[r] = 0;
  for (int32_t k = 0; k < n; ++k) {
    int32_t r = root_of[k];
    if (r != INT32_MAX && size[r] > 1 && !is_root_seen[r]) { is_root_seen[r] = 1; ids[n_tracks++] = r; }
  }
  qsort(ids, (size_t)n_tracks, sizeof(int32_t), i32_cmp); /* std::map<int, Track> iterates ascending */
  int32_t* slot = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int32_t t = 0; t < n_tracks; ++t) slot[ids[t]] = t;
  int64_t* tptr = (int64_t*)calloc((size_t)(n_tracks + 1), sizeof(int64_t));
  for (int32_t k = 0; k < n; ++k) {
    int32_t r = root_of[k];
    if (r != INT32_MAX && size[r] > 1) ++tptr[slot[r] + 1];
  }
  for (int32_t t = 0; t < n_tracks; ++t) tptr[t + 1] += tptr[t];
  int32_t* eimg = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_entries + 1));
  int32_t* efeat = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_entries + 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_tracks + 1));
  memcpy(fill, tptr, sizeof(int64_t) * (size_t)(n_tracks + 1));
  for (int32_t k = 0; k < n; ++k) { /* node order = ascending (image, feature) = std::map<int,int> order */
    int32_t r = root_of[k];
    if (r != INT32_MAX && size[r] > 1) {
      int64_t pos = fill[slot[r]]++;
      eimg[pos] = nodes[k].img;
      efeat[pos] = nodes[k].feat;
    }
  }
  *track_id_out = ids;
  *track_ptr_out = tptr;
  *entry_image_out = eimg;
  *entry_feature_out = efeat;
  free(nodes); free(parent); free(rank); free(size); free(root_of); free(key); free(n_img); free(bad);
  free(is_root_seen); free(last_img); free(slot); free(fill);
  return n_tracks;
}

/* ------------------------------------------------------------------------------------------ */
/* [Ceres-1.14] generic trust-region Levenberg-Marquardt loop                                  */
/* trust_region_minimizer.cc (Minimize / IterationZero / ComputeTrustRegionStep / ...),        */
/* levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc (monotonic steps)           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int n_free; /* tangent dimension */
  int n_amb;  /* ambient dimension of x */
  void* ctx;
  double (*cost)(void* ctx, const double* x);
  /* residuals + Jacobian at x stored in ctx (weighted, unscaled); g = J^T r (tangent, unscaled) */
  double (*linearize)(void* ctx, const double* x, double* g);
  void (*col_sqnorm)(void* ctx, double* out); /* of the stored (possibly scaled) Jacobian */
  void (*scale_cols)(void* ctx, const double* s);
  int (*solve)(void* ctx, const double* D, double* y); /* argmin |J y - r|^2 + |D y|^2; 0 = ok */
  double (*model_cost_change)(void* ctx, const double* step); /* -(J step)^T (r + J step / 2) */
  void (*plus)(void* ctx, const double* x, const double* delta, double* x_out);
  double (*x_norm)(void* ctx, const double* x);               /* ambient norm over problem blocks */
  double (*diff_norm)(void* ctx, const double* a, const double* b);
  double (*grad_max)(void* ctx, const double* g);
} lm_problem;

static void trace_push(orc_lm_trace* t, double cost, double cost_change, double radius, double rho, int accepted)
{
  if (!t || t->count >= t->capacity) return;
  int i = t->count++;
  if (t->cost) t->cost[i] = cost;
  if (t->cost_change) t->cost_change[i] = cost_change;
  if (t->radius) t->radius[i] = radius;
  if (t->rho) t->rho[i] = rho;
  if (t->accepted) t->accepted[i] = accepted;
}

static int all_finite(const double* v, int n)
{
  for (int i = 0; i < n; ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

static void lm_minimize(lm_problem* P, double* x_inout, const orc_lm_options* o, orc_lm_summary* S, orc_lm_trace* T)
{
  const int nf = P->n_free, na = P->n_amb;
  double* x = (double*)malloc(sizeof(double) * (size_t)na);
  double* xc = (double*)malloc(sizeof(double) * (size_t)na);
  double* best = (double*)malloc(sizeof(double) * (size_t)na);
  double* g = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  double* scale = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  double* diag = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  double* lmD = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  double* step = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  double* delta = (double*)malloc(sizeof(double) * (size_t)(nf + 1));
  memcpy(x, x_inout, sizeof(double) * (size_t)na);
  memcpy(best, x, sizeof(double) * (size_t)na);
  if (T) T->count = 0;

  /* LevenbergMarquardtStrategy state */
  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0;

  memset(S, 0, sizeof(*S));
  S->termination_type = ORC_NO_CONVERGENCE;

  /* --- IterationZero + EvaluateGradientAndJacobian --- */
  double x_cost = P->linearize(P->ctx, x, g);
  S->num_jacobian_evals = 1;
  if (!isfinite(x_cost)) {
    S->termination_type = ORC_FAILURE;
    goto done;
  }
  for (int i = 0; i < nf; ++i) scale[i] = 1.0;
  if (o->jacobi_scaling) {
    P->col_sqnorm(P->ctx, scale);
    for (int i = 0; i < nf; ++i) scale[i] = 1.0 / (1.0 + sqrt(scale[i]));
    P->scale_cols(P->ctx, scale);
  }
  double x_norm = P->x_norm(P->ctx, x);
  double grad_max = P->grad_max(P->ctx, g);
  double minimum_cost = x_cost;
  S->initial_cost = x_cost;
  S->final_cost = x_cost;

  int iteration = 0;               /* index of the iteration summary being finalised */
  int step_is_successful = 1;      /* iteration 0 */
  int num_consecutive_invalid = 0;
  int n_summaries = 0;
  double it_cost = x_cost, it_cost_change = 0.0, it_rho = 0.0;
  int it_flag = 1;

  for (;;) {
    /* --- FinalizeIterationAndCheckIfMinimizerCanContinue --- */
    if (step_is_successful) {
      ++S->num_successful_steps;
      if (x_cost < minimum_cost || n_summaries == 0) {
        minimum_cost = x_cost;
        memcpy(best, x, sizeof(double) * (size_t)na);
      }
    }
    else {
      ++S->num_unsuccessful_steps;
    }
    trace_push(T, it_cost, it_cost_change, radius, it_rho, it_flag);
    ++n_summaries;
    if (it_cost < S->final_cost) S->final_cost = it_cost; /* SetSummaryFinalCost: min over iterations */
    S->final_gradient_max_norm = grad_max;
    if (iteration >= o->max_num_iterations) { /* MaxSolverIterationsReached */
      S->termination_type = ORC_NO_CONVERGENCE;
      break;
    }
    if (step_is_successful && grad_max <= o->gradient_tolerance) { /* GradientToleranceReached */
      S->termination_type = ORC_CONVERGENCE;
      break;
    }
    if (radius <= o->min_trust_region_radius) { /* MinTrustRegionRadiusReached */
      S->termination_type = ORC_CONVERGENCE;
      break;
    }

    ++iteration;
    ++S->num_lm_steps;
    step_is_successful = 0;
    it_cost = x_cost; it_cost_change = 0.0; it_rho = 0.0; it_flag = -1;

    /* --- ComputeTrustRegionStep: LevenbergMarquardtStrategy::ComputeStep --- */
    if (!reuse_diagonal) {
      P->col_sqnorm(P->ctx, diag);
      for (int i = 0; i < nf; ++i) diag[i] = fmin(fmax(diag[i], o->min_lm_diagonal), o->max_lm_diagonal);
    }
    for (int i = 0; i < nf; ++i) lmD[i] = sqrt(diag[i] / radius);
    int solve_fail = P->solve(P->ctx, lmD, step);
    ++S->num_linear_solves;
    reuse_diagonal = 1;
    int step_is_valid = 0;
    double model_cost_change = 0.0;
    if (!solve_fail && all_finite(step, nf)) {
      for (int i = 0; i < nf; ++i) step[i] = -step[i];
      model_cost_change = P->model_cost_change(P->ctx, step);
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      /* --- HandleInvalidStep --- */
      ++num_consecutive_invalid;
      if (num_consecutive_invalid >= o->max_num_consecutive_invalid_steps) {
        S->termination_type = ORC_FAILURE;
        break;
      }
      radius *= 0.5; /* StepIsInvalid */
      reuse_diagonal = 0;
      continue;
    }
    num_consecutive_invalid = 0;
    for (int i = 0; i < nf; ++i) delta[i] = step[i] * scale[i]; /* undo Jacobi scaling */

    /* --- ComputeCandidatePointAndEvaluateCost --- */
    P->plus(P->ctx, x, delta, xc);
    double candidate_cost = P->cost(P->ctx, xc);
    if (!isfinite(candidate_cost)) candidate_cost = DBL_MAX;

    /* --- ParameterToleranceReached --- */
    double step_norm = P->diff_norm(P->ctx, x, xc);
    if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
      S->termination_type = ORC_CONVERGENCE;
      break;
    }
    /* --- FunctionToleranceReached --- */
    double cost_change = x_cost - candidate_cost;
    if (fabs(cost_change) <= o->function_tolerance * x_cost) {
      S->termination_type = ORC_CONVERGENCE;
      break;
    }
    /* --- IsStepSuccessful: TrustRegionStepEvaluator::StepQuality (monotonic) --- */
    double rho = cost_change / model_cost_change;
    it_cost_change = cost_change;
    it_rho = rho;
    if (rho > o->min_relative_decrease) {
      /* --- HandleSuccessfulStep --- */
      memcpy(x, xc, sizeof(double) * (size_t)na);
      x_norm = P->x_norm(P->ctx, x);
      x_cost = P->linearize(P->ctx, x, g);
      ++S->num_jacobian_evals;
      if (!isfinite(x_cost)) {
        S->termination_type = ORC_FAILURE;
        break;
      }
      if (o->jacobi_scaling) P->scale_cols(P->ctx, scale);
      grad_max = P->grad_max(P->ctx, g);
      step_is_successful = 1;
      it_cost = x_cost;
      it_flag = 1;
      /* StepAccepted */
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3));
      radius = fmin(o->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
    }
    else {
      /* --- HandleUnsuccessfulStep / StepRejected --- */
      it_cost = candidate_cost;
      it_flag = 0;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
    }
  }
  S->num_iterations = n_summaries - 1;

done:
  S->final_radius = radius;
  memcpy(x_inout, best, sizeof(double) * (size_t)na);
  free(x); free(xc); free(best); free(g); free(scale); free(diag); free(lmD); free(step); free(delta);
}

/* ------------------------------------------------------------------------------------------ */
/* dense Cholesky (lower, in place), used for the reduced camera system                        */
/* ------------------------------------------------------------------------------------------ */
static int chol_lower(double* A, int n)
{ /* right-looking, row-major; returns 0 ok */
  const int NB = 48;
  for (int k0 = 0; k0 < n; k0 += NB) {
    int kb = (k0 + NB < n) ? NB : n - k0;
    /* factor diagonal block + panel below (unblocked columns inside the panel) */
    for (int k = k0; k < k0 + kb; ++k) {
      double d = A[(size_t)k * n + k];
      for (int j = k0; j < k; ++j) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
      if (!(d > 0.0) || !isfinite(d)) return 1;
      d = sqrt(d);
      A[(size_t)k * n + k] = d;
      double inv = 1.0 / d;
#pragma omp parallel for schedule(static) if (n - k > 256)
      for (int i = k + 1; i < n; ++i) {
        double v = A[(size_t)i * n + k];
        for (int j = k0; j < k; ++j) v -= A[(size_t)i * n + j] * A[(size_t)k * n + j];
        A[(size_t)i * n + k] = v * inv;
      }
    }
    /* trailing update: A[i][j] -= sum_k L[i][k] L[j][k], i >= j >= k0 + kb */
    int t0 = k0 + kb;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = t0; i < n; ++i) {
      const double* Li = A + (size_t)i * n + k0;
      for (int j = t0; j <= i; ++j) {
        const double* Lj = A + (size_t)j * n + k0;
        double acc = 0;
        for (int k = 0; k < kb; ++k) acc += Li[k] * Lj[k];
        A[(size_t)i * n + j] -= acc;
      }
    }
  }
  return 0;
}

static void chol_solve(const double* L, int n, double* b)
{
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int j = 0; j < i; ++j) v -= L[(size_t)i * n + j] * b[j];
    b[i] = v / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int j = i + 1; j < n; ++j) v -= L[(size_t)j * n + i] * b[j];
    b[i] = v / L[(size_t)i * n + i];
  }
}

/* [Ceres-1.14] InvertPSDMatrix<3>: selfadjointView.llt().solve(Identity) */
static int inv3_llt(const double* A, double* Ainv)
{
  double l00 = A[0];
  if (!(l00 > 0)) return 1;
  l00 = sqrt(l00);
  double l10 = A[3] / l00, l20 = A[6] / l00;
  double d1 = A[4] - l10 * l10;
  if (!(d1 > 0)) return 1;
  double l11 = sqrt(d1);
  double l21 = (A[7] - l20 * l10) / l11;
  double d2 = A[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return 1;
  double l22 = sqrt(d2);
  for (int c = 0; c < 3; ++c) {
    double b0 = c == 0, b1 = c == 1, b2 = c == 2;
    double y0 = b0 / l00;
    double y1 = (b1 - l10 * y0) / l11;
    double y2 = (b2 - l20 * y0 - l21 * y1) / l22;
    double x2 = y2 / l22;
    double x1 = (y1 - l21 * x2) / l11;
    double x0 = (y0 - l10 * x1 - l20 * x2) / l00;
    Ainv[c] = x0; Ainv[3 + c] = x1; Ainv[6 + c] = x2;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* PTZ-IBA problem: evaluation with Ceres-style numeric or closed-form Jacobians               */
/* ------------------------------------------------------------------------------------------ */
#define MAX_NCF 9
#define CFREE_DISP 15 /* cfree values 15, 16, 17 denote the three displacement parameters (not part of the camera 15-vector) */

typedef struct {
  const orc_ba_problem* p;
  int jac_mode;
  int ncf;               /* free camera parameters per camera */
  int cfree[MAX_NCF];    /* indices into the 15-vector */
  int has_tlw;
  int has_disp;          /* PTZRayDistDisp: the global 3-parameter displacement block (disp_param_, ptzray_optimizer.cc:655).  It
                          * is carried as three extra slots of every camera that all map (cmap) to the slots of camera 0 --
                          * one parameter block shared by every residual, the same device as for shared intrinsics. */
  int n_cs;              /* camera-side system size = ncf * n_cam + 6 * has_tlw */
  int n_free;
  int n_amb;
  uint8_t* cam_active;   /* camera has >= 1 residual block (problem_.HasParameterBlock) */
  /* shared intrinsics: tangent index of camera i's free slot k (its own slot, or the slot of the first camera of its
   * intrinsics group for the intrinsic parameters); slots that nobody maps to stay in the vector as zero columns,
   * exactly like the reference's never-read fy */
  int* cmap;             /* [n_cam * ncf] */
  uint8_t* counts_intr;  /* [n_cam] 1 = this camera's copy of the intrinsics block is the one counted in |x| */
  int* first_of_group;   /* [n_cam] first camera (lowest index) with the same intrinsics id */
  /* stored linearisation (weighted; scaled in place by scale_cols) */
  double* Jc;   /* [n_obs][2][ncf] */
  double* Jr;   /* [n_obs][2][3]   */
  double* r;    /* [n_obs][2]      */
  double* cterm; /* [n_obs] per-observation cost terms */
  double* Jc3;  /* [n_obs3d][2][ncf] */
  double* Jt3;  /* [n_obs3d][2][6]  */
  double* r3;   /* [n_obs3d][2]     */
  /* solve workspace */
  double* S;    /* n_cs x n_cs */
  double* E;    /* [n_ray][9]  (V + D^2)^-1 */
  double* yr;   /* [n_ray][3] */
  int64_t* ray_ptr; /* [n_ray + 1] observation ranges */
} ba_ctx;

/* ambient x layout: [cam 15*n_cam | ray 3*n_ray | tlw 6 | disp 3] ; tangent: [cam ncf*n_cam | tlw 6*has | ray 3*n_ray] */
static inline const double* X_disp(const ba_ctx* c, const double* x) { return x + 15 * (size_t)c->p->n_cam + 3 * (size_t)c->p->n_ray + 6; }
static inline const double* X_cam(const ba_ctx* c, const double* x, int i) { (void)c; return x + 15 * (size_t)i; }
static inline const double* X_ray(const ba_ctx* c, const double* x, int j) { return x + 15 * (size_t)c->p->n_cam + 3 * (size_t)j; }
static inline const double* X_tlw(const ba_ctx* c, const double* x) { return x + 15 * (size_t)c->p->n_cam + 3 * (size_t)c->p->n_ray; }

static void cam_to_blocks(const double* cam, double* intr, double* extr)
{ /* SetUpInitialCameraParams, ptzray_optimizer.cc:647-651 */
  intr[0] = cam[0]; intr[1] = cam[1]; intr[2] = cam[2]; intr[3] = cam[3];
  intr[4] = cam[10]; intr[5] = cam[11]; intr[6] = cam[12]; intr[7] = cam[13]; intr[8] = cam[14];
  for (int k = 0; k < 6; ++k) extr[k] = cam[4 + k];
}
/* 15-vector index -> (block, index): intr index or 9 + extr index */
static int cam_idx_to_block18(int ci)
{
  if (ci <= 3) return ci;
  if (ci >= 10) return 4 + (ci - 10);
  return 9 + (ci - 4);
}

static void res2d2d(int type, const double* intr, const double* disp, const double* extr, const double* ray, const float* uv,
                    double* res)
{
  switch (type) {
    case ORC_PTZRay: orc_res_ptzray(intr, extr, ray, uv, res); break;
    case ORC_PTZRayDist: orc_res_ptzray_dist(intr, extr, ray, uv, res); break;
    case ORC_PTZRayDistDisp: orc_res_ptzray_dist_disp(intr, disp, extr, ray, uv, res); break;
    default: orc_res_ptzray_fxfy_dist(intr, extr, ray, uv, res); break;
  }
}
static void res2d3d(int type, const double* intr, const double* disp, const double* extr, const double* tlw, const float* uv,
                    const double* xyz, double* res)
{
  if (type == ORC_PTZRayDistDisp) orc_res_reproj2d3d_disp(intr, disp, extr, tlw, uv, xyz, res);
  else orc_res_reproj2d3d(intr, extr, tlw, uv, xyz, res);
}

/* [Ceres-1.14] numeric_diff.h NumericDiff<..., CENTRAL, ...>::EvaluateJacobianForParameterBlock:
 * delta = max(sqrt(eps), |x_j| * 1e-6); column = (f(x + delta e_j) - f(x - delta e_j)) * (1/delta / 2) */
static inline double nd_step(double xj)
{
  double min_step = sqrt(DBL_EPSILON);
  double s = fabs(xj) * 1e-6;
  return s > min_step ? s : min_step;
}

/* Evaluate one 2D-2D block: residual, Jacobian wrt the 18 block parameters [intr9 | extr6 | ... ] and ray3,
 * by central differences over ALL columns, exactly as NumericDiffCostFunction<F, CENTRAL, 2, 9, 6, 3>
 * (ptzray_optimizer.cc:60,133,197) does (37 functor calls). */
static void block2d2d_numeric(int type, const double* cam, const double* disp, const double* ray, const float* uv, double* res,
                              double* J15, double* Jray, double* Jd)
{
  double intr[9], extr[6], rr[3] = {ray[0], ray[1], ray[2]}, dd[3] = {0, 0, 0};
  if (disp) { dd[0] = disp[0]; dd[1] = disp[1]; dd[2] = disp[2]; }
  cam_to_blocks(cam, intr, extr);
  res2d2d(type, intr, dd, extr, rr, uv, res);
  if (type == ORC_PTZRayDistDisp) { /* NumericDiffCostFunction<PTZRayDistDispFactor, CENTRAL, 2, 9, 3, 6, 3> (:263) */
    double gp[2], gm[2];
    for (int j = 0; j < 3; ++j) {
      double x0 = dd[j], d = nd_step(x0);
      dd[j] = x0 + d; res2d2d(type, intr, dd, extr, rr, uv, gp);
      dd[j] = x0 - d; res2d2d(type, intr, dd, extr, rr, uv, gm);
      dd[j] = x0;
      double one_over = 1.0 / d; one_over /= 2;
      Jd[j] = (gp[0] - gm[0]) * one_over;
      Jd[3 + j] = (gp[1] - gm[1]) * one_over;
    }
  }
  double fp[2], fm[2];
  double J18[2][15];
  for (int j = 0; j < 9; ++j) {
    double x0 = intr[j], d = nd_step(x0);
    intr[j] = x0 + d; res2d2d(type, intr, dd, extr, rr, uv, fp);
    intr[j] = x0 - d; res2d2d(type, intr, dd, extr, rr, uv, fm);
    intr[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    J18[0][j] = (fp[0] - fm[0]) * one_over;
    J18[1][j] = (fp[1] - fm[1]) * one_over;
  }
  for (int j = 0; j < 6; ++j) {
    double x0 = extr[j], d = nd_step(x0);
    extr[j] = x0 + d; res2d2d(type, intr, dd, extr, rr, uv, fp);
    extr[j] = x0 - d; res2d2d(type, intr, dd, extr, rr, uv, fm);
    extr[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    J18[0][9 + j] = (fp[0] - fm[0]) * one_over;
    J18[1][9 + j] = (fp[1] - fm[1]) * one_over;
  }
  for (int j = 0; j < 3; ++j) {
    double x0 = rr[j], d = nd_step(x0);
    rr[j] = x0 + d; res2d2d(type, intr, dd, extr, rr, uv, fp);
    rr[j] = x0 - d; res2d2d(type, intr, dd, extr, rr, uv, fm);
    rr[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    Jray[j] = (fp[0] - fm[0]) * one_over;
    Jray[3 + j] = (fp[1] - fm[1]) * one_over;
  }
  for (int ci = 0; ci < 15; ++ci) {
    int b = cam_idx_to_block18(ci);
    J15[ci] = J18[0][b];
    J15[15 + ci] = J18[1][b];
  }
}

/* closed-form version of the same block.  J15: d res / d cam15 (only columns that can be free are
 * filled: 0, 1, 4..6, 10), Jray: d res / d ray. */
static void block2d2d_analytic(int type, const double* cam, const double* disp, const double* ray, const float* uv, double* res,
                               double* J15, double* Jray, double* Jd)
{
  memset(J15, 0, sizeof(double) * 30);
  double R[9], dR[27];
  orc_rodrigues_jac(cam + 4, R, dR);
  double fx = cam[0], fy = (type == ORC_PTZRayFxfyDist) ? cam[1] : cam[0], cx = cam[2], cy = cam[3];
  double X[3], dXdray[9]; /* point fed to R, and its derivative wrt ray */
  if (type == ORC_PTZRayDist) {
    X[0] = ray[0]; X[1] = ray[1]; X[2] = ray[2];
    for (int e = 0; e < 9; ++e) dXdray[e] = (e % 4 == 0) ? 1.0 : 0.0;
  }
  else {
    double n = vec3_norm(ray), in = 1.0 / n;
    X[0] = ray[0] * in; X[1] = ray[1] * in; X[2] = ray[2] * in;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) dXdray[3 * i + j] = ((i == j ? 1.0 : 0.0) - X[i] * X[j]) * in;
  }
  double P[3];
  mat3_mul_vec(R, X, P);
  if (type == ORC_PTZRayDist && P[2] < 0) {
    res[0] = 1000000.0; res[1] = 1000000.0;
    memset(Jray, 0, sizeof(double) * 6);
    return;
  }
  double ddf = 0; /* d displacement / d fx */
  if (type == ORC_PTZRayDistDisp) {
    P[2] += disp[0] + disp[1] * fx + disp[2] * fx * fx;
    ddf = disp[1] + 2.0 * disp[2] * fx;
  }
  double iz = 1.0 / P[2];
  double x = P[0] / P[2], y = P[1] / P[2]; /* same arithmetic as the functor */
  /* d(x,y)/dP */
  double dpi[6] = {iz, 0, -x * iz, 0, iz, -y * iz};
  double xd = x, yd = y, B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (type != ORC_PTZRay) {
    brown(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], &xd, &yd);
    brown_jac(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], B, dk1);
  }
  if (type == ORC_PTZRay) { /* same arithmetic as the functor: (f Px + cx Pz) / Pz */
    double px = fx * P[0] + cx * P[2], py = fx * P[1] + cy * P[2];
    res[0] = (double)uv[0] - px / P[2];
    res[1] = (double)uv[1] - py / P[2];
  }
  else {
    res[0] = (double)uv[0] - (fx * xd + cx);
    res[1] = (double)uv[1] - (fy * yd + cy);
  }
  /* d pred / dP = diag(fx, fy) * B * dpi  (2x3) */
  double M[6];
  for (int j = 0; j < 3; ++j) {
    M[j] = fx * (B[0] * dpi[j] + B[1] * dpi[3 + j]);
    M[3 + j] = fy * (B[2] * dpi[j] + B[3] * dpi[3 + j]);
  }
  /* focal */
  if (type == ORC_PTZRayFxfyDist) {
    J15[0] = -xd; J15[15 + 1] = -yd;
  }
  else {
    J15[0] = -xd; J15[15 + 0] = -yd;
  }
  if (type != ORC_PTZRay) { /* k1 */
    J15[10] = -fx * dk1[0];
    J15[15 + 10] = -fy * dk1[1];
  }
  if (type == ORC_PTZRayDistDisp) { /* P.z also moves with fx; displacement block: dP.z/d(d0,d1,d2) = (1, fx, fx^2) */
    J15[0] -= M[2] * ddf;
    J15[15 + 0] -= M[5] * ddf;
    const double pw[3] = {1.0, fx, fx * fx};
    for (int k = 0; k < 3; ++k) { Jd[k] = -M[2] * pw[k]; Jd[3 + k] = -M[5] * pw[k]; }
  }
  /* rotation: dP/dr_k = dR_k X */
  for (int k = 0; k < 3; ++k) {
    double dP[3];
    mat3_mul_vec(dR + 9 * k, X, dP);
    J15[4 + k] = -(M[0] * dP[0] + M[1] * dP[1] + M[2] * dP[2]);
    J15[15 + 4 + k] = -(M[3] * dP[0] + M[4] * dP[1] + M[5] * dP[2]);
  }
  /* ray: dP/dray = R dX/dray */
  double RdX[9];
  mat3_mul(R, dXdray, RdX);
  for (int j = 0; j < 3; ++j) {
    Jray[j] = -(M[0] * RdX[j] + M[1] * RdX[3 + j] + M[2] * RdX[6 + j]);
    Jray[3 + j] = -(M[3] * RdX[j] + M[4] * RdX[3 + j] + M[5] * RdX[6 + j]);
  }
}

/* 2D-3D block (Reproj2d3dFactor, NumericDiffCostFunction<..., 2, 9, 6, 6>, ptzray_optimizer.cc:330) */
static void block2d3d_numeric(int type, const double* cam, const double* disp, const double* tlw, const float* uv,
                              const double* xyz, double* res, double* J15, double* Jt, double* Jd)
{
  double intr[9], extr[6], tt[6], dd[3] = {0, 0, 0};
  if (disp) { dd[0] = disp[0]; dd[1] = disp[1]; dd[2] = disp[2]; }
  cam_to_blocks(cam, intr, extr);
  memcpy(tt, tlw, sizeof(tt));
  res2d3d(type, intr, dd, extr, tt, uv, xyz, res);
  if (type == ORC_PTZRayDistDisp) { /* NumericDiffCostFunction<Reproj2d3dDispFactor, CENTRAL, 2, 9, 3, 6, 6> (:400) */
    double gp[2], gm[2];
    for (int j = 0; j < 3; ++j) {
      double x0 = dd[j], d = nd_step(x0);
      dd[j] = x0 + d; res2d3d(type, intr, dd, extr, tt, uv, xyz, gp);
      dd[j] = x0 - d; res2d3d(type, intr, dd, extr, tt, uv, xyz, gm);
      dd[j] = x0;
      double one_over = 1.0 / d; one_over /= 2;
      Jd[j] = (gp[0] - gm[0]) * one_over;
      Jd[3 + j] = (gp[1] - gm[1]) * one_over;
    }
  }
  double fp[2], fm[2], J18[2][15];
  for (int j = 0; j < 9; ++j) {
    double x0 = intr[j], d = nd_step(x0);
    intr[j] = x0 + d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fp);
    intr[j] = x0 - d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fm);
    intr[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    J18[0][j] = (fp[0] - fm[0]) * one_over; J18[1][j] = (fp[1] - fm[1]) * one_over;
  }
  for (int j = 0; j < 6; ++j) {
    double x0 = extr[j], d = nd_step(x0);
    extr[j] = x0 + d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fp);
    extr[j] = x0 - d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fm);
    extr[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    J18[0][9 + j] = (fp[0] - fm[0]) * one_over; J18[1][9 + j] = (fp[1] - fm[1]) * one_over;
  }
  for (int j = 0; j < 6; ++j) {
    double x0 = tt[j], d = nd_step(x0);
    tt[j] = x0 + d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fp);
    tt[j] = x0 - d; res2d3d(type, intr, dd, extr, tt, uv, xyz, fm);
    tt[j] = x0;
    double one_over = 1.0 / d; one_over /= 2;
    Jt[j] = (fp[0] - fm[0]) * one_over; Jt[6 + j] = (fp[1] - fm[1]) * one_over;
  }
  for (int ci = 0; ci < 15; ++ci) {
    int b = cam_idx_to_block18(ci);
    J15[ci] = J18[0][b];
    J15[15 + ci] = J18[1][b];
  }
}

static void block2d3d_analytic(int type, const double* cam, const double* disp, const double* tlw, const float* uv,
                               const double* xyz, double* res, double* J15, double* Jt, double* Jd)
{
  memset(J15, 0, sizeof(double) * 30);
  double R[9], dR[27], Rlw[9], dRlw[27];
  orc_rodrigues_jac(cam + 4, R, dR);
  orc_rodrigues_jac(tlw, Rlw, dRlw);
  double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
  double Xl[3], P[3];
  mat3_mul_vec(Rlw, xyz, Xl);
  Xl[0] += tlw[3]; Xl[1] += tlw[4]; Xl[2] += tlw[5];
  mat3_mul_vec(R, Xl, P);
  double ddf = 0;
  if (type == ORC_PTZRayDistDisp) {
    P[2] += disp[0] + disp[1] * fx + disp[2] * fx * fx;
    ddf = disp[1] + 2.0 * disp[2] * fx;
  }
  double iz = 1.0 / P[2], x = P[0] / P[2], y = P[1] / P[2]; /* same arithmetic as the functor */
  double dpi[6] = {iz, 0, -x * iz, 0, iz, -y * iz};
  double xd, yd, B[4], dk1[2];
  brown(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], &xd, &yd);
  brown_jac(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], B, dk1);
  res[0] = (double)uv[0] - (fx * xd + cx);
  res[1] = (double)uv[1] - (fy * yd + cy);
  double M[6];
  for (int j = 0; j < 3; ++j) {
    M[j] = fx * (B[0] * dpi[j] + B[1] * dpi[3 + j]);
    M[3 + j] = fy * (B[2] * dpi[j] + B[3] * dpi[3 + j]);
  }
  J15[0] = -xd;
  J15[15 + 1] = -yd;
  J15[10] = -fx * dk1[0];
  J15[15 + 10] = -fy * dk1[1];
  if (type == ORC_PTZRayDistDisp) {
    J15[0] -= M[2] * ddf;
    J15[15 + 0] -= M[5] * ddf;
    const double pw[3] = {1.0, fx, fx * fx};
    for (int k = 0; k < 3; ++k) { Jd[k] = -M[2] * pw[k]; Jd[3 + k] = -M[5] * pw[k]; }
  }
  for (int k = 0; k < 3; ++k) {
    double dP[3];
    mat3_mul_vec(dR + 9 * k, Xl, dP);
    J15[4 + k] = -(M[0] * dP[0] + M[1] * dP[1] + M[2] * dP[2]);
    J15[15 + 4 + k] = -(M[3] * dP[0] + M[4] * dP[1] + M[5] * dP[2]);
    double dXl[3], dP2[3];
    mat3_mul_vec(dRlw + 9 * k, xyz, dXl);
    mat3_mul_vec(R, dXl, dP2);
    Jt[k] = -(M[0] * dP2[0] + M[1] * dP2[1] + M[2] * dP2[2]);
    Jt[6 + k] = -(M[3] * dP2[0] + M[4] * dP2[1] + M[5] * dP2[2]);
  }
  for (int k = 0; k < 3; ++k) { /* translation part: dP = R e_k */
    Jt[3 + k] = -(M[0] * R[k] + M[1] * R[3 + k] + M[2] * R[6 + k]);
    Jt[6 + 3 + k] = -(M[3] * R[k] + M[4] * R[3 + k] + M[5] * R[6 + k]);
  }
}

int32_t orc_ba_cam_free_dim(int32_t factor_type) { return factor_type == ORC_PTZRay ? 5 : (factor_type == ORC_PTZRayDistDisp ? 9 : 6); }

static int ba_ctx_init(ba_ctx* c, const orc_ba_problem* p, int jac_mode)
{
  memset(c, 0, sizeof(*c));
  c->p = p;
  c->jac_mode = jac_mode;
  /* SubsetParameterization (ptzray_optimizer.cc:863,870,881): free intr {fx,fy} (+k1 for *Dist*), free extr rvec */
  if (p->factor_type == ORC_PTZRay) {
    c->ncf = 5;
    int f[5] = {0, 1, 4, 5, 6};
    memcpy(c->cfree, f, sizeof(f));
  }
  else if (p->factor_type == ORC_PTZRayDistDisp) {
    /* intrinsics {fx, fy, k1} (:868-871), rvec, and the displacement block disp_param_ (no parameterization: all 3 free) */
    c->ncf = 9;
    c->has_disp = 1;
    int f[9] = {0, 1, 10, 4, 5, 6, CFREE_DISP, CFREE_DISP + 1, CFREE_DISP + 2};
    memcpy(c->cfree, f, sizeof(f));
  }
  else {
    c->ncf = 6;
    int f[6] = {0, 1, 10, 4, 5, 6};
    memcpy(c->cfree, f, sizeof(f));
  }
  c->has_tlw = p->n_obs3d > 0;
  c->n_cs = c->ncf * p->n_cam + 6 * c->has_tlw;
  c->n_free = c->n_cs + 3 * p->n_ray;
  c->n_amb = 15 * p->n_cam + 3 * p->n_ray + 6 + 3;
  c->cam_active = (uint8_t*)calloc((size_t)p->n_cam + 1, 1);
  for (int64_t a = 0; a < p->n_obs; ++a) {
    if (p->obs_cam[a] < 0 || p->obs_cam[a] >= p->n_cam || p->obs_ray[a] < 0 || p->obs_ray[a] >= p->n_ray) return 1;
    if (a > 0 && p->obs_ray[a] < p->obs_ray[a - 1]) return 1;
    c->cam_active[p->obs_cam[a]] = 1;
  }
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    if (p->obs3d_cam[a] < 0 || p->obs3d_cam[a] >= p->n_cam) return 1;
    c->cam_active[p->obs3d_cam[a]] = 1;
  }
  c->cmap = (int*)malloc(sizeof(int) * ((size_t)p->n_cam * c->ncf + 1));
  c->counts_intr = (uint8_t*)calloc((size_t)p->n_cam + 1, 1);
  c->first_of_group = (int*)malloc(sizeof(int) * ((size_t)p->n_cam + 1));
  for (int i = 0; i < p->n_cam; ++i) {
    int first = i;
    if (p->ic_of_cam)
      for (int m = 0; m < i; ++m)
        if (p->ic_of_cam[m] == p->ic_of_cam[i]) { first = m; break; }
    c->first_of_group[i] = first;
    for (int k = 0; k < c->ncf; ++k) {
      int is_intr = c->cfree[k] < 4 || c->cfree[k] >= 10;
      if (c->cfree[k] >= CFREE_DISP) c->cmap[i * c->ncf + k] = k; /* the one displacement block: slots of camera 0 */
      else c->cmap[i * c->ncf + k] = (is_intr ? first : i) * c->ncf + k;
    }
  }
  /* the shared block is in the problem if any member has residuals; it is counted once, at its first active member */
  for (int i = 0; i < p->n_cam; ++i) {
    if (!c->cam_active[i]) continue;
    int seen = 0;
    for (int m = 0; m < i; ++m)
      if (c->cam_active[m] && c->first_of_group[m] == c->first_of_group[i]) { seen = 1; break; }
    c->counts_intr[i] = !seen;
  }
  c->Jc = (double*)malloc(sizeof(double) * (size_t)(p->n_obs * 2 * c->ncf + 1));
  c->Jr = (double*)malloc(sizeof(double) * (size_t)(p->n_obs * 6 + 1));
  c->r = (double*)malloc(sizeof(double) * (size_t)(p->n_obs * 2 + 1));
  c->cterm = (double*)malloc(sizeof(double) * (size_t)(p->n_obs + 1));
  c->Jc3 = (double*)malloc(sizeof(double) * (size_t)(p->n_obs3d * 2 * c->ncf + 1));
  c->Jt3 = (double*)malloc(sizeof(double) * (size_t)(p->n_obs3d * 12 + 1));
  c->r3 = (double*)malloc(sizeof(double) * (size_t)(p->n_obs3d * 2 + 1));
  c->S = (double*)malloc(sizeof(double) * ((size_t)c->n_cs * c->n_cs + 1));
  c->E = (double*)malloc(sizeof(double) * (size_t)(9 * p->n_ray + 1));
  c->yr = (double*)malloc(sizeof(double) * (size_t)(3 * p->n_ray + 1));
  c->ray_ptr = (int64_t*)calloc((size_t)p->n_ray + 2, sizeof(int64_t));
  for (int64_t a = 0; a < p->n_obs; ++a) ++c->ray_ptr[p->obs_ray[a] + 1];
  for (int32_t j = 0; j < p->n_ray; ++j) c->ray_ptr[j + 1] += c->ray_ptr[j];
  return 0;
}

static void ba_ctx_free(ba_ctx* c)
{
  free(c->cterm); free(c->cam_active); free(c->cmap); free(c->counts_intr); free(c->first_of_group); free(c->Jc); free(c->Jr); free(c->r); free(c->Jc3); free(c->Jt3); free(c->r3);
  free(c->S); free(c->E); free(c->yr); free(c->ray_ptr);
}

static double ba_cost(void* vc, const double* x)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  double cost = 0;
  /* per-observation terms in parallel, then a serial sum: bitwise reproducible for any thread count */
#pragma omp parallel for schedule(static)
  for (int64_t a = 0; a < p->n_obs; ++a) {
    double intr[9], extr[6], res[2];
    cam_to_blocks(X_cam(c, x, p->obs_cam[a]), intr, extr);
    res2d2d(p->factor_type, intr, X_disp(c, x), extr, X_ray(c, x, p->obs_ray[a]), p->obs_uv + 2 * a, res);
    /* ScaledLoss(NULL, w): cost = 0.5 * w * |r|^2  (ptzray_optimizer.cc:805-806) */
    c->cterm[a] = 0.5 * (p->ray_weight[p->obs_ray[a]] * (res[0] * res[0] + res[1] * res[1]));
  }
  for (int64_t a = 0; a < p->n_obs; ++a) cost += c->cterm[a];
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    double intr[9], extr[6], res[2];
    cam_to_blocks(X_cam(c, x, p->obs3d_cam[a]), intr, extr);
    res2d3d(p->factor_type, intr, X_disp(c, x), extr, X_tlw(c, x), p->obs3d_uv + 2 * a, p->obs3d_xyz + 3 * a, res);
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
  }
  return cost;
}

static double ba_linearize(void* vc, const double* x, double* g)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf;
  double cost = 0;
#pragma omp parallel for schedule(static)
  for (int64_t a = 0; a < p->n_obs; ++a) {
    double res[2], J15[30], Jray[6], Jd[6] = {0, 0, 0, 0, 0, 0};
    const double* cam = X_cam(c, x, p->obs_cam[a]);
    const double* ray = X_ray(c, x, p->obs_ray[a]);
    if (c->jac_mode == ORC_JAC_NUMERIC) block2d2d_numeric(p->factor_type, cam, X_disp(c, x), ray, p->obs_uv + 2 * a, res, J15, Jray, Jd);
    else block2d2d_analytic(p->factor_type, cam, X_disp(c, x), ray, p->obs_uv + 2 * a, res, J15, Jray, Jd);
    double w = p->ray_weight[p->obs_ray[a]];
    double sw = sqrt(w); /* Corrector: residuals, jacobians *= sqrt(rho') */
    c->cterm[a] = 0.5 * (w * (res[0] * res[0] + res[1] * res[1]));
    c->r[2 * a] = res[0] * sw;
    c->r[2 * a + 1] = res[1] * sw;
    for (int k = 0; k < ncf; ++k) {
      const int f = c->cfree[k];
      c->Jc[(2 * a) * ncf + k] = (f >= CFREE_DISP ? Jd[f - CFREE_DISP] : J15[f]) * sw;
      c->Jc[(2 * a + 1) * ncf + k] = (f >= CFREE_DISP ? Jd[3 + f - CFREE_DISP] : J15[15 + f]) * sw;
    }
    for (int k = 0; k < 6; ++k) c->Jr[6 * a + k] = Jray[k] * sw;
  }
  for (int64_t a = 0; a < p->n_obs; ++a) cost += c->cterm[a];
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    double res[2], J15[30], Jt[12], Jd[6] = {0, 0, 0, 0, 0, 0};
    const double* cam = X_cam(c, x, p->obs3d_cam[a]);
    if (c->jac_mode == ORC_JAC_NUMERIC)
      block2d3d_numeric(p->factor_type, cam, X_disp(c, x), X_tlw(c, x), p->obs3d_uv + 2 * a, p->obs3d_xyz + 3 * a, res, J15, Jt, Jd);
    else
      block2d3d_analytic(p->factor_type, cam, X_disp(c, x), X_tlw(c, x), p->obs3d_uv + 2 * a, p->obs3d_xyz + 3 * a, res, J15, Jt, Jd);
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    c->r3[2 * a] = res[0];
    c->r3[2 * a + 1] = res[1];
    for (int k = 0; k < ncf; ++k) {
      /* PTZRay keeps k1 constant; for *Dist* types k1 (cfree[2] = 10) is free */
      const int f = c->cfree[k];
      c->Jc3[(2 * a) * ncf + k] = f >= CFREE_DISP ? Jd[f - CFREE_DISP] : J15[f];
      c->Jc3[(2 * a + 1) * ncf + k] = f >= CFREE_DISP ? Jd[3 + f - CFREE_DISP] : J15[15 + f];
    }
    memcpy(c->Jt3 + 12 * (size_t)a, Jt, sizeof(double) * 12);
  }
  /* gradient = J^T r (tangent space, unscaled) */
  if (g) {
    memset(g, 0, sizeof(double) * (size_t)c->n_free);
    double* gc = g;
    double* gt = g + ncf * p->n_cam;
    double* gr = g + c->n_cs;
    for (int64_t a = 0; a < p->n_obs; ++a) {
      int ci = p->obs_cam[a], rj = p->obs_ray[a];
      for (int k = 0; k < ncf; ++k)
        gc[c->cmap[ci * ncf + k]] += c->Jc[(2 * a) * ncf + k] * c->r[2 * a] + c->Jc[(2 * a + 1) * ncf + k] * c->r[2 * a + 1];
      for (int k = 0; k < 3; ++k) gr[3 * rj + k] += c->Jr[6 * a + k] * c->r[2 * a] + c->Jr[6 * a + 3 + k] * c->r[2 * a + 1];
    }
    for (int32_t a = 0; a < p->n_obs3d; ++a) {
      int ci = p->obs3d_cam[a];
      for (int k = 0; k < ncf; ++k)
        gc[c->cmap[ci * ncf + k]] += c->Jc3[(2 * a) * ncf + k] * c->r3[2 * a] + c->Jc3[(2 * a + 1) * ncf + k] * c->r3[2 * a + 1];
      for (int k = 0; k < 6; ++k) gt[k] += c->Jt3[12 * a + k] * c->r3[2 * a] + c->Jt3[12 * a + 6 + k] * c->r3[2 * a + 1];
    }
  }
  return cost;
}

static void ba_col_sqnorm(void* vc, double* out)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf;
  memset(out, 0, sizeof(double) * (size_t)c->n_free);
  double* oc = out;
  double* ot = out + ncf * p->n_cam;
  double* orr = out + c->n_cs;
  for (int64_t a = 0; a < p->n_obs; ++a) {
    int ci = p->obs_cam[a], rj = p->obs_ray[a];
    for (int k = 0; k < ncf; ++k) {
      double j0 = c->Jc[(2 * a) * ncf + k], j1 = c->Jc[(2 * a + 1) * ncf + k];
      oc[c->cmap[ci * ncf + k]] += j0 * j0 + j1 * j1;
    }
    for (int k = 0; k < 3; ++k) orr[3 * rj + k] += c->Jr[6 * a + k] * c->Jr[6 * a + k] + c->Jr[6 * a + 3 + k] * c->Jr[6 * a + 3 + k];
  }
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    int ci = p->obs3d_cam[a];
    for (int k = 0; k < ncf; ++k) {
      double j0 = c->Jc3[(2 * a) * ncf + k], j1 = c->Jc3[(2 * a + 1) * ncf + k];
      oc[c->cmap[ci * ncf + k]] += j0 * j0 + j1 * j1;
    }
    for (int k = 0; k < 6; ++k) ot[k] += c->Jt3[12 * a + k] * c->Jt3[12 * a + k] + c->Jt3[12 * a + 6 + k] * c->Jt3[12 * a + 6 + k];
  }
}

static void ba_scale_cols(void* vc, const double* s)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf;
  const double* sc = s;
  const double* st = s + ncf * p->n_cam;
  const double* sr = s + c->n_cs;
  for (int64_t a = 0; a < p->n_obs; ++a) {
    int ci = p->obs_cam[a], rj = p->obs_ray[a];
    for (int k = 0; k < ncf; ++k) {
      c->Jc[(2 * a) * ncf + k] *= sc[c->cmap[ci * ncf + k]];
      c->Jc[(2 * a + 1) * ncf + k] *= sc[c->cmap[ci * ncf + k]];
    }
    for (int k = 0; k < 3; ++k) {
      c->Jr[6 * a + k] *= sr[3 * rj + k];
      c->Jr[6 * a + 3 + k] *= sr[3 * rj + k];
    }
  }
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    int ci = p->obs3d_cam[a];
    for (int k = 0; k < ncf; ++k) {
      c->Jc3[(2 * a) * ncf + k] *= sc[c->cmap[ci * ncf + k]];
      c->Jc3[(2 * a + 1) * ncf + k] *= sc[c->cmap[ci * ncf + k]];
    }
    for (int k = 0; k < 6; ++k) {
      c->Jt3[12 * a + k] *= st[k];
      c->Jt3[12 * a + 6 + k] *= st[k];
    }
  }
}

/* [Ceres-1.14] SchurComplementSolver / SchurEliminator (SPARSE_SCHUR, ptzray_optimizer.cc:471):
 * e-blocks = rays (3x3).  Solves (J^T J + D^2) y = J^T r; any exact elimination order gives the
 * same y up to round-off, so the reduced camera system is factored densely here. */
/* Add v to entry (i, j) AND (j, i) of the symmetric camera system, of which only the lower triangle is stored.
 * With shared intrinsics two different (camera, slot) pairs can map to one index: then both mirrored contributions land on
 * the same diagonal entry. */
static inline void S_add_pair(double* S, int n, int i, int j, double v)
{
  if (i == j) S[(size_t)i * n + i] += 2.0 * v;
  else if (i > j) S[(size_t)i * n + j] += v;
  else S[(size_t)j * n + i] += v;
}
/* Add v to entry (i, j) of a diagonal block (k >= l visited once): mirrored implicitly by the symmetric storage. */
static inline void S_add_lower(double* S, int n, int i, int j, double v)
{
  if (i >= j) S[(size_t)i * n + j] += v;
  else S[(size_t)j * n + i] += v;
}

static int ba_solve(void* vc, const double* D, double* y)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf, n = c->n_cs;
  const double* Dc = D;
  const double* Dr = D + c->n_cs;
  double* S = c->S;
  double* b = y; /* camera-side rhs lives in y[0..n) */
  memset(S, 0, sizeof(double) * (size_t)n * n);
  memset(y, 0, sizeof(double) * (size_t)c->n_free);
  for (int i = 0; i < n; ++i) S[(size_t)i * n + i] = Dc[i] * Dc[i];
  /* camera diagonal blocks and rhs from 2D-2D observations */
  for (int64_t a = 0; a < p->n_obs; ++a) {
    int ci = p->obs_cam[a];
    const double* j0 = c->Jc + (2 * a) * ncf;
    const double* j1 = j0 + ncf;
    for (int k = 0; k < ncf; ++k) {
      b[c->cmap[ci * ncf + k]] += j0[k] * c->r[2 * a] + j1[k] * c->r[2 * a + 1];
      for (int l = 0; l <= k; ++l) S_add_lower(S, n, c->cmap[ci * ncf + k], c->cmap[ci * ncf + l], j0[k] * j0[l] + j1[k] * j1[l]);
    }
  }
  /* 2D-3D observations: camera block, tlw block, camera-tlw coupling */
  const int t0 = ncf * p->n_cam;
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    int ci = p->obs3d_cam[a];
    const double* j0 = c->Jc3 + (2 * a) * ncf;
    const double* j1 = j0 + ncf;
    const double* q0 = c->Jt3 + 12 * (size_t)a;
    const double* q1 = q0 + 6;
    for (int k = 0; k < ncf; ++k) {
      b[c->cmap[ci * ncf + k]] += j0[k] * c->r3[2 * a] + j1[k] * c->r3[2 * a + 1];
      for (int l = 0; l <= k; ++l) S_add_lower(S, n, c->cmap[ci * ncf + k], c->cmap[ci * ncf + l], j0[k] * j0[l] + j1[k] * j1[l]);
    }
    for (int k = 0; k < 6; ++k) {
      b[t0 + k] += q0[k] * c->r3[2 * a] + q1[k] * c->r3[2 * a + 1];
      for (int l = 0; l <= k; ++l) S[(size_t)(t0 + k) * n + t0 + l] += q0[k] * q0[l] + q1[k] * q1[l];
      for (int l = 0; l < ncf; ++l) S[(size_t)(t0 + k) * n + c->cmap[ci * ncf + l]] += q0[k] * j0[l] + q1[k] * j1[l];
    }
  }
  /* eliminate rays */
  int fail = 0;
  for (int32_t j = 0; j < p->n_ray; ++j) {
    int64_t a0 = c->ray_ptr[j], a1 = c->ray_ptr[j + 1];
    double V[9] = {0}, br[3] = {0};
    for (int64_t a = a0; a < a1; ++a) {
      const double* q0 = c->Jr + 6 * a;
      const double* q1 = q0 + 3;
      for (int k = 0; k < 3; ++k) {
        br[k] += q0[k] * c->r[2 * a] + q1[k] * c->r[2 * a + 1];
        for (int l = 0; l < 3; ++l) V[3 * k + l] += q0[k] * q0[l] + q1[k] * q1[l];
      }
    }
    for (int k = 0; k < 3; ++k) V[4 * k] += Dr[3 * j + k] * Dr[3 * j + k];
    double* E = c->E + 9 * (size_t)j;
    if (inv3_llt(V, E)) { fail = 1; break; }
    double z[3]; /* E * br */
    mat3_mul_vec(E, br, z);
    y[c->n_cs + 3 * j] = br[0]; y[c->n_cs + 3 * j + 1] = br[1]; y[c->n_cs + 3 * j + 2] = br[2];
    /* W_a = Jc_a^T Jr_a (ncf x 3), Y_a = W_a E */
    double Wbuf[64][MAX_NCF * 3], Ybuf[64][MAX_NCF * 3];
    int64_t L = a1 - a0;
    double (*W)[MAX_NCF * 3] = Wbuf;
    double (*Y)[MAX_NCF * 3] = Ybuf;
    double* heap = NULL;
    if (L > 64) {
      heap = (double*)malloc(sizeof(double) * (size_t)L * MAX_NCF * 3 * 2);
      W = (double(*)[MAX_NCF * 3])heap;
      Y = (double(*)[MAX_NCF * 3])(heap + (size_t)L * MAX_NCF * 3);
    }
    for (int64_t a = a0; a < a1; ++a) {
      const double* j0 = c->Jc + (2 * a) * ncf;
      const double* j1 = j0 + ncf;
      const double* q0 = c->Jr + 6 * a;
      const double* q1 = q0 + 3;
      double* Wa = W[a - a0];
      double* Ya = Y[a - a0];
      for (int k = 0; k < ncf; ++k)
        for (int l = 0; l < 3; ++l) Wa[3 * k + l] = j0[k] * q0[l] + j1[k] * q1[l];
      for (int k = 0; k < ncf; ++k)
        for (int l = 0; l < 3; ++l) Ya[3 * k + l] = Wa[3 * k] * E[l] + Wa[3 * k + 1] * E[3 + l] + Wa[3 * k + 2] * E[6 + l];
      int ci = p->obs_cam[a];
      for (int k = 0; k < ncf; ++k) b[c->cmap[ci * ncf + k]] -= Wa[3 * k] * z[0] + Wa[3 * k + 1] * z[1] + Wa[3 * k + 2] * z[2];
    }
    for (int64_t a = a0; a < a1; ++a)
      for (int64_t bb = a0; bb < a1; ++bb) {
        int ci = p->obs_cam[a], cj = p->obs_cam[bb];
        if (cj > ci) continue; /* lower triangle of S (block level) */
        const double* Ya = Y[a - a0];
        const double* Wb = W[bb - a0];
        for (int k = 0; k < ncf; ++k)
          for (int l = 0; l < ncf; ++l) {
            if (ci == cj && l > k) continue;
            const double v = -(Ya[3 * k] * Wb[3 * l] + Ya[3 * k + 1] * Wb[3 * l + 1] + Ya[3 * k + 2] * Wb[3 * l + 2]);
            if (ci == cj) S_add_lower(S, n, c->cmap[ci * ncf + k], c->cmap[cj * ncf + l], v);
            else S_add_pair(S, n, c->cmap[ci * ncf + k], c->cmap[cj * ncf + l], v);
          }
      }
    free(heap);
  }
  if (fail) return 1;
  if (chol_lower(S, n)) return 1;
  chol_solve(S, n, b);
  /* back-substitute: y_r = E (b_r - sum_a W_a^T y_c) */
  for (int32_t j = 0; j < p->n_ray; ++j) {
    double t[3] = {y[c->n_cs + 3 * j], y[c->n_cs + 3 * j + 1], y[c->n_cs + 3 * j + 2]};
    for (int64_t a = c->ray_ptr[j]; a < c->ray_ptr[j + 1]; ++a) {
      int ci = p->obs_cam[a];
      const double* j0 = c->Jc + (2 * a) * ncf;
      const double* j1 = j0 + ncf;
      const double* q0 = c->Jr + 6 * a;
      const double* q1 = q0 + 3;
      double m0 = 0, m1 = 0; /* Jc_a y_c */
      for (int k = 0; k < ncf; ++k) { m0 += j0[k] * b[c->cmap[ci * ncf + k]]; m1 += j1[k] * b[c->cmap[ci * ncf + k]]; }
      for (int l = 0; l < 3; ++l) t[l] -= q0[l] * m0 + q1[l] * m1;
    }
    mat3_mul_vec(c->E + 9 * (size_t)j, t, y + c->n_cs + 3 * j);
  }
  return 0;
}

static double ba_model_cost_change(void* vc, const double* step)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf;
  const double* sc = step;
  const double* st = step + ncf * p->n_cam;
  const double* sr = step + c->n_cs;
  double acc = 0;
  for (int64_t a = 0; a < p->n_obs; ++a) {
    int ci = p->obs_cam[a], rj = p->obs_ray[a];
    double m0 = 0, m1 = 0;
    for (int k = 0; k < ncf; ++k) { m0 += c->Jc[(2 * a) * ncf + k] * sc[c->cmap[ci * ncf + k]]; m1 += c->Jc[(2 * a + 1) * ncf + k] * sc[c->cmap[ci * ncf + k]]; }
    for (int k = 0; k < 3; ++k) { m0 += c->Jr[6 * a + k] * sr[3 * rj + k]; m1 += c->Jr[6 * a + 3 + k] * sr[3 * rj + k]; }
    acc += m0 * (c->r[2 * a] + m0 / 2.0) + m1 * (c->r[2 * a + 1] + m1 / 2.0);
  }
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    int ci = p->obs3d_cam[a];
    double m0 = 0, m1 = 0;
    for (int k = 0; k < ncf; ++k) { m0 += c->Jc3[(2 * a) * ncf + k] * sc[c->cmap[ci * ncf + k]]; m1 += c->Jc3[(2 * a + 1) * ncf + k] * sc[c->cmap[ci * ncf + k]]; }
    for (int k = 0; k < 6; ++k) { m0 += c->Jt3[12 * a + k] * st[k]; m1 += c->Jt3[12 * a + 6 + k] * st[k]; }
    acc += m0 * (c->r3[2 * a] + m0 / 2.0) + m1 * (c->r3[2 * a + 1] + m1 / 2.0);
  }
  return -acc;
}

static void ba_plus(void* vc, const double* x, const double* delta, double* xo)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  const int ncf = c->ncf;
  memcpy(xo, x, sizeof(double) * (size_t)c->n_amb);
  for (int i = 0; i < p->n_cam; ++i)
    for (int k = 0; k < ncf; ++k)
      if (c->cfree[k] < CFREE_DISP) xo[15 * (size_t)i + c->cfree[k]] += delta[c->cmap[i * ncf + k]];
  if (c->has_disp)
    for (int k = 0; k < ncf; ++k)
      if (c->cfree[k] >= CFREE_DISP)
        xo[15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray + 6 + (c->cfree[k] - CFREE_DISP)] += delta[c->cmap[k]];
  if (c->has_tlw)
    for (int k = 0; k < 6; ++k) xo[15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray + k] += delta[ncf * p->n_cam + k];
  for (int j = 0; j < 3 * p->n_ray; ++j) xo[15 * (size_t)p->n_cam + j] += delta[c->n_cs + j];
}

/* ambient norms run over the parameter blocks that are in the ceres::Problem: intr(9)+extr(6) of
 * cameras with >= 1 residual, every ray, and tlw when 2D-3D residuals exist */
static double ba_diff_norm(void* vc, const double* a, const double* b)
{
  ba_ctx* c = (ba_ctx*)vc;
  const orc_ba_problem* p = c->p;
  double acc = 0;
  for (int i = 0; i < p->n_cam; ++i) {
    if (!c->cam_active[i]) continue;
    for (int k = 0; k < 15; ++k) {
      /* a shared intrinsics block (indices 0-3, 10-14) is one parameter block: counted at one member only */
      if ((k < 4 || k >= 10) && !c->counts_intr[i]) continue;
      double d = a[15 * (size_t)i + k] - (b ? b[15 * (size_t)i + k] : 0.0);
      acc += d * d;
    }
  }
  size_t o = 15 * (size_t)p->n_cam;
  for (size_t j = 0; j < 3 * (size_t)p->n_ray; ++j) {
    double d = a[o + j] - (b ? b[o + j] : 0.0);
    acc += d * d;
  }
  if (c->has_tlw)
    for (int k = 0; k < 6; ++k) {
      double d = a[o + 3 * (size_t)p->n_ray + k] - (b ? b[o + 3 * (size_t)p->n_ray + k] : 0.0);
      acc += d * d;
    }
  if (c->has_disp)
    for (int k = 0; k < 3; ++k) {
      double d = a[o + 3 * (size_t)p->n_ray + 6 + k] - (b ? b[o + 3 * (size_t)p->n_ray + 6 + k] : 0.0);
      acc += d * d;
    }
  return sqrt(acc);
}
static double ba_x_norm(void* vc, const double* x) { return ba_diff_norm(vc, x, NULL); }
static double ba_grad_max(void* vc, const double* g)
{
  ba_ctx* c = (ba_ctx*)vc;
  double m = 0;
  for (int i = 0; i < c->n_free; ++i) m = fmax(m, fabs(g[i]));
  return m;
}

static void set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int32_t orc_ba_solve(const orc_ba_problem* p, double* cam, double* ray, double* tlw, const orc_lm_options* o,
                     orc_lm_summary* s, orc_lm_trace* trace)
{
  return orc_ba_solve_disp(p, cam, ray, tlw, NULL, o, s, trace);
}

int32_t orc_ba_solve_disp(const orc_ba_problem* p, double* cam, double* ray, double* tlw, double* disp, const orc_lm_options* o,
                          orc_lm_summary* s, orc_lm_trace* trace)
{
  ba_ctx c;
  if (ba_ctx_init(&c, p, o->jacobian_mode)) { ba_ctx_free(&c); return 1; }
  set_threads(o->num_threads);
  double* x = (double*)malloc(sizeof(double) * (size_t)c.n_amb);
  memcpy(x, cam, sizeof(double) * 15 * (size_t)p->n_cam);
  /* intrinsics_param_.insert({ic_id, ...}) keeps the FIRST camera's values for a shared block (ptzray_optimizer.cc:645-650) */
  for (int i = 0; i < p->n_cam; ++i) {
    const int f = c.first_of_group[i];
    if (f == i) continue;
    for (int k = 0; k < 15; ++k)
      if (k < 4 || k >= 10) x[15 * (size_t)i + k] = x[15 * (size_t)f + k];
  }
  memcpy(x + 15 * (size_t)p->n_cam, ray, sizeof(double) * 3 * (size_t)p->n_ray);
  memcpy(x + 15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray, tlw, sizeof(double) * 6);
  for (int k = 0; k < 3; ++k) x[15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray + 6 + k] = disp ? disp[k] : 0.0; /* :655 */
  lm_problem P = {c.n_free, c.n_amb, &c, ba_cost, ba_linearize, ba_col_sqnorm, ba_scale_cols, ba_solve,
                  ba_model_cost_change, ba_plus, ba_x_norm, ba_diff_norm, ba_grad_max};
  lm_minimize(&P, x, o, s, trace);
  s->num_residuals = (int32_t)(2 * p->n_obs + 2 * p->n_obs3d);
  memcpy(cam, x, sizeof(double) * 15 * (size_t)p->n_cam);
  memcpy(ray, x + 15 * (size_t)p->n_cam, sizeof(double) * 3 * (size_t)p->n_ray);
  memcpy(tlw, x + 15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray, sizeof(double) * 6);
  if (disp) memcpy(disp, x + 15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray + 6, sizeof(double) * 3);
  free(x);
  ba_ctx_free(&c);
  return 0;
}

int32_t orc_ba_residuals(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw, double* res)
{
  return orc_ba_residuals_disp(p, cam, ray, tlw, NULL, res);
}

int32_t orc_ba_residuals_disp(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                              const double* disp, double* res)
{
  const double zero3[3] = {0, 0, 0};
  if (!disp) disp = zero3;
  for (int64_t a = 0; a < p->n_obs; ++a) {
    double intr[9], extr[6];
    cam_to_blocks(cam + 15 * (size_t)p->obs_cam[a], intr, extr);
    res2d2d(p->factor_type, intr, disp, extr, ray + 3 * (size_t)p->obs_ray[a], p->obs_uv + 2 * a, res + 2 * a);
  }
  for (int32_t a = 0; a < p->n_obs3d; ++a) {
    double intr[9], extr[6];
    cam_to_blocks(cam + 15 * (size_t)p->obs3d_cam[a], intr, extr);
    res2d3d(p->factor_type, intr, disp, extr, tlw, p->obs3d_uv + 2 * a, p->obs3d_xyz + 3 * a, res + 2 * p->n_obs + 2 * a);
  }
  return 0;
}

int32_t orc_ba_linearize(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                         int32_t jacobian_mode, double* cost, double* g_c, double* U, double* g_r, double* V, double* W)
{
  return orc_ba_linearize_disp(p, cam, ray, tlw, NULL, jacobian_mode, cost, g_c, U, g_r, V, W);
}

int32_t orc_ba_linearize_disp(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                              const double* disp, int32_t jacobian_mode, double* cost, double* g_c, double* U, double* g_r,
                              double* V, double* W)
{
  ba_ctx c;
  if (ba_ctx_init(&c, p, jacobian_mode)) { ba_ctx_free(&c); return 1; }
  const int ncf = c.ncf;
  double* x = (double*)malloc(sizeof(double) * (size_t)c.n_amb);
  memcpy(x, cam, sizeof(double) * 15 * (size_t)p->n_cam);
  memcpy(x + 15 * (size_t)p->n_cam, ray, sizeof(double) * 3 * (size_t)p->n_ray);
  memcpy(x + 15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray, tlw, sizeof(double) * 6);
  for (int k = 0; k < 3; ++k) x[15 * (size_t)p->n_cam + 3 * (size_t)p->n_ray + 6 + k] = disp ? disp[k] : 0.0;
  double* g = (double*)malloc(sizeof(double) * (size_t)c.n_free);
  double cst = ba_linearize(&c, x, g);
  if (cost) *cost = cst;
  if (g_c) memcpy(g_c, g, sizeof(double) * (size_t)ncf * p->n_cam);
  if (g_r) memcpy(g_r, g + c.n_cs, sizeof(double) * 3 * (size_t)p->n_ray);
  if (U) {
    memset(U, 0, sizeof(double) * (size_t)ncf * ncf * p->n_cam);
    for (int64_t a = 0; a < p->n_obs; ++a) {
      int ci = p->obs_cam[a];
      const double* j0 = c.Jc + (2 * a) * ncf;
      const double* j1 = j0 + ncf;
      for (int k = 0; k < ncf; ++k)
        for (int l = 0; l < ncf; ++l) U[((size_t)ci * ncf + k) * ncf + l] += j0[k] * j0[l] + j1[k] * j1[l];
    }
    for (int32_t a = 0; a < p->n_obs3d; ++a) {
      int ci = p->obs3d_cam[a];
      const double* j0 = c.Jc3 + (2 * a) * ncf;
      const double* j1 = j0 + ncf;
      for (int k = 0; k < ncf; ++k)
        for (int l = 0; l < ncf; ++l) U[((size_t)ci * ncf + k) * ncf + l] += j0[k] * j0[l] + j1[k] * j1[l];
    }
  }
  if (V) {
    memset(V, 0, sizeof(double) * 9 * (size_t)p->n_ray);
    for (int64_t a = 0; a < p->n_obs; ++a) {
      int rj = p->obs_ray[a];
      const double* q0 = c.Jr + 6 * a;
      const double* q1 = q0 + 3;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) V[9 * (size_t)rj + 3 * k + l] += q0[k] * q0[l] + q1[k] * q1[l];
    }
  }
  if (W) {
    for (int64_t a = 0; a < p->n_obs; ++a) {
      const double* j0 = c.Jc + (2 * a) * ncf;
      const double* j1 = j0 + ncf;
      const double* q0 = c.Jr + 6 * a;
      const double* q1 = q0 + 3;
      for (int k = 0; k < ncf; ++k)
        for (int l = 0; l < 3; ++l) W[(size_t)a * ncf * 3 + 3 * k + l] = j0[k] * q0[l] + j1[k] * q1[l];
    }
  }
  free(g);
  free(x);
  ba_ctx_free(&c);
  return 0;
}

/* Pix2Ray  ptzray_optimizer.cc:768-797: ray = normalise(mean_i normalise(R_i^-1 K_i^-1 [u,v,1])) */
void orc_pix2ray(const orc_ba_problem* p, const double* cam, double* ray)
{
  int64_t a = 0;
  for (int32_t j = 0; j < p->n_ray; ++j) {
    double acc[3] = {0, 0, 0};
    int64_t cnt = 0;
    for (; a < p->n_obs && p->obs_ray[a] == j; ++a) {
      const double* cv = cam + 15 * (size_t)p->obs_cam[a];
      double R[9], Ri[9];
      orc_rodrigues(cv + 4, R);
      mat3_inv(R, Ri);
      /* K^-1 [u, v, 1] with K = [[fx,0,cx],[0,fy,cy],[0,0,1]] */
      double q[3] = {((double)p->obs_uv[2 * a] - cv[2]) / cv[0], ((double)p->obs_uv[2 * a + 1] - cv[3]) / cv[1], 1.0};
      double t[3];
      mat3_mul_vec(Ri, q, t);
      double n = vec3_norm(t);
      acc[0] += t[0] / n; acc[1] += t[1] / n; acc[2] += t[2] / n;
      ++cnt;
    }
    acc[0] /= (double)cnt; acc[1] /= (double)cnt; acc[2] /= (double)cnt;
    double n = vec3_norm(acc);
    ray[3 * j] = acc[0] / n; ray[3 * j + 1] = acc[1] / n; ray[3 * j + 2] = acc[2] / n;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* KRT single-view LM  (krt_optimizer.cc:265-404)                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const orc_krt_problem* p;
  int jac_mode;
  int nf;
  int fidx[6];
  double k1[4];
  double dist1[5];
  double* J; /* [2M][nf] */
  double* r; /* [2M] */
  double* A; /* QR workspace (2M + nf) x nf, column-major */
  double* rhs;
} krt_ctx;

/* cvProjectPoints2Internal (OpenCV 4.5.3 calibration.cpp) for one point, no tilt, 5 distortion coefficients */
static void cv_project_point(const double* R, const double* t, double fx, double fy, double cx, double cy, const double* k,
                             const double* X, double* uv)
{
  double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  z = z ? 1. / z : 1;
  x *= z; y *= z;
  double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
  double xd = x * cdist + k[2] * a1 + k[3] * a2;
  double yd = y * cdist + k[2] * a3 + k[3] * a1;
  uv[0] = xd * fx + cx;
  uv[1] = yd * fy + cy;
}

void orc_res_2d3d_krt(const double* cam, int32_t fxfy, const float* pt2d, const double* pt3d_local, double* res)
{
  /* krt_optimizer.cc:202-216 (param[1] = param[0]) / :228-241; cam.rvec() is Rodrigues(R(rvec)), the round trip is
   * the identity to round-off and is not restated */
  double R[9], uv[2];
  orc_rodrigues(cam + 4, R);
  cv_project_point(R, cam + 7, cam[0], fxfy ? cam[1] : cam[0], cam[2], cam[3], cam + 10, pt3d_local, uv);
  res[0] = (double)pt2d[0] - uv[0];
  res[1] = (double)pt2d[1] - uv[1];
}

void orc_krt_point_to_local(const double* ref, const double* Xw, double* Xl)
{
  double R[9];
  orc_rodrigues(ref + 4, R);
  mat3_mul_vec(R, Xw, Xl);
  Xl[0] += ref[7]; Xl[1] += ref[8]; Xl[2] += ref[9];
}

static void krt_res(const krt_ctx* c, const double* cam, int m, double* res)
{
  const orc_krt_problem* p = c->p;
  double cv[15];
  memcpy(cv, cam, sizeof(cv));
  if (m >= p->n_match) { /* 2D-3D block (krt_optimizer.cc:364-381) */
    int i = m - p->n_match;
    orc_res_2d3d_krt(cv, p->factor_type == ORC_KRT_Fxfy || p->factor_type == ORC_KRT_FxfyDist, p->pts2d + 2 * i,
                     p->pts3d_local + 3 * i, res);
    return;
  }
  switch (p->factor_type) {
    case ORC_KRT_F:
      orc_res_2d2d(cv, c->k1, p->uv_ref + 2 * m, p->uv_cur + 2 * m, res);
      break;
    case ORC_KRT_FDist:
      orc_res_2d2d_dist(cv, c->k1, c->dist1, p->uv_ref + 2 * m, p->uv_cur + 2 * m, res);
      break;
    default: { /* Fxfy / FxfyDist (krt_optimizer.cc:52-71, 141-192): dead from the CLI */
      double R[9], ray1[3], P[3];
      orc_rodrigues(cv + 4, R);
      if (p->factor_type == ORC_KRT_Fxfy) {
        /* Factor2d2dFxfy does NOT normalise ray1 (:61) -- irrelevant after the perspective divide */
        krt_ray1(c->k1, (double)p->uv_ref[2 * m], (double)p->uv_ref[2 * m + 1], ray1);
        mat3_mul_vec(R, ray1, P);
        res[0] = (double)p->uv_cur[2 * m] - (cv[0] * P[0] + cv[2] * P[2]) / P[2];
        res[1] = (double)p->uv_cur[2 * m + 1] - (cv[1] * P[1] + cv[3] * P[2]) / P[2];
      }
      else {
        float und[2];
        orc_undistort_point(c->k1, c->dist1, p->uv_ref + 2 * m, und);
        if (und[0] < 0 || und[0] >= c->k1[2] * 2 || und[1] < 0 || und[1] >= c->k1[3] * 2) { res[0] = res[1] = 0; return; }
        krt_ray1(c->k1, (double)und[0], (double)und[1], ray1);
        mat3_mul_vec(R, ray1, P);
        double x = P[0] / P[2], y = P[1] / P[2], xd, yd;
        brown(x, y, cv[10], cv[11], cv[12], cv[13], cv[14], &xd, &yd);
        res[0] = (double)p->uv_cur[2 * m] - (cv[0] * xd + cv[2]);
        res[1] = (double)p->uv_cur[2 * m + 1] - (cv[1] * yd + cv[3]);
      }
    }
  }
}

static void krt_block_analytic(const krt_ctx* c, const double* cam, int m, double* res, double* J15)
{
  const orc_krt_problem* p = c->p;
  memset(J15, 0, sizeof(double) * 30);
  int type = p->factor_type;
  int dist = (type == ORC_KRT_FDist || type == ORC_KRT_FxfyDist);
  int fxfy = (type == ORC_KRT_Fxfy || type == ORC_KRT_FxfyDist);
  if (m >= p->n_match) { /* 2D-3D block: closed-form Jacobian of the projectPoints residual */
    const double* X = p->pts3d_local + 3 * (m - p->n_match);
    double R[9], dR[27], Q[3];
    orc_rodrigues_jac(cam + 4, R, dR);
    mat3_mul_vec(R, X, Q);
    double P[3] = {Q[0] + cam[7], Q[1] + cam[8], Q[2] + cam[9]};
    double iz = P[2] ? 1.0 / P[2] : 1.0, x = P[0] * iz, y = P[1] * iz;
    double fx = cam[0], fy = fxfy ? cam[1] : cam[0];
    double xd, yd, B[4], dk1[2];
    /* stored (d0..d4) read by OpenCV as (k1,k2,p1,p2,k3) */
    brown(x, y, cam[10], cam[11], cam[14], cam[12], cam[13], &xd, &yd);
    brown_jac(x, y, cam[10], cam[11], cam[14], cam[12], cam[13], B, dk1);
    res[0] = (double)p->pts2d[2 * (m - p->n_match)] - (xd * fx + cam[2]);
    res[1] = (double)p->pts2d[2 * (m - p->n_match) + 1] - (yd * fy + cam[3]);
    double dpi[6] = {iz, 0, -x * iz, 0, iz, -y * iz}, M[6];
    for (int j = 0; j < 3; ++j) {
      M[j] = fx * (B[0] * dpi[j] + B[1] * dpi[3 + j]);
      M[3 + j] = fy * (B[2] * dpi[j] + B[3] * dpi[3 + j]);
    }
    if (fxfy) { J15[0] = -xd; J15[15 + 1] = -yd; }
    else { J15[0] = -xd; J15[15] = -yd; }
    if (dist) { J15[10] = -fx * dk1[0]; J15[15 + 10] = -fy * dk1[1]; }
    for (int k = 0; k < 3; ++k) {
      double dP[3];
      mat3_mul_vec(dR + 9 * k, X, dP);
      J15[4 + k] = -(M[0] * dP[0] + M[1] * dP[1] + M[2] * dP[2]);
      J15[15 + 4 + k] = -(M[3] * dP[0] + M[4] * dP[1] + M[5] * dP[2]);
    }
    return;
  }
  double u1 = p->uv_ref[2 * m], v1 = p->uv_ref[2 * m + 1];
  if (dist) {
    float und[2];
    orc_undistort_point(c->k1, c->dist1, p->uv_ref + 2 * m, und);
    if (und[0] < 0 || und[0] >= c->k1[2] * 2 || und[1] < 0 || und[1] >= c->k1[3] * 2) { res[0] = res[1] = 0; return; }
    u1 = und[0]; v1 = und[1];
  }
  double R[9], dR[27], X[3], P[3];
  orc_rodrigues_jac(cam + 4, R, dR);
  krt_ray1(c->k1, u1, v1, X);
  mat3_mul_vec(R, X, P);
  double fx = cam[0], fy = fxfy ? cam[1] : cam[0], cx = cam[2], cy = cam[3];
  double iz = 1.0 / P[2], x = P[0] / P[2], y = P[1] / P[2]; /* same arithmetic as the functor */
  double dpi[6] = {iz, 0, -x * iz, 0, iz, -y * iz};
  double xd = x, yd = y, B[4] = {1, 0, 0, 1}, dk1[2] = {0, 0};
  if (dist) {
    brown(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], &xd, &yd);
    brown_jac(x, y, cam[10], cam[11], cam[12], cam[13], cam[14], B, dk1);
    res[0] = (double)p->uv_cur[2 * m] - (fx * xd + cx);
    res[1] = (double)p->uv_cur[2 * m + 1] - (fy * yd + cy);
  }
  else {
    res[0] = (double)p->uv_cur[2 * m] - (fx * P[0] + cx * P[2]) / P[2];
    res[1] = (double)p->uv_cur[2 * m + 1] - (fy * P[1] + cy * P[2]) / P[2];
  }
  double M[6];
  for (int j = 0; j < 3; ++j) {
    M[j] = fx * (B[0] * dpi[j] + B[1] * dpi[3 + j]);
    M[3 + j] = fy * (B[2] * dpi[j] + B[3] * dpi[3 + j]);
  }
  if (fxfy) { J15[0] = -xd; J15[15 + 1] = -yd; }
  else { J15[0] = -xd; J15[15] = -yd; }
  if (dist) { J15[10] = -fx * dk1[0]; J15[15 + 10] = -fy * dk1[1]; }
  for (int k = 0; k < 3; ++k) {
    double dP[3];
    mat3_mul_vec(dR + 9 * k, X, dP);
    J15[4 + k] = -(M[0] * dP[0] + M[1] * dP[1] + M[2] * dP[2]);
    J15[15 + 4 + k] = -(M[3] * dP[0] + M[4] * dP[1] + M[5] * dP[2]);
  }
}

static int krt_blocks(const krt_ctx* c) { return c->p->n_match + (c->p->pts2d ? c->p->n_pt : 0); }

static double krt_cost(void* vc, const double* x)
{
  krt_ctx* c = (krt_ctx*)vc;
  double cost = 0;
  for (int m = 0; m < krt_blocks(c); ++m) {
    double res[2];
    krt_res(c, x, m, res);
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
  }
  return cost;
}

static double krt_linearize(void* vc, const double* x, double* g)
{
  krt_ctx* c = (krt_ctx*)vc;
  const int nf = c->nf, M = krt_blocks(c);
  double cost = 0;
  for (int m = 0; m < M; ++m) {
    double res[2], J15[30];
    if (c->jac_mode == ORC_JAC_NUMERIC) {
      /* NumericDiffCostFunction<Factor2d2d*, CENTRAL, 2, 15> (krt_optimizer.cc:47,75,136,196): 31 functor calls */
      double xx[15], fp[2], fm[2];
      memcpy(xx, x, sizeof(xx));
      krt_res(c, xx, m, res);
      for (int j = 0; j < 15; ++j) {
        double x0 = xx[j], d = nd_step(x0);
        xx[j] = x0 + d; krt_res(c, xx, m, fp);
        xx[j] = x0 - d; krt_res(c, xx, m, fm);
        xx[j] = x0;
        double one_over = 1.0 / d; one_over /= 2;
        J15[j] = (fp[0] - fm[0]) * one_over;
        J15[15 + j] = (fp[1] - fm[1]) * one_over;
      }
    }
    else {
      krt_block_analytic(c, x, m, res, J15);
    }
    cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    c->r[2 * m] = res[0];
    c->r[2 * m + 1] = res[1];
    for (int k = 0; k < nf; ++k) {
      c->J[(size_t)(2 * m) * nf + k] = J15[c->fidx[k]];
      c->J[(size_t)(2 * m + 1) * nf + k] = J15[15 + c->fidx[k]];
    }
  }
  if (g) {
    for (int k = 0; k < nf; ++k) {
      double acc = 0;
      for (int row = 0; row < 2 * M; ++row) acc += c->J[(size_t)row * nf + k] * c->r[row];
      g[k] = acc;
    }
  }
  return cost;
}

static void krt_col_sqnorm(void* vc, double* out)
{
  krt_ctx* c = (krt_ctx*)vc;
  for (int k = 0; k < c->nf; ++k) {
    double acc = 0;
    for (int row = 0; row < 2 * krt_blocks(c); ++row) acc += c->J[(size_t)row * c->nf + k] * c->J[(size_t)row * c->nf + k];
    out[k] = acc;
  }
}
static void krt_scale_cols(void* vc, const double* s)
{
  krt_ctx* c = (krt_ctx*)vc;
  for (int row = 0; row < 2 * krt_blocks(c); ++row)
    for (int k = 0; k < c->nf; ++k) c->J[(size_t)row * c->nf + k] *= s[k];
}

/* [Ceres-1.14] DenseQRSolver (DENSE_QR, krt_optimizer.cc:389): x = [J; D].householderQr().solve([r; 0]) */
static int krt_solve(void* vc, const double* D, double* y)
{
  krt_ctx* c = (krt_ctx*)vc;
  const int nf = c->nf, rows = 2 * krt_blocks(c), R = rows + nf;
  double* A = c->A; /* column-major R x nf */
  double* b = c->rhs;
  for (int k = 0; k < nf; ++k) {
    for (int row = 0; row < rows; ++row) A[(size_t)k * R + row] = c->J[(size_t)row * nf + k];
    for (int l = 0; l < nf; ++l) A[(size_t)k * R + rows + l] = (l == k) ? D[k] : 0.0;
  }
  for (int row = 0; row < rows; ++row) b[row] = c->r[row];
  for (int l = 0; l < nf; ++l) b[rows + l] = 0.0;
  /* Householder QR (Eigen::HouseholderQR, unblocked: make_householder + applyHouseholderOnTheLeft) */
  for (int k = 0; k < nf; ++k) {
    double* col = A + (size_t)k * R;
    double c0 = col[k], tail = 0;
    for (int i = k + 1; i < R; ++i) tail += col[i] * col[i];
    double tau, beta;
    if (tail <= DBL_MIN) {
      tau = 0; beta = c0;
      for (int i = k + 1; i < R; ++i) col[i] = 0;
    }
    else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < R; ++i) col[i] /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    col[k] = beta;
    /* apply H = I - tau v v^T (v = [1; essential]) to remaining columns and rhs */
    for (int l = k + 1; l <= nf; ++l) {
      double* t = (l < nf) ? A + (size_t)l * R : b;
      double dot = t[k];
      for (int i = k + 1; i < R; ++i) dot += col[i] * t[i];
      dot *= tau;
      t[k] -= dot;
      for (int i = k + 1; i < R; ++i) t[i] -= dot * col[i];
    }
  }
  for (int k = nf - 1; k >= 0; --k) {
    double v = b[k];
    for (int l = k + 1; l < nf; ++l) v -= A[(size_t)l * R + k] * y[l];
    double d = A[(size_t)k * R + k];
    if (d == 0.0) return 1;
    y[k] = v / d;
  }
  return 0;
}

static double krt_model_cost_change(void* vc, const double* step)
{
  krt_ctx* c = (krt_ctx*)vc;
  double acc = 0;
  for (int row = 0; row < 2 * krt_blocks(c); ++row) {
    double m = 0;
    for (int k = 0; k < c->nf; ++k) m += c->J[(size_t)row * c->nf + k] * step[k];
    acc += m * (c->r[row] + m / 2.0);
  }
  return -acc;
}
static void krt_plus(void* vc, const double* x, const double* d, double* xo)
{
  krt_ctx* c = (krt_ctx*)vc;
  memcpy(xo, x, sizeof(double) * 15);
  for (int k = 0; k < c->nf; ++k) xo[c->fidx[k]] += d[k];
}
static double krt_diff_norm(void* vc, const double* a, const double* b)
{
  (void)vc;
  double acc = 0;
  for (int k = 0; k < 15; ++k) {
    double d = a[k] - (b ? b[k] : 0.0);
    acc += d * d;
  }
  return sqrt(acc);
}
static double krt_x_norm(void* vc, const double* x) { return krt_diff_norm(vc, x, NULL); }
static double krt_grad_max(void* vc, const double* g)
{
  krt_ctx* c = (krt_ctx*)vc;
  double m = 0;
  for (int k = 0; k < c->nf; ++k) m = fmax(m, fabs(g[k]));
  return m;
}

int32_t orc_krt_solve(const orc_krt_problem* p, double* cam, const orc_lm_options* o, orc_lm_summary* s, orc_lm_trace* trace)
{
  krt_ctx c;
  memset(&c, 0, sizeof(c));
  c.p = p;
  c.jac_mode = o->jacobian_mode;
  /* SubsetParameterization, krt_optimizer.cc:321-346 (free indices in ascending order) */
  switch (p->factor_type) {
    case ORC_KRT_F: { int f[4] = {0, 4, 5, 6}; c.nf = 4; memcpy(c.fidx, f, sizeof(f)); break; }
    case ORC_KRT_FDist: { int f[5] = {0, 4, 5, 6, 10}; c.nf = 5; memcpy(c.fidx, f, sizeof(f)); break; }
    case ORC_KRT_Fxfy: { int f[5] = {0, 1, 4, 5, 6}; c.nf = 5; memcpy(c.fidx, f, sizeof(f)); break; }
    default: { int f[6] = {0, 1, 4, 5, 6, 10}; c.nf = 6; memcpy(c.fidx, f, sizeof(f)); break; }
  }
  /* cam_ref_local: K and dist of the reference, R = I, t = 0 (krt_optimizer.cc:272-276) */
  c.k1[0] = p->cam_ref[0]; c.k1[1] = p->cam_ref[1]; c.k1[2] = p->cam_ref[2]; c.k1[3] = p->cam_ref[3];
  for (int k = 0; k < 5; ++k) c.dist1[k] = p->cam_ref[10 + k];
  int rows = 2 * krt_blocks(&c);
  c.J = (double*)malloc(sizeof(double) * ((size_t)rows * c.nf + 1));
  c.r = (double*)malloc(sizeof(double) * ((size_t)rows + 1));
  c.A = (double*)malloc(sizeof(double) * ((size_t)(rows + c.nf) * c.nf + 1));
  c.rhs = (double*)malloc(sizeof(double) * ((size_t)rows + c.nf + 1));
  set_threads(o->num_threads);
  lm_problem P = {c.nf, 15, &c, krt_cost, krt_linearize, krt_col_sqnorm, krt_scale_cols, krt_solve,
                  krt_model_cost_change, krt_plus, krt_x_norm, krt_diff_norm, krt_grad_max};
  lm_minimize(&P, cam, o, s, trace);
  s->num_residuals = rows;
  free(c.J); free(c.r); free(c.A); free(c.rhs);
  return 0;
}

static void cam_vec_to_Rt(const double* cam, double* R, double* t)
{
  orc_rodrigues(cam + 4, R);
  t[0] = cam[7]; t[1] = cam[8]; t[2] = cam[9];
}

/* krt_optimizer.cc:269-284: cam_curr_local = (K, dist, R_cur R_ref^-1, -R_cur R_ref^-1 t_ref + t_cur).ToVector() */
void orc_krt_world_to_local(const double* ref, const double* cur, double* out)
{
  double Rr[9], tr[3], Rc[9], tc[3], Rri[9], Rl[9];
  cam_vec_to_Rt(ref, Rr, tr);
  cam_vec_to_Rt(cur, Rc, tc);
  mat3_inv(Rr, Rri);
  mat3_mul(Rc, Rri, Rl);
  double t[3];
  mat3_mul_vec(Rl, tr, t);
  memcpy(out, cur, sizeof(double) * 15);
  orc_rodrigues_inv(Rl, out + 4);
  out[7] = -t[0] + tc[0]; out[8] = -t[1] + tc[1]; out[9] = -t[2] + tc[2];
}

/* krt_optimizer.cc:535-567 */
void orc_krt_local_to_world(const double* ref, const double* loc, int32_t factor_type, double* out)
{
  double cv[15];
  memcpy(cv, loc, sizeof(cv));
  if (factor_type == ORC_KRT_F || factor_type == ORC_KRT_FDist) cv[1] = cv[0]; /* fx = fy (:543) */
  double Rr[9], tr[3], Rl[9], tl[3], Rw[9], t[3];
  cam_vec_to_Rt(ref, Rr, tr);
  cam_vec_to_Rt(cv, Rl, tl);
  mat3_mul(Rl, Rr, Rw);
  mat3_mul_vec(Rl, tr, t);
  memcpy(out, cv, sizeof(double) * 15);
  orc_rodrigues_inv(Rw, out + 4);
  out[7] = t[0] + tl[0]; out[8] = t[1] + tl[1]; out[9] = t[2] + tl[2];
}

/* KRTOptimizer::CheckResults  krt_optimizer.cc:504-533 */
int32_t orc_krt_check(const orc_lm_summary* s, const double* cam, double max_reproj_error)
{
  double final_reproj_error = sqrt(2) * sqrt((2 * s->final_cost) / s->num_residuals);
  if (s->termination_type != ORC_CONVERGENCE) return 0;
  if (final_reproj_error >= max_reproj_error) return 0;
  double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
  double fov_x = atan(cx / fx) * 2 * 180 / M_PI;
  double fov_y = atan(cy / fy) * 2 * 180 / M_PI;
  if (fov_x < 0 || fov_x > 170 || fov_y < 0 || fov_y > 170) return 0;
  return 1;
}

/* run_ptz_reloc.cc:68-118 as a loop over packed queries (see ptz_oracle.h) */
int32_t orc_krt_solve_batch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                            const double* cam_ref_world, double* cam_cur_world, int32_t factor_type, double max_reproj_error,
                            const orc_lm_options* o, orc_lm_summary* summaries, int32_t* accepted, int32_t num_threads)
{
  orc_lm_options oo = *o;
  oo.num_threads = 1; /* the threads are spent on whole queries */
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(num_threads)
  for (int32_t q = 0; q < n_query; ++q) {
    orc_krt_problem p;
    memset(&p, 0, sizeof(p));
    p.n_match = (int32_t)(match_ptr[q + 1] - match_ptr[q]);
    p.uv_ref = uv_ref + 2 * match_ptr[q];
    p.uv_cur = uv_cur + 2 * match_ptr[q];
    p.cam_ref = cam_ref_world + 15 * (size_t)q;
    p.factor_type = factor_type;
    double loc[15];
    orc_krt_world_to_local(p.cam_ref, cam_cur_world + 15 * (size_t)q, loc);
    orc_lm_summary s;
    memset(&s, 0, sizeof(s));
    orc_krt_solve(&p, loc, &oo, &s, NULL);
    const int32_t ok = orc_krt_check(&s, loc, max_reproj_error);
    if (ok) orc_krt_local_to_world(p.cam_ref, loc, factor_type, cam_cur_world + 15 * (size_t)q);
    if (summaries) summaries[q] = s;
    if (accepted) accepted[q] = ok;
  }
  return 0;
}
