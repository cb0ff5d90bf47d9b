// factor_harness.cc -- TEST INFRASTRUCTURE.  Host instantiation of the device math in
// ptz-calib_amd/csrc/ptz_factor.h so the algebra can be checked against the oracle without a GPU.
// Never part of the product library.
#include "../../ptz-calib_amd/csrc/ptz_factor.h"

using namespace ptz;

static void fill_camblk(const double* cam15, double* cb)
{
  rodrigues(cam15 + 4, cb + CB_R);
  so3_left_jacobian(cam15 + 4, cb + CB_JL);
  cb[CB_F] = cam15[0]; cb[CB_CX] = cam15[2]; cb[CB_CY] = cam15[3]; cb[CB_FY] = cam15[1];
  for (int k = 0; k < 5; ++k) cb[CB_K + k] = cam15[10 + k];
  for (int k = 0; k < 6; ++k) cb[CB_S + k] = 1.0;
}

// ba_pair_side (the factored Jacobians of k_schur's phase 2) brought back to ba_linearize's form: Jc = G with the rotation
// columns taken through Jl, Jr = -MR / |X|.  type 0..3; disp only read for type 3.  Returns 0 for the zero-Jacobian branch.
template <int T> static int pair_side_t(const double* cb, const double* ray, double* Jc, double* Jr)
{
  constexpr int NC = BaDims<T>::NC, ROT0 = BaDims<T>::ROT0;
  double Xn[3], inv_n, MR[2][3], G[2][NC];
  ba_ray_point<T>(ray, Xn, inv_n);
  const bool ok = ba_pair_side<T>(cb + CB_R, cb[CB_F], T == 2 ? cb[CB_FY] : cb[CB_F], cb + CB_K, cb + CB_D, Xn, MR, G);
  if (!ok) { for (int i = 0; i < 2 * NC; ++i) Jc[i] = 0; for (int i = 0; i < 6; ++i) Jr[i] = 0; return 0; }
  const double* Jl = cb + CB_JL;
  for (int r = 0; r < 2; ++r) {
    for (int k = 0; k < NC; ++k) Jc[r * NC + k] = G[r][k];
    for (int k = 0; k < 3; ++k)
      Jc[r * NC + ROT0 + k] = G[r][ROT0] * Jl[k] + G[r][ROT0 + 1] * Jl[3 + k] + G[r][ROT0 + 2] * Jl[6 + k];
    for (int k = 0; k < 3; ++k) Jr[r * 3 + k] = -MR[r][k] * inv_n;
  }
  return 1;
}

extern "C" {
void h_rodrigues(const double* r, double* R, double* Jl) { rodrigues(r, R); so3_left_jacobian(r, Jl); }

// Jc: [2][NC], Jr: [2][3]
void h_ba_linearize(int type, const double* cam15, const double* ray, const float* uv, double* res, double* Jc, double* Jr)
{
  double cb[CAMBLK];
  fill_camblk(cam15, cb);
  double jr[2][3];
  if (type == 0) {
    double jc[2][4];
    ba_linearize<0>(cb, ray, uv[0], uv[1], res, jc, jr);
    for (int i = 0; i < 8; ++i) Jc[i] = (&jc[0][0])[i];
  }
  else if (type == 1) {
    double jc[2][5];
    ba_linearize<1>(cb, ray, uv[0], uv[1], res, jc, jr);
    for (int i = 0; i < 10; ++i) Jc[i] = (&jc[0][0])[i];
  }
  else {
    double jc[2][6];
    ba_linearize<2>(cb, ray, uv[0], uv[1], res, jc, jr);
    for (int i = 0; i < 12; ++i) Jc[i] = (&jc[0][0])[i];
  }
  for (int i = 0; i < 6; ++i) Jr[i] = (&jr[0][0])[i];
}
void h_ba_residual(int type, const double* cam15, const double* ray, const float* uv, double* res)
{
  double cb[CAMBLK];
  fill_camblk(cam15, cb);
  if (type == 0) ba_residual<0>(cb, ray, uv[0], uv[1], res);
  else ba_residual<1>(cb, ray, uv[0], uv[1], res);
}
// directional variant used by the evaluation kernel: v = camera step in ba_linearize's column order (NC entries)
void h_ba_step_dir(int type, const double* cam15, const double* ray, const float* uv, const double* v, double* res, double* p, double* Jr)
{
  double cb[CAMBLK];
  fill_camblk(cam15, cb);
  const int nc = type == 0 ? 4 : (type == 1 ? 5 : 6);
  const double* Jl = cb + CB_JL;
  double om[3];
  for (int r = 0; r < 3; ++r) om[r] = Jl[3 * r] * v[nc - 3] + Jl[3 * r + 1] * v[nc - 2] + Jl[3 * r + 2] * v[nc - 1];
  double jr[2][3];
  if (type == 0) ba_step_dir<0>(cb, ray, uv[0], uv[1], v, om, res, p, jr);
  else if (type == 1) ba_step_dir<1>(cb, ray, uv[0], uv[1], v, om, res, p, jr);
  else ba_step_dir<2>(cb, ray, uv[0], uv[1], v, om, res, p, jr);
  for (int i = 0; i < 6; ++i) Jr[i] = (&jr[0][0])[i];
}
// KRT: cam15 current (local frame), k1[4], dist1[5] reference intrinsics/distortion
void h_krt_eval(int ktype, const double* cam15, const double* k1, const double* dist1, const float* uv1, const float* uv2,
                double* res, double* J)
{
  double R[9], Jl[9];
  rodrigues(cam15 + 4, R);
  so3_left_jacobian(cam15 + 4, Jl);
  double u1 = uv1[0], v1 = uv1[1];
  bool skip = false;
  if (ktype & 1) {
    float ou, ov;
    undistort_point(k1[0], k1[1], k1[2], k1[3], dist1, uv1[0], uv1[1], ou, ov);
    skip = (ou < 0 || ou >= k1[2] * 2 || ov < 0 || ov >= k1[3] * 2);
    u1 = ou; v1 = ov;
  }
  double X[3] = {(u1 - k1[2]) / k1[0], (v1 - k1[3]) / k1[1], 1.0};
  double n = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
  double ray1[3] = {X[0] / n, X[1] / n, X[2] / n};
  const double fy = (ktype & 2) ? cam15[1] : cam15[0];
#define H_KRT(T)                                                                                                       \
  {                                                                                                                    \
    double j[2][KrtDims<T>::NF];                                                                                       \
    krt_eval<T, true>(R, Jl, cam15[0], fy, cam15[2], cam15[3], cam15 + 10, ray1, skip, uv2[0], uv2[1], res, j);      \
    for (int i = 0; i < 2 * KrtDims<T>::NF; ++i) J[i] = (&j[0][0])[i];                                               \
  }
  switch (ktype) {
    case 0: H_KRT(0) break;
    case 1: H_KRT(1) break;
    case 2: H_KRT(2) break;
    default: H_KRT(3) break;
  }
#undef H_KRT
}
// KRT 2D-3D block: cam15 current (local frame), Xl local point
void h_krt_eval_2d3d(int ktype, const double* cam15, const double* Xl, const float* uv, double* res, double* J)
{
  double R[9], Jl[9];
  rodrigues(cam15 + 4, R);
  so3_left_jacobian(cam15 + 4, Jl);
  const double fy = (ktype & 2) ? cam15[1] : cam15[0];
#define H_KRT3(T)                                                                                                      \
  {                                                                                                                    \
    double j[2][KrtDims<T>::NF];                                                                                       \
    krt_eval_2d3d<T, true>(R, Jl, cam15[0], fy, cam15[2], cam15[3], cam15 + 10, cam15 + 7, Xl, uv[0], uv[1], res, j);  \
    for (int i = 0; i < 2 * KrtDims<T>::NF; ++i) J[i] = (&j[0][0])[i];                                               \
  }
  switch (ktype) {
    case 0: H_KRT3(0) break;
    case 1: H_KRT3(1) break;
    case 2: H_KRT3(2) break;
    default: H_KRT3(3) break;
  }
#undef H_KRT3
}
// F3: cam15, tlw6, xyz, uv -> res[2], Jc[2][5+factor], Jt[2][6]
void h_reproj2d3d(int factor, const double* cam15, const double* tlw, const double* xyz, const float* uv, double* res, double* Jc, double* Jt)
{
  double cb[CAMBLK], tl[TLWBLK];
  fill_camblk(cam15, cb);
  rodrigues(tlw, tl);
  so3_left_jacobian(tlw, tl + 9);
  tl[18] = tlw[3]; tl[19] = tlw[4]; tl[20] = tlw[5];
  double jt[2][6];
  if (factor == 0) {
    double jc[2][5];
    reproj2d3d_eval<0, true>(cb, tl, xyz, uv[0], uv[1], res, jc, jt);
    for (int i = 0; i < 10; ++i) Jc[i] = (&jc[0][0])[i];
  }
  else {
    double jc[2][6];
    reproj2d3d_eval<1, true>(cb, tl, xyz, uv[0], uv[1], res, jc, jt);
    for (int i = 0; i < 12; ++i) Jc[i] = (&jc[0][0])[i];
  }
  for (int i = 0; i < 12; ++i) Jt[i] = (&jt[0][0])[i];
}
int h_ba_pair_side(int type, const double* cam15, const double* disp, const double* ray, double* Jc, double* Jr)
{
  double cb[CAMBLK_DISP];
  fill_camblk(cam15, cb);
  for (int k = 0; k < 3; ++k) cb[CB_D + k] = disp ? disp[k] : 0.0;
  if (type == 0) return pair_side_t<0>(cb, ray, Jc, Jr);
  if (type == 1) return pair_side_t<1>(cb, ray, Jc, Jr);
  if (type == 2) return pair_side_t<2>(cb, ray, Jc, Jr);
  return pair_side_t<3>(cb, ray, Jc, Jr);
}
// PTZRayDistDisp: disp = (d0, d1, d2); Jc: [2][8] columns [f, k1, r1, r2, r3, d0, d1, d2]
void h_ba_linearize_disp(const double* cam15, const double* disp, const double* ray, const float* uv, double* res, double* Jc, double* Jr)
{
  double cb[CAMBLK_DISP];
  fill_camblk(cam15, cb);
  for (int k = 0; k < 3; ++k) cb[CB_D + k] = disp[k];
  double jc[2][8], jr[2][3];
  ba_linearize<3>(cb, ray, uv[0], uv[1], res, jc, jr);
  for (int i = 0; i < 16; ++i) Jc[i] = (&jc[0][0])[i];
  for (int i = 0; i < 6; ++i) Jr[i] = (&jr[0][0])[i];
}
void h_ba_residual_disp(const double* cam15, const double* disp, const double* ray, const float* uv, double* res)
{
  double cb[CAMBLK_DISP];
  fill_camblk(cam15, cb);
  for (int k = 0; k < 3; ++k) cb[CB_D + k] = disp[k];
  ba_residual<3>(cb, ray, uv[0], uv[1], res);
}
// v: step in the column order of h_ba_linearize_disp
void h_ba_step_dir_disp(const double* cam15, const double* disp, const double* ray, const float* uv, const double* v, double* res, double* p, double* Jr)
{
  double cb[CAMBLK_DISP];
  fill_camblk(cam15, cb);
  for (int k = 0; k < 3; ++k) cb[CB_D + k] = disp[k];
  const double* Jl = cb + CB_JL;
  double om[3];
  for (int r = 0; r < 3; ++r) om[r] = Jl[3 * r] * v[2] + Jl[3 * r + 1] * v[3] + Jl[3 * r + 2] * v[4];
  const double sv[5] = {v[0], v[1], v[5], v[6], v[7]};  // the evaluation kernel's layout: [non-rotation columns | om]
  double jr[2][3];
  ba_step_dir<3>(cb, ray, uv[0], uv[1], sv, om, res, p, jr);
  for (int i = 0; i < 6; ++i) Jr[i] = (&jr[0][0])[i];
}
// Reproj2d3dDispFactor: Jc [2][9] columns [fx, fy, k1, r1, r2, r3, d0, d1, d2]
void h_reproj2d3d_disp(const double* cam15, const double* disp, const double* tlw, const double* xyz, const float* uv, double* res, double* Jc, double* Jt)
{
  double cb[CAMBLK_DISP], tl[TLWBLK];
  fill_camblk(cam15, cb);
  for (int k = 0; k < 3; ++k) cb[CB_D + k] = disp[k];
  rodrigues(tlw, tl);
  so3_left_jacobian(tlw, tl + 9);
  tl[18] = tlw[3]; tl[19] = tlw[4]; tl[20] = tlw[5];
  double jt[2][6], jc[2][9];
  reproj2d3d_eval<1, true, true>(cb, tl, xyz, uv[0], uv[1], res, jc, jt);
  for (int i = 0; i < 18; ++i) Jc[i] = (&jc[0][0])[i];
  for (int i = 0; i < 12; ++i) Jt[i] = (&jt[0][0])[i];
}
int h_inv3(const double* A6, double* Ai6) { return inv3_spd(A6, Ai6) ? 1 : 0; }
}
