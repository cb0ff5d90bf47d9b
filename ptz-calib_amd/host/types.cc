#include "types.h"

#include "small_linalg.h"

#include <cfloat>
#include <cmath>

namespace ptzcalib {

Mat33 Mul(const Mat33& a, const Mat33& b)
{
  Mat33 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  return c;
}
Vec3 Mul(const Mat33& a, const Vec3& v)
{
  return {a[0] * v[0] + a[1] * v[1] + a[2] * v[2], a[3] * v[0] + a[4] * v[1] + a[5] * v[2], a[6] * v[0] + a[7] * v[1] + a[8] * v[2]};
}
Mat33 Transpose(const Mat33& a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
Mat33 Inverse(const Mat33& S)
{
  const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  const double d = 1.0 / det;
  return {(S[4] * S[8] - S[5] * S[7]) * d, (S[2] * S[7] - S[1] * S[8]) * d, (S[1] * S[5] - S[2] * S[4]) * d,
          (S[5] * S[6] - S[3] * S[8]) * d, (S[0] * S[8] - S[2] * S[6]) * d, (S[2] * S[3] - S[0] * S[5]) * d,
          (S[3] * S[7] - S[4] * S[6]) * d, (S[1] * S[6] - S[0] * S[7]) * d, (S[0] * S[4] - S[1] * S[3]) * d};
}

Mat33 Rodrigues(const Vec3& r)
{
  const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) return Eye3();
  const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
  const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
  return {c + c1 * x * x,     c1 * x * y - s * z, c1 * x * z + s * y,
          c1 * x * y + s * z, c + c1 * y * y,     c1 * y * z - s * x,
          c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z};
}

Vec3 RodriguesInv(const Mat33& R_in)
{
  // cv::Rodrigues (matrix -> vector) first replaces R by U V^T of its SVD, which is what turns the scaled
  // near-rotation K_j^-1 H K_i of the registration step (ptz_incremental_optimizer.cc:391-394) into a rotation.
  Mat33 R = R_in;
  {
    Mat33 Um, Vm;
    double s[3];
    JacobiSVD3(R_in.data(), Um.data(), s, Vm.data());  // (the 3 x 3 case of JacobiSVD without heap: same rotations, same bits)
    if (s[2] > 0) R = Mul(Um, Transpose(Vm));
  }
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) return {0, 0, 0};
    double t;
    t = (R[0] + 1) * 0.5; rx = std::sqrt(t > 0 ? t : 0.);
    t = (R[4] + 1) * 0.5; ry = std::sqrt(t > 0 ? t : 0.) * (R[1] < 0 ? -1. : 1.);
    t = (R[8] + 1) * 0.5; rz = std::sqrt(t > 0 ? t : 0.) * (R[2] < 0 ? -1. : 1.);
    if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
    theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
    return {rx * theta, ry * theta, rz * theta};
  }
  const double vth = theta / (2 * s);
  return {rx * vth, ry * vth, rz * vth};
}

Vec3 Camera::t_wc() const
{
  const Vec3 v = Mul(Inverse(R_), t_);
  return {-v[0], -v[1], -v[2]};
}

std::vector<double> Camera::ToVector() const
{
  std::vector<double> v(15);
  v[0] = K_[0]; v[1] = K_[4]; v[2] = K_[2]; v[3] = K_[5];
  const Vec3 rv = RodriguesInv(R_);
  v[4] = rv[0]; v[5] = rv[1]; v[6] = rv[2];
  v[7] = t_[0]; v[8] = t_[1]; v[9] = t_[2];
  for (int k = 0; k < 5; ++k) v[10 + k] = dist_[k];
  return v;
}

void Camera::FromVector(const std::vector<double>& v)
{
  if (v.size() != 15) throw std::invalid_argument("Expected camera vector size: 15, actual size :" + std::to_string(v.size()));
  K_ = {v[0], 0, v[2], 0, v[1], v[3], 0, 0, 1};
  R_ = Rodrigues({v[4], v[5], v[6]});
  t_ = {v[7], v[8], v[9]};
  for (int k = 0; k < 5; ++k) dist_[k] = v[10 + k];
}

}  // namespace ptzcalib
