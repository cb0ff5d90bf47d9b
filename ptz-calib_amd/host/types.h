// types.h -- data types that cross the optimizer boundary, mirroring src/core/types.h of the reference
// (ImageFeatures :17-22, MatchesInfo :24-32, Ray :34-45, Camera :47-72) without OpenCV: fixed-size
// row-major matrices replace cv::Mat (CV_64F), plain structs replace cv::Point2f / KeyPoint / DMatch / Size.
#pragma once

#include <array>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptzcalib {

struct Point2f { float x = 0, y = 0; Point2f() = default; Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3d { double x = 0, y = 0, z = 0; Point3d() = default; Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {} };
struct Size { int width = 0, height = 0; };
struct KeyPoint { Point2f pt; };                   // only .pt is read by the optimizers
struct DMatch { int queryIdx = 0, trainIdx = 0; };  // (feature of src image, feature of dst image)

using Mat33 = std::array<double, 9>;  // row-major
using Vec3 = std::array<double, 3>;
using Vec5 = std::array<double, 5>;

inline Mat33 Eye3() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
Mat33 Mul(const Mat33& a, const Mat33& b);
Vec3 Mul(const Mat33& a, const Vec3& v);
Mat33 Transpose(const Mat33& a);
Mat33 Inverse(const Mat33& a);                 // closed-form 3x3 inverse (cv::Mat::inv() on 3x3)
Mat33 Rodrigues(const Vec3& rvec);             // cv::Rodrigues vector -> matrix
Vec3 RodriguesInv(const Mat33& R);             // cv::Rodrigues matrix -> vector

struct ImageFeatures {
  long img_idx = 0;
  Size img_size;
  std::vector<KeyPoint> keypoints;
};

struct MatchesInfo {
  long src_img_idx = 0, dst_img_idx = 0;
  std::vector<DMatch> matches;
  std::vector<unsigned char> inliers_mask;
  int num_inliers = 0;
  Mat33 H = Eye3();
  bool H_empty = true;     // cv::Mat::empty() of the reference's H
  double confidence = 0;
};

struct Ray {
  int id_ = 0;
  Point3d pt3d_;
  Point2f uv_;
  Ray(int id, const Vec3& pt3d, const Point2f& uv) : id_(id), pt3d_(pt3d[0], pt3d[1], pt3d[2]), uv_(uv) {}
};

// Camera: K (3x3), R (3x3), t (3x1), dist (5x1: k1,k2,k3,p1,p2), types.h:47-72
class Camera {
 public:
  Camera() : K_(Eye3()), R_(Eye3()), t_{0, 0, 0}, dist_{0, 0, 0, 0, 0} {}
  Camera(const Mat33& K, const Mat33& R, const Vec3& t, const Vec5& dist) : K_(K), R_(R), t_(t), dist_(dist) {}
  const Mat33& K() const { return K_; }
  Mat33& K() { return K_; }
  const Mat33& R() const { return R_; }
  Mat33& R() { return R_; }
  Vec3 rvec() const { return RodriguesInv(R_); }
  const Vec3& t() const { return t_; }
  Vec3& t() { return t_; }
  Vec3 t_wc() const;  // -R^-1 t
  const Vec5& dist() const { return dist_; }
  Vec5& dist() { return dist_; }
  // [fx, fy, cx, cy, r1, r2, r3, t1, t2, t3, k1, k2, k3, p1, p2]  (types.cc:32-57)
  std::vector<double> ToVector() const;
  void FromVector(const std::vector<double>& v);  // throws std::invalid_argument unless v.size() == 15 (types.cc:61-63)

 private:
  Mat33 K_, R_;
  Vec3 t_;
  Vec5 dist_;
};

}  // namespace ptzcalib
