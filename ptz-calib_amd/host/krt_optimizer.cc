#include "krt_optimizer.h"

#include <cmath>

#include "../csrc/ptz_factor.h"
#include <stdexcept>

namespace ptzcalib {

KRTOptimizer::KRTOptimizer(int max_iter, double max_reproj_error, FACTOR_TYPE factor_type)
    : factor_type_(factor_type), max_iter_(max_iter), max_reproj_error_(max_reproj_error)
{
}

void KRTOptimizer::SetInitParams(const Mat33& K, const Mat33& R, const Vec3& t, const Vec5& dist)
{
  cam_curr_world_ = Camera(K, R, t, dist);
}

void KRTOptimizer::Add2d2dConstraints(const Camera& cam_ref, const std::vector<KeyPoint>& kpts_ref,
                                      const std::vector<KeyPoint>& kpts_curr, const std::vector<DMatch>& matches)
{  // krt_optimizer.cc:265-316: one residual block per match (uv1 = reference keypoint, uv2 = current keypoint)
  cam_ref_ = cam_ref;
  has_2d2d_ = true;
  for (const auto& m : matches) {
    uv_ref_.push_back(kpts_ref[m.queryIdx].pt.x);
    uv_ref_.push_back(kpts_ref[m.queryIdx].pt.y);
    uv_cur_.push_back(kpts_curr[m.trainIdx].pt.x);
    uv_cur_.push_back(kpts_curr[m.trainIdx].pt.y);
  }
}

void KRTOptimizer::Add2d3dConstraints(const std::vector<Point2f>& pts2d, const std::vector<Point3d>& pts3d)
{  // krt_optimizer.cc:350-383: one Factor2d3d(Fxfy)Dist block per point; the move into the local frame (:357-362) is done
   // by the device code from the reference camera given to Add2d2dConstraints, which therefore has to come first (the
   // reference multiplies by an empty R_local_world_ otherwise and OpenCV throws)
  if (pts2d.size() != pts3d.size() || pts2d.empty()) return;
  if (!has_2d2d_) throw std::logic_error("KRTOptimizer::Add2d3dConstraints before Add2d2dConstraints");
  for (size_t i = 0; i < pts2d.size(); ++i) {
    pts2d_.push_back(pts2d[i].x);
    pts2d_.push_back(pts2d[i].y);
    pts3d_.push_back(pts3d[i].x);
    pts3d_.push_back(pts3d[i].y);
    pts3d_.push_back(pts3d[i].z);
  }
}

bool KRTOptimizer::Solve(Mat33& K, Mat33& R, Vec3& t, Vec5& dist)
{
  if (uv_ref_.empty()) return false;
  const int64_t match_ptr[2] = {0, static_cast<int64_t>(uv_ref_.size() / 2)};
  const int64_t point_ptr[2] = {0, static_cast<int64_t>(pts2d_.size() / 2)};
  const bool p3 = !pts2d_.empty();
  std::vector<double> ref = cam_ref_.ToVector(), cur = cam_curr_world_.ToVector();
  ptz_lm_options opt;
  ptz_lm_options_default(&opt);
  opt.max_num_iterations = max_iter_;  // krt_optimizer.cc:388
  opt.device_id = device_id_;
  int32_t accepted = 0;
  if (ptz_krt_solve_batch_2d3d(1, match_ptr, uv_ref_.data(), uv_cur_.data(), p3 ? point_ptr : nullptr, pts2d_.data(),
                               pts3d_.data(), ref.data(), cur.data(), static_cast<int32_t>(factor_type_), max_reproj_error_,
                               &opt, &summary_, &accepted, nullptr) != PTZ_OK)
    return false;
  num_iter_ = summary_.num_successful_steps;  // krt_optimizer.cc:396
  if (!accepted) return false;                // CheckResults, :504-533
  cam_curr_world_.FromVector(cur);            // ObtainRefinedCameraParams, :535-567 (done on the device, world frame)
  K = cam_curr_world_.K();
  dist = cam_curr_world_.dist();
  R = cam_curr_world_.R();
  t = cam_curr_world_.t();
  return true;
}

double KRTOptimizer::Cal2d2dReprojError(const Camera& cam_ref, const std::vector<KeyPoint>& kpts_ref,
                                        const std::vector<KeyPoint>& kpts_curr, const std::vector<DMatch>& matches)
{  // krt_optimizer.cc:406-455: the factor of the configured type at the current camera, in the local frame of cam_ref
   // (rotation of the current camera relative to the reference); the same functor arithmetic as the device code
  const Mat33 Rl = Mul(cam_curr_world_.R(), Inverse(cam_ref.R()));
  const Mat33 Kr = cam_ref.K(), Kc = cam_curr_world_.K();
  const Vec5 dr = cam_ref.dist(), dc = cam_curr_world_.dist();
  const bool with_dist = factor_type_ == FDist || factor_type_ == FxfyDist;
  const double fx = Kc[0], fy = (factor_type_ == Fxfy || factor_type_ == FxfyDist) ? Kc[4] : Kc[0];
  double s0 = 0, s1 = 0;
  for (const auto& m : matches) {
    const Point2f a = kpts_ref[m.queryIdx].pt, b = kpts_curr[m.trainIdx].pt;
    double u = a.x, v = a.y, res[2], unused[2][4];
    bool skip = false;
    if (with_dist) {  // :89-101
      float ou, ov;
      ptz::undistort_point(Kr[0], Kr[4], Kr[2], Kr[5], dr.data(), a.x, a.y, ou, ov);
      skip = ou < 0 || ou >= Kr[2] * 2 || ov < 0 || ov >= Kr[5] * 2;
      u = ou; v = ov;
    }
    Vec3 x = {(u - Kr[2]) / Kr[0], (v - Kr[5]) / Kr[4], 1.0};
    const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + 1.0);
    x = {x[0] / n, x[1] / n, x[2] / n};
    if (with_dist) ptz::krt_eval<1, false>(Rl.data(), nullptr, fx, fy, Kc[2], Kc[5], dc.data(), x.data(), skip, b.x, b.y, res, nullptr);
    else ptz::krt_eval<0, false>(Rl.data(), nullptr, fx, fy, Kc[2], Kc[5], dc.data(), x.data(), skip, b.x, b.y, res, unused);
    s0 += res[0] * res[0];
    s1 += res[1] * res[1];
  }
  return std::sqrt((s0 + s1) / static_cast<double>(matches.size()));
}

double KRTOptimizer::Cal2d3dReprojError(const std::vector<Point2f>& pts2d, const std::vector<Point3d>& pts3d)
{  // krt_optimizer.cc:457-500: Factor2d3d(Fxfy)Dist at the current camera; R_l X_l + t_l with X_l = R_ref X_w + t_ref equals
   // R_cur X_w + t_cur, so the world-frame camera is used directly.  cv::projectPoints reads dist as (k1,k2,p1,p2,k3).
  if (pts2d.size() != pts3d.size() || pts2d.empty()) return -1;
  const Mat33 R = cam_curr_world_.R(), K = cam_curr_world_.K();
  const Vec3 t = cam_curr_world_.t();
  const Vec5 d = cam_curr_world_.dist();
  const double fx = K[0], fy = (factor_type_ == Fxfy || factor_type_ == FxfyDist) ? K[4] : K[0];
  double s0 = 0, s1 = 0;
  for (size_t i = 0; i < pts2d.size(); ++i) {
    const Vec3 P = Mul(R, Vec3{pts3d[i].x, pts3d[i].y, pts3d[i].z});
    double z = P[2] + t[2];
    z = z != 0.0 ? 1.0 / z : 1.0;
    const double x = (P[0] + t[0]) * z, y = (P[1] + t[1]) * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double cd = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6;
    const double xd = x * cd + d[2] * (2 * x * y) + d[3] * (r2 + 2 * x * x);
    const double yd = y * cd + d[2] * (r2 + 2 * y * y) + d[3] * (2 * x * y);
    const double r0 = pts2d[i].x - (xd * fx + K[2]), r1 = pts2d[i].y - (yd * fy + K[5]);
    s0 += r0 * r0;
    s1 += r1 * r1;
  }
  return std::sqrt((s0 + s1) / static_cast<double>(pts2d.size()));
}

}  // namespace ptzcalib
