#include "krt_optimizer.h"

#include <cmath>

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
  for (const auto& m : matches) {
    uv_ref_.push_back(kpts_ref[m.queryIdx].pt.x);
    uv_ref_.push_back(kpts_ref[m.queryIdx].pt.y);
    uv_cur_.push_back(kpts_curr[m.trainIdx].pt.x);
    uv_cur_.push_back(kpts_curr[m.trainIdx].pt.y);
  }
}

void KRTOptimizer::Add2d3dConstraints(const std::vector<Point2f>& pts2d, const std::vector<Point3d>& pts3d)
{
  if (pts2d.size() != pts3d.size() || pts2d.empty()) return;
  has_2d3d_ = true;  // never called by the reference's applications; not implemented on the device path
}

bool KRTOptimizer::Solve(Mat33& K, Mat33& R, Vec3& t, Vec5& dist)
{
  if (has_2d3d_ || uv_ref_.empty()) return false;
  const int64_t match_ptr[2] = {0, static_cast<int64_t>(uv_ref_.size() / 2)};
  std::vector<double> ref = cam_ref_.ToVector(), cur = cam_curr_world_.ToVector();
  ptz_lm_options opt;
  ptz_lm_options_default(&opt);
  opt.max_num_iterations = max_iter_;  // krt_optimizer.cc:388
  opt.device_id = device_id_;
  int32_t accepted = 0;
  if (ptz_krt_solve_batch(1, match_ptr, uv_ref_.data(), uv_cur_.data(), ref.data(), cur.data(),
                          static_cast<int32_t>(factor_type_), max_reproj_error_, &opt, &summary_, &accepted,
                          nullptr) != PTZ_OK)
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
{  // krt_optimizer.cc:406-455 for the F factor (rotation of the current camera relative to the reference)
  const Mat33 Rl = Mul(cam_curr_world_.R(), Inverse(cam_ref.R()));
  const Mat33 Kr = cam_ref.K(), Kc = cam_curr_world_.K();
  double s0 = 0, s1 = 0;
  for (const auto& m : matches) {
    const Point2f a = kpts_ref[m.queryIdx].pt, b = kpts_curr[m.trainIdx].pt;
    Vec3 x = {(a.x - Kr[2]) / Kr[0], (a.y - Kr[5]) / Kr[4], 1.0};
    const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + 1.0);
    x = {x[0] / n, x[1] / n, x[2] / n};
    const Vec3 P = Mul(Rl, x);
    const double r0 = b.x - (Kc[0] * P[0] + Kc[2] * P[2]) / P[2], r1 = b.y - (Kc[0] * P[1] + Kc[5] * P[2]) / P[2];
    s0 += r0 * r0;
    s1 += r1 * r1;
  }
  return std::sqrt((s0 + s1) / static_cast<double>(matches.size()));
}

}  // namespace ptzcalib
