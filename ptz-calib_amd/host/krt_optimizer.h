// krt_optimizer.h -- KRTOptimizer with the reference's public interface (src/core/krt_optimizer.h:108-145).
// Solve() hands the single-view problem to the MI355X library (ptz_krt_solve_batch with one query); the batched
// entry point is what relocalization over many queries should call directly.
#pragma once

#include <vector>

#include "../../include/ptz_calib_amd.h"
#include "types.h"

namespace ptzcalib {

class KRTOptimizer {
 public:
  enum FACTOR_TYPE { F, FDist, Fxfy, FxfyDist };
  KRTOptimizer(int max_iter, double max_reproj_error, FACTOR_TYPE factor_type);
  void SetInitParams(const Mat33& K, const Mat33& R, const Vec3& t, const Vec5& dist);
  void Add2d2dConstraints(const Camera& cam_ref, const std::vector<KeyPoint>& kpts_ref, const std::vector<KeyPoint>& kpts_curr,
                          const std::vector<DMatch>& matches);
  void Add2d3dConstraints(const std::vector<Point2f>& pts2d, const std::vector<Point3d>& pts3d);
  bool Solve(Mat33& K, Mat33& R, Vec3& t, Vec5& dist);
  double Cal2d2dReprojError(const Camera& cam_ref, const std::vector<KeyPoint>& kpts_ref, const std::vector<KeyPoint>& kpts_curr,
                            const std::vector<DMatch>& matches);
  double Cal2d3dReprojError(const std::vector<Point2f>& pts2d, const std::vector<Point3d>& pts3d);
  void SetFixedFocal() { set_fixed_focal_ = true; }  // a flag nobody reads, as in the reference (krt_optimizer.cc:502)
  int num_iter_ = 0;

  const ptz_lm_summary& summary() const { return summary_; }
  void SetDevice(int device_id) { device_id_ = device_id; }

 private:
  Camera cam_curr_world_, cam_ref_;
  std::vector<float> uv_ref_, uv_cur_, pts2d_;
  std::vector<double> pts3d_;
  bool has_2d2d_ = false, set_fixed_focal_ = false;
  FACTOR_TYPE factor_type_;
  int max_iter_;
  double max_reproj_error_;
  int device_id_ = 0;
  ptz_lm_summary summary_{};
};

}  // namespace ptzcalib
