// ptzray_optimizer.h -- PTZRayOptimizer with the reference's public interface
// (src/core/ptzray_optimizer.h:110-129): same constructors, Solve overloads, error accessors, FACTOR_TYPE.
// Where the reference owns a ceres::Problem and calls ceres::Solve (ptzray_optimizer.cc:469-475), this class
// packs the problem into flat arrays and calls the MI355X library through its C-ABI (ptz_ba_solve).
#pragma once

#include <memory>
#include <unordered_set>
#include <vector>

#include "../../include/ptz_calib_amd.h"
#include "tracks.h"
#include "types.h"

namespace ptzcalib {

enum FACTOR_TYPE { PTZRay, PTZRayDist, PTZRayFxfyDist, PTZRayDistDisp };

// Packed form of one global-BA problem (what crosses the C-ABI).
struct PackedBA {
  std::vector<long> cam_image;     // compact camera id -> image id (ascending)
  std::vector<int> ray_track;      // compact ray id -> track id (ascending)
  std::vector<float> obs_uv;       // [2*n_obs]
  std::vector<int32_t> obs_cam, obs_ray;
  std::vector<double> ray_weight;  // full track length (ptzray_optimizer.cc:805)
  std::vector<double> cam;         // [15*n_cam]
  std::vector<double> ray;         // [3*n_ray] Pix2Ray initialisation
  std::vector<float> obs3d_uv;     // [2*n_obs3d] annotation pixels, cameras ascending (ptzray_optimizer.cc:894-917)
  std::vector<double> obs3d_xyz;   // [3*n_obs3d]
  std::vector<int32_t> obs3d_cam;  // compact camera id
  std::vector<int32_t> ic_of_cam;  // intrinsics block id per compact camera (SetSharedIntrinsics)
  std::array<double, 6> tlw{{0, 0, 0, 0, 0, 0}};  // T_l_w: initial value before Solve, refined value after
  bool tlw_init_ok = false;        // SetInitTransLocalToWorld() found a view that passed the PnP gates
};

// Tracks of one match table, built once and shared by many solves: the reference's map plus the same content as flat
// arrays in the map's iteration order.
struct SharedTracks {
  std::vector<int> id;        // track ids, ascending
  std::vector<int64_t> ptr;   // [n_tracks + 1]
  std::vector<int> img, feat; // views of track k: ptr[k] .. ptr[k+1], image ids ascending
};

class PTZRayOptimizer {
 public:
  PTZRayOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                  const std::vector<Camera>& cameras, const std::vector<std::vector<Point2f>>& pixels,
                  const std::vector<std::vector<Point3d>>& pts3d, const std::unordered_set<long>& cam_ids, int max_iter,
                  FACTOR_TYPE type);
  PTZRayOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                  const std::vector<Camera>& cameras, const std::unordered_set<long>& cam_ids, int max_iter, FACTOR_TYPE type);
  bool Solve(std::vector<Camera>& cameras);
  bool Solve(std::vector<Camera>& cameras, std::vector<std::vector<Ray>>& rays);
  double final_reproj_error_all() const { return final_reproj_error_all_; }
  double final_reproj_error_2d2d() const { ComputeErrors(); return final_reproj_error_2d2d_; }
  double final_reproj_error_2d3d() const { ComputeErrors(); return final_reproj_error_2d3d_; }
  void SetSharedIntrinsics(const std::vector<long>& shared_ic_ids);
  static void T_l_w(const double* tlw, Mat33& R_l_w, Vec3& t_l_w);

  // extras (not in the reference): the packed problem and the solver summary of the last Solve
  const PackedBA& packed() const { return packed_; }
  const ptz_lm_summary& summary() const { return summary_; }
  const std::array<double, 6>& initial_tlw() const { return tlw_init_; }
  const std::array<double, 3>& displacement() const { return disp_; }
  void SetDevice(int device_id) { device_id_ = device_id; }
  double device_ms() const { return device_ms_; }  // wall time of the ptz_ba_solve call of the last Solve
  // The tracks depend on the match table only, not on the candidate set: a caller that solves many candidate subsets of
  // one match table (PtzIncrementalOptimizer) builds them once and shares them instead of repeating FindTracks().
  void UseTracks(std::shared_ptr<const SharedTracks> tracks) { shared_tracks_ = std::move(tracks); }
  static std::shared_ptr<const SharedTracks> BuildTracks(const std::vector<MatchesInfo>& matches_info);
  // The same tracks resident on the device (ptz_rig_create): a solve over a candidate subset is then a VIEW of them -- the
  // packed problem is built on the device, the host neither walks the tracks nor uploads observations (2D-2D residuals without
  // shared intrinsics; anything else packs on the host as before).  Same arrays, same initial rays, same bits.
  void UseRig(const ptz_rig* rig) { rig_ = rig; }
  // Same as the 2D-2D constructor but WITHOUT the deep copies of features / matches (ptzray_optimizer.h:145-149): the
  // caller keeps both alive until Solve returns.
  struct Borrow {};
  PTZRayOptimizer(Borrow, const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                  const std::vector<Camera>& cameras, const std::unordered_set<long>& cam_ids, int max_iter, FACTOR_TYPE type);

 private:
  bool CheckValid() const;
  bool SolveImpl(std::vector<Camera>& cameras, std::vector<std::vector<Ray>>* rays);
  bool SolveView(std::vector<Camera>& cameras);
  void FindTracks();
  bool isCandidate(long image_id) const { return cam_ids_.count(image_id) != 0; }
  void Pack();
  const Tracks& tracks() const { return tracks_; }  // own tracks (FindTracks); empty when UseTracks() supplied the shared flat form
  bool SetInitTransLocalToWorld();

  std::vector<Camera> cameras_;
  std::vector<ImageFeatures> features_own_;
  std::vector<MatchesInfo> matches_info_own_;
  const std::vector<ImageFeatures>& features_;
  const std::vector<MatchesInfo>& matches_info_;
  std::shared_ptr<const SharedTracks> shared_tracks_;
  const ptz_rig* rig_ = nullptr;
  std::vector<std::vector<Point2f>> pixels_;
  std::vector<std::vector<Point3d>> pts3d_;
  size_t num_cams_ = 0;
  std::unordered_set<long> cam_ids_;
  std::vector<long> shared_ic_ids_;
  FACTOR_TYPE type_;
  Tracks tracks_;
  int track_len_ = 0, max_track_len_ = 0, min_track_len_ = 0;
  int max_iter_ = 100;
  int device_id_ = 0;
  double device_ms_ = 0;
  PackedBA packed_;
  std::array<double, 6> tlw_init_{{0, 0, 0, 0, 0, 0}};
  std::array<double, 3> disp_{{0, 0, 0}};  // disp_param_ (ptzray_optimizer.h: PTZRayDistDisp), refined by Solve
  ptz_lm_summary summary_{};
  double init_reproj_error_all_ = 0, final_reproj_error_all_ = 0;
  void ComputeErrors() const;  // unweighted 2D-2D / 2D-3D RMS of the solved state, on first use
  mutable bool errors_ready_ = true;
  mutable double final_reproj_error_2d2d_ = 0, final_reproj_error_2d3d_ = 0;
};

}  // namespace ptzcalib
