#include "ptzray_optimizer.h"

#include "device_batcher.h"

#include <cstdio>

#include <chrono>
#include <cmath>
#include <limits>
#include <numeric>

#include "epnp.h"

namespace ptzcalib {

PTZRayOptimizer::PTZRayOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                                 const std::vector<Camera>& cameras, const std::vector<std::vector<Point2f>>& pixels,
                                 const std::vector<std::vector<Point3d>>& pts3d, const std::unordered_set<long>& cam_ids,
                                 int max_iter, FACTOR_TYPE type)
    : cameras_(cameras), features_own_(features), matches_info_own_(matches_info), features_(features_own_),
      matches_info_(matches_info_own_), pixels_(pixels), pts3d_(pts3d), num_cams_(cameras.size()), type_(type), max_iter_(max_iter)
{
  // empty cam_ids => every camera is a candidate (ptzray_optimizer.cc:418-425)
  if (cam_ids.empty()) for (size_t i = 0; i < cameras_.size(); ++i) cam_ids_.insert(static_cast<long>(i));
  else cam_ids_ = cam_ids;
  shared_ic_ids_.resize(cameras_.size());
  std::iota(shared_ic_ids_.begin(), shared_ic_ids_.end(), 0);  // intrinsics are NOT shared by default (:427-428)
}

PTZRayOptimizer::PTZRayOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                                 const std::vector<Camera>& cameras, const std::unordered_set<long>& cam_ids, int max_iter,
                                 FACTOR_TYPE type)
    : PTZRayOptimizer(features, matches_info, cameras, {}, {}, cam_ids, max_iter, type)
{
}

PTZRayOptimizer::PTZRayOptimizer(Borrow, const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                                 const std::vector<Camera>& cameras, const std::unordered_set<long>& cam_ids, int max_iter,
                                 FACTOR_TYPE type)
    : cameras_(cameras), features_(features), matches_info_(matches_info), num_cams_(cameras.size()), type_(type), max_iter_(max_iter)
{
  if (cam_ids.empty()) for (size_t i = 0; i < cameras_.size(); ++i) cam_ids_.insert(static_cast<long>(i));
  else cam_ids_ = cam_ids;
  shared_ic_ids_.resize(cameras_.size());
  std::iota(shared_ic_ids_.begin(), shared_ic_ids_.end(), 0);
}

std::shared_ptr<const SharedTracks> PTZRayOptimizer::BuildTracks(const std::vector<MatchesInfo>& matches_info)
{
  auto st = std::make_shared<SharedTracks>();
  TracksBuilder builder;
  builder.Build(matches_info);
  builder.Filter(4);
  // flat arrays in the iteration order of the reference's map of maps (track id ascending, image id ascending); the maps
  // themselves (a heap node per view) are not built for the shared form
  builder.ExportFlat(st->id, st->ptr, st->img, st->feat);
  return st;
}

void PTZRayOptimizer::SetSharedIntrinsics(const std::vector<long>& shared_ic_ids)
{
  if (shared_ic_ids.size() != cameras_.size()) return;  // length mismatch: ignored with a warning in the reference (:499-502)
  shared_ic_ids_ = shared_ic_ids;
}

void PTZRayOptimizer::T_l_w(const double* tlw, Mat33& R_l_w, Vec3& t_l_w)
{
  R_l_w = Rodrigues({tlw[0], tlw[1], tlw[2]});
  t_l_w = {tlw[3], tlw[4], tlw[5]};
}

bool PTZRayOptimizer::CheckValid() const
{  // ptzray_optimizer.cc:515-535
  if (num_cams_ == 0) return false;
  if (features_.size() != num_cams_) return false;
  if (max_iter_ <= 0) return false;
  if (!pixels_.empty()) {
    if (pixels_.size() != num_cams_ || pts3d_.size() != num_cams_) return false;
    for (size_t i = 0; i < num_cams_; ++i)
      if (pixels_[i].size() != pts3d_[i].size()) return false;
  }
  return true;
}

void PTZRayOptimizer::FindTracks()
{  // ptzray_optimizer.cc:537-552
  if (shared_tracks_) return;  // built once by the caller (the statistics of Length() are only logged by the reference)
  TracksBuilder builder;
  builder.Build(matches_info_);
  builder.Filter(4);
  builder.ExportToSTL(tracks_);
  Length(tracks_, track_len_, max_track_len_, min_track_len_);
}

// T_l_w from the first candidate view whose annotations pass the PnP gates (ptzray_optimizer.cc:562-633).
bool PTZRayOptimizer::SetInitTransLocalToWorld()
{
  packed_.tlw = {{0, 0, 0, 0, 0, 0}};
  packed_.tlw_init_ok = false;
  for (size_t i = 0; i < num_cams_; ++i) {
    if (!isCandidate(static_cast<long>(i))) continue;
    if (pixels_.empty() || pixels_[i].empty()) continue;
    Mat33 R;
    Vec3 tvec;
    if (!SolvePnPEPnP(pts3d_[i], pixels_[i], cameras_[i].K(), cameras_[i].dist(), R, tvec)) continue;  // :572-576
    // cv::solvePnP hands back rvec; the reference converts it back with cv::Rodrigues (:578-579)
    R = Rodrigues(RodriguesInv(R));
    const Vec3 p0 = Mul(R, Vec3{pts3d_[i][0].x, pts3d_[i][0].y, pts3d_[i][0].z});
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (p0[2] + tvec[2] < 0 || det < 0.0) continue;  // :581-586
    // cv::projectPoints on float32 copies of the points, no distortion, float32 predictions (:588-596)
    const Mat33& K = cameras_[i].K();
    double err = 0;
    for (size_t j = 0; j < pts3d_[i].size(); ++j) {
      const Vec3 Xf = {static_cast<double>(static_cast<float>(pts3d_[i][j].x)), static_cast<double>(static_cast<float>(pts3d_[i][j].y)),
                       static_cast<double>(static_cast<float>(pts3d_[i][j].z))};
      const Vec3 X = Mul(R, Xf);
      const double x = (X[0] + tvec[0]) / (X[2] + tvec[2]), y = (X[1] + tvec[1]) / (X[2] + tvec[2]);
      const float pu = static_cast<float>(K[0] * x + K[2]), pv = static_cast<float>(K[4] * y + K[5]);
      err += static_cast<double>((pu - pixels_[i][j].x) * (pu - pixels_[i][j].x) + (pv - pixels_[i][j].y) * (pv - pixels_[i][j].y));
    }
    err = std::sqrt(err / static_cast<double>(pts3d_[i].size()));
    if (err > 300) continue;  // :602-605
    // T_l_w = T_i_l^-1 T_i_w (:615-621)
    const Mat33 Rt = Transpose(cameras_[i].R());
    const Mat33 R_l_w = Mul(Rt, R);
    const Vec3& ti = cameras_[i].t();
    const Vec3 t_l_w = Mul(Rt, Vec3{tvec[0] - ti[0], tvec[1] - ti[1], tvec[2] - ti[2]});
    const Vec3 rv = RodriguesInv(R_l_w);
    packed_.tlw = {{rv[0], rv[1], rv[2], t_l_w[0], t_l_w[1], t_l_w[2]}};
    packed_.tlw_init_ok = true;
    return true;
  }
  return false;
}

// SetUpInitialCameraParams (:635-670) + AddConstraints2d2d ordering (:799-850) + Pix2Ray (:768-797)
void PTZRayOptimizer::Pack()
{
  PackedBA& p = packed_;
  {
    PackedBA fresh;
    fresh.tlw = p.tlw;
    fresh.tlw_init_ok = p.tlw_init_ok;
    p = fresh;
  }
  std::vector<int> cam_of_image(num_cams_, -1);
  for (size_t i = 0; i < num_cams_; ++i) {
    if (!isCandidate(static_cast<long>(i))) continue;
    cam_of_image[i] = static_cast<int>(p.cam_image.size());
    p.cam_image.push_back(static_cast<long>(i));
    const std::vector<double> v = cameras_[i].ToVector();
    p.cam.insert(p.cam.end(), v.begin(), v.end());
  }
  // R^-1 K^-1 once per camera: cv::Mat evaluates R.inv() * K.inv() * uv from the left (ptzray_optimizer.cc:786)
  std::vector<Mat33> RKinv(p.cam_image.size());
  for (size_t c = 0; c < p.cam_image.size(); ++c)
    RKinv[c] = Mul(Inverse(cameras_[p.cam_image[c]].R()), Inverse(cameras_[p.cam_image[c]].K()));
  std::vector<char> is_cand(num_cams_, 0);
  for (size_t i = 0; i < num_cams_; ++i) is_cand[i] = isCandidate(static_cast<long>(i));
  {  // upper bounds (every view of every track a candidate): one allocation instead of the doubling sequence
    size_t max_obs = 0, max_ray = 0;
    if (shared_tracks_) { max_obs = shared_tracks_->img.size(); max_ray = shared_tracks_->id.size(); }
    else { max_ray = tracks_.size(); for (const auto& te : tracks_) max_obs += te.second.size(); }
    p.obs_uv.reserve(2 * max_obs); p.obs_cam.reserve(max_obs); p.obs_ray.reserve(max_obs);
    p.ray_track.reserve(max_ray); p.ray_weight.reserve(max_ray); p.ray.reserve(3 * max_ray);
  }
  auto add_track = [&](int track_id, size_t track_len, auto&& for_each_view) {
    Vec3 acc = {0, 0, 0};
    size_t n_cand = 0;
    const int ray_id = static_cast<int>(p.ray_track.size());
    for_each_view([&](int image, int feature) {
      if (image < 0 || static_cast<size_t>(image) >= num_cams_ || !is_cand[image]) return;
      const int c = cam_of_image[image];
      const Point2f pt = features_[image].keypoints[feature].pt;
      p.obs_uv.push_back(pt.x);
      p.obs_uv.push_back(pt.y);
      p.obs_cam.push_back(c);
      p.obs_ray.push_back(ray_id);
      const Vec3 t = Mul(RKinv[c], Vec3{pt.x, pt.y, 1.0});
      const double n = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      acc[0] += t[0] / n; acc[1] += t[1] / n; acc[2] += t[2] / n;
      ++n_cand;
    });
    if (n_cand == 0) return;  // no residual block is ever added for this track: not a parameter of the problem
    p.ray_track.push_back(track_id);
    p.ray_weight.push_back(static_cast<double>(track_len));  // FULL track length, also when only some views are candidates (:805)
    acc[0] /= n_cand; acc[1] /= n_cand; acc[2] /= n_cand;
    const double n = std::sqrt(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
    p.ray.push_back(acc[0] / n); p.ray.push_back(acc[1] / n); p.ray.push_back(acc[2] / n);
  };
  if (shared_tracks_) {
    const SharedTracks& st = *shared_tracks_;
    for (size_t k = 0; k < st.id.size(); ++k) {
      // skip tracks without a candidate view before touching anything else (most tracks, early in an incremental run)
      bool any = false;
      for (int64_t e = st.ptr[k]; e < st.ptr[k + 1] && !any; ++e) any = st.img[e] >= 0 && static_cast<size_t>(st.img[e]) < num_cams_ && is_cand[st.img[e]] != 0;
      if (!any) continue;
      add_track(st.id[k], static_cast<size_t>(st.ptr[k + 1] - st.ptr[k]), [&](auto&& view) {
        for (int64_t e = st.ptr[k]; e < st.ptr[k + 1]; ++e) view(st.img[e], st.feat[e]);
      });
    }
  }
  else {
    for (const auto& te : tracks_)  // std::map: ascending track id, ascending image id inside
      add_track(te.first, te.second.size(), [&](auto&& view) {
        for (const auto& kv : te.second) view(kv.first, kv.second);
      });
  }
  // AddConstraints2d3d (:887-923): candidate cameras ascending, annotation order within a camera
  for (size_t i = 0; i < num_cams_; ++i) {
    if (!isCandidate(static_cast<long>(i))) continue;
    if (pixels_.empty() || pixels_[i].empty()) continue;
    for (size_t j = 0; j < pixels_[i].size(); ++j) {
      p.obs3d_uv.push_back(pixels_[i][j].x); p.obs3d_uv.push_back(pixels_[i][j].y);
      p.obs3d_xyz.push_back(pts3d_[i][j].x); p.obs3d_xyz.push_back(pts3d_[i][j].y); p.obs3d_xyz.push_back(pts3d_[i][j].z);
      p.obs3d_cam.push_back(cam_of_image[i]);
    }
  }
}

// host-side Reproj2d3dFactor residual for the read-back statistics (CalReprojError2d3d, :1030-1072; functor :268-326)
// disp != nullptr: Reproj2d3dDispFactor (:334-396)
static void Residual2d3d(const double* c, const double* tlw, const double* Xw, float u, float v, double* res, const double* disp = nullptr)
{
  const Mat33 Rl = Rodrigues({tlw[0], tlw[1], tlw[2]});
  Vec3 Xl = Mul(Rl, Vec3{Xw[0], Xw[1], Xw[2]});
  Xl = {Xl[0] + tlw[3], Xl[1] + tlw[4], Xl[2] + tlw[5]};
  Vec3 P = Mul(Rodrigues({c[4], c[5], c[6]}), Xl);
  if (disp) P[2] += disp[0] + disp[1] * c[0] + disp[2] * c[0] * c[0];
  const double px = P[0] / P[2], py = P[1] / P[2];
  const double r2 = px * px + py * py, r4 = r2 * r2, r6 = r2 * r2 * r2;
  const double rad = 1.0 + c[10] * r2 + c[11] * r4 + c[12] * r6;
  const double xd = px * rad + 2.0 * c[13] * px * py + c[14] * (r2 + 2.0 * px * px);
  const double yd = py * rad + 2.0 * c[14] * px * py + c[13] * (r2 + 2.0 * py * py);
  res[0] = static_cast<double>(u) - (c[0] * xd + c[2]);
  res[1] = static_cast<double>(v) - (c[1] * yd + c[3]);
}

// host-side evaluation of the 2D-2D residual for the read-back statistics (CalReprojError2d2d, :970-1028)
static void Residual2d2d(FACTOR_TYPE type, const double* c, const Mat33& R, const double* X, float u, float v, double* res,
                         const double* disp = nullptr)
{
  Vec3 x = {X[0], X[1], X[2]};
  if (type != PTZRayDist) {  // PTZRay and PTZRayFxfyDist normalise the ray (:42, :161)
    const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    x = {x[0] / n, x[1] / n, x[2] / n};
  }
  Vec3 P = Mul(R, x);
  if (type == PTZRayDistDisp) P[2] += disp[0] + disp[1] * c[0] + disp[2] * c[0] * c[0];  // :233-236
  if (type == PTZRay) {
    res[0] = static_cast<double>(u) - (c[0] * P[0] + c[2] * P[2]) / P[2];
    res[1] = static_cast<double>(v) - (c[0] * P[1] + c[3] * P[2]) / P[2];
    return;
  }
  if (type == PTZRayDist && P[2] < 0) { res[0] = res[1] = 1000000.0; return; }  // :97-102, PTZRayDist only
  const double px = P[0] / P[2], py = P[1] / P[2];
  const double r2 = px * px + py * py, r4 = r2 * r2, r6 = r2 * r2 * r2;
  const double rad = 1.0 + c[10] * r2 + c[11] * r4 + c[12] * r6;
  const double xd = px * rad + 2.0 * c[13] * px * py + c[14] * (r2 + 2.0 * px * px);
  const double yd = py * rad + 2.0 * c[14] * px * py + c[13] * (r2 + 2.0 * py * py);
  res[0] = static_cast<double>(u) - (c[0] * xd + c[2]);
  res[1] = static_cast<double>(v) - ((type == PTZRayFxfyDist ? c[1] : c[0]) * yd + c[3]);  // fy read by PTZRayFxfyDist (:167)
}

void PTZRayOptimizer::ComputeErrors() const
{
  if (errors_ready_) return;
  errors_ready_ = true;
  const PackedBA& p = packed_;
  if (p.obs_cam.empty()) return;
  const std::vector<double>& cam = p.cam;
  const std::vector<double>& ray = p.ray;
  double sum0 = 0, sum1 = 0;
  std::vector<Mat33> Rc(p.cam_image.size());  // one cv::Rodrigues per camera instead of one per residual
  for (size_t c = 0; c < Rc.size(); ++c) Rc[c] = Rodrigues({cam[15 * c + 4], cam[15 * c + 5], cam[15 * c + 6]});
  for (size_t a = 0; a < p.obs_cam.size(); ++a) {
    double res[2];
    Residual2d2d(type_, &cam[15 * static_cast<size_t>(p.obs_cam[a])], Rc[p.obs_cam[a]], &ray[3 * static_cast<size_t>(p.obs_ray[a])], p.obs_uv[2 * a],
                 p.obs_uv[2 * a + 1], res, disp_.data());
    sum0 += res[0] * res[0];
    sum1 += res[1] * res[1];
  }
  final_reproj_error_2d2d_ = std::sqrt((sum0 + sum1) / static_cast<double>(p.obs_cam.size()));
  if (p.obs3d_cam.empty()) {
    final_reproj_error_2d3d_ = std::numeric_limits<double>::quiet_NaN();  // sqrt(0 / 0) in the reference when there are no annotations
  }
  else {
    double s0 = 0, s1 = 0;
    for (size_t a = 0; a < p.obs3d_cam.size(); ++a) {
      double res[2];
      Residual2d3d(&cam[15 * static_cast<size_t>(p.obs3d_cam[a])], p.tlw.data(), &p.obs3d_xyz[3 * a], p.obs3d_uv[2 * a], p.obs3d_uv[2 * a + 1], res,
                   type_ == PTZRayDistDisp ? disp_.data() : nullptr);
      s0 += res[0] * res[0];
      s1 += res[1] * res[1];
    }
    final_reproj_error_2d3d_ = std::sqrt((s0 + s1) / static_cast<double>(p.obs3d_cam.size()));
  }
}

bool PTZRayOptimizer::Solve(std::vector<Camera>& cameras)
{
  // the reference builds the ray lists and throws them away here (ptzray_optimizer.cc:448-452); nothing reads them
  return SolveImpl(cameras, nullptr);
}

bool PTZRayOptimizer::Solve(std::vector<Camera>& cameras, std::vector<std::vector<Ray>>& rays) { return SolveImpl(cameras, &rays); }

// Solve over a view of the resident rig: candidate cameras (ascending image id, SetUpInitialCameraParams' order :635-670) and their
// R^-1 K^-1 for Pix2Ray (:786) are all the host prepares; residual blocks, weights, ray initialisation and the solve are the
// device's (ptz_ba_batch_create_views, ptz_ba_batch_set_state_pix2ray).
bool PTZRayOptimizer::SolveView(std::vector<Camera>& cameras)
{
  PackedBA& p = packed_;
  p = PackedBA();
  std::vector<int32_t> cam_image;
  for (size_t i = 0; i < num_cams_; ++i) {
    if (!isCandidate(static_cast<long>(i))) continue;
    p.cam_image.push_back(static_cast<long>(i));
    cam_image.push_back(static_cast<int32_t>(i));
    const std::vector<double> v = cameras_[i].ToVector();
    p.cam.insert(p.cam.end(), v.begin(), v.end());
  }
  if (cam_image.empty()) return false;
  std::vector<double> rkinv(9 * cam_image.size());
  for (size_t c = 0; c < cam_image.size(); ++c) {
    const Mat33 m = Mul(Inverse(cameras_[cam_image[c]].R()), Inverse(cameras_[cam_image[c]].K()));  // as Pack() evaluates it
    for (int k = 0; k < 9; ++k) rkinv[9 * c + k] = m[k];
  }
  ptz_lm_options opt;
  ptz_lm_options_default(&opt);
  opt.max_num_iterations = max_iter_;  // ptzray_optimizer.cc:470
  opt.device_id = device_id_;
  const int32_t ftype = (type_ == PTZRay) ? PTZ_BA_PTZRay : (type_ == PTZRayDist ? PTZ_BA_PTZRayDist : PTZ_BA_PTZRayFxfyDist);
  ptz_rig_view view{rig_, static_cast<int32_t>(cam_image.size()), cam_image.data()};
  std::vector<double> cam = p.cam;
  tlw_init_ = p.tlw;
  disp_ = {{0.0, 0.0, 0.0}};
  const auto t_dev = std::chrono::steady_clock::now();
  const int32_t rc = DeviceBaSolveView(&view, ftype, cam.data(), rkinv.data(), &opt, &summary_);
  device_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev).count();
  if (rc == PTZ_ENOOBS) return false;  // no track has a candidate view: no residual block, not a problem (the packed path returns false there too)
  if (rc != PTZ_OK) {
    fprintf(stderr, "[ptzcalib] PTZRayOptimizer::Solve: device solve not run, ptz_ba_batch_create_views / solve returned %d\n", (int)rc);
    return false;
  }
  init_reproj_error_all_ = std::sqrt(2.0) * std::sqrt((2 * summary_.initial_cost) / summary_.num_residuals);
  final_reproj_error_all_ = std::sqrt(2.0) * std::sqrt((2 * summary_.final_cost) / summary_.num_residuals);
  p.cam = cam;
  errors_ready_ = true;  // (the unweighted statistics need the packed residuals; callers of this path -- the incremental pipeline -- never read them)
  final_reproj_error_2d2d_ = final_reproj_error_2d3d_ = std::numeric_limits<double>::quiet_NaN();
  if (summary_.termination_type != PTZ_CONVERGENCE) return false;  // :482-488
  // ObtainRefinedCameraParams (:672-766) with T_l_w = identity (no annotations on this path)
  if (cameras.size() < num_cams_) cameras.resize(num_cams_);
  for (size_t c = 0; c < p.cam_image.size(); ++c) {
    std::vector<double> param(cam.begin() + 15 * c, cam.begin() + 15 * (c + 1));
    if (type_ != PTZRayFxfyDist) param[1] = param[0];  // fy := fx (:705-706)
    Camera& out = cameras[p.cam_image[c]];
    out.FromVector(param);
    // (the packed path composes with T_l_w = (I, 0) here: R I and R 0 + t, the same values)
    const Mat33 R_l_w = Rodrigues({0.0, 0.0, 0.0});
    const Vec3 Rt = Mul(out.R(), Vec3{0.0, 0.0, 0.0});
    out.t() = {Rt[0] + out.t()[0], Rt[1] + out.t()[1], Rt[2] + out.t()[2]};
    out.R() = Mul(out.R(), R_l_w);
  }
  return true;
}

bool PTZRayOptimizer::SolveImpl(std::vector<Camera>& cameras, std::vector<std::vector<Ray>>* rays_out)
{
  if (!CheckValid()) return false;
  FindTracks();
  SetInitTransLocalToWorld();
  if (type_ != PTZRay && type_ != PTZRayDist && type_ != PTZRayFxfyDist && type_ != PTZRayDistDisp) return false;
  // A view of device-resident tracks (UseRig): nothing is packed on the host but the candidate cameras themselves.
  bool view_path = rig_ != nullptr && rays_out == nullptr && pixels_.empty() && type_ != PTZRayDistDisp;
  for (size_t i = 0; view_path && i < num_cams_; ++i)
    if (isCandidate(static_cast<long>(i)) && shared_ic_ids_[i] != static_cast<long>(i)) view_path = false;
  if (view_path) return SolveView(cameras);
  Pack();
  PackedBA& p = packed_;
  if (p.obs_cam.empty() || p.ray_track.empty()) return false;

  ptz_ba_problem prob{};
  prob.n_cam = static_cast<int32_t>(p.cam_image.size());
  prob.n_ray = static_cast<int32_t>(p.ray_track.size());
  prob.n_obs = static_cast<int64_t>(p.obs_cam.size());
  prob.obs_uv = p.obs_uv.data();
  prob.obs_cam = p.obs_cam.data();
  prob.obs_ray = p.obs_ray.data();
  prob.ray_weight = p.ray_weight.data();
  prob.factor_type = (type_ == PTZRay) ? PTZ_BA_PTZRay : (type_ == PTZRayDist ? PTZ_BA_PTZRayDist : (type_ == PTZRayFxfyDist ? PTZ_BA_PTZRayFxfyDist : PTZ_BA_PTZRayDistDisp));
  // SetSharedIntrinsics: the intrinsics block id of every candidate camera (ptzray_optimizer.cc:643-650)
  p.ic_of_cam.clear();
  bool shared = false;
  for (long image : p.cam_image) {
    p.ic_of_cam.push_back(static_cast<int32_t>(shared_ic_ids_[image]));
    shared |= shared_ic_ids_[image] != image;
  }
  if (shared) prob.ic_of_cam = p.ic_of_cam.data();
  prob.n_obs3d = static_cast<int32_t>(p.obs3d_cam.size());
  if (prob.n_obs3d > 0) {
    prob.obs3d_uv = p.obs3d_uv.data();
    prob.obs3d_xyz = p.obs3d_xyz.data();
    prob.obs3d_cam = p.obs3d_cam.data();
  }
  ptz_lm_options opt;
  ptz_lm_options_default(&opt);
  opt.max_num_iterations = max_iter_;  // ptzray_optimizer.cc:470
  opt.device_id = device_id_;
  std::vector<double> cam = p.cam, ray = p.ray;
  double tlw[6];
  tlw_init_ = p.tlw;
  for (int k = 0; k < 6; ++k) tlw[k] = p.tlw[k];
  const auto t_dev = std::chrono::steady_clock::now();
  disp_ = {{0.0, 0.0, 0.0}};  // disp_param_ starts at zero (:655)
  const int32_t rc = type_ == PTZRayDistDisp ? ptz_ba_solve_disp(&prob, cam.data(), ray.data(), tlw, disp_.data(), &opt, &summary_)
                                             : DeviceBaSolve(&prob, cam.data(), ray.data(), tlw, &opt, &summary_);  // (through the thread's DeviceBatcher, if any)
  device_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev).count();
  if (rc != PTZ_OK) {
    // not a convergence failure: the device path refused or could not run the problem.  Say so -- the reference's callers
    // (PtzIncrementalOptimizer withdraws the newest image after a failed bundle adjustment) would otherwise mistake it for one.
    fprintf(stderr, "[ptzcalib] PTZRayOptimizer::Solve: device solve not run, ptz_ba_solve returned %d (%s)\n", (int)rc,
            rc == PTZ_EUNSUPPORTED ? "factor type not on the device path" : rc == PTZ_ELIMIT ? "problem dimension beyond the device path's limits, see PTZ_ELIMIT"
            : rc == PTZ_ENOMEM ? "out of device memory" : rc == PTZ_ENODEVICE ? "no usable HIP device / runtime error" : "malformed problem");
    return false;
  }

  // CalReprojError (:960-968)
  init_reproj_error_all_ = std::sqrt(2.0) * std::sqrt((2 * summary_.initial_cost) / summary_.num_residuals);
  final_reproj_error_all_ = std::sqrt(2.0) * std::sqrt((2 * summary_.final_cost) / summary_.num_residuals);
  // CalReprojError2d2d / 2d3d (:970-1072) walk every residual on the host: evaluated on first use of the accessors (the
  // incremental pipeline never reads them), from the solved state kept in packed_
  p.cam = cam;
  p.ray = ray;
  for (int k = 0; k < 6; ++k) p.tlw[k] = tlw[k];
  errors_ready_ = false;

  if (summary_.termination_type != PTZ_CONVERGENCE) return false;  // :482-488

  // ObtainRefinedCameraParams (:672-766)
  if (cameras.size() < num_cams_) cameras.resize(num_cams_);
  Mat33 R_l_w;
  Vec3 t_l_w;
  T_l_w(tlw, R_l_w, t_l_w);
  for (size_t c = 0; c < p.cam_image.size(); ++c) {
    std::vector<double> param(cam.begin() + 15 * c, cam.begin() + 15 * (c + 1));
    // fy := fx for PTZRay / PTZRayDist, also after fy was a free parameter of the annotations (:705-706); PTZRayFxfyDist keeps
    // its own fy (:683-685)
    if (type_ != PTZRayFxfyDist) param[1] = param[0];
    // the displacement block goes into t_z (:693, :714; zero for every other type)
    param[9] += disp_[0] + disp_[1] * param[0] + disp_[2] * param[0] * param[0];
    Camera& out = cameras[p.cam_image[c]];
    out.FromVector(param);
    // local -> world: T_i_w = T_i_l T_l_w (:729-740)
    const Vec3 Rt = Mul(out.R(), t_l_w);
    out.t() = {Rt[0] + out.t()[0], Rt[1] + out.t()[1], Rt[2] + out.t()[2]};
    out.R() = Mul(out.R(), R_l_w);
  }
  const Mat33 R_w_l = Transpose(R_l_w);
  const Vec3 Rtt = Mul(R_w_l, t_l_w);
  if (rays_out) {
    rays_out->clear();
    rays_out->resize(num_cams_);
  }
  for (size_t j = 0; rays_out && j < p.ray_track.size(); ++j) {
    std::vector<std::vector<Ray>>& rays = *rays_out;
    const Vec3 rl = Mul(R_w_l, Vec3{ray[3 * j], ray[3 * j + 1], ray[3 * j + 2]});
    const Vec3 ray_w = {rl[0] - Rtt[0], rl[1] - Rtt[1], rl[2] - Rtt[2]};  // R_w_l ray_l + t_w_l (:746-754)
    if (shared_tracks_) {
      const SharedTracks& st = *shared_tracks_;
      const size_t k = static_cast<size_t>(std::lower_bound(st.id.begin(), st.id.end(), p.ray_track[j]) - st.id.begin());
      for (int64_t e = st.ptr[k]; e < st.ptr[k + 1]; ++e)
        rays[st.img[e]].emplace_back(p.ray_track[j], ray_w, features_[st.img[e]].keypoints[st.feat[e]].pt);
    }
    else {
      for (const auto& kv : tracks_.at(p.ray_track[j]))
        rays[kv.first].emplace_back(p.ray_track[j], ray_w, features_[kv.first].keypoints[kv.second].pt);
    }
  }
  p.cam = cam;
  p.ray = ray;
  for (int k = 0; k < 6; ++k) p.tlw[k] = tlw[k];
  return true;
}

}  // namespace ptzcalib
