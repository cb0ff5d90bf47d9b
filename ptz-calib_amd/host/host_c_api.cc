// host_c_api.cc -- flat C entry points over the C++ host classes so that pytest (ctypes) can drive them.
// Not part of the reference's surface; the C++ classes are.
#include <cstdlib>
#include <cstring>

#include "krt_optimizer.h"
#include "ptzray_optimizer.h"

using namespace ptzcalib;

namespace {
void BuildInputs(int n_img, const int64_t* kp_ptr, const float* kp_xy, const int32_t* img_wh, int n_pairs, const int64_t* src,
                 const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t, const double* cam15,
                 std::vector<ImageFeatures>& feats, std::vector<MatchesInfo>& mis, std::vector<Camera>& cams)
{
  feats.resize(n_img);
  cams.resize(n_img);
  for (int i = 0; i < n_img; ++i) {
    feats[i].img_idx = i;
    if (img_wh) { feats[i].img_size.width = img_wh[2 * i]; feats[i].img_size.height = img_wh[2 * i + 1]; }
    for (int64_t k = kp_ptr[i]; k < kp_ptr[i + 1]; ++k) {
      KeyPoint kp;
      kp.pt = Point2f(kp_xy[2 * k], kp_xy[2 * k + 1]);
      feats[i].keypoints.push_back(kp);
    }
    if (cam15) cams[i].FromVector(std::vector<double>(cam15 + 15 * i, cam15 + 15 * (i + 1)));
  }
  mis.resize(n_pairs);
  for (int p = 0; p < n_pairs; ++p) {
    mis[p].src_img_idx = src[p];
    mis[p].dst_img_idx = dst[p];
    for (int64_t k = match_ptr[p]; k < match_ptr[p + 1]; ++k) {
      DMatch m;
      m.queryIdx = q[k];
      m.trainIdx = t[k];
      mis[p].matches.push_back(m);
    }
  }
}
template <typename T> T* Dup(const std::vector<T>& v)
{
  T* p = static_cast<T*>(malloc(sizeof(T) * (v.size() + 1)));
  if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}
}  // namespace

extern "C" {

void ptzh_free(void* p) { free(p); }

// TracksBuilder Build/Filter/ExportToSTL; same output convention as the oracle's orc_tracks_build
int32_t ptzh_tracks_build(int32_t n_pairs, const int64_t* src, const int64_t* dst, const int64_t* match_ptr, const int32_t* q,
                          const int32_t* t, int32_t min_len, int32_t** track_id, int64_t** track_ptr, int32_t** eimg, int32_t** efeat)
{
  std::vector<MatchesInfo> mis(n_pairs);
  for (int p = 0; p < n_pairs; ++p) {
    mis[p].src_img_idx = src[p];
    mis[p].dst_img_idx = dst[p];
    for (int64_t k = match_ptr[p]; k < match_ptr[p + 1]; ++k) { DMatch m; m.queryIdx = q[k]; m.trainIdx = t[k]; mis[p].matches.push_back(m); }
  }
  TracksBuilder b;
  b.Build(mis);
  b.Filter(min_len);
  Tracks tr;
  b.ExportToSTL(tr);
  std::vector<int32_t> ids, ei, ef;
  std::vector<int64_t> ptr{0};
  for (const auto& te : tr) {
    ids.push_back(te.first);
    for (const auto& kv : te.second) { ei.push_back(kv.first); ef.push_back(kv.second); }
    ptr.push_back(static_cast<int64_t>(ei.size()));
  }
  *track_id = Dup(ids); *track_ptr = Dup(ptr); *eimg = Dup(ei); *efeat = Dup(ef);
  return static_cast<int32_t>(ids.size());
}

// Full PTZRayOptimizer::Solve through the C++ class.  cam15 [15*n_img] in/out (ToVector layout).
// packed_* outputs (malloc'ed, may be NULL): the packed problem the class handed to the C-ABI.
int32_t ptzh_ptzray_solve(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, int32_t n_pairs, const int64_t* src,
                          const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t, double* cam15,
                          const int64_t* cand_ids, int32_t n_cand, int32_t max_iter, int32_t type, int32_t solve_on_device,
                          double* errors3, ptz_lm_summary* summary, int32_t* n_obs_out, int32_t* n_ray_out, float** p_uv,
                          int32_t** p_cam, int32_t** p_ray, double** p_w, double** p_cam15, double** p_ray3, int64_t** p_cam_image)
{
  std::vector<ImageFeatures> feats;
  std::vector<MatchesInfo> mis;
  std::vector<Camera> cams;
  BuildInputs(n_img, kp_ptr, kp_xy, nullptr, n_pairs, src, dst, match_ptr, q, t, cam15, feats, mis, cams);
  std::unordered_set<long> ids;
  for (int i = 0; i < n_cand; ++i) ids.insert(static_cast<long>(cand_ids[i]));
  PTZRayOptimizer opt(feats, mis, cams, ids, solve_on_device ? max_iter : 0, static_cast<FACTOR_TYPE>(type));
  bool ok = false;
  if (solve_on_device) {
    ok = opt.Solve(cams);
  }
  else {
    // packing only (CPU tests): run the pre-solve stages of Solve through a friend-free path: Solve() with max_iter 0 fails
    // CheckValid before touching the device, so re-create with max_iter 1 and call the public pieces via packed().
    PTZRayOptimizer opt2(feats, mis, cams, ids, 1, static_cast<FACTOR_TYPE>(type));
    opt2.Solve(cams);  // without a device ptz_ba_solve returns PTZ_ENODEVICE -> Solve returns false, packed() is filled
    const PackedBA& p = opt2.packed();
    if (n_obs_out) *n_obs_out = static_cast<int32_t>(p.obs_cam.size());
    if (n_ray_out) *n_ray_out = static_cast<int32_t>(p.ray_track.size());
    if (p_uv) *p_uv = Dup(p.obs_uv);
    if (p_cam) *p_cam = Dup(p.obs_cam);
    if (p_ray) *p_ray = Dup(p.obs_ray);
    if (p_w) *p_w = Dup(p.ray_weight);
    if (p_cam15) *p_cam15 = Dup(p.cam);
    if (p_ray3) *p_ray3 = Dup(p.ray);
    if (p_cam_image) { std::vector<int64_t> ci(p.cam_image.begin(), p.cam_image.end()); *p_cam_image = Dup(ci); }
    return 0;
  }
  const PackedBA& p = opt.packed();
  if (n_obs_out) *n_obs_out = static_cast<int32_t>(p.obs_cam.size());
  if (n_ray_out) *n_ray_out = static_cast<int32_t>(p.ray_track.size());
  if (p_uv) *p_uv = Dup(p.obs_uv);
  if (p_cam) *p_cam = Dup(p.obs_cam);
  if (p_ray) *p_ray = Dup(p.obs_ray);
  if (p_w) *p_w = Dup(p.ray_weight);
  if (p_cam15) *p_cam15 = Dup(p.cam);
  if (p_ray3) *p_ray3 = Dup(p.ray);
  if (p_cam_image) { std::vector<int64_t> ci(p.cam_image.begin(), p.cam_image.end()); *p_cam_image = Dup(ci); }
  if (errors3) { errors3[0] = opt.final_reproj_error_all(); errors3[1] = opt.final_reproj_error_2d2d(); errors3[2] = opt.final_reproj_error_2d3d(); }
  if (summary) *summary = opt.summary();
  if (ok)
    for (int i = 0; i < n_img; ++i) {
      const std::vector<double> v = cams[i].ToVector();
      memcpy(cam15 + 15 * i, v.data(), sizeof(double) * 15);
    }
  return ok ? 1 : 0;
}

// KRTOptimizer through the C++ class: one query.  cam_cur15 in (initial, world) / out (refined, world).
int32_t ptzh_krt_solve(const double* cam_ref15, double* cam_cur15, int32_t n_match, const float* uv_ref, const float* uv_cur,
                       int32_t max_iter, double max_reproj_error, int32_t type, int32_t* num_iter, ptz_lm_summary* summary)
{
  Camera ref, cur;
  ref.FromVector(std::vector<double>(cam_ref15, cam_ref15 + 15));
  cur.FromVector(std::vector<double>(cam_cur15, cam_cur15 + 15));
  std::vector<KeyPoint> kr(n_match), kc(n_match);
  std::vector<DMatch> ms(n_match);
  for (int m = 0; m < n_match; ++m) {
    kr[m].pt = Point2f(uv_ref[2 * m], uv_ref[2 * m + 1]);
    kc[m].pt = Point2f(uv_cur[2 * m], uv_cur[2 * m + 1]);
    ms[m].queryIdx = m; ms[m].trainIdx = m;
  }
  KRTOptimizer opt(max_iter, max_reproj_error, static_cast<KRTOptimizer::FACTOR_TYPE>(type));
  opt.SetInitParams(cur.K(), cur.R(), cur.t(), cur.dist());
  opt.Add2d2dConstraints(ref, kr, kc, ms);
  Mat33 K, R; Vec3 t; Vec5 dist;
  const bool ok = opt.Solve(K, R, t, dist);
  if (num_iter) *num_iter = opt.num_iter_;
  if (summary) *summary = opt.summary();
  if (ok) {
    const std::vector<double> v = Camera(K, R, t, dist).ToVector();
    memcpy(cam_cur15, v.data(), sizeof(double) * 15);
  }
  return ok ? 1 : 0;
}

}  // extern "C"
