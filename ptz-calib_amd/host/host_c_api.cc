// host_c_api.cc -- flat C entry points over the C++ host classes so that pytest (ctypes) can drive them.
// Not part of the reference's surface; the C++ classes are.
#include <cstdlib>
#include <cstring>

#include "data_io.h"
#include "epnp.h"
#include "homography.h"
#include "image_size.h"
#include "json_mini.h"
#include "krt_optimizer.h"
#include <algorithm>
#include <chrono>
#include <stdexcept>
#include <memory>
#include <unordered_set>

#include "ptz_incremental_optimizer.h"
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

// Camera::FromVector -> Camera::ToVector round trip (cv::Rodrigues both ways) for the tests
void ptzh_camera_roundtrip(const double* in15, double* out15, double* R9)
{
  Camera c;
  c.FromVector(std::vector<double>(in15, in15 + 15));
  const std::vector<double> v = c.ToVector();
  memcpy(out15, v.data(), sizeof(double) * 15);
  if (R9) memcpy(R9, c.R().data(), sizeof(double) * 9);
}
// rotation vector of an arbitrary (possibly scaled, noisy) 3x3 matrix, as cv::Rodrigues(matrix) gives it
void ptzh_rodrigues_inv(const double* R9, double* rvec)
{
  Mat33 R;
  for (int k = 0; k < 9; ++k) R[k] = R9[k];
  const Vec3 r = RodriguesInv(R);
  rvec[0] = r[0]; rvec[1] = r[1]; rvec[2] = r[2];
}

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
  // the flat export (what the shared-track path of the optimizers uses) must be the map export, entry for entry
  {
    std::vector<int> fid, fimg, ffeat;
    std::vector<int64_t> fptr;
    b.ExportFlat(fid, fptr, fimg, ffeat);
    const bool same = fid.size() == ids.size() && fptr == ptr && std::equal(fid.begin(), fid.end(), ids.begin()) &&
                      fimg.size() == ei.size() && std::equal(fimg.begin(), fimg.end(), ei.begin()) &&
                      std::equal(ffeat.begin(), ffeat.end(), ef.begin());
    if (!same) return -2;
  }
  *track_id = Dup(ids); *track_ptr = Dup(ptr); *eimg = Dup(ei); *efeat = Dup(ef);
  return static_cast<int32_t>(ids.size());
}

namespace {
struct PackedOut {
  int32_t *n_obs, *n_ray;
  float** uv; int32_t** cam; int32_t** ray; double** w; double** cam15; double** ray3; int64_t** cam_image;
};
void ExportPacked(const PackedBA& p, const PackedOut& o)
{
  if (o.n_obs) *o.n_obs = static_cast<int32_t>(p.obs_cam.size());
  if (o.n_ray) *o.n_ray = static_cast<int32_t>(p.ray_track.size());
  if (o.uv) *o.uv = Dup(p.obs_uv);
  if (o.cam) *o.cam = Dup(p.obs_cam);
  if (o.ray) *o.ray = Dup(p.obs_ray);
  if (o.w) *o.w = Dup(p.ray_weight);
  if (o.cam15) *o.cam15 = Dup(p.cam);
  if (o.ray3) *o.ray3 = Dup(p.ray);
  if (o.cam_image) { std::vector<int64_t> ci(p.cam_image.begin(), p.cam_image.end()); *o.cam_image = Dup(ci); }
}
}  // namespace

// Full PTZRayOptimizer::Solve through the C++ class.  cam15 [15*n_img] in/out (ToVector layout).
// Annotations (may be NULL): ann_ptr [n_img+1] prefix offsets into ann_uv [2*] / ann_xyz [3*] (pixels_ / pts3d_ of the
// reference's second constructor).  tlw_out [14]: initial T_l_w (6), refined T_l_w (6), 1.0 if the PnP initialisation passed, number of packed 2D-3D blocks.
// packed_* outputs (malloc'ed, may be NULL): the packed problem the class handed to the C-ABI.
// solve_on_device == 0: packing only (CPU tests); without a device ptz_ba_solve returns PTZ_ENODEVICE, Solve returns
// false and packed() holds the packed problem and the initial T_l_w.
int32_t ptzh_ptzray_georef(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, int32_t n_pairs, const int64_t* src,
                           const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t, double* cam15,
                           const int64_t* ann_ptr, const float* ann_uv, const double* ann_xyz, const int64_t* cand_ids,
                           int32_t n_cand, int32_t max_iter, int32_t type, int32_t solve_on_device, double* errors3,
                           ptz_lm_summary* summary, double* tlw_out, int32_t* n_obs_out, int32_t* n_ray_out, float** p_uv,
                           int32_t** p_cam, int32_t** p_ray, double** p_w, double** p_cam15, double** p_ray3, int64_t** p_cam_image)
{
  std::vector<ImageFeatures> feats;
  std::vector<MatchesInfo> mis;
  std::vector<Camera> cams;
  BuildInputs(n_img, kp_ptr, kp_xy, nullptr, n_pairs, src, dst, match_ptr, q, t, cam15, feats, mis, cams);
  std::unordered_set<long> ids;
  for (int i = 0; i < n_cand; ++i) ids.insert(static_cast<long>(cand_ids[i]));
  std::vector<std::vector<Point2f>> pixels;
  std::vector<std::vector<Point3d>> pts3d;
  if (ann_ptr) {
    pixels.resize(n_img);
    pts3d.resize(n_img);
    for (int i = 0; i < n_img; ++i)
      for (int64_t k = ann_ptr[i]; k < ann_ptr[i + 1]; ++k) {
        pixels[i].emplace_back(ann_uv[2 * k], ann_uv[2 * k + 1]);
        pts3d[i].emplace_back(ann_xyz[3 * k], ann_xyz[3 * k + 1], ann_xyz[3 * k + 2]);
      }
  }
  PTZRayOptimizer opt(feats, mis, cams, pixels, pts3d, ids, solve_on_device ? max_iter : 1, static_cast<FACTOR_TYPE>(type));
  const std::array<double, 6> zero{{0, 0, 0, 0, 0, 0}};
  const PackedOut po{n_obs_out, n_ray_out, p_uv, p_cam, p_ray, p_w, p_cam15, p_ray3, p_cam_image};
  if (!solve_on_device) {
    opt.Solve(cams);
    const PackedBA& p = opt.packed();
    ExportPacked(p, po);
    if (tlw_out) {
      for (int k = 0; k < 6; ++k) { tlw_out[k] = p.tlw[k]; tlw_out[6 + k] = zero[k]; }
      tlw_out[12] = p.tlw_init_ok ? 1.0 : 0.0;
    tlw_out[13] = static_cast<double>(p.obs3d_cam.size());
      tlw_out[13] = static_cast<double>(p.obs3d_cam.size());
    }
    return 0;
  }
  const bool ok = opt.Solve(cams);
  const PackedBA& p = opt.packed();
  if (tlw_out) {
    for (int k = 0; k < 6; ++k) { tlw_out[k] = opt.initial_tlw()[k]; tlw_out[6 + k] = p.tlw[k]; }
    tlw_out[12] = p.tlw_init_ok ? 1.0 : 0.0;
    tlw_out[13] = static_cast<double>(p.obs3d_cam.size());
  }
  ExportPacked(p, po);
  if (errors3) { errors3[0] = opt.final_reproj_error_all(); errors3[1] = opt.final_reproj_error_2d2d(); errors3[2] = opt.final_reproj_error_2d3d(); }
  if (summary) *summary = opt.summary();
  if (ok)
    for (int i = 0; i < n_img; ++i) {
      const std::vector<double> v = cams[i].ToVector();
      memcpy(cam15 + 15 * i, v.data(), sizeof(double) * 15);
    }
  return ok ? 1 : 0;
}

// Same as ptzh_ptzray_georef without annotations, with PTZRayOptimizer::SetSharedIntrinsics(shared_ic_ids [n_img]) applied.
int32_t ptzh_ptzray_solve_shared(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, int32_t n_pairs, const int64_t* src,
                                 const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t, double* cam15,
                                 const int64_t* shared_ic_ids, int32_t max_iter, int32_t type, ptz_lm_summary* summary)
{
  std::vector<ImageFeatures> feats;
  std::vector<MatchesInfo> mis;
  std::vector<Camera> cams;
  BuildInputs(n_img, kp_ptr, kp_xy, nullptr, n_pairs, src, dst, match_ptr, q, t, cam15, feats, mis, cams);
  PTZRayOptimizer opt(feats, mis, cams, {}, max_iter, static_cast<FACTOR_TYPE>(type));
  opt.SetSharedIntrinsics(std::vector<long>(shared_ic_ids, shared_ic_ids + n_img));
  const bool ok = opt.Solve(cams);
  if (summary) *summary = opt.summary();
  if (ok)
    for (int i = 0; i < n_img; ++i) {
      const std::vector<double> v = cams[i].ToVector();
      memcpy(cam15 + 15 * i, v.data(), sizeof(double) * 15);
    }
  return ok ? 1 : 0;
}

int32_t ptzh_ptzray_solve(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, int32_t n_pairs, const int64_t* src,
                          const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t, double* cam15,
                          const int64_t* cand_ids, int32_t n_cand, int32_t max_iter, int32_t type, int32_t solve_on_device,
                          double* errors3, ptz_lm_summary* summary, int32_t* n_obs_out, int32_t* n_ray_out, float** p_uv,
                          int32_t** p_cam, int32_t** p_ray, double** p_w, double** p_cam15, double** p_ray3, int64_t** p_cam_image)
{
  return ptzh_ptzray_georef(n_img, kp_ptr, kp_xy, n_pairs, src, dst, match_ptr, q, t, cam15, nullptr, nullptr, nullptr, cand_ids,
                            n_cand, max_iter, type, solve_on_device, errors3, summary, nullptr, n_obs_out, n_ray_out, p_uv, p_cam,
                            p_ray, p_w, p_cam15, p_ray3, p_cam_image);
}

// cv::solvePnP(..., SOLVEPNP_EPNP) replacement on its own: K9 row-major, dist5 (k1,k2,k3,p1,p2), out R9 row-major, t3.
int32_t ptzh_epnp(int32_t n, const double* xyz, const float* uv, const double* K9, const double* dist5, double* R9, double* t3)
{
  std::vector<Point3d> pw;
  std::vector<Point2f> px;
  for (int i = 0; i < n; ++i) { pw.emplace_back(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]); px.emplace_back(uv[2 * i], uv[2 * i + 1]); }
  Mat33 K, R; Vec5 d; Vec3 tv;
  for (int k = 0; k < 9; ++k) K[k] = K9[k];
  for (int k = 0; k < 5; ++k) d[k] = dist5[k];
  if (!SolvePnPEPnP(pw, px, K, d, R, tv)) return 0;
  for (int k = 0; k < 9; ++k) R9[k] = R[k];
  for (int k = 0; k < 3; ++k) t3[k] = tv[k];
  return 1;
}

// PtzIncrementalOptimizer::Solve.  Pairs are the non-empty cells of the reference's N x N match table in table order;
// H [9*n_pairs] row-major, h_valid [n_pairs] (0 = cv::Mat::empty()), confidence [n_pairs].  cam15 in/out.
// registered [n_img] out (0/1).  events: up to max_events rows of (kind, a, b, success); returns the number of events;
// *solved = Solve's return value.
int32_t ptzh_incremental_solve(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, const int32_t* img_wh, int32_t n_pairs,
                               const int64_t* src, const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t,
                               const double* H, const int32_t* h_valid, const double* confidence, double* cam15,
                               const int64_t* seeds, int32_t n_seeds, int32_t max_iter, int32_t* registered, int64_t* events,
                               int32_t max_events, int64_t* lm_iterations, int32_t* solved, double* timing7)
{  // timing7: the optimizer's five figures, then the wall time of its construction and of Solve() alone (milliseconds) -- the
   // conversion of the flat arrays into ImageFeatures / MatchesInfo objects above them is this harness's, not the library's
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<ImageFeatures> feats;
  std::vector<MatchesInfo> mis;
  std::vector<Camera> cams;
  BuildInputs(n_img, kp_ptr, kp_xy, img_wh, n_pairs, src, dst, match_ptr, q, t, cam15, feats, mis, cams);
  for (int p = 0; p < n_pairs; ++p) {
    for (int k = 0; k < 9; ++k) mis[p].H[k] = H[9 * p + k];
    mis[p].H_empty = h_valid[p] == 0;
    mis[p].confidence = confidence[p];
  }
  const double t_c0 = now_ms();
  PtzIncrementalOptimizer opt(feats, mis, cams, max_iter);
  if (n_seeds > 0) opt.SetSeedImageId(std::vector<long>(seeds, seeds + n_seeds));
  std::unordered_set<long> reg;
  const double t_s0 = now_ms();
  const bool ok = opt.Solve(cams, reg);
  const double t_s1 = now_ms();
  const auto& ev = opt.events();
  const int32_t ne = static_cast<int32_t>(ev.size()) < max_events ? static_cast<int32_t>(ev.size()) : max_events;
  for (int32_t e = 0; e < ne; ++e) {
    events[4 * e] = ev[e].kind; events[4 * e + 1] = ev[e].a; events[4 * e + 2] = ev[e].b; events[4 * e + 3] = ev[e].success ? 1 : 0;
  }
  if (lm_iterations) *lm_iterations = opt.lm_iterations();
  if (timing7) {
    memcpy(timing7, opt.timing_ms(), sizeof(double) * 5);
    timing7[5] = t_s0 - t_c0;
    timing7[6] = t_s1 - t_s0;
  }
  if (solved) *solved = ok ? 1 : 0;
  if (!ok) return ne;
  for (int i = 0; i < n_img; ++i) {
    registered[i] = reg.count(i) ? 1 : 0;
    const std::vector<double> v = cams[i].ToVector();
    memcpy(cam15 + 15 * i, v.data(), sizeof(double) * 15);
  }
  return ne;
}

// ---- several rigs in lock step (PtzIncrementalOptimizer::SolveBatch): create one handle per rig, solve them together, read
//      every rig's result back ------------------------------------------------------------------------------------------------
struct IncRig {
  std::vector<ImageFeatures> feats;
  std::vector<MatchesInfo> mis;
  std::vector<Camera> cams, out;
  std::unique_ptr<PtzIncrementalOptimizer> opt;
  std::unordered_set<long> reg;
  bool ok = false;
};

void* ptzh_inc_create(int32_t n_img, const int64_t* kp_ptr, const float* kp_xy, const int32_t* img_wh, int32_t n_pairs,
                      const int64_t* src, const int64_t* dst, const int64_t* match_ptr, const int32_t* q, const int32_t* t,
                      const double* H, const int32_t* h_valid, const double* confidence, const double* cam15, const int64_t* seeds,
                      int32_t n_seeds, int32_t max_iter)
{
  IncRig* r = new IncRig();
  BuildInputs(n_img, kp_ptr, kp_xy, img_wh, n_pairs, src, dst, match_ptr, q, t, cam15, r->feats, r->mis, r->cams);
  for (int p = 0; p < n_pairs; ++p) {
    for (int k = 0; k < 9; ++k) r->mis[p].H[k] = H[9 * p + k];
    r->mis[p].H_empty = h_valid[p] == 0;
    r->mis[p].confidence = confidence[p];
  }
  // (the tables were built for this optimizer only: moved in, not copied a second time)
  r->opt.reset(new PtzIncrementalOptimizer(PtzIncrementalOptimizer::TakeInputs{}, std::move(r->feats), std::move(r->mis), std::move(r->cams), max_iter));
  if (n_seeds > 0) r->opt->SetSeedImageId(std::vector<long>(seeds, seeds + n_seeds));
  return r;
}

// stats8: rounds, bundle-adjustment batches, problems in them, registration launches, queries in them, ms inside the batched
// bundle adjustments, ms inside the registration launches, wall ms of the whole call.  Returns the number of rigs solved.
int32_t ptzh_inc_solve_batch(void** handles, int32_t n, int32_t device_id, double* stats8)
{
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<PtzIncrementalOptimizer*> rigs;
  for (int32_t i = 0; i < n; ++i) {
    IncRig* r = static_cast<IncRig*>(handles[i]);
    r->opt->SetDevice(device_id);
    rigs.push_back(r->opt.get());
  }
  std::vector<std::vector<Camera>> cams;
  std::vector<std::unordered_set<long>> regs;
  PtzIncrementalOptimizer::BatchStats st;
  const std::vector<char> ok = PtzIncrementalOptimizer::SolveBatch(rigs, cams, regs, &st);
  int32_t solved = 0;
  for (int32_t i = 0; i < n; ++i) {
    IncRig* r = static_cast<IncRig*>(handles[i]);
    r->ok = ok[i] != 0;
    r->out = std::move(cams[i]);
    r->reg = std::move(regs[i]);
    solved += r->ok ? 1 : 0;
  }
  if (stats8) {
    stats8[0] = static_cast<double>(st.rounds); stats8[1] = static_cast<double>(st.ba_batches); stats8[2] = static_cast<double>(st.ba_problems);
    stats8[3] = static_cast<double>(st.krt_launches); stats8[4] = static_cast<double>(st.krt_queries); stats8[5] = st.ba_ms; stats8[6] = st.krt_ms;
    stats8[7] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return solved;
}

int32_t ptzh_inc_result(void* handle, double* cam15, int32_t* registered, int64_t* events, int32_t max_events, int64_t* lm_iterations,
                        int32_t* solved)
{
  IncRig* r = static_cast<IncRig*>(handle);
  const auto& ev = r->opt->events();
  const int32_t ne = static_cast<int32_t>(ev.size()) < max_events ? static_cast<int32_t>(ev.size()) : max_events;
  for (int32_t e = 0; e < ne; ++e) {
    events[4 * e] = ev[e].kind; events[4 * e + 1] = ev[e].a; events[4 * e + 2] = ev[e].b; events[4 * e + 3] = ev[e].success ? 1 : 0;
  }
  if (lm_iterations) *lm_iterations = r->opt->lm_iterations();
  if (solved) *solved = r->ok ? 1 : 0;
  if (!r->ok) return ne;
  const int n_img = static_cast<int>(r->opt->NumImages());
  for (int i = 0; i < n_img; ++i) {
    registered[i] = r->reg.count(i) ? 1 : 0;
    const std::vector<double> v = r->out[i].ToVector();
    memcpy(cam15 + 15 * i, v.data(), sizeof(double) * 15);
  }
  return ne;
}

void ptzh_inc_destroy(void* handle) { delete static_cast<IncRig*>(handle); }

// ---- file formats (data_io) for the tests: every probe answers with a JSON text (malloc'ed, free with ptzh_free) ------
static char* DupText(const std::string& s)
{
  char* p = static_cast<char*>(malloc(s.size() + 1));
  memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

// cmd = "load": a = image dir, b = feature dir  -> {ok, fnames, sizes, n_keypoints, first_keypoint, pairs:[{src,dst,n,H,H_empty,confidence}]}
// cmd = "annotation": a = annotation json, b = image dir (for the file names) -> {ok, pixels, pts3d}
// cmd = "rewrite": a = camera json in, b = camera json out (ReadFromJson -> SaveToJson)  -> {ok, names}
// cmd = "image_size": a = file -> {ok, width, height}
// cmd = "json": a = JSON text -> {ok, dump}
char* ptzh_io_probe(const char* cmd_c, const char* a_c, const char* b_c)
{
  const std::string cmd = cmd_c ? cmd_c : "", a = a_c ? a_c : "", b = b_c ? b_c : "";
  Json out = Json::Object();
  if (cmd == "image_size") {
    Size sz;
    const bool ok = ReadImageSize(a, sz);
    out["ok"] = Json::Bool(ok);
    out["width"] = Json::Int(sz.width);
    out["height"] = Json::Int(sz.height);
  }
  else if (cmd == "json") {
    Json v;
    std::string err;
    const bool ok = Json::Parse(a, v, &err);
    out["ok"] = Json::Bool(ok);
    out["dump"] = Json::String(ok ? v.dump(4) : err);
  }
  else if (cmd == "load") {
    std::vector<std::string> fnames;
    std::vector<ImageFeatures> features;
    std::vector<Size> sizes;
    const bool ok = LoadImgsAndFeatures(a, b, fnames, features, sizes);
    out["ok"] = Json::Bool(ok);
    Json jn = Json::Array(), js = Json::Array(), jk = Json::Array(), jf = Json::Array();
    for (size_t i = 0; i < fnames.size(); ++i) {
      jn.push_back(Json::String(fnames[i]));
      Json s = Json::Array(); s.push_back(Json::Int(sizes[i].width)); s.push_back(Json::Int(sizes[i].height)); js.push_back(s);
      jk.push_back(Json::Int(static_cast<long long>(features[i].keypoints.size())));
      Json f = Json::Array();
      if (!features[i].keypoints.empty()) { f.push_back(Json::Float(features[i].keypoints[0].pt.x)); f.push_back(Json::Float(features[i].keypoints[0].pt.y)); }
      jf.push_back(f);
    }
    out["fnames"] = jn; out["sizes"] = js; out["n_keypoints"] = jk; out["first_keypoint"] = jf;
    Json jp = Json::Array();
    if (ok) {
      std::vector<MatchesInfo> mis;
      LoadMatchesInfo(b + "/pairs_matches.txt", fnames, features, mis);
      out["table_cells"] = Json::Int(static_cast<long long>(mis.size()));
      for (const MatchesInfo& mi : mis) {
        if (mi.matches.empty()) continue;
        Json p = Json::Object();
        p["src"] = Json::Int(mi.src_img_idx); p["dst"] = Json::Int(mi.dst_img_idx); p["n"] = Json::Int(static_cast<long long>(mi.matches.size()));
        p["H"] = Json::FloatArray(std::vector<double>(mi.H.begin(), mi.H.end()));
        p["H_empty"] = Json::Bool(mi.H_empty);
        p["confidence"] = Json::Float(mi.confidence);
        p["first_match"] = Json::FloatArray({static_cast<double>(mi.matches[0].queryIdx), static_cast<double>(mi.matches[0].trainIdx)});
        jp.push_back(p);
      }
    }
    out["pairs"] = jp;
  }
  else if (cmd == "annotation") {
    std::vector<std::string> fnames;
    std::vector<ImageFeatures> features;
    std::vector<Size> sizes;
    LoadImgsAndFeatures(b, b, fnames, features, sizes);
    std::vector<std::vector<Point2f>> pixels;
    std::vector<std::vector<Point3d>> pts3d;
    const bool ok = LoadAnnotation(a, fnames, pixels, pts3d);
    out["ok"] = Json::Bool(ok);
    Json jp = Json::Array(), jq = Json::Array();
    for (size_t i = 0; i < pixels.size(); ++i) {
      Json pi = Json::Array(), qi = Json::Array();
      for (size_t k = 0; k < pixels[i].size(); ++k) {
        pi.push_back(Json::FloatArray({pixels[i][k].x, pixels[i][k].y}));
        qi.push_back(Json::FloatArray({pts3d[i][k].x, pts3d[i][k].y, pts3d[i][k].z}));
      }
      jp.push_back(pi); jq.push_back(qi);
    }
    out["pixels"] = jp; out["pts3d"] = jq;
  }
  else if (cmd == "rewrite") {
    std::vector<Camera> cams;
    std::vector<std::string> names;
    std::vector<std::vector<Point2f>> pixels;
    std::vector<std::vector<Point3d>> pts3d;
    std::vector<Size> sizes;
    bool ok = ReadFromJson(a, cams, names, pixels, pts3d, sizes);
    if (ok) ok = SaveToJson(cams, names, pixels, pts3d, b);
    out["ok"] = Json::Bool(ok);
    Json jn = Json::Array();
    for (const std::string& n : names) jn.push_back(Json::String(n));
    out["names"] = jn;
  }
  else out["ok"] = Json::Bool(false);
  return DupText(out.dump(4));
}

// cv::findHomography(src, dst, RANSAC, thresh, mask) replacement: returns 1 and fills H9 / mask[n], or 0 (empty H)
int32_t ptzh_find_homography(int32_t n, const float* src, const float* dst, double thresh, double* H9, unsigned char* mask)
{
  std::vector<Point2f> a, b;
  for (int i = 0; i < n; ++i) { a.emplace_back(src[2 * i], src[2 * i + 1]); b.emplace_back(dst[2 * i], dst[2 * i + 1]); }
  Mat33 H;
  std::vector<unsigned char> m;
  if (!FindHomographyRansac(a, b, thresh, H, &m)) return 0;
  for (int k = 0; k < 9; ++k) H9[k] = H[k];
  if (mask) memcpy(mask, m.data(), m.size());
  return 1;
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

// The same with KRTOptimizer::Add2d3dConstraints (world points) after Add2d2dConstraints; reproj2 = {Cal2d2dReprojError,
// Cal2d3dReprojError} after the solve.  order_swapped != 0 calls Add2d3dConstraints first (expected to throw): returns -2.
int32_t ptzh_krt_solve_2d3d(const double* cam_ref15, double* cam_cur15, int32_t n_match, const float* uv_ref, const float* uv_cur,
                            int32_t n_pt, const float* pts2d, const double* pts3d, int32_t max_iter, double max_reproj_error,
                            int32_t type, int32_t order_swapped, int32_t* num_iter, ptz_lm_summary* summary, double* reproj2)
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
  std::vector<Point2f> p2(n_pt);
  std::vector<Point3d> p3(n_pt);
  for (int i = 0; i < n_pt; ++i) {
    p2[i] = Point2f(pts2d[2 * i], pts2d[2 * i + 1]);
    p3[i] = Point3d(pts3d[3 * i], pts3d[3 * i + 1], pts3d[3 * i + 2]);
  }
  KRTOptimizer opt(max_iter, max_reproj_error, static_cast<KRTOptimizer::FACTOR_TYPE>(type));
  opt.SetInitParams(cur.K(), cur.R(), cur.t(), cur.dist());
  if (order_swapped) {
    try { opt.Add2d3dConstraints(p2, p3); }
    catch (const std::logic_error&) { return -2; }
    return -1;
  }
  opt.Add2d2dConstraints(ref, kr, kc, ms);
  opt.Add2d3dConstraints(p2, p3);
  Mat33 K, R; Vec3 t; Vec5 dist;
  const bool ok = opt.Solve(K, R, t, dist);
  if (num_iter) *num_iter = opt.num_iter_;
  if (summary) *summary = opt.summary();
  if (reproj2) {
    reproj2[0] = opt.Cal2d2dReprojError(ref, kr, kc, ms);
    reproj2[1] = opt.Cal2d3dReprojError(p2, p3);
  }
  if (ok) {
    const std::vector<double> v = Camera(K, R, t, dist).ToVector();
    memcpy(cam_cur15, v.data(), sizeof(double) * 15);
  }
  return ok ? 1 : 0;
}

}  // extern "C"
