#include "ptz_incremental_optimizer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <set>

#include <thread>

#include "../../include/ptz_calib_amd.h"
#include "device_batcher.h"
#include "ptzray_optimizer.h"

namespace ptzcalib {

long PtzIncrementalOptimizer::kMaxNumImages = 100000;
float PtzIncrementalOptimizer::kBaGlobalImagesRatio = 1.1f;

namespace {
struct ScopedMs {
  double& acc;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit ScopedMs(double& a) : acc(a) {}
  ~ScopedMs() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
// Images with a positive score, best first.  std::sort (not stable_sort) with the reference's comparator, so that ties
// fall the way they do there under the same standard library (ptz_incremental_optimizer.cc:192-203).
std::vector<long> RankByScore(const std::vector<float>& score)
{
  std::vector<long> order(score.size());
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) -> bool { return score[a] > score[b]; });
  std::vector<long> out;
  for (long id : order) {
    if (score[id] <= 0.0f) break;
    out.push_back(id);
  }
  return out;
}

// initial guess of an uncalibrated view: f = 1.2 max(w, h), principal point at the image centre (:322-329)
void SetDefaultIntrinsics(Camera& cam, const Size& size)
{
  constexpr double ratio = 1.2;
  const double focal = ratio * std::max(size.width, size.height);
  cam.K()[0] = cam.K()[4] = focal;
  cam.K()[2] = 0.5 * size.width;
  cam.K()[5] = 0.5 * size.height;
}

// rotation of view j predicted from registered view i through the pair homography: R_j = K_j^-1 H_ji K_i R_i (:391-394).
// The product is a rotation only up to scale and noise; it is orthonormalised when it is converted to a rotation
// vector (RodriguesInv), as cv::Rodrigues does in the reference.
Mat33 RotationFromHomography(const Mat33& K_j, const Mat33& H_j_i, const Camera& cam_i)
{
  return Mul(Mul(Mul(Inverse(K_j), H_j_i), cam_i.K()), cam_i.R());
}
}  // namespace

PtzIncrementalOptimizer::PtzIncrementalOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                                                 const std::vector<Camera>& cameras, int max_iter)
    : cameras_(cameras), features_(features), matches_info_(matches_info), max_iter_(max_iter)
{
}

PtzIncrementalOptimizer::PtzIncrementalOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                                                 const std::vector<Camera>& cameras, const std::vector<std::string>& names, int max_iter)
    : cameras_(cameras), features_(features), matches_info_(matches_info), names_(names), max_iter_(max_iter)
{
}

PtzIncrementalOptimizer::PtzIncrementalOptimizer(TakeInputs, std::vector<ImageFeatures>&& features, std::vector<MatchesInfo>&& matches_info,
                                                 std::vector<Camera>&& cameras, int max_iter)
    : cameras_(std::move(cameras)), features_(std::move(features)), matches_info_(std::move(matches_info)), max_iter_(max_iter)
{
}

PtzIncrementalOptimizer::~PtzIncrementalOptimizer()
{
  if (rig_) ptz_rig_destroy(rig_);
  if (match_table_) ptz_krt_table_destroy(match_table_);
}

void PtzIncrementalOptimizer::SetSeedImageId(const std::vector<long>& image_ids) { seed_image_ids_ = image_ids; }

bool PtzIncrementalOptimizer::CheckValid() const
{  // :140-146
  return !(features_.empty() || features_.size() != cameras_.size() || max_iter_ <= 0);
}

long PtzIncrementalOptimizer::ImagePairToPairId(long image_id1, long image_id2) const
{
  return image_id1 < image_id2 ? image_id1 * kMaxNumImages + image_id2 : image_id2 * kMaxNumImages + image_id1;
}

// Main loop (:39-131).
bool PtzIncrementalOptimizer::Solve(std::vector<Camera>& cameras, std::unordered_set<long>& reg_image_ids)
{
  if (!CheckValid()) return false;
  const bool trace_setup = getenv("PTZ_INC_TIMING") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double ts0 = now_ms();
  if (!tracks_) tracks_ = PTZRayOptimizer::BuildTracks(matches_info_);
  const double ts1 = now_ms();
  if (!rig_ && !tracks_->id.empty() && !(getenv("PTZ_IBA_VIEWS") && atoi(getenv("PTZ_IBA_VIEWS")) == 0)) {
    // the tracks go to the device once: image id and pixel of every view (PTZ_IBA_VIEWS=0: pack every bundle adjustment on the host)
    const SharedTracks& st = *tracks_;
    std::vector<float> uv(2 * st.img.size());
    bool ok = true;
    for (size_t e = 0; e < st.img.size() && ok; ++e) {
      ok = st.img[e] >= 0 && static_cast<size_t>(st.img[e]) < features_.size() && st.feat[e] >= 0 &&
           static_cast<size_t>(st.feat[e]) < features_[st.img[e]].keypoints.size();
      if (!ok) break;
      const Point2f pt = features_[st.img[e]].keypoints[st.feat[e]].pt;
      uv[2 * e] = pt.x; uv[2 * e + 1] = pt.y;
    }
    if (ok && ptz_rig_create(static_cast<int32_t>(features_.size()), static_cast<int32_t>(st.id.size()), st.ptr.data(), st.img.data(), uv.data(), device_id_,
                             &rig_) != PTZ_OK)
      rig_ = nullptr;  // (the bundle adjustments then pack on the host)
  }
  if (!match_table_ && !matches_info_.empty() && !(getenv("PTZ_IBA_MATCH_TABLE") && atoi(getenv("PTZ_IBA_MATCH_TABLE")) == 0)) {
    // ... and so do the matched pixels of every table entry: a registration attempt is then its entry's number and two cameras
    // (PTZ_IBA_MATCH_TABLE=0: the attempts' pixels are packed for every launch)
    std::vector<int64_t> ptr(matches_info_.size() + 1, 0);
    for (size_t e = 0; e < matches_info_.size(); ++e) ptr[e + 1] = ptr[e] + static_cast<int64_t>(matches_info_[e].matches.size());
    std::vector<float> uv_ref(2 * static_cast<size_t>(ptr.back())), uv_cur(2 * static_cast<size_t>(ptr.back()));
    bool ok = true;
    for (size_t e = 0; e < matches_info_.size() && ok; ++e) {
      const MatchesInfo& mi = matches_info_[e];
      if (mi.matches.empty()) continue;
      ok = mi.src_img_idx >= 0 && static_cast<size_t>(mi.src_img_idx) < features_.size() && mi.dst_img_idx >= 0 &&
           static_cast<size_t>(mi.dst_img_idx) < features_.size();
      if (!ok) break;
      const std::vector<KeyPoint>& kr = features_[mi.src_img_idx].keypoints;
      const std::vector<KeyPoint>& kc = features_[mi.dst_img_idx].keypoints;
      float* pr = uv_ref.data() + 2 * ptr[e];
      float* pc = uv_cur.data() + 2 * ptr[e];
      for (const DMatch& m : mi.matches) {
        if (m.queryIdx < 0 || static_cast<size_t>(m.queryIdx) >= kr.size() || m.trainIdx < 0 || static_cast<size_t>(m.trainIdx) >= kc.size()) { ok = false; break; }
        const Point2f a = kr[m.queryIdx].pt, b = kc[m.trainIdx].pt;
        *pr++ = a.x; *pr++ = a.y; *pc++ = b.x; *pc++ = b.y;
      }
    }
    if (ok && ptz_krt_table_create(static_cast<int32_t>(matches_info_.size()), ptr.data(), uv_ref.data(), uv_cur.data(), device_id_, &match_table_) != PTZ_OK)
      match_table_ = nullptr;  // (the attempts then pack their pixels)
  }
  if (trace_setup) fprintf(stderr, "[inc] setup: tracks %.2f ms, resident tracks + match table %.2f ms\n", ts1 - ts0, now_ms() - ts1);
  by_dst_.assign(features_.size(), {});
  for (size_t e = 0; e < matches_info_.size(); ++e) {
    const MatchesInfo& mi = matches_info_[e];
    if (!mi.H_empty && mi.dst_img_idx >= 0 && static_cast<size_t>(mi.dst_img_idx) < by_dst_.size()) by_dst_[mi.dst_img_idx].push_back(e);
  }

  const int kInitNumTrials = 50;
  for (int num_trials = 0; num_trials < kInitNumTrials; ++num_trials) {
    long image_id1, image_id2;
    if (!FindInitialImagePair(image_id1, image_id2)) return false;
    const bool seeded = RegisterInitialImagePair(image_id1, image_id2);
    events_.push_back({Event::kInitPair, image_id1, image_id2, seeded});
    if (!seeded) continue;

    AdjustGlobalBundle();
    size_t ba_prev_num_reg_images = NumRegImages();

    bool reg_next_success = true;
    while (reg_next_success) {
      reg_next_success = false;
      const std::vector<long> next_image_ids = FindNextImages();
      if (next_image_ids.empty()) break;

      // The walk below ends at the next successful bundle adjustment, i.e. after about 0.1 x (model size) registrations:
      // solve the attempts of that many leading candidates (plus a little slack for failures) in one launch now.
      attempt_cache_.clear();
      {
        const float need = kBaGlobalImagesRatio * static_cast<float>(ba_prev_num_reg_images) - static_cast<float>(NumRegImages());
        const size_t ahead = static_cast<size_t>(std::max(1.0f, std::ceil(need))) + 2;
        SpeculateRegistrations(next_image_ids, 0, ahead);
      }
      for (size_t reg_trial = 0; reg_trial < next_image_ids.size(); ++reg_trial) {
        const long image_id = next_image_ids[reg_trial];
        reg_next_success = RegisterNextImage(image_id);
        if (reg_next_success && static_cast<float>(NumRegImages()) >= kBaGlobalImagesRatio * static_cast<float>(ba_prev_num_reg_images)) {
          if (AdjustGlobalBundle()) {
            ba_prev_num_reg_images = NumRegImages();
            break;  // re-rank the remaining images against the refined model
          }
          reg_image_ids_.erase(image_id);  // only the newest image is withdrawn, as in the reference (:99)
          reg_next_success = false;
        }
        if (!reg_next_success) {
          // a seed that cannot grow to three images within 30 attempts is abandoned (:105-113); the outer loop then
          // runs the closing bundle adjustment and returns -- the reference does not go back for another seed either
          const size_t kMinNumInitialRegTrials = 30;
          const size_t kMinModelSize = 3;
          if (reg_trial >= kMinNumInitialRegTrials && NumRegImages() < kMinModelSize) break;
        }
      }
    }

    AdjustGlobalBundle();
    reg_image_ids = reg_image_ids_;
    cameras = cameras_;
    return true;
  }
  return false;  // 50 seeds failed (the reference falls off the end of the function here)
}

// N rigs side by side (run_ptzba_synthetic.sh:4-13 runs them one after the other): every optimizer's Solve() on a host thread
// of its own, its device calls through one DeviceBatcher, so that each round of the lock step is ONE batched bundle
// adjustment and ONE registration launch for all rigs.  Decisions and results are those of N solo runs.
std::vector<char> PtzIncrementalOptimizer::SolveBatch(const std::vector<PtzIncrementalOptimizer*>& rigs, std::vector<std::vector<Camera>>& cameras,
                                                      std::vector<std::unordered_set<long>>& reg_image_ids, BatchStats* stats)
{
  const size_t n = rigs.size();
  cameras.resize(n);
  reg_image_ids.resize(n);
  std::vector<char> ok(n, 0);
  // ONE lock step over all rigs: with the rigs' tracks resident on the device a round's bundle adjustments are one device-built
  // batch (~19 problems for 64 rigs of 200 views) and its registration attempts one launch beside it, and the host work of a round
  // is what the rigs do themselves, in parallel.  (Round 3 packed every problem on the host and had to split the rigs into
  // cohorts of four to overlap that work: 1.9 problems per batch.)  PTZ_IBA_COHORTS deals the rigs to several independent lock
  // steps (rig i -> cohort i % K); measured, 64 rigs x 200 views, ms inside this call: 1: 444, 2: 408, 3: 396, 16: 610 -- a second
  // cohort overlaps one's host work with the other's device rounds at half the batch size.  Results do not depend on it.
  size_t n_cohorts = 1;
  if (const char* e = getenv("PTZ_IBA_COHORTS")) n_cohorts = static_cast<size_t>(std::max(1, atoi(e)));
  n_cohorts = std::max<size_t>(1, std::min(n_cohorts, n));
  std::vector<std::unique_ptr<DeviceBatcher>> batchers;
  for (size_t c = 0; c < n_cohorts; ++c) batchers.emplace_back(new DeviceBatcher(static_cast<int>((n - c + n_cohorts - 1) / n_cohorts)));
  std::vector<std::thread> th;
  th.reserve(n);
  for (size_t i = 0; i < n; ++i)
    th.emplace_back([&, i] {
      DeviceBatcher* batcher = batchers[i % n_cohorts].get();
      DeviceBatcher::Scope scope(batcher);
      ok[i] = rigs[i]->Solve(cameras[i], reg_image_ids[i]) ? 1 : 0;
      batcher->ClientDone();
    });
  for (std::thread& t : th) t.join();
  if (stats) {
    *stats = BatchStats();
    for (const auto& bp : batchers) {
      const DeviceBatcher::Stats s = bp->stats();
      stats->rounds += s.rounds; stats->ba_batches += s.ba_batches; stats->ba_problems += s.ba_problems;
      stats->krt_launches += s.krt_launches; stats->krt_queries += s.krt_queries; stats->ba_ms += s.ba_ms; stats->krt_ms += s.krt_ms;
    }
  }
  return ok;
}

bool PtzIncrementalOptimizer::FindInitialImagePair(long& image_id1, long& image_id2)
{  // :148-176
  const std::vector<long> firsts = seed_image_ids_.empty() ? FindFirstInitialImage() : seed_image_ids_;
  for (long first : firsts) {
    image_id1 = first;
    for (long second : FindSecondInitialImage(first)) {
      image_id2 = second;
      const long pair_id = ImagePairToPairId(first, second);
      if (init_image_pairs_.count(pair_id) > 0) continue;
      init_image_pairs_.insert(pair_id);
      return true;
    }
  }
  image_id1 = image_id2 = std::numeric_limits<long>::max();
  return false;
}

std::vector<long> PtzIncrementalOptimizer::FindFirstInitialImage() const
{  // :178-204: total matching confidence of every image (float accumulation)
  ScopedMs tm(timing_ms_[0]);
  std::vector<float> score(features_.size(), 0.0f);
  for (const MatchesInfo& mi : matches_info_) {
    const float confidence = static_cast<float>(mi.confidence);
    score[mi.src_img_idx] += confidence;
    score[mi.dst_img_idx] += confidence;
  }
  return RankByScore(score);
}

std::vector<long> PtzIncrementalOptimizer::FindSecondInitialImage(long image_id1) const
{  // :206-244: partners of image_id1 with a mean match displacement of at least 50 px
  ScopedMs tm(timing_ms_[0]);
  std::vector<float> score(features_.size(), 0.0f);
  const float kMinPixelDiff = 50;
  for (const MatchesInfo& mi : matches_info_) {
    const long src = mi.src_img_idx, dst = mi.dst_img_idx;
    if (mi.matches.empty()) continue;
    const bool is_src = image_id1 == src, is_dst = image_id1 == dst;
    if (is_src == is_dst) continue;  // unrelated pair, or the image paired with itself
    if (CalPixelDiff(src, dst, mi.matches) < kMinPixelDiff) continue;
    score[is_src ? dst : src] += static_cast<float>(mi.confidence);
  }
  return RankByScore(score);
}

std::vector<long> PtzIncrementalOptimizer::FindNextImages() const
{  // :246-296: unregistered images next to the model, by total confidence towards registered images
  ScopedMs tm(timing_ms_[0]);
  std::vector<float> score(features_.size(), 0.0f);
  const size_t kMaxRegTrials = 4;
  // (per-image flags first: the loop below asks four questions per table entry, ~8 500 entries per 200-view rig, every cycle)
  std::vector<char> is_exhausted(features_.size(), 0), is_reg(features_.size(), 0);
  for (const auto& kv : num_reg_trials_)
    if (kv.second > kMaxRegTrials && kv.first >= 0 && static_cast<size_t>(kv.first) < is_exhausted.size()) is_exhausted[kv.first] = 1;
  for (long id : reg_image_ids_)
    if (id >= 0 && static_cast<size_t>(id) < is_reg.size()) is_reg[id] = 1;
  auto flag = [](const std::vector<char>& f, long id) { return id >= 0 && static_cast<size_t>(id) < f.size() && f[id] != 0; };
  for (const MatchesInfo& mi : matches_info_) {
    const long src = mi.src_img_idx, dst = mi.dst_img_idx;
    if (src == dst || mi.H_empty) continue;
    if (flag(is_exhausted, src) || flag(is_exhausted, dst)) continue;
    const bool src_in = flag(is_reg, src), dst_in = flag(is_reg, dst);
    if (src_in == dst_in) continue;  // both registered, or neither
    score[src_in ? dst : src] += static_cast<float>(mi.confidence);
  }
  return RankByScore(score);
}

float PtzIncrementalOptimizer::CalPixelDiff(long image_id1, long image_id2, const std::vector<DMatch>& matches) const
{  // :298-312: float accumulator, each distance taken in double (cv::norm of a Point2f)
  float total = 0.0f;
  for (const DMatch& m : matches) {
    const Point2f a = features_[image_id1].keypoints[m.queryIdx].pt, b = features_[image_id2].keypoints[m.trainIdx].pt;
    const float dx = a.x - b.x, dy = a.y - b.y;
    total += std::sqrt(static_cast<double>(dx) * dx + static_cast<double>(dy) * dy);
  }
  return total * 1.0f / matches.size();
}

void PtzIncrementalOptimizer::SetInitialImagePairParameters(long image_id1, long image_id2)
{  // :322-352
  SetDefaultIntrinsics(cameras_[image_id1], features_[image_id1].img_size);
  cameras_[image_id1].R() = Eye3();
  SetDefaultIntrinsics(cameras_[image_id2], features_[image_id2].img_size);
  for (const MatchesInfo& mi : matches_info_) {
    // only the table entry stored in this direction is used (:344); a pair listed the other way round leaves R_2 as it was
    if (mi.src_img_idx == image_id1 && mi.dst_img_idx == image_id2) {
      cameras_[image_id2].R() = RotationFromHomography(cameras_[image_id2].K(), mi.H, cameras_[image_id1]);
      break;
    }
  }
}

bool PtzIncrementalOptimizer::RunBundle(const std::unordered_set<long>& ids)
{
  ScopedMs tm(timing_ms_[1]);
  if (getenv("PTZ_INC_DEBUG") && atoi(getenv("PTZ_INC_DEBUG")) > 1)
    for (long id : std::set<long>(ids.begin(), ids.end())) {
      const std::vector<double> v = cameras_[id].ToVector();
      fprintf(stderr, "[inc]   in  cam %ld: %.17g %.17g %.17g %.17g | %.17g %.17g %.17g\n", id, v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    }
  PTZRayOptimizer optimizer(PTZRayOptimizer::Borrow{}, features_, matches_info_, cameras_, ids, max_iter_, PTZRay);
  optimizer.UseTracks(tracks_);
  optimizer.UseRig(rig_);
  optimizer.SetDevice(device_id_);
  const bool ok = optimizer.Solve(cameras_);
  timing_ms_[2] += optimizer.device_ms();
  lm_iterations_ += optimizer.summary().num_iterations;
  if (getenv("PTZ_INC_DEBUG") && atoi(getenv("PTZ_INC_DEBUG")) > 1)
    for (long id : std::set<long>(ids.begin(), ids.end())) {
      const std::vector<double> v = cameras_[id].ToVector();
      fprintf(stderr, "[inc]   out cam %ld: %.17g %.17g %.17g %.17g | %.17g %.17g %.17g\n", id, v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    }
  if (getenv("PTZ_INC_DEBUG")) {
    double cs = 0;
    for (long id : ids) { const std::vector<double> v = cameras_[id].ToVector(); for (int k = 0; k < 7; ++k) cs += v[k] * (k + 1); }
    fprintf(stderr, "[inc] BA n=%zu it=%d ok=%d init_cost=%.17g final_cost=%.17g cams=%.17g\n", ids.size(), optimizer.summary().num_iterations,
            (int)ok, optimizer.summary().initial_cost, optimizer.summary().final_cost, cs);
  }
  events_.push_back({Event::kGlobalBA, static_cast<long>(ids.size()), optimizer.summary().num_iterations, ok});
  return ok;
}

bool PtzIncrementalOptimizer::RegisterInitialImagePair(long image_id1, long image_id2)
{  // :354-375
  num_reg_trials_[image_id1] += 1;
  num_reg_trials_[image_id2] += 1;
  init_image_pairs_.insert(ImagePairToPairId(image_id1, image_id2));
  SetInitialImagePairParameters(image_id1, image_id2);
  const bool ok = RunBundle({image_id1, image_id2});
  if (ok) {
    reg_image_ids_.insert(image_id1);
    reg_image_ids_.insert(image_id2);
  }
  return ok;
}

bool PtzIncrementalOptimizer::AdjustGlobalBundle() { return RunBundle(reg_image_ids_); }  // :420-440

// Batched KRT solves of the given table entries (each: registered reference mi.src -> image mi.dst); results go to the
// cache.  One ptz_krt_solve_batch launch for all of them (queries are independent: one wave each).
void PtzIncrementalOptimizer::SolveAttempts(const std::vector<const MatchesInfo*>& todo)
{
  if (todo.empty()) return;
  const size_t n = todo.size();
  const bool resident = match_table_ != nullptr;
  std::vector<int64_t> match_ptr(n + 1, 0);
  size_t total = 0;
  if (!resident) for (const MatchesInfo* mi : todo) total += mi->matches.size();
  std::vector<float> uv_ref, uv_cur;
  uv_ref.reserve(2 * total); uv_cur.reserve(2 * total);
  std::vector<ptz_krt_attempt> attempts(resident ? n : 0);
  std::vector<double> cam_ref(15 * n), cam_cur(15 * n);
  std::vector<Attempt> res(n);
  // (a registered camera is the reference of many attempts of one launch: its 15-vector -- an SVD inside -- is made once)
  std::unordered_map<long, std::vector<double>> ref_vec;
  for (size_t q = 0; q < n; ++q) {
    const MatchesInfo& mi = *todo[q];
    const Camera& cam_i = cameras_[mi.src_img_idx];
    const Camera& cam_j = cameras_[mi.dst_img_idx];
    const Camera init(cam_i.K(), RotationFromHomography(cam_i.K(), mi.H, cam_i), cam_j.t(), cam_j.dist());  // K_j := K_i (:392)
    res[q].init_K = init.K();
    res[q].init_R = init.R();
    auto rv = ref_vec.find(mi.src_img_idx);
    if (rv == ref_vec.end()) rv = ref_vec.emplace(mi.src_img_idx, cam_i.ToVector()).first;
    const std::vector<double>& vr = rv->second;
    const std::vector<double> vc = init.ToVector();
    std::copy(vr.begin(), vr.end(), cam_ref.begin() + 15 * q);
    std::copy(vc.begin(), vc.end(), cam_cur.begin() + 15 * q);
    if (resident) {  // the entry's pixels are on the device already
      attempts[q].table = match_table_;
      attempts[q].entry = static_cast<int32_t>(todo[q] - matches_info_.data());
      continue;
    }
    const std::vector<KeyPoint>& kr = features_[mi.src_img_idx].keypoints;
    const std::vector<KeyPoint>& kc = features_[mi.dst_img_idx].keypoints;
    for (const DMatch& m : mi.matches) {
      const Point2f a = kr[m.queryIdx].pt, b = kc[m.trainIdx].pt;
      uv_ref.push_back(a.x); uv_ref.push_back(a.y);
      uv_cur.push_back(b.x); uv_cur.push_back(b.y);
    }
    match_ptr[q + 1] = static_cast<int64_t>(uv_ref.size() / 2);
  }
  ptz_lm_options opt;
  ptz_lm_options_default(&opt);
  opt.max_num_iterations = 100;  // :396
  opt.device_id = device_id_;
  std::vector<ptz_lm_summary> summaries(n);
  std::vector<int32_t> accepted(n, 0);
  const auto t_dev = std::chrono::steady_clock::now();
  const int32_t rc = resident ? DeviceKrtSolveAttempts(static_cast<int32_t>(n), attempts.data(), cam_ref.data(), cam_cur.data(), PTZ_KRT_F,
                                                       /*max_reproj_error=*/100.0, &opt, summaries.data(), accepted.data(), nullptr)
                              : DeviceKrtSolveBatch(static_cast<int32_t>(n), match_ptr.data(), uv_ref.data(), uv_cur.data(), cam_ref.data(),
                                                    cam_cur.data(), PTZ_KRT_F, /*max_reproj_error=*/100.0, &opt, summaries.data(),
                                                    accepted.data(), nullptr);
  timing_ms_[4] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dev).count();
  for (size_t q = 0; q < n; ++q) {
    res[q].accepted = rc == PTZ_OK && accepted[q] != 0;
    if (res[q].accepted) std::copy(cam_cur.begin() + 15 * q, cam_cur.begin() + 15 * (q + 1), res[q].refined.begin());
    attempt_cache_[todo[q]] = res[q];
  }
}

void PtzIncrementalOptimizer::SpeculateRegistrations(const std::vector<long>& next_image_ids, size_t first, size_t count)
{
  ScopedMs tm(timing_ms_[3]);
  std::vector<const MatchesInfo*> todo;
  for (size_t r = first; r < next_image_ids.size() && r < first + count; ++r)
    for (size_t e : by_dst_[next_image_ids[r]]) {
      const MatchesInfo& mi = matches_info_[e];
      if (IsRegistered(mi.src_img_idx) && !attempt_cache_.count(&mi)) todo.push_back(&mi);
    }
  SolveAttempts(todo);
}

bool PtzIncrementalOptimizer::RegisterNextImage(long image_id)
{  // :377-418
  ScopedMs tm(timing_ms_[3]);
  num_reg_trials_[image_id] += 1;
  // every table entry (registered i -> image_id) with a homography is one attempt, tried in table order until one is accepted
  std::vector<const MatchesInfo*> attempts;
  for (size_t e : by_dst_[image_id])
    if (IsRegistered(matches_info_[e].src_img_idx)) attempts.push_back(&matches_info_[e]);
  if (attempts.empty()) {
    events_.push_back({Event::kRegister, image_id, -1, false});
    return false;
  }
  // attempts solved ahead of time are looked up; the ones that are not (their reference was registered after the
  // speculation) and come before the first accepted one are solved now, together
  std::vector<const MatchesInfo*> todo;
  for (const MatchesInfo* mi : attempts) {
    const auto it = attempt_cache_.find(mi);
    if (it == attempt_cache_.end()) todo.push_back(mi);
    else if (it->second.accepted) break;
  }
  SolveAttempts(todo);
  Camera& cam_j = cameras_[image_id];
  for (const MatchesInfo* mi : attempts) {
    const Attempt& at = attempt_cache_.at(mi);
    if (!at.accepted) continue;
    Camera refined;
    refined.FromVector(std::vector<double>(at.refined.begin(), at.refined.end()));
    cam_j.K() = refined.K();  // t and dist are not taken over (:406-407)
    cam_j.R() = refined.R();
    reg_image_ids_.insert(image_id);
    events_.push_back({Event::kRegister, image_id, mi->src_img_idx, true});
    return true;
  }
  // all attempts failed: the camera keeps the initial guess of the last attempt (:392-394 run before every solve)
  const Attempt& last = attempt_cache_.at(attempts.back());
  cam_j.K() = last.init_K;
  cam_j.R() = last.init_R;
  events_.push_back({Event::kRegister, image_id, -1, false});
  return false;
}

}  // namespace ptzcalib
