// ptz_incremental_optimizer.h -- PTZ-IBA orchestration with the reference's public interface
// (src/core/ptz_incremental_optimizer.h:21-44): seed pair, view-by-view registration, global bundle adjustment
// whenever the model grew by kBaGlobalImagesRatio.  Every solve runs on the MI355X library:
//  * RegisterNextImage gathers ALL (registered reference -> image) attempts of one image into a single
//    ptz_krt_solve_batch launch and takes the first accepted one in table order, which is exactly what the
//    reference's sequential loop with early exit returns (ptz_incremental_optimizer.cc:383-415) because the
//    attempts do not depend on each other;
//  * the attempts of the images that the current ranking will try next are solved ahead of time in one launch
//    (SpeculateRegistrations): an attempt depends only on its reference camera, its match list and its homography, none
//    of which changes before the next bundle adjustment, so looking its result up later is the same as solving it then;
//  * the feature tracks are built once per match table and shared by all bundle adjustments (the reference rebuilds
//    them from the N^2 table inside every PTZRayOptimizer::Solve, ptzray_optimizer.cc:537-552), and the feature /
//    match tables are borrowed instead of deep-copied.
#pragma once

#include <array>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "ptzray_optimizer.h"
#include "tracks.h"
#include "types.h"

namespace ptzcalib {

class PtzIncrementalOptimizer {
 public:
  PtzIncrementalOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                          const std::vector<Camera>& cameras, int max_iter);
  PtzIncrementalOptimizer(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                          const std::vector<Camera>& cameras, const std::vector<std::string>& names, int max_iter);

  // the same with the inputs MOVED in (a caller that built them for this optimizer only: the C API of the batch driver) -- the
  // reference's constructors deep-copy (ptz_incremental_optimizer.h:31-36), which for a 200-view rig is 12 MB of match lists
  struct TakeInputs {};
  PtzIncrementalOptimizer(TakeInputs, std::vector<ImageFeatures>&& features, std::vector<MatchesInfo>&& matches_info,
                          std::vector<Camera>&& cameras, int max_iter);
  size_t NumImages() const { return features_.size(); }
  ~PtzIncrementalOptimizer();
  PtzIncrementalOptimizer(const PtzIncrementalOptimizer&) = delete;
  PtzIncrementalOptimizer& operator=(const PtzIncrementalOptimizer&) = delete;
  bool Solve(std::vector<Camera>& cameras, std::unordered_set<long>& reg_image_ids);
  void SetSeedImageId(const std::vector<long>& image_ids);

  static long kMaxNumImages;
  static float kBaGlobalImagesRatio;

  // extras (not in the reference): the sequence of decisions, for tests and for throughput accounting
  struct Event {
    enum Kind { kInitPair = 0, kRegister = 1, kGlobalBA = 2 };
    int kind;
    long a, b;        // kInitPair: the two images; kRegister: image, reference it was registered against (-1 = none);
                      // kGlobalBA: number of registered images, LM iterations
    bool success;
  };
  const std::vector<Event>& events() const { return events_; }
  long lm_iterations() const { return lm_iterations_; }  // summed over all bundle adjustments
  // wall-clock split of Solve in milliseconds: [0] ranking, [1] bundle adjustments (pack + device), [2] device part of [1],
  // [3] registrations (pack + device), [4] device part of [3]
  const double* timing_ms() const { return timing_ms_; }
  void SetDevice(int device_id) { device_id_ = device_id; }
  // extras: several rigs in lock step on one GPU (device_batcher.h): rigs[i]->Solve(cameras[i], reg_image_ids[i]) for every i,
  // each on its own host thread, their bundle adjustments and registration attempts batched round by round.  Returns the
  // Solve() results.  Decisions, cameras and events of every rig are those of its solo run.
  struct BatchStats { long rounds = 0, ba_batches = 0, ba_problems = 0, krt_launches = 0, krt_queries = 0; double ba_ms = 0, krt_ms = 0; };
  static std::vector<char> SolveBatch(const std::vector<PtzIncrementalOptimizer*>& rigs, std::vector<std::vector<Camera>>& cameras,
                                      std::vector<std::unordered_set<long>>& reg_image_ids, BatchStats* stats = nullptr);

 private:
  bool CheckValid() const;
  bool FindInitialImagePair(long& image_id1, long& image_id2);
  std::vector<long> FindFirstInitialImage() const;
  std::vector<long> FindSecondInitialImage(long image_id1) const;
  std::vector<long> FindNextImages() const;
  float CalPixelDiff(long image_id1, long image_id2, const std::vector<DMatch>& matches) const;
  bool RegisterInitialImagePair(long image_id1, long image_id2);
  bool RegisterNextImage(long image_id);
  // one (registered reference -> image) attempt of RegisterNextImage and what came out of it
  struct Attempt {
    bool accepted = false;
    std::array<double, 15> refined{};  // world-frame camera vector when accepted
    Mat33 init_K = Eye3(), init_R = Eye3();  // what the reference leaves in cameras_[image] when the attempt fails (:392-394)
  };
  void SolveAttempts(const std::vector<const MatchesInfo*>& todo);  // results into attempt_cache_
  void SpeculateRegistrations(const std::vector<long>& next_image_ids, size_t first, size_t count);
  void SetInitialImagePairParameters(long image_id1, long image_id2);
  bool AdjustGlobalBundle();
  bool RunBundle(const std::unordered_set<long>& ids);
  long ImagePairToPairId(long image_id1, long image_id2) const;
  size_t NumRegImages() const { return reg_image_ids_.size(); }
  bool IsRegistered(long id) const { return reg_image_ids_.count(id) != 0; }

  std::vector<Camera> cameras_;
  std::vector<ImageFeatures> features_;
  std::vector<MatchesInfo> matches_info_;
  std::vector<std::string> names_;
  int max_iter_;
  int device_id_ = 0;
  std::unordered_set<long> init_image_pairs_;             // every pair is tried once as a seed
  std::unordered_map<long, size_t> num_reg_trials_;       // registration attempts per image
  std::unordered_set<long> reg_image_ids_;
  std::vector<long> seed_image_ids_;
  std::shared_ptr<const SharedTracks> tracks_;
  // the same tracks resident on the device (ptz_rig_create, once per rig): every bundle adjustment of the run is a view of them
  // ("keep tracks and packed observations resident and grow them instead of rebuilding")
  ptz_rig* rig_ = nullptr;
  ptz_krt_table* match_table_ = nullptr;  // the table entries' matched pixels, resident on the device (entry = index into matches_info_)
  std::vector<std::vector<size_t>> by_dst_;  // table entries (indices into matches_info_, ascending) per destination image
  std::unordered_map<const MatchesInfo*, Attempt> attempt_cache_;  // valid until the next successful bundle adjustment
  std::vector<Event> events_;
  long lm_iterations_ = 0;
  mutable double timing_ms_[5] = {0, 0, 0, 0, 0};
};

}  // namespace ptzcalib
