// device_batcher.h -- several optimizers in lock step on one GPU.
//
// The reference calibrates its scenes one process after the other (run_ptzba_synthetic.sh:4-13); inside a scene
// PtzIncrementalOptimizer::Solve (src/core/ptz_incremental_optimizer.cc:39-126) is a strictly sequential chain of ~40 bundle
// adjustments and ~200 single-view registrations, each far too small to fill a GPU.  Scenes never interact, so N of them can
// walk that chain side by side: every optimizer runs UNCHANGED on a host thread of its own, and the two device calls it makes
// (ptz_ba_solve, ptz_krt_solve_batch) go through this rendezvous instead of straight to the library.  When every optimizer
// that is still running has a call pending, the pending bundle adjustments become ONE ptz_ba_batch and the pending
// registration attempts ONE ptz_krt_solve_batch launch; the results are handed back and the threads go on.  A scene inside a
// batch has the bits of its solo solve (the library's reductions are fixed-order, scenes never share a sum), so every decision
// of every optimizer is the one it would have taken alone.  PtzIncrementalOptimizer::SolveBatch runs several such lock steps
// (cohorts of four rigs) side by side, each with a DeviceBatcher of its own.
#pragma once

#include <semaphore.h>

#include <cstdint>
#include <mutex>
#include <vector>

#include "../../include/ptz_calib_amd.h"

namespace ptzcalib {

class DeviceBatcher {
 public:
  explicit DeviceBatcher(int n_clients);
  // what a client thread calls in place of the C-ABI entry points of the same signature (blocks until its round has run)
  int32_t BaSolve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt, ptz_lm_summary* summary);
  int32_t KrtSolveBatch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, const double* cam_ref,
                        double* cam_cur, int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                        ptz_lm_summary* summaries, int32_t* accepted, double* device_ms);
  // registration attempts over device-resident match tables (ptz_krt_solve_attempts): attempts [n_query], cameras as above
  int32_t KrtSolveAttempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur, int32_t factor_type,
                           double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted, double* device_ms);
  // a bundle adjustment over a VIEW of device-resident tracks (ptz_ba_batch_create_views): cam [15 n_cam] in / out, rkinv [9 n_cam]
  int32_t BaSolveView(const ptz_rig_view* view, int32_t factor_type, double* cam, const double* rkinv, const ptz_lm_options* opt,
                      ptz_lm_summary* summary);
  void ClientDone();  // the calling client makes no further calls (its optimizer has returned)

  struct Stats {
    long rounds = 0, ba_batches = 0, ba_problems = 0, krt_launches = 0, krt_queries = 0;
    double ba_ms = 0, krt_ms = 0;  // wall time inside the library calls
  };
  Stats stats() const { return stats_; }

  // the batcher of the calling thread (nullptr: calls go straight to the library)
  static DeviceBatcher* Current();
  struct Scope {
    explicit Scope(DeviceBatcher* b);
    ~Scope();
    DeviceBatcher* prev;
  };

 private:
  struct BaReq {
    const ptz_ba_problem* p; double *cam, *ray, *tlw; const ptz_lm_options* opt; ptz_lm_summary* summary; int32_t rc;
  };
  struct KrtReq {
    int32_t n_query; const int64_t* match_ptr; const float *uv_ref, *uv_cur; const double* cam_ref; double* cam_cur;
    int32_t factor_type; double max_reproj_error; const ptz_lm_options* opt; ptz_lm_summary* summaries; int32_t* accepted;
    double* device_ms; int32_t rc;
    const ptz_krt_attempt* attempts;  // not null: the queries are entries of resident tables (match_ptr / uv_* unused)
  };
  struct BavReq {
    const ptz_rig_view* view; int32_t factor_type; double* cam; const double* rkinv; const ptz_lm_options* opt; ptz_lm_summary* summary; int32_t rc;
  };
  void RunBaViews(std::vector<BavReq*>& reqs);
  static void RunBaViewBatch(std::vector<BavReq*>& reqs);  // one ptz_ba_batch_create_views for all of them
  std::vector<BavReq*> bav_;
  void Arrive(std::unique_lock<std::mutex>& lk);  // called with the request already queued; returns with `lk` released
  void ReleaseWaiters(std::unique_lock<std::mutex>& lk);  // after a round: wakes every waiting client (releases `lk` first)
  void RunRound();                                // executes and clears the queues (lock held: every other client is waiting)
  void RunBa(std::vector<BaReq*>& reqs);
  static void RunBaBatch(std::vector<BaReq*>& reqs);  // one ptz_ba_batch for all of them
  void RunKrt(std::vector<KrtReq*>& reqs);

  std::mutex mu_;
  std::vector<sem_t*> waiters_;  // the clients asleep until the round has run (one semaphore each, on their stacks)
  int active_, waiting_ = 0;
  uint64_t generation_ = 0;
  std::vector<BaReq*> ba_;
  std::vector<KrtReq*> krt_;
  Stats stats_;
  double trace_late_[3] = {0, 0, 0}, trace_first_ = 0;  // PTZ_BATCHER_TRACE: latest arrival of a round by request type
  double last_end_ms_ = 0;  // PTZ_BATCHER_TRACE: when the previous round ended
};

// The device calls of the optimizer classes: through the calling thread's DeviceBatcher when it has one.
int32_t DeviceBaSolve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt, ptz_lm_summary* summary);
int32_t DeviceBaSolveView(const ptz_rig_view* view, int32_t factor_type, double* cam, const double* rkinv, const ptz_lm_options* opt,
                          ptz_lm_summary* summary);
int32_t DeviceKrtSolveBatch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, const double* cam_ref,
                            double* cam_cur, int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                            ptz_lm_summary* summaries, int32_t* accepted, double* device_ms);
int32_t DeviceKrtSolveAttempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur, int32_t factor_type,
                               double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted, double* device_ms);

}  // namespace ptzcalib
