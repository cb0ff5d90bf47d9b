#include "device_batcher.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace ptzcalib {

namespace {
thread_local DeviceBatcher* t_batcher = nullptr;
double NowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool Trace() { static const bool on = getenv("PTZ_BATCHER_TRACE") != nullptr; return on; }  // per-call timings on stderr
bool SameOptions(const ptz_lm_options* a, const ptz_lm_options* b) { return memcmp(a, b, sizeof(ptz_lm_options)) == 0; }
}  // namespace

DeviceBatcher* DeviceBatcher::Current() { return t_batcher; }
DeviceBatcher::Scope::Scope(DeviceBatcher* b) : prev(t_batcher) { t_batcher = b; }
DeviceBatcher::Scope::~Scope() { t_batcher = prev; }

DeviceBatcher::DeviceBatcher(int n_clients) : active_(n_clients) {}

// Waiting clients sleep on a semaphore of their own (on their stack) and are posted one by one after the round: the sixty-odd
// clients of a lock step then wake side by side.  (They used to share one condition variable: every woken thread had to take the
// batcher's mutex in turn before it could leave -- measured, 64 rigs: ~0.6 ms from the end of a round until the last client was
// back at work, a quarter of the round.)
void DeviceBatcher::Arrive(std::unique_lock<std::mutex>& lk)
{
  ++waiting_;
  if (waiting_ == active_) {  // the last one to arrive runs the round for everybody
    RunRound();
    ReleaseWaiters(lk);
    return;
  }
  sem_t sem;
  sem_init(&sem, 0, 0);
  waiters_.push_back(&sem);
  lk.unlock();
  while (sem_wait(&sem) != 0) {}  // (EINTR)
  sem_destroy(&sem);
}

void DeviceBatcher::ReleaseWaiters(std::unique_lock<std::mutex>& lk)
{
  waiting_ = 0;
  ++generation_;
  std::vector<sem_t*> ws;
  ws.swap(waiters_);
  lk.unlock();  // (the results are in the clients' request blocks; nothing below touches the batcher)
  for (sem_t* s : ws) sem_post(s);
}

void DeviceBatcher::ClientDone()
{
  std::unique_lock<std::mutex> lk(mu_);
  --active_;
  if (active_ > 0 && waiting_ == active_) {  // everybody else was waiting for this client
    RunRound();
    ReleaseWaiters(lk);
  }
}

int32_t DeviceBatcher::BaSolve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt, ptz_lm_summary* summary)
{
  ptz_lm_options dflt;
  if (!opt) { ptz_lm_options_default(&dflt); opt = &dflt; }
  BaReq req{p, cam, ray, tlw, opt, summary, PTZ_EINVAL};
  std::unique_lock<std::mutex> lk(mu_);
  ba_.push_back(&req);
  Arrive(lk);
  return req.rc;
}

int32_t DeviceBatcher::BaSolveView(const ptz_rig_view* view, int32_t factor_type, double* cam, const double* rkinv, const ptz_lm_options* opt,
                                   ptz_lm_summary* summary)
{
  ptz_lm_options dflt;
  if (!opt) { ptz_lm_options_default(&dflt); opt = &dflt; }
  BavReq req{view, factor_type, cam, rkinv, opt, summary, PTZ_EINVAL};
  std::unique_lock<std::mutex> lk(mu_);
  bav_.push_back(&req);
  if (Trace() && last_end_ms_ > 0) trace_late_[1] = std::max(trace_late_[1], NowMs() - last_end_ms_);
  Arrive(lk);
  return req.rc;
}

int32_t DeviceBatcher::KrtSolveBatch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, const double* cam_ref,
                                     double* cam_cur, int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                                     ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  ptz_lm_options dflt;
  if (!opt) { ptz_lm_options_default(&dflt); opt = &dflt; }
  KrtReq req{n_query, match_ptr, uv_ref, uv_cur, cam_ref, cam_cur, factor_type, max_reproj_error, opt, summaries, accepted, device_ms, PTZ_EINVAL, nullptr};
  std::unique_lock<std::mutex> lk(mu_);
  krt_.push_back(&req);
  if (Trace() && last_end_ms_ > 0) { trace_late_[2] = std::max(trace_late_[2], NowMs() - last_end_ms_); if (trace_first_ == 0) trace_first_ = NowMs() - last_end_ms_; }
  Arrive(lk);
  return req.rc;
}

int32_t DeviceBatcher::KrtSolveAttempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur, int32_t factor_type,
                                        double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  ptz_lm_options dflt;
  if (!opt) { ptz_lm_options_default(&dflt); opt = &dflt; }
  KrtReq req{n_query, nullptr, nullptr, nullptr, cam_ref, cam_cur, factor_type, max_reproj_error, opt, summaries, accepted, device_ms, PTZ_EINVAL, attempts};
  std::unique_lock<std::mutex> lk(mu_);
  krt_.push_back(&req);
  if (Trace() && last_end_ms_ > 0) { trace_late_[2] = std::max(trace_late_[2], NowMs() - last_end_ms_); if (trace_first_ == 0) trace_first_ = NowMs() - last_end_ms_; }
  Arrive(lk);
  return req.rc;
}

void DeviceBatcher::RunRound()
{
  ++stats_.rounds;
  const double t_round = NowMs();
  if (Trace() && last_end_ms_ > 0)
    fprintf(stderr, "batcher round clients=%d ba=%zu views=%zu krt=%zu: the clients' own work since the last round %.2f ms (last view request after %.2f, last registration request after %.2f, first after %.2f)\n", active_, ba_.size(), bav_.size(),
            krt_.size(), t_round - last_end_ms_, trace_late_[1], trace_late_[2], trace_first_);
  trace_late_[0] = trace_late_[1] = trace_late_[2] = 0; trace_first_ = 0;
  // The round's registration launches and its bundle adjustments do not depend on each other (they belong to different rigs): the
  // registration side -- mostly host-side packing of the matches -- runs on a thread of its own beside the bundle adjustments.
  std::thread krt_thread;
  if (!krt_.empty()) {
    krt_thread = std::thread([this] {
      std::vector<char> done_krt(krt_.size(), 0);
      for (size_t i = 0; i < krt_.size(); ++i) {
        if (done_krt[i]) continue;
        std::vector<KrtReq*> group{krt_[i]};
        done_krt[i] = 1;
        for (size_t j = i + 1; j < krt_.size(); ++j) {
          if (done_krt[j] || krt_[j]->factor_type != krt_[i]->factor_type || krt_[j]->max_reproj_error != krt_[i]->max_reproj_error ||
              !SameOptions(krt_[j]->opt, krt_[i]->opt) || (krt_[j]->attempts == nullptr) != (krt_[i]->attempts == nullptr))
            continue;
          group.push_back(krt_[j]);
          done_krt[j] = 1;
        }
        RunKrt(group);
      }
    });
  }
  // requests that can share a launch: same factor type and options (a batch has ONE options block); a problem with shared
  // intrinsics or annotations is solved on its own (nothing in the incremental pipeline makes those)
  std::vector<char> done_ba(ba_.size(), 0);
  for (size_t i = 0; i < ba_.size(); ++i) {
    if (done_ba[i]) continue;
    std::vector<BaReq*> group{ba_[i]};
    done_ba[i] = 1;
    const bool plain = !ba_[i]->p->ic_of_cam && ba_[i]->p->n_obs3d == 0;
    for (size_t j = i + 1; plain && j < ba_.size(); ++j) {
      if (done_ba[j] || ba_[j]->p->ic_of_cam || ba_[j]->p->n_obs3d != 0) continue;
      if (ba_[j]->p->factor_type != ba_[i]->p->factor_type || !SameOptions(ba_[j]->opt, ba_[i]->opt)) continue;
      group.push_back(ba_[j]);
      done_ba[j] = 1;
    }
    RunBa(group);
  }
  std::vector<char> done_bav(bav_.size(), 0);
  for (size_t i = 0; i < bav_.size(); ++i) {  // views of resident rigs: one device-built batch per (factor type, options)
    if (done_bav[i]) continue;
    std::vector<BavReq*> group{bav_[i]};
    done_bav[i] = 1;
    for (size_t j = i + 1; j < bav_.size(); ++j) {
      if (done_bav[j] || bav_[j]->factor_type != bav_[i]->factor_type || !SameOptions(bav_[j]->opt, bav_[i]->opt)) continue;
      group.push_back(bav_[j]);
      done_bav[j] = 1;
    }
    RunBaViews(group);
  }
  if (krt_thread.joinable()) krt_thread.join();
  last_end_ms_ = NowMs();
  if (Trace()) fprintf(stderr, "batcher round ran %.2f ms\n", last_end_ms_ - t_round);
  ba_.clear();
  bav_.clear();
  krt_.clear();
}

void DeviceBatcher::RunBa(std::vector<BaReq*>& reqs)
{
  const double t0 = NowMs();
  ++stats_.ba_batches;
  stats_.ba_problems += static_cast<long>(reqs.size());
  // A batch costs host time on both sides of its device time (structure of every problem, uploads, read-back): a large group
  // goes as up to eight batches on as many host threads, so that one batch's host work overlaps another's device work.  A
  // problem's result does not depend on the batch it is in.
  size_t parts = reqs.size() >= 32 ? 8 : (reqs.size() >= 16 ? 4 : (reqs.size() >= 8 ? 2 : 1));  // (64 rigs: 4 -> 806, 8 -> 768, 12 -> 825 ms in the rounds)
  if (const char* e = getenv("PTZ_BATCHER_PARTS")) parts = std::max<size_t>(1, std::min<size_t>(reqs.size(), static_cast<size_t>(atoi(e))));
  if (parts > 1) {
    std::vector<std::vector<BaReq*>> chunk(parts);
    for (size_t i = 0; i < reqs.size(); ++i) chunk[i * parts / reqs.size()].push_back(reqs[i]);
    std::vector<std::thread> th;
    for (size_t k = 1; k < parts; ++k) th.emplace_back([&, k] { RunBaBatch(chunk[k]); });
    RunBaBatch(chunk[0]);
    for (std::thread& t : th) t.join();
  }
  else RunBaBatch(reqs);
  stats_.ba_ms += NowMs() - t0;
}

void DeviceBatcher::RunBaBatch(std::vector<BaReq*>& reqs)
{
  if (reqs.size() == 1) {
    BaReq& r = *reqs[0];
    r.rc = ptz_ba_solve(r.p, r.cam, r.ray, r.tlw, r.opt, r.summary);
    return;
  }
  const int32_t n = static_cast<int32_t>(reqs.size());
  std::vector<ptz_ba_problem> probs(n);
  size_t n_cam = 0, n_ray = 0;
  for (int32_t i = 0; i < n; ++i) {
    probs[i] = *reqs[i]->p;
    n_cam += static_cast<size_t>(probs[i].n_cam);
    n_ray += static_cast<size_t>(probs[i].n_ray);
  }
  std::vector<double> cam(15 * n_cam), ray(3 * n_ray), tlw(6 * static_cast<size_t>(n), 0.0);
  {
    size_t co = 0, ro = 0;
    for (int32_t i = 0; i < n; ++i) {
      const BaReq& r = *reqs[i];
      memcpy(cam.data() + 15 * co, r.cam, sizeof(double) * 15 * probs[i].n_cam);
      memcpy(ray.data() + 3 * ro, r.ray, sizeof(double) * 3 * probs[i].n_ray);
      if (r.tlw) memcpy(tlw.data() + 6 * static_cast<size_t>(i), r.tlw, sizeof(double) * 6);
      co += probs[i].n_cam; ro += probs[i].n_ray;
    }
  }
  std::vector<ptz_lm_summary> summ(n);
  ptz_ba_batch* b = nullptr;
  const double t0 = NowMs();
  int32_t rc = ptz_ba_batch_create(n, probs.data(), reqs[0]->opt, &b);
  const double t1 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_set_state(b, cam.data(), ray.data(), tlw.data());
  const double t2 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_solve(b, summ.data());
  const double t3 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_get_state(b, cam.data(), ray.data(), tlw.data());
  const double t4 = NowMs();
  if (b) ptz_ba_batch_destroy(b);
  if (Trace())
    fprintf(stderr, "batcher ba n=%d cams=%zu rays=%zu create %.2f set %.2f solve %.2f get %.2f destroy %.2f ms\n", n, n_cam, n_ray, t1 - t0,
            t2 - t1, t3 - t2, t4 - t3, NowMs() - t4);
  if (rc != PTZ_OK && rc != PTZ_ENODEVICE && rc != PTZ_ENOMEM) {
    // one malformed / oversized problem must not fail its neighbours: every request gets its own verdict
    for (BaReq* r : reqs) r->rc = ptz_ba_solve(r->p, r->cam, r->ray, r->tlw, r->opt, r->summary);
    return;
  }
  size_t co = 0, ro = 0;
  for (int32_t i = 0; i < n; ++i) {
    BaReq& r = *reqs[i];
    r.rc = rc;
    if (rc == PTZ_OK) {
      memcpy(r.cam, cam.data() + 15 * co, sizeof(double) * 15 * probs[i].n_cam);
      memcpy(r.ray, ray.data() + 3 * ro, sizeof(double) * 3 * probs[i].n_ray);
      if (r.tlw) memcpy(r.tlw, tlw.data() + 6 * static_cast<size_t>(i), sizeof(double) * 6);
      if (r.summary) *r.summary = summ[i];
    }
    co += probs[i].n_cam; ro += probs[i].n_ray;
  }
}

namespace {
// one view alone: what a solo PTZRayOptimizer::Solve over a view does
int32_t SolveOneView(const ptz_rig_view* view, int32_t factor_type, double* cam, const double* rkinv, const ptz_lm_options* opt, ptz_lm_summary* summary)
{
  ptz_ba_batch* b = nullptr;
  int32_t rc = ptz_ba_batch_create_views(1, view, factor_type, opt, &b);
  if (rc == PTZ_OK) rc = ptz_ba_batch_set_state_pix2ray(b, cam, rkinv);
  ptz_lm_summary s{};
  if (rc == PTZ_OK) rc = ptz_ba_batch_solve(b, &s);
  if (rc == PTZ_OK) rc = ptz_ba_batch_get_state(b, cam, nullptr, nullptr);
  if (b) ptz_ba_batch_destroy(b);
  if (rc == PTZ_OK && summary) *summary = s;
  return rc;
}
}  // namespace

void DeviceBatcher::RunBaViews(std::vector<BavReq*>& reqs)
{
  const double t0 = NowMs();
  // The packed problems are built on the device: a round costs the host little besides the tile plans, so it goes as ONE batch;
  // PTZ_BATCHER_PARTS splits it over host threads (batches overlap each other's host and device work)
  size_t parts = 1;
  if (const char* e = getenv("PTZ_BATCHER_PARTS")) parts = std::max<size_t>(1, std::min<size_t>(reqs.size(), static_cast<size_t>(atoi(e))));
  stats_.ba_batches += static_cast<long>(parts);
  stats_.ba_problems += static_cast<long>(reqs.size());
  if (parts > 1) {
    std::vector<std::vector<BavReq*>> chunk(parts);
    for (size_t i = 0; i < reqs.size(); ++i) chunk[i * parts / reqs.size()].push_back(reqs[i]);
    std::vector<std::thread> th;
    for (size_t k = 1; k < parts; ++k) th.emplace_back([&, k] { RunBaViewBatch(chunk[k]); });
    RunBaViewBatch(chunk[0]);
    for (std::thread& t : th) t.join();
  }
  else RunBaViewBatch(reqs);
  stats_.ba_ms += NowMs() - t0;
}

void DeviceBatcher::RunBaViewBatch(std::vector<BavReq*>& reqs)
{
  const int32_t n = static_cast<int32_t>(reqs.size());
  if (n == 1) {
    BavReq& r = *reqs[0];
    r.rc = SolveOneView(r.view, r.factor_type, r.cam, r.rkinv, r.opt, r.summary);
    return;
  }
  std::vector<ptz_rig_view> views(n);
  size_t n_cam = 0;
  for (int32_t i = 0; i < n; ++i) { views[i] = *reqs[i]->view; n_cam += static_cast<size_t>(views[i].n_cam); }
  std::vector<double> cam(15 * n_cam), rk(9 * n_cam);
  {
    size_t co = 0;
    for (int32_t i = 0; i < n; ++i) {
      memcpy(cam.data() + 15 * co, reqs[i]->cam, sizeof(double) * 15 * views[i].n_cam);
      memcpy(rk.data() + 9 * co, reqs[i]->rkinv, sizeof(double) * 9 * views[i].n_cam);
      co += static_cast<size_t>(views[i].n_cam);
    }
  }
  std::vector<ptz_lm_summary> summ(n);
  ptz_ba_batch* b = nullptr;
  const double t0 = NowMs();
  int32_t rc = ptz_ba_batch_create_views(n, views.data(), reqs[0]->factor_type, reqs[0]->opt, &b);
  const double t1 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_set_state_pix2ray(b, cam.data(), rk.data());
  const double t2 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_solve(b, summ.data());
  const double t3 = NowMs();
  if (rc == PTZ_OK) rc = ptz_ba_batch_get_state(b, cam.data(), nullptr, nullptr);
  const double t4 = NowMs();
  if (b) ptz_ba_batch_destroy(b);
  if (Trace())
    fprintf(stderr, "batcher views n=%d cams=%zu create %.2f set %.2f solve %.2f get %.2f destroy %.2f ms\n", n, n_cam, t1 - t0, t2 - t1, t3 - t2, t4 - t3,
            NowMs() - t4);
  if (rc != PTZ_OK && rc != PTZ_ENODEVICE && rc != PTZ_ENOMEM) {
    // a view without candidate observations (or an oversized one) must not fail its neighbours: every request gets its own verdict
    // (an empty view is an everyday event of the incremental pipeline; anything else doubles the round's device work and says so)
    if (rc != PTZ_ENOOBS) fprintf(stderr, "[ptzcalib] view batch of %d refused with code %d: its views are solved one by one\n", (int)n, (int)rc);
    for (BavReq* r : reqs) r->rc = SolveOneView(r->view, r->factor_type, r->cam, r->rkinv, r->opt, r->summary);
    return;
  }
  size_t co = 0;
  for (int32_t i = 0; i < n; ++i) {
    BavReq& r = *reqs[i];
    r.rc = rc;
    if (rc == PTZ_OK) {
      memcpy(r.cam, cam.data() + 15 * co, sizeof(double) * 15 * views[i].n_cam);
      if (r.summary) *r.summary = summ[i];
    }
    co += static_cast<size_t>(views[i].n_cam);
  }
}

void DeviceBatcher::RunKrt(std::vector<KrtReq*>& reqs)
{
  const double t0 = NowMs();
  ++stats_.krt_launches;
  if (reqs[0]->attempts) {  // entries of resident tables: the launch's arguments are the (table, entry) pairs and the cameras, nothing else moves
    size_t nq = 0;
    for (const KrtReq* r : reqs) nq += static_cast<size_t>(r->n_query);
    stats_.krt_queries += static_cast<long>(nq);
    std::vector<ptz_krt_attempt> att(nq);
    std::vector<double> cam_ref(15 * nq), cam_cur(15 * nq);
    std::vector<ptz_lm_summary> summ(nq);
    std::vector<int32_t> acc(nq, 0);
    size_t q0 = 0;
    for (const KrtReq* r : reqs) {
      memcpy(att.data() + q0, r->attempts, sizeof(ptz_krt_attempt) * r->n_query);
      memcpy(cam_ref.data() + 15 * q0, r->cam_ref, sizeof(double) * 15 * r->n_query);
      memcpy(cam_cur.data() + 15 * q0, r->cam_cur, sizeof(double) * 15 * r->n_query);
      q0 += static_cast<size_t>(r->n_query);
    }
    const double t1 = NowMs();
    double dev_ms = 0;
    ptz_lm_options opt = *reqs[0]->opt;
    if (opt.krt_lanes_per_query == 0) opt.krt_lanes_per_query = 64;  // (the form of a rig's own launches: see DeviceKrtSolveBatch)
    const int32_t rc = ptz_krt_solve_attempts(static_cast<int32_t>(nq), att.data(), cam_ref.data(), cam_cur.data(), reqs[0]->factor_type,
                                              reqs[0]->max_reproj_error, &opt, summ.data(), acc.data(), &dev_ms);
    if (Trace()) fprintf(stderr, "batcher krt reqs=%zu queries=%zu matches=resident pack %.2f call %.2f device %.3f ms\n", reqs.size(), nq, t1 - t0, NowMs() - t1, dev_ms);
    q0 = 0;
    for (KrtReq* r : reqs) {
      r->rc = rc;
      if (rc == PTZ_OK) {
        memcpy(r->cam_cur, cam_cur.data() + 15 * q0, sizeof(double) * 15 * r->n_query);
        if (r->summaries) memcpy(r->summaries, summ.data() + q0, sizeof(ptz_lm_summary) * r->n_query);
        if (r->accepted) memcpy(r->accepted, acc.data() + q0, sizeof(int32_t) * r->n_query);
        if (r->device_ms) *r->device_ms = dev_ms;
      }
      q0 += static_cast<size_t>(r->n_query);
    }
    stats_.krt_ms += NowMs() - t0;
    return;
  }
  if (reqs.size() == 1) {
    KrtReq& r = *reqs[0];
    stats_.krt_queries += r.n_query;
    r.rc = ptz_krt_solve_batch(r.n_query, r.match_ptr, r.uv_ref, r.uv_cur, r.cam_ref, r.cam_cur, r.factor_type, r.max_reproj_error, r.opt,
                               r.summaries, r.accepted, r.device_ms);
    stats_.krt_ms += NowMs() - t0;
    return;
  }
  size_t nq = 0, nm = 0;
  for (const KrtReq* r : reqs) { nq += static_cast<size_t>(r->n_query); nm += static_cast<size_t>(r->match_ptr[r->n_query] - r->match_ptr[0]); }
  stats_.krt_queries += static_cast<long>(nq);
  std::vector<int64_t> ptr(nq + 1, 0);
  std::vector<float> uv_ref(2 * nm), uv_cur(2 * nm);
  std::vector<double> cam_ref(15 * nq), cam_cur(15 * nq);
  std::vector<ptz_lm_summary> summ(nq);
  std::vector<int32_t> acc(nq, 0);
  {
    // offsets first, then the copies on a few threads (a round of 64 rigs merges ~4 MB of matches: 0.6 ms on one thread)
    std::vector<size_t> q_at(reqs.size()), m_at(reqs.size());
    size_t q0 = 0, m0 = 0;
    for (size_t k = 0; k < reqs.size(); ++k) {
      const KrtReq* r = reqs[k];
      q_at[k] = q0; m_at[k] = m0;
      q0 += static_cast<size_t>(r->n_query);
      m0 += static_cast<size_t>(r->match_ptr[r->n_query] - r->match_ptr[0]);
    }
    auto copy_range = [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; ++k) {
        const KrtReq* r = reqs[k];
        const int64_t base = r->match_ptr[0];
        const size_t m = static_cast<size_t>(r->match_ptr[r->n_query] - base);
        for (int32_t q = 0; q < r->n_query; ++q) ptr[q_at[k] + q + 1] = static_cast<int64_t>(m_at[k]) + (r->match_ptr[q + 1] - base);
        memcpy(uv_ref.data() + 2 * m_at[k], r->uv_ref + 2 * base, sizeof(float) * 2 * m);
        memcpy(uv_cur.data() + 2 * m_at[k], r->uv_cur + 2 * base, sizeof(float) * 2 * m);
        memcpy(cam_ref.data() + 15 * q_at[k], r->cam_ref, sizeof(double) * 15 * r->n_query);
        memcpy(cam_cur.data() + 15 * q_at[k], r->cam_cur, sizeof(double) * 15 * r->n_query);
      }
    };
    const size_t n_thr = nm > (size_t)200000 ? std::min<size_t>(4, reqs.size()) : 1;
    if (n_thr > 1) {
      std::vector<std::thread> th;
      for (size_t t = 1; t < n_thr; ++t) th.emplace_back(copy_range, reqs.size() * t / n_thr, reqs.size() * (t + 1) / n_thr);
      copy_range(0, reqs.size() / n_thr);
      for (std::thread& x : th) x.join();
    }
    else copy_range(0, reqs.size());
  }
  const double t1 = NowMs();
  double dev_ms = 0;
  // the merged launch must give every query the bits of its own small launch: the form is pinned to what those pick (a wave per
  // query, the latency form), whatever the merged size
  ptz_lm_options opt = *reqs[0]->opt;
  if (opt.krt_lanes_per_query == 0) opt.krt_lanes_per_query = 64;
  const int32_t rc = ptz_krt_solve_batch(static_cast<int32_t>(nq), ptr.data(), uv_ref.data(), uv_cur.data(), cam_ref.data(), cam_cur.data(),
                                         reqs[0]->factor_type, reqs[0]->max_reproj_error, &opt, summ.data(), acc.data(), &dev_ms);
  if (Trace()) fprintf(stderr, "batcher krt reqs=%zu queries=%zu matches=%zu pack %.2f call %.2f device %.3f ms\n", reqs.size(), nq, nm, t1 - t0, NowMs() - t1, dev_ms);
  size_t q0 = 0;
  for (KrtReq* r : reqs) {
    r->rc = rc;
    if (rc == PTZ_OK) {
      memcpy(r->cam_cur, cam_cur.data() + 15 * q0, sizeof(double) * 15 * r->n_query);
      if (r->summaries) memcpy(r->summaries, summ.data() + q0, sizeof(ptz_lm_summary) * r->n_query);
      if (r->accepted) memcpy(r->accepted, acc.data() + q0, sizeof(int32_t) * r->n_query);
      if (r->device_ms) *r->device_ms = dev_ms;
    }
    q0 += static_cast<size_t>(r->n_query);
  }
  stats_.krt_ms += NowMs() - t0;
}

int32_t DeviceBaSolve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt, ptz_lm_summary* summary)
{
  if (DeviceBatcher* b = DeviceBatcher::Current()) return b->BaSolve(p, cam, ray, tlw, opt, summary);
  return ptz_ba_solve(p, cam, ray, tlw, opt, summary);
}

int32_t DeviceBaSolveView(const ptz_rig_view* view, int32_t factor_type, double* cam, const double* rkinv, const ptz_lm_options* opt,
                          ptz_lm_summary* summary)
{
  if (DeviceBatcher* b = DeviceBatcher::Current()) return b->BaSolveView(view, factor_type, cam, rkinv, opt, summary);
  return SolveOneView(view, factor_type, cam, rkinv, opt, summary);
}

int32_t DeviceKrtSolveBatch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, const double* cam_ref,
                            double* cam_cur, int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                            ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  // The orchestration's registration launches always use ONE lane form -- a wave per query, the latency form -- whatever their size
  // and whether or not they are merged with other rigs' launches: the form is part of a query's bits (ptz_lm_options::
  // krt_lanes_per_query), and a rig must take the same decisions alone, in a lock step of any size and under PTZ_KRT_GROUP.
  ptz_lm_options pinned;
  if (opt) pinned = *opt; else ptz_lm_options_default(&pinned);
  if (pinned.krt_lanes_per_query == 0) pinned.krt_lanes_per_query = 64;
  if (DeviceBatcher* b = DeviceBatcher::Current())
    return b->KrtSolveBatch(n_query, match_ptr, uv_ref, uv_cur, cam_ref, cam_cur, factor_type, max_reproj_error, &pinned, summaries, accepted, device_ms);
  return ptz_krt_solve_batch(n_query, match_ptr, uv_ref, uv_cur, cam_ref, cam_cur, factor_type, max_reproj_error, &pinned, summaries, accepted, device_ms);
}

int32_t DeviceKrtSolveAttempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur, int32_t factor_type,
                               double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  ptz_lm_options pinned;  // (one lane form for the orchestration's launches: see DeviceKrtSolveBatch)
  if (opt) pinned = *opt; else ptz_lm_options_default(&pinned);
  if (pinned.krt_lanes_per_query == 0) pinned.krt_lanes_per_query = 64;
  if (DeviceBatcher* b = DeviceBatcher::Current())
    return b->KrtSolveAttempts(n_query, attempts, cam_ref, cam_cur, factor_type, max_reproj_error, &pinned, summaries, accepted, device_ms);
  return ptz_krt_solve_attempts(n_query, attempts, cam_ref, cam_cur, factor_type, max_reproj_error, &pinned, summaries, accepted, device_ms);
}

}  // namespace ptzcalib
