// ptz_ba.hip -- PTZ-IBA global bundle adjustment on MI355X (gfx950): batched Levenberg-Marquardt over
// independent scenes, everything between the packed observations and the refined parameters on device.
//
// Replaces PTZRayOptimizer::Solve (src/core/ptzray_optimizer.cc:454-489) from AddConstraints2d2d
// (:799-885) through ceres::Solve (:469-475): residual blocks with ScaledLoss(track length) weights,
// SubsetParameterization masks (free: focal, rvec, ray [, k1]), the Ceres 1.14 trust-region LM policy
// (Jacobi scaling, LM diagonal clamp, step-quality radius update,
// function/parameter/gradient tolerances), Schur elimination of the 3x3 ray blocks (row S5) and the
// solve of the reduced camera system (ptz_chol.hip).
//
// Per LM pass and scene (all scenes of the batch advance in lock-step, finished scenes early-out):
//   lm_pre      finalize the previous iteration, termination checks               [1 block / scene]
//   ray_prep    E = (V + D^2)^-1 per ray, z = E g_r                                [thread / ray]
//   cam_diag    LM diagonal of the camera blocks                                   [thread / camera]
//   schur       row-block i of the reduced system: T_a = W_a E staged in LDS, S_ii, b_i, then
//               S_ij = - sum_{tracks seen by i and j} T_a W_b^T for j < i          [workgroup / camera]
//               (b is written as row n of the padded S)
//   cholesky    S y_c = b  (ptz_chol.hip: panel + MFMA syrk per 64-wide block column, back-substitution)
//   cam_update  candidate camera, its rotation block                              [thread / camera]
//   eval        y_r = E (g_r - sum_a Jr_a^T Jc_a y_c), candidate ray, model cost change
//               -(J d)^T (r + J d / 2) and candidate cost (Jacobians recomputed)   [thread / ray]
//   lm_post     step validity, tolerances, rho, accept/reject, radius              [1 block / scene]
//   (if accepted) cam_prep, lin_ray, lin_cam: re-linearise at the new point
// Observation records are 16 B (2 x f32 pixel, i32 camera, i32 ray); a workgroup of lin_ray/eval/backsub
// stages its scene's camera blocks (rotation, SO(3) Jacobian, intrinsics, Jacobi scales) in LDS once and
// every thread then reads them by camera id.  All reductions are fixed-order (bitwise reproducible).
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "ptz_common.h"
#include "ptz_pool.h"
#include "ptz_factor.h"

#include "ptz_ba_kernels.h"
#include "ptz_view_kernels.h"
#include <rocprim/rocprim.hpp>

namespace ptz {

// =============================================================================================================
// host side
// =============================================================================================================
enum ProfSlot { P_LIN = 0, P_LMCTL, P_RAYPREP, P_CLEAR, P_SCHUR, P_RHS, P_CHOL_PANEL, P_CHOL_SYRK, P_CHOL_BACK,
                P_BACKSUB, P_EVAL, P_SYNC, P_NSLOT };
static const char* kSlotNames[PTZ_PROF_SLOTS] = {"linearize", "lm_control", "ray_prep", "clear", "schur", "rhs",
                                                 "chol_panel", "chol_syrk", "chol_backsolve", "backsub", "eval",
                                                 "host_sync", "", "", "", ""};

}  // namespace ptz

using namespace ptz;

// chol_factor_solve split so that the three kernel families can be timed separately
namespace ptz {
void chol_factor_solve_profiled(const CholBatch& cb, double* x, hipStream_t stream, void* prof, bool fused);
}

// The launch shape of one LM pass.  A batch's passes start at full size; once few of its scenes are still active the host
// switches to compacted shapes whose grids cover `slots` scenes (blockIdx.y -> scene through the device's compacted list), with
// the workgroup size, the kernel variants and the factorisation path a batch of that size would get -- the stragglers of a
// 1000-scene batch then cost what a handful of scenes cost, not a thousand empty workgroups per kernel.
struct PassShape {
  int slots = 0;          // grid extent over scenes
  bool compact = false;   // blockIdx.y is a slot of the compacted list
  int ray_block = 1024;   // rays per workgroup of the ray-centric kernels
  bool small_blocks = false, fused = false;  // SMALL kernel variants; one-launch-per-column factorisation
  bool fuse_ctl = false;  // LM control and the camera update inside k_eval / k_lin_cam (no k_lm_pre / k_lm_post / k_cam_update launches)
  int eval_lanes = 1;     // lanes per ray in k_eval (4: the form of a few scenes, see k_eval)
  int max_chunk = 0;
  size_t lin_smem = 0, eval_smem = 0;
};

struct ptz_ba_batch {
  int n_scene = 0, type = 0, nc = 4, device = 0;
  std::vector<SceneDev> scenes;
  int total_cam = 0, total_ray = 0, total_obs = 0, total_pair = 0, total_ent = 0, total_run = 0, total_chunk = 0;  // total_chunk: partial-sum slots (waves of 64 rays + 1 per scene)
  int ray_block = RAY_BLOCK;
  bool schur_tg = false;         // a camera with more observations than k_schur's LDS table holds: table in global memory
  bool schur_w = false;          // PTZ_BA_SCHUR_W=1: round 2's Schur kernel over materialised W rows (kept for A/B measurements)
  bool schur_f = false;          // PTZRay, table in LDS: k_schur_f (factored 8-double rows, three workgroups per compute unit; PTZ_BA_SCHUR_F=0: k_schur)
  bool gtab = false;             // camera tables too large for LDS: the GTAB instantiations read them from global memory
  int n_group_hint(int n) const { if (const char* e = getenv("PTZ_BA_STREAMS")) return std::max(1, atoi(e)); return n >= 32 ? 2 : 1; }
  int max_cam = 0, max_ray = 0, max_chunk = 0, max_pair = 0, max_n = 0, max_cam_obs = 0, max_cam_ent = 0, max_cam_pair = 0, max_cam_run = 0;
  ptz_lm_options opt;
  Dev d;
  std::vector<void*> allocs;
  hipEvent_t create_ev = nullptr;    // end of ptz_ba_batch_create's work on `io`
  double* rkinv_dev = nullptr;       // ptz_ba_batch_set_state_pix2ray: the cameras' R^-1 K^-1 [9 total_cam]
  std::vector<void*> staged_pinned;  // staging blocks of uploads still in flight on `io` (released behind the next wait for it)
  void release_staged() { for (void* p : staged_pinned) ptzpool::pinned_release(p); staged_pinned.clear(); }
  hipStream_t stream = nullptr;   // stream of the group being enqueued (LAUNCH / prof_* use it)
  hipStream_t io = nullptr;       // uploads of ptz_ba_batch_create (before the group streams exist); see copy_on()
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // scenes are split into independent groups, one HIP stream each, so that the latency-bound kernels of one
  // group (diagonal-tile factorisation, back-substitution, LM control) overlap the throughput kernels of another
  int n_group = 1;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> fork_ev, join_ev;
  std::vector<int> group_first, group_count;
  std::vector<Dev> dg;
  // Cholesky look-ahead: one auxiliary stream and two events per group stream
  std::vector<hipStream_t> aux;
  std::vector<hipEvent_t> la_ev;
  bool lookahead = true;
  bool left_looking = true;  // left-looking column updates instead of right-looking trailing updates
  int group_of(hipStream_t st) const { for (size_t g = 0; g < streams.size(); ++g) if (streams[g] == st) return (int)g; return -1; }
  hipStream_t aux_stream(hipStream_t st) const { const int g = group_of(st); return (lookahead && g >= 0 && g < (int)aux.size()) ? aux[g] : nullptr; }
  void lookahead_events(hipStream_t st, hipEvent_t* t, hipEvent_t* r) const { const int g = group_of(st); *t = la_ev[2 * g]; *r = la_ev[2 * g + 1]; }
  // pass pipeline control (Dev::grp_ctl / host_ctl): 4 ints per scene group, device block + pinned host mirror
  int* d_ctl = nullptr;
  int* h_ctl = nullptr;      // pinned; read by the host while the device writes (fine-grained host memory)
  int* h_ctl_dev = nullptr;  // the device's address of h_ctl
  int ctl_groups = 0;        // groups the control blocks were sized for
  int ahead = 3;             // LM passes the host may have enqueued beyond the last one known to have reached its step evaluation
  // one captured LM pass per scene group (kernel arguments never change during a batch's life), replayed per pass
  bool use_graph = true;
  std::vector<PassShape> shapes;                        // [0] = full size, then compacted shapes by ascending slot count
  std::vector<std::vector<hipGraphExec_t>> pass_graph;  // [group][shape]
  int* d_act = nullptr;                                 // compacted scene lists, n_scene ints (each group its own range)
  bool compaction = true;
  double *cam0 = nullptr, *ray0 = nullptr, *tlw0 = nullptr;  // device copies of the initial state
  std::vector<int> sched_kmin;  // CholBatch::sched_kmin
  double* dsp0 = nullptr;  // PTZRayDistDisp: initial displacement block, one copy per camera (zeros unless ptz_ba_batch_set_disp)
  int has3d = 0, total_o3 = 0;
  // Rays are renumbered inside the library, longest track first (see build_pairs): ray_perm[ray_off + j] = the caller's
  // scene-local index of internal ray j.  set_state / get_state / linearize translate.
  std::vector<int> ray_perm;
  const int* d_ray_perm = nullptr;  // the same on the device (ptz_ba_batch_get_state gathers the result there)
  // shared intrinsics: per global camera, the global index of the first camera of its group (source of the initial values)
  std::vector<int> first_of_group;
  int max_grp = 0;
  bool has_state = false;
  bool views_mode = false;  // built by ptz_ba_batch_create_views: structure arrays at upper-bound extents, ray order on the device only
  int n_solves = 0;  // solves of this batch so far
  double last_ms = 0;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, int>> ev_used;  // (slot, event index of the start; stop = +1)
  double prof_ms[PTZ_PROF_SLOTS] = {0};
  int64_t prof_n[PTZ_PROF_SLOTS] = {0};

  template <typename T> int alloc(T** p, size_t count)
  {
    void* q = nullptr;
    if (ptzpool::dev_acquire(device, sizeof(T) * std::max<size_t>(count, 1), &q) != hipSuccess) return PTZ_ENOMEM;
    allocs.push_back(q);
    *p = (T*)q;
    return PTZ_OK;
  }
  void prof_begin(int slot)
  {
    if (!profiling) return;
    if (ev_used.size() * 2 + 2 > ev_pool.size()) {
      for (int i = 0; i < 2; ++i) { hipEvent_t e; (void)ptzpool::event_acquire(device, true, &e); ev_pool.push_back(e); }
    }
    const int idx = (int)ev_used.size() * 2;
    ev_used.push_back({slot, idx});
    (void)hipEventRecord(ev_pool[idx], stream);
  }
  void prof_end()
  {
    if (!profiling) return;
    (void)hipEventRecord(ev_pool[ev_used.back().second + 1], stream);
  }
  void prof_collect()
  {
    for (auto& u : ev_used) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev_pool[u.second], ev_pool[u.second + 1]) == hipSuccess) {
        prof_ms[u.first] += ms;
        prof_n[u.first] += 1;
      }
    }
    ev_used.clear();
  }
};

namespace {

// dynamic LDS of k_schur: T table, scratch, staged entries, pair offsets, reduction strip
// Elimination order of a reduced camera system at tile granularity.
//
// bisect_tiles: the tiles `ids` (natural numbers, ascending) with adjacency adj(a, e); looks for a separator made of a prefix
// [0, w1) and an interval [m, m + w2) of the list -- the two cuts of a ring whose images are numbered along it, or the middle
// cut of a band -- such that the rest falls apart; the parts are dealt to two lanes.  cls[i] = 0 / 1 (lane, 0 the longer) or
// 2 (separator).  The dependent chain of the factorisation of these tiles is then max(lane) + separator instead of their
// number; returns false unless that saves at least `min_saving`.
template <typename Adj>
inline bool bisect_tiles(const std::vector<int>& ids, Adj&& adj, int min_tiles, int min_saving, std::vector<int>& cls)
{
  const int nf = (int)ids.size();
  if (nf < min_tiles) return false;
  int best_cost = nf + 1;
  std::vector<int> comp(nf), stack;
  bool found = false;
  cls.assign(nf, 2);
  for (int w1 = 0; w1 <= 4; ++w1)
    for (int w2 = 0; w2 <= 4; ++w2)
      for (int m = w1; m + w2 <= nf; ++m) {
        if (w2 == 0 && m > w1) break;
        if (w1 + w2 == 0) continue;
        int ncomp = 0;  // tiles outside the separator: connected components
        for (int t = 0; t < nf; ++t) comp[t] = (t < w1 || (t >= m && t < m + w2)) ? -2 : -1;
        for (int t = 0; t < nf; ++t) {
          if (comp[t] != -1) continue;
          comp[t] = ncomp;
          stack.assign(1, t);
          while (!stack.empty()) {
            const int u = stack.back(); stack.pop_back();
            for (int v = 0; v < nf; ++v)
              if (comp[v] == -1 && adj(ids[u], ids[v])) { comp[v] = ncomp; stack.push_back(v); }
          }
          ++ncomp;
        }
        if (ncomp < 2) continue;
        std::vector<int> size(ncomp, 0), order(ncomp);
        for (int t = 0; t < nf; ++t) if (comp[t] >= 0) ++size[comp[t]];
        for (int c = 0; c < ncomp; ++c) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](int a, int e) { return size[a] > size[e]; });
        int la = 0, lb = 0;
        std::vector<int> lane_of(ncomp);
        for (int c : order) { if (la <= lb) { lane_of[c] = 0; la += size[c]; } else { lane_of[c] = 1; lb += size[c]; } }
        const int cost = std::max(la, lb) + (w1 + w2);
        if (cost > nf - min_saving || cost >= best_cost) continue;
        found = true;
        best_cost = cost;
        const int big = la >= lb ? 0 : 1;  // lane 0 is the longer one
        for (int t = 0; t < nf; ++t) cls[t] = comp[t] < 0 ? 2 : (lane_of[comp[t]] == big ? 0 : 1);
      }
  return found;
}

// m0: lower-triangular tile adjacency (nt x nt, natural order, not closed under fill); tiles >= first_dense (T_l_w block, rhs
// row, padding) stay last, in place.  The free tiles are bisected, and each lane once more when that shortens it (a band of
// four tiles with two sub-diagonals: ends first, middle after): the order is lane 0 (its parts, its separator), lane 1, the
// top separator, the tail.  Which block columns can then be factored side by side is read off the filled structure
// (level_schedule); perm[t] = position of natural tile t; lanes at positions [0, lane_a) and [lane_a, lane_a + lane_b).
// Returns false (natural order) unless the top-level bisection shortens the chain by two or more.
inline bool plan_dissection(int nt, int first_dense, const unsigned char* m0, int* perm, int* lane_a, int* lane_b, bool nested = true)
{
  const int nf = std::min(first_dense, nt);
  auto adj = [&](int a, int e) { return a == e ? false : (a > e ? m0[a * nt + e] : m0[e * nt + a]) != 0; };
  std::vector<int> ids(nf), cls;
  for (int t = 0; t < nf; ++t) ids[t] = t;
  if (!bisect_tiles(ids, adj, 6, 2, cls)) return false;
  int pos = 0, n_lane[2] = {0, 0};
  for (int lane = 0; lane < 2; ++lane) {
    std::vector<int> mine, sub;
    for (int t = 0; t < nf; ++t) if (cls[t] == lane) mine.push_back(t);
    n_lane[lane] = (int)mine.size();
    if (nested && bisect_tiles(mine, adj, 4, 1, sub)) {
      for (int pass = 0; pass < 3; ++pass)
        for (size_t k = 0; k < mine.size(); ++k) if (sub[k] == pass) perm[mine[k]] = pos++;
    }
    else for (int t : mine) perm[t] = pos++;
  }
  for (int t = 0; t < nf; ++t) if (cls[t] == 2) perm[t] = pos++;
  for (int t = nf; t < nt; ++t) perm[t] = pos++;
  *lane_a = n_lane[0]; *lane_b = n_lane[1];
  return true;
}

// Steps of the factorisation from the FILLED tile structure m (nt x nt, lower, in elimination order): block column t can be
// factored once every column k < t with m[t][k] != 0 is done, so its step is one more than the largest of theirs; columns of
// one step neither depend on nor couple with each other.  sched[step * CHOL_STEP_COLS + slot] = column or -1, ascending inside
// a step.  A step with more than CHOL_STEP_COLS columns spills its extra columns into later steps (still correct: a column
// may always be factored later than its earliest step as long as its dependants move with it -- so the levels are recomputed
// with the spilled columns' new steps).  Returns the number of steps.
inline int level_schedule(int nt, const unsigned char* m, int* sched)
{
  std::vector<int> level(nt, 0), used(nt + 1, 0);
  for (int t = 0; t < nt; ++t) {
    int lv = 0;
    for (int k = 0; k < t; ++k)
      if (m[t * nt + k]) lv = std::max(lv, level[k] + 1);
    while (used[lv] >= CHOL_STEP_COLS) ++lv;  // (levels are at most nt - 1: column t has at most t predecessors)
    level[t] = lv;
    ++used[lv];
  }
  int steps = 0;
  for (int t = 0; t < nt; ++t) steps = std::max(steps, level[t] + 1);
  for (int i = 0; i < nt * CHOL_STEP_COLS; ++i) sched[i] = -1;
  std::vector<int> fill(steps, 0);
  for (int t = 0; t < nt; ++t) sched[level[t] * CHOL_STEP_COLS + fill[level[t]]++] = t;
  return steps;
}

// threads of a k_schur workgroup = the most runs a camera's entries are cut into (ptz_ba_kernels.h schur_threads<TYPE>())
inline int schur_threads_of(int factor_type) { (void)factor_type; return 256; }

// NW: camera columns with a 2D-2D Jacobian (Dims<TYPE>::NW); max_ent: entries of the largest camera
inline size_t schur_lds_bytes(int max_obs, int NC, int np, bool legacy, int threads, int NW = 0, int max_ent = 0, int row = 0)
{
  // legacy (k_schur_w): T rows of the largest camera, the reduction strip, the scene's tile order (one int per 64 columns)
  if (legacy) return sizeof(double) * ((size_t)max_obs * NC * 3 + (size_t)(SCHUR_THREADS / 64) * (NC + NC * (NC + 1) / 2) + (size_t)(np / CHOL_NB + 1) / 2 + 1);
  // k_schur: the reduction strip, the tile order, then ONE region that first holds, per observation of the largest camera, its T'
  // row and the ray's direction (NW * 3 + 3 doubles rounded up to an odd count, NW <= NC) and later one sum per run (NW^2 | 1)
  if (NW <= 0) NW = NC;
  const size_t table = (size_t)max_obs * (row > 0 ? row : ((NW * 3 + 3) | 1)), sums = (size_t)threads * ((NW * NW) | 1);  // row > 0: k_schur_f's factored rows
  return sizeof(double) * ((size_t)(threads / 64) * (NW + NW * (NW + 1) / 2) + (size_t)(np / CHOL_NB + 2) / 2 + std::max(table, sums) + 2) +
         2 * (size_t)((max_ent + 3) & ~3);
}

// Every host <-> device copy of a batch goes through a stream of its own and waits for THAT stream only.  hipMemcpy / hipMemset
// run on the null stream, which synchronises with every other (blocking) stream of the process: batches driven from several
// host threads (the lock-step PTZ-IBA, ptz_ba_solve_sharded's dealers) then wait for each other's solves inside their copies.
inline hipError_t copy_on(hipStream_t st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
  if (bytes == 0) return hipSuccess;
  // small copies go through a pinned block of the pool: an asynchronous copy from / to pageable memory waits inside the runtime
  // (the same interrupt-driven wait stream_wait() avoids); large ones keep the direct path, their transfer time dominates
  void* pin = nullptr;
  if (bytes <= ((size_t)16 << 20) && ptzpool::pinned_acquire(bytes, &pin) == hipSuccess) {
    hipError_t e;
    if (kind == hipMemcpyHostToDevice) {
      memcpy(pin, src, bytes);
      e = hipMemcpyAsync(dst, pin, bytes, kind, st);
      if (e == hipSuccess) e = stream_wait(st);
    }
    else {
      e = hipMemcpyAsync(pin, src, bytes, kind, st);
      if (e == hipSuccess) e = stream_wait(st);
      if (e == hipSuccess) memcpy(dst, pin, bytes);
    }
    ptzpool::pinned_release(pin);
    return e;
  }
  (void)hipGetLastError();
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  if (e == hipSuccess) e = stream_wait(st);
  return e;
}

template <typename T> int upload(ptz_ba_batch* b, const std::vector<T>& h, const T** dev)
{
  T* p = nullptr;
  int rc = b->alloc(&p, h.size());
  if (rc) return rc;
  if (copy_on(b->io, p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return PTZ_ENODEVICE;
  *dev = p;
  return PTZ_OK;
}

// Many small host arrays -> ONE device block through ONE pinned staging buffer and one copy (a synchronous hipMemcpy from
// pageable memory costs 10-15 us each whatever its size; a batch has about twenty of them).  Large batches, whose structure
// would need a staging buffer of more than 64 MB, keep the per-array uploads.
// std::vector whose resize() leaves trivially-constructible elements uninitialised (the batch-wide observation arrays are
// sized once and then filled in place by the builder threads; zero-filling 600 MB first costs more than building them)
template <typename T> struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using RawVec = std::vector<T, NoInitAlloc<T>>;

struct StagedUpload {
  struct Item { const void* src; size_t bytes; const void** dst; };
  std::vector<Item> items;
  template <typename T, typename A> void add(const std::vector<T, A>& h, const T** dev)
  {
    items.push_back({h.data(), sizeof(T) * h.size(), reinterpret_cast<const void**>(dev)});
  }
  // wait = false: the copy is left in flight on b->io (kernels enqueued there see the data; the caller's next wait for that stream
  // -- followed by b->release_staged() -- ends it)
  int commit(ptz_ba_batch* b, bool wait = true)
  {
    auto up = [](size_t x) { return (std::max<size_t>(x, 1) + 255) & ~(size_t)255; };
    size_t total = 0;
    for (const Item& it : items) total += up(it.bytes);
    void* pinned = nullptr;
    if (total > ((size_t)64 << 20) || ptzpool::pinned_acquire(total, &pinned) != hipSuccess) {
      (void)hipGetLastError();
      for (const Item& it : items) {
        char* p = nullptr;
        int rc = b->alloc(&p, it.bytes);
        if (rc) return rc;
        if (copy_on(b->io, p, it.src, it.bytes, hipMemcpyHostToDevice) != hipSuccess) return PTZ_ENODEVICE;
        *it.dst = p;
      }
      return PTZ_OK;
    }
    char* dev = nullptr;
    int rc = b->alloc(&dev, total);
    if (rc) { ptzpool::pinned_release(pinned); return rc; }
    size_t off = 0;
    for (const Item& it : items) {
      if (it.bytes) memcpy((char*)pinned + off, it.src, it.bytes);
      *it.dst = dev + off;
      off += up(it.bytes);
    }
    hipError_t e = hipMemcpyAsync(dev, pinned, total, hipMemcpyHostToDevice, b->io);
    if (e == hipSuccess && !wait) { b->staged_pinned.push_back(pinned); return PTZ_OK; }
    if (e == hipSuccess) e = stream_wait(b->io);
    ptzpool::pinned_release(pinned);
    return e == hipSuccess ? PTZ_OK : PTZ_ENODEVICE;
  }
};

#define LAUNCH(kern, grid, block, smem, ...) ptz::launch(kern, grid, block, smem, b->stream, __VA_ARGS__)
// the ray-centric kernels come in four shapes: small / large workgroups x camera tables staged in LDS / read from global memory
#define PTZ_LAUNCH_RAY(kern, grid, smem, dev)                                                                       \
  do {                                                                                                              \
    if (sh.small_blocks) {                                                                                          \
      if (b->gtab) LAUNCH((kern<TYPE, true, true>), grid, dim3(sh.ray_block), smem, dev);                           \
      else LAUNCH((kern<TYPE, true, false>), grid, dim3(sh.ray_block), smem, dev);                                  \
    }                                                                                                               \
    else {                                                                                                          \
      if (b->gtab) LAUNCH((kern<TYPE, false, true>), grid, dim3(sh.ray_block), smem, dev);                          \
      else LAUNCH((kern<TYPE, false, false>), grid, dim3(sh.ray_block), smem, dev);                                 \
    }                                                                                                               \
  } while (0)

// k_pairs' dynamic LDS beyond 64 KiB (very wide rigs): the attribute is set ONCE per device (hipFuncSetAttribute under a lock on every
// batch creation was a measurable share of a view batch's 0.4 ms of enqueueing in the 64-rig lock step)
static void pairs_lds_cap(int device)
{
  static std::mutex mu;
  static bool done[64] = {};
  std::lock_guard<std::mutex> lk(mu);
  if (done[device & 63]) return;
  (void)hipFuncSetAttribute((const void*)k_pairs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  (void)hipFuncSetAttribute((const void*)k_pairs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  done[device & 63] = true;
}

template <int TYPE> void enqueue_linearize(ptz_ba_batch* b)
{
  const Dev& d = b->d;
  const PassShape& sh = b->shapes[0];
  b->prof_begin(P_LIN);
  LAUNCH(k_cam_prep<TYPE>, dim3((b->max_cam + 63) / 64, b->n_scene), dim3(64), 0, d);
  PTZ_LAUNCH_RAY(k_lin_ray, dim3(sh.max_chunk, b->n_scene), sh.lin_smem, d);
  if (d.W) LAUNCH((k_lin_cam<TYPE, true>), dim3((b->max_cam + 3) / 4, b->n_scene), dim3(256), 0, d);  // parity tests / legacy Schur path
  else LAUNCH((k_lin_cam<TYPE, false>), dim3((b->max_cam + 3) / 4, b->n_scene), dim3(256), 0, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_lin_3d<TYPE>, dim3(b->n_scene), dim3(256), 0, d);
  if (d.shared) LAUNCH(k_group_grad<TYPE>, dim3(b->n_scene), dim3(256), 0, d);
  b->prof_end();
}

static void make_groups(ptz_ba_batch* b)
{
  const int B = b->n_scene;
  // per-family timings are only meaningful when no other group's kernels share the device: profiling runs one group
  const int G = b->profiling ? 1 : std::max(1, std::min(std::min(b->n_group, B), b->ctl_groups));
  b->group_first.clear(); b->group_count.clear(); b->dg.clear();
  for (int g = 0; g < G; ++g) {
    const int lo = (int)((int64_t)B * g / G), hi = (int)((int64_t)B * (g + 1) / G);
    b->group_first.push_back(lo);
    b->group_count.push_back(hi - lo);
    Dev d = b->d;  // per-scene arrays are re-based; everything else is addressed through SceneDev offsets
    const size_t np = d.chol.np, nt = np / CHOL_NB;
    d.n_scene = hi - lo;
    d.scene += lo; d.lm += lo; d.active += lo; d.ray_fail += lo; d.tail_cnt += 2 * (size_t)lo;
    d.yc += (size_t)lo * np;
    d.chol.count = hi - lo;
    d.chol.A += (size_t)lo * np * np;
    if (d.chol.Linv) d.chol.Linv += (size_t)lo * nt * CHOL_NB * CHOL_NB;
    d.chol.Ldiag += (size_t)lo * nt * CHOL_NB * CHOL_NB;
    d.chol.Dinv += (size_t)lo * nt * 4 * 16 * 16;
    d.chol.n += lo; d.chol.fail += lo; d.chol.active = d.active;
    if (d.chol.tmask) d.chol.tmask += (size_t)lo * nt * nt;
    if (d.chol.sched) d.chol.sched += (size_t)lo * nt * CHOL_STEP_COLS;
    if (d.tperm) { d.tperm += (size_t)lo * nt; d.chol.xperm = d.tperm; }
    d.grp_ctl = b->d_ctl + 4 * g;
    d.host_ctl = b->h_ctl_dev + 4 * g;
    d.act = b->d_act + lo;
    d.use_act = 0;
    if (d.chol.L) d.chol.L = b->d.chol.L + (size_t)g * std::min(b->n_scene, CHOL_CHAIN_SLOTS) * np * np;  // slots of finished L tiles per group
    if (d.chol.chain_ctl) d.chol.chain_ctl = b->d.chol.chain_ctl + (size_t)g * chol_chain_ctl_ints((int)np);  // (a block per stream)
    if (d.chol.bs_items) { d.chol.bs_items += (size_t)lo * 4 * chol_backsolve_max_groups((int)np); d.chol.bs_groups += lo; }
    b->dg.push_back(d);
  }
  if ((int)b->pass_graph.size() != G || (G > 0 && b->pass_graph[0].size() != b->shapes.size())) {  // the grouping changed (profiling on / off): recorded passes are stale
    for (auto& v : b->pass_graph) for (auto ge : v) if (ge) (void)hipGraphExecDestroy(ge);
    b->pass_graph.assign(G, std::vector<hipGraphExec_t>(b->shapes.size(), nullptr));
  }
  while ((int)b->streams.size() < G) {
    hipStream_t st; (void)ptzpool::stream_acquire(b->device, &st); b->streams.push_back(st);
    hipEvent_t e1, e2; (void)ptzpool::event_acquire(b->device, false, &e1); (void)ptzpool::event_acquire(b->device, false, &e2);
    b->fork_ev.push_back(e1); b->join_ev.push_back(e2);
    hipStream_t ax; (void)ptzpool::stream_acquire(b->device, &ax); b->aux.push_back(ax);
    hipEvent_t e3, e4; (void)ptzpool::event_acquire(b->device, false, &e3); (void)ptzpool::event_acquire(b->device, false, &e4);
    b->la_ev.push_back(e3); b->la_ev.push_back(e4);
  }
}

// one LM pass of one group in launch shape `sh`, enqueued on b->stream; returns after enqueueing (no synchronisation).
// A pass = one trust-region step and the bookkeeping that OPENS the next one (k_lm_pre last: the first iteration is opened by
// solve_impl's prologue), so that launch shapes with and without launches of their own for LM control can follow one another.
template <int TYPE> void enqueue_pass(ptz_ba_batch* b, const Dev& dgrp, const PassShape& sh)
{
  constexpr int NC = Dims<TYPE>::NC;
  Dev d = dgrp;
  d.use_act = sh.compact ? 1 : 0;
  d.ray_block = sh.ray_block;
  d.chol.count = sh.slots;
  d.chol.act = sh.compact ? d.act : nullptr;
  d.chol.act_n = d.grp_ctl + 2;
  if (!sh.fused) d.chol.L = nullptr;  // (the second matrix marks the one-launch-per-column path)
  // Scenes without annotation residuals and shared blocks: the camera-side linearisation of the CANDIDATE is evaluated before the
  // step is judged (k_lin_cam, Dev::spec_lin), and one control point per pass judges the step and opens the next iteration
  // (lm_step_wave) -- a launch of its own (k_lm_step), or in launch shapes of a few scenes the tail of k_lin_cam (Dev::fuse_ctl, with
  // the camera update in k_eval's prologue).  The others keep k_lm_post / k_lin_cam at the accepted point / k_lm_pre.
  // (Not for large launch shapes: there the speculative linearisation of the steps that end up rejected costs more than the second
  //  control launch it saves -- C4: linearise 19 -> 27 ms per solve against 3 ms of control.  Both forms take the same decisions
  //  with the same bits, so a scene may change between them from pass to pass.  Round 2's Schur kernel reads W rows of the
  //  CURRENT point, which are not double-buffered.)
  const bool fast = TYPE < 3 && !d.shared && !b->schur_w && sh.fuse_ctl;
  const bool fuse = sh.fuse_ctl && fast;
  d.spec_lin = fast ? 1 : 0;
  d.fuse_ctl = fuse ? 1 : 0;
  const int B = sh.slots;
  hipStream_t st = b->stream;
  const int schur_thr = schur_threads<TYPE>();
  const size_t schur_smem = schur_lds_bytes(b->schur_tg ? 0 : b->max_cam_obs, NC, d.chol.np, b->schur_w, schur_thr, Dims<TYPE>::NW, b->max_cam_ent,
                                            b->schur_f ? SCHUR_F_ROW : 0);
  b->prof_begin(P_RAYPREP);
  {
    const int nt = d.chol.np / CHOL_NB;
    const int per = std::max(1, sh.ray_block / 256);  // tiles of the lower triangle a workgroup clears
    if (d.chol.tmask) LAUNCH(k_ray_prep<TYPE>, dim3(sh.max_chunk + (nt * (nt + 1) / 2 + per - 1) / per, B), dim3(sh.ray_block), 0, d, sh.max_chunk);
    else {  // dense debugging path (PTZ_BA_DENSE_CHOL): whole matrices zeroed by a memset
      LAUNCH(k_ray_prep<TYPE>, dim3(sh.max_chunk, B), dim3(sh.ray_block), 0, d, sh.max_chunk);
      chol_clear(d.chol, st);
    }
  }
  if (d.shared) LAUNCH(k_group_diag<TYPE>, dim3(b->max_grp * NC, B), dim3(64), 0, d);
  b->prof_end();
  b->prof_begin(P_SCHUR);
  if (b->schur_w) {
    if (b->schur_tg) LAUNCH((k_schur_w<TYPE, true>), dim3(b->max_cam, B), dim3(SCHUR_THREADS), schur_smem, d);
    else LAUNCH((k_schur_w<TYPE, false>), dim3(b->max_cam, B), dim3(SCHUR_THREADS), schur_smem, d);
  }
  else if (b->schur_f) {
    if constexpr (Dims<TYPE>::FACTOR == 0) {
      if (b->schur_tg) LAUNCH((k_schur_f<TYPE, true>), dim3(b->max_cam, B), dim3(256), schur_smem, d);
      else LAUNCH((k_schur_f<TYPE, false>), dim3(b->max_cam, B), dim3(256), schur_smem, d);
    }
  }
  else if (b->schur_tg) LAUNCH((k_schur<TYPE, true>), dim3(b->max_cam, B), dim3(schur_thr), schur_smem, d);
  else LAUNCH((k_schur<TYPE, false>), dim3(b->max_cam, B), dim3(schur_thr), schur_smem, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_schur_3d<TYPE>, dim3(B), dim3(64), 0, d);
  if (d.shared) LAUNCH(k_fold_system<TYPE>, dim3(B), dim3(1024), sizeof(double) * (size_t)(b->max_n + 4) + (size_t)(d.chol.np / CHOL_NB) * (d.chol.np / CHOL_NB), d);
  b->prof_end();
  chol_factor_solve_profiled(d.chol, d.yc, st, b, sh.fused);
  if (d.shared) LAUNCH(k_group_expand<TYPE>, dim3(B), dim3(256), 0, d);
  if (!fuse) {
    b->prof_begin(P_BACKSUB);
    LAUNCH(k_cam_update<TYPE>, dim3((b->max_cam + 63) / 64, B), dim3(64), 0, d);
    b->prof_end();
  }
  b->prof_begin(P_EVAL);
  if constexpr (TYPE < 3) {
    if (fuse && sh.eval_lanes == 4) LAUNCH((k_eval<TYPE, true, false, true, 4>), dim3(sh.max_chunk, B), dim3(sh.ray_block * 4), sh.eval_smem, d);
    else if (fuse) LAUNCH((k_eval<TYPE, true, false, true>), dim3(sh.max_chunk, B), dim3(sh.ray_block), sh.eval_smem, d);
    else PTZ_LAUNCH_RAY(k_eval, dim3(sh.max_chunk, B), sh.eval_smem, d);
  }
  else PTZ_LAUNCH_RAY(k_eval, dim3(sh.max_chunk, B), sh.eval_smem, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_eval_3d<TYPE>, dim3(B), dim3(256), 0, d);
  b->prof_end();
  if (!fast) {
    b->prof_begin(P_LMCTL);
    LAUNCH(k_lm_post<TYPE>, dim3(B), dim3(LM_THREADS), 0, d);
    b->prof_end();
  }
  {
    const Dev& dd = d;
    b->prof_begin(P_LIN);
    // (no k_lin_ray inside a pass: the ray side of an accepted step's linearisation was left by k_eval's second pass,
    //  LmState::ray_lin_ready; k_lin_ray runs for iteration zero only, enqueue_linearize)
    if (dd.W && b->schur_w) LAUNCH((k_lin_cam<TYPE, true>), dim3((b->max_cam + 3) / 4, B), dim3(256), 0, dd);
    else LAUNCH((k_lin_cam<TYPE, false>), dim3((b->max_cam + 3) / 4, B), dim3(256), 0, dd);
    if (Dims<TYPE>::HAS3D) LAUNCH(k_lin_3d<TYPE>, dim3(B), dim3(256), 0, dd);
    if (dd.shared) LAUNCH(k_group_grad<TYPE>, dim3(B), dim3(256), 0, dd);
    b->prof_end();
  }
  b->prof_begin(P_LMCTL);
  if (!fast) LAUNCH(k_lm_pre<TYPE>, dim3(B), dim3(LM_THREADS), 0, d);
  else if (!fuse) LAUNCH(k_lm_step<TYPE>, dim3(B), dim3(LM_THREADS), 0, d);
  if (b->shapes.size() > 1 && b->compaction) LAUNCH(k_compact, dim3(1), dim3(1024), 0, d);  // (batches too small for a compacted shape skip it)
  b->prof_end();
}

template <int TYPE> int solve_impl(ptz_ba_batch* b, ptz_lm_summary* out)
{
  constexpr int NC = Dims<TYPE>::NC;
  const Dev& d = b->d;
  const int B = b->n_scene;
  const bool dbg = getenv("PTZ_BA_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  double t_enq = 0, t_sync = 0;
  make_groups(b);
  const int G = (int)b->dg.size();
  hipStream_t s0 = b->streams[0];
  b->stream = s0;
  for (int i = 0; i < 4 * b->ctl_groups; ++i) __atomic_store_n(&b->h_ctl[i], 0, __ATOMIC_RELEASE);  // nothing of this batch is in flight
  for (int g = 0; g < G; ++g) __atomic_store_n(&b->h_ctl[4 * g + 2], b->group_count[g], __ATOMIC_RELEASE);
  PTZ_HIP_TRY(hipEventRecord(b->ev0, s0));
  // x <- initial state, scales <- 1, LM state reset (whole batch, stream 0)
  {
    const size_t n15 = (size_t)15 * b->total_cam, n3 = (size_t)3 * b->total_ray, nnc = (size_t)b->total_cam * NC;
    const size_t most = std::max(std::max(n15, n3), std::max(std::max(nnc, (size_t)6 * B), d.dsp_x ? d.dsp_stride : (size_t)0));
    LAUNCH(k_solve_init, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, d, (const double*)b->cam0, (const double*)b->ray0, (const double*)b->dsp0,
           (const double*)b->tlw0, n15, n3, nnc);
  }
  // IterationZero: evaluate, Jacobi scales from the column norms, re-evaluate scaled
  enqueue_linearize<TYPE>(b);
  if (b->opt.jacobi_scaling) {
    LAUNCH(k_jacobi_scale<TYPE>, dim3((std::max(b->max_cam, b->max_ray) + 255) / 256, B), dim3(256), 0, d);
    if (d.shared) LAUNCH(k_group_scale<TYPE>, dim3(b->max_grp * NC, B), dim3(64), 0, d);
    enqueue_linearize<TYPE>(b);
  }
  for (int g = 0; g < G; ++g) hipLaunchKernelGGL(k_ctl_reset, dim3(1), dim3(64), 0, s0, b->dg[g]);
  // iteration zero's books are closed and the first step is opened here; every pass ends with the same bookkeeping for the next
  b->prof_begin(P_LMCTL);
  for (int g = 0; g < G; ++g) {
    LAUNCH(k_lm_pre<TYPE>, dim3(b->group_count[g]), dim3(LM_THREADS), 0, b->dg[g]);
    if (b->shapes.size() > 1 && b->compaction) LAUNCH(k_compact, dim3(1), dim3(1024), 0, b->dg[g]);
  }
  b->prof_end();
  // fork: every group's stream continues after the common prologue
  PTZ_HIP_TRY(hipEventRecord(b->fork_ev[0], s0));
  for (int g = 1; g < G; ++g) PTZ_HIP_TRY(hipStreamWaitEvent(b->streams[g], b->fork_ev[0], 0));
  const int max_it = b->opt.max_num_iterations;
  // Every group runs its own pass pipeline and the host is not part of it: LM control lives on the device (k_lm_pre /
  // k_lm_post), every kernel of a pass returns at once for a scene that has terminated, so the host simply keeps `ahead`
  // passes enqueued beyond the last one the device is known to have reached, and stops when the device reports that the
  // group's last scene has retired (two words in pinned memory per group, polled; no stream synchronisation, no copy).
  // Passes enqueued past that point are empty launches.  Groups drift out of phase, so the latency-bound part of one
  // group's pass (block-column chain of the factorisation, LM control) overlaps the throughput kernels of another.
  // (a batch's FIRST solve enqueues its passes eagerly: recording and instantiating a graph costs more than it saves when the
  // batch is solved once, as every bundle adjustment of the incremental pipeline is)
  const bool graph = b->use_graph && !b->profiling && !b->lookahead && b->d.chol.tmask != nullptr && b->n_solves > 0;
  ++b->n_solves;
  // Passes per replayed graph.  Between two graph launches on a stream the device idles ~8.4 us (kernel trace of the one-rig solve:
  // the six kernels of a pass abut, the next pass's first kernel starts 8.4-8.8 us behind the last) -- 4 % of a one-rig pass.  Launch
  // shapes of a few scenes therefore replay graphs of SEVERAL passes (PTZ_BA_GRAPH_PASSES; the passes past a scene's end are the empty
  // launches they always were: at most that many more of them per solve).
  static const int graph_passes_few = [] { const char* e = getenv("PTZ_BA_GRAPH_PASSES"); return e ? std::max(1, std::min(16, atoi(e))) : 1; }();
  auto graph_passes = [&](const PassShape& sh) { return sh.slots <= 8 ? graph_passes_few : 1; };
  auto graph_of = [&](int g, int si) -> hipGraphExec_t {  // built on first use: one node per launch, a linear chain
    if (b->pass_graph[g][si]) return b->pass_graph[g][si];
    GraphRecorder rec;
    bool ok = hipGraphCreate(&rec.graph, 0) == hipSuccess;
    if (ok) {
      b->stream = b->streams[g];
      g_recorder = &rec;
      PassShape shg = b->shapes[si];
      if (si == 0) shg.slots = b->group_count[g];
      for (int rep = 0; rep < graph_passes(shg); ++rep) enqueue_pass<TYPE>(b, b->dg[g], shg);
      g_recorder = nullptr;
      ok = rec.ok && hipGraphInstantiate(&b->pass_graph[g][si], rec.graph, nullptr, nullptr, 0) == hipSuccess;
    }
    if (rec.graph) (void)hipGraphDestroy(rec.graph);
    if (!ok) { (void)hipGetLastError(); b->pass_graph[g][si] = nullptr; b->use_graph = false; }
    return b->pass_graph[g][si];
  };
  std::vector<char> galive(G, 1);
  std::vector<int> enq(G, 0), shape_used;
  // Watchdog.  The loop below only ever waits on words the DEVICE writes; if a launch was refused, a kernel faulted or the
  // progress word stops moving for any other reason, nothing would ever clear the wait.  So whenever no group has made
  // progress for `watchdog_ms`, every waiting group's stream is asked for its state: an error ends the solve with
  // PTZ_ENODEVICE; an IDLE stream whose passes have all run without the progress word catching up is credited with them
  // (`credit`) and fed further passes -- the pipeline degrades to enqueue-and-wait, results unchanged -- and a group that needs
  // that more often than a solve has passes is given up on.  Never a spin without end, never a re-exec.
  std::vector<int> credit(G, 0), stalls(G, 0);
  double watchdog_ms = 50.0;
  if (const char* e = getenv("PTZ_BA_WATCHDOG_MS")) watchdog_ms = std::max(0.1, atof(e));
  double t_progress = now();
  int watchdog_rc = PTZ_OK;
  int alive = G;
  while (alive > 0) {
    bool progressed = false;
    for (int g = 0; g < G; ++g) {
      if (!galive[g]) continue;
      if (__atomic_load_n(&b->h_ctl[4 * g + 1], __ATOMIC_ACQUIRE)) { galive[g] = 0; --alive; progressed = true; continue; }
      if (enq[g] >= max_it) { galive[g] = 0; --alive; progressed = true; continue; }  // every step the options allow is enqueued (the last pass's k_lm_pre closes the books)
      if (enq[g] - std::max(__atomic_load_n(&b->h_ctl[4 * g], __ATOMIC_ACQUIRE), credit[g]) >= b->ahead) continue;
      const double te0 = now();
      b->stream = b->streams[g];
      const bool last = false;
      // launch shape: the smallest one that covers the scenes last reported active (the count only ever decreases, so a stale
      // value is an upper bound); full size while more than the largest compacted shape are
      int si = 0;
      if (b->compaction && !last) {
        const int cnt = __atomic_load_n(&b->h_ctl[4 * g + 2], __ATOMIC_ACQUIRE);
        for (int k = 1; k < (int)b->shapes.size(); ++k)
          if (b->shapes[k].slots >= cnt && b->shapes[k].slots < b->group_count[g]) { si = k; break; }
      }
      PassShape sh = b->shapes[si];
      if (si == 0) { sh.slots = b->group_count[g]; }
      if (dbg) { if (shape_used.size() < b->shapes.size()) shape_used.resize(b->shapes.size(), 0); ++shape_used[si]; }
      // (one or two rigs: the passes are enqueued as they are -- a pass is ~200 us of device time against ~25 us of host time for its
      //  six launches, and between two REPLAYED graphs the device idles 8.4 us: 5.20 -> 5.09 ms per 25-iteration solve of the C2 rig,
      //  A/B on one box; PTZ_BA_GRAPH_FEW=1 replays graphs there too)
      static const bool graph_few = [] { const char* e = getenv("PTZ_BA_GRAPH_FEW"); return e && atoi(e) != 0; }();
      hipGraphExec_t ge = (!last && graph && b->use_graph && (sh.slots > 2 || graph_few)) ? graph_of(g, si) : nullptr;
      if (ge) PTZ_HIP_TRY(hipGraphLaunch(ge, b->streams[g]));
      else { b->stream = b->streams[g]; enqueue_pass<TYPE>(b, b->dg[g], sh); }
      enq[g] += ge ? graph_passes(sh) : 1;
      progressed = true;
      t_enq += now() - te0;
    }
    if (progressed) { t_progress = now(); continue; }
    const double ts0 = now();
    __builtin_ia32_pause();
    if (ts0 - t_progress > watchdog_ms) {
      for (int g = 0; g < G && watchdog_rc == PTZ_OK; ++g) {
        if (!galive[g]) continue;
        const hipError_t q = hipStreamQuery(b->streams[g]);
        if (q == hipErrorNotReady) continue;  // still working: a long pass, not a stall
        const hipError_t le = hipGetLastError();
        if (q != hipSuccess || le != hipSuccess) {
          fprintf(stderr, "[ptz_ba] scene group %d: the device reported %s after %d enqueued LM passes; giving up on this solve\n", g,
                  hipGetErrorName(q != hipSuccess ? q : le), enq[g]);
          watchdog_rc = PTZ_ENODEVICE;
          break;
        }
        // idle and healthy: everything enqueued has run.  Retired meanwhile?  Then the next sweep sees it.
        if (__atomic_load_n(&b->h_ctl[4 * g + 1], __ATOMIC_ACQUIRE)) continue;
        if (__atomic_load_n(&b->h_ctl[4 * g], __ATOMIC_ACQUIRE) >= enq[g]) continue;  // the word caught up
        if (++stalls[g] > max_it + 8) {
          fprintf(stderr, "[ptz_ba] scene group %d: no progress reports from the device (%d enqueued passes ran to completion without one); giving up on this solve\n", g, enq[g]);
          watchdog_rc = PTZ_ENODEVICE;
          break;
        }
        credit[g] = enq[g];
      }
      if (watchdog_rc != PTZ_OK) break;
      t_progress = now();
    }
    t_sync += now() - ts0;
  }
  if (watchdog_rc != PTZ_OK) {
    for (int g = 0; g < G; ++g) (void)stream_wait(b->streams[g]);  // nothing of this batch may be in flight when the caller frees it
    (void)hipGetLastError();
    return watchdog_rc;
  }
  if (dbg) for (size_t k = 0; k < shape_used.size(); ++k) fprintf(stderr, "[ptz_ba] launch shape %zu (%d slots%s): %d passes\n", k, b->shapes[k].slots, b->shapes[k].compact ? ", compacted" : "", shape_used[k]);
  if (dbg) for (int g = 0; g < G; ++g) fprintf(stderr, "[ptz_ba] group %d: %d passes enqueued, %d reached by the device when the host stopped\n", g, enq[g], b->h_ctl[4 * g]);
  // join
  for (int g = 1; g < G; ++g) {
    PTZ_HIP_TRY(hipEventRecord(b->join_ev[g], b->streams[g]));
    PTZ_HIP_TRY(hipStreamWaitEvent(s0, b->join_ev[g], 0));
  }
  b->stream = s0;
  PTZ_HIP_TRY(hipEventRecord(b->ev1, s0));
  // the scenes' LM states come back behind the last pass, inside the solve's ONE wait
  void* lm_pin = nullptr;
  if (ptzpool::pinned_acquire(sizeof(LmState) * B, &lm_pin) == hipSuccess) {
    if (hipMemcpyAsync(lm_pin, d.lm, sizeof(LmState) * B, hipMemcpyDeviceToHost, s0) != hipSuccess) { (void)hipGetLastError(); ptzpool::pinned_release(lm_pin); lm_pin = nullptr; }
  }
  else (void)hipGetLastError();
  {
    const hipError_t we = stream_wait(s0);
    b->release_staged();  // (uploads of create / set_state that were left in flight)
    if (we != hipSuccess) { if (lm_pin) ptzpool::pinned_release(lm_pin); PTZ_HIP_TRY(we); }
  }
  {
    const hipError_t le = hipGetLastError();  // a kernel launch that was refused (resources, arguments) must not pass for a solve
    if (le != hipSuccess) { if (lm_pin) ptzpool::pinned_release(lm_pin); PTZ_HIP_TRY(le); }
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, b->ev0, b->ev1);
  b->last_ms = ms;
  b->prof_collect();
  if (dbg) fprintf(stderr, "[ptz_ba] groups %d: total %.2f ms, enqueue %.2f ms, sync-wait %.2f ms, device %.2f ms\n", G, now() - t_start, t_enq, t_sync, ms);
  {
    std::vector<LmState> h(B);
    if (lm_pin) { memcpy(h.data(), lm_pin, sizeof(LmState) * B); ptzpool::pinned_release(lm_pin); }
    else PTZ_HIP_TRY(copy_on(b->stream, h.data(), d.lm, sizeof(LmState) * B, hipMemcpyDeviceToHost));
    int timeouts = 0;
    for (int i = 0; i < B; ++i) timeouts += h[i].chain_timeouts;
    if (timeouts) {  // never a silently different trajectory: a hand-over that did not arrive is a device problem, not a rejected step
      fprintf(stderr, "[ptz_ba] %d linear solve(s) of this batch lost a tile hand-over of the one-launch factorisation (bounded wait ran out); "
                      "the solve is reported as failed\n", timeouts);
      return PTZ_ENODEVICE;
    }
    for (int i = 0; out && i < B; ++i) {
      ptz_lm_summary& s = out[i];
      s.termination_type = h[i].termination;
      s.num_iterations = h[i].n_summaries - 1;
      s.num_lm_steps = h[i].num_lm_steps;
      s.num_successful_steps = h[i].num_successful;
      s.num_unsuccessful_steps = h[i].num_unsuccessful;
      s.num_residuals = 2 * b->scenes[i].n_obs + 2 * b->scenes[i].n_o3;
      s.num_linear_solves = h[i].num_linear_solves;
      s.num_jacobian_evals = h[i].num_jac_evals;
      s.initial_cost = h[i].initial_cost;
      s.final_cost = h[i].final_cost;
      s.final_radius = h[i].radius;
      s.final_gradient_max_norm = h[i].grad_max;
    }
  }
  return PTZ_OK;
}

}  // namespace

namespace ptz {
// defined here (needs ptz_ba_batch) but uses the kernels of ptz_chol.hip through chol_factor_solve pieces
void chol_factor_solve_profiled(const CholBatch& cb, double* x, hipStream_t stream, void* prof, bool fused)
{
  // One-step look-ahead: after the triangular solve of block column k, the small update of block column k+1 stays on
  // the main stream, so the (latency-bound) diagonal factorisation and triangular solve of step k+1 start at once,
  // while the bulk of the trailing update (tile columns >= k+2) runs on the auxiliary stream.
  ptz_ba_batch* b = (ptz_ba_batch*)prof;
  const int nt = cb.np / CHOL_NB;
  hipStream_t aux = b->aux_stream(stream);
  const bool la = aux != nullptr && nt >= 3;
  hipEvent_t evT = nullptr, evR = nullptr;
  if (la) b->lookahead_events(stream, &evT, &evR);
  bool rest_pending = false;
  if (fused) {  // a few scenes: the whole factorisation in one launch (chol_chain_kernel), or one launch per step of the schedule (chol_col_step_kernel)
    if (chol_chain_enabled(cb)) {
      b->prof_begin(P_CHOL_SYRK);
      chol_chain_launch(cb, stream);
      b->prof_end();
    }
    else {
      b->prof_begin(P_CHOL_PANEL);
      chol_diag_launch(cb, -1, stream);
      b->prof_end();
      for (int st = 0; st + 1 < chol_step_count(cb); ++st) {
        b->prof_begin(P_CHOL_SYRK);
        chol_col_step_launch(cb, st, stream);
        b->prof_end();
      }
    }
    b->prof_begin(P_CHOL_BACK);
    chol_backsolve_launch(cb, x, stream);
    b->prof_end();
    return;
  }
  if (b->left_looking) {
    for (int k = 0; k < nt; ++k) {
      if (k > 0) {
        b->prof_begin(P_CHOL_SYRK);
        chol_update_col_launch(cb, k, stream, /*fuse_diag=*/true);  // also factors the diagonal tile of column k
        b->prof_end();
      }
      b->prof_begin(P_CHOL_PANEL);
      chol_panel_launch(cb, k, stream, /*diag_done=*/k > 0);
      b->prof_end();
    }
    b->prof_begin(P_CHOL_BACK);
    chol_tile_inverse_launch(cb, stream);
    chol_backsolve_launch(cb, x, stream);
    b->prof_end();
    return;
  }
  for (int k = 0; k < nt; ++k) {
    const int m = nt - k - 1;
    b->prof_begin(P_CHOL_PANEL);
    chol_panel_launch(cb, k, stream, /*diag_done=*/k > 0);  // the trailing update of step k - 1 factored this diagonal tile
    b->prof_end();
    if (m <= 0) continue;
    if (!la) {
      b->prof_begin(P_CHOL_SYRK);
      chol_syrk_launch(cb, k, stream, 0, /*fuse_diag=*/true);
      b->prof_end();
      continue;
    }
    (void)hipEventRecord(evT, stream);              // panel k (L_ik tiles) is final
    (void)hipStreamWaitEvent(aux, evT, 0);
    if (rest_pending) (void)hipStreamWaitEvent(stream, evR, 0);  // column k+1 was last touched by rest(k-1)
    b->prof_begin(P_CHOL_SYRK);
    chol_syrk_launch(cb, k, stream, 1, /*fuse_diag=*/true);
    b->prof_end();
    if (m >= 2) {
      hipStream_t keep = b->stream;
      b->stream = aux;
      b->prof_begin(P_CHOL_SYRK);
      chol_syrk_launch(cb, k, aux, 2);
      b->prof_end();
      b->stream = keep;
      (void)hipEventRecord(evR, aux);
      rest_pending = true;
    }
  }
  if (la && rest_pending) (void)hipStreamWaitEvent(stream, evR, 0);
  b->prof_begin(P_CHOL_BACK);
  chol_tile_inverse_launch(cb, stream);
  chol_backsolve_launch(cb, x, stream);
  b->prof_end();
}
}  // namespace ptz

// One rig's tracks, resident in HBM (ptz_rig_create), and the bounds a view of it is sized by.
struct ptz_rig {
  int device = 0, n_img = 0, n_track = 0;
  int64_t n_view = 0;
  const int* d_trk_ptr = nullptr;
  const int* d_trk_img = nullptr;
  const float2* d_trk_uv = nullptr;
  int64_t ent_bound = 0;           // camera-pair entries of the whole rig: sum over the tracks of L (L - 1) / 2
  int max_track_len = 0;           // views of the rig's longest track: bounds the length field of a view batch's sort key (ptz_ba_batch_create_views)
  std::vector<int> img_obs, img_ent;  // per image: its views, and its entries as the HIGHER camera of a pair
  std::vector<void*> allocs;
};

// =============================================================================================================
// C-ABI
// =============================================================================================================
extern "C" {

void ptz_lm_options_default(ptz_lm_options* o)
{
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 200;
  o->device_id = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
}

const char* ptz_version(void) { return "ptz-calib_amd 0.1 (gfx950)"; }

int32_t ptz_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t ptz_ba_plan_tile_order(int32_t nt, int32_t first_dense, const uint8_t* mask, int32_t* perm, int32_t* lanes,
                               int32_t* sched, int32_t* n_steps)
{
  if (nt <= 0 || !mask || !perm || !lanes) return PTZ_EINVAL;
  int la = 0, lb = 0;
  const bool planned = plan_dissection(nt, first_dense, mask, perm, &la, &lb);
  if (!planned) { for (int t = 0; t < nt; ++t) perm[t] = t; la = lb = 0; }
  lanes[0] = la; lanes[1] = lb;
  if (sched && n_steps) {  // as ptz_ba_batch_create does: permuted mask, fill, levels
    std::vector<unsigned char> m((size_t)nt * nt, 0);
    for (int a = 0; a < nt; ++a)
      for (int e = 0; e <= a; ++e)
        if (mask[a * nt + e]) m[std::max(perm[a], perm[e]) * nt + std::min(perm[a], perm[e])] = 1;
    for (int k = 0; k < nt; ++k)
      for (int x = k + 1; x < nt; ++x) {
        if (!m[x * nt + k]) continue;
        for (int y = k + 1; y <= x; ++y)
          if (m[y * nt + k]) m[x * nt + y] = 1;
      }
    if (planned) n_steps[0] = level_schedule(nt, m.data(), sched);
    else {
      for (int i = 0; i < nt * CHOL_STEP_COLS; ++i) sched[i] = -1;
      for (int t = 0; t < nt; ++t) sched[CHOL_STEP_COLS * t] = t;
      n_steps[0] = nt;
    }
  }
  return planned ? 1 : 0;
}

int32_t ptz_ba_cam_block_dim(int32_t factor_type)
{
  if (factor_type == PTZ_BA_PTZRay) return 4;
  if (factor_type == PTZ_BA_PTZRayDist) return 5;
  if (factor_type == PTZ_BA_PTZRayFxfyDist) return 6;
  if (factor_type == PTZ_BA_PTZRayDistDisp) return 8;  // [f, k1, r1, r2, r3, d0, d1, d2]
  return PTZ_EUNSUPPORTED;
}

int32_t ptz_ba_batch_cam_block_dim(const ptz_ba_batch* b) { return b ? b->nc : PTZ_EINVAL; }

void ptz_ba_batch_destroy(ptz_ba_batch* b)
{
#ifdef PTZ_CHOL_TIMELINE
  if (b && b->d.chol.np > 0) chol_chain_timeline_print(b->d.chol.np / 64);
#endif
  if (!b) return;
  DeviceGuard guard(b->device);
  // nothing of this batch may still be running when its memory is handed to the next one
  for (auto st : b->streams) (void)stream_wait(st);
  for (auto st : b->aux) (void)stream_wait(st);
  if (b->io) (void)stream_wait(b->io);
  b->release_staged();
  const int dv = b->device;
  for (void* p : b->allocs) ptzpool::dev_release(dv, p);
  for (auto e : b->ev_pool) ptzpool::event_release(dv, true, e);
  if (b->h_ctl) ptzpool::pinned_release(b->h_ctl);
  for (auto& v : b->pass_graph) for (auto ge : v) if (ge) (void)hipGraphExecDestroy(ge);
  ptzpool::event_release(dv, true, b->ev0);
  ptzpool::event_release(dv, true, b->ev1);
  if (b->create_ev) ptzpool::event_release(dv, false, b->create_ev);
  for (auto st : b->streams) ptzpool::stream_release(dv, st);
  for (auto st : b->aux) ptzpool::stream_release(dv, st);
  ptzpool::stream_release(dv, b->io);
  for (auto e : b->la_ev) ptzpool::event_release(dv, false, e);
  for (auto e : b->fork_ev) ptzpool::event_release(dv, false, e);
  for (auto e : b->join_ev) ptzpool::event_release(dv, false, e);
  delete b;
}

namespace {
// Camera-pair entry lists of one scene (the off-diagonal blocks of the reduced system): for every ray, every (a, b) with
// cam(a) > cam(b); a is stored as its position in cam(a)'s observation list (the LDS slot of T_a in k_schur).  Independent of
// every other scene apart from the base offset of its observations, so batches build these lists on several host threads.
struct PairBuild {
  std::vector<int> pci, pcj, pptr, pbrow;  // pptr: scene-local entry offsets (n_pair + 1); pbrow: first W row of camera cj
  std::vector<unsigned> ent;        // copied into the batch-wide array (in parallel, per wave) and released
  std::vector<uint2> runs;          // k_schur's runs: {first entry (scene-local here), pair among the camera's | entries << 16}
  std::vector<int> prun;            // first run of every pair (scene-local), n_pair + 1
  int64_t n_ent = 0;
  int n_pair = 0, max_cam_obs = 0, max_cam_ent = 0, max_cam_pair = 0, max_cam_run = 0, err = PTZ_OK;
};

// Where a scene's observation-side arrays go in the batch-wide host arrays (all offsets are prefix sums of the scene sizes,
// known before any scene is built, so the worker threads write them in place).
struct ObsDest {
  float2* uv; int* cam; int* ray; int* camobs; int* camray; float2* camuv;  // + obs_off
  int* rayptr; double* w;                                      // + ray_off (+ scene index for the pointer array)
  int* camptr; int* campair; int* camrun;                      // + cam_off + scene index
  int* wpos;                                                   // + obs_off
};  // (member order = the order of the initialiser in ptz_ba_batch_create)

// Scratch vectors of build_pairs, handed from call to call (a batch builds its scenes on short-lived worker threads, several
// batches at a time in the lock-step PTZ-IBA): a vector that keeps its capacity costs no allocation and, above all, no page
// faults -- fresh memory was a third of the structure stage's time.
struct BuildScratch {
  std::vector<int> len, first, start, cnt_ray, cnt_cam, fill, pos, pair_cnt, pair_fill, cam_first, cam_ent;
  std::vector<char> asc;
};
struct ScratchPool {
  std::mutex mu;
  std::vector<BuildScratch*> free_list;
  BuildScratch* get()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_list.empty()) { BuildScratch* s = free_list.back(); free_list.pop_back(); return s; }
    }
    return new BuildScratch();
  }
  void put(BuildScratch* s)
  {
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() < 32) { free_list.push_back(s); return; }
    delete s;
  }
};
inline ScratchPool* scratch_pool()
{
  static ScratchPool* p = new ScratchPool();  // never destroyed: worker threads may outlive static destructors
  return p;
}
struct ScratchLease {
  BuildScratch* s;
  ScratchLease() : s(scratch_pool()->get()) {}
  ~ScratchLease() { scratch_pool()->put(s); }
};

void build_pairs(const ptz_ba_problem& p_in, int obase, int ray_off, const ObsDest& od, PairBuild& out, int* ray_perm, int max_runs,
                 bool pairs_on_device = false)
{
  // Internal ray order: by track length, longest first, then by the track's first camera (stable).  The ray-centric kernels give one lane to a ray and walk its
  // observations; with the caller's order a wave of 64 rays waits for its longest track (4 .. 19 observations on a C2 rig, the
  // mean over waves of the longest is 14.5 against a mean length of 7.5), sorted it does not (1.01 x the mean).  Observations
  // keep their order inside a track; ray_perm[j] = the caller's index of internal ray j.
  ScratchLease lease;
  BuildScratch& sc = *lease.s;
  ptz_ba_problem p = p_in;
  std::vector<int>& cnt_ray = sc.cnt_ray;
  std::vector<int>& cnt_cam = sc.cnt_cam;
  cnt_ray.assign(p.n_ray + 1, 0);
  cnt_cam.assign(p.n_cam + 1, 0);
  {
    std::vector<int>&len = sc.len, &first = sc.first, &start = sc.start;
    len.assign(p.n_ray, 0);
    first.assign(p.n_ray + 1, 0);
    for (int64_t a = 0; a < p.n_obs; ++a) ++len[p_in.obs_ray[a]];
    int max_len = 0;
    for (int j = 0; j < p.n_ray; ++j) {
      if (len[j] == 0) { out.err = PTZ_EINVAL; return; }  // every ray has >= 1 observation
      first[j + 1] = first[j] + len[j];
      max_len = std::max(max_len, len[j]);
    }
    // counting sort by (length descending, first camera of the track ascending), stable: rays of one wave then also look at
    // the same few cameras at the same time, and their reads of the camera table in LDS are broadcasts instead of bank conflicts
    // (coarser spatial keys in front of the length -- 2 to 16 sectors of the camera range -- were measured: no further gain)
    const size_t nkey = (size_t)(max_len + 1) * (size_t)std::max(p.n_cam, 1);
    auto key = [&](int j) {
      const int c = p_in.obs_cam[first[j]];
      return (size_t)(max_len - len[j]) * (size_t)p.n_cam + (size_t)(c >= 0 && c < p.n_cam ? c : 0);
    };
    start.assign(nkey + 2, 0);
    for (int j = 0; j < p.n_ray; ++j) ++start[key(j) + 1];
    for (size_t l = 0; l <= nkey; ++l) start[l + 1] += start[l];
    for (int j = 0; j < p.n_ray; ++j) ray_perm[start[key(j)]++] = j;
    // the observations in the internal order go straight into the batch-wide arrays (no copy of the problem in between)
    int at = 0;
    for (int jn = 0; jn < p.n_ray; ++jn) {
      const int jo = ray_perm[jn];
      od.w[jn] = p_in.ray_weight[jo];
      cnt_ray[jn] = at;
      for (int a = first[jo]; a < first[jo + 1]; ++a, ++at) {
        const int c = p_in.obs_cam[a];
        od.uv[at] = make_float2(p_in.obs_uv[2 * a], p_in.obs_uv[2 * a + 1]);
        od.cam[at] = c;
        od.ray[at] = jn;
        ++cnt_cam[c + 1];
      }
    }
    cnt_ray[p.n_ray] = at;
    p.obs_uv = reinterpret_cast<const float*>(od.uv); p.obs_cam = od.cam; p.obs_ray = od.ray; p.ray_weight = od.w;
  }
  for (int j = 0; j <= p.n_ray; ++j) od.rayptr[j] = obase + cnt_ray[j];
  for (int c = 0; c < p.n_cam; ++c) cnt_cam[c + 1] += cnt_cam[c];
  std::vector<int>& pos = sc.pos;
  pos.resize(p.n_obs);
  {  // camera-major observation lists; pos[a] = rank of observation a in its camera's list (camera-major order = ascending a)
    std::vector<int>& fill = sc.fill;
    fill.assign(cnt_cam.begin(), cnt_cam.end() - 1);
    for (int64_t a = 0; a < p.n_obs; ++a) {
      const int c = p.obs_cam[a];
      const int slot = fill[c]++;
      pos[a] = slot - cnt_cam[c];
      od.camobs[slot] = obase + (int)a;
      od.camray[slot] = ray_off + p.obs_ray[a];
      od.camuv[slot] = od.uv[a];
      od.wpos[a] = obase + slot;
    }
    for (int c = 0; c < p.n_cam; ++c) out.max_cam_obs = std::max(out.max_cam_obs, fill[c] - cnt_cam[c]);
    for (int c = 0; c <= p.n_cam; ++c) od.camptr[c] = obase + cnt_cam[c];
  }
  if (pairs_on_device) {
    // pairs, entries and runs are made by k_pairs from the arrays above (ptz_ba_batch_create): only the entry count is needed here
    int64_t ne = 0;
    for (int j = 0; j < p.n_ray; ++j) { const int64_t L = cnt_ray[j + 1] - cnt_ray[j]; ne += L * (L - 1) / 2; }
    if (ne > 0x7fffffff) { out.err = PTZ_EINVAL; return; }
    out.n_ent = ne;
    out.n_pair = 0;
    return;
  }
  // counting sort by (ci, cj): pairs ascending in ci * n_cam + cj, the entries of a pair in ray order (stable).
  // The (a, b) pairs with cam(a) > cam(b) of every ray are walked twice in ray order -- once to count them per camera pair,
  // once to drop them into their slots -- rather than kept as a list (round 2 kept two 4-byte indices per entry, 3 MB per C2
  // scene written and re-read: 6.4 -> 4.0 ms per scene on one host thread).  Observations of a track come camera-ascending
  // from the packing (track asc, image asc: ptzray_optimizer.cc:801-850), in which case the pairs are simply (a, b < a); any
  // other order takes the general double loop.  (Walking camera by camera instead, with n_cam counters that stay in L1, was
  // measured: slower, the walk then chases observation -> track -> cameras in random order.)
  const size_t ncc = (size_t)p.n_cam * p.n_cam;
  std::vector<int>& pair_cnt = sc.pair_cnt;
  pair_cnt.assign(ncc, 0);
  std::vector<char>& asc = sc.asc;
  asc.resize(p.n_ray);
  int64_t n_ent = 0;
  auto for_each_entry = [&](auto&& fn) -> bool {  // fn(a, bb, slot of (cam(a), cam(bb)) in the n_cam x n_cam table)
    for (int j = 0; j < p.n_ray; ++j) {
      const int r0 = cnt_ray[j], r1 = cnt_ray[j + 1];
      if (asc[j]) {
        for (int a = r0 + 1; a < r1; ++a) {
          const size_t row = (size_t)p.obs_cam[a] * p.n_cam;
          for (int bb = r0; bb < a; ++bb) fn(a, bb, row + p.obs_cam[bb]);
        }
      }
      else {
        for (int a = r0; a < r1; ++a)
          for (int bb = r0; bb < r1; ++bb) {
            const int ci = p.obs_cam[a], cj = p.obs_cam[bb];
            if (ci == cj && a != bb) return false;  // an image appears once per track (tracks.cc:77)
            if (ci > cj) fn(a, bb, (size_t)ci * p.n_cam + cj);
          }
      }
    }
    return true;
  };
  for (int j = 0; j < p.n_ray; ++j) {
    bool ascending = true;
    for (int a = cnt_ray[j] + 1; a < cnt_ray[j + 1]; ++a) ascending &= p.obs_cam[a] > p.obs_cam[a - 1];
    asc[j] = ascending ? 1 : 0;
  }
  if (!for_each_entry([&](int, int, size_t cell) { ++pair_cnt[cell]; ++n_ent; })) { out.err = PTZ_EINVAL; return; }
  if (n_ent > 0x7fffffff) { out.err = PTZ_EINVAL; return; }
  int npair = 0;
  std::vector<int>&cam_first = sc.cam_first, &cam_ent = sc.cam_ent, &pair_fill = sc.pair_fill;
  cam_first.assign(p.n_cam + 1, -1);
  cam_ent.assign(p.n_cam, 0);
  pair_fill.resize(ncc);  // next free entry slot of a pair (scene-local); only the cells of existing pairs are read
  {
    int run = 0;
    for (int ci = 0; ci < p.n_cam; ++ci)
      for (int cj = 0; cj < ci; ++cj) {
        const int c = pair_cnt[(size_t)ci * p.n_cam + cj];
        if (c == 0) continue;
        out.pci.push_back(ci);
        out.pcj.push_back(cj);
        out.pbrow.push_back(obase + cnt_cam[cj]);
        out.pptr.push_back(run);
        if (cam_first[ci] < 0) cam_first[ci] = npair;
        pair_fill[(size_t)ci * p.n_cam + cj] = run;
        cam_ent[ci] += c;
        run += c;
        ++npair;
      }
    out.pptr.push_back(run);
  }
  out.ent.resize((size_t)n_ent);
  for_each_entry([&](int a, int bb, size_t cell) {
    out.ent[pair_fill[cell]++] = (unsigned)pos[a] | ((unsigned)pos[bb] << 16);  // (LDS slot of T_a, W row of b relative to camera cj's first row)
  });
  // per-camera pair ranges (pairs are sorted by ci): cameras without pairs get an empty range
  cam_first[p.n_cam] = npair;
  for (int c = 0; c < p.n_cam; ++c) out.max_cam_ent = std::max(out.max_cam_ent, cam_ent[c]);
  for (int c = p.n_cam - 1; c >= 0; --c) if (cam_first[c] < 0) cam_first[c] = cam_first[c + 1];
  for (int c = 0; c <= p.n_cam; ++c) od.campair[c] = cam_first[c];
  for (int c = 0; c < p.n_cam; ++c) out.max_cam_pair = std::max(out.max_cam_pair, cam_first[c + 1] - cam_first[c]);
  // Runs of k_schur's second phase.  A camera's entries (its pairs one after the other) are cut into pieces of one length L
  // that never straddle two pairs, L the smallest for which the camera has at most max_runs (= threads) pieces: a thread of the
  // camera's workgroup then sums exactly one piece, and all threads have the same amount of work whatever the lengths of the
  // pairs.  (A view with more pairs than threads gets one run per pair and several rounds; k_schur then keeps its table in
  // global memory, see schur_tg.)
  out.prun.assign(npair + 1, 0);
  for (int c = 0; c < p.n_cam; ++c) {
    od.camrun[c] = (int)out.runs.size();
    const int p0 = cam_first[c], p1 = cam_first[c + 1];
    if (p1 == p0) continue;
    int max_len = 0;
    for (int pi = p0; pi < p1; ++pi) max_len = std::max(max_len, out.pptr[pi + 1] - out.pptr[pi]);
    auto pieces = [&](int L) { int64_t n = 0; for (int pi = p0; pi < p1; ++pi) n += (out.pptr[pi + 1] - out.pptr[pi] + L - 1) / L; return n; };
    // smallest L in [1, max_len] with pieces(L) <= max_runs (max_len if there is none).  pieces() is non-increasing in L and
    // pieces(L) >= entries / L, so the search starts at ceil(entries / max_runs) -- usually the answer or one below it -- and
    // walks up (a bisection over [1, max_len] cost nine rounds of integer divisions per camera, 0.3 ms per 170-view scene)
    const int total = out.pptr[p1] - out.pptr[p0];
    int lo = std::max(1, std::min(max_len, (total + max_runs - 1) / max_runs));
    while (lo < max_len && pieces(lo) > max_runs) ++lo;
    const int L = std::min(lo, 65535);
    for (int pi = p0; pi < p1; ++pi) {
      out.prun[pi] = (int)out.runs.size();
      for (int e = out.pptr[pi]; e < out.pptr[pi + 1]; e += L)
        out.runs.push_back(make_uint2((unsigned)e, (unsigned)(pi - p0) | ((unsigned)std::min(L, out.pptr[pi + 1] - e) << 16)));
    }
    out.max_cam_run = std::max(out.max_cam_run, (int)out.runs.size() - od.camrun[c]);
  }
  od.camrun[p.n_cam] = (int)out.runs.size();
  out.prun[npair] = (int)out.runs.size();
  out.n_pair = npair;
  out.n_ent = n_ent;
}
}  // namespace

static int32_t create_impl(int32_t n, const ptz_ba_problem* problems, const ptz_rig_view* views, int32_t view_type, const ptz_lm_options* opt,
                           ptz_ba_batch** out);
int32_t ptz_ba_batch_create(int32_t n, const ptz_ba_problem* problems, const ptz_lm_options* opt, ptz_ba_batch** out)
{
  if (n <= 0 || !problems || !out) return PTZ_EINVAL;
  return create_impl(n, problems, nullptr, 0, opt, out);
}
int32_t ptz_ba_batch_create_views(int32_t n, const ptz_rig_view* views, int32_t factor_type, const ptz_lm_options* opt, ptz_ba_batch** out)
{
  if (n <= 0 || !views || !out) return PTZ_EINVAL;
  if (factor_type != PTZ_BA_PTZRay && factor_type != PTZ_BA_PTZRayDist && factor_type != PTZ_BA_PTZRayFxfyDist) return PTZ_EUNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    const ptz_rig_view& v = views[i];
    if (!v.rig || v.n_cam <= 0 || !v.cam_image || v.rig->n_track <= 0) return PTZ_EINVAL;
    for (int c = 0; c < v.n_cam; ++c)
      if (v.cam_image[c] < 0 || v.cam_image[c] >= v.rig->n_img || (c > 0 && v.cam_image[c] <= v.cam_image[c - 1])) return PTZ_EINVAL;
  }
  return create_impl(n, nullptr, views, factor_type, opt, out);
}

// views == nullptr: the n packed problems.  Else: the packed problems of n views of resident rigs, built on the device
// (ptz_view_kernels.h) -- the structure arrays then have the extents of the WHOLE rigs (nothing is read back to size them),
// each scene's real counts live in its SceneDev.
static int32_t create_impl(int32_t n, const ptz_ba_problem* problems, const ptz_rig_view* views, int32_t view_type, const ptz_lm_options* opt,
                           ptz_ba_batch** out)
{
  *out = nullptr;
  const bool vm = views != nullptr;
  std::vector<ptz_ba_problem> vprob;
  if (vm) {  // stand-ins that carry the sizes the layout is made for
    vprob.resize(n);
    for (int i = 0; i < n; ++i) {
      memset(&vprob[i], 0, sizeof(ptz_ba_problem));
      vprob[i].n_cam = views[i].n_cam; vprob[i].n_ray = views[i].rig->n_track; vprob[i].n_obs = views[i].rig->n_view;
      vprob[i].factor_type = view_type;
    }
    problems = vprob.data();
  }
  const bool dbg_t = getenv("PTZ_BA_DEBUG_TIMING") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tc0 = now_ms();
  double tc1 = 0, tc2 = 0, tc3 = 0, ts_obs = 0, ts_ent = 0;
  ptz_lm_options o;
  if (opt) o = *opt; else ptz_lm_options_default(&o);
  if (o.max_num_iterations <= 0) return PTZ_EINVAL;  // CheckValid, ptzray_optimizer.cc:521
  const int type = problems[0].factor_type;
  if (type != PTZ_BA_PTZRay && type != PTZ_BA_PTZRayDist && type != PTZ_BA_PTZRayFxfyDist && type != PTZ_BA_PTZRayDistDisp) return PTZ_EUNSUPPORTED;
  const bool disp = type == PTZ_BA_PTZRayDistDisp;
  int has3d = 0;
  std::vector<int> cam_hist;
  int pre_max_cam_obs = 0, pre_max_cam = 0;  // (decides whether the pair lists are built on the device, before anything is built)
  // ---- validate + sizes (host only; no device touched before this passes)
  for (int i = 0; vm && i < n; ++i) {
    for (int c = 0; c < views[i].n_cam; ++c) pre_max_cam_obs = std::max(pre_max_cam_obs, views[i].rig->img_obs[views[i].cam_image[c]]);
    pre_max_cam = std::max(pre_max_cam, views[i].n_cam);
  }
  for (int i = 0; !vm && i < n; ++i) {
    const ptz_ba_problem& p = problems[i];
    if (p.factor_type != type) return PTZ_EINVAL;
    if (p.n_cam <= 0 || p.n_ray <= 0 || p.n_obs <= 0) return PTZ_EINVAL;  // num_cams_ == 0 -> false (:517)
    if (!p.obs_uv || !p.obs_cam || !p.obs_ray || !p.ray_weight) return PTZ_EINVAL;
    if (p.n_obs3d < 0 || (p.n_obs3d > 0 && (!p.obs3d_uv || !p.obs3d_xyz || !p.obs3d_cam))) return PTZ_EINVAL;
    for (int a = 0; a < p.n_obs3d; ++a)
      if (p.obs3d_cam[a] < 0 || p.obs3d_cam[a] >= p.n_cam) return PTZ_EINVAL;
    if (p.n_obs3d > 0) has3d = 1;
    cam_hist.assign(p.n_cam, 0);
    for (int64_t a = 0; a < p.n_obs; ++a) {
      if (p.obs_cam[a] < 0 || p.obs_cam[a] >= p.n_cam || p.obs_ray[a] < 0 || p.obs_ray[a] >= p.n_ray) return PTZ_EINVAL;
      if (a > 0 && p.obs_ray[a] < p.obs_ray[a - 1]) return PTZ_EINVAL;
      ++cam_hist[p.obs_cam[a]];
    }
    for (int c = 0; c < p.n_cam; ++c) pre_max_cam_obs = std::max(pre_max_cam_obs, cam_hist[c]);
    pre_max_cam = std::max(pre_max_cam, p.n_cam);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= o.device_id) return PTZ_ENODEVICE;
  PTZ_DEVICE_GUARD(o.device_id);

  // fy becomes a live column with annotation residuals; PTZRayFxfyDist has it anyway
  const int NC = type == PTZ_BA_PTZRayFxfyDist ? 6 : ((type == PTZ_BA_PTZRay) ? 4 : (disp ? 8 : 5)) + has3d;
  const int CBS = disp ? CAMBLK_DISP + 1 : CAMBLK + 1, CDS = disp ? CAMBLK_DISP + 1 : CANDBLK + 1;  // Dims<TYPE>::CBS / CDS
  ptz_ba_batch* b = new ptz_ba_batch();
  b->has3d = has3d;
  b->n_scene = n; b->type = type; b->nc = NC; b->opt = o; b->device = o.device_id;
  if (ptzpool::stream_acquire(b->device, &b->io) != hipSuccess) { delete b; return PTZ_ENODEVICE; }
  // rays per workgroup of the ray-centric kernels: one rig's ~13 k rays on 1024-ray workgroups keep 14 compute units busy,
  // so a few scenes use small workgroups; large batches amortise the LDS camera tables over many rays (results do not
  // depend on it: per-ray sums are reduced per wave of 64 rays)
  b->ray_block = n <= 4 ? 128 : (n <= 32 ? 256 : RAY_BLOCK);
  if (const char* e = getenv("PTZ_BA_RAY_BLOCK")) b->ray_block = std::min(RAY_BLOCK, std::max(64, (atoi(e) / 64) * 64));
  RawVec<float2> h_uv, h_camuv;
  RawVec<int> h_cam, h_ray, h_camobs, h_camray, h_wpos;
  std::vector<int> h_rayptr, h_camptr, h_pci, h_pcj, h_pptr;
  RawVec<unsigned> h_ent;
  std::vector<int> h_pbrow;
  std::vector<int> h_campair, h_camrun, h_prun;
  std::vector<uint2> h_runs;
  std::vector<double> h_w, h_o3xyz;
  std::vector<float2> h_o3uv;
  std::vector<int> h_o3cam;
  std::vector<int> h_grpptr, h_grpmem;
  std::vector<unsigned char> h_grpcls;
  if (!vm) {
    size_t to = 0, tr = 0, tc = 0;
    for (int i = 0; i < n; ++i) { to += (size_t)problems[i].n_obs; tr += (size_t)problems[i].n_ray; tc += (size_t)problems[i].n_cam; }
    h_uv.reserve(to); h_cam.reserve(to); h_ray.reserve(to); h_camobs.reserve(to); h_wpos.reserve(to); h_camray.reserve(to);
    h_rayptr.reserve(tr + n); h_w.reserve(tr); h_camptr.reserve(tc + n); h_campair.reserve(tc + n);
    h_ent.reserve(to * 4);
  }
  std::vector<unsigned char> h_camflag;
  int total_grp = 0;
  bool any_shared = false;
  int64_t tot_obs = 0, tot_ent = 0;
  for (int i = 0; i < n; ++i) {
    tot_obs += problems[i].n_obs;
  }
  if (tot_obs > 0x7fffffff) { delete b; return PTZ_EINVAL; }
  if (!vm) { h_uv.reserve(tot_obs); h_cam.reserve(tot_obs); h_ray.reserve(tot_obs); h_camobs.reserve(tot_obs); }
  // the pair lists of the scenes are built a wave at a time on up to 8 host threads (PTZ_BA_HOST_THREADS)
  std::vector<int> obs_base(n, 0), ray_base(n, 0), cam_base(n, 0);
  for (int i = 1; i < n; ++i) {
    obs_base[i] = obs_base[i - 1] + (int)problems[i - 1].n_obs;
    ray_base[i] = ray_base[i - 1] + problems[i - 1].n_ray;
    cam_base[i] = cam_base[i - 1] + problems[i - 1].n_cam;
  }
  if (!vm) {
    const size_t tr = (size_t)ray_base[n - 1] + problems[n - 1].n_ray, tc = (size_t)cam_base[n - 1] + problems[n - 1].n_cam;
    h_uv.resize(tot_obs); h_camuv.resize(tot_obs); h_cam.resize(tot_obs); h_ray.resize(tot_obs); h_camobs.resize(tot_obs); h_camray.resize(tot_obs);
    h_rayptr.resize(tr + n); h_w.resize(tr); h_camptr.resize(tc + n); h_campair.resize(tc + n); h_camrun.resize(tc + n); h_wpos.resize(tot_obs);
    b->ray_perm.resize((size_t)ray_base[n - 1] + problems[n - 1].n_ray);
  }
  else h_camptr.assign((size_t)cam_base[n - 1] + problems[n - 1].n_cam + n, 0);  // (read by the per-camera flags below: no camera has residuals as far as they are concerned)
  b->views_mode = vm;
  int n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("PTZ_BA_HOST_THREADS")) n_threads = std::max(1, atoi(e));
  const int wave_scenes = 4 * n_threads;
  // Pair lists on the device (k_pairs) unless the round-2 kernels are asked for, the per-camera bitmaps do not fit in LDS, or
  // PTZ_BA_GPU_STRUCT=0 (the host builder: 2.1 of its 3.6 ms per 170-view problem are these lists)
  bool gpu_pairs = !getenv("PTZ_BA_SCHUR_W") || atoi(getenv("PTZ_BA_SCHUR_W")) == 0;
  const size_t pairs_lds = sizeof(unsigned) * ((size_t)pre_max_cam * ((pre_max_cam_obs + 31) / 32) + 5 * (size_t)pre_max_cam + 8);
  if (pairs_lds > 150 * 1024 || pre_max_cam_obs > 65535) gpu_pairs = false;
  if (const char* e = getenv("PTZ_BA_GPU_STRUCT")) gpu_pairs = gpu_pairs && atoi(e) != 0;
  if (vm && (pairs_lds > 150 * 1024 || pre_max_cam_obs > 65535)) { delete b; return PTZ_ELIMIT; }  // (a view is built on the device or not at all)
  if (vm) gpu_pairs = true;
  // PTZ_BA_GPU_STRUCT_CHECK=1 (tests): the host builds its lists as well and the device's must equal them word for word
  const bool pairs_check = gpu_pairs && getenv("PTZ_BA_GPU_STRUCT_CHECK") != nullptr;
  std::vector<PairBuild> wave;
  int wave_first = 0;
  for (int i = 0; i < n; ++i) {
    const ptz_ba_problem& p = problems[i];
    SceneDev s;
    s.n_cam = p.n_cam; s.n_ray = p.n_ray; s.n_obs = (int)p.n_obs;
    s.cam_off = b->total_cam; s.ray_off = b->total_ray; s.obs_off = b->total_obs;
    s.pair_off = b->total_pair; s.ent_off = b->total_ent; s.part_off = b->total_chunk;
    s.n_chunk = (p.n_ray + b->ray_block - 1) / b->ray_block;
    s.n_wave = (p.n_ray + 63) / 64;
    s.n = NC * p.n_cam + 6 * has3d;
    s.idx = i;
    s.o3_off = b->total_o3; s.n_o3 = p.n_obs3d;
    for (int a = 0; a < p.n_obs3d; ++a) {
      h_o3uv.push_back(make_float2(p.obs3d_uv[2 * a], p.obs3d_uv[2 * a + 1]));
      h_o3cam.push_back(p.obs3d_cam[a]);
      for (int k = 0; k < 3; ++k) h_o3xyz.push_back(p.obs3d_xyz[3 * a + k]);
    }
    b->total_o3 += p.n_obs3d;
    // observations, ray ranges, camera-major lists, camera-pair entry lists: built ahead of this loop, a wave of scenes at a
    // time on several threads (build_pairs); the observation-side arrays are written in place, the pair lists appended here
    const double tsb = now_ms();
    if (vm) {
      // extents of the WHOLE rig (the real counts are the device's to find, and nothing is read back to size an array): pairs of
      // the candidates, the rig's entries, at most max(runs per workgroup, pairs of a camera) runs per camera
      const ptz_rig* rg = views[i].rig;
      const int64_t pair_bound = (int64_t)p.n_cam * (p.n_cam - 1) / 2;
      const int64_t run_bound = (int64_t)p.n_cam * std::max(schur_threads_of(type), p.n_cam);
      if ((int64_t)b->total_ent + rg->ent_bound > 0x7fffffff || (int64_t)b->total_pair + pair_bound > 0x7fffffff ||
          (int64_t)b->total_run + run_bound > 0x7fffffff) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
      s.run_off = b->total_run;
      s.n_pair = 0;
      b->total_run += (int)run_bound;
      b->total_pair += (int)pair_bound;
      b->total_ent += (int)rg->ent_bound;
      for (int c = 0; c < p.n_cam; ++c) {
        b->max_cam_obs = std::max(b->max_cam_obs, rg->img_obs[views[i].cam_image[c]]);
        b->max_cam_ent = std::max(b->max_cam_ent, rg->img_ent[views[i].cam_image[c]]);
      }
      b->max_cam_pair = std::max(b->max_cam_pair, p.n_cam - 1);
      b->max_cam_run = std::max(b->max_cam_run, std::max(schur_threads_of(type), p.n_cam - 1));
    }
    else {
      if (i >= wave_first + (int)wave.size()) {
        wave_first = i;
        const int wn = std::min(n - i, wave_scenes);
        wave.assign(wn, PairBuild());
        auto work = [&](int t0, int step) {
          for (int k = t0; k < wn; k += step) {
            const int sidx = wave_first + k;
            const ObsDest od = {h_uv.data() + obs_base[sidx], h_cam.data() + obs_base[sidx], h_ray.data() + obs_base[sidx],
                                h_camobs.data() + obs_base[sidx], h_camray.data() + obs_base[sidx], h_camuv.data() + obs_base[sidx],
                                h_rayptr.data() + ray_base[sidx] + sidx, h_w.data() + ray_base[sidx],
                                h_camptr.data() + cam_base[sidx] + sidx, h_campair.data() + cam_base[sidx] + sidx,
                                h_camrun.data() + cam_base[sidx] + sidx, h_wpos.data() + obs_base[sidx]};
            build_pairs(problems[sidx], obs_base[sidx], ray_base[sidx], od, wave[k], b->ray_perm.data() + ray_base[sidx], schur_threads_of(type), gpu_pairs && !pairs_check);
          }
        };
        const int nt = std::min(n_threads, wn);
        auto run = [&](auto&& fn) {
          if (nt <= 1) { fn(0, 1); return; }
          std::vector<std::thread> th;
          for (int t = 1; t < nt; ++t) th.emplace_back(fn, t, nt);
          fn(0, nt);
          for (auto& x : th) x.join();
        };
        run(work);
        // the entry lists of the wave go into the batch-wide array in one resize and a parallel copy
        std::vector<size_t> ent_at(wn);
        size_t at = h_ent.size();
        bool wave_ok = true;
        for (int k = 0; k < wn; ++k) { ent_at[k] = at; at += wave[k].ent.size(); wave_ok &= wave[k].err == PTZ_OK; }
        if (!wave_ok || at > 0x7fffffff) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
        h_ent.resize(at);
        run([&](int t0, int step) {
          for (int k = t0; k < wn; k += step) {
            if (!wave[k].ent.empty()) memcpy(h_ent.data() + ent_at[k], wave[k].ent.data(), sizeof(unsigned) * wave[k].ent.size());
            std::vector<unsigned>().swap(wave[k].ent);
          }
        });
      }
      PairBuild& pb = wave[i - wave_first];
      if (pb.err != PTZ_OK || (int64_t)b->total_ent + pb.n_ent > 0x7fffffff) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
      h_pci.insert(h_pci.end(), pb.pci.begin(), pb.pci.end());
      h_pcj.insert(h_pcj.end(), pb.pcj.begin(), pb.pcj.end());
      h_pbrow.insert(h_pbrow.end(), pb.pbrow.begin(), pb.pbrow.end());
      for (int v : pb.pptr) h_pptr.push_back(b->total_ent + v);
      for (const uint2& r : pb.runs) h_runs.push_back(make_uint2(r.x + (unsigned)b->total_ent, r.y));
      h_prun.insert(h_prun.end(), pb.prun.begin(), pb.prun.end());
      s.run_off = b->total_run;
      b->total_run += (int)pb.runs.size();
      b->max_cam_run = std::max(b->max_cam_run, pb.max_cam_run);
      b->max_cam_obs = std::max(b->max_cam_obs, pb.max_cam_obs);
      b->max_cam_ent = std::max(b->max_cam_ent, pb.max_cam_ent);
      b->max_cam_pair = std::max(b->max_cam_pair, pb.max_cam_pair);
      s.n_pair = pb.n_pair;
      b->total_ent += (int)pb.n_ent;
      b->total_pair += pb.n_pair;
      ts_ent += now_ms() - tsb;
      pb = PairBuild();  // release this scene's lists
    }
    // shared intrinsics groups (ids are arbitrary integers; members in ascending camera order; groups in order of their
    // first member); a camera alone in its group needs nothing
    {
      s.grp_off = total_grp;
      s.n_grp = 0;
      std::vector<int> first(p.n_cam);
      for (int c = 0; c < p.n_cam; ++c) {
        first[c] = c;
        if (p.ic_of_cam)
          for (int m = 0; m < c; ++m)
            if (p.ic_of_cam[m] == p.ic_of_cam[c]) { first[c] = m; break; }
      }
      for (int c = 0; c < p.n_cam; ++c) b->first_of_group.push_back(b->total_cam + first[c]);
      std::vector<char> counted(p.n_cam, 0);  // group (by first member) already has its counting camera
      bool disp_counted = false;
      for (int c = 0; c < p.n_cam; ++c) {
        const int* camptr_s = h_camptr.data() + cam_base[i] + i;  // filled by build_pairs for this scene
        const bool has_res = camptr_s[c + 1] > camptr_s[c];
        unsigned char flag = 0;
        if (has_res && !counted[first[c]]) { flag = 1; counted[first[c]] = 1; }
        if (disp && has_res && !disp_counted) { flag |= 2; disp_counted = true; }
        h_camflag.push_back(flag);
      }
      h_grpptr.push_back((int)h_grpmem.size());
      for (int c = 0; c < p.n_cam; ++c) {
        if (first[c] != c) continue;
        int members = 0;
        for (int m = c; m < p.n_cam; ++m) members += first[m] == c;
        if (members < 2) continue;
        for (int m = c; m < p.n_cam; ++m) if (first[m] == c) h_grpmem.push_back(m);
        h_grpptr.push_back((int)h_grpmem.size());
        h_grpcls.push_back(0);
        ++s.n_grp;
        any_shared = true;
      }
      if (disp) {  // the displacement block is one parameter of the whole problem: a group of all cameras, over the d slots
        for (int m = 0; m < p.n_cam; ++m) h_grpmem.push_back(m);
        h_grpptr.push_back((int)h_grpmem.size());
        h_grpcls.push_back(1);
        ++s.n_grp;
        any_shared = true;
      }
      total_grp += s.n_grp;
      b->max_grp = std::max(b->max_grp, s.n_grp);
    }
    b->total_cam += p.n_cam; b->total_ray += p.n_ray; b->total_obs += (int)p.n_obs; b->total_chunk += s.n_wave + 1;
    b->max_cam = std::max(b->max_cam, p.n_cam); b->max_ray = std::max(b->max_ray, p.n_ray);
    b->max_chunk = std::max(b->max_chunk, s.n_chunk); b->max_pair = std::max(b->max_pair, s.n_pair);
    b->max_n = std::max(b->max_n, s.n);
    b->scenes.push_back(s);
  }
  (void)tot_ent;

  tc1 = now_ms();
  Dev& d = b->d;
  memset(&d, 0, sizeof(d));
  d.n_scene = n;
  int rc = PTZ_OK;
#define TRY(x) do { rc = (x); if (rc) { ptz_ba_batch_destroy(b); return rc; } } while (0)
  std::vector<unsigned char> h_adj;  // views: tile adjacency of every scene's reduced system, marked on the device
  if (vm) {
    // ---- the packed problems of the views, on the device (ptz_view_kernels.h): observations in the library's order, ray and camera
    // lists, camera pairs; one read-back at the end (counts, tile adjacency), nothing in between
    const int nt_e = chol_padded_order(b->max_n) / CHOL_NB;
    std::vector<ViewDev> hv(n);
    std::vector<int> h_map, h_camimg;
    size_t trk_total = 0;
    int max_trk = 0;
    int64_t view_key_top = 0;  // largest (longest candidate track bound) * cameras + cameras of the views: what the sort keys must hold
    for (int i = 0; i < n; ++i) {
      const ptz_rig* rg = views[i].rig;
      if (rg->device != b->device) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
      ViewDev& v = hv[i];
      v.trk_ptr = rg->d_trk_ptr; v.trk_img = rg->d_trk_img; v.trk_uv = rg->d_trk_uv;
      v.n_track = rg->n_track; v.n_img = rg->n_img; v.n_cam = views[i].n_cam;
      v.map_off = (int)h_map.size(); v.trk_off = (int)trk_total;
      v.n_ray = 0; v.n_obs = 0; v.max_len = 0;
      h_map.insert(h_map.end(), (size_t)rg->n_img, -1);
      for (int c = 0; c < v.n_cam; ++c) { h_map[v.map_off + views[i].cam_image[c]] = c; h_camimg.push_back(views[i].cam_image[c]); }
      trk_total += (size_t)rg->n_track;
      max_trk = std::max(max_trk, rg->n_track);
      // the sort key of the internal ray order holds (longest - length) * cameras + first camera in 22 bits, the track in 24.  What
      // k_view_keys puts there is below (the view's longest candidate track) * cameras, and a candidate track is no longer than the
      // rig's longest track or the number of candidate images -- NOT the rig's mean length: one long track in a wide view would
      // otherwise spill into the bits of the view number, and the batch-wide sort would silently be a different order.
      // That bound belongs to the BATCH-WIDE sort (k_view_keys, taken below when a rig has more than 16 384 tracks or on request);
      // the one-launch sort (k_view_sort) keeps a 32-bit key of its own per view, whose padding value 2 << bits(top) must fit: top < 2^29.
      view_key_top = std::max(view_key_top, (int64_t)std::min(rg->max_track_len, v.n_cam) * v.n_cam + v.n_cam);
      if (rg->n_track >= (1 << 24)) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
    }
    const bool block_sort = max_trk <= 16384 && !(getenv("PTZ_BA_VIEW_BLOCK_SORT") && atoi(getenv("PTZ_BA_VIEW_BLOCK_SORT")) == 0);
    if (view_key_top >= (block_sort ? ((int64_t)1 << 29) : ((int64_t)1 << 22))) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
    const ViewDev* dviews_c = nullptr;
    const int *d_map = nullptr, *d_camimg = nullptr, *d_chunkoff = nullptr;
    std::vector<int> h_chunkoff(n);
    size_t chunk_total = 0;  // [chunks of 256 rays][cameras] counters of the views (k_view_hist), bounds from the rigs' track counts
    for (int i = 0; i < n; ++i) {
      h_chunkoff[i] = (int)chunk_total;
      chunk_total += (size_t)((views[i].rig->n_track + 255) / 256) * views[i].n_cam;
      if (chunk_total > 0x7fffffffu) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
    }
    {
      StagedUpload up;
      up.add(b->scenes, &d.scene);
      up.add(hv, &dviews_c);
      up.add(h_map, &d_map);
      up.add(h_camimg, &d_camimg);
      up.add(h_chunkoff, &d_chunkoff);
      TRY(up.commit(b, /*wait=*/false));
    }
    ViewBuild vb;
    memset(&vb, 0, sizeof(vb));
    vb.views = const_cast<ViewDev*>(dviews_c);
    vb.cam_of_image = d_map; vb.cam_image = d_camimg;
    vb.scene = const_cast<SceneDev*>(d.scene);
    vb.chunk_off = d_chunkoff;
    if (b->max_cam <= VIEW_LDS_CAMS) TRY(b->alloc(&vb.chunk_cnt, chunk_total));  // (unused by the wide path)
    unsigned long long *key_out = nullptr;
    int* val_out = nullptr;
    TRY(b->alloc(&vb.t_len, trk_total)); TRY(b->alloc(&vb.t_first, trk_total)); TRY(b->alloc(&vb.t_ext, trk_total));
    TRY(b->alloc(&vb.key_in, trk_total)); TRY(b->alloc(&vb.val_in, trk_total)); TRY(b->alloc(&key_out, trk_total)); TRY(b->alloc(&val_out, trk_total));
    TRY(b->alloc(&vb.ray_trk, (size_t)b->total_ray)); TRY(b->alloc(&vb.ray_len, (size_t)b->total_ray)); TRY(b->alloc(&vb.cam_cnt, (size_t)b->total_cam));
    TRY(b->alloc(&vb.obs_uv, (size_t)b->total_obs)); TRY(b->alloc(&vb.obs_cam, (size_t)b->total_obs)); TRY(b->alloc(&vb.obs_ray, (size_t)b->total_obs));
    TRY(b->alloc(&vb.ray_ptr, (size_t)b->total_ray + n)); TRY(b->alloc(&vb.cam_ptr, (size_t)b->total_cam + n));
    TRY(b->alloc(&vb.cam_obs, (size_t)b->total_obs)); TRY(b->alloc(&vb.wpos, (size_t)b->total_obs)); TRY(b->alloc(&vb.cam_ray, (size_t)b->total_obs));
    TRY(b->alloc(&vb.cam_uv, (size_t)b->total_obs)); TRY(b->alloc(&vb.ray_w, (size_t)b->total_ray)); TRY(b->alloc(&vb.ray_perm, (size_t)b->total_ray));
    d.obs_uv = vb.obs_uv; d.obs_cam = vb.obs_cam; d.obs_ray = vb.obs_ray; d.ray_ptr = vb.ray_ptr; d.cam_ptr = vb.cam_ptr; d.cam_obs = vb.cam_obs;
    d.wpos = vb.wpos; d.cam_ray = vb.cam_ray; d.cam_uv = vb.cam_uv; d.ray_w = vb.ray_w; b->d_ray_perm = vb.ray_perm;
    hipStream_t st = b->io;
    const dim3 gtrk((max_trk + 255) / 256, n);
    const double tv0 = now_ms();
    hipLaunchKernelGGL(k_view_tracks, gtrk, dim3(256), 0, st, vb);
    hipLaunchKernelGGL(k_view_scan, dim3(n), dim3(1024), 0, st, vb);
    // the views' internal ray order: one launch (a workgroup per view sorts its tracks) when every view fits, else the batch-wide sort
    if (block_sort) {
      {
        static std::mutex sort_mu;
        static bool sort_cap[64] = {};
        std::lock_guard<std::mutex> lk(sort_mu);
        const int dv = b->device & 63;
        if (!sort_cap[dv]) { (void)hipFuncSetAttribute((const void*)k_view_sort<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); sort_cap[dv] = true; }
      }
      if (max_trk <= 4096) hipLaunchKernelGGL(k_view_sort<4>, dim3(n), dim3(1024), sizeof(rocprim::block_radix_sort<unsigned, 1024, 4, int>::storage_type), st, vb, val_out);
      else hipLaunchKernelGGL(k_view_sort<16>, dim3(n), dim3(1024), sizeof(rocprim::block_radix_sort<unsigned, 1024, 16, int>::storage_type), st, vb, val_out);
    }
    else {
      hipLaunchKernelGGL(k_view_keys, gtrk, dim3(256), 0, st, vb);
      unsigned end_bit = 47;
      while ((1u << (end_bit - 47)) < (unsigned)n) ++end_bit;
      size_t tmp_bytes = 0;
      void* tmp = nullptr;
      // (the low 24 bits of a key are the track number and the keys are generated in (view, track) order: a STABLE sort on the bits
      //  above them gives the same order in four digit passes instead of seven)
      const unsigned begin_bit = 24;
      if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, vb.key_in, key_out, vb.val_in, val_out, trk_total, begin_bit, end_bit, st) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
      char* tmpc = nullptr;
      TRY(b->alloc(&tmpc, tmp_bytes + 256));
      tmp = tmpc;
      if (rocprim::radix_sort_pairs(tmp, tmp_bytes, vb.key_in, key_out, vb.val_in, val_out, trk_total, begin_bit, end_bit, st) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    }
    const double tva = now_ms();
    hipLaunchKernelGGL(k_view_rays, gtrk, dim3(256), 0, st, vb, (const int*)val_out);
    hipLaunchKernelGGL(k_view_rayscan, dim3(n), dim3(1024), 0, st, vb, b->ray_block);
    // (PTZ_BA_DEBUG_VIEW_WIDE=1: the path of views wider than VIEW_LDS_CAMS for any view -- tests: the same lists)
    const bool view_wide = b->max_cam > VIEW_LDS_CAMS || (getenv("PTZ_BA_DEBUG_VIEW_WIDE") && atoi(getenv("PTZ_BA_DEBUG_VIEW_WIDE")) != 0);
    if (!view_wide) {  // camera-major lists by counting: chunk histograms, running sums, placement
      hipLaunchKernelGGL(k_view_hist, gtrk, dim3(256), sizeof(int) * (size_t)b->max_cam, st, vb);
      hipLaunchKernelGGL(k_view_chunkscan, dim3(n), dim3(1024), 0, st, vb);
      hipLaunchKernelGGL(k_view_place, gtrk, dim3(256), sizeof(unsigned long long) * 4 * (size_t)b->max_cam, st, vb);
    }
    else {
      hipLaunchKernelGGL(k_view_obs, gtrk, dim3(256), 0, st, vb);
      hipLaunchKernelGGL(k_view_camscan, dim3(n), dim3(1024), 0, st, vb);
      hipLaunchKernelGGL(k_view_camlists_wide, dim3(b->max_cam, n), dim3(256), 0, st, vb);
    }
    const double tvb = now_ms();
    // camera pairs, entry lists, runs: k_pairs as for any batch, into arrays of the bounds' extents (no sizing read-back)
    PairsDev pa;
    memset(&pa, 0, sizeof(pa));
    pa.scene = d.scene; pa.obs_cam = d.obs_cam; pa.obs_ray = d.obs_ray; pa.ray_ptr = d.ray_ptr; pa.cam_ptr = d.cam_ptr; pa.cam_obs = d.cam_obs;
    pa.wpos = d.wpos;
    pa.max_runs = schur_threads_of(type);
    int *d_cnt = nullptr, *d_off3 = nullptr, *d_tot = nullptr, *d_err = nullptr;
    unsigned char* d_adj = nullptr;
    TRY(b->alloc(&d_cnt, (size_t)3 * b->total_cam));
    TRY(b->alloc(&d_off3, (size_t)3 * b->total_cam));
    TRY(b->alloc(&d_tot, (size_t)6 * n + 1));
    d_err = d_tot + 6 * n;
    TRY(b->alloc(&d_adj, (size_t)n * nt_e * nt_e));
    pa.cam_cnt = d_cnt; pa.cam_off3 = d_off3; pa.scene_tot = d_tot; pa.err = d_err;
    TRY(b->alloc(&pa.pci, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pcj, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pbrow, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pptr, (size_t)b->total_pair + n));
    TRY(b->alloc(&pa.prun, (size_t)b->total_pair + n));
    TRY(b->alloc(&pa.campair, (size_t)b->total_cam + n));
    TRY(b->alloc(&pa.camrun, (size_t)b->total_cam + n));
    TRY(b->alloc(&pa.runs, (size_t)b->total_run));
    TRY(b->alloc(&pa.ent, (size_t)b->total_ent));
    pairs_lds_cap(b->device);
    const double tvc = now_ms();
    if (hipMemsetAsync(d_err, 0, sizeof(int), st) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    hipLaunchKernelGGL(k_pairs<false>, dim3(b->max_cam, n), dim3(256), pairs_lds, st, pa);
    hipLaunchKernelGGL(k_pair_scan, dim3((n + 63) / 64), dim3(64), 0, st, d.scene, n, (const int*)d_cnt, d_off3, d_tot);
    hipLaunchKernelGGL(k_view_pairs_patch, dim3((n + 63) / 64), dim3(64), 0, st, vb.scene, n, (const int*)d_tot);
    hipLaunchKernelGGL(k_pairs<true>, dim3(b->max_cam, n), dim3(256), pairs_lds, st, pa);
    hipLaunchKernelGGL(k_view_adjacency, dim3(n), dim3(256), 0, st, d.scene, (const int*)pa.pci, (const int*)pa.pcj, NC, nt_e, d_adj);
    d.pair_ci = pa.pci; d.pair_cj = pa.pcj; d.pair_brow = pa.pbrow; d.pair_ptr = pa.pptr; d.pair_run = pa.prun;
    d.cam_pair = pa.campair; d.cam_run = pa.camrun; d.run_rec = pa.runs; d.ent = pa.ent;
    // the one read-back: the views' counts, the pair totals, the error word, the tile adjacency
    const double tvd = now_ms();
    std::vector<int> h_tot((size_t)6 * n + 1);
    h_adj.resize((size_t)n * nt_e * nt_e);
    {
      hipError_t e = hipMemcpyAsync(hv.data(), dviews_c, sizeof(ViewDev) * n, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(h_tot.data(), d_tot, sizeof(int) * (6 * n + 1), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(h_adj.data(), d_adj, h_adj.size(), hipMemcpyDeviceToHost, st);
      const double tv1 = now_ms();
      if (e == hipSuccess) e = stream_wait(st);
      if (dbg_t) fprintf(stderr, "[ptz_ba_create] views: allocations + upload %.3f ms, build enqueued in %.3f ms, waited %.3f ms for it | tracks..sort %.3f, rays..place %.3f, pair arrays %.3f, memset + pairs launches %.3f, copies back %.3f\n", tv0 - tc1, tv1 - tv0, now_ms() - tv1,
                         tva - tv0, tvb - tva, tvc - tvb, tvd - tvc, tv1 - tvd);
      b->release_staged();
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess) { (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    }
    if (h_tot[(size_t)6 * n]) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }  // an image twice in one track (tracks.cc:77)
    b->max_pair = 0; b->max_cam_pair = 0; b->max_cam_ent = 0; b->max_cam_run = 0;
    for (int i = 0; i < n; ++i) {
      if (hv[i].n_obs <= 0 || hv[i].n_ray <= 0) { ptz_ba_batch_destroy(b); return PTZ_ENOOBS; }  // no candidate observation: not a problem (:517); its own code, so that callers can tell it from a malformed view
      SceneDev& sd = b->scenes[i];
      const int* t = h_tot.data() + 6 * (size_t)i;
      sd.n_ray = hv[i].n_ray; sd.n_obs = hv[i].n_obs;
      sd.n_wave = (sd.n_ray + 63) / 64; sd.n_chunk = (sd.n_ray + b->ray_block - 1) / b->ray_block;
      sd.n_pair = t[0];
      b->max_pair = std::max(b->max_pair, t[0]);
      b->max_cam_pair = std::max(b->max_cam_pair, t[3]);
      b->max_cam_ent = std::max(b->max_cam_ent, t[4]);
      b->max_cam_run = std::max(b->max_cam_run, t[5]);
    }
    gpu_pairs = false;  // (done)
  }
  else {
    StagedUpload up;
    up.add(b->scenes, &d.scene);
    up.add(h_uv, &d.obs_uv);
    up.add(h_cam, &d.obs_cam);
    up.add(h_ray, &d.obs_ray);
    up.add(h_rayptr, &d.ray_ptr);
    up.add(h_camptr, &d.cam_ptr);
    up.add(h_camobs, &d.cam_obs);
    up.add(h_wpos, &d.wpos);
    up.add(h_camray, &d.cam_ray);
    up.add(h_camuv, &d.cam_uv);
    up.add(h_pci, &d.pair_ci);
    up.add(h_pcj, &d.pair_cj);
    up.add(h_pbrow, &d.pair_brow);
    up.add(h_pptr, &d.pair_ptr);
    up.add(h_campair, &d.cam_pair);
    up.add(h_camrun, &d.cam_run);
    up.add(h_runs, &d.run_rec);
    up.add(h_prun, &d.pair_run);
    up.add(h_ent, &d.ent);
    up.add(h_w, &d.ray_w);
    up.add(h_o3uv, &d.o3_uv);
    up.add(h_o3xyz, &d.o3_xyz);
    up.add(h_o3cam, &d.o3_cam);
    up.add(b->ray_perm, &b->d_ray_perm);
    TRY(up.commit(b));
  }
  if (gpu_pairs) {
    // camera pairs, entry lists and runs on the device (k_pairs): count, scan, read the totals back (the arrays are sized
    // exactly), write; then the pairs' cameras come back for the tile structure below
    PairsDev pa;
    memset(&pa, 0, sizeof(pa));
    pa.scene = d.scene; pa.obs_cam = d.obs_cam; pa.obs_ray = d.obs_ray; pa.ray_ptr = d.ray_ptr; pa.cam_ptr = d.cam_ptr; pa.cam_obs = d.cam_obs;
    pa.wpos = d.wpos;
    pa.max_runs = schur_threads_of(type);
    int *d_cnt = nullptr, *d_off3 = nullptr, *d_tot = nullptr, *d_err = nullptr;
    TRY(b->alloc(&d_cnt, (size_t)3 * b->total_cam));
    TRY(b->alloc(&d_off3, (size_t)3 * b->total_cam));
    TRY(b->alloc(&d_tot, (size_t)6 * n));
    TRY(b->alloc(&d_err, 1));
    pa.cam_cnt = d_cnt; pa.cam_off3 = d_off3; pa.scene_tot = d_tot; pa.err = d_err;
    pairs_lds_cap(b->device);  // dynamic LDS beyond 64 KiB for very wide rigs
    if (hipMemsetAsync(d_err, 0, sizeof(int), b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    hipLaunchKernelGGL(k_pairs<false>, dim3(b->max_cam, n), dim3(256), pairs_lds, b->io, pa);
    hipLaunchKernelGGL(k_pair_scan, dim3((n + 63) / 64), dim3(64), 0, b->io, d.scene, n, (const int*)d_cnt, d_off3, d_tot);
    std::vector<int> h_tot((size_t)6 * n + 1);
    {
      hipError_t e = hipMemcpyAsync(h_tot.data(), d_tot, sizeof(int) * 6 * n, hipMemcpyDeviceToHost, b->io);
      if (e == hipSuccess) e = hipMemcpyAsync(h_tot.data() + 6 * n, d_err, sizeof(int), hipMemcpyDeviceToHost, b->io);
      if (e == hipSuccess) e = stream_wait(b->io);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess) { (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    }
    if (h_tot[(size_t)6 * n]) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }  // an image twice in one track (tracks.cc:77)
    const int chk_pair = b->total_pair, chk_run = b->total_run, chk_mcp = b->max_cam_pair, chk_mce = b->max_cam_ent, chk_mcr = b->max_cam_run;
    const std::vector<SceneDev> chk_scenes = pairs_check ? b->scenes : std::vector<SceneDev>();
    if (pairs_check) { b->max_cam_pair = 0; b->max_cam_ent = 0; b->max_cam_run = 0; }
    b->total_pair = 0; b->total_run = 0; b->max_pair = 0;
    for (int i = 0; i < n; ++i) {
      SceneDev& sd = b->scenes[i];
      const int* t = h_tot.data() + 6 * (size_t)i;
      int64_t ne = 0;  // (the host's count of the scene's entries, from the track lengths)
      ne = (i + 1 < n ? b->scenes[i + 1].ent_off : b->total_ent) - sd.ent_off;
      if (t[1] != ne || (int64_t)b->total_pair + t[0] > 0x7fffffff || (int64_t)b->total_run + t[2] > 0x7fffffff) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
      sd.n_pair = t[0];
      sd.pair_off = b->total_pair;
      sd.run_off = b->total_run;
      b->total_pair += t[0];
      b->total_run += t[2];
      b->max_pair = std::max(b->max_pair, t[0]);
      b->max_cam_pair = std::max(b->max_cam_pair, t[3]);
      b->max_cam_ent = std::max(b->max_cam_ent, t[4]);
      b->max_cam_run = std::max(b->max_cam_run, t[5]);
    }
    if (copy_on(b->io, const_cast<SceneDev*>(d.scene), b->scenes.data(), sizeof(SceneDev) * n, hipMemcpyHostToDevice) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    TRY(b->alloc(&pa.pci, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pcj, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pbrow, (size_t)b->total_pair));
    TRY(b->alloc(&pa.pptr, (size_t)b->total_pair + n));
    TRY(b->alloc(&pa.prun, (size_t)b->total_pair + n));
    TRY(b->alloc(&pa.campair, (size_t)b->total_cam + n));
    TRY(b->alloc(&pa.camrun, (size_t)b->total_cam + n));
    TRY(b->alloc(&pa.runs, (size_t)b->total_run));
    TRY(b->alloc(&pa.ent, (size_t)b->total_ent));
    hipLaunchKernelGGL(k_pairs<true>, dim3(b->max_cam, n), dim3(256), pairs_lds, b->io, pa);
    d.pair_ci = pa.pci; d.pair_cj = pa.pcj; d.pair_brow = pa.pbrow; d.pair_ptr = pa.pptr; d.pair_run = pa.prun;
    d.cam_pair = pa.campair; d.cam_run = pa.camrun; d.run_rec = pa.runs; d.ent = pa.ent;
    if (pairs_check) {
      long bad = 0;
      if (chk_pair != b->total_pair || chk_run != b->total_run || chk_mcp != b->max_cam_pair || chk_mce != b->max_cam_ent || chk_mcr != b->max_cam_run) ++bad;
      for (int i = 0; i < n && !bad; ++i)
        if (chk_scenes[i].n_pair != b->scenes[i].n_pair || chk_scenes[i].pair_off != b->scenes[i].pair_off || chk_scenes[i].run_off != b->scenes[i].run_off) ++bad;
      auto cmp = [&](const void* dev, const void* host, size_t bytes, const char* name) {
        std::vector<unsigned char> tmp(bytes + 1);
        if (bytes && (hipMemcpyAsync(tmp.data(), dev, bytes, hipMemcpyDeviceToHost, b->io) != hipSuccess || stream_wait(b->io) != hipSuccess)) { ++bad; return; }
        if (bytes && memcmp(tmp.data(), host, bytes) != 0) { ++bad; fprintf(stderr, "[ptz_ba_create] device-built %s differs from the host builder's\n", name); }
      };
      if (!bad) {
        cmp(pa.pci, h_pci.data(), sizeof(int) * h_pci.size(), "pair_ci");
        cmp(pa.pcj, h_pcj.data(), sizeof(int) * h_pcj.size(), "pair_cj");
        cmp(pa.pbrow, h_pbrow.data(), sizeof(int) * h_pbrow.size(), "pair_brow");
        cmp(pa.pptr, h_pptr.data(), sizeof(int) * h_pptr.size(), "pair_ptr");
        cmp(pa.prun, h_prun.data(), sizeof(int) * h_prun.size(), "pair_run");
        cmp(pa.campair, h_campair.data(), sizeof(int) * h_campair.size(), "cam_pair");
        cmp(pa.camrun, h_camrun.data(), sizeof(int) * h_camrun.size(), "cam_run");
        cmp(pa.runs, h_runs.data(), sizeof(uint2) * h_runs.size(), "run_rec");
        cmp(pa.ent, h_ent.data(), sizeof(unsigned) * h_ent.size(), "ent");
      }
      else fprintf(stderr, "[ptz_ba_create] device-built pair totals differ from the host builder's\n");
      if (bad) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
    }
    h_pci.resize(b->total_pair);
    h_pcj.resize(b->total_pair);
    {
      hipError_t e = b->total_pair ? hipMemcpyAsync(h_pci.data(), pa.pci, sizeof(int) * b->total_pair, hipMemcpyDeviceToHost, b->io) : hipSuccess;
      if (e == hipSuccess && b->total_pair) e = hipMemcpyAsync(h_pcj.data(), pa.pcj, sizeof(int) * b->total_pair, hipMemcpyDeviceToHost, b->io);
      if (e == hipSuccess) e = stream_wait(b->io);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess) { (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    }
  }
  d.tlw_stride = (size_t)n * 6;
  TRY(b->alloc(&d.tlw_x, 2 * d.tlw_stride));
  TRY(b->alloc(&b->tlw0, d.tlw_stride));
  d.tlwblk_stride = (size_t)n * TLWBLK;
  TRY(b->alloc(&d.tlwblk, 2 * d.tlwblk_stride));
  TRY(b->alloc(&d.tlwcand, (size_t)n * TLWBLK));
  TRY(b->alloc(&d.scale_t, (size_t)n * 6));
  TRY(b->alloc(&d.diag_t, (size_t)n * 6));
  TRY(b->alloc(&d.Ut, (size_t)n * 36));
  TRY(b->alloc(&d.gt, (size_t)n * 6));
  TRY(b->alloc(&d.dt, (size_t)n * 6));
  TRY(b->alloc(&d.Jc3, (size_t)b->total_o3 * 2 * NC));
  TRY(b->alloc(&d.Jt3, (size_t)b->total_o3 * 12));
  TRY(b->alloc(&d.r3, (size_t)b->total_o3 * 2));
  if (hipMemsetAsync(b->tlw0, 0, sizeof(double) * d.tlw_stride, b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
  d.cam_stride = (size_t)b->total_cam * 15;
  d.ray_stride = (size_t)b->total_ray * 3;
  TRY(b->alloc(&d.cam_x, 2 * d.cam_stride));
  TRY(b->alloc(&d.ray_x, 2 * d.ray_stride));
  TRY(b->alloc(&b->cam0, d.cam_stride));
  if (disp) {
    d.dsp_stride = (size_t)b->total_cam * 3;
    TRY(b->alloc(&d.dsp_x, 2 * d.dsp_stride));
    TRY(b->alloc(&b->dsp0, d.dsp_stride));
    if (hipMemsetAsync(b->dsp0, 0, sizeof(double) * d.dsp_stride, b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }  // :655
  }
  TRY(b->alloc(&b->ray0, d.ray_stride));
  d.camblk_stride = ((size_t)b->total_cam * CBS + 2 + 1) & ~(size_t)1;
  TRY(b->alloc(&d.camblk, 2 * d.camblk_stride));
  TRY(b->alloc(&d.candblk, (size_t)b->total_cam * CDS + 2));
  TRY(b->alloc(&d.scale_c, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.scale_r, (size_t)b->total_ray * 3));
  TRY(b->alloc(&d.U, (size_t)2 * b->total_cam * NC * NC));
  TRY(b->alloc(&d.gc, (size_t)2 * b->total_cam * NC));
  TRY(b->alloc(&d.costc, (size_t)2 * b->total_cam));
  d.lin_cams = b->total_cam;
  TRY(b->alloc(&d.diag_c, (size_t)2 * b->total_cam * NC));
  TRY(b->alloc(&d.dc, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.dct, (size_t)b->total_cam * (NC | 1) + 4));
  // (two halves each: k_eval leaves the ray-side linearisation of its candidate in the half that LmState.cur does not select)
  d.V_stride = (size_t)b->total_ray * 6; d.gr_stride = (size_t)b->total_ray * 3;
  d.rayrec_stride = (size_t)b->total_ray * 8; d.plin_stride = (size_t)b->total_chunk * 2;
  TRY(b->alloc(&d.V, 2 * d.V_stride));
  TRY(b->alloc(&d.gr, 2 * d.gr_stride));
  TRY(b->alloc(&d.diag_r, (size_t)b->total_ray * 3));
  TRY(b->alloc(&d.E, (size_t)b->total_ray * EZS));  // the rays' records, as EZS / 2 planes of 16-byte pieces (e_piece)
  d.e_stride = (size_t)b->total_ray;
  d.shared = any_shared ? 1 : 0;
  if (any_shared) {
    TRY(upload(b, h_grpptr, &d.grp_ptr));
    TRY(upload(b, h_grpmem, &d.grp_mem));
    TRY(upload(b, h_camflag, &d.cam_flag));
    if (disp) TRY(upload(b, h_grpcls, &d.grp_cls));
    TRY(b->alloc(&d.gfold, (size_t)b->total_cam * NC));
  }
  // W = Jc^T Jr rows are not materialised by the solver (k_schur rebuilds them from the rays); ptz_ba_batch_linearize allocates
  // them when a caller asks for them, PTZ_BA_SCHUR_W=1 brings round 2's kernels back for A/B measurements
  if (const char* e = getenv("PTZ_BA_SCHUR_W")) b->schur_w = atoi(e) != 0;
  if (b->schur_w) TRY(b->alloc(&d.W, (size_t)b->total_obs * (disp ? 24 : type == PTZ_BA_PTZRayFxfyDist ? 18 : 16)));  // room for the widest row stride
  TRY(b->alloc(&d.rayrec, 2 * d.rayrec_stride));
  TRY(b->alloc(&d.partial, (size_t)b->total_chunk * 4));
  TRY(b->alloc(&d.partial_lin, 2 * d.plin_stride));
  d.ray_block = b->ray_block;
  TRY(b->alloc(&d.camstep, (size_t)b->total_cam * 2));
  TRY(b->alloc(&d.cam_gmax, (size_t)2 * b->total_cam));
  TRY(b->alloc(&d.tail_cnt, (size_t)2 * n));
  TRY(b->alloc(&d.lm, (size_t)n));
  TRY(b->alloc(&d.active, (size_t)n));
  TRY(b->alloc(&d.ray_fail, (size_t)n));
  // reduced camera systems
  d.chol.count = n;
  d.chol.np = chol_padded_order(b->max_n);
  // the systems' orders and (below) their tile structure, elimination order, step schedule and back-substitution lists: ONE upload
  StagedUpload up2;
  std::vector<int> hn(n);
  for (int i = 0; i < n; ++i) hn[i] = b->scenes[i].n;
  up2.add(hn, &d.chol.n);
  std::vector<unsigned char> hm;
  std::vector<int> h_tperm, h_sched, h_groups;
  std::vector<BsItem> h_items;
  TRY(b->alloc(&d.chol.A, (size_t)n * d.chol.np * d.chol.np));
  {
    // the one-launch-per-column factorisation (a few scenes, or the last few active scenes of a large batch) publishes its
    // finished L tiles in a second matrix, indexed by launch slot: up to CHOL_CHAIN_SLOTS slots per scene group
    const int groups = std::max(1, std::min(b->n_group_hint(n), n));
    const size_t slots = (size_t)std::min(n, CHOL_CHAIN_SLOTS) * groups;
    TRY(b->alloc(&d.chol.L, slots * d.chol.np * d.chol.np));
    if (hipMemsetAsync(d.chol.L, 0, sizeof(double) * slots * d.chol.np * d.chol.np, b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
    TRY(b->alloc(&b->d_act, (size_t)n));
    const size_t ctl_ints = chol_chain_ctl_ints(d.chol.np) * groups;
    TRY(b->alloc(&d.chol.chain_ctl, ctl_ints));
    if (hipMemsetAsync(d.chol.chain_ctl, 0, sizeof(int) * ctl_ints, b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
  }
  TRY(b->alloc(&d.chol.Ldiag, (size_t)n * (d.chol.np / CHOL_NB) * CHOL_NB * CHOL_NB));
  TRY(b->alloc(&d.chol.Dinv, (size_t)n * (d.chol.np / CHOL_NB) * 4 * 16 * 16));
  TRY(b->alloc(&d.chol.Linv, (size_t)n * (d.chol.np / CHOL_NB) * CHOL_NB * CHOL_NB));
  TRY(b->alloc(&d.chol.fail, (size_t)n));
  d.chol.active = d.active;
  TRY(b->alloc(&d.yc, (size_t)n * d.chol.np));
  tc2 = now_ms();
  // Tile-level structure of every reduced camera system: cameras that share a track couple their tiles, the T_l_w
  // block and the rhs row couple to everything; closed under the fill of the right-looking factorisation.  The tiles are
  // eliminated in the order plan_dissection picks (arcs of a ring side by side, separators last) when that shortens the
  // dependent chain of the factorisation; Dev::tperm carries the order to the kernels, CholBatch::sched the steps.
  if (!getenv("PTZ_BA_DENSE_CHOL")) {
    const int nt = d.chol.np / CHOL_NB;
    bool dissect = true;
    bool nested = true;
    if (const char* e = getenv("PTZ_BA_ORDER")) { dissect = strcmp(e, "natural") != 0; nested = strcmp(e, "flat") != 0; }
    hm.assign((size_t)n * nt * nt, 0);
    h_tperm.resize((size_t)n * nt); h_sched.assign((size_t)n * nt * CHOL_STEP_COLS, -1);
    int max_steps = 0;
    bool any_plan = false;
    std::vector<unsigned char> m0((size_t)nt * nt);
    for (int i = 0; i < n; ++i) {
      const SceneDev& sd = b->scenes[i];
      std::fill(m0.begin(), m0.end(), 0);
      auto tile_lo = [&](int cam) { return (cam * NC) / CHOL_NB; };
      auto tile_hi = [&](int cam) { return (cam * NC + NC - 1) / CHOL_NB; };
      if (vm) memcpy(m0.data(), h_adj.data() + (size_t)i * nt * nt, (size_t)nt * nt);  // (marked on the device by the same rule: k_view_adjacency)
      for (int c = 0; !vm && c < sd.n_cam; ++c)
        for (int a = tile_lo(c); a <= tile_hi(c); ++a)
          for (int e = tile_lo(c); e <= a; ++e) m0[a * nt + e] = 1;
      for (int p = 0; !vm && p < sd.n_pair; ++p) {
        const int ci = h_pci[sd.pair_off + p], cj = h_pcj[sd.pair_off + p];
        for (int a = tile_lo(ci); a <= tile_hi(ci); ++a)
          for (int e = tile_lo(cj); e <= tile_hi(cj); ++e) {
            if (a >= e) m0[a * nt + e] = 1; else m0[e * nt + a] = 1;
          }
      }
      // shared intrinsics: the fold puts a dense row / column at every group's representative (its last camera)
      if (any_shared) {
        const int* gp = h_grpptr.data() + sd.grp_off + i;
        for (int g = 0; g < sd.n_grp; ++g) {
          const int rep = h_grpmem[gp[g + 1] - 1];
          for (int a = tile_lo(rep); a <= tile_hi(rep); ++a) {
            for (int e = 0; e <= a; ++e) m0[a * nt + e] = 1;
            for (int r = a; r < nt; ++r) m0[r * nt + a] = 1;
          }
        }
      }
      // dense rows: the global block (if any) and the rhs row, index n_cam * NC .. n
      const int first_dense = (sd.n_cam * NC) / CHOL_NB;
      for (int a = first_dense; a < nt; ++a)
        for (int e = 0; e <= a; ++e) m0[a * nt + e] = 1;
      int* perm = h_tperm.data() + (size_t)i * nt;
      int* sched = h_sched.data() + (size_t)i * nt * CHOL_STEP_COLS;
      int lane_a = 0, lane_b = 0;
      const bool planned = dissect && sd.n_grp == 0 && plan_dissection(nt, first_dense, m0.data(), perm, &lane_a, &lane_b, nested);
      if (!planned) for (int t = 0; t < nt; ++t) perm[t] = t;
      if (planned) any_plan = true;
      unsigned char* m = hm.data() + (size_t)i * nt * nt;
      for (int a = 0; a < nt; ++a)
        for (int e = 0; e <= a; ++e) {
          if (!m0[a * nt + e]) continue;
          const int pa = perm[a], pe = perm[e];
          m[std::max(pa, pe) * nt + std::min(pa, pe)] = 1;
        }
      for (int k = 0; k < nt; ++k)
        for (int x = k + 1; x < nt; ++x) {
          if (!m[x * nt + k]) continue;
          for (int y = k + 1; y <= x; ++y)
            if (m[y * nt + k]) m[x * nt + y] = 1;
        }
      // steps: from the filled structure when the order was planned, one block column after the other otherwise
      int steps = 0;
      if (planned) steps = level_schedule(nt, m, sched);
      else for (int t = 0; t < nt; ++t, ++steps) sched[CHOL_STEP_COLS * steps] = t;
      max_steps = std::max(max_steps, steps);
      if ((int)b->sched_kmin.size() < steps) b->sched_kmin.resize(steps, nt);
      for (int st = 0; st < steps; ++st) b->sched_kmin[st] = std::min(b->sched_kmin[st], sched[CHOL_STEP_COLS * st]);  // (slot 0 of a step holds its smallest column)
      if (dbg_t && i == 0) fprintf(stderr, "[ptz_ba_create] scene 0: %d tiles, elimination %s: lanes %d + %d, %d steps\n", nt, planned ? "dissected" : "natural", lane_a, lane_b, steps);
    }
    up2.add(hm, &d.chol.tmask);
    if (any_plan) {
      up2.add(h_tperm, &d.tperm);  // (d.chol.xperm = d.tperm once the upload has placed it)
      up2.add(h_sched, &d.chol.sched);
      d.chol.n_steps = max_steps;
      d.chol.sched_kmin = b->sched_kmin.data();
    }
    {  // the back-substitution's work list, per scene (the kernel would otherwise make it itself, every launch: ~6 us of one wave)
      const int mg = chol_backsolve_max_groups(d.chol.np);
      const char* e = getenv("PTZ_BA_BACKSOLVE_HOST_LIST");  // 0: the kernel makes the list itself (tests: the same list, the same bits)
      // (mg <= 1024: chol_backsolve_kernel keeps the kinds of the host list's groups in a fixed array of that many)
      if ((!e || atoi(e) != 0) && mg <= 1024 && sizeof(double) * ((size_t)d.chol.np + 1024) + sizeof(BsItem) * 4 * (size_t)mg <= 150 * 1024) {  // (the kernel's own list form applies)
        h_items.resize((size_t)n * 4 * mg);
        h_groups.resize(n);
        // a few systems: the two arcs of a dissected system on a workgroup each (chol_backsolve_arcs); PTZ_BA_BACKSOLVE_SPLIT=0: one workgroup
        const char* es = getenv("PTZ_BA_BACKSOLVE_SPLIT");
        const bool want_split = any_plan && n <= 8 && nt <= 64 && !(es && atoi(es) == 0);
        for (int i = 0; i < n; ++i) {
          bool split = false;
          h_groups[i] = chol_backsolve_plan(d.chol.np, b->scenes[i].n, hm.data() + (size_t)i * nt * nt,
                                            any_plan ? h_sched.data() + (size_t)i * nt * CHOL_STEP_COLS : nullptr, max_steps, h_items.data() + (size_t)i * 4 * mg,
                                            want_split, &split);
          if (split) d.chol.bs_split = 1;
        }
        up2.add(h_items, &d.chol.bs_items);
        up2.add(h_groups, &d.chol.bs_groups);
      }
    }
    // tiles outside the structure are never written again: zero everything once (the block may be a recycled one)
    if (hipMemsetAsync(d.chol.A, 0, sizeof(double) * (size_t)n * d.chol.np * d.chol.np, b->io) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
  }
  TRY(up2.commit(b, /*wait=*/false));  // (the wait for the zero fills at the end of this function ends it)
  d.chol.xperm = d.tperm;
#undef TRY
  d.cam_x0 = b->cam0; d.ray_x0 = b->ray0;
  d.opt.max_num_iterations = o.max_num_iterations;
  d.opt.max_consecutive_invalid = o.max_num_consecutive_invalid_steps;
  d.opt.jacobi_scaling = o.jacobi_scaling;
  d.opt.initial_radius = o.initial_trust_region_radius;
  d.opt.max_radius = o.max_trust_region_radius;
  d.opt.min_radius = o.min_trust_region_radius;
  d.opt.min_relative_decrease = o.min_relative_decrease;
  d.opt.min_lm_diagonal = o.min_lm_diagonal;
  d.opt.max_lm_diagonal = o.max_lm_diagonal;
  d.opt.function_tolerance = o.function_tolerance;
  d.opt.gradient_tolerance = o.gradient_tolerance;
  d.opt.parameter_tolerance = o.parameter_tolerance;
  // Two independently pipelined groups for batches: the groups drift out of phase, so the latency-bound block-column chain of
  // one overlaps the throughput kernels of the other (measured on the 256-scene C2 batch: 35.0k -> 36.8k LM it/s; four groups
  // 36.2k, six slower than one).  Small batches stay in one group.
  b->n_group = n >= 32 ? 2 : 1;
  if (const char* e = getenv("PTZ_BA_STREAMS")) b->n_group = std::max(1, atoi(e));
  b->lookahead = n >= 8;  // look-ahead pays for mid-size batches; a single scene is better off with fewer launches
  if (const char* e = getenv("PTZ_BA_LOOKAHEAD")) b->lookahead = atoi(e) != 0;
  // batches: left-looking column updates (half the tile traffic, no trailing-update launches); a few scenes: right-looking,
  // whose updates of one column step spread over many workgroups instead of looping inside one (measured: 256 scenes
  // 141.9 -> 137.8 ms, single scene 16.4 -> 19.4 ms with the left-looking form)
  b->left_looking = n >= 8;
  if (const char* e = getenv("PTZ_BA_CHOL_LEFT")) b->left_looking = atoi(e) != 0;
  // (small batches -- the view batches of the incremental pipeline converge in two or three passes -- run two ahead: every pass
  //  enqueued beyond the last real one is a dozen empty launches; large batches, whose passes last milliseconds, keep three)
  b->ahead = n <= 32 ? 2 : 3;
  if (const char* e = getenv("PTZ_BA_AHEAD")) b->ahead = std::max(1, atoi(e));
  if (const char* e = getenv("PTZ_BA_DEBUG_STALL")) b->d.debug_stall = std::max(0, atoi(e));
  if (const char* e = getenv("PTZ_BA_DEBUG_CHAIN_SPIN")) b->d.chol.chain_spin_limit = std::max(0, atoi(e));  // tests: hand-overs that time out
  b->d.chol.chain_ready_whole = 1;  // (Dev is zero-filled at creation)
  b->d.chol.chain_pair = 1;
  b->d.chol.chain_w0 = 1;
  if (const char* e = getenv("PTZ_BA_CHAIN_W0")) b->d.chol.chain_w0 = atoi(e) != 0;
  if (const char* e = getenv("PTZ_BA_CHAIN_PAIR")) b->d.chol.chain_pair = atoi(e) != 0;
  if (const char* e = getenv("PTZ_BA_CHAIN_READY_WHOLE")) b->d.chol.chain_ready_whole = atoi(e) != 0;       // A/B: 0 = four block rounds also for finished columns
  if (const char* e = getenv("PTZ_BA_GRAPH")) b->use_graph = atoi(e) != 0;
  b->ctl_groups = std::max(1, std::min(b->n_group, n));
  if (b->alloc(&b->d_ctl, (size_t)4 * b->ctl_groups) != PTZ_OK ||
      ptzpool::pinned_acquire(sizeof(int) * 4 * b->ctl_groups, (void**)&b->h_ctl) != hipSuccess ||
      hipHostGetDevicePointer((void**)&b->h_ctl_dev, b->h_ctl, 0) != hipSuccess) {
    ptz_ba_batch_destroy(b);
    return PTZ_ENODEVICE;
  }
  make_groups(b);
  b->stream = b->streams.empty() ? nullptr : b->streams[0];
  if (b->stream == nullptr || ptzpool::event_acquire(b->device, true, &b->ev0) != hipSuccess ||
      ptzpool::event_acquire(b->device, true, &b->ev1) != hipSuccess) {
    ptz_ba_batch_destroy(b);
    return PTZ_ENODEVICE;
  }
  // kernels that stage camera tables need > 64 KiB of dynamic LDS for large rigs
  // Rigs whose camera tables do not fit in LDS (the 160 KiB hold ~340 cameras) read them from global memory instead: the
  // reference has no cap on the number of views (ptzray_optimizer.cc:799-885)
  b->gtab = sizeof(double) * ((size_t)b->max_cam * (CBS + CDS + (NC | 1)) + 16 + 18) + (size_t)OBS_PREFETCH_BYTES * 256 > 160 * 1024;
  if (const char* e = getenv("PTZ_BA_GLOBAL_TABLES")) b->gtab = atoi(e) != 0;
  b->compaction = true;
  if (const char* e = getenv("PTZ_BA_COMPACT")) b->compaction = atoi(e) != 0;
  auto make_shape = [&](int slots, bool compact) {
    PassShape sh;
    sh.slots = slots; sh.compact = compact;
    // rays per workgroup of the ray-centric kernels: one rig's ~13 k rays on 1024-ray workgroups keep 14 compute units busy,
    // so a few scenes use small workgroups; large batches amortise the LDS camera tables over many rays (results do not
    // depend on it: per-ray sums are reduced per wave of 64 rays)
    sh.ray_block = slots <= 4 ? 128 : (slots <= 32 ? 256 : RAY_BLOCK);
    if (const char* e = getenv("PTZ_BA_RAY_BLOCK")) sh.ray_block = std::min(RAY_BLOCK, std::max(64, (atoi(e) / 64) * 64));
    sh.small_blocks = sh.ray_block <= 256;
    // one launch per factorisation (chol_chain_kernel) or per step (chol_col_step_kernel) for a few systems; up to CHOL_CHAIN_SLOTS when
    // the one-launch form takes them (chol_chain_fits)
    sh.fused = slots <= 8 || chol_chain_fits(slots, b->d.chol.np);
    if (const char* e = getenv("PTZ_BA_CHOL_FUSED")) sh.fused = atoi(e) != 0 && sh.fused;
    // a few scenes: LM control and the camera update ride in the tails / prologue of k_eval and k_lin_cam (three launches less per pass)
    sh.fuse_ctl = slots <= 8 && sh.small_blocks && !b->gtab;
    if (const char* e = getenv("PTZ_BA_FUSE_CTL")) sh.fuse_ctl = sh.fuse_ctl && atoi(e) != 0;
    sh.eval_lanes = (sh.fuse_ctl && sh.ray_block <= 128) ? 4 : 1;
    if (const char* e = getenv("PTZ_BA_EVAL_LANES")) sh.eval_lanes = (atoi(e) == 4 && sh.fuse_ctl && sh.ray_block <= 128) ? 4 : 1;  // (tests: the same bits either way)
    sh.max_chunk = (b->max_ray + sh.ray_block - 1) / sh.ray_block;
    const size_t obs_lds = sh.small_blocks ? (size_t)OBS_PREFETCH_BYTES * sh.ray_block : 0;
    if (b->gtab) { sh.eval_smem = sizeof(double) * 16 + obs_lds + 16; sh.lin_smem = obs_lds + 16; }
    else {
      sh.eval_smem = sizeof(double) * ((size_t)b->max_cam * (CBS + CDS + (NC | 1)) + 16 + 18) + obs_lds;
      sh.lin_smem = sizeof(double) * ((size_t)b->max_cam * CBS + 6) + obs_lds;
    }
    return sh;
  };
  b->shapes.clear();
  b->shapes.push_back(make_shape(n, false));
  // (two slots: the last rigs of a large batch get the one-launch factorisation, chol_chain_kernel; a batch of up to eight keeps
  //  its single shape -- a second one costs it a k_compact launch per pass)
  for (int sl = n > 8 ? 2 : 8; sl < n; sl *= 4) b->shapes.push_back(make_shape(sl, true));
  b->ray_block = b->shapes[0].ray_block;
  b->d.ray_block = b->ray_block;
  const int eval_smem = (int)b->shapes[0].eval_smem, lin_smem = (int)b->shapes[0].lin_smem;
  // k_schur keeps a camera's T_a rows in LDS (up to ~1700 observations of one view); beyond that the table goes to global
  // memory.  What remains is the 16-bit position inside the camera-pair entry records: 65535 observations per view.
  if (b->max_cam_obs > 65535) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
  // (also when a view has more runs than a workgroup has threads -- more pairs than that: the several rounds k_schur then needs
  // cannot reuse the table's LDS space for their sums)
  const int NW2d = NC - ((has3d && type != PTZ_BA_PTZRayFxfyDist) ? 1 : 0);  // Dims<TYPE>::NW
  b->d.schur_ent_cap = b->max_cam_ent;
  for (auto& dgp : b->dg) dgp.schur_ent_cap = b->max_cam_ent;
  // PTZRay: the factored rows of k_schur_f (8 doubles per observation instead of 15: three workgroups per compute unit, and
  // the LDS table holds views of ~2000 observations); PTZ_BA_SCHUR_F=0 keeps k_schur (A/B measurements)
  b->schur_f = type == PTZ_BA_PTZRay && !b->schur_w;
  if (const char* e = getenv("PTZ_BA_SCHUR_F")) b->schur_f = b->schur_f && atoi(e) != 0;
  const int frow = b->schur_f ? SCHUR_F_ROW : 0;
  b->d.e_fold = b->schur_f ? 1 : 0;
  for (auto& dgp : b->dg) dgp.e_fold = b->d.e_fold;
  b->schur_tg = schur_lds_bytes(b->max_cam_obs, NC, b->d.chol.np, b->schur_w, schur_threads_of(type), NW2d, b->max_cam_ent, frow) > 160 * 1024 ||
                (!b->schur_w && b->max_cam_run > schur_threads_of(type));
  if (!b->schur_w && schur_lds_bytes(0, NC, b->d.chol.np, false, schur_threads_of(type), NW2d, b->max_cam_ent, frow) > 160 * 1024) {
    ptz_ba_batch_destroy(b);  // a view with more pair entries than the LDS copy of its list holds (~60 000)
    return PTZ_ELIMIT;
  }
  if (const char* e = getenv("PTZ_BA_SCHUR_GLOBAL_T")) b->schur_tg = atoi(e) != 0;
  if (b->schur_tg) {
    const int rc2 = b->alloc(&b->d.Tbuf, (size_t)b->total_obs * ((NC * 3 + 3) | 1));
    if (rc2) { ptz_ba_batch_destroy(b); return rc2; }
    for (auto& dgp : b->dg) dgp.Tbuf = b->d.Tbuf;
  }
  if (eval_smem > 160 * 1024 || lin_smem > 160 * 1024) { ptz_ba_batch_destroy(b); return PTZ_ELIMIT; }
  {
    // The cap on dynamic LDS is a property of the kernel, not of a batch: raise it to the hardware limit once per device
    // and instantiation, so that batches of different sizes can live side by side (a per-batch value would let a small
    // batch created later lower the cap under a large one).
    static std::mutex attr_mu;
    static std::vector<char> attr_done;
    std::lock_guard<std::mutex> lk(attr_mu);
    if ((int)attr_done.size() <= o.device_id) attr_done.resize(o.device_id + 1, 0);
    if (!attr_done[o.device_id]) {
      bool attr_ok = true;
      auto raise_cap = [&](const void* fn) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { attr_ok = false; return; }
        const int cap = 160 * 1024 - (int)fa.sharedSizeBytes;  // the statically declared part counts against the same 160 KB
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess) attr_ok = false;
      };
#define PTZ_SET_ATTR(T)                        \
      raise_cap((const void*)k_schur<T, false>);      \
      raise_cap((const void*)k_schur<T, true>);       \
      raise_cap((const void*)k_schur_w<T, false>);    \
      raise_cap((const void*)k_schur_w<T, true>);     \
      raise_cap((const void*)k_eval<T, true, false>);       \
      raise_cap((const void*)k_eval<T, false, false>);      \
      raise_cap((const void*)k_lin_ray<T, true, false>);    \
      raise_cap((const void*)k_lin_ray<T, false, false>);
      raise_cap((const void*)k_eval<0, true, false, true>);
      raise_cap((const void*)k_eval<0, true, false, true, 4>);
      raise_cap((const void*)k_eval<1, true, false, true, 4>);
      raise_cap((const void*)k_eval<2, true, false, true, 4>);
      raise_cap((const void*)k_schur_f<0, false>);
      raise_cap((const void*)k_schur_f<0, true>);
      raise_cap((const void*)k_schur_f<4, false>);
      raise_cap((const void*)k_schur_f<4, true>);
      raise_cap((const void*)k_eval<1, true, false, true>);
      raise_cap((const void*)k_eval<2, true, false, true>);
      PTZ_SET_ATTR(0) PTZ_SET_ATTR(1) PTZ_SET_ATTR(2) PTZ_SET_ATTR(3) PTZ_SET_ATTR(4) PTZ_SET_ATTR(5) PTZ_SET_ATTR(6) PTZ_SET_ATTR(7)
#undef PTZ_SET_ATTR
      if (!attr_ok) { (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
      attr_done[o.device_id] = 1;
    }
  }
  // The zero fills and the staged upload are still in flight on `io`: the batch's first stream waits for them ON THE DEVICE (every
  // later entry point works on that stream or behind it), the host does not -- it goes on to the caller's set_state / solve while
  // the device finishes the structure.  (The staging blocks are released behind the next host-side wait: solve, destroy.)
  if (ptzpool::event_acquire(b->device, false, &b->create_ev) != hipSuccess || hipEventRecord(b->create_ev, b->io) != hipSuccess ||
      hipStreamWaitEvent(b->stream, b->create_ev, 0) != hipSuccess) {
    (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE;
  }
  tc3 = now_ms();
  if (dbg_t) fprintf(stderr, "[ptz_ba_create] host structure %.2f ms (observations %.2f, pair entries %.2f), uploads + allocations %.2f ms, mask + rest %.2f ms\n", tc1 - tc0, ts_obs, ts_ent, tc2 - tc1, tc3 - tc2);
  *out = b;
  return PTZ_OK;
}

// the ray order of a batch built from views lives on the device: fetched when a host-side entry point needs it
static int32_t ensure_host_ray_perm(ptz_ba_batch* b)
{
  if (!b->ray_perm.empty() || !b->d_ray_perm) return PTZ_OK;
  b->ray_perm.resize((size_t)b->total_ray);
  if (copy_on(b->stream, b->ray_perm.data(), b->d_ray_perm, sizeof(int) * b->ray_perm.size(), hipMemcpyDeviceToHost) != hipSuccess) return PTZ_ENODEVICE;
  return PTZ_OK;
}

int32_t ptz_ba_batch_set_state(ptz_ba_batch* b, const double* cam, const double* ray, const double* tlw)
{
  if (!b || !cam || !ray) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  if (int32_t rcp = ensure_host_ray_perm(b)) return rcp;
  // cameras (a shared block starting from its first camera's values), rays in the library's order and T_l_w are put together in
  // ONE pinned block, sent with three asynchronous copies and waited for ONCE (three staged copies with a wait each cost the
  // lock-step PTZ-IBA 0.4 ms per bundle adjustment); batches whose state is larger than the staging limit copy array by array
  const size_t nc = (size_t)15 * b->total_cam, nr = (size_t)3 * b->total_ray, nt = (size_t)6 * b->n_scene;
  const size_t bytes = sizeof(double) * (nc + nr + nt);
  void* pin = nullptr;
  std::vector<double> heap;
  double* st = nullptr;
  if (bytes <= ((size_t)32 << 20) && ptzpool::pinned_acquire(bytes, &pin) == hipSuccess) st = static_cast<double*>(pin);
  else { (void)hipGetLastError(); heap.resize(nc + nr + nt); st = heap.data(); }
  double *c = st, *r = st + nc, *t = st + nc + nr;
  memcpy(c, cam, sizeof(double) * nc);
  if (b->d.shared) {
    // a shared block starts from its first camera's values (intrinsics_param_.insert, ptzray_optimizer.cc:645-650)
    for (int i = 0; i < b->total_cam; ++i) {
      const int f = b->first_of_group[i];
      if (f == i) continue;
      for (int k = 0; k < 15; ++k)
        if (k < 4 || k >= 10) c[15 * (size_t)i + k] = c[15 * (size_t)f + k];
    }
  }
  for (const SceneDev& sd : b->scenes)
    for (int j = 0; j < sd.n_ray; ++j) {
      const double* src = ray + 3 * ((size_t)sd.ray_off + b->ray_perm[sd.ray_off + j]);
      double* dst = r + 3 * ((size_t)sd.ray_off + j);
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
  if (tlw) memcpy(t, tlw, sizeof(double) * nt);
  else memset(t, 0, sizeof(double) * nt);
  hipError_t e = hipMemcpyAsync(b->cam0, c, sizeof(double) * nc, hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(b->ray0, r, sizeof(double) * nr, hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(b->tlw0, t, sizeof(double) * nt, hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = stream_wait(b->stream);
  ptzpool::pinned_release(pin);
  PTZ_HIP_TRY(e);
  b->has_state = true;
  return PTZ_OK;
}

int32_t ptz_ba_batch_solve(ptz_ba_batch* b, ptz_lm_summary* summaries)
{
  if (!b || !b->has_state) return PTZ_EINVAL;
  clear_stale_error(__func__);
  PTZ_DEVICE_GUARD(b->device);
  switch (b->type + 4 * b->has3d) {  // Dims<TYPE>
    case 0: return solve_impl<0>(b, summaries);
    case 1: return solve_impl<1>(b, summaries);
    case 2: return solve_impl<2>(b, summaries);
    case 3: return solve_impl<3>(b, summaries);
    case 4: return solve_impl<4>(b, summaries);
    case 5: return solve_impl<5>(b, summaries);
    case 6: return solve_impl<6>(b, summaries);
    default: return solve_impl<7>(b, summaries);
  }
}

int32_t ptz_ba_batch_get_state(ptz_ba_batch* b, double* cam, double* ray, double* tlw)
{
  if (!b) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  // The current half of every scene's double-buffered state is gathered ON THE DEVICE, in the caller's order (the library
  // numbers a scene's rays by track length, ray_perm), and comes back in one copy per array: round 2 issued two small
  // synchronous copies per scene (6 ms for 64 rigs, 30 ms for the 1000 scenes of C4).
  const size_t nc = (size_t)b->total_cam * 15, nr = (size_t)b->total_ray * 3, nt = (size_t)b->n_scene * 6;
  void* stage = nullptr;
  if (ptzpool::dev_acquire(b->device, sizeof(double) * (nc + nr + nt), &stage) != hipSuccess) return PTZ_ENOMEM;
  double* s_cam = static_cast<double*>(stage);
  double* s_tlw = s_cam + nc;
  double* s_ray = s_tlw + nt;  // (last: a caller that does not want the rays does not pay for their copy)
  hipLaunchKernelGGL(k_gather_state, dim3(std::max(1, (std::max(b->max_cam * 15, b->max_ray) + 255) / 256), b->n_scene), dim3(256), 0, b->stream, b->d,
                     b->d_ray_perm, s_cam, s_ray, s_tlw);
  // the gathered state comes back in ONE copy into a pinned block (the kernel and the copy are ordered on the batch's stream: one
  // wait), from where the caller's arrays are filled
  const size_t bytes = sizeof(double) * (nc + nt + (ray ? nr : 0));
  void* pin = nullptr;
  hipError_t e = hipSuccess;
  if (bytes <= ((size_t)32 << 20) && ptzpool::pinned_acquire(bytes, &pin) == hipSuccess) {
    e = hipMemcpyAsync(pin, stage, bytes, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = stream_wait(b->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) {
      const double* p = static_cast<const double*>(pin);
      if (cam) memcpy(cam, p, sizeof(double) * nc);
      if (tlw) memcpy(tlw, p + nc, sizeof(double) * nt);
      if (ray) memcpy(ray, p + nc + nt, sizeof(double) * nr);
    }
    ptzpool::pinned_release(pin);
  }
  else {
    (void)hipGetLastError();
    e = stream_wait(b->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && cam) e = copy_on(b->stream, cam, s_cam, sizeof(double) * nc, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ray) e = copy_on(b->stream, ray, s_ray, sizeof(double) * nr, hipMemcpyDeviceToHost);
    if (e == hipSuccess && tlw) e = copy_on(b->stream, tlw, s_tlw, sizeof(double) * nt, hipMemcpyDeviceToHost);
  }
  ptzpool::dev_release(b->device, stage);
  if (e != hipSuccess) { (void)hipGetLastError(); return PTZ_ENODEVICE; }
  return PTZ_OK;
}

int32_t ptz_ba_batch_set_disp(ptz_ba_batch* b, const double* disp)
{
  if (!b || !disp) return PTZ_EINVAL;
  if (b->type != PTZ_BA_PTZRayDistDisp) return PTZ_EUNSUPPORTED;
  PTZ_DEVICE_GUARD(b->device);
  std::vector<double> v((size_t)3 * b->total_cam);  // every camera carries a copy of its scene's block
  for (const SceneDev& sd : b->scenes)
    for (int c = 0; c < sd.n_cam; ++c)
      for (int k = 0; k < 3; ++k) v[3 * ((size_t)sd.cam_off + c) + k] = disp[3 * (size_t)sd.idx + k];
  PTZ_HIP_TRY(copy_on(b->stream, b->dsp0, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice));
  return PTZ_OK;
}

int32_t ptz_ba_batch_get_disp(ptz_ba_batch* b, double* disp)
{
  if (!b || !disp) return PTZ_EINVAL;
  if (b->type != PTZ_BA_PTZRayDistDisp) return PTZ_EUNSUPPORTED;
  PTZ_DEVICE_GUARD(b->device);
  std::vector<LmState> h(b->n_scene);
  PTZ_HIP_TRY(copy_on(b->stream, h.data(), b->d.lm, sizeof(LmState) * b->n_scene, hipMemcpyDeviceToHost));
  for (int i = 0; i < b->n_scene; ++i)  // camera 0's copy (all copies of a scene are equal)
    PTZ_HIP_TRY(copy_on(b->stream, disp + 3 * (size_t)i, b->d.dsp_x + h[i].cur * b->d.dsp_stride + 3 * (size_t)b->scenes[i].cam_off,
                          sizeof(double) * 3, hipMemcpyDeviceToHost));
  return PTZ_OK;
}

int32_t ptz_ba_batch_last_solve_ms(const ptz_ba_batch* b, double* ms)
{
  if (!b || !ms) return PTZ_EINVAL;
  *ms = b->last_ms;
  return PTZ_OK;
}

int32_t ptz_ba_batch_set_profiling(ptz_ba_batch* b, int32_t enable)
{
  if (!b) return PTZ_EINVAL;
  b->profiling = enable != 0;
  for (int i = 0; i < PTZ_PROF_SLOTS; ++i) { b->prof_ms[i] = 0; b->prof_n[i] = 0; }
  return PTZ_OK;
}

int32_t ptz_ba_batch_get_profile(const ptz_ba_batch* b, double* ms_per_slot, int64_t* launches_per_slot, const char** slot_names)
{
  if (!b) return PTZ_EINVAL;
  for (int i = 0; i < PTZ_PROF_SLOTS; ++i) {
    if (ms_per_slot) ms_per_slot[i] = b->prof_ms[i];
    if (launches_per_slot) launches_per_slot[i] = b->prof_n[i];
    if (slot_names) slot_names[i] = kSlotNames[i];
  }
  return PTZ_OK;
}

int32_t ptz_ba_batch_pix2ray(ptz_ba_batch* b)
{
  if (!b || !b->has_state) return PTZ_EINVAL;
  clear_stale_error(__func__);
  PTZ_DEVICE_GUARD(b->device);
  hipLaunchKernelGGL(k_pix2ray, dim3(b->shapes[0].max_chunk, b->n_scene), dim3(b->shapes[0].ray_block), 0, b->stream, b->d, b->cam0, b->ray0);
  PTZ_HIP_TRY(stream_wait(b->stream));
  PTZ_HIP_TRY(hipGetLastError());
  return PTZ_OK;
}

int32_t ptz_ba_batch_linearize(ptz_ba_batch* b, int32_t index, double* cost, double* g_c, double* U, double* g_r, double* V, double* W)
{
  clear_stale_error(__func__);
  if (!b || !b->has_state || index < 0 || index >= b->n_scene) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  if (int32_t rcp = ensure_host_ray_perm(b)) return rcp;
  if (W && !b->d.W) {  // the rows exist only for this call's callers (the solver does not materialise them)
    const bool disp = b->type == PTZ_BA_PTZRayDistDisp;
    const int rcw = b->alloc(&b->d.W, (size_t)b->total_obs * (disp ? 24 : b->type == PTZ_BA_PTZRayFxfyDist ? 18 : 16));
    if (rcw) return rcw;
  }
  const Dev& d = b->d;
  const int NC = b->nc;
  hipStream_t st = b->stream;
  PTZ_HIP_TRY(hipMemcpyAsync(d.cam_x, b->cam0, sizeof(double) * 15 * b->total_cam, hipMemcpyDeviceToDevice, st));
  PTZ_HIP_TRY(hipMemcpyAsync(d.ray_x, b->ray0, sizeof(double) * 3 * b->total_ray, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_reset, dim3((b->n_scene + 63) / 64), dim3(64), 0, st, d);
  hipLaunchKernelGGL(k_fill, dim3(((size_t)b->total_cam * NC + 255) / 256), dim3(256), 0, st, d.scale_c, (size_t)b->total_cam * NC, 1.0);
  hipLaunchKernelGGL(k_fill, dim3(((size_t)b->total_ray * 3 + 255) / 256), dim3(256), 0, st, d.scale_r, (size_t)b->total_ray * 3, 1.0);
  PTZ_HIP_TRY(hipMemcpyAsync(d.tlw_x, b->tlw0, sizeof(double) * 6 * b->n_scene, hipMemcpyDeviceToDevice, st));
  PTZ_HIP_TRY(hipMemcpyAsync(d.tlw_x + d.tlw_stride, b->tlw0, sizeof(double) * 6 * b->n_scene, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_fill, dim3((6 * b->n_scene + 255) / 256), dim3(256), 0, st, d.scale_t, (size_t)6 * b->n_scene, 1.0);
  if (d.dsp_x) PTZ_HIP_TRY(hipMemcpyAsync(d.dsp_x, b->dsp0, sizeof(double) * d.dsp_stride, hipMemcpyDeviceToDevice, st));
  switch (b->type + 4 * b->has3d) {  // Dims<TYPE>
    case 0: enqueue_linearize<0>(b); break;
    case 1: enqueue_linearize<1>(b); break;
    case 2: enqueue_linearize<2>(b); break;
    case 3: enqueue_linearize<3>(b); break;
    case 4: enqueue_linearize<4>(b); break;
    case 5: enqueue_linearize<5>(b); break;
    case 6: enqueue_linearize<6>(b); break;
    default: enqueue_linearize<7>(b); break;
  }
  PTZ_HIP_TRY(stream_wait(st));
  PTZ_HIP_TRY(hipGetLastError());
  const SceneDev& s = b->scenes[index];
  if (cost) {
    std::vector<double> c(s.n_cam);
    PTZ_HIP_TRY(copy_on(b->stream, c.data(), d.costc + s.cam_off, sizeof(double) * s.n_cam, hipMemcpyDeviceToHost));
    double t = 0;
    for (double v : c) t += v;
    *cost = t;
  }
  if (g_c) PTZ_HIP_TRY(copy_on(b->stream, g_c, d.gc + (size_t)s.cam_off * NC, sizeof(double) * NC * s.n_cam, hipMemcpyDeviceToHost));
  if (U) PTZ_HIP_TRY(copy_on(b->stream, U, d.U + (size_t)s.cam_off * NC * NC, sizeof(double) * NC * NC * s.n_cam, hipMemcpyDeviceToHost));
  const int* perm = b->ray_perm.data() + s.ray_off;  // internal ray j -> the caller's ray index
  if (g_r) {
    std::vector<double> g3((size_t)s.n_ray * 3);
    PTZ_HIP_TRY(copy_on(b->stream, g3.data(), d.gr + (size_t)s.ray_off * 3, sizeof(double) * 3 * s.n_ray, hipMemcpyDeviceToHost));
    for (int j = 0; j < s.n_ray; ++j)
      for (int k = 0; k < 3; ++k) g_r[(size_t)perm[j] * 3 + k] = g3[(size_t)j * 3 + k];
  }
  if (V) {
    std::vector<double> v6((size_t)s.n_ray * 6);
    PTZ_HIP_TRY(copy_on(b->stream, v6.data(), d.V + (size_t)s.ray_off * 6, sizeof(double) * 6 * s.n_ray, hipMemcpyDeviceToHost));
    for (int j = 0; j < s.n_ray; ++j) {
      const double* p = &v6[(size_t)j * 6];
      double* q = V + (size_t)perm[j] * 9;
      q[0] = p[0]; q[1] = p[1]; q[2] = p[3]; q[3] = p[1]; q[4] = p[2]; q[5] = p[4]; q[6] = p[3]; q[7] = p[4]; q[8] = p[5];
    }
  }
  if (W) {  // rows are stored camera-major; return them in observation order
    const int NW = b->type == PTZ_BA_PTZRayFxfyDist ? 6 : NC - b->has3d;  // W carries the 2D-2D columns only (PTZRayDistDisp: 8)
    const int ws = (NW * 3 + 1) & ~1;        // Dims<TYPE>::WS
    std::vector<double> rows((size_t)s.n_obs * ws);
    std::vector<int> wp(s.n_obs);
    PTZ_HIP_TRY(copy_on(b->stream, rows.data(), d.W + (size_t)s.obs_off * ws, sizeof(double) * ws * s.n_obs, hipMemcpyDeviceToHost));
    PTZ_HIP_TRY(copy_on(b->stream, wp.data(), d.wpos + s.obs_off, sizeof(int) * s.n_obs, hipMemcpyDeviceToHost));
    // internal observation order is (internal ray, order inside the track); the caller's is (caller's ray, same inner order)
    std::vector<int> rp(s.n_ray + 1);
    PTZ_HIP_TRY(copy_on(b->stream, rp.data(), d.ray_ptr + s.ray_off + s.idx, sizeof(int) * (s.n_ray + 1), hipMemcpyDeviceToHost));
    std::vector<int> ext_first(s.n_ray + 1, 0);
    for (int j = 0; j < s.n_ray; ++j) ext_first[perm[j] + 1] = rp[j + 1] - rp[j];
    for (int r = 0; r < s.n_ray; ++r) ext_first[r + 1] += ext_first[r];
    for (int j = 0; j < s.n_ray; ++j)
      for (int a = rp[j]; a < rp[j + 1]; ++a) {
        const int a_int = a - s.obs_off, a_ext = ext_first[perm[j]] + (a - rp[j]);
        memcpy(W + (size_t)a_ext * NW * 3, &rows[(size_t)(wp[a_int] - s.obs_off) * ws], sizeof(double) * NW * 3);
      }
  }
  return PTZ_OK;
}

int32_t ptz_rig_create(int32_t n_img, int32_t n_track, const int64_t* trk_ptr, const int32_t* trk_img, const float* trk_uv, int32_t device_id,
                       ptz_rig** out)
{
  if (n_img <= 0 || n_track <= 0 || !trk_ptr || !trk_img || !trk_uv || !out) return PTZ_EINVAL;
  *out = nullptr;
  const int64_t nv = trk_ptr[n_track];
  if (trk_ptr[0] != 0 || nv <= 0 || nv > 0x7fffffff) return PTZ_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id) return PTZ_ENODEVICE;
  ptz_rig* r = new ptz_rig();
  r->device = device_id; r->n_img = n_img; r->n_track = n_track; r->n_view = nv;
  r->img_obs.assign(n_img, 0); r->img_ent.assign(n_img, 0);
  std::vector<int> ptr32(n_track + 1);
  for (int t = 0; t <= n_track; ++t) ptr32[t] = (int)trk_ptr[t];
  for (int t = 0; t < n_track; ++t) {
    const int64_t L = trk_ptr[t + 1] - trk_ptr[t];
    if (L <= 0) { delete r; return PTZ_EINVAL; }
    r->ent_bound += L * (L - 1) / 2;
    r->max_track_len = (int)std::max<int64_t>(r->max_track_len, L);
    for (int64_t e = trk_ptr[t]; e < trk_ptr[t + 1]; ++e) {
      const int im = trk_img[e];
      if (im < 0 || im >= n_img || (e > trk_ptr[t] && im <= trk_img[e - 1])) { delete r; return PTZ_EINVAL; }  // images ascend inside a track
      ++r->img_obs[im];
      r->img_ent[im] += (int)(e - trk_ptr[t]);  // pairs in which this view is the higher camera
    }
  }
  if (r->ent_bound > 0x7fffffff) { delete r; return PTZ_ELIMIT; }
  PTZ_DEVICE_GUARD(device_id);
  hipStream_t st = nullptr;
  if (ptzpool::stream_acquire(device_id, &st) != hipSuccess) { delete r; return PTZ_ENODEVICE; }
  auto up = [&](const void* src, size_t bytes, const void** dst) -> bool {
    void* q = nullptr;
    if (ptzpool::dev_acquire(device_id, bytes, &q) != hipSuccess) return false;
    r->allocs.push_back(q);
    *dst = q;
    return copy_on(st, q, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  const bool ok = up(ptr32.data(), sizeof(int) * ptr32.size(), (const void**)&r->d_trk_ptr) && up(trk_img, sizeof(int) * (size_t)nv, (const void**)&r->d_trk_img) &&
                  up(trk_uv, sizeof(float) * 2 * (size_t)nv, (const void**)&r->d_trk_uv);
  ptzpool::stream_release(device_id, st);
  if (!ok) { ptz_rig_destroy(r); return PTZ_ENODEVICE; }
  *out = r;
  return PTZ_OK;
}

void ptz_rig_destroy(ptz_rig* r)
{
  if (!r) return;
  DeviceGuard guard(r->device);
  for (void* p : r->allocs) ptzpool::dev_release(r->device, p);
  delete r;
}

int32_t ptz_ba_batch_set_state_pix2ray(ptz_ba_batch* b, const double* cam, const double* rkinv)
{
  if (!b || !cam || !rkinv) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  clear_stale_error(__func__);
  // cameras and their R^-1 K^-1 in one pinned block, two copies, the ray kernel behind them, one wait
  const size_t nc = (size_t)15 * b->total_cam, nk = (size_t)9 * b->total_cam;
  void* pin = nullptr;
  std::vector<double> heap;
  double* stg = nullptr;
  if (ptzpool::pinned_acquire(sizeof(double) * (nc + nk), &pin) == hipSuccess) stg = static_cast<double*>(pin);
  else { (void)hipGetLastError(); heap.resize(nc + nk); stg = heap.data(); }
  memcpy(stg, cam, sizeof(double) * nc);
  memcpy(stg + nc, rkinv, sizeof(double) * nk);
  // (no wait: the solve that follows runs on the same stream; the staging block lives until that solve's wait, the R^-1 K^-1 block
  //  until the batch goes)
  if (!b->rkinv_dev && b->alloc(&b->rkinv_dev, nk) != PTZ_OK) { if (pin) ptzpool::pinned_release(pin); return PTZ_ENOMEM; }
  double* dk = b->rkinv_dev;  // (one block per batch, reused by later calls: they are ordered on the batch's stream)
  hipError_t e = hipMemcpyAsync(b->cam0, stg, sizeof(double) * nc, hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dk, stg + nc, sizeof(double) * nk, hipMemcpyHostToDevice, b->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b->tlw0, 0, sizeof(double) * 6 * b->n_scene, b->stream);
  if (e == hipSuccess)
    hipLaunchKernelGGL(k_view_pix2ray, dim3((b->max_ray + 255) / 256, b->n_scene), dim3(256), 0, b->stream, b->d, (const double*)b->cam0, (const double*)dk, b->ray0);
  if (e == hipSuccess && pin) b->staged_pinned.push_back(pin);  // (in flight)
  else { (void)stream_wait(b->stream); if (pin) ptzpool::pinned_release(pin); }  // pageable staging or a refused copy: nothing may still read it
  if (e == hipSuccess) e = hipGetLastError();
  PTZ_HIP_TRY(e);
  b->has_state = true;
  return PTZ_OK;
}

int32_t ptz_debug_batch_initial_rays(ptz_ba_batch* b, double* ray)
{
  if (!b || !ray || !b->has_state) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  if (int32_t rcp = ensure_host_ray_perm(b)) return rcp;
  std::vector<double> r((size_t)3 * b->total_ray);
  if (copy_on(b->stream, r.data(), b->ray0, sizeof(double) * r.size(), hipMemcpyDeviceToHost) != hipSuccess) return PTZ_ENODEVICE;
  for (const SceneDev& sd : b->scenes)
    for (int j = 0; j < sd.n_ray; ++j)
      for (int k = 0; k < 3; ++k) ray[3 * ((size_t)sd.ray_off + b->ray_perm[sd.ray_off + j]) + k] = r[3 * ((size_t)sd.ray_off + j) + k];
  return PTZ_OK;
}

int32_t ptz_debug_batch_structure_hash(ptz_ba_batch* b, uint64_t* hash)
{
  if (!b || !hash) return PTZ_EINVAL;
  PTZ_DEVICE_GUARD(b->device);
  const Dev& d = b->d;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* c = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
  std::vector<unsigned char> buf;
  auto fetch = [&](const void* dev, size_t bytes) -> const unsigned char* {
    buf.resize(bytes + 8);
    if (bytes && copy_on(b->stream, buf.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    return buf.data();
  };
  // every scene's arrays over their real extents, positions made scene-relative (a batch built from views keeps gaps between scenes)
  std::vector<SceneDev> sc(b->n_scene);
  if (copy_on(b->stream, sc.data(), d.scene, sizeof(SceneDev) * b->n_scene, hipMemcpyDeviceToHost) != hipSuccess) return PTZ_ENODEVICE;
  for (int i = 0; i < b->n_scene; ++i) {
    const SceneDev& s = sc[i];
    const int counts[6] = {s.n_cam, s.n_ray, s.n_obs, s.n_pair, s.n_wave, s.n};
    mix(counts, sizeof(counts));
    auto plain = [&](const void* base, size_t elem, size_t first, size_t count) { const unsigned char* p = fetch((const char*)base + elem * first, elem * count); if (p) mix(p, elem * count); return p != nullptr; };
    auto rel = [&](const int* base, size_t first, size_t count, int minus) {
      const unsigned char* p = fetch(base + first, sizeof(int) * count);
      if (!p) return false;
      std::vector<int> v(count);
      memcpy(v.data(), p, sizeof(int) * count);
      for (int& x : v) x -= minus;
      mix(v.data(), sizeof(int) * count);
      return true;
    };
    bool ok = plain(d.obs_uv, sizeof(float2), s.obs_off, s.n_obs) && plain(d.obs_cam, 4, s.obs_off, s.n_obs) && plain(d.obs_ray, 4, s.obs_off, s.n_obs);
    ok = ok && rel(d.ray_ptr, (size_t)s.ray_off + s.idx, (size_t)s.n_ray + 1, s.obs_off) && rel(d.cam_ptr, (size_t)s.cam_off + s.idx, (size_t)s.n_cam + 1, s.obs_off);
    ok = ok && rel(d.cam_obs, s.obs_off, s.n_obs, s.obs_off) && rel(d.wpos, s.obs_off, s.n_obs, s.obs_off) && rel(d.cam_ray, s.obs_off, s.n_obs, s.ray_off);
    ok = ok && plain(d.cam_uv, sizeof(float2), s.obs_off, s.n_obs) && plain(d.ray_w, 8, s.ray_off, s.n_ray) && plain(b->d_ray_perm, 4, s.ray_off, s.n_ray);
    ok = ok && plain(d.pair_ci, 4, s.pair_off, s.n_pair) && plain(d.pair_cj, 4, s.pair_off, s.n_pair) && rel(d.pair_brow, s.pair_off, s.n_pair, s.obs_off);
    ok = ok && rel(d.pair_ptr, (size_t)s.pair_off + s.idx, (size_t)s.n_pair + 1, s.ent_off) && plain(d.pair_run, 4, (size_t)s.pair_off + s.idx, (size_t)s.n_pair + 1);
    ok = ok && plain(d.cam_pair, 4, (size_t)s.cam_off + s.idx, (size_t)s.n_cam + 1) && plain(d.cam_run, 4, (size_t)s.cam_off + s.idx, (size_t)s.n_cam + 1);
    if (!ok) return PTZ_ENODEVICE;
    // entries and runs: extents from the scene's own last offsets
    std::vector<int> last(2);
    if (copy_on(b->stream, &last[0], d.pair_ptr + s.pair_off + s.idx + s.n_pair, 4, hipMemcpyDeviceToHost) != hipSuccess) return PTZ_ENODEVICE;
    if (copy_on(b->stream, &last[1], d.cam_run + s.cam_off + s.idx + s.n_cam, 4, hipMemcpyDeviceToHost) != hipSuccess) return PTZ_ENODEVICE;
    const int n_ent = last[0] - s.ent_off, n_run = last[1];
    if (!plain(d.ent, 4, s.ent_off, (size_t)std::max(n_ent, 0))) return PTZ_ENODEVICE;
    {
      const unsigned char* p = fetch(d.run_rec + s.run_off, sizeof(uint2) * (size_t)std::max(n_run, 0));
      if (!p) return PTZ_ENODEVICE;
      std::vector<uint2> v((size_t)std::max(n_run, 0));
      memcpy(v.data(), p, sizeof(uint2) * v.size());
      for (uint2& x : v) x.x -= (unsigned)s.ent_off;
      mix(v.data(), sizeof(uint2) * v.size());
    }
    const int tail[2] = {n_ent, n_run};
    mix(tail, sizeof(tail));
  }
  *hash = h;
  return PTZ_OK;
}

void ptz_trim_cache(void) { ptzpool::trim(); }

int32_t ptz_ba_solve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt, ptz_lm_summary* summary)
{
  if (!p || !cam || !ray) return PTZ_EINVAL;
  const bool dbg = getenv("PTZ_BA_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  ptz_ba_batch* b = nullptr;
  int rc = ptz_ba_batch_create(1, p, opt, &b);
  if (rc) return rc;
  const double t1 = now();
  rc = ptz_ba_batch_set_state(b, cam, ray, tlw);
  const double t2 = now();
  if (!rc) rc = ptz_ba_batch_solve(b, summary);
  const double t3 = now();
  if (!rc) rc = ptz_ba_batch_get_state(b, cam, ray, tlw);
  const double t4 = now();
  ptz_ba_batch_destroy(b);
  if (dbg) fprintf(stderr, "[ptz_ba_solve] n_cam %d n_obs %lld: create %.2f set_state %.2f solve %.2f get_state %.2f destroy %.2f ms\n", p->n_cam,
                   (long long)p->n_obs, t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
  return rc;
}

int32_t ptz_ba_solve_disp(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, double* disp, const ptz_lm_options* opt,
                          ptz_lm_summary* summary)
{
  if (!p || !cam || !ray) return PTZ_EINVAL;
  if (p->factor_type != PTZ_BA_PTZRayDistDisp) return PTZ_EUNSUPPORTED;
  ptz_ba_batch* b = nullptr;
  int rc = ptz_ba_batch_create(1, p, opt, &b);
  if (rc) return rc;
  rc = ptz_ba_batch_set_state(b, cam, ray, tlw);
  if (!rc && disp) rc = ptz_ba_batch_set_disp(b, disp);
  if (!rc) rc = ptz_ba_batch_solve(b, summary);
  if (!rc) rc = ptz_ba_batch_get_state(b, cam, ray, tlw);
  if (!rc && disp) rc = ptz_ba_batch_get_disp(b, disp);
  ptz_ba_batch_destroy(b);
  return rc;
}

// Many scenes over several devices of one node, from one process: scenes are dealt longest-first to the least loaded device
// (they never interact -- run_ptzba_synthetic.sh:4-13 runs them as separate processes), every device gets one host thread
// that creates, solves and reads back its own batch, and the results are scattered back into the caller's arrays.  No
// collective: there is nothing to exchange.
int32_t ptz_ba_solve_sharded(int32_t n, const ptz_ba_problem* problems, double* cam, double* ray, double* tlw,
                             const int32_t* device_ids, int32_t n_devices, const ptz_lm_options* opt, ptz_lm_summary* summaries)
{
  if (n <= 0 || !problems || !cam || !ray || !device_ids || n_devices <= 0) return PTZ_EINVAL;
  ptz_lm_options base;
  if (opt) base = *opt; else ptz_lm_options_default(&base);
  std::vector<size_t> cam_off(n + 1, 0), ray_off(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    if (problems[i].n_cam <= 0 || problems[i].n_ray <= 0) return PTZ_EINVAL;
    cam_off[i + 1] = cam_off[i] + (size_t)problems[i].n_cam;
    ray_off[i + 1] = ray_off[i] + (size_t)problems[i].n_ray;
  }
  // longest-first greedy assignment by observation count
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int bq) { return problems[a].n_obs > problems[bq].n_obs; });
  std::vector<std::vector<int>> shard(n_devices);
  std::vector<int64_t> load(n_devices, 0);
  for (int i : order) {
    int best = 0;
    for (int dv = 1; dv < n_devices; ++dv) if (load[dv] < load[best]) best = dv;
    shard[best].push_back(i);
    load[best] += problems[i].n_obs;
  }
  std::vector<int> rcs(n_devices, PTZ_OK);
  auto run = [&](int dv) {
    std::vector<int>& ids = shard[dv];
    if (ids.empty()) return;
    std::sort(ids.begin(), ids.end());  // problem order inside the shard
    std::vector<ptz_ba_problem> probs;
    std::vector<double> c, r, t;
    for (int i : ids) {
      probs.push_back(problems[i]);
      c.insert(c.end(), cam + 15 * cam_off[i], cam + 15 * cam_off[i + 1]);
      r.insert(r.end(), ray + 3 * ray_off[i], ray + 3 * ray_off[i + 1]);
      if (tlw) t.insert(t.end(), tlw + 6 * (size_t)i, tlw + 6 * (size_t)(i + 1));
    }
    ptz_lm_options o = base;
    o.device_id = device_ids[dv];
    ptz_ba_batch* b = nullptr;
    int rc = ptz_ba_batch_create((int32_t)ids.size(), probs.data(), &o, &b);
    std::vector<ptz_lm_summary> summ(ids.size());
    if (!rc) rc = ptz_ba_batch_set_state(b, c.data(), r.data(), tlw ? t.data() : nullptr);
    if (!rc) rc = ptz_ba_batch_solve(b, summ.data());
    if (!rc) rc = ptz_ba_batch_get_state(b, c.data(), r.data(), tlw ? t.data() : nullptr);
    if (b) ptz_ba_batch_destroy(b);
    rcs[dv] = rc;
    if (rc) return;
    size_t co = 0, ro = 0;
    for (size_t k = 0; k < ids.size(); ++k) {
      const int i = ids[k];
      const size_t nc = (size_t)problems[i].n_cam, nr = (size_t)problems[i].n_ray;
      memcpy(cam + 15 * cam_off[i], c.data() + 15 * co, sizeof(double) * 15 * nc);
      memcpy(ray + 3 * ray_off[i], r.data() + 3 * ro, sizeof(double) * 3 * nr);
      if (tlw) memcpy(tlw + 6 * (size_t)i, t.data() + 6 * k, sizeof(double) * 6);
      if (summaries) summaries[i] = summ[k];
      co += nc; ro += nr;
    }
  };
  std::vector<std::thread> th;
  for (int dv = 1; dv < n_devices; ++dv) th.emplace_back(run, dv);
  run(0);
  for (auto& x : th) x.join();
  for (int rc : rcs) if (rc) return rc;
  return PTZ_OK;
}

// Diagnostic (not part of the drop-in boundary, used by tools/probes/probe_host_structure.py): the host-side structure stage of
// ptz_ba_batch_create -- build_pairs of every problem on `n_threads` threads, into scratch arrays -- without touching a device.
int32_t ptz_debug_host_structure(int32_t n, const ptz_ba_problem* problems, int32_t n_threads, int32_t reps, double* ms_per_rep, uint64_t* hash_out)
{
  if (n <= 0 || !problems || n_threads <= 0 || reps <= 0 || !ms_per_rep) return PTZ_EINVAL;
  std::vector<size_t> ob(n + 1, 0), rb(n + 1, 0), cb(n + 1, 0);
  for (int i = 0; i < n; ++i) { ob[i + 1] = ob[i] + (size_t)problems[i].n_obs; rb[i + 1] = rb[i] + problems[i].n_ray; cb[i + 1] = cb[i] + problems[i].n_cam; }
  const auto t0 = std::chrono::steady_clock::now();
  for (int rep = 0; rep < reps; ++rep) {
    RawVec<float2> uv(ob[n]), camuv(ob[n]);
    RawVec<int> cam(ob[n]), ray(ob[n]), camobs(ob[n]), camray(ob[n]), wpos(ob[n]);
    std::vector<int> rayptr(rb[n] + n), camptr(cb[n] + n), campair(cb[n] + n), camrun(cb[n] + n), perm(rb[n]);
    std::vector<double> w(rb[n]);
    std::vector<PairBuild> wave(n);
    auto work = [&](int t, int step) {
      for (int k = t; k < n; k += step) {
        const ObsDest od = {uv.data() + ob[k], cam.data() + ob[k], ray.data() + ob[k], camobs.data() + ob[k], camray.data() + ob[k], camuv.data() + ob[k],
                            rayptr.data() + rb[k] + k, w.data() + rb[k], camptr.data() + cb[k] + k, campair.data() + cb[k] + k,
                            camrun.data() + cb[k] + k, wpos.data() + ob[k]};
        build_pairs(problems[k], (int)ob[k], (int)rb[k], od, wave[k], perm.data() + rb[k], schur_threads_of(problems[k].factor_type));
      }
    };
    const int nt = std::min(n_threads, n);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t, nt);
    work(0, nt);
    for (auto& x : th) x.join();
    for (int k = 0; k < n; ++k) if (wave[k].err != PTZ_OK) return wave[k].err;
    if (hash_out && rep == 0) {  // FNV-1a over everything the stage produces, in a fixed order
      uint64_t h = 1469598103934665603ull;
      auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* c = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
      mix(uv.data(), sizeof(float2) * ob[n]); mix(camuv.data(), sizeof(float2) * ob[n]); mix(cam.data(), 4 * ob[n]); mix(ray.data(), 4 * ob[n]);
      mix(camobs.data(), 4 * ob[n]); mix(camray.data(), 4 * ob[n]); mix(wpos.data(), 4 * ob[n]);
      mix(rayptr.data(), 4 * rayptr.size()); mix(camptr.data(), 4 * camptr.size()); mix(campair.data(), 4 * campair.size()); mix(camrun.data(), 4 * camrun.size());
      mix(perm.data(), 4 * perm.size()); mix(w.data(), 8 * w.size());
      for (int k = 0; k < n; ++k) {
        const PairBuild& pb = wave[k];
        mix(pb.pci.data(), 4 * pb.pci.size()); mix(pb.pcj.data(), 4 * pb.pcj.size()); mix(pb.pptr.data(), 4 * pb.pptr.size()); mix(pb.pbrow.data(), 4 * pb.pbrow.size());
        mix(pb.ent.data(), 4 * pb.ent.size()); mix(pb.runs.data(), 8 * pb.runs.size()); mix(pb.prun.data(), 4 * pb.prun.size());
        const int64_t v[7] = {pb.n_ent, pb.n_pair, pb.max_cam_obs, pb.max_cam_ent, pb.max_cam_pair, pb.max_cam_run, pb.err};
        mix(v, sizeof(v));
      }
      *hash_out = h;
    }
  }
  *ms_per_rep = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  return PTZ_OK;
}

}  // extern "C"
