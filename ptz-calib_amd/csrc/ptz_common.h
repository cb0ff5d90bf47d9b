// ptz_common.h -- shared host/device helpers for the HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <tuple>
#include <utility>

#include "../../include/ptz_calib_amd.h"

namespace ptz {

#define PTZ_HIP_TRY(expr)                                                                      \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      fprintf(stderr, "[ptzcalib] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e), __FILE__, \
              __LINE__, #expr);                                                                \
      return (_e == hipErrorOutOfMemory) ? PTZ_ENOMEM : PTZ_ENODEVICE;                         \
    }                                                                                          \
  } while (0)

constexpr int WAVE = 64;

// Entry points check hipGetLastError() after their own launches, so an error some other user of the runtime left behind on this
// thread (PyTorch, an earlier failed call) has to go first -- but not silently: it is reported once per occurrence.
inline void clear_stale_error(const char* where)
{
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    fprintf(stderr, "[ptzcalib] %s: cleared a HIP error that an earlier call on this thread left behind (%s); it is not this call's\n", where, hipGetErrorName(e));
}

// Every C-ABI entry point runs on the device it was given and leaves the calling thread's current device as it found it
// (a PyTorch caller's later launches must not land on another GPU because a batch was destroyed at GC time).
struct DeviceGuard {
  int prev = -1;
  bool changed = false, ok = true;
  explicit DeviceGuard(int dev)
  {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; changed = ok; }
  }
  ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define PTZ_DEVICE_GUARD(dev)              \
  ptz::DeviceGuard _ptz_guard(dev);        \
  if (!_ptz_guard.ok) return PTZ_ENODEVICE

// Wait for a stream by POLLING it.  hipStreamSynchronize parks the thread on an interrupt-driven signal wait, and with several
// host threads driving their own streams (the lock-step PTZ-IBA's four batch threads, the sharded dealers) those waits were
// measured to return 20-40 ms after the work had finished (rocprofv3 --hip-trace: hipStreamSynchronize max 40.7 ms around
// solves of 1.5 ms, several threads released at the same instant).  hipStreamQuery reads the completion signal directly.
// The first 200 us spin (the common case: a copy or a small solve), after that the thread yields between polls.
inline hipError_t stream_wait(hipStream_t st)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if ((spins & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) std::this_thread::yield();
    else __builtin_ia32_pause();
  }
}

// ---- kernel launches that can be recorded into a hipGraph instead of being issued -----------------------------------------
// One LM pass is the same ~25 launches with the same arguments every time, so it is built ONCE as a graph of kernel nodes
// (a linear chain) and replayed per pass: one host call instead of 25.  The graph is built explicitly, node by node, not by
// stream capture: capture puts a stream into a state that other host threads' synchronous runtime calls (another shard's
// batch creation on the same device) can stall on, which deadlocked ptz_ba_solve_sharded.
struct GraphRecorder {
  hipGraph_t graph = nullptr;
  hipGraphNode_t last = nullptr;
  bool ok = true;
};
inline thread_local GraphRecorder* g_recorder = nullptr;  // set while a pass is being recorded on this thread

template <typename... KArgs, typename... Args>
inline void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t st, Args&&... args)
{
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
  if (!g_recorder) {
    hipLaunchKernelGGL(kern, grid, block, smem, st, std::forward<Args>(args)...);
    return;
  }
  std::tuple<KArgs...> vals(static_cast<KArgs>(args)...);  // the kernel's own parameter types; copied by the runtime at node creation
  void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
  {
    size_t i = 0;
    std::apply([&](auto&... v) { ((ptrs[i++] = (void*)&v), ...); }, vals);
  }
  hipKernelNodeParams p;
  memset(&p, 0, sizeof(p));
  p.func = (void*)kern;
  p.gridDim = grid;
  p.blockDim = block;
  p.sharedMemBytes = (unsigned)smem;
  p.kernelParams = ptrs;
  p.extra = nullptr;
  hipGraphNode_t node = nullptr;
  GraphRecorder& r = *g_recorder;
  if (hipGraphAddKernelNode(&node, r.graph, r.last ? &r.last : nullptr, r.last ? 1 : 0, &p) != hipSuccess) { r.ok = false; (void)hipGetLastError(); }
  else r.last = node;
}

// ---- XCD-aware block remap -------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs by linear block id.  For 2-D grids (x = item, y = scene) this
// bijective remap gives each XCD a contiguous range of the logical (scene, item) space, so the workgroups that
// share a scene's data share one XCD's L2.  Placement is a speed matter only; results never depend on it.
__device__ __forceinline__ void xcd_remap(int& bx, int& by)
{
  const unsigned gx = gridDim.x, total = gridDim.x * gridDim.y;
  const unsigned L = blockIdx.x + gx * blockIdx.y;
  const unsigned q = total / 8, r = total % 8, xcd = L % 8, idx = L / 8;
  const unsigned Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bx = (int)(Lp % gx);
  by = (int)(Lp / gx);
}

// ---- deterministic reductions ----------------------------------------------------------------------
// ---- cross-lane exchanges without the LDS crossbar (gfx950) ------------------------------------------------------------------
// v_permlane32_swap / v_permlane16_swap exchange the upper half (the odd 16-lane rows) of one register with the lower half (the even
// rows) of another: exactly the trade of a reduce-scatter's halving step at lane distance 32 / 16 -- after the swap a + b is, in the
// lower lanes, a(l) + a(l + d) and, in the upper lanes, b(l - d) + b(l): the sums the shuffle form (keep + shfl_xor(send)) makes, same
// operands, same bits.  Distances 8, 4, 2, 1 are DPP moves inside a row of 16 lanes.
__device__ __forceinline__ void swap_halves32(double& a, double& b)
{
  auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap_rows16(double& a, double& b)
{
  auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// the value of lane (l ^ D) for D = 8, 4, 2, 1 (all lanes active)
template <int D> __device__ __forceinline__ double lane_xor_dpp(double v)
{
  static_assert(D == 8 || D == 4 || D == 2 || D == 1, "inside a row of 16 lanes");
  if (D == 8) return dpp_mov<0x128>(v);       // row_ror:8
  if (D == 2) return dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  if (D == 1) return dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  // D == 4: lanes of the even banks of four read four lanes up (row_shl:4), those of the odd banks four lanes down (row_shr:4)
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x104, 0xf, 0x5, false);
  lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x114, 0xf, 0xa, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x104, 0xf, 0x5, false);
  hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x114, 0xf, 0xa, false);
  return __hiloint2double(hi, lo);
}

// Butterfly over the 64 lanes of a wave: every lane ends with the same, order-fixed sum -- lane l adds the value of lane l ^ 32, then
// l ^ 16, .. l ^ 1.  Round 5: the exchanges are register swaps (distance 32, 16) and DPP moves (8 .. 1) instead of twelve
// ds_bpermute_b32 through the LDS crossbar; the tree and therefore the bits are the shuffle form's.  ALL 64 lanes must be active.
__device__ __forceinline__ double wave_sum(double v)
{
  { double t = v; swap_halves32(v, t); v += t; }   // v = [lo, lo], t = [hi, hi]
  { double t = v; swap_rows16(v, t); v += t; }
  v += lane_xor_dpp<8>(v);
  v += lane_xor_dpp<4>(v);
  v += lane_xor_dpp<2>(v);
  v += lane_xor_dpp<1>(v);
  return v;
}
__device__ __forceinline__ double wave_max(double v)  // (the same butterfly as wave_sum: no trip through the LDS crossbar)
{
  { double t = v; swap_halves32(v, t); v = fmax(v, t); }
  { double t = v; swap_rows16(v, t); v = fmax(v, t); }
  v = fmax(v, lane_xor_dpp<8>(v));
  v = fmax(v, lane_xor_dpp<4>(v));
  v = fmax(v, lane_xor_dpp<2>(v));
  v = fmax(v, lane_xor_dpp<1>(v));
  return v;
}

// Block-wide sum (blockDim.x multiple of 64, <= 1024).  scratch: >= 16 doubles of LDS.  All threads
// receive the result; waves are added in wave order, so the result is bitwise reproducible.
__device__ __forceinline__ double block_sum(double v, double* scratch)
{
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}
__device__ __forceinline__ double block_max(double v, double* scratch)
{
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = scratch[0];
  for (int i = 1; i < nw; ++i) t = fmax(t, scratch[i]);
  return t;
}

// Four reductions in one pass of the block tree (kind bit k set: value k is a maximum, else a sum).  Value by value the
// same operations in the same order as block_sum / block_max; scratch: [4 * waves].
__device__ __forceinline__ void block_reduce4(double (&v)[4], unsigned max_mask, double* scratch)
{
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = ((max_mask >> k) & 1u) ? wave_max(v[k]) : wave_sum(v[k]);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) scratch[4 * w + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if ((max_mask >> k) & 1u) {
      double t = scratch[k];
      for (int i = 1; i < nw; ++i) t = fmax(t, scratch[4 * i + k]);
      v[k] = t;
    }
    else {
      double t = 0;
      for (int i = 0; i < nw; ++i) t += scratch[4 * i + k];
      v[k] = t;
    }
  }
}

// ---- dense Cholesky (ptz_chol.hip) -------------------------------------------------------------------
// `count` independent SPD systems stored as padded row-major matrices A[count][np][np] (lower triangle
// read).  Row n_i of system i holds rhs^T in columns [0, n_i) and a huge diagonal (CHOL_BIG), so the
// factorisation performs the forward substitution; rows > n_i are identity.  After chol_factor the
// strictly-lower tiles of A hold L, Ldiag holds the factored diagonal tiles, and chol_backsolve writes
// x[count][np].
constexpr int CHOL_NB = 64;
constexpr double CHOL_BIG = 1e300;
constexpr int CHOL_STEP_COLS = 4;  // block columns one step of the schedule can hold
struct CholBatch {
  int count = 0;
  int np = 0;             // padded order, multiple of CHOL_NB, >= max(n_i) + 1
  double* A = nullptr;    // device
  // Where the off-diagonal tiles of L live once they are final.  nullptr: in place in A (the multi-launch paths).  The
  // one-launch-per-column path (chol_col_step_kernel) cannot publish L_ik in place -- other workgroups of the same launch
  // still read A_ik -- and writes a second matrix [count][np][np]; the back-substitution reads L from there.
  double* L = nullptr;
  // device [count][np/NB][NB*NB] or nullptr: full inverses of the factored diagonal tiles (lower triangular, row-major).  Written
  // by the one-launch-per-column path beside the critical chain (a spare workgroup of the next launch); with them the
  // back-substitution is two matrix-vector products per block column instead of a blocked triangular solve.
  double* Linv = nullptr;
  // Compacted launches (thin LM passes of a large batch): blockIdx.y is a SLOT, act[slot] the system it stands for, *act_n the
  // number of slots in use.  nullptr: slot == system.  The second matrix L of the one-launch-per-column path is indexed by slot.
  const int* act = nullptr;
  const int* act_n = nullptr;
  double* Ldiag = nullptr;  // device [count][np/NB][NB*NB]
  double* Dinv = nullptr;   // device [count][np/NB][4][16*16]: inverses of the 16x16 diagonal blocks of L_kk
  const int* n = nullptr;   // device [count]
  int* fail = nullptr;      // device [count]: bit 0 set if a pivot <= 0 is met in rows < n_i, bit 1 if a hand-over of the one-launch
                            // factorisation did not arrive within its bounded wait (chol_chain_kernel)
  const int* active = nullptr;  // device [count] or nullptr; systems with active == 0 are skipped
  // device [count][nt * nt] or nullptr (dense): tmask[i * nt + j] != 0  <=>  tile (i, j), i >= j, of L can be non-zero
  // (structure of the reduced camera system closed under the fill of a tile-level symbolic factorisation; the tile row
  // holding the rhs row is full).  Tiles outside the mask stay exactly zero and their panel / update work is skipped:
  // the tile-granular counterpart of the sparse Cholesky behind the reference's SPARSE_SCHUR (ptzray_optimizer.cc:471).
  const unsigned char* tmask = nullptr;
  // Step schedule of the one-launch-per-step path: device [count][nt][CHOL_STEP_COLS] block columns (or -1, ascending, slot 0
  // always used) that step s factors side by side -- columns of one step never couple (the arcs of a dissected ring and their
  // halves), so their panels, updates and the diagonal tiles of the next step proceed in ONE launch and the dependent chain is
  // as long as the longest arc piece plus the separators.  nullptr: step s = block column s.  n_steps: steps of the longest
  // schedule in the batch.
  const int* sched = nullptr;
  int n_steps = 0;
  const int* sched_kmin = nullptr;  // HOST memory [n_steps]: smallest block column any system factors in step s (sizes the launch)
  // device [chol_chain_ctl_ints(np)] or nullptr: tickets, generation and per-slot tile flags of chol_chain_kernel -- the
  // factorisation of a few systems as ONE launch whose workgroups hand their tiles on through flags (zeroed once at creation;
  // one block per stream that factors)
  int* chain_ctl = nullptr;
  int chain_spin_limit = 0;  // polls a chain hand-over waits for before it gives up (0: the default, 2^21; tests shorten it)
  int chain_w0 = 1;          // the chain kernels' diagonal tile with one wave on the chain and no barrier on it (diag_factor_tile_w0; PTZ_BA_CHAIN_W0=0: the four-wave form)
  int chain_pair = 1;        // chol_chain_kernel: the last two columns of a diagonal tile side by side when their producers end together (PTZ_BA_CHAIN_PAIR=0: in turn)
  int chain_ready_whole = 1; // chol_chain_kernel: a column found finished is applied in one go (PTZ_BA_CHAIN_READY_WHOLE=0: block rounds, A/B)
  // device [count][4 * chol_backsolve_max_groups(np)] work items of the back-substitution in execution order and [count] their
  // group counts (chol_backsolve_plan, made on the host with the structure), or nullptr: the kernel makes the list itself
  const struct BsItem* bs_items = nullptr;
  const int* bs_groups = nullptr;
  // 1: the host's list carries owners (BsItem::kind's bits 4 and 5, chol_backsolve_plan) for some system of the batch -- the
  // back-substitution then runs TWO workgroups per system, see chol_backsolve_arcs
  int bs_split = 0;
  // device [count][np / NB] or nullptr: elimination order of the tiles (the bundle adjustment's Dev::tperm).  With it the
  // back-substitution writes x in the CALLER's numbering -- x[c] = solution of row tperm[c / NB] * NB + c % NB -- so that the
  // reader of the solution (the camera update, on the critical path of a pass) needs no lookup in front of its loads
  const int* xperm = nullptr;
};
// one 64 x 64 tile product of the back-substitution: kind 0 empty slot, 1 x = M^T t (inverse of a diagonal tile), 2 t -= M^T x
struct BsItem { long long off; int ld, in_off, out_off, kind; };
inline int chol_backsolve_max_groups(int np)
{
  const int nt = np / CHOL_NB;
  int g = nt;  // one group per diagonal inverse + the tiles of row k four at a time
  for (int k = 0; k < nt; ++k) g += (k + 3) / 4;
  return g;
}
// The two arcs of a dissected system (ptz_ba.hip's elimination plan) do not couple: below the separator tiles -- the single-column
// steps at the end of the schedule -- the tiles fall into two connected components of the structure, and the back-substitution of
// one never reads what the other's writes.  One compute unit takes the factor in at 64 bytes per clock, which is what the
// back-substitution of one rig is bound by, so the list is dealt to TWO workgroups: both walk the separator tiles' chain (its 15 of 76
// tiles twice, but no word passes between them -- each finds the same x_k, bit for bit), each then only the tiles whose results land
// in ITS arc.  owner[t]: 0 separator / padding (both), 1 / 2 the arcs; false: no such split (natural order, one arc, more components).
inline bool chol_backsolve_arcs(int nt, int n, const unsigned char* tm, const int* sched, int n_steps_sched, unsigned char* owner)
{
  const int NB = CHOL_NB;
  for (int t = 0; t < nt; ++t) owner[t] = 0;
  if (!sched || !tm || nt > 64) return false;
  int first_top_step = n_steps_sched;  // steps [first_top_step, n_steps) have one column each
  for (int st = n_steps_sched - 1; st >= 0; --st) {
    int cols = 0;
    for (int c = 0; c < CHOL_STEP_COLS; ++c) cols += sched[st * CHOL_STEP_COLS + c] >= 0 && sched[st * CHOL_STEP_COLS + c] * NB < n;
    if (cols > 1) break;
    first_top_step = st;
  }
  if (first_top_step <= 0) return false;
  int comp[64];
  for (int t = 0; t < nt; ++t) comp[t] = -2;  // -2: separator / padding, -1: arc tile not yet labelled
  for (int st = 0; st < first_top_step; ++st)
    for (int c = 0; c < CHOL_STEP_COLS; ++c) { const int k = sched[st * CHOL_STEP_COLS + c]; if (k >= 0 && k * NB < n) comp[k] = -1; }
  int n_comp = 0;
  for (int t0 = 0; t0 < nt; ++t0) {
    if (comp[t0] != -1) continue;
    int stack[64], sp = 0;
    stack[sp++] = t0; comp[t0] = n_comp;
    while (sp) {
      const int a = stack[--sp];
      for (int b = 0; b < nt; ++b) {
        if (comp[b] != -1) continue;
        const bool coupled = a > b ? tm[a * nt + b] != 0 : tm[b * nt + a] != 0;
        if (coupled) { comp[b] = n_comp; stack[sp++] = b; }
      }
    }
    ++n_comp;
  }
  if (n_comp != 2) return false;
  for (int t = 0; t < nt; ++t) owner[t] = comp[t] < 0 ? 0 : (unsigned char)(comp[t] + 1);
  return true;
}
// The list chol_backsolve_kernel would make for itself (same items, same order), made once on the host: backward over the steps of
// the factorisation, the diagonal inverses of a step in one group, the tiles of its rows four to a group.  Returns the groups.
// split (may be null): set to whether the items carry owners -- kind | owner << 4, owner = chol_backsolve_arcs of the tile the item
// WRITES (0: both workgroups) -- which they do when the system has two arcs and `want_split`.
inline int chol_backsolve_plan(int np, int n, const unsigned char* tm, const int* sched, int n_steps_sched, BsItem* items, bool want_split = false,
                               bool* split = nullptr)
{
  const int nt = np / CHOL_NB, NB = CHOL_NB;
  const int n_steps = sched ? n_steps_sched : nt;
  int g = 0;
  for (int st = n_steps - 1; st >= 0; --st) {
    int col[CHOL_STEP_COLS];
    for (int c = 0; c < CHOL_STEP_COLS; ++c) col[c] = -1;
    if (st < nt) {
      if (!sched) col[0] = st;
      else for (int c = 0; c < CHOL_STEP_COLS; ++c) col[c] = sched[st * CHOL_STEP_COLS + c];
    }
    bool any_col = false;
    for (int c = 0; c < CHOL_STEP_COLS; ++c) {
      if (col[c] >= 0 && col[c] * NB >= n) col[c] = -1;
      any_col |= col[c] >= 0;
    }
    if (!any_col) continue;
    for (int c = 0; c < 4; ++c) {
      const int kd = c < CHOL_STEP_COLS ? col[c] : -1;
      items[4 * g + c] = kd >= 0 ? BsItem{(long long)kd * (NB * NB), NB, kd * NB, kd * NB, 1} : BsItem{0, 0, 0, 0, 0};
    }
    ++g;
    int filled = 0;
    for (int c = 0; c < CHOL_STEP_COLS; ++c) {
      const int k = col[c];
      if (k < 0) continue;
      for (int tj = 0; tj < k; ++tj)
        if (!tm || tm[k * nt + tj]) items[4 * g + filled++] = BsItem{(long long)(k * NB) * np + (long long)tj * NB, np, k * NB, tj * NB, 2};
    }
    const int padded = (filled + 3) & ~3;
    for (int i = filled; i < padded; ++i) items[4 * g + i] = BsItem{0, 0, 0, 0, 0};
    g += padded / 4;
  }
  if (split) *split = false;
  unsigned char owner[64];
  if (want_split && chol_backsolve_arcs(nt, n, tm, sched, n_steps_sched, owner)) {
    for (int i = 0; i < 4 * g; ++i)
      if (items[i].kind != 0) items[i].kind |= (int)owner[items[i].out_off / NB] << 4;
    if (split) *split = true;
  }
  return g;
}
// ints behind CholBatch::chain_ctl: [0] next ticket, [1] workgroups done, [2] generation of the last finished launch, then for
// each of up to CHOL_CHAIN_SLOTS slots 4 nt "block b of diagonal tile k factored" flags and nt * nt "tile (i, k) final" flags (value = generation)
// ("diagonal tile k factored" is FOUR flags, one per 16-column block of the tile: consumers take the blocks as they come)
constexpr int CHOL_CHAIN_SLOTS = 32;  // most systems one chol_chain_kernel launch factors (slots of finished L tiles and flags per stream)
inline size_t chol_chain_ctl_ints(int np) { const size_t nt = (size_t)np / CHOL_NB; return 4 + (size_t)CHOL_CHAIN_SLOTS * (4 * nt + nt * nt); }
// whether `count` systems of order np are factored by ONE launch (chol_chain_kernel) -- ptz_chol.hip
bool chol_chain_fits(int count, int np);
__device__ __forceinline__ int chol_system_of(const CholBatch& cb, int slot)
{
  if (!cb.act) return slot;
  return slot < *cb.act_n ? cb.act[slot] : -1;
}
inline int chol_padded_order(int n_max) { return ((n_max + 1 + CHOL_NB - 1) / CHOL_NB) * CHOL_NB; }
// enqueue factorisation + solve on `stream`; x: device [count][np]
void chol_factor_solve(const CholBatch& cb, double* x, hipStream_t stream);
// the three kernel families of chol_factor_solve, individually (for per-family timing)
void chol_panel_launch(const CholBatch& cb, int k, hipStream_t stream, bool diag_done = false);  // diag (unless already done) + trsm
void chol_syrk_launch(const CholBatch& cb, int k, hipStream_t stream, int mode, bool fuse_diag = false);  // mode 0 all, 1 column k+1, 2 rest; fuse_diag: also factor tile (k+1, k+1)
void chol_backsolve_launch(const CholBatch& cb, double* x, hipStream_t stream);
void chol_tile_inverse_launch(const CholBatch& cb, hipStream_t stream);  // Linv of every diagonal tile (multi-launch paths, before the back-substitution)
void chol_diag_launch(const CholBatch& cb, int k, hipStream_t stream);
bool chol_chain_enabled(const CholBatch& cb);                               // the one-launch factorisation applies to this batch
void chol_chain_launch(const CholBatch& cb, hipStream_t stream);            // diagonal tiles, steps and tile inverses of a few systems in ONE launch       // factor the diagonal tile of block column k (k < 0: of the columns of step 0)
inline int chol_step_count(const CholBatch& cb) { return cb.sched ? cb.n_steps : cb.np / CHOL_NB; }
void chol_col_step_launch(const CholBatch& cb, int k, hipStream_t stream);   // few systems: trsm + trailing update + next diagonal tile, one launch
void chol_update_col_launch(const CholBatch& cb, int j, hipStream_t stream, bool fuse_diag = false);  // left-looking: column j -= all earlier columns; fuse_diag: also factor tile (j, j)
// helper kernel launcher: zero A, set padding identity / CHOL_BIG (rows >= n_i) for all systems
void chol_clear(const CholBatch& cb, hipStream_t stream);
#ifdef PTZ_CHOL_TIMELINE
void chol_chain_timeline_print(int nt);  // probe builds: the tiles' times of the LAST chol_chain_kernel launch (system 0), printed by a kernel of its own
#endif

}  // namespace ptz
