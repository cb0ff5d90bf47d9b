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
#include <vector>

#include "ptz_common.h"
#include "ptz_pool.h"
#include "ptz_factor.h"

namespace ptz {

namespace {

constexpr int RAY_BLOCK = 1024;  // rays per workgroup in the ray-centric kernels
constexpr int CBS = CAMBLK + 1;    // LDS stride of a camera block (33 doubles: odd -> no same-field bank conflicts)
constexpr int CDS = CANDBLK + 1;   // LDS stride of a candidate block (19)
// W row stride: see Dims<TYPE>::WS

struct SceneDev {
  int n_cam, n_ray, n_obs, n_pair;
  int cam_off, ray_off, obs_off, pair_off;
  int ent_off;   // first camera-pair entry
  int part_off;  // first partial-sum slot (one per ray chunk)
  int n_chunk;   // ceil(n_ray / RAY_BLOCK)
  int n;         // NC * n_cam: order of the reduced camera system
  int idx;       // global scene index (the CSR pointer arrays carry one extra entry per preceding scene)
  int o3_off, n_o3;  // 2D-3D annotation observations of the scene
  int grp_off, n_grp;  // shared-intrinsics groups with >= 2 cameras (ranges into grp_ptr, which carries n_grp + 1 entries per scene)
};

struct LmState {
  double radius, decrease_factor;
  double x_cost, x_norm, grad_max;
  double candidate_cost, model_cost_change;
  double initial_cost, final_cost, it_cost;
  int reuse_diagonal, need_linearize, cur, iteration, n_summaries, step_is_successful;
  int num_consecutive_invalid, termination;
  int num_successful, num_unsuccessful, num_lm_steps, num_linear_solves, num_jac_evals;
  int pad;
};

struct Opt {  // device copy of the solver options
  int max_num_iterations, max_consecutive_invalid, jacobi_scaling;
  double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
};

// Everything the kernels read, by value.
struct Dev {
  int n_scene;
  const SceneDev* scene;
  // observations (sorted ray-major) and structure
  const float2* obs_uv;
  const int* obs_cam;   // scene-local camera id
  const int* obs_ray;   // scene-local ray id
  const int* ray_ptr;   // [total_ray + n_scene] per scene n_ray + 1 entries, global obs index
  const int* cam_ptr;   // [total_cam + n_scene] per scene n_cam + 1 entries into cam_obs
  const int* cam_obs;   // global obs index, camera-major
  const int* wpos;      // [total_obs] row of W that holds observation a (camera-major position)
  const int* cam_ray;   // [total_obs] global ray id, camera-major (same order as cam_obs)
  const int* pair_ci;   // scene-local camera ids, ci >= cj
  const int* pair_cj;
  const int* pair_ptr;  // [total_pair + n_scene] per scene n_pair + 1 entries, global entry index
  const int* cam_pair;  // [total_cam + n_scene] per scene n_cam + 1 entries: scene-local pair range of each camera ci
  const int2* ent;      // (position of obs a in ci's observation list, global obs index b of cj); ci > cj only
  const double* ray_w;
  // state: two buffers, LmState.cur selects the current one
  double* cam_x;  // [2][total_cam][15]
  double* ray_x;  // [2][total_ray][3]
  size_t cam_stride, ray_stride;
  const double* cam_x0;
  const double* ray_x0;
  // per-camera blocks
  double* camblk;    // [total_cam][CAMBLK]   at x
  double* candblk;   // [total_cam][CANDBLK]  at the candidate
  double* scale_c;   // [total_cam][NC]
  double* scale_r;   // [total_ray][3]
  double* U;         // [total_cam][NC*NC]
  double* gc;        // [total_cam][NC]
  double* costc;     // [total_cam]
  double* diag_c;    // [total_cam][NC]
  double* dc;        // [total_cam][NC] scaled-space camera step
  double* V;         // [total_ray][6]
  double* gr;        // [total_ray][3]
  double* diag_r;    // [total_ray][3]
  double* E;         // [total_ray][6]
  double* z;         // [total_ray][3]
  double* W;         // [total_obs][Dims::WS] rows W_a = Jc^T Jr (NW x 3), camera-major
  double* partial;   // [total_chunk + n_scene][2] (one extra slot per scene for the 2D-3D terms)
  // 2D-3D annotation residuals (georeferencing); per-scene arrays below are indexed by the GLOBAL scene index
  const float2* o3_uv;  // [total_o3]
  const double* o3_xyz; // [total_o3][3] world points
  const int* o3_cam;    // [total_o3] scene-local camera id
  double* tlw_x;     // [2][n_scene_total][6]
  size_t tlw_stride;
  double* tlwblk;    // [n_scene_total][TLWBLK] at x
  double* tlwcand;   // [n_scene_total][TLWBLK] at the candidate (R_lw and t used)
  double* scale_t;   // [n_scene_total][6]
  double* diag_t;    // [n_scene_total][6]
  double* Ut;        // [n_scene_total][36]
  double* gt;        // [n_scene_total][6]
  double* dt;        // [n_scene_total][6] scaled-space tlw step
  double* Jc3;       // [total_o3][2][NC] (scaled)
  double* Jt3;       // [total_o3][2][6]  (scaled)
  double* r3;        // [total_o3][2]
  // shared intrinsics (SetSharedIntrinsics): groups of cameras whose intrinsic columns are one parameter
  int shared;            // 0 = no group anywhere in the batch: none of the group kernels is launched
  const int* grp_ptr;    // per scene n_grp + 1 offsets into grp_mem (global), at [scene.grp_off + scene.idx ...]
  const int* grp_mem;    // scene-local camera ids of a group, ascending; the LAST one is the representative
  const unsigned char* cam_flag;  // [total_cam] bit 0: this camera's intrinsics block is counted in |x| (one per group)
  double* gfold;         // [total_cam][NC] gradient with the shared slots folded onto the representative
  // LM
  LmState* lm;
  int* active;       // [n_scene]
  int* ray_fail;     // [n_scene]
  Opt opt;
  // reduced camera system
  CholBatch chol;
  double* yc;        // [n_scene][np]
};

__device__ __forceinline__ const double* cur_cam(const Dev& d, const SceneDev& s, const LmState& st)
{
  return d.cam_x + (size_t)st.cur * d.cam_stride + (size_t)s.cam_off * 15;
}
__device__ __forceinline__ const double* cur_ray(const Dev& d, const SceneDev& s, const LmState& st)
{
  return d.ray_x + (size_t)st.cur * d.ray_stride + (size_t)s.ray_off * 3;
}

// TYPE = factor (0 PTZRay, 1 PTZRayDist, 2 PTZRayFxfyDist) + 3 * has3d.
//   NW columns of a camera carry a non-zero 2D-2D Jacobian: [f, (k1), r1, r2, r3]; PTZRayFxfyDist [fx, fy, k1, r1, r2, r3]
//   NC free camera parameters: without annotations the same set (the reference's always-zero fy column of PTZRay /
//   PTZRayDist is not materialised); with 2D-3D annotation residuals fy becomes live (Reproj2d3dFactor reads it,
//   ptzray_optimizer.cc:273): [f, fy, (k1), r1, r2, r3], and the 6-dof T_l_w block joins the reduced system.
template <int TYPE> struct Dims {
  static constexpr int FACTOR = TYPE % 3, HAS3D = TYPE / 3;
  static constexpr int FXFY = FACTOR == 2;  // fy is a 2D-2D column
  static constexpr int F3 = FACTOR ? 1 : 0;  // Reproj2d3dFactor variant: k1 free or not (same functor for Dist / FxfyDist)
  static constexpr int NW = 4 + (FACTOR >= 1) + FXFY;
  static constexpr int NC = NW + (HAS3D && !FXFY);
  static constexpr int NG = 6 * HAS3D;  // size of the global (tlw) block
  // doubles per observation row of W = Jc^T Jr (NW x 3), rounded up to a 16-byte multiple: 12 (96 B) / 16 (128 B).
  // Measured on MI355X: unpadded 96-B rows beat 128-B-aligned rows (less write/stream traffic outweighs line straddling).
  static constexpr int WS = (NW * 3 + 1) & ~1;
  // position of 2D-2D column k inside the NC block
  static __host__ __device__ constexpr int pos(int k) { return NC != NW ? (k == 0 ? 0 : k + 1) : k; }
  // index of free parameter k of the NC block in the Camera 15-vector
  static __host__ __device__ constexpr int at(int k)
  {
    // PTZRay: f r1 r2 r3 | PTZRayDist: f k1 r.. | +3D: f fy r.. | f fy k1 r.. | PTZRayFxfyDist (with or without 3D): fx fy k1 r..
    constexpr int FYL = (NC != NW) || FXFY;  // fy occupies slot 1
    return k == 0 ? 0 : (FYL && k == 1) ? 1 : (FACTOR && k == 1 + FYL) ? 10 : 4 + (k - (NC - 3));
  }
};

__device__ __forceinline__ void fill_camblk(const double* c15, double* cb, bool with_jl)
{
  double R[9];
  rodrigues(c15 + 4, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) cb[CB_R + i] = R[i];
  cb[CB_F] = c15[0]; cb[CB_CX] = c15[2]; cb[CB_CY] = c15[3]; cb[CB_FY] = c15[1];
#pragma unroll
  for (int i = 0; i < 5; ++i) cb[CB_K + i] = c15[10 + i];
  if (with_jl) {
    double Jl[9];
    so3_left_jacobian(c15 + 4, Jl);
#pragma unroll
    for (int i = 0; i < 9; ++i) cb[CB_JL + i] = Jl[i];
  }
}

// ---- cam_prep: rotation / SO(3) Jacobian / intrinsics / scales of every camera at x ------------------
template <int TYPE>
__global__ void k_cam_prep(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cam) return;
  double* cb = d.camblk + (size_t)(s.cam_off + i) * CAMBLK;
  fill_camblk(cur_cam(d, s, st) + (size_t)i * 15, cb, true);
#pragma unroll
  for (int k = 0; k < NC; ++k) cb[CB_S + k] = d.scale_c[(size_t)(s.cam_off + i) * NC + k];
  if (Dims<TYPE>::HAS3D && i == 0) {
    const double* t = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    double* tb = d.tlwblk + (size_t)s.idx * TLWBLK;
    double R[9], Jl[9];
    const double rv[3] = {t[0], t[1], t[2]};
    rodrigues(rv, R);
    so3_left_jacobian(rv, Jl);
    for (int k = 0; k < 9; ++k) { tb[k] = R[k]; tb[9 + k] = Jl[k]; }
    tb[18] = t[3]; tb[19] = t[4]; tb[20] = t[5];
  }
}

// stage a scene's camera table into LDS (stride words per camera)
__device__ __forceinline__ void stage_table(const double* __restrict__ src, double* dst, int count)
{
  for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
}
// same, re-striding rows of SRC doubles to DST doubles in LDS
template <int SRC, int DST>
__device__ __forceinline__ void stage_rows(const double* __restrict__ src, double* dst, int rows)
{
  for (int i = threadIdx.x; i < rows * SRC; i += blockDim.x) dst[(i / SRC) * DST + (i % SRC)] = src[i];
}

// ---- lin_ray: per-ray linearisation ---------------------------------------------------------------------
// thread = ray: for every observation of the ray evaluate residual + Jacobians, apply sqrt(w) and the
// Jacobi scales, accumulate V = sum Jr^T Jr and g_r = sum Jr^T r.  (The W_a = Jc^T Jr rows are written by k_lin_cam, whose
// lanes walk a camera's observations in the order of its W rows: one sequential stream instead of a 96-byte scatter.)
template <int TYPE>
__global__ __launch_bounds__(RAY_BLOCK) void k_lin_ray(Dev d)
{
  constexpr int NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize || blockIdx.x >= s.n_chunk) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  stage_rows<CAMBLK, CBS>(d.camblk + (size_t)s.cam_off * CAMBLK, lds, s.n_cam);
  __syncthreads();
  const int j = blockIdx.x * RAY_BLOCK + threadIdx.x;
  if (j >= s.n_ray) return;
  const int gj = s.ray_off + j;
  const double* X = cur_ray(d, s, st) + (size_t)j * 3;
  const double Xr[3] = {X[0], X[1], X[2]};
  const double sr[3] = {d.scale_r[(size_t)gj * 3], d.scale_r[(size_t)gj * 3 + 1], d.scale_r[(size_t)gj * 3 + 2]};
  const double sw = sqrt(d.ray_w[gj]);
  const int* rp = d.ray_ptr + s.ray_off + s.idx;
  double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int a = rp[j]; a < rp[j + 1]; ++a) {
    const float2 uv = d.obs_uv[a];
    const double* cb = lds + d.obs_cam[a] * CBS;
    double res[2], Jc[2][NW], Jr[2][3];
    ba_linearize<F>(cb, Xr, uv.x, uv.y, res, Jc, Jr);
    res[0] *= sw; res[1] *= sw;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double m = sw * sr[k]; Jr[0][k] *= m; Jr[1][k] *= m; }
    V[0] += Jr[0][0] * Jr[0][0] + Jr[1][0] * Jr[1][0];
    V[1] += Jr[0][1] * Jr[0][0] + Jr[1][1] * Jr[1][0];
    V[2] += Jr[0][1] * Jr[0][1] + Jr[1][1] * Jr[1][1];
    V[3] += Jr[0][2] * Jr[0][0] + Jr[1][2] * Jr[1][0];
    V[4] += Jr[0][2] * Jr[0][1] + Jr[1][2] * Jr[1][1];
    V[5] += Jr[0][2] * Jr[0][2] + Jr[1][2] * Jr[1][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] += Jr[0][k] * res[0] + Jr[1][k] * res[1];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) d.V[(size_t)gj * 6 + k] = V[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) d.gr[(size_t)gj * 3 + k] = g[k];
}

// ---- lin_cam: per-camera blocks -------------------------------------------------------------------------
// wave = camera: lanes stride over the camera's observation list, U = sum Jc^T Jc, g_c = sum Jc^T r,
// cost = 1/2 sum w |r|^2, reduced with a fixed butterfly; every lane also stores the row W_a = Jc^T Jr of its observation
// (row index = position in the camera-major list, so a wave writes one contiguous stretch of W).
template <int TYPE>
__global__ __launch_bounds__(256) void k_lin_cam(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= s.n_cam) return;
  const int lane = threadIdx.x & 63;
  const int gi = s.cam_off + i;
  double cb[CAMBLK];
#pragma unroll
  for (int k = 0; k < CAMBLK; ++k) cb[k] = d.camblk[(size_t)gi * CAMBLK + k];
  const double* rays = cur_ray(d, s, st);
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  double U[NW * (NW + 1) / 2], g[NW], cost = 0;
#pragma unroll
  for (int k = 0; k < NW * (NW + 1) / 2; ++k) U[k] = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) g[k] = 0;
  for (int q = cp[i] + lane; q < cp[i + 1]; q += 64) {
    const int a = d.cam_obs[q];
    const float2 uv = d.obs_uv[a];
    const int j = d.obs_ray[a];
    const double Xr[3] = {rays[(size_t)j * 3], rays[(size_t)j * 3 + 1], rays[(size_t)j * 3 + 2]};
    double res[2], Jc[2][NW], Jr[2][3];
    ba_linearize<F>(cb, Xr, uv.x, uv.y, res, Jc, Jr);
    const double w = d.ray_w[s.ray_off + j];
    const double sw = sqrt(w);
    cost += 0.5 * (w * (res[0] * res[0] + res[1] * res[1]));
    res[0] *= sw; res[1] *= sw;
#pragma unroll
    for (int k = 0; k < NW; ++k) { const double m = sw * cb[CB_S + Dims<TYPE>::pos(k)]; Jc[0][k] *= m; Jc[1][k] *= m; }
    {
      const double* sr = d.scale_r + (size_t)(s.ray_off + j) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double m = sw * sr[k]; Jr[0][k] *= m; Jr[1][k] *= m; }
      double* Wa = d.W + (size_t)q * Dims<TYPE>::WS;
#pragma unroll
      for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) Wa[3 * k + l] = Jc[0][k] * Jr[0][l] + Jc[1][k] * Jr[1][l];
    }
    int e = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      g[k] += Jc[0][k] * res[0] + Jc[1][k] * res[1];
#pragma unroll
      for (int l = 0; l <= k; ++l) U[e++] += Jc[0][k] * Jc[0][l] + Jc[1][k] * Jc[1][l];
    }
  }
  cost = wave_sum(cost);
#pragma unroll
  for (int k = 0; k < NW; ++k) g[k] = wave_sum(g[k]);
#pragma unroll
  for (int k = 0; k < NW * (NW + 1) / 2; ++k) U[k] = wave_sum(U[k]);
  if (lane == 0) {
    d.costc[gi] = cost;
    if (NC != NW) {  // the fy row/column has no 2D-2D contribution; k_lin_3d adds the annotation terms
#pragma unroll
      for (int k = 0; k < NC * NC; ++k) d.U[(size_t)gi * NC * NC + k] = 0;
#pragma unroll
      for (int k = 0; k < NC; ++k) d.gc[(size_t)gi * NC + k] = 0;
    }
    int e = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int pk = Dims<TYPE>::pos(k);
      d.gc[(size_t)gi * NC + pk] = g[k];
#pragma unroll
      for (int l = 0; l <= k; ++l) {
        const int pl = Dims<TYPE>::pos(l);
        d.U[(size_t)gi * NC * NC + pk * NC + pl] = U[e];
        d.U[(size_t)gi * NC * NC + pl * NC + pk] = U[e];
        ++e;
      }
    }
  }
}

// ---- lin_3d: 2D-3D annotation residuals (AddConstraints2d3d, ptzray_optimizer.cc:887-923; weight 1) ----------
// One workgroup per scene: thread = annotation point (closed-form Jacobians of Reproj2d3dFactor w.r.t. the camera
// block [fx, fy, (k1), rvec] and the T_l_w block), then thread 0 adds the few blocks to U_i, g_i, cost_i and builds the
// T_l_w diagonal block / gradient in observation order (fixed order, no atomics).
template <int TYPE>
__global__ __launch_bounds__(256) void k_lin_3d(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = blockIdx.x;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  const double* tb = d.tlwblk + (size_t)s.idx * TLWBLK;
  const double* stl = d.scale_t + (size_t)s.idx * 6;
  for (int o = threadIdx.x; o < s.n_o3; o += 256) {
    const int go = s.o3_off + o;
    const int ci = d.o3_cam[go];
    const double* cb = d.camblk + (size_t)(s.cam_off + ci) * CAMBLK;
    const float2 uv = d.o3_uv[go];
    const double xyz[3] = {d.o3_xyz[(size_t)go * 3], d.o3_xyz[(size_t)go * 3 + 1], d.o3_xyz[(size_t)go * 3 + 2]};
    double res[2], Jc[2][5 + Dims<TYPE>::F3], Jt[2][6];
    reproj2d3d_eval<Dims<TYPE>::F3, true>(cb, tb, xyz, uv.x, uv.y, res, Jc, Jt);
    for (int k = 0; k < NC; ++k) { d.Jc3[(size_t)go * 2 * NC + k] = Jc[0][k] * cb[CB_S + k]; d.Jc3[(size_t)go * 2 * NC + NC + k] = Jc[1][k] * cb[CB_S + k]; }
    for (int k = 0; k < 6; ++k) { d.Jt3[(size_t)go * 12 + k] = Jt[0][k] * stl[k]; d.Jt3[(size_t)go * 12 + 6 + k] = Jt[1][k] * stl[k]; }
    d.r3[(size_t)go * 2] = res[0]; d.r3[(size_t)go * 2 + 1] = res[1];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double Ut[36], gt[6];
  for (int k = 0; k < 36; ++k) Ut[k] = 0;
  for (int k = 0; k < 6; ++k) gt[k] = 0;
  for (int o = 0; o < s.n_o3; ++o) {
    const int go = s.o3_off + o;
    const int gi = s.cam_off + d.o3_cam[go];
    const double* j0 = d.Jc3 + (size_t)go * 2 * NC;
    const double* j1 = j0 + NC;
    const double* q0 = d.Jt3 + (size_t)go * 12;
    const double* q1 = q0 + 6;
    const double r0 = d.r3[(size_t)go * 2], r1 = d.r3[(size_t)go * 2 + 1];
    d.costc[gi] += 0.5 * (r0 * r0 + r1 * r1);
    for (int k = 0; k < NC; ++k) {
      d.gc[(size_t)gi * NC + k] += j0[k] * r0 + j1[k] * r1;
      for (int l = 0; l < NC; ++l) d.U[(size_t)gi * NC * NC + k * NC + l] += j0[k] * j0[l] + j1[k] * j1[l];
    }
    for (int k = 0; k < 6; ++k) {
      gt[k] += q0[k] * r0 + q1[k] * r1;
      for (int l = 0; l < 6; ++l) Ut[k * 6 + l] += q0[k] * q0[l] + q1[k] * q1[l];
    }
  }
  for (int k = 0; k < 36; ++k) d.Ut[(size_t)s.idx * 36 + k] = Ut[k];
  for (int k = 0; k < 6; ++k) d.gt[(size_t)s.idx * 6 + k] = gt[k];
}

// ---- Jacobi scaling (Ceres: s_j = 1 / (1 + |J_:j|), computed once at iteration 0) -----------------------
template <int TYPE>
__global__ void k_jacobi_scale(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < s.n_cam) {
    const int gi = s.cam_off + t;
#pragma unroll
    for (int k = 0; k < NC; ++k) d.scale_c[(size_t)gi * NC + k] = 1.0 / (1.0 + sqrt(d.U[(size_t)gi * NC * NC + k * NC + k]));
  }
  if (t < s.n_ray) {
    const int gj = s.ray_off + t;
    d.scale_r[(size_t)gj * 3 + 0] = 1.0 / (1.0 + sqrt(d.V[(size_t)gj * 6 + 0]));
    d.scale_r[(size_t)gj * 3 + 1] = 1.0 / (1.0 + sqrt(d.V[(size_t)gj * 6 + 2]));
    d.scale_r[(size_t)gj * 3 + 2] = 1.0 / (1.0 + sqrt(d.V[(size_t)gj * 6 + 5]));
  }
  if (Dims<TYPE>::HAS3D && t < 6) d.scale_t[(size_t)s.idx * 6 + t] = 1.0 / (1.0 + sqrt(d.Ut[(size_t)s.idx * 36 + t * 7]));
}

// ---- shared intrinsics (PTZRayOptimizer::SetSharedIntrinsics, ptzray_optimizer.cc:497-505) ------------------------------
// Cameras of a group share ONE intrinsics parameter block.  The per-camera pipeline above stays as it is (every camera
// still carries its own copy of the intrinsic columns, with identical values and steps); what makes the copies one
// parameter is a change of variables x_cam = P x_shared applied where it matters:
//   * column norms, hence Jacobi scales and LM diagonals, are those of the stacked group column (k_group_scale, k_group_diag);
//   * the reduced camera system is folded, S' = P^T S P, b' = P^T b, onto the group's representative (k_fold_system),
//     solved, and the representative's step is copied back to the members (k_group_expand);
//   * gradient norm and |x| count the block once (k_group_grad, cam_flag in k_lm_pre / k_lm_post).
// The representative is the LAST camera of the group, so that the dense rows the fold creates sit at the bottom of the
// system and cause no extra fill above them.
template <int TYPE> __device__ __forceinline__ bool is_intr_slot(int k)
{
  const int a = Dims<TYPE>::at(k);
  return a < 4 || a >= 10;
}

// thread = (group, slot): group-wide Jacobi scale from the summed squared column norms (members ascending)
template <int TYPE>
__global__ void k_group_scale(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = t / NC, k = t % NC;
  if (g >= s.n_grp || !is_intr_slot<TYPE>(k)) return;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  double sum = 0;
  for (int e = gp[g]; e < gp[g + 1]; ++e) sum += d.U[(size_t)(s.cam_off + d.grp_mem[e]) * NC * NC + k * NC + k];
  const double sc_g = 1.0 / (1.0 + sqrt(sum));
  for (int e = gp[g]; e < gp[g + 1]; ++e) d.scale_c[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k] = sc_g;
}

// LM diagonal of a shared slot: clamp(sum of the members' diagonal entries); each member carries an equal share so that
// the fold of S adds them back up
template <int TYPE>
__global__ void k_group_diag(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (d.lm[sc].reuse_diagonal) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = t / NC, k = t % NC;
  if (g >= s.n_grp || !is_intr_slot<TYPE>(k)) return;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  double sum = 0;
  for (int e = gp[g]; e < gp[g + 1]; ++e) sum += d.U[(size_t)(s.cam_off + d.grp_mem[e]) * NC * NC + k * NC + k];
  const double share = fmin(fmax(sum, d.opt.min_lm_diagonal), d.opt.max_lm_diagonal) / (double)(gp[g + 1] - gp[g]);
  for (int e = gp[g]; e < gp[g + 1]; ++e) d.diag_c[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k] = share;
}

// gradient with the shared slots folded onto the representative (others 0); every other slot copied.
// One workgroup per scene: copy, barrier, fold.
template <int TYPE>
__global__ void k_group_grad(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.x;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  for (int t = threadIdx.x; t < s.n_cam * NC; t += blockDim.x) d.gfold[(size_t)s.cam_off * NC + t] = d.gc[(size_t)s.cam_off * NC + t];
  __syncthreads();
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  for (int t = threadIdx.x; t < s.n_grp * NC; t += blockDim.x) {
    const int g = t / NC, k = t % NC;
    if (!is_intr_slot<TYPE>(k)) continue;
    double sum = 0;
    for (int e = gp[g]; e < gp[g + 1]; ++e) sum += d.gc[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k];
    for (int e = gp[g]; e < gp[g + 1]; ++e)
      d.gfold[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k] = (e == gp[g + 1] - 1) ? sum : 0.0;
  }
}

// S' = P^T S P, b' = P^T b in place on the lower-triangular storage (row n = right-hand side), one workgroup per scene,
// one (group, slot) after the other.  For slot index(m) = m * NC + k of the members m:
//   v[c]  = sum_m Sfull[index(m)][c]            for every column c = 0 .. n (c = n is the right-hand side)
//   S'[rep][c] = v[c]  (c not a member slot),   S'[rep][rep] = sum_m v[index(m)],
//   rows / columns of the other members: 0, diagonal 1, right-hand side 0 (their step is copied from the representative).
template <int TYPE>
__global__ __launch_bounds__(1024) void k_fold_system(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (s.n_grp == 0) return;
  const int np = d.chol.np, n = s.n;
  double* A = d.chol.A + (size_t)sc * np * np;
  extern __shared__ double v[];  // [n + 1]
  __shared__ double dsum;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  auto at = [&](int i, int c) -> double& { return i >= c ? A[(size_t)i * np + c] : A[(size_t)c * np + i]; };
  for (int g = 0; g < s.n_grp; ++g) {
    const int e0 = gp[g], e1 = gp[g + 1];
    const int rep = d.grp_mem[e1 - 1];
    for (int k = 0; k < NC; ++k) {
      if (!is_intr_slot<TYPE>(k)) continue;
      for (int c = threadIdx.x; c <= n; c += blockDim.x) {
        double sum = 0;
        for (int e = e0; e < e1; ++e) sum += at(d.grp_mem[e] * NC + k, c);
        v[c] = sum;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        double sum = 0;
        for (int e = e0; e < e1; ++e) sum += v[d.grp_mem[e] * NC + k];
        dsum = sum;
      }
      __syncthreads();
      const int ri = rep * NC + k;
      for (int c = threadIdx.x; c <= n; c += blockDim.x) {
        // is c one of the member slots of this (group, slot)?
        bool member = false;
        if (c < n && c % NC == k) {
          const int cam = c / NC;
          for (int e = e0; e < e1 && !member; ++e) member = d.grp_mem[e] == cam;
        }
        if (!member) {
          at(ri, c) = v[c];
          for (int e = e0; e < e1 - 1; ++e) at(d.grp_mem[e] * NC + k, c) = 0.0;
        }
      }
      __syncthreads();
      // member x member block: representative diagonal = folded sum, other members identity, cross entries 0
      for (int t = threadIdx.x; t < (e1 - e0) * (e1 - e0); t += blockDim.x) {
        const int a = t / (e1 - e0), bq = t % (e1 - e0);
        if (bq > a) continue;
        const int ia = d.grp_mem[e0 + a] * NC + k, ib = d.grp_mem[e0 + bq] * NC + k;
        double val = 0.0;
        if (a == bq) val = (a == e1 - e0 - 1) ? dsum : 1.0;
        at(ia, ib) = val;
      }
      __syncthreads();
    }
  }
}

// after the solve: the members of a group take the representative's step for the shared slots
template <int TYPE>
__global__ void k_group_expand(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  double* y = d.yc + (size_t)sc * d.chol.np;
  for (int t = threadIdx.x; t < s.n_grp * NC; t += blockDim.x) {
    const int g = t / NC, k = t % NC;
    if (!is_intr_slot<TYPE>(k)) continue;
    const double yr = y[d.grp_mem[gp[g + 1] - 1] * NC + k];
    for (int e = gp[g]; e < gp[g + 1] - 1; ++e) y[d.grp_mem[e] * NC + k] = yr;
  }
}

// ---- lm_pre: TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue ---------------------
constexpr int LM_THREADS = 1024;  // the LM bookkeeping kernels are one workgroup per scene: wide, to shorten their reductions
template <int TYPE>
__global__ __launch_bounds__(LM_THREADS) void k_lm_pre(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  LmState& st = d.lm[sc];
  __shared__ double scratch[16];
  const int tid = threadIdx.x;
  if (st.step_is_successful) {
    // a fresh linearisation exists: cost, gradient max-norm (unscaled gradient), |x|
    double c = 0, gm = 0, xn = 0;
    const double* cam = cur_cam(d, s, st);
    const int* cp = d.cam_ptr + s.cam_off + s.idx;
    for (int i = tid; i < s.n_cam; i += LM_THREADS) {
      const int gi = s.cam_off + i;
      c += d.costc[gi];
      const double* gsrc = d.shared ? d.gfold : d.gc;  // shared intrinsics: the group's gradient sits at its representative
      for (int k = 0; k < NC; ++k) gm = fmax(gm, fabs(gsrc[(size_t)gi * NC + k] / d.scale_c[(size_t)gi * NC + k]));
      if (cp[i + 1] > cp[i]) {  // parameter blocks of cameras without residuals are not in the problem
        const bool intr = !d.shared || (d.cam_flag[gi] & 1);  // a shared intrinsics block is ONE block: counted once
        for (int k = 0; k < 15; ++k)
          if (intr || (k >= 4 && k < 10)) xn += cam[(size_t)i * 15 + k] * cam[(size_t)i * 15 + k];
      }
    }
    const double* ray = cur_ray(d, s, st);
    for (int j = tid; j < s.n_ray; j += LM_THREADS) {
      const int gj = s.ray_off + j;
      for (int k = 0; k < 3; ++k) {
        gm = fmax(gm, fabs(d.gr[(size_t)gj * 3 + k] / d.scale_r[(size_t)gj * 3 + k]));
        xn += ray[(size_t)j * 3 + k] * ray[(size_t)j * 3 + k];
      }
    }
    // fixed-order cost: per-thread partial sums over a strided camera set, then the block tree
    if (Dims<TYPE>::HAS3D && tid == 0) {
      const double* tl = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
      for (int k = 0; k < 6; ++k) {
        if (s.n_o3 > 0) xn += tl[k] * tl[k];  // the T_l_w block is in the problem only when annotation residuals exist
        gm = fmax(gm, fabs(d.gt[(size_t)s.idx * 6 + k] / d.scale_t[(size_t)s.idx * 6 + k]));
      }
    }
    c = block_sum(c, scratch);
    gm = block_max(gm, scratch);
    xn = block_sum(xn, scratch);
    if (tid == 0) {
      st.x_cost = c;
      st.it_cost = c;
      st.grad_max = gm;
      st.x_norm = sqrt(xn);
      st.need_linearize = 0;
      ++st.num_jac_evals;
      if (st.n_summaries == 0) { st.initial_cost = c; st.final_cost = c; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (st.step_is_successful) ++st.num_successful; else ++st.num_unsuccessful;
    if (st.it_cost < st.final_cost) st.final_cost = st.it_cost;
    ++st.n_summaries;
    if (st.iteration >= d.opt.max_num_iterations) { st.termination = PTZ_NO_CONVERGENCE; d.active[sc] = 0; }
    else if (st.step_is_successful && st.grad_max <= d.opt.gradient_tolerance) { st.termination = PTZ_CONVERGENCE; d.active[sc] = 0; }
    else if (st.radius <= d.opt.min_radius) { st.termination = PTZ_CONVERGENCE; d.active[sc] = 0; }
    else {
      ++st.iteration;
      ++st.num_lm_steps;
      st.step_is_successful = 0;
      d.ray_fail[sc] = 0;
    }
  }
}

// ---- ray_prep: LevenbergMarquardtStrategy diagonal + SchurEliminator e-block inverse --------------------
template <int TYPE>
__global__ __launch_bounds__(RAY_BLOCK) void k_ray_prep(Dev d)
{
  const int sc = blockIdx.y;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int j = blockIdx.x * RAY_BLOCK + threadIdx.x;
  if (j >= s.n_ray) return;
  const int gj = s.ray_off + j;
  double V[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) V[k] = d.V[(size_t)gj * 6 + k];
  double dg[3];
  if (!st.reuse_diagonal) {
    dg[0] = fmin(fmax(V[0], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
    dg[1] = fmin(fmax(V[2], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
    dg[2] = fmin(fmax(V[5], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
#pragma unroll
    for (int k = 0; k < 3; ++k) d.diag_r[(size_t)gj * 3 + k] = dg[k];
  }
  else {
#pragma unroll
    for (int k = 0; k < 3; ++k) dg[k] = d.diag_r[(size_t)gj * 3 + k];
  }
  // D = sqrt(diag / radius); V + D^2
  const double D0 = sqrt(dg[0] / st.radius), D1 = sqrt(dg[1] / st.radius), D2 = sqrt(dg[2] / st.radius);
  V[0] += D0 * D0; V[2] += D1 * D1; V[5] += D2 * D2;
  double E[6];
  if (!inv3_spd(V, E)) {
    d.ray_fail[sc] = 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) E[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) d.E[(size_t)gj * 6 + k] = E[k];
  const double g0 = d.gr[(size_t)gj * 3], g1 = d.gr[(size_t)gj * 3 + 1], g2 = d.gr[(size_t)gj * 3 + 2];
  d.z[(size_t)gj * 3 + 0] = E[0] * g0 + E[1] * g1 + E[3] * g2;
  d.z[(size_t)gj * 3 + 1] = E[1] * g0 + E[2] * g1 + E[4] * g2;
  d.z[(size_t)gj * 3 + 2] = E[3] * g0 + E[4] * g1 + E[5] * g2;
}

template <int TYPE>
__global__ void k_cam_diag(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cam || st.reuse_diagonal) return;
  const int gi = s.cam_off + i;
  if (Dims<TYPE>::HAS3D && i == 0)
    for (int k = 0; k < 6; ++k)
      d.diag_t[(size_t)s.idx * 6 + k] = fmin(fmax(d.Ut[(size_t)s.idx * 36 + k * 7], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
#pragma unroll
  for (int k = 0; k < NC; ++k)
    d.diag_c[(size_t)gi * NC + k] = fmin(fmax(d.U[(size_t)gi * NC * NC + k * NC + k], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
}

// ---- schur: one workgroup per camera ci -------------------------------------------------------------------
// Phase 1 (all threads, strided over ci's observations a): T_a = W_a E_ray(a) into LDS; at the same time the
//   diagonal block S_ii = U_i + D_i^2 - sum_a T_a W_a^T and the right-hand side b_i = g_i - sum_a W_a z_ray(a)
//   (z = E g_r) are reduced over the workgroup.  b is stored as row n of the padded system.
// Phase 2 (16-lane groups over ci's camera pairs (ci, cj < ci)): each lane takes entries e = lane, lane+16, ...
//   of the pair (T_a from LDS, W_b = one aligned 128-B line from L2), accumulates the whole NC x NC product in
//   registers, the group is reduced with a fixed butterfly and lane 0 stores S_ij = -sum.
#ifndef PTZ_SCHUR_THREADS
#define PTZ_SCHUR_THREADS 256
#endif
#ifndef PTZ_SCHUR_WAVES
#define PTZ_SCHUR_WAVES 3
#endif
constexpr int SCHUR_THREADS = PTZ_SCHUR_THREADS;
template <int TYPE>
__global__ __launch_bounds__(SCHUR_THREADS, PTZ_SCHUR_WAVES) void k_schur(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW;
  constexpr int NU = NW * (NW + 1) / 2;
  constexpr int NT = NW * 3;
  constexpr int TS = NT;  // row stride of the T table in LDS (an odd stride was measured: 25 % slower, it breaks the 16-byte reads of phase 2)
  int ci, sc;
  xcd_remap(ci, sc);
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (ci >= s.n_cam) return;
  const LmState& st = d.lm[sc];
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  const int o0 = cp[ci], no = cp[ci + 1] - o0;
  const int* cpair = d.cam_pair + s.cam_off + s.idx;
  const int* pp = d.pair_ptr + s.pair_off + s.idx;
  const int pr0 = cpair[ci], npr = cpair[ci + 1] - pr0;   // this camera's pairs
  const int eb = 0;                                        // entries are addressed by global index
  double* T = lds;                                  // [no][TS]
  double* strip = lds + (size_t)no * TS;            // [waves][NC + NU] reduction strip
  const int2* ents = d.ent + eb;                    // (a slot, W row of b), this camera's contiguous range
  const int* pps = pp + pr0;                        // entry offsets of this camera's pairs (global entry index)
  double bsum[NW], D[NU];
#pragma unroll
  for (int k = 0; k < NW; ++k) bsum[k] = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) D[k] = 0;
  for (int q = threadIdx.x; q < no; q += SCHUR_THREADS) {
    const int gj = d.cam_ray[o0 + q];  // global ray id of the q-th observation of this camera
    const double z0 = d.z[(size_t)gj * 3], z1 = d.z[(size_t)gj * 3 + 1], z2 = d.z[(size_t)gj * 3 + 2];
    const double* E = d.E + (size_t)gj * 6;
    const double e0 = E[0], e1 = E[1], e2 = E[2], e3 = E[3], e4 = E[4], e5 = E[5];
    double w[NT];
    const double* Wa = d.W + (size_t)(o0 + q) * Dims<TYPE>::WS;  // camera-major rows: sequential stream
#pragma unroll
    for (int k = 0; k < NT; ++k) w[k] = Wa[k];
    int e = 0;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const double w0 = w[3 * p], w1 = w[3 * p + 1], w2 = w[3 * p + 2];
      bsum[p] += w0 * z0 + w1 * z1 + w2 * z2;
      const double t0 = w0 * e0 + w1 * e1 + w2 * e3, t1 = w0 * e1 + w1 * e2 + w2 * e4, t2 = w0 * e3 + w1 * e4 + w2 * e5;
      T[q * TS + 3 * p] = t0; T[q * TS + 3 * p + 1] = t1; T[q * TS + 3 * p + 2] = t2;
#pragma unroll
      for (int qq = 0; qq <= p; ++qq) D[e++] += t0 * w[3 * qq] + t1 * w[3 * qq + 1] + t2 * w[3 * qq + 2];
    }
  }
  // one pass of the block tree for all NC + NU sums (fixed order: lanes by butterfly, waves in wave order)
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NV = NW + NU;
    double v[NV];
#pragma unroll
    for (int k = 0; k < NW; ++k) v[k] = wave_sum(bsum[k]);
#pragma unroll
    for (int k = 0; k < NU; ++k) v[NW + k] = wave_sum(D[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) strip[wv * NV + k] = v[k];
    }
    __syncthreads();  // also orders the T / ents / pps stores before phase 2
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double t = 0;
      for (int i = 0; i < SCHUR_THREADS / 64; ++i) t += strip[i * NV + k];
      v[k] = t;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) bsum[k] = v[k];
#pragma unroll
    for (int k = 0; k < NU; ++k) D[k] = v[NW + k];
  }
  const int np = d.chol.np;
  double* A = d.chol.A + (size_t)sc * np * np;
  if (threadIdx.x == 0) {
    const int gi = s.cam_off + ci;
    double* row = A + (size_t)s.n * np;
    // full NC x NC block: U (2D-2D + annotation terms) + D^2 - sum T W^T (the latter only on the NW x NW 2D-2D columns);
    // assembled in registers, stored once (both triangles: the block stays symmetric)
    double blk[NC * NC], rhs[NC];
#pragma unroll
    for (int p = 0; p < NC; ++p) {
      rhs[p] = d.gc[(size_t)gi * NC + p];
#pragma unroll
      for (int qq = 0; qq <= p; ++qq) blk[p * NC + qq] = d.U[(size_t)gi * NC * NC + p * NC + qq];
      const double Dd = sqrt(d.diag_c[(size_t)gi * NC + p] / st.radius);
      blk[p * NC + p] += Dd * Dd;
    }
    int e = 0;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      rhs[Dims<TYPE>::pos(p)] -= bsum[p];
#pragma unroll
      for (int qq = 0; qq <= p; ++qq) blk[Dims<TYPE>::pos(p) * NC + Dims<TYPE>::pos(qq)] -= D[e++];
    }
#pragma unroll
    for (int p = 0; p < NC; ++p) {
      row[ci * NC + p] = rhs[p];
#pragma unroll
      for (int qq = 0; qq <= p; ++qq) {
        A[(size_t)(ci * NC + p) * np + ci * NC + qq] = blk[p * NC + qq];
        A[(size_t)(ci * NC + qq) * np + ci * NC + p] = blk[p * NC + qq];
      }
    }
  }
  // ---- phase 2: off-diagonal blocks of row-block ci (index data and T from LDS, W_b lines from L2/HBM)
  const int l = threadIdx.x & 15;
  for (int pl = (threadIdx.x >> 4); pl < npr; pl += SCHUR_THREADS / 16) {
    double acc[NW * NW];
#pragma unroll
    for (int k = 0; k < NW * NW; ++k) acc[k] = 0;
    const int e1 = pps[pl + 1];
    int e = pps[pl] + l;
    // two entries per trip while both exist (both W_b rows in flight together), then at most one single entry
    for (; e + 16 < e1; e += 32) {
      const int2 ab0 = ents[e];
      const int2 ab1 = ents[e + 16];
      const double* Wb0 = d.W + (size_t)ab0.y * Dims<TYPE>::WS;
      const double* Wb1 = d.W + (size_t)ab1.y * Dims<TYPE>::WS;
      double wb0[NT], wb1[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) { wb0[k] = Wb0[k]; wb1[k] = Wb1[k]; }
      const double* Ta0 = T + ab0.x * TS;
      const double* Ta1 = T + ab1.x * TS;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        const double t0 = Ta0[3 * p], t1 = Ta0[3 * p + 1], t2 = Ta0[3 * p + 2];
        const double u0 = Ta1[3 * p], u1 = Ta1[3 * p + 1], u2 = Ta1[3 * p + 2];
#pragma unroll
        for (int q = 0; q < NW; ++q)
          acc[p * NW + q] += (t0 * wb0[3 * q] + t1 * wb0[3 * q + 1] + t2 * wb0[3 * q + 2]) +
                             (u0 * wb1[3 * q] + u1 * wb1[3 * q + 1] + u2 * wb1[3 * q + 2]);
      }
    }
    if (e < e1) {
      const int2 ab0 = ents[e];
      const double* Wb0 = d.W + (size_t)ab0.y * Dims<TYPE>::WS;
      double wb0[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) wb0[k] = Wb0[k];
      const double* Ta0 = T + ab0.x * TS;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        const double t0 = Ta0[3 * p], t1 = Ta0[3 * p + 1], t2 = Ta0[3 * p + 2];
#pragma unroll
        for (int q = 0; q < NW; ++q) acc[p * NW + q] += t0 * wb0[3 * q] + t1 * wb0[3 * q + 1] + t2 * wb0[3 * q + 2];
      }
    }
    // reduce-scatter over the 16 lanes of the group: after the steps with masks 8, 4, 2, 1 lane l holds the complete
    // sum of block element l (fixed order); every lane then stores its own element.  Elements >= NW*NW (NW = 5 keeps
    // 25 values) take a second pass with the lanes that are left.
    const int cj = d.pair_cj[s.pair_off + pr0 + pl];
    double* S = A + (size_t)(ci * NC) * np + cj * NC;
#pragma unroll
    for (int base = 0; base < NW * NW; base += 16) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (base + k < NW * NW) ? acc[(base + k < NW * NW) ? base + k : 0] : 0.0;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) {
        const bool up = (l & m) != 0;
#pragma unroll
        for (int k = 0; k < m; ++k) {
          const double keep = up ? v[k + m] : v[k];
          const double send = up ? v[k] : v[k + m];
          v[k] = keep + __shfl_xor(send, m, 16);
        }
      }
      const int el = base + l;
      if (el < NW * NW) S[(size_t)Dims<TYPE>::pos(el / NW) * np + Dims<TYPE>::pos(el % NW)] = -v[0];
    }
  }
}

// ---- schur_3d: rows of the T_l_w block in the reduced system (it is not coupled to the rays) ---------------------
//   S_tt = U_t + D_t^2,  S_t,cam(i) = sum_{annotations of camera i} Jt^T Jc,  b_t = g_t
template <int TYPE>
__global__ __launch_bounds__(64) void k_schur_3d(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int np = d.chol.np;
  double* A = d.chol.A + (size_t)sc * np * np;
  const int t0 = NC * s.n_cam;
  const int k = threadIdx.x;  // one lane per T_l_w row
  if (k >= 6) return;
  for (int l = 0; l <= k; ++l) {
    double v = d.Ut[(size_t)s.idx * 36 + k * 6 + l];
    if (l == k) {
      const double Dd = sqrt(d.diag_t[(size_t)s.idx * 6 + k] / st.radius);
      v += Dd * Dd;
    }
    A[(size_t)(t0 + k) * np + t0 + l] = v;
  }
  A[(size_t)s.n * np + t0 + k] = d.gt[(size_t)s.idx * 6 + k];
  for (int o = 0; o < s.n_o3; ++o) {  // observation order: deterministic accumulation into the (zeroed) coupling row
    const int go = s.o3_off + o;
    const int ci = d.o3_cam[go];
    const double q0 = d.Jt3[(size_t)go * 12 + k], q1 = d.Jt3[(size_t)go * 12 + 6 + k];
    const double* j0 = d.Jc3 + (size_t)go * 2 * NC;
    for (int l = 0; l < NC; ++l) A[(size_t)(t0 + k) * np + ci * NC + l] += q0 * j0[l] + q1 * j0[NC + l];
  }
}

// ---- cam_update: candidate cameras and their residual-side blocks ----------------------------------------
template <int TYPE>
__global__ void k_cam_update(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cam) return;
  const int gi = s.cam_off + i;
  const double* x = cur_cam(d, s, st) + (size_t)i * 15;
  double c15[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) c15[k] = x[k];
  const double* y = d.yc + (size_t)sc * d.chol.np + (size_t)i * NC;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double step = -y[k];
    d.dc[(size_t)gi * NC + k] = step;
    c15[Dims<TYPE>::at(k)] += step * d.scale_c[(size_t)gi * NC + k];
  }
  double* xc = d.cam_x + (size_t)(st.cur ^ 1) * d.cam_stride + (size_t)gi * 15;
#pragma unroll
  for (int k = 0; k < 15; ++k) xc[k] = c15[k];
  double cb[CANDBLK];
  fill_camblk(c15, cb, false);
#pragma unroll
  for (int k = 0; k < CANDBLK; ++k) d.candblk[(size_t)gi * CANDBLK + k] = cb[k];
  if (Dims<TYPE>::HAS3D && i == 0) {
    const double* t = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    double* tc = d.tlw_x + (size_t)(st.cur ^ 1) * d.tlw_stride + (size_t)s.idx * 6;
    const double* yt = d.yc + (size_t)sc * d.chol.np + (size_t)NC * s.n_cam;
    double tn[6];
    for (int k = 0; k < 6; ++k) {
      const double step = -yt[k];
      d.dt[(size_t)s.idx * 6 + k] = step;
      tn[k] = t[k] + step * d.scale_t[(size_t)s.idx * 6 + k];
      tc[k] = tn[k];
    }
    double* tb = d.tlwcand + (size_t)s.idx * TLWBLK;
    double R[9];
    const double rv[3] = {tn[0], tn[1], tn[2]};
    rodrigues(rv, R);
    for (int k = 0; k < 9; ++k) tb[k] = R[k];
    tb[18] = tn[3]; tb[19] = tn[4]; tb[20] = tn[5];
  }
}

// ---- eval: ray back-substitution, model cost change and candidate cost in one ray-centric pass -----------------
//   y_r = E (g_r - sum_a Jr_a^T (Jc_a y_c))            (SchurEliminator::BackSubstitute; W_a = Jc_a^T Jr_a is not
//                                                        re-read: the Jacobian blocks are recomputed, flops are free)
//   candidate ray = x + scale * (-y_r)
//   model_cost_change = -(J d)^T (r + J d / 2)          (TrustRegionMinimizer::ComputeTrustRegionStep)
//   candidate_cost    = 1/2 sum w |r(x + delta)|^2
// The scaled camera step d_c = -y_c is staged in LDS next to the camera tables of x and of the candidate.
template <int TYPE>
__global__ __launch_bounds__(RAY_BLOCK) void k_eval(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  const int sc = blockIdx.y;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (blockIdx.x >= s.n_chunk) return;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int DCS = NC | 1;                // odd stride for the step table as well
  double* tab = lds;                         // [n_cam][CBS]
  double* ctab = tab + s.n_cam * CBS;        // [n_cam][CDS]
  double* dct = ctab + s.n_cam * CDS;        // [n_cam][DCS] scaled camera step
  double* scratch = dct + s.n_cam * DCS;     // [16]
  stage_rows<CAMBLK, CBS>(d.camblk + (size_t)s.cam_off * CAMBLK, tab, s.n_cam);
  stage_rows<CANDBLK, CDS>(d.candblk + (size_t)s.cam_off * CANDBLK, ctab, s.n_cam);
  stage_rows<NC, DCS>(d.dc + (size_t)s.cam_off * NC, dct, s.n_cam);
  __syncthreads();
  const int j = blockIdx.x * RAY_BLOCK + threadIdx.x;
  double mcc = 0, cost = 0;
  if (j < s.n_ray) {
    const int gj = s.ray_off + j;
    const double* X = cur_ray(d, s, st) + (size_t)j * 3;
    const double Xr[3] = {X[0], X[1], X[2]};
    const double sr[3] = {d.scale_r[(size_t)gj * 3], d.scale_r[(size_t)gj * 3 + 1], d.scale_r[(size_t)gj * 3 + 2]};
    const double w = d.ray_w[gj];
    const double sw = sqrt(w);
    const int* rp = d.ray_ptr + s.ray_off + s.idx;
    const int a0 = rp[j], a1 = rp[j + 1];
    // pass 1 (one linearisation per observation): with p_a = Jc_a d_c (camera part of J d),
    //   t  = g_r + sum_a Jr_a^T p_a                      -> y_r = E t, ray step d_r = -y_r
    //   s1 = sum_a p_a . (r_a + p_a / 2)
    // and, since J d = p_a + Jr_a d_r per observation, the ray's share of (J d)^T (r + J d / 2) is
    //   s1 + d_r . t + 1/2 d_r^T V d_r      (V = sum_a Jr_a^T Jr_a is the stored, undamped ray block)
    double t0 = d.gr[(size_t)gj * 3], t1 = d.gr[(size_t)gj * 3 + 1], t2 = d.gr[(size_t)gj * 3 + 2];
    double s1 = 0;
    for (int a = a0; a < a1; ++a) {
      const float2 uv = d.obs_uv[a];
      const int ci = d.obs_cam[a];
      const double* cb = tab + ci * CBS;
      double res[2], Jc[2][NW], Jr[2][3];
      ba_linearize<F>(cb, Xr, uv.x, uv.y, res, Jc, Jr);
      double m0 = 0, m1 = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) { const double m = sw * cb[CB_S + Dims<TYPE>::pos(k)] * dct[ci * DCS + Dims<TYPE>::pos(k)]; m0 += Jc[0][k] * m; m1 += Jc[1][k] * m; }
      s1 += m0 * (res[0] * sw + m0 / 2.0) + m1 * (res[1] * sw + m1 / 2.0);
      t0 += sw * sr[0] * (Jr[0][0] * m0 + Jr[1][0] * m1);
      t1 += sw * sr[1] * (Jr[0][1] * m0 + Jr[1][1] * m1);
      t2 += sw * sr[2] * (Jr[0][2] * m0 + Jr[1][2] * m1);
    }
    const double* E = d.E + (size_t)gj * 6;
    // step = -y_r (Ceres solves J y = r and negates)
    const double ds[3] = {-(E[0] * t0 + E[1] * t1 + E[3] * t2), -(E[1] * t0 + E[2] * t1 + E[4] * t2), -(E[3] * t0 + E[4] * t1 + E[5] * t2)};
    const double Xn[3] = {Xr[0] + ds[0] * sr[0], Xr[1] + ds[1] * sr[1], Xr[2] + ds[2] * sr[2]};
    double* xc = d.ray_x + (size_t)(st.cur ^ 1) * d.ray_stride + (size_t)gj * 3;
    xc[0] = Xn[0]; xc[1] = Xn[1]; xc[2] = Xn[2];
    {
      const double* V = d.V + (size_t)gj * 6;  // [v00 v10 v11 v20 v21 v22]
      const double q0 = V[0] * ds[0] + V[1] * ds[1] + V[3] * ds[2];
      const double q1 = V[1] * ds[0] + V[2] * ds[1] + V[4] * ds[2];
      const double q2 = V[3] * ds[0] + V[4] * ds[1] + V[5] * ds[2];
      mcc = s1 + (ds[0] * t0 + ds[1] * t1 + ds[2] * t2) + 0.5 * (ds[0] * q0 + ds[1] * q1 + ds[2] * q2);
    }
    // pass 2: candidate cost (residuals only)
    for (int a = a0; a < a1; ++a) {
      const float2 uv = d.obs_uv[a];
      double rc[2];
      ba_residual<F>(ctab + d.obs_cam[a] * CDS, Xn, uv.x, uv.y, rc);
      cost += 0.5 * (w * (rc[0] * rc[0] + rc[1] * rc[1]));
    }
  }
  mcc = block_sum(mcc, scratch);
  cost = block_sum(cost, scratch);
  if (threadIdx.x == 0) {
    d.partial[(size_t)(s.part_off + blockIdx.x) * 2] = mcc;
    d.partial[(size_t)(s.part_off + blockIdx.x) * 2 + 1] = cost;
  }
}

// ---- eval_3d: annotation residuals' share of the model cost change and of the candidate cost -------------------
template <int TYPE>
__global__ __launch_bounds__(256) void k_eval_3d(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  __shared__ double scratch[16];
  double mcc = 0, cost = 0;
  for (int o = threadIdx.x; o < s.n_o3; o += 256) {
    const int go = s.o3_off + o;
    const int gi = s.cam_off + d.o3_cam[go];
    const double* j0 = d.Jc3 + (size_t)go * 2 * NC;
    const double* q0 = d.Jt3 + (size_t)go * 12;
    double m0 = 0, m1 = 0;
    for (int k = 0; k < NC; ++k) { const double st_ = d.dc[(size_t)gi * NC + k]; m0 += j0[k] * st_; m1 += j0[NC + k] * st_; }
    for (int k = 0; k < 6; ++k) { const double st_ = d.dt[(size_t)s.idx * 6 + k]; m0 += q0[k] * st_; m1 += q0[6 + k] * st_; }
    mcc += m0 * (d.r3[(size_t)go * 2] + m0 / 2.0) + m1 * (d.r3[(size_t)go * 2 + 1] + m1 / 2.0);
    double cb[CAMBLK];
    for (int k = 0; k < CANDBLK; ++k) cb[k] = d.candblk[(size_t)gi * CANDBLK + k];
    const float2 uv = d.o3_uv[go];
    const double xyz[3] = {d.o3_xyz[(size_t)go * 3], d.o3_xyz[(size_t)go * 3 + 1], d.o3_xyz[(size_t)go * 3 + 2]};
    double rc[2], Jc[2][5 + Dims<TYPE>::F3], Jt[2][6];
    reproj2d3d_eval<Dims<TYPE>::F3, false>(cb, d.tlwcand + (size_t)s.idx * TLWBLK, xyz, uv.x, uv.y, rc, Jc, Jt);
    cost += 0.5 * (rc[0] * rc[0] + rc[1] * rc[1]);
  }
  mcc = block_sum(mcc, scratch);
  cost = block_sum(cost, scratch);
  if (threadIdx.x == 0) {
    d.partial[(size_t)(s.part_off + s.n_chunk) * 2] = mcc;
    d.partial[(size_t)(s.part_off + s.n_chunk) * 2 + 1] = cost;
  }
}

// ---- lm_post: the body of TrustRegionMinimizer::Minimize after the step has been computed --------------
template <int TYPE>
__global__ __launch_bounds__(LM_THREADS) void k_lm_post(Dev d)
{
  const int sc = blockIdx.x;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  LmState& st = d.lm[sc];
  __shared__ double scratch[16];
  const int tid = threadIdx.x;
  // chunk partials in chunk order (thread-strided, then the fixed block tree)
  double mcc = 0, cost = 0;
  for (int c = tid; c < s.n_chunk + Dims<TYPE>::HAS3D; c += LM_THREADS) {
    mcc += d.partial[(size_t)(s.part_off + c) * 2];
    cost += d.partial[(size_t)(s.part_off + c) * 2 + 1];
  }
  mcc = -block_sum(mcc, scratch);
  cost = block_sum(cost, scratch);
  // |x - x_candidate| and |x_candidate| over the parameter blocks that are in the problem
  const double* cam = cur_cam(d, s, st);
  const double* camc = d.cam_x + (size_t)(st.cur ^ 1) * d.cam_stride + (size_t)s.cam_off * 15;
  const double* ray = cur_ray(d, s, st);
  const double* rayc = d.ray_x + (size_t)(st.cur ^ 1) * d.ray_stride + (size_t)s.ray_off * 3;
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  double dn = 0, cn = 0;
  for (int i = tid; i < s.n_cam; i += LM_THREADS) {
    if (cp[i + 1] <= cp[i]) continue;
    const bool intr = !d.shared || (d.cam_flag[s.cam_off + i] & 1);
    for (int k = 0; k < 15; ++k) {
      if (!intr && (k < 4 || k >= 10)) continue;
      const double a = cam[(size_t)i * 15 + k], b = camc[(size_t)i * 15 + k];
      dn += (a - b) * (a - b);
      cn += b * b;
    }
  }
  for (int j = tid; j < s.n_ray * 3; j += LM_THREADS) {
    const double a = ray[j], b = rayc[j];
    dn += (a - b) * (a - b);
    cn += b * b;
  }
  if (Dims<TYPE>::HAS3D && tid == 0 && s.n_o3 > 0) {
    const double* ta = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    const double* tb_ = d.tlw_x + (size_t)(st.cur ^ 1) * d.tlw_stride + (size_t)s.idx * 6;
    for (int k = 0; k < 6; ++k) { dn += (ta[k] - tb_[k]) * (ta[k] - tb_[k]); cn += tb_[k] * tb_[k]; }
  }
  dn = block_sum(dn, scratch);
  cn = block_sum(cn, scratch);
  if (tid != 0) return;
  const Opt& o = d.opt;
  ++st.num_linear_solves;
  st.reuse_diagonal = 1;  // LevenbergMarquardtStrategy::ComputeStep
  const bool solve_fail = d.ray_fail[sc] || d.chol.fail[sc];
  const bool valid = !solve_fail && isfinite(mcc) && isfinite(dn) && mcc > 0.0;
  st.model_cost_change = mcc;
  st.it_cost = st.x_cost;
  if (!valid) {  // HandleInvalidStep
    ++st.num_consecutive_invalid;
    if (st.num_consecutive_invalid >= o.max_consecutive_invalid) { st.termination = PTZ_FAILURE; d.active[sc] = 0; return; }
    st.radius *= 0.5;  // StepIsInvalid
    st.reuse_diagonal = 0;
    return;
  }
  st.num_consecutive_invalid = 0;
  if (!isfinite(cost)) cost = 1.7976931348623157e308;
  st.candidate_cost = cost;
  // ParameterToleranceReached
  if (sqrt(dn) <= o.parameter_tolerance * (st.x_norm + o.parameter_tolerance)) { st.termination = PTZ_CONVERGENCE; d.active[sc] = 0; return; }
  // FunctionToleranceReached
  const double cost_change = st.x_cost - cost;
  if (fabs(cost_change) <= o.function_tolerance * st.x_cost) { st.termination = PTZ_CONVERGENCE; d.active[sc] = 0; return; }
  const double rho = cost_change / mcc;  // TrustRegionStepEvaluator::StepQuality, monotonic steps
  if (rho > o.min_relative_decrease) {
    // HandleSuccessfulStep: x <- candidate; the Jacobian is re-evaluated by the kernels that follow
    st.cur ^= 1;
    st.need_linearize = 1;
    st.step_is_successful = 1;
    const double t = 2.0 * rho - 1.0;
    st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);  // StepAccepted
    st.radius = fmin(o.max_radius, st.radius);
    st.decrease_factor = 2.0;
    st.reuse_diagonal = 0;
  }
  else {
    // HandleUnsuccessfulStep / StepRejected
    st.it_cost = cost;
    st.radius = st.radius / st.decrease_factor;
    st.decrease_factor *= 2.0;
    st.reuse_diagonal = 1;
  }
}

// ---- reset / init -----------------------------------------------------------------------------------------
__global__ void k_reset(Dev d)
{
  const int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc >= d.n_scene) return;
  LmState st;
  memset(&st, 0, sizeof(st));
  st.radius = d.opt.initial_radius;
  st.decrease_factor = 2.0;
  st.need_linearize = 1;
  st.step_is_successful = 1;
  st.termination = PTZ_NO_CONVERGENCE;
  d.lm[sc] = st;
  d.active[sc] = 1;
  d.ray_fail[sc] = 0;
}
__global__ void k_fill(double* p, size_t n, double v)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// Pix2Ray (ptzray_optimizer.cc:768-797): ray = normalise(mean_i normalise(R_i^-1 K_i^-1 [u, v, 1]))
__global__ __launch_bounds__(RAY_BLOCK) void k_pix2ray(Dev d, double* cam0, double* ray0)
{
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const int j = blockIdx.x * RAY_BLOCK + threadIdx.x;
  if (j >= s.n_ray) return;
  const int* rp = d.ray_ptr + s.ray_off + s.idx;
  double acc[3] = {0, 0, 0};
  int cnt = 0;
  for (int a = rp[j]; a < rp[j + 1]; ++a) {
    const double* c = cam0 + (size_t)(s.cam_off + d.obs_cam[a]) * 15;
    double R[9];
    rodrigues(c + 4, R);
    const float2 uv = d.obs_uv[a];
    const double q0 = ((double)uv.x - c[2]) / c[0], q1 = ((double)uv.y - c[3]) / c[1];
    // R^-1 = R^T for a rotation (the reference inverts numerically; identical to round-off)
    double t0 = R[0] * q0 + R[3] * q1 + R[6], t1 = R[1] * q0 + R[4] * q1 + R[7], t2 = R[2] * q0 + R[5] * q1 + R[8];
    const double n = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    acc[0] += t0 / n; acc[1] += t1 / n; acc[2] += t2 / n;
    ++cnt;
  }
  acc[0] /= cnt; acc[1] /= cnt; acc[2] /= cnt;
  const double n = sqrt(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
  double* out = ray0 + (size_t)(s.ray_off + j) * 3;
  out[0] = acc[0] / n; out[1] = acc[1] / n; out[2] = acc[2] / n;
}

}  // namespace

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
void chol_factor_solve_profiled(const CholBatch& cb, double* x, hipStream_t stream, void* prof);
}

struct ptz_ba_batch {
  int n_scene = 0, type = 0, nc = 4, device = 0;
  std::vector<SceneDev> scenes;
  int total_cam = 0, total_ray = 0, total_obs = 0, total_pair = 0, total_ent = 0, total_chunk = 0;
  int max_cam = 0, max_ray = 0, max_chunk = 0, max_pair = 0, max_n = 0, max_cam_obs = 0, max_cam_ent = 0, max_cam_pair = 0;
  ptz_lm_options opt;
  Dev d;
  std::vector<void*> allocs;
  hipStream_t stream = nullptr;   // stream of the group being enqueued (LAUNCH / prof_* use it)
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
  int* h_active = nullptr;  // pinned
  double *cam0 = nullptr, *ray0 = nullptr, *tlw0 = nullptr;  // device copies of the initial state
  int has3d = 0, total_o3 = 0;
  // shared intrinsics: per global camera, the global index of the first camera of its group (source of the initial values)
  std::vector<int> first_of_group;
  int max_grp = 0;
  bool has_state = false;
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
inline size_t schur_lds_bytes(int max_obs, int max_ent, int max_pair, int NC)
{
  (void)max_ent; (void)max_pair;
  return sizeof(double) * ((size_t)max_obs * NC * 3 + (size_t)(SCHUR_THREADS / 64) * (NC + NC * (NC + 1) / 2));
}

template <typename T> int upload(ptz_ba_batch* b, const std::vector<T>& h, const T** dev)
{
  T* p = nullptr;
  int rc = b->alloc(&p, h.size());
  if (rc) return rc;
  if (!h.empty() && hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return PTZ_ENODEVICE;
  *dev = p;
  return PTZ_OK;
}

#define LAUNCH(kern, grid, block, smem, ...) hipLaunchKernelGGL(kern, grid, block, smem, b->stream, __VA_ARGS__)

template <int TYPE> void enqueue_linearize(ptz_ba_batch* b)
{
  const Dev& d = b->d;
  b->prof_begin(P_LIN);
  LAUNCH(k_cam_prep<TYPE>, dim3((b->max_cam + 63) / 64, b->n_scene), dim3(64), 0, d);
  LAUNCH(k_lin_ray<TYPE>, dim3(b->max_chunk, b->n_scene), dim3(RAY_BLOCK), sizeof(double) * b->max_cam * CBS, d);
  LAUNCH(k_lin_cam<TYPE>, dim3((b->max_cam + 3) / 4, b->n_scene), dim3(256), 0, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_lin_3d<TYPE>, dim3(b->n_scene), dim3(256), 0, d);
  if (d.shared) LAUNCH(k_group_grad<TYPE>, dim3(b->n_scene), dim3(256), 0, d);
  b->prof_end();
}

static void make_groups(ptz_ba_batch* b)
{
  const int B = b->n_scene;
  // per-family timings are only meaningful when no other group's kernels share the device: profiling runs one group
  const int G = b->profiling ? 1 : std::max(1, std::min(b->n_group, B));
  b->group_first.clear(); b->group_count.clear(); b->dg.clear();
  for (int g = 0; g < G; ++g) {
    const int lo = (int)((int64_t)B * g / G), hi = (int)((int64_t)B * (g + 1) / G);
    b->group_first.push_back(lo);
    b->group_count.push_back(hi - lo);
    Dev d = b->d;  // per-scene arrays are re-based; everything else is addressed through SceneDev offsets
    const size_t np = d.chol.np, nt = np / CHOL_NB;
    d.n_scene = hi - lo;
    d.scene += lo; d.lm += lo; d.active += lo; d.ray_fail += lo;
    d.yc += (size_t)lo * np;
    d.chol.count = hi - lo;
    d.chol.A += (size_t)lo * np * np;
    d.chol.Ldiag += (size_t)lo * nt * CHOL_NB * CHOL_NB;
    d.chol.Dinv += (size_t)lo * nt * 4 * 16 * 16;
    d.chol.n += lo; d.chol.fail += lo; d.chol.active = d.active;
    if (d.chol.tmask) d.chol.tmask += (size_t)lo * nt * nt;
    b->dg.push_back(d);
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

// one LM pass of one group, enqueued on b->stream; returns after enqueueing (no synchronisation)
template <int TYPE> void enqueue_pass(ptz_ba_batch* b, const Dev& d, bool last)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int B = d.n_scene;
  hipStream_t st = b->stream;
  const size_t eval_smem = sizeof(double) * ((size_t)b->max_cam * (CBS + CDS + (NC | 1)) + 16);
  const size_t schur_smem = schur_lds_bytes(b->max_cam_obs, b->max_cam_ent, b->max_cam_pair, NC);
  b->prof_begin(P_LMCTL);
  LAUNCH(k_lm_pre<TYPE>, dim3(B), dim3(LM_THREADS), 0, d);
  b->prof_end();
  if (last) return;
  b->prof_begin(P_RAYPREP);
  LAUNCH(k_ray_prep<TYPE>, dim3(b->max_chunk, B), dim3(RAY_BLOCK), 0, d);
  LAUNCH(k_cam_diag<TYPE>, dim3((b->max_cam + 63) / 64, B), dim3(64), 0, d);
  if (d.shared) LAUNCH(k_group_diag<TYPE>, dim3((b->max_grp * NC + 63) / 64, B), dim3(64), 0, d);
  b->prof_end();
  b->prof_begin(P_CLEAR);
  chol_clear(d.chol, st);
  b->prof_end();
  b->prof_begin(P_SCHUR);
  LAUNCH(k_schur<TYPE>, dim3(b->max_cam, B), dim3(SCHUR_THREADS), schur_smem, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_schur_3d<TYPE>, dim3(B), dim3(64), 0, d);
  if (d.shared) LAUNCH(k_fold_system<TYPE>, dim3(B), dim3(1024), sizeof(double) * (size_t)(b->max_n + 2), d);
  b->prof_end();
  chol_factor_solve_profiled(d.chol, d.yc, st, b);
  if (d.shared) LAUNCH(k_group_expand<TYPE>, dim3(B), dim3(256), 0, d);
  b->prof_begin(P_BACKSUB);
  LAUNCH(k_cam_update<TYPE>, dim3((b->max_cam + 63) / 64, B), dim3(64), 0, d);
  b->prof_end();
  b->prof_begin(P_EVAL);
  LAUNCH(k_eval<TYPE>, dim3(b->max_chunk, B), dim3(RAY_BLOCK), eval_smem, d);
  if (Dims<TYPE>::HAS3D) LAUNCH(k_eval_3d<TYPE>, dim3(B), dim3(256), 0, d);
  b->prof_end();
  b->prof_begin(P_LMCTL);
  LAUNCH(k_lm_post<TYPE>, dim3(B), dim3(LM_THREADS), 0, d);
  b->prof_end();
  {
    const Dev& dd = d;
    b->prof_begin(P_LIN);
    LAUNCH(k_cam_prep<TYPE>, dim3((b->max_cam + 63) / 64, B), dim3(64), 0, dd);
    LAUNCH(k_lin_ray<TYPE>, dim3(b->max_chunk, B), dim3(RAY_BLOCK), sizeof(double) * b->max_cam * CBS, dd);
    LAUNCH(k_lin_cam<TYPE>, dim3((b->max_cam + 3) / 4, B), dim3(256), 0, dd);
    if (Dims<TYPE>::HAS3D) LAUNCH(k_lin_3d<TYPE>, dim3(B), dim3(256), 0, dd);
    if (dd.shared) LAUNCH(k_group_grad<TYPE>, dim3(B), dim3(256), 0, dd);
    b->prof_end();
  }
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
  PTZ_HIP_TRY(hipEventRecord(b->ev0, s0));
  // x <- initial state, scales <- 1, LM state reset (whole batch, stream 0)
  PTZ_HIP_TRY(hipMemcpyAsync(d.cam_x, b->cam0, sizeof(double) * 15 * b->total_cam, hipMemcpyDeviceToDevice, s0));
  PTZ_HIP_TRY(hipMemcpyAsync(d.ray_x, b->ray0, sizeof(double) * 3 * b->total_ray, hipMemcpyDeviceToDevice, s0));
  LAUNCH(k_reset, dim3((B + 63) / 64), dim3(64), 0, d);
  LAUNCH(k_fill, dim3(((size_t)b->total_cam * NC + 255) / 256), dim3(256), 0, d.scale_c, (size_t)b->total_cam * NC, 1.0);
  LAUNCH(k_fill, dim3(((size_t)b->total_ray * 3 + 255) / 256), dim3(256), 0, d.scale_r, (size_t)b->total_ray * 3, 1.0);
  // both halves of the double buffer: without annotation residuals no kernel ever writes the candidate half, and the
  // accepted-step parity decides which half is read back
  PTZ_HIP_TRY(hipMemcpyAsync(d.tlw_x, b->tlw0, sizeof(double) * 6 * B, hipMemcpyDeviceToDevice, s0));
  PTZ_HIP_TRY(hipMemcpyAsync(d.tlw_x + d.tlw_stride, b->tlw0, sizeof(double) * 6 * B, hipMemcpyDeviceToDevice, s0));
  LAUNCH(k_fill, dim3((6 * B + 255) / 256), dim3(256), 0, d.scale_t, (size_t)6 * B, 1.0);
  // IterationZero: evaluate, Jacobi scales from the column norms, re-evaluate scaled
  enqueue_linearize<TYPE>(b);
  if (b->opt.jacobi_scaling) {
    LAUNCH(k_jacobi_scale<TYPE>, dim3((std::max(b->max_cam, b->max_ray) + 255) / 256, B), dim3(256), 0, d);
    if (d.shared) LAUNCH(k_group_scale<TYPE>, dim3((b->max_grp * NC + 63) / 64, B), dim3(64), 0, d);
    enqueue_linearize<TYPE>(b);
  }
  // fork: every group's stream continues after the common prologue
  PTZ_HIP_TRY(hipEventRecord(b->fork_ev[0], s0));
  for (int g = 1; g < G; ++g) PTZ_HIP_TRY(hipStreamWaitEvent(b->streams[g], b->fork_ev[0], 0));
  const int max_it = b->opt.max_num_iterations;
  // Every group runs its own pass pipeline: the host waits for a group's active flags, enqueues that group's next pass at once
  // and only then turns to the next group.  The groups therefore drift out of phase, and the latency-bound part of one
  // group's pass (block-column chain of the factorisation, LM control) overlaps the throughput kernels of another.
  std::vector<char> galive(G, 1);
  std::vector<int> gpass(G, 0);
  auto enqueue_group = [&](int g) -> int {
    const double te0 = now();
    b->stream = b->streams[g];
    enqueue_pass<TYPE>(b, b->dg[g], gpass[g] == max_it);
    if (gpass[g] < max_it) {
      b->prof_begin(P_SYNC);
      PTZ_HIP_TRY(hipMemcpyAsync(b->h_active + b->group_first[g], b->dg[g].active, sizeof(int) * b->group_count[g],
                                 hipMemcpyDeviceToHost, b->streams[g]));
      b->prof_end();
    }
    t_enq += now() - te0;
    return PTZ_OK;
  };
  int alive = G;
  for (int g = 0; g < G; ++g) {
    if (enqueue_group(g) != PTZ_OK) return PTZ_ENODEVICE;
    if (max_it == 0) { galive[g] = 0; --alive; }
  }
  while (alive > 0) {
    for (int g = 0; g < G; ++g) {
      if (!galive[g]) continue;
      const double ts0 = now();
      PTZ_HIP_TRY(hipStreamSynchronize(b->streams[g]));
      bool any = false;
      int na = 0;
      for (int i = 0; i < b->group_count[g]; ++i) { const bool a = b->h_active[b->group_first[g] + i] != 0; any |= a; na += a; }
      t_sync += now() - ts0;
      if (dbg) fprintf(stderr, "[ptz_ba] group %d pass %d: %d scenes active afterwards (t = %.3f ms)\n", g, gpass[g], na, now() - t_start);
      if (!any) { galive[g] = 0; --alive; continue; }
      ++gpass[g];
      if (enqueue_group(g) != PTZ_OK) return PTZ_ENODEVICE;
      if (gpass[g] == max_it) { galive[g] = 0; --alive; }  // the last pass only closes the books (k_lm_pre)
    }
  }
  // join
  for (int g = 1; g < G; ++g) {
    PTZ_HIP_TRY(hipEventRecord(b->join_ev[g], b->streams[g]));
    PTZ_HIP_TRY(hipStreamWaitEvent(s0, b->join_ev[g], 0));
  }
  b->stream = s0;
  PTZ_HIP_TRY(hipEventRecord(b->ev1, s0));
  PTZ_HIP_TRY(hipStreamSynchronize(s0));
  PTZ_HIP_TRY(hipGetLastError());  // a kernel launch that was refused (resources, arguments) must not pass for a solve
  float ms = 0;
  (void)hipEventElapsedTime(&ms, b->ev0, b->ev1);
  b->last_ms = ms;
  b->prof_collect();
  if (dbg) fprintf(stderr, "[ptz_ba] groups %d: total %.2f ms, enqueue %.2f ms, sync-wait %.2f ms, device %.2f ms\n", G, now() - t_start, t_enq, t_sync, ms);
  if (out) {
    std::vector<LmState> h(B);
    PTZ_HIP_TRY(hipMemcpy(h.data(), d.lm, sizeof(LmState) * B, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; ++i) {
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
void chol_factor_solve_profiled(const CholBatch& cb, double* x, hipStream_t stream, void* prof)
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
  chol_backsolve_launch(cb, x, stream);
  b->prof_end();
}
}  // namespace ptz

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

int32_t ptz_ba_cam_block_dim(int32_t factor_type)
{
  if (factor_type == PTZ_BA_PTZRay) return 4;
  if (factor_type == PTZ_BA_PTZRayDist) return 5;
  if (factor_type == PTZ_BA_PTZRayFxfyDist) return 6;
  return PTZ_EUNSUPPORTED;
}

int32_t ptz_ba_batch_cam_block_dim(const ptz_ba_batch* b) { return b ? b->nc : PTZ_EINVAL; }

void ptz_ba_batch_destroy(ptz_ba_batch* b)
{
  if (!b) return;
  (void)hipSetDevice(b->device);
  // nothing of this batch may still be running when its memory is handed to the next one
  for (auto st : b->streams) (void)hipStreamSynchronize(st);
  for (auto st : b->aux) (void)hipStreamSynchronize(st);
  const int dv = b->device;
  for (void* p : b->allocs) ptzpool::dev_release(dv, p);
  for (auto e : b->ev_pool) ptzpool::event_release(dv, true, e);
  if (b->h_active) ptzpool::pinned_release(b->h_active);
  ptzpool::event_release(dv, true, b->ev0);
  ptzpool::event_release(dv, true, b->ev1);
  for (auto st : b->streams) ptzpool::stream_release(dv, st);
  for (auto st : b->aux) ptzpool::stream_release(dv, st);
  for (auto e : b->la_ev) ptzpool::event_release(dv, false, e);
  for (auto e : b->fork_ev) ptzpool::event_release(dv, false, e);
  for (auto e : b->join_ev) ptzpool::event_release(dv, false, e);
  delete b;
}

int32_t ptz_ba_batch_create(int32_t n, const ptz_ba_problem* problems, const ptz_lm_options* opt, ptz_ba_batch** out)
{
  if (n <= 0 || !problems || !out) return PTZ_EINVAL;
  *out = nullptr;
  const bool dbg_t = getenv("PTZ_BA_DEBUG_TIMING") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tc0 = now_ms();
  double tc1 = 0, tc2 = 0, tc3 = 0, ts_obs = 0, ts_ent = 0;
  ptz_lm_options o;
  if (opt) o = *opt; else ptz_lm_options_default(&o);
  if (o.max_num_iterations <= 0) return PTZ_EINVAL;  // CheckValid, ptzray_optimizer.cc:521
  const int type = problems[0].factor_type;
  if (type != PTZ_BA_PTZRay && type != PTZ_BA_PTZRayDist && type != PTZ_BA_PTZRayFxfyDist) return PTZ_EUNSUPPORTED;
  int has3d = 0;
  // ---- validate + sizes (host only; no device touched before this passes)
  for (int i = 0; i < n; ++i) {
    const ptz_ba_problem& p = problems[i];
    if (p.factor_type != type) return PTZ_EINVAL;
    if (p.n_cam <= 0 || p.n_ray <= 0 || p.n_obs <= 0) return PTZ_EINVAL;  // num_cams_ == 0 -> false (:517)
    if (!p.obs_uv || !p.obs_cam || !p.obs_ray || !p.ray_weight) return PTZ_EINVAL;
    if (p.n_obs3d < 0 || (p.n_obs3d > 0 && (!p.obs3d_uv || !p.obs3d_xyz || !p.obs3d_cam))) return PTZ_EINVAL;
    for (int a = 0; a < p.n_obs3d; ++a)
      if (p.obs3d_cam[a] < 0 || p.obs3d_cam[a] >= p.n_cam) return PTZ_EINVAL;
    if (p.n_obs3d > 0) has3d = 1;
    for (int64_t a = 0; a < p.n_obs; ++a) {
      if (p.obs_cam[a] < 0 || p.obs_cam[a] >= p.n_cam || p.obs_ray[a] < 0 || p.obs_ray[a] >= p.n_ray) return PTZ_EINVAL;
      if (a > 0 && p.obs_ray[a] < p.obs_ray[a - 1]) return PTZ_EINVAL;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= o.device_id) return PTZ_ENODEVICE;
  PTZ_HIP_TRY(hipSetDevice(o.device_id));

  // fy becomes a live column with annotation residuals; PTZRayFxfyDist has it anyway
  const int NC = type == PTZ_BA_PTZRayFxfyDist ? 6 : ((type == PTZ_BA_PTZRay) ? 4 : 5) + has3d;
  ptz_ba_batch* b = new ptz_ba_batch();
  b->has3d = has3d;
  b->n_scene = n; b->type = type; b->nc = NC; b->opt = o; b->device = o.device_id;
  std::vector<float2> h_uv;
  std::vector<int> h_cam, h_ray, h_rayptr, h_camptr, h_camobs, h_pci, h_pcj, h_pptr, h_wpos, h_camray;
  std::vector<int2> h_ent;
  std::vector<int> h_campair;
  std::vector<double> h_w, h_o3xyz;
  std::vector<float2> h_o3uv;
  std::vector<int> h_o3cam;
  std::vector<int> h_grpptr, h_grpmem;
  {
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
  h_uv.reserve(tot_obs); h_cam.reserve(tot_obs); h_ray.reserve(tot_obs); h_camobs.reserve(tot_obs);
  for (int i = 0; i < n; ++i) {
    const ptz_ba_problem& p = problems[i];
    SceneDev s;
    s.n_cam = p.n_cam; s.n_ray = p.n_ray; s.n_obs = (int)p.n_obs;
    s.cam_off = b->total_cam; s.ray_off = b->total_ray; s.obs_off = b->total_obs;
    s.pair_off = b->total_pair; s.ent_off = b->total_ent; s.part_off = b->total_chunk;
    s.n_chunk = (p.n_ray + RAY_BLOCK - 1) / RAY_BLOCK;
    s.n = NC * p.n_cam + 6 * has3d;
    s.idx = i;
    s.o3_off = b->total_o3; s.n_o3 = p.n_obs3d;
    for (int a = 0; a < p.n_obs3d; ++a) {
      h_o3uv.push_back(make_float2(p.obs3d_uv[2 * a], p.obs3d_uv[2 * a + 1]));
      h_o3cam.push_back(p.obs3d_cam[a]);
      for (int k = 0; k < 3; ++k) h_o3xyz.push_back(p.obs3d_xyz[3 * a + k]);
    }
    b->total_o3 += p.n_obs3d;
    // observations, ray ranges
    const double tsa = now_ms();
    const int obase = s.obs_off;
    std::vector<int> cnt_ray(p.n_ray + 1, 0), cnt_cam(p.n_cam + 1, 0);
    for (int64_t a = 0; a < p.n_obs; ++a) {
      h_uv.push_back(make_float2(p.obs_uv[2 * a], p.obs_uv[2 * a + 1]));
      h_cam.push_back(p.obs_cam[a]);
      h_ray.push_back(p.obs_ray[a]);
      ++cnt_ray[p.obs_ray[a] + 1];
      ++cnt_cam[p.obs_cam[a] + 1];
    }
    for (int j = 0; j < p.n_ray; ++j) {
      if (cnt_ray[j + 1] == 0) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }  // every ray has >= 1 observation
      cnt_ray[j + 1] += cnt_ray[j];
    }
    for (int j = 0; j <= p.n_ray; ++j) h_rayptr.push_back(obase + cnt_ray[j]);
    for (int j = 0; j < p.n_ray; ++j) h_w.push_back(p.ray_weight[j]);
    // camera-major observation lists
    for (int c = 0; c < p.n_cam; ++c) cnt_cam[c + 1] += cnt_cam[c];
    {
      std::vector<int> fill(cnt_cam.begin(), cnt_cam.end() - 1);
      const size_t base = h_camobs.size();
      h_camobs.resize(base + p.n_obs);
      h_camray.resize(base + p.n_obs);
      for (int64_t a = 0; a < p.n_obs; ++a) {
        const int slot = fill[p.obs_cam[a]]++;
        h_camobs[base + slot] = obase + (int)a;
        h_camray[base + slot] = s.ray_off + p.obs_ray[a];
      }
      for (int c = 0; c <= p.n_cam; ++c) h_camptr.push_back(obase + cnt_cam[c]);
    }
    const double tsb = now_ms();
    ts_obs += tsb - tsa;
    // camera-pair entry lists (off-diagonal blocks): for every ray, every (a, b) with cam(a) > cam(b);
    // a is stored as its position in cam(a)'s observation list (the LDS slot of T_a in k_schur)
    {
      std::vector<int> pos(p.n_obs);
      {
        std::vector<int> fill(p.n_cam, 0);
        for (int64_t a = 0; a < p.n_obs; ++a) pos[a] = fill[p.obs_cam[a]]++;   // camera-major order = ascending a
        for (int c = 0; c < p.n_cam; ++c) b->max_cam_obs = std::max(b->max_cam_obs, fill[c]);
        for (int64_t a = 0; a < p.n_obs; ++a) h_wpos.push_back(obase + cnt_cam[p.obs_cam[a]] + pos[a]);
      }
      // counting sort by (ci, cj): pairs ascending in ci * n_cam + cj, the entries of a pair in ray order (stable).
      // One pass over the rays lists the (a, b) pairs with cam(a) > cam(b) in ray order and counts them per camera pair;
      // the fill pass then runs over that flat list only.  Observations of a track come camera-ascending from the packing
      // (track asc, image asc: ptzray_optimizer.cc:801-850), in which case the pairs are simply (a, b < a); any other order
      // takes the general double loop.
      const size_t ncc = (size_t)p.n_cam * p.n_cam;
      std::vector<int> pair_cnt(ncc, 0);
      std::vector<int> ea, ebq;
      ea.reserve((size_t)p.n_obs * 4);
      ebq.reserve((size_t)p.n_obs * 4);
      for (int j = 0; j < p.n_ray; ++j) {
        const int r0 = cnt_ray[j], r1 = cnt_ray[j + 1];
        bool ascending = true;
        for (int a = r0 + 1; a < r1; ++a) ascending &= p.obs_cam[a] > p.obs_cam[a - 1];
        if (ascending) {
          for (int a = r0 + 1; a < r1; ++a) {
            const size_t row = (size_t)p.obs_cam[a] * p.n_cam;
            for (int bb = r0; bb < a; ++bb) {
              ++pair_cnt[row + p.obs_cam[bb]];
              ea.push_back(a);
              ebq.push_back(bb);
            }
          }
        }
        else {
          for (int a = r0; a < r1; ++a)
            for (int bb = r0; bb < r1; ++bb) {
              const int ci = p.obs_cam[a], cj = p.obs_cam[bb];
              if (ci == cj && a != bb) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }  // an image appears once per track (tracks.cc:77)
              if (ci <= cj) continue;
              ++pair_cnt[(size_t)ci * p.n_cam + cj];
              ea.push_back(a);
              ebq.push_back(bb);
            }
        }
      }
      const int64_t n_ent = (int64_t)ea.size();
      if ((int64_t)b->total_ent + n_ent > 0x7fffffff) { ptz_ba_batch_destroy(b); return PTZ_EINVAL; }
      int npair = 0;
      std::vector<int> cam_first(p.n_cam + 1, -1), cam_ent(p.n_cam, 0);
      std::vector<int> pair_fill(ncc, -1);  // next free entry slot of a pair (scene-local)
      {
        int run = 0;
        for (int ci = 0; ci < p.n_cam; ++ci)
          for (int cj = 0; cj < ci; ++cj) {
            const int c = pair_cnt[(size_t)ci * p.n_cam + cj];
            if (c == 0) continue;
            h_pci.push_back(ci);
            h_pcj.push_back(cj);
            h_pptr.push_back(b->total_ent + run);
            if (cam_first[ci] < 0) cam_first[ci] = npair;
            pair_fill[(size_t)ci * p.n_cam + cj] = run;
            cam_ent[ci] += c;
            run += c;
            ++npair;
          }
        h_pptr.push_back(b->total_ent + run);
      }
      {
        const size_t ebase = h_ent.size();
        h_ent.resize(ebase + (size_t)n_ent);
        for (int64_t e = 0; e < n_ent; ++e) {
          const int a = ea[e], bb = ebq[e];
          const int cj = p.obs_cam[bb];
          const int slot = pair_fill[(size_t)p.obs_cam[a] * p.n_cam + cj]++;
          h_ent[ebase + slot] = make_int2(pos[a], obase + cnt_cam[cj] + pos[bb]);  // (LDS slot of T_a, W row of b)
        }
      }
      ts_ent += now_ms() - tsb;
      const int64_t n_keys = n_ent;
      // per-camera pair ranges (pairs are sorted by ci): cameras without pairs get an empty range
      cam_first[p.n_cam] = npair;
      for (int c = 0; c < p.n_cam; ++c) b->max_cam_ent = std::max(b->max_cam_ent, cam_ent[c]);
      for (int c = p.n_cam - 1; c >= 0; --c) if (cam_first[c] < 0) cam_first[c] = cam_first[c + 1];
      for (int c = 0; c <= p.n_cam; ++c) h_campair.push_back(cam_first[c]);
      for (int c = 0; c < p.n_cam; ++c) b->max_cam_pair = std::max(b->max_cam_pair, cam_first[c + 1] - cam_first[c]);
      s.n_pair = npair;
      b->total_ent += (int)n_keys;
      b->total_pair += npair;
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
      for (int c = 0; c < p.n_cam; ++c) {
        const bool has_res = cnt_cam[c + 1] > cnt_cam[c];
        unsigned char flag = 0;
        if (has_res && !counted[first[c]]) { flag = 1; counted[first[c]] = 1; }
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
        ++s.n_grp;
        any_shared = true;
      }
      total_grp += s.n_grp;
      b->max_grp = std::max(b->max_grp, s.n_grp);
    }
    b->total_cam += p.n_cam; b->total_ray += p.n_ray; b->total_obs += (int)p.n_obs; b->total_chunk += s.n_chunk + 1;
    b->max_cam = std::max(b->max_cam, p.n_cam); b->max_ray = std::max(b->max_ray, p.n_ray);
    b->max_chunk = std::max(b->max_chunk, s.n_chunk); b->max_pair = std::max(b->max_pair, s.n_pair);
    b->max_n = std::max(b->max_n, s.n);
    b->scenes.push_back(s);
  }
  (void)tot_ent;
  // LDS budget of the eval kernel bounds the camera count of a scene (160 KiB per workgroup)
  if (sizeof(double) * ((size_t)b->max_cam * (CBS + CDS + NC + 1) + 16) > 160 * 1024) { ptz_ba_batch_destroy(b); return PTZ_EUNSUPPORTED; }

  tc1 = now_ms();
  Dev& d = b->d;
  memset(&d, 0, sizeof(d));
  d.n_scene = n;
  int rc = PTZ_OK;
#define TRY(x) do { rc = (x); if (rc) { ptz_ba_batch_destroy(b); return rc; } } while (0)
  TRY(upload(b, b->scenes, &d.scene));
  TRY(upload(b, h_uv, &d.obs_uv));
  TRY(upload(b, h_cam, &d.obs_cam));
  TRY(upload(b, h_ray, &d.obs_ray));
  TRY(upload(b, h_rayptr, &d.ray_ptr));
  TRY(upload(b, h_camptr, &d.cam_ptr));
  TRY(upload(b, h_camobs, &d.cam_obs));
  TRY(upload(b, h_wpos, &d.wpos));
  TRY(upload(b, h_camray, &d.cam_ray));
  TRY(upload(b, h_pci, &d.pair_ci));
  TRY(upload(b, h_pcj, &d.pair_cj));
  TRY(upload(b, h_pptr, &d.pair_ptr));
  TRY(upload(b, h_campair, &d.cam_pair));
  TRY(upload(b, h_ent, &d.ent));
  TRY(upload(b, h_w, &d.ray_w));
  TRY(upload(b, h_o3uv, &d.o3_uv));
  TRY(upload(b, h_o3xyz, &d.o3_xyz));
  TRY(upload(b, h_o3cam, &d.o3_cam));
  d.tlw_stride = (size_t)n * 6;
  TRY(b->alloc(&d.tlw_x, 2 * d.tlw_stride));
  TRY(b->alloc(&b->tlw0, d.tlw_stride));
  TRY(b->alloc(&d.tlwblk, (size_t)n * TLWBLK));
  TRY(b->alloc(&d.tlwcand, (size_t)n * TLWBLK));
  TRY(b->alloc(&d.scale_t, (size_t)n * 6));
  TRY(b->alloc(&d.diag_t, (size_t)n * 6));
  TRY(b->alloc(&d.Ut, (size_t)n * 36));
  TRY(b->alloc(&d.gt, (size_t)n * 6));
  TRY(b->alloc(&d.dt, (size_t)n * 6));
  TRY(b->alloc(&d.Jc3, (size_t)b->total_o3 * 2 * NC));
  TRY(b->alloc(&d.Jt3, (size_t)b->total_o3 * 12));
  TRY(b->alloc(&d.r3, (size_t)b->total_o3 * 2));
  if (hipMemset(b->tlw0, 0, sizeof(double) * d.tlw_stride) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
  d.cam_stride = (size_t)b->total_cam * 15;
  d.ray_stride = (size_t)b->total_ray * 3;
  TRY(b->alloc(&d.cam_x, 2 * d.cam_stride));
  TRY(b->alloc(&d.ray_x, 2 * d.ray_stride));
  TRY(b->alloc(&b->cam0, d.cam_stride));
  TRY(b->alloc(&b->ray0, d.ray_stride));
  TRY(b->alloc(&d.camblk, (size_t)b->total_cam * CAMBLK));
  TRY(b->alloc(&d.candblk, (size_t)b->total_cam * CANDBLK));
  TRY(b->alloc(&d.scale_c, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.scale_r, (size_t)b->total_ray * 3));
  TRY(b->alloc(&d.U, (size_t)b->total_cam * NC * NC));
  TRY(b->alloc(&d.gc, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.costc, (size_t)b->total_cam));
  TRY(b->alloc(&d.diag_c, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.dc, (size_t)b->total_cam * NC));
  TRY(b->alloc(&d.V, (size_t)b->total_ray * 6));
  TRY(b->alloc(&d.gr, (size_t)b->total_ray * 3));
  TRY(b->alloc(&d.diag_r, (size_t)b->total_ray * 3));
  TRY(b->alloc(&d.E, (size_t)b->total_ray * 6));
  TRY(b->alloc(&d.z, (size_t)b->total_ray * 3));
  d.shared = any_shared ? 1 : 0;
  if (any_shared) {
    TRY(upload(b, h_grpptr, &d.grp_ptr));
    TRY(upload(b, h_grpmem, &d.grp_mem));
    TRY(upload(b, h_camflag, &d.cam_flag));
    TRY(b->alloc(&d.gfold, (size_t)b->total_cam * NC));
  }
  TRY(b->alloc(&d.W, (size_t)b->total_obs * (type == PTZ_BA_PTZRayFxfyDist ? 18 : 16)));  // room for the widest row stride
  TRY(b->alloc(&d.partial, (size_t)b->total_chunk * 2));
  TRY(b->alloc(&d.lm, (size_t)n));
  TRY(b->alloc(&d.active, (size_t)n));
  TRY(b->alloc(&d.ray_fail, (size_t)n));
  // reduced camera systems
  d.chol.count = n;
  d.chol.np = chol_padded_order(b->max_n);
  {
    std::vector<int> hn(n);
    for (int i = 0; i < n; ++i) hn[i] = b->scenes[i].n;
    const int* dn = nullptr;
    TRY(upload(b, hn, &dn));
    d.chol.n = dn;
  }
  TRY(b->alloc(&d.chol.A, (size_t)n * d.chol.np * d.chol.np));
  TRY(b->alloc(&d.chol.Ldiag, (size_t)n * (d.chol.np / CHOL_NB) * CHOL_NB * CHOL_NB));
  TRY(b->alloc(&d.chol.Dinv, (size_t)n * (d.chol.np / CHOL_NB) * 4 * 16 * 16));
  TRY(b->alloc(&d.chol.fail, (size_t)n));
  d.chol.active = d.active;
  TRY(b->alloc(&d.yc, (size_t)n * d.chol.np));
  tc2 = now_ms();
  // Tile-level structure of every reduced camera system: cameras that share a track couple their tiles, the T_l_w
  // block and the rhs row couple to everything; closed under the fill of the right-looking factorisation.
  if (!getenv("PTZ_BA_DENSE_CHOL")) {
    const int nt = d.chol.np / CHOL_NB;
    std::vector<unsigned char> hm((size_t)n * nt * nt, 0);
    for (int i = 0; i < n; ++i) {
      const SceneDev& sd = b->scenes[i];
      unsigned char* m = hm.data() + (size_t)i * nt * nt;
      auto tile_lo = [&](int cam) { return (cam * NC) / CHOL_NB; };
      auto tile_hi = [&](int cam) { return (cam * NC + NC - 1) / CHOL_NB; };
      for (int c = 0; c < sd.n_cam; ++c)
        for (int a = tile_lo(c); a <= tile_hi(c); ++a)
          for (int e = tile_lo(c); e <= a; ++e) m[a * nt + e] = 1;
      for (int p = 0; p < sd.n_pair; ++p) {
        const int ci = h_pci[sd.pair_off + p], cj = h_pcj[sd.pair_off + p];
        for (int a = tile_lo(ci); a <= tile_hi(ci); ++a)
          for (int e = tile_lo(cj); e <= tile_hi(cj); ++e) {
            if (a >= e) m[a * nt + e] = 1; else m[e * nt + a] = 1;
          }
      }
      // shared intrinsics: the fold puts a dense row / column at every group's representative (its last camera)
      if (any_shared) {
        const int* gp = h_grpptr.data() + sd.grp_off + i;
        for (int g = 0; g < sd.n_grp; ++g) {
          const int rep = h_grpmem[gp[g + 1] - 1];
          for (int a = tile_lo(rep); a <= tile_hi(rep); ++a) {
            for (int e = 0; e <= a; ++e) m[a * nt + e] = 1;
            for (int r = a; r < nt; ++r) m[r * nt + a] = 1;
          }
        }
      }
      // dense rows: the global block (if any) and the rhs row, index n_cam * NC .. n
      for (int a = (sd.n_cam * NC) / CHOL_NB; a < nt; ++a)
        for (int e = 0; e <= a; ++e) m[a * nt + e] = 1;
      for (int k = 0; k < nt; ++k)
        for (int x = k + 1; x < nt; ++x) {
          if (!m[x * nt + k]) continue;
          for (int y = k + 1; y <= x; ++y)
            if (m[y * nt + k]) m[x * nt + y] = 1;
        }
    }
    const unsigned char* dm = nullptr;
    TRY(upload(b, hm, &dm));
    d.chol.tmask = dm;
    // tiles outside the structure are never written again: zero everything once (the block may be a recycled one)
    if (hipMemset(d.chol.A, 0, sizeof(double) * (size_t)n * d.chol.np * d.chol.np) != hipSuccess) { ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
  }
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
  make_groups(b);
  b->stream = b->streams.empty() ? nullptr : b->streams[0];
  if (b->stream == nullptr || ptzpool::event_acquire(b->device, true, &b->ev0) != hipSuccess ||
      ptzpool::event_acquire(b->device, true, &b->ev1) != hipSuccess ||
      ptzpool::pinned_acquire(sizeof(int) * n, (void**)&b->h_active) != hipSuccess) {
    ptz_ba_batch_destroy(b);
    return PTZ_ENODEVICE;
  }
  // kernels that stage camera tables need > 64 KiB of dynamic LDS for large rigs
  const int eval_smem = (int)(sizeof(double) * ((size_t)b->max_cam * (CBS + CDS + (NC | 1)) + 16));
  const int lin_smem = (int)(sizeof(double) * (size_t)b->max_cam * CBS);
  const int schur_smem = (int)schur_lds_bytes(b->max_cam_obs, b->max_cam_ent, b->max_cam_pair, NC);
  if (schur_smem > 160 * 1024) { ptz_ba_batch_destroy(b); return PTZ_EUNSUPPORTED; }
  if (eval_smem > 160 * 1024 || lin_smem > 160 * 1024) { ptz_ba_batch_destroy(b); return PTZ_EUNSUPPORTED; }
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
      raise_cap((const void*)k_schur<T>);      \
      raise_cap((const void*)k_eval<T>);       \
      raise_cap((const void*)k_lin_ray<T>);
      PTZ_SET_ATTR(0) PTZ_SET_ATTR(1) PTZ_SET_ATTR(2) PTZ_SET_ATTR(3) PTZ_SET_ATTR(4) PTZ_SET_ATTR(5)
#undef PTZ_SET_ATTR
      if (!attr_ok) { (void)hipGetLastError(); ptz_ba_batch_destroy(b); return PTZ_ENODEVICE; }
      attr_done[o.device_id] = 1;
    }
  }
  tc3 = now_ms();
  if (dbg_t) fprintf(stderr, "[ptz_ba_create] host structure %.2f ms (observations %.2f, pair entries %.2f), uploads + allocations %.2f ms, mask + rest %.2f ms\n", tc1 - tc0, ts_obs, ts_ent, tc2 - tc1, tc3 - tc2);
  *out = b;
  return PTZ_OK;
}

int32_t ptz_ba_batch_set_state(ptz_ba_batch* b, const double* cam, const double* ray, const double* tlw)
{
  if (!b || !cam || !ray) return PTZ_EINVAL;
  PTZ_HIP_TRY(hipSetDevice(b->device));
  if (b->d.shared) {
    // a shared block starts from its first camera's values (intrinsics_param_.insert, ptzray_optimizer.cc:645-650)
    std::vector<double> c(cam, cam + 15 * (size_t)b->total_cam);
    for (int i = 0; i < b->total_cam; ++i) {
      const int f = b->first_of_group[i];
      if (f == i) continue;
      for (int k = 0; k < 15; ++k)
        if (k < 4 || k >= 10) c[15 * (size_t)i + k] = c[15 * (size_t)f + k];
    }
    PTZ_HIP_TRY(hipMemcpy(b->cam0, c.data(), sizeof(double) * 15 * b->total_cam, hipMemcpyHostToDevice));
  }
  else PTZ_HIP_TRY(hipMemcpy(b->cam0, cam, sizeof(double) * 15 * b->total_cam, hipMemcpyHostToDevice));
  PTZ_HIP_TRY(hipMemcpy(b->ray0, ray, sizeof(double) * 3 * b->total_ray, hipMemcpyHostToDevice));
  if (tlw) PTZ_HIP_TRY(hipMemcpy(b->tlw0, tlw, sizeof(double) * 6 * b->n_scene, hipMemcpyHostToDevice));
  else PTZ_HIP_TRY(hipMemset(b->tlw0, 0, sizeof(double) * 6 * b->n_scene));
  b->has_state = true;
  return PTZ_OK;
}

int32_t ptz_ba_batch_solve(ptz_ba_batch* b, ptz_lm_summary* summaries)
{
  if (!b || !b->has_state) return PTZ_EINVAL;
  PTZ_HIP_TRY(hipSetDevice(b->device));
  switch (b->type + 3 * b->has3d) {  // Dims<TYPE>
    case 0: return solve_impl<0>(b, summaries);
    case 1: return solve_impl<1>(b, summaries);
    case 2: return solve_impl<2>(b, summaries);
    case 3: return solve_impl<3>(b, summaries);
    case 4: return solve_impl<4>(b, summaries);
    default: return solve_impl<5>(b, summaries);
  }
}

int32_t ptz_ba_batch_get_state(ptz_ba_batch* b, double* cam, double* ray, double* tlw)
{
  if (!b) return PTZ_EINVAL;
  PTZ_HIP_TRY(hipSetDevice(b->device));
  std::vector<LmState> h(b->n_scene);
  PTZ_HIP_TRY(hipMemcpy(h.data(), b->d.lm, sizeof(LmState) * b->n_scene, hipMemcpyDeviceToHost));
  for (int i = 0; i < b->n_scene; ++i) {
    const SceneDev& s = b->scenes[i];
    const int cur = h[i].cur;
    if (cam) PTZ_HIP_TRY(hipMemcpy(cam + (size_t)s.cam_off * 15, b->d.cam_x + cur * b->d.cam_stride + (size_t)s.cam_off * 15,
                                   sizeof(double) * 15 * s.n_cam, hipMemcpyDeviceToHost));
    if (ray) PTZ_HIP_TRY(hipMemcpy(ray + (size_t)s.ray_off * 3, b->d.ray_x + cur * b->d.ray_stride + (size_t)s.ray_off * 3,
                                   sizeof(double) * 3 * s.n_ray, hipMemcpyDeviceToHost));
  }
  if (tlw)
    for (int i = 0; i < b->n_scene; ++i)
      PTZ_HIP_TRY(hipMemcpy(tlw + 6 * (size_t)i, b->d.tlw_x + h[i].cur * b->d.tlw_stride + 6 * (size_t)i, sizeof(double) * 6, hipMemcpyDeviceToHost));
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
  PTZ_HIP_TRY(hipSetDevice(b->device));
  hipLaunchKernelGGL(k_pix2ray, dim3(b->max_chunk, b->n_scene), dim3(RAY_BLOCK), 0, b->stream, b->d, b->cam0, b->ray0);
  PTZ_HIP_TRY(hipStreamSynchronize(b->stream));
  PTZ_HIP_TRY(hipGetLastError());
  return PTZ_OK;
}

int32_t ptz_ba_batch_linearize(ptz_ba_batch* b, int32_t index, double* cost, double* g_c, double* U, double* g_r, double* V, double* W)
{
  if (!b || !b->has_state || index < 0 || index >= b->n_scene) return PTZ_EINVAL;
  PTZ_HIP_TRY(hipSetDevice(b->device));
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
  switch (b->type + 3 * b->has3d) {  // Dims<TYPE>
    case 0: enqueue_linearize<0>(b); break;
    case 1: enqueue_linearize<1>(b); break;
    case 2: enqueue_linearize<2>(b); break;
    case 3: enqueue_linearize<3>(b); break;
    case 4: enqueue_linearize<4>(b); break;
    default: enqueue_linearize<5>(b); break;
  }
  PTZ_HIP_TRY(hipStreamSynchronize(st));
  PTZ_HIP_TRY(hipGetLastError());
  const SceneDev& s = b->scenes[index];
  if (cost) {
    std::vector<double> c(s.n_cam);
    PTZ_HIP_TRY(hipMemcpy(c.data(), d.costc + s.cam_off, sizeof(double) * s.n_cam, hipMemcpyDeviceToHost));
    double t = 0;
    for (double v : c) t += v;
    *cost = t;
  }
  if (g_c) PTZ_HIP_TRY(hipMemcpy(g_c, d.gc + (size_t)s.cam_off * NC, sizeof(double) * NC * s.n_cam, hipMemcpyDeviceToHost));
  if (U) PTZ_HIP_TRY(hipMemcpy(U, d.U + (size_t)s.cam_off * NC * NC, sizeof(double) * NC * NC * s.n_cam, hipMemcpyDeviceToHost));
  if (g_r) PTZ_HIP_TRY(hipMemcpy(g_r, d.gr + (size_t)s.ray_off * 3, sizeof(double) * 3 * s.n_ray, hipMemcpyDeviceToHost));
  if (V) {
    std::vector<double> v6((size_t)s.n_ray * 6);
    PTZ_HIP_TRY(hipMemcpy(v6.data(), d.V + (size_t)s.ray_off * 6, sizeof(double) * 6 * s.n_ray, hipMemcpyDeviceToHost));
    for (int j = 0; j < s.n_ray; ++j) {
      const double* p = &v6[(size_t)j * 6];
      double* q = V + (size_t)j * 9;
      q[0] = p[0]; q[1] = p[1]; q[2] = p[3]; q[3] = p[1]; q[4] = p[2]; q[5] = p[4]; q[6] = p[3]; q[7] = p[4]; q[8] = p[5];
    }
  }
  if (W) {  // rows are stored camera-major; return them in observation order
    const int NW = b->type == PTZ_BA_PTZRayFxfyDist ? 6 : NC - b->has3d;  // W carries the 2D-2D columns only
    const int ws = (NW * 3 + 1) & ~1;        // Dims<TYPE>::WS
    std::vector<double> rows((size_t)s.n_obs * ws);
    std::vector<int> wp(s.n_obs);
    PTZ_HIP_TRY(hipMemcpy(rows.data(), d.W + (size_t)s.obs_off * ws, sizeof(double) * ws * s.n_obs, hipMemcpyDeviceToHost));
    PTZ_HIP_TRY(hipMemcpy(wp.data(), d.wpos + s.obs_off, sizeof(int) * s.n_obs, hipMemcpyDeviceToHost));
    for (int a = 0; a < s.n_obs; ++a) memcpy(W + (size_t)a * NW * 3, &rows[(size_t)(wp[a] - s.obs_off) * ws], sizeof(double) * NW * 3);
  }
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

}  // extern "C"
