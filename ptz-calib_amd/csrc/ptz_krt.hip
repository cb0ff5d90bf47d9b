// ptz_krt.hip -- single-view 4-/5-parameter Levenberg-Marquardt, batched over queries (gfx950).
//
// Replaces KRTOptimizer::Add2d2dConstraints + Solve + CheckResults + ObtainRefinedCameraParams
// (src/core/krt_optimizer.cc:265-348, 385-404, 504-567) as driven by the relocalization loop
// (src/app/run_ptz_reloc.cc:68-118) and by RegisterNextImage (src/core/ptz_incremental_optimizer.cc:377-418).
// The reference builds one ceres::Problem per query (NumericDiffCostFunction over all 15 camera
// entries, DENSE_QR); here one 64-lane wave owns one query and runs the whole trust-region loop
// (Ceres 1.14 policy) without leaving the kernel:
//   lanes stride over the query's matches (16 B records: 2 x f32 reference pixel, 2 x f32 current pixel),
//   J^T J (4x4 or 5x5), J^T r and the cost are reduced with a fixed butterfly, every lane then solves the
//   damped normal equations redundantly in registers.  DENSE_QR on [J; D] and Cholesky on J^T J + D^2
//   give the same step up to round-off (the Jacobi-scaled 4-/5-column Jacobian is well conditioned).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#include "ptz_common.h"
#include "ptz_pool.h"
#include "ptz_factor.h"

namespace ptz {
namespace {

struct KrtOpt {
  int max_num_iterations, max_consecutive_invalid, jacobi_scaling;
  double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance, max_reproj_error;
  int lanes_per_query;  // host side only (launch_krt)
};

// 15-vector index of free parameter k: F {0,4,5,6}, FDist {0,4,5,6,10}, Fxfy {0,1,4,5,6}, FxfyDist {0,1,4,5,6,10}
template <int KTYPE> struct KFree {
  static __device__ __forceinline__ int at(int k)
  {
    constexpr int ROT0 = KrtDims<KTYPE>::ROT0;
    return k < ROT0 ? k : (k < ROT0 + 3 ? 4 + (k - ROT0) : 10);
  }
};

// in-register Cholesky solve of an NF x NF SPD system (row-major full storage); false if not SPD
template <int NF>
__device__ __forceinline__ bool spd_solve(double* A, double* b)
{
  double inv[NF];  // 1 / L_jj
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    double d = A[j * NF + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j * NF + k] * A[j * NF + k];
    if (!(d > 0.0)) return false;
    // (round 6) ONE reciprocal per column -- 1 / sqrt(d) to working precision (rsq + two Newton steps) -- instead of a square root and
    // 2 (NF - 1 - j) + 2 IEEE divisions by it: a division is ~18 instructions on this chip and the solve had twenty of them, every
    // lane of the group its own copy.  The factor and the solution differ from the divided form in the last bits (the step is held
    // to the oracle's QR step at 1e-6 either way).
    double rs = __builtin_amdgcn_rsq(d);
    rs = rs * (1.5 - 0.5 * d * rs * rs);
    rs = rs * (1.5 - 0.5 * d * rs * rs);
    inv[j] = rs;
    A[j * NF + j] = d * rs;
#pragma unroll
    for (int i = j + 1; i < NF; ++i) {
      double v = A[i * NF + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= A[i * NF + k] * A[j * NF + k];
      A[i * NF + j] = v * rs;
    }
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= A[i * NF + k] * b[k];
    b[i] = v * inv[i];
  }
#pragma unroll
  for (int i = NF - 1; i >= 0; --i) {
    double v = b[i];
#pragma unroll
    for (int k = i + 1; k < NF; ++k) v -= A[k * NF + i] * b[k];
    b[i] = v * inv[i];
  }
  return true;
}

template <int KTYPE>
struct MatchEval {
  // constant part of a match: unit ray of the reference pixel in the local frame (krt_optimizer.cc:31-33,
  // 89-104) and the border guard of the distortion variant (:97-101)
  static __device__ __forceinline__ void ray1(const double* kref, const double* dref, float u1, float v1, double r[3], bool& skip)
  {
    double u = u1, v = v1;
    skip = false;
    if (KTYPE & 1) {
      float ou, ov;
      undistort_point(kref[0], kref[1], kref[2], kref[3], dref, u1, v1, ou, ov);
      skip = (ou < 0 || ou >= kref[2] * 2 || ov < 0 || ov >= kref[3] * 2);
      u = ou; v = ov;
    }
    const double X0 = (u - kref[2]) / kref[0], X1 = (v - kref[3]) / kref[1];
    const double n = sqrt(X0 * X0 + X1 * X1 + 1.0);
    r[0] = X0 / n; r[1] = X1 / n; r[2] = 1.0 / n;
  }
};

// Lanes per query.  G = 64 (a wave per query) is the latency form: a registration attempt of the incremental pipeline or a
// handful of queries are as fast as they can be.  G = 16 (four queries per wave) is the throughput form for launches of
// thousands of queries of a few hundred matches each: with 128 matches a wave of 64 has two matches per lane and then spends
// as long on its 15 six-step reductions and on 64 redundant copies of one 4 x 4 solve as on the matches; 16 lanes take eight
// matches each, reduce in four steps, and a wave's redundant solves serve four queries.  The two forms sum in different
// orders: a query's bits depend on the form, never on its neighbours in the launch (ptz_krt_solve_batch picks the form from
// the launch size alone, krt_group_size()).
// (the butterfly v += v[lane ^ off], off = G / 2 .. 1, with the partners fetched by v_permlane32/16_swap and DPP instead of
//  ds_bpermute -- 21 sums of four to six steps per linearisation: the same partners, the same sums, the same bits)
template <int G> __device__ __forceinline__ double group_sum(double v)
{
  static_assert(G == 64 || G == 16, "a wave or a DPP row of lanes per query");
  if (G == 64) return wave_sum(v);
  v += lane_xor_dpp<8>(v);
  v += lane_xor_dpp<4>(v);
  v += lane_xor_dpp<2>(v);
  v += lane_xor_dpp<1>(v);
  return v;
}
template <int G> struct KrtCache { static constexpr int N = G == 64 ? 256 : 128; };  // matches per query whose constant part is cached
// P3: the queries also carry 2D-3D constraints (KRTOptimizer::Add2d3dConstraints, krt_optimizer.cc:350-383): world points,
// moved into the local frame of the reference camera as the reference does (:357-362), one residual block each.
#ifndef PTZ_KRT_OCC_DIST
#define PTZ_KRT_OCC_DIST 2
#endif
#ifndef PTZ_KRT_OCC_PLAIN
#define PTZ_KRT_OCC_PLAIN 2
#endif
template <int KTYPE, bool P3, int G>
__global__ __launch_bounds__(256, (KTYPE & 1) ? PTZ_KRT_OCC_DIST : PTZ_KRT_OCC_PLAIN) void k_krt(int n_query, const long long* __restrict__ match_ptr, const float2* __restrict__ uv_ref,
                                             const float2* __restrict__ uv_cur, const long long* __restrict__ point_ptr,
                                             const float2* __restrict__ pt_uv, const double* __restrict__ pt_xyz,
                                             const double* __restrict__ cam_ref, double* __restrict__ cam_cur, KrtOpt o,
                                             ptz_lm_summary* __restrict__ summ, int* __restrict__ accepted,
                                             const long long* __restrict__ cur_delta)
{
  constexpr int NF = KrtDims<KTYPE>::NF;
  constexpr int NH = NF * (NF + 1) / 2;
  constexpr int QPB = 256 / G, KRT_CACHE = KrtCache<G>::N;  // queries per workgroup
  const int q = blockIdx.x * QPB + (int)threadIdx.x / G;
  if (q >= n_query) return;
  const int lane = threadIdx.x % G;
  // cur_delta = nullptr: the queries' matches one after the other (CSR offsets).  Otherwise (ptz_krt_solve_attempts: matches of
  // resident tables) match_ptr holds a [begin, end) PAIR per query, relative to uv_ref, and the query's current-image pixels
  // start cur_delta[q] elements further on in uv_cur than its reference pixels in uv_ref (the tables are separate allocations).
  const long long m0 = cur_delta ? match_ptr[2 * q] : match_ptr[q], m1 = cur_delta ? match_ptr[2 * q + 1] : match_ptr[q + 1];
  const int M = (int)(m1 - m0);
  if (cur_delta) uv_cur += cur_delta[q];
  const long long p0 = P3 ? point_ptr[q] : 0;
  const int NP = P3 ? (int)(point_ptr[q + 1] - p0) : 0;
  // ---- world -> local frame of the reference camera (krt_optimizer.cc:269-284)
  double ref[15], x[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) { ref[k] = cam_ref[(size_t)q * 15 + k]; x[k] = cam_cur[(size_t)q * 15 + k]; }
  double Rref[9], Rcur[9], RrefT[9], Rl[9];
  rodrigues(ref + 4, Rref);
  rodrigues(x + 4, Rcur);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) RrefT[3 * i + j] = Rref[3 * j + i];
  mat3_mul(Rcur, RrefT, Rl);
  {
    double rv[3];
    rodrigues_inv(Rl, rv);
    const double t0 = Rl[0] * ref[7] + Rl[1] * ref[8] + Rl[2] * ref[9];
    const double t1 = Rl[3] * ref[7] + Rl[4] * ref[8] + Rl[5] * ref[9];
    const double t2 = Rl[6] * ref[7] + Rl[7] * ref[8] + Rl[8] * ref[9];
    x[4] = rv[0]; x[5] = rv[1]; x[6] = rv[2];
    x[7] = -t0 + x[7]; x[8] = -t1 + x[8]; x[9] = -t2 + x[9];
  }
  const double kref[4] = {ref[0], ref[1], ref[2], ref[3]};
  const double dref[5] = {ref[10], ref[11], ref[12], ref[13], ref[14]};
  // The unit ray of a reference pixel (and, with distortion, its iterative undistortion) does not depend on the parameters
  // being optimised: computed once per match, kept in a group-private LDS strip for the first KRT_CACHE matches of the query
  // (the reference recomputes it in every functor call, krt_optimizer.cc:31-33, 89-104).  A unit ray has r[2] = 1 / n > 0:
  // r[2] = 0 marks a match the border guard skips.
  __shared__ double ray_cache[QPB][KRT_CACHE][3];
  double (*rc)[3] = ray_cache[threadIdx.x / G];
  for (int m = lane; m < M && m < KRT_CACHE; m += G) {
    const float2 a = uv_ref[m0 + m];
    double r1[3];
    bool skip;
    MatchEval<KTYPE>::ray1(kref, dref, a.x, a.y, r1, skip);
    rc[m][0] = r1[0]; rc[m][1] = r1[1]; rc[m][2] = skip ? 0.0 : r1[2];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): every lane reads back only what it wrote
  auto local_point = [&](int i, double Xl[3]) {  // R_local_world X_w + t_local_world (krt_optimizer.cc:357-362)
    const double* X = pt_xyz + 3 * (p0 + i);
    Xl[0] = Rref[0] * X[0] + Rref[1] * X[1] + Rref[2] * X[2] + ref[7];
    Xl[1] = Rref[3] * X[0] + Rref[4] * X[1] + Rref[5] * X[2] + ref[8];
    Xl[2] = Rref[6] * X[0] + Rref[7] * X[1] + Rref[8] * X[2] + ref[9];
  };
  auto match_ray = [&](int m, double r1[3], bool& skip) {
    if (m < KRT_CACHE) { r1[0] = rc[m][0]; r1[1] = rc[m][1]; r1[2] = rc[m][2]; skip = r1[2] == 0.0; }
    else { const float2 a = uv_ref[m0 + m]; MatchEval<KTYPE>::ray1(kref, dref, a.x, a.y, r1, skip); }
  };

  // residual-only pass at a camera vector.  exact: the 2D-2D quotients as IEEE divisions (the final cost's evaluation, below)
  auto eval_cost_impl = [&](const double* c, auto exact_tag) -> double {
    constexpr bool EXACT = decltype(exact_tag)::value;
    double R[9];
    rodrigues(c + 4, R);
    double cost = 0;
    for (int m = lane; m < M; m += G) {
      const float2 bq = uv_cur[m0 + m];
      double r1[3], res[2], J[2][NF];
      bool skip;
      match_ray(m, r1, skip);
      krt_eval<KTYPE, false, EXACT>(R, nullptr, c[0], (KTYPE & 2) ? c[1] : c[0], c[2], c[3], c + 10, r1, skip, bq.x, bq.y, res, J);
      cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
    }
    if (P3) {
      for (int i = lane; i < NP; i += G) {
        const float2 bq = pt_uv[p0 + i];
        double Xl[3], res[2], J[2][NF];
        local_point(i, Xl);
        krt_eval_2d3d<KTYPE, false>(R, nullptr, c[0], (KTYPE & 2) ? c[1] : c[0], c[2], c[3], c + 10, c + 7, Xl, bq.x, bq.y, res, J);
        cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
      }
    }
    return group_sum<G>(cost);
  };
  auto eval_cost = [&](const double* c) -> double { return eval_cost_impl(c, std::false_type{}); };
  // full linearisation: H = J^T J (packed lower), g = J^T r, cost
  double H[NH], g[NF];
  auto linearize = [&](const double* c) -> double {
    double R[9], Jl[9];
    rodrigues(c + 4, R);
    so3_left_jacobian(c + 4, Jl);
    double cost = 0;
#pragma unroll
    for (int k = 0; k < NH; ++k) H[k] = 0;
#pragma unroll
    for (int k = 0; k < NF; ++k) g[k] = 0;
    for (int m = lane; m < M; m += G) {
      const float2 bq = uv_cur[m0 + m];
      double r1[3], res[2], J[2][NF];
      bool skip;
      match_ray(m, r1, skip);
      krt_eval<KTYPE, true>(R, Jl, c[0], (KTYPE & 2) ? c[1] : c[0], c[2], c[3], c + 10, r1, skip, bq.x, bq.y, res, J);
      cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
      int e = 0;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        g[k] += J[0][k] * res[0] + J[1][k] * res[1];
#pragma unroll
        for (int l = 0; l <= k; ++l) H[e++] += J[0][k] * J[0][l] + J[1][k] * J[1][l];
      }
    }
    if (P3) {
      for (int i = lane; i < NP; i += G) {
        const float2 bq = pt_uv[p0 + i];
        double Xl[3], res[2], J[2][NF];
        local_point(i, Xl);
        krt_eval_2d3d<KTYPE, true>(R, Jl, c[0], (KTYPE & 2) ? c[1] : c[0], c[2], c[3], c + 10, c + 7, Xl, bq.x, bq.y, res, J);
        cost += 0.5 * (res[0] * res[0] + res[1] * res[1]);
        int e = 0;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          g[k] += J[0][k] * res[0] + J[1][k] * res[1];
#pragma unroll
          for (int l = 0; l <= k; ++l) H[e++] += J[0][k] * J[0][l] + J[1][k] * J[1][l];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) H[k] = group_sum<G>(H[k]);
#pragma unroll
    for (int k = 0; k < NF; ++k) g[k] = group_sum<G>(g[k]);
    return group_sum<G>(cost);
  };
  auto norm15 = [&](const double* c) -> double {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) s += c[k] * c[k];
    return sqrt(s);
  };

  // ---- [Ceres 1.14] TrustRegionMinimizer, LevenbergMarquardtStrategy
  double radius = o.initial_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double x_cost = linearize(x);
  double scale[NF], diag[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) scale[k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[k * (k + 1) / 2 + k])) : 1.0;
  double x_norm = norm15(x);
  double grad_max = 0;
#pragma unroll
  for (int k = 0; k < NF; ++k) grad_max = fmax(grad_max, fabs(g[k]));
  const double initial_cost = x_cost;
  double final_cost = x_cost, it_cost = x_cost;
  int iteration = 0, n_summaries = 0, termination = PTZ_NO_CONVERGENCE;
  int n_succ = 0, n_unsucc = 0, n_steps = 0, n_solves = 0, n_jac = 1, consec_invalid = 0;
  bool step_ok = true;
  for (;;) {
    if (step_ok) ++n_succ; else ++n_unsucc;
    if (it_cost < final_cost) final_cost = it_cost;
    ++n_summaries;
    if (iteration >= o.max_num_iterations) { termination = PTZ_NO_CONVERGENCE; break; }
    if (step_ok && grad_max <= o.gradient_tolerance) { termination = PTZ_CONVERGENCE; break; }
    if (radius <= o.min_radius) { termination = PTZ_CONVERGENCE; break; }
    ++iteration; ++n_steps;
    step_ok = false;
    it_cost = x_cost;
    // scaled normal equations (J_s = J diag(scale))
    double A[NF * NF], b[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      b[k] = g[k] * scale[k];
#pragma unroll
      for (int l = 0; l <= k; ++l) {
        const double v = H[k * (k + 1) / 2 + l] * scale[k] * scale[l];
        A[k * NF + l] = v; A[l * NF + k] = v;
      }
    }
    if (!reuse_diagonal) {
#pragma unroll
      for (int k = 0; k < NF; ++k) diag[k] = fmin(fmax(A[k * NF + k], o.min_lm_diagonal), o.max_lm_diagonal);
    }
    // (the scaled system is not kept beside the solve's working copy: the model cost change below takes its entries from the packed
    //  H and the scales again -- the same products, H_kl s_k s_l and g_k s_k -- which frees 30 doubles of registers across the solve)
#pragma unroll
    for (int k = 0; k < NF; ++k) { const double D = sqrt(diag[k] / radius); A[k * NF + k] += D * D; }
    const bool solved = spd_solve<NF>(A, b);
    ++n_solves;
    reuse_diagonal = true;
    double step[NF], mcc = 0;
    bool valid = solved;
#pragma unroll
    for (int k = 0; k < NF; ++k) { step[k] = -b[k]; valid = valid && isfinite(step[k]); }
    if (valid) {
      // -(J step)^T (r + J step / 2) = -(step^T g + step^T H step / 2)
      double sg = 0, shs = 0;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        sg += step[k] * (g[k] * scale[k]);
        double t = 0;
#pragma unroll
        for (int l = 0; l < NF; ++l) {
          const int hi = k > l ? k : l, lo = k > l ? l : k;
          t += (H[hi * (hi + 1) / 2 + lo] * scale[hi] * scale[lo]) * step[l];
        }
        shs += step[k] * t;
      }
      mcc = -(sg + 0.5 * shs);
      valid = mcc > 0.0;
    }
    if (!valid) {
      if (++consec_invalid >= o.max_consecutive_invalid) { termination = PTZ_FAILURE; break; }
      radius *= 0.5;
      reuse_diagonal = false;
      continue;
    }
    consec_invalid = 0;
    double xc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) xc[k] = x[k];
#pragma unroll
    for (int k = 0; k < NF; ++k) xc[KFree<KTYPE>::at(k)] += step[k] * scale[k];
    double cand = eval_cost(xc);
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    double dn = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) dn += (x[k] - xc[k]) * (x[k] - xc[k]);
    if (sqrt(dn) <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { termination = PTZ_CONVERGENCE; break; }
    const double cost_change = x_cost - cand;
    if (fabs(cost_change) <= o.function_tolerance * x_cost) { termination = PTZ_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > o.min_relative_decrease) {
#pragma unroll
      for (int k = 0; k < 15; ++k) x[k] = xc[k];
      x_norm = norm15(x);
      x_cost = linearize(x);
      ++n_jac;
      grad_max = 0;
#pragma unroll
      for (int k = 0; k < NF; ++k) grad_max = fmax(grad_max, fabs(g[k]));
      step_ok = true;
      it_cost = x_cost;
      const double t = 2.0 * rho - 1.0;
      radius = fmin(o.max_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
      reuse_diagonal = false;
    }
    else {
      it_cost = cand;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  // ---- CheckResults (krt_optimizer.cc:504-533) + ObtainRefinedCameraParams (:535-567)
  // The accept test compares the final reprojection error with max_reproj_error: the final cost is evaluated once more at the final
  // point with the reference functor's own divisions (the iterations use reciprocal products, ptz_factor.h); one residual pass of the
  // ~16 a query takes.  (Ceres' final_cost is the cost of the point it returns.)
  if (termination == PTZ_CONVERGENCE) final_cost = eval_cost_impl(x, std::true_type{});
  // [Ceres 1.14] a non-finite cost at the initial point (a non-finite pixel, a ray in the image plane) is FAILURE before any iteration:
  // nothing is refined, the summary carries zeros (the oracle's lm_minimize, IterationZero).  The loop above has then spent its
  // max_consecutive_invalid steps on not-a-numbers without moving x (a branch in front of it costs every query 136 bytes of scratch);
  // its bookkeeping is replaced here.
  const bool bad0 = !isfinite(initial_cost);
  if (bad0) { termination = PTZ_FAILURE; n_summaries = 1; n_steps = 0; n_succ = 0; n_unsucc = 0; n_solves = 0; n_jac = 1; final_cost = 0.0; }
  const int num_residuals = 2 * (M + NP);
  const double final_reproj = sqrt(2.0) * sqrt((2 * final_cost) / num_residuals);
  bool ok = (termination == PTZ_CONVERGENCE) && !(final_reproj >= o.max_reproj_error);
  {
    const double fov_x = atan(x[2] / x[0]) * 2 * 180 / M_PI, fov_y = atan(x[3] / x[1]) * 2 * 180 / M_PI;
    if (fov_x < 0 || fov_x > 170 || fov_y < 0 || fov_y > 170) ok = false;
  }
  if (lane == 0) {
    ptz_lm_summary s;
    s.termination_type = termination;
    s.num_iterations = n_summaries - 1;
    s.num_lm_steps = n_steps;
    s.num_successful_steps = n_succ;
    s.num_unsuccessful_steps = n_unsucc;
    s.num_residuals = num_residuals;
    s.num_linear_solves = n_solves;
    s.num_jacobian_evals = n_jac;
    s.initial_cost = bad0 ? 0.0 : initial_cost;
    s.final_cost = final_cost;
    s.final_radius = bad0 ? o.initial_radius : radius;
    s.final_gradient_max_norm = bad0 ? 0.0 : grad_max;
    summ[q] = s;
    accepted[q] = ok ? 1 : 0;
    if (ok) {
      if (!(KTYPE & 2)) x[1] = x[0];  // fx = fy for F / FDist (krt_optimizer.cc:540-551)
      double Rloc[9], Rw[9], rv[3];
      rodrigues(x + 4, Rloc);
      mat3_mul(Rloc, Rref, Rw);
      rodrigues_inv(Rw, rv);
      const double t0 = Rloc[0] * ref[7] + Rloc[1] * ref[8] + Rloc[2] * ref[9];
      const double t1 = Rloc[3] * ref[7] + Rloc[4] * ref[8] + Rloc[5] * ref[9];
      const double t2 = Rloc[6] * ref[7] + Rloc[7] * ref[8] + Rloc[8] * ref[9];
      double* out = cam_cur + (size_t)q * 15;
#pragma unroll
      for (int k = 0; k < 15; ++k) out[k] = x[k];
      out[4] = rv[0]; out[5] = rv[1]; out[6] = rv[2];
      out[7] = t0 + x[7]; out[8] = t1 + x[8]; out[9] = t2 + x[9];
    }
  }
}

KrtOpt make_krt_opt(const ptz_lm_options& o, double max_reproj_error)
{
  KrtOpt ko;
  ko.max_num_iterations = o.max_num_iterations;
  ko.max_consecutive_invalid = o.max_num_consecutive_invalid_steps;
  ko.jacobi_scaling = o.jacobi_scaling;
  ko.initial_radius = o.initial_trust_region_radius;
  ko.max_radius = o.max_trust_region_radius;
  ko.min_radius = o.min_trust_region_radius;
  ko.min_relative_decrease = o.min_relative_decrease;
  ko.min_lm_diagonal = o.min_lm_diagonal;
  ko.max_lm_diagonal = o.max_lm_diagonal;
  ko.function_tolerance = o.function_tolerance;
  ko.gradient_tolerance = o.gradient_tolerance;
  ko.parameter_tolerance = o.parameter_tolerance;
  ko.max_reproj_error = max_reproj_error;
  ko.lanes_per_query = o.krt_lanes_per_query;
  return ko;
}

// lanes per query of a launch of n_query queries (see k_krt): ptz_lm_options::krt_lanes_per_query, else PTZ_KRT_GROUP, else by launch size
// (the throughput form only once the launch fills the chip -- 256 compute units x 8 waves x 4 queries: below that a launch
// lasts as long as its slowest query, which sixteen lanes make four times slower)
inline int krt_group_size(int n_query, int requested)
{
  if (requested == 16 || requested == 64) return requested;
  if (const char* e = getenv("PTZ_KRT_GROUP")) { const int g = atoi(e); if (g == 16 || g == 64) return g; }
  return n_query >= 16384 ? 16 : 64;
}

// one launch over device-resident queries (all pointers are device pointers; d_pptr = nullptr: no 2D-3D constraints)
void launch_krt(int n_query, const long long* d_ptr, const float2* d_ref, const float2* d_cur, const long long* d_pptr,
                const float2* d_puv, const double* d_pxyz, const double* d_cref, double* d_ccur, int factor_type, const KrtOpt& ko,
                ptz_lm_summary* d_sum, int* d_acc, hipStream_t st, const long long* d_delta = nullptr)
{
  const int G = krt_group_size(n_query, ko.lanes_per_query);
  const int qpb = 256 / G;
  const dim3 grid((n_query + qpb - 1) / qpb), block(256);
  const bool p3 = d_pptr != nullptr;
#define PTZ_KRT_LAUNCH(T, P)                                                                                                   \
  do {                                                                                                                         \
    if (G == 16) hipLaunchKernelGGL((k_krt<T, P, 16>), grid, block, 0, st, n_query, d_ptr, d_ref, d_cur, d_pptr, d_puv, d_pxyz, \
                                    d_cref, d_ccur, ko, d_sum, d_acc, d_delta);                                                \
    else hipLaunchKernelGGL((k_krt<T, P, 64>), grid, block, 0, st, n_query, d_ptr, d_ref, d_cur, d_pptr, d_puv, d_pxyz, d_cref, \
                            d_ccur, ko, d_sum, d_acc, d_delta);                                                                \
  } while (0)
  switch (factor_type * 2 + (p3 ? 1 : 0)) {
    case 0: PTZ_KRT_LAUNCH(0, false); break;
    case 1: PTZ_KRT_LAUNCH(0, true); break;
    case 2: PTZ_KRT_LAUNCH(1, false); break;
    case 3: PTZ_KRT_LAUNCH(1, true); break;
    case 4: PTZ_KRT_LAUNCH(2, false); break;
    case 5: PTZ_KRT_LAUNCH(2, true); break;
    case 6: PTZ_KRT_LAUNCH(3, false); break;
    default: PTZ_KRT_LAUNCH(3, true); break;
  }
#undef PTZ_KRT_LAUNCH
}

}  // namespace
}  // namespace ptz

using namespace ptz;

extern "C" int32_t ptz_krt_solve_batch_device(int32_t n_query, const int64_t* d_match_ptr, const float* d_uv_ref, const float* d_uv_cur,
                                              const int64_t* d_point_ptr, const float* d_pts2d, const double* d_pts3d,
                                              const double* d_cam_ref, double* d_cam_cur, int32_t factor_type,
                                              double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* d_summaries,
                                              int32_t* d_accepted, void* hip_stream)
{
  if (n_query <= 0 || !d_match_ptr || !d_uv_ref || !d_uv_cur || !d_cam_ref || !d_cam_cur || !d_summaries || !d_accepted) return PTZ_EINVAL;
  if (d_point_ptr && (!d_pts2d || !d_pts3d)) return PTZ_EINVAL;
  if (factor_type < PTZ_KRT_F || factor_type > PTZ_KRT_FxfyDist) return PTZ_EUNSUPPORTED;
  ptz_lm_options o;
  if (opt) o = *opt; else ptz_lm_options_default(&o);
  clear_stale_error(__func__);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PTZ_ENODEVICE;
  // The launch goes to the device that owns the caller's buffers, on the caller's stream: a caller holding tensors and a
  // stream on GPU 1 must not need to repeat that in opt->device_id.  An explicit, different device_id is a contradiction.
  int device = -1;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_cam_cur) != hipSuccess) { (void)hipGetLastError(); return PTZ_EINVAL; }
    device = attr.device;
    if (hip_stream) {
      hipDevice_t sdev = -1;
      if (hipStreamGetDevice((hipStream_t)hip_stream, &sdev) != hipSuccess) { (void)hipGetLastError(); return PTZ_EINVAL; }
      if ((int)sdev != device) return PTZ_EINVAL;  // stream and buffers live on different devices
    }
    if (device < 0 || device >= ndev) return PTZ_EINVAL;
    if (opt && opt->device_id != 0 && opt->device_id != device) return PTZ_EINVAL;
  }
  PTZ_DEVICE_GUARD(device);
  launch_krt(n_query, (const long long*)d_match_ptr, (const float2*)d_uv_ref, (const float2*)d_uv_cur, (const long long*)d_point_ptr,
             (const float2*)d_pts2d, d_pts3d, d_cam_ref, d_cam_cur, factor_type, make_krt_opt(o, max_reproj_error), d_summaries, d_accepted,
             (hipStream_t)hip_stream);
  PTZ_HIP_TRY(hipGetLastError());
  return PTZ_OK;
}

extern "C" int32_t ptz_krt_solve_batch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                                       const double* cam_ref, double* cam_cur, int32_t factor_type, double max_reproj_error,
                                       const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  return ptz_krt_solve_batch_2d3d(n_query, match_ptr, uv_ref, uv_cur, nullptr, nullptr, nullptr, cam_ref, cam_cur, factor_type,
                                  max_reproj_error, opt, summaries, accepted, device_ms);
}

extern "C" int32_t ptz_krt_solve_batch_2d3d(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                                            const int64_t* point_ptr, const float* pts2d, const double* pts3d,
                                            const double* cam_ref, double* cam_cur, int32_t factor_type,
                                            double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries,
                                            int32_t* accepted, double* device_ms)
{
  if (n_query <= 0 || !match_ptr || !uv_ref || !uv_cur || !cam_ref || !cam_cur || !summaries || !accepted) return PTZ_EINVAL;
  const bool p3 = point_ptr != nullptr;
  if (p3 && (!pts2d || !pts3d)) return PTZ_EINVAL;
  if (p3)
    for (int q = 0; q < n_query; ++q)
      if (point_ptr[q + 1] < point_ptr[q]) return PTZ_EINVAL;
  const int64_t np = p3 ? point_ptr[n_query] : 0;
  if (factor_type < PTZ_KRT_F || factor_type > PTZ_KRT_FxfyDist) return PTZ_EUNSUPPORTED;
  ptz_lm_options o;
  if (opt) o = *opt; else ptz_lm_options_default(&o);
  for (int q = 0; q < n_query; ++q)
    if (match_ptr[q + 1] < match_ptr[q]) return PTZ_EINVAL;
  clear_stale_error(__func__);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= o.device_id) return PTZ_ENODEVICE;
  PTZ_DEVICE_GUARD(o.device_id);
  const int64_t nm = match_ptr[n_query];
  // one pooled device block for everything the launch touches: [inputs | cam_cur (in/out) | summaries | accepted]
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_ptr = 0, o_ref = o_ptr + up(sizeof(long long) * (n_query + 1)), o_cur = o_ref + up(sizeof(float2) * (nm > 0 ? nm : 1)),
               o_cref = o_cur + up(sizeof(float2) * (nm > 0 ? nm : 1)), o_pptr = o_cref + up(sizeof(double) * 15 * n_query),
               o_puv = o_pptr + up(sizeof(long long) * (n_query + 1)), o_pxyz = o_puv + up(sizeof(float2) * (np > 0 ? np : 1)),
               o_ccur = o_pxyz + up(sizeof(double) * 3 * (np > 0 ? np : 1)), o_sum = o_ccur + up(sizeof(double) * 15 * n_query),
               o_acc = o_sum + up(sizeof(ptz_lm_summary) * n_query), total = o_acc + up(sizeof(int) * n_query);
  struct Held {
    int dev; char* base = nullptr; void* pinned = nullptr; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Held()
    {
      if (st) (void)stream_wait(st);
      ptzpool::dev_release(dev, base);
      ptzpool::pinned_release(pinned);
      ptzpool::stream_release(dev, st);
      ptzpool::event_release(dev, true, e0);
      ptzpool::event_release(dev, true, e1);
    }
  } h;
  h.dev = o.device_id;
  if (ptzpool::dev_acquire(h.dev, total, (void**)&h.base) != hipSuccess) return PTZ_ENOMEM;
  PTZ_HIP_TRY(ptzpool::stream_acquire(h.dev, &h.st));
  PTZ_HIP_TRY(ptzpool::event_acquire(h.dev, true, &h.e0));
  PTZ_HIP_TRY(ptzpool::event_acquire(h.dev, true, &h.e1));
  long long* d_ptr = (long long*)(h.base + o_ptr);
  float2 *d_ref = (float2*)(h.base + o_ref), *d_cur = (float2*)(h.base + o_cur);
  double *d_cref = (double*)(h.base + o_cref), *d_ccur = (double*)(h.base + o_ccur);
  ptz_lm_summary* d_sum = (ptz_lm_summary*)(h.base + o_sum);
  int* d_acc = (int*)(h.base + o_acc);
  long long* d_pptr = p3 ? (long long*)(h.base + o_pptr) : nullptr;
  float2* d_puv = (float2*)(h.base + o_puv);
  double* d_pxyz = (double*)(h.base + o_pxyz);
  // Small launches (a registration, a handful of queries) go through one pinned staging buffer: one copy in, one copy out,
  // instead of eight synchronous-by-nature pageable copies of 10-15 us each.  Large ones keep the per-array copies.
  const bool staged = total <= ((size_t)64 << 20) && ptzpool::pinned_acquire(total, &h.pinned) == hipSuccess;
  if (!staged) (void)hipGetLastError();
  if (staged) {
    char* ps = (char*)h.pinned;
    memcpy(ps + o_ptr, match_ptr, sizeof(long long) * (n_query + 1));
    if (nm > 0) { memcpy(ps + o_ref, uv_ref, sizeof(float2) * nm); memcpy(ps + o_cur, uv_cur, sizeof(float2) * nm); }
    memcpy(ps + o_cref, cam_ref, sizeof(double) * 15 * n_query);
    memcpy(ps + o_ccur, cam_cur, sizeof(double) * 15 * n_query);
    if (p3) {
      memcpy(ps + o_pptr, point_ptr, sizeof(long long) * (n_query + 1));
      if (np > 0) { memcpy(ps + o_puv, pts2d, sizeof(float2) * np); memcpy(ps + o_pxyz, pts3d, sizeof(double) * 3 * np); }
    }
    PTZ_HIP_TRY(hipMemcpyAsync(h.base, ps, o_sum, hipMemcpyHostToDevice, h.st));
  }
  else {
    if (p3) {
      PTZ_HIP_TRY(hipMemcpyAsync(d_pptr, point_ptr, sizeof(long long) * (n_query + 1), hipMemcpyHostToDevice, h.st));
      if (np > 0) {
        PTZ_HIP_TRY(hipMemcpyAsync(d_puv, pts2d, sizeof(float2) * np, hipMemcpyHostToDevice, h.st));
        PTZ_HIP_TRY(hipMemcpyAsync(d_pxyz, pts3d, sizeof(double) * 3 * np, hipMemcpyHostToDevice, h.st));
      }
    }
    PTZ_HIP_TRY(hipMemcpyAsync(d_ptr, match_ptr, sizeof(long long) * (n_query + 1), hipMemcpyHostToDevice, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(d_ref, uv_ref, sizeof(float2) * nm, hipMemcpyHostToDevice, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(d_cur, uv_cur, sizeof(float2) * nm, hipMemcpyHostToDevice, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(d_cref, cam_ref, sizeof(double) * 15 * n_query, hipMemcpyHostToDevice, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(d_ccur, cam_cur, sizeof(double) * 15 * n_query, hipMemcpyHostToDevice, h.st));
  }
  PTZ_HIP_TRY(hipEventRecord(h.e0, h.st));
  launch_krt(n_query, d_ptr, d_ref, d_cur, d_pptr, d_puv, d_pxyz, d_cref, d_ccur, factor_type, make_krt_opt(o, max_reproj_error), d_sum,
             d_acc, h.st);
  PTZ_HIP_TRY(hipEventRecord(h.e1, h.st));
  if (staged) {
    char* ps = (char*)h.pinned;
    PTZ_HIP_TRY(hipMemcpyAsync(ps + o_ccur, h.base + o_ccur, total - o_ccur, hipMemcpyDeviceToHost, h.st));
    PTZ_HIP_TRY(stream_wait(h.st));
    PTZ_HIP_TRY(hipGetLastError());  // a refused kernel launch must not pass for a solve
    memcpy(cam_cur, ps + o_ccur, sizeof(double) * 15 * n_query);
    memcpy(summaries, ps + o_sum, sizeof(ptz_lm_summary) * n_query);
    memcpy(accepted, ps + o_acc, sizeof(int) * n_query);
  }
  else {
    PTZ_HIP_TRY(hipMemcpyAsync(cam_cur, d_ccur, sizeof(double) * 15 * n_query, hipMemcpyDeviceToHost, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(summaries, d_sum, sizeof(ptz_lm_summary) * n_query, hipMemcpyDeviceToHost, h.st));
    PTZ_HIP_TRY(hipMemcpyAsync(accepted, d_acc, sizeof(int) * n_query, hipMemcpyDeviceToHost, h.st));
    PTZ_HIP_TRY(stream_wait(h.st));
    PTZ_HIP_TRY(hipGetLastError());  // a refused kernel launch must not pass for a solve
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, h.e0, h.e1);
  if (device_ms) *device_ms = ms;
  return PTZ_OK;
}

// ---- registration attempts over RESIDENT match tables ------------------------------------------------------------------------
// PtzIncrementalOptimizer::RegisterNextImage (ptz_incremental_optimizer.cc:377-418) solves, for an unregistered image j, one KRT
// problem per table entry (registered i -> j): the entry's matches never change during a run, only the two cameras do.  A rig's
// table goes to the device once (ptz_krt_table_create); an attempt is then {table, entry, reference camera, initial camera} --
// thirty doubles instead of the entry's pixels packed, merged and staged again for every launch.  Same kernel, same bits.
struct ptz_krt_table {
  int device = 0;
  int32_t n_entry = 0;
  int64_t n_match = 0;
  std::vector<int64_t> ptr;  // [n_entry + 1]
  float2* d_ref = nullptr;   // [n_match]
  float2* d_cur = nullptr;   // [n_match]
};

extern "C" int32_t ptz_krt_table_create(int32_t n_entry, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, int32_t device_id,
                                        ptz_krt_table** out)
{
  if (!out) return PTZ_EINVAL;
  *out = nullptr;
  if (n_entry <= 0 || !match_ptr || match_ptr[0] != 0) return PTZ_EINVAL;
  for (int32_t e = 0; e < n_entry; ++e)
    if (match_ptr[e + 1] < match_ptr[e]) return PTZ_EINVAL;
  const int64_t nm = match_ptr[n_entry];
  if (nm > 0 && (!uv_ref || !uv_cur)) return PTZ_EINVAL;
  clear_stale_error(__func__);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || ndev <= device_id) return PTZ_ENODEVICE;
  PTZ_DEVICE_GUARD(device_id);
  std::unique_ptr<ptz_krt_table> t(new ptz_krt_table());
  t->device = device_id; t->n_entry = n_entry; t->n_match = nm;
  t->ptr.assign(match_ptr, match_ptr + n_entry + 1);
  const size_t bytes = sizeof(float2) * (size_t)std::max<int64_t>(nm, 1);
  void* blk = nullptr;
  if (ptzpool::dev_acquire(device_id, 2 * bytes, &blk) != hipSuccess) return PTZ_ENOMEM;
  t->d_ref = static_cast<float2*>(blk);
  t->d_cur = reinterpret_cast<float2*>(static_cast<char*>(blk) + bytes);
  if (nm > 0) {
    hipStream_t st = nullptr;
    hipError_t e = ptzpool::stream_acquire(device_id, &st);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_ref, uv_ref, sizeof(float2) * nm, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_cur, uv_cur, sizeof(float2) * nm, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = stream_wait(st);
    if (st) ptzpool::stream_release(device_id, st);
    if (e != hipSuccess) { (void)hipGetLastError(); ptzpool::dev_release(device_id, blk); return PTZ_ENODEVICE; }
  }
  *out = t.release();
  return PTZ_OK;
}

extern "C" void ptz_krt_table_destroy(ptz_krt_table* t)
{
  if (!t) return;
  DeviceGuard guard(t->device);
  ptzpool::dev_release(t->device, t->d_ref);  // (one block: d_cur lies behind d_ref)
  delete t;
}

extern "C" int32_t ptz_krt_solve_attempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur,
                                          int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                                          ptz_lm_summary* summaries, int32_t* accepted, double* device_ms)
{
  if (n_query <= 0 || !attempts || !cam_ref || !cam_cur || !summaries || !accepted) return PTZ_EINVAL;
  if (factor_type < PTZ_KRT_F || factor_type > PTZ_KRT_FxfyDist) return PTZ_EUNSUPPORTED;
  ptz_lm_options o;
  if (opt) o = *opt; else ptz_lm_options_default(&o);
  const ptz_krt_table* t0 = attempts[0].table;
  if (!t0) return PTZ_EINVAL;
  const int device = t0->device;  // the launch goes where the tables live
  if (opt && opt->device_id != 0 && opt->device_id != device) return PTZ_EINVAL;
  for (int q = 0; q < n_query; ++q) {
    const ptz_krt_table* t = attempts[q].table;
    if (!t || t->device != device || attempts[q].entry < 0 || attempts[q].entry >= t->n_entry) return PTZ_EINVAL;
  }
  clear_stale_error(__func__);
  PTZ_DEVICE_GUARD(device);
  // one pooled device block, one pinned staging block: [ranges | deltas | cam_ref | cam_cur (in / out) | summaries | accepted]
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_rng = 0, o_del = o_rng + up(sizeof(long long) * 2 * n_query), o_cref = o_del + up(sizeof(long long) * n_query),
               o_ccur = o_cref + up(sizeof(double) * 15 * n_query), o_sum = o_ccur + up(sizeof(double) * 15 * n_query),
               o_acc = o_sum + up(sizeof(ptz_lm_summary) * n_query), total = o_acc + up(sizeof(int) * n_query);
  struct Held {
    int dev; char* base = nullptr; void* pinned = nullptr; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Held()
    {
      if (st) (void)stream_wait(st);
      ptzpool::dev_release(dev, base);
      ptzpool::pinned_release(pinned);
      ptzpool::stream_release(dev, st);
      ptzpool::event_release(dev, true, e0);
      ptzpool::event_release(dev, true, e1);
    }
  } h;
  h.dev = device;
  if (ptzpool::dev_acquire(h.dev, total, (void**)&h.base) != hipSuccess) return PTZ_ENOMEM;
  if (ptzpool::pinned_acquire(total, &h.pinned) != hipSuccess) { (void)hipGetLastError(); return PTZ_ENOMEM; }
  PTZ_HIP_TRY(ptzpool::stream_acquire(h.dev, &h.st));
  PTZ_HIP_TRY(ptzpool::event_acquire(h.dev, true, &h.e0));
  PTZ_HIP_TRY(ptzpool::event_acquire(h.dev, true, &h.e1));
  char* ps = (char*)h.pinned;
  long long* rng = (long long*)(ps + o_rng);
  long long* del = (long long*)(ps + o_del);
  for (int q = 0; q < n_query; ++q) {
    const ptz_krt_table* t = attempts[q].table;
    // the table's pixels, in elements from the first table's (the tables are separate allocations of one address space: the
    // distances are taken on the addresses, not by pointer subtraction; every block of the pool is 256-byte aligned)
    auto dist = [](const float2* a, const float2* b0) {
      return (long long)((reinterpret_cast<intptr_t>(a) - reinterpret_cast<intptr_t>(b0)) / (intptr_t)sizeof(float2));
    };
    const long long at = dist(t->d_ref, t0->d_ref);
    rng[2 * q] = at + t->ptr[attempts[q].entry];
    rng[2 * q + 1] = at + t->ptr[attempts[q].entry + 1];
    del[q] = dist(t->d_cur, t0->d_cur) - at;
  }
  memcpy(ps + o_cref, cam_ref, sizeof(double) * 15 * n_query);
  memcpy(ps + o_ccur, cam_cur, sizeof(double) * 15 * n_query);
  PTZ_HIP_TRY(hipMemcpyAsync(h.base, ps, o_sum, hipMemcpyHostToDevice, h.st));
  PTZ_HIP_TRY(hipEventRecord(h.e0, h.st));
  launch_krt(n_query, (const long long*)(h.base + o_rng), t0->d_ref, t0->d_cur, nullptr, nullptr, nullptr, (const double*)(h.base + o_cref),
             (double*)(h.base + o_ccur), factor_type, make_krt_opt(o, max_reproj_error), (ptz_lm_summary*)(h.base + o_sum), (int*)(h.base + o_acc), h.st,
             (const long long*)(h.base + o_del));
  PTZ_HIP_TRY(hipEventRecord(h.e1, h.st));
  PTZ_HIP_TRY(hipMemcpyAsync(ps + o_ccur, h.base + o_ccur, total - o_ccur, hipMemcpyDeviceToHost, h.st));
  PTZ_HIP_TRY(stream_wait(h.st));
  PTZ_HIP_TRY(hipGetLastError());  // a refused kernel launch must not pass for a solve
  memcpy(cam_cur, ps + o_ccur, sizeof(double) * 15 * n_query);
  memcpy(summaries, ps + o_sum, sizeof(ptz_lm_summary) * n_query);
  memcpy(accepted, ps + o_acc, sizeof(int) * n_query);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, h.e0, h.e1);
  if (device_ms) *device_ms = ms;
  return PTZ_OK;
}

// Queries over several devices of one node, from one process: contiguous chunks of (nearly) equal total match count, one host
// thread per device, results written straight into the caller's arrays (the queries are independent, run_ptz_reloc.cc:68).
extern "C" int32_t ptz_krt_solve_batch_sharded(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                                               const int64_t* point_ptr, const float* pts2d, const double* pts3d,
                                               const double* cam_ref, double* cam_cur, int32_t factor_type, double max_reproj_error,
                                               const int32_t* device_ids, int32_t n_devices, const ptz_lm_options* opt,
                                               ptz_lm_summary* summaries, int32_t* accepted)
{
  if (n_query <= 0 || !match_ptr || !device_ids || n_devices <= 0 || !summaries || !accepted) return PTZ_EINVAL;
  ptz_lm_options base;
  if (opt) base = *opt; else ptz_lm_options_default(&base);
  const int nd = std::min(n_devices, n_query);
  // chunk boundaries: query index at which the running match count passes k / nd of the total
  std::vector<int> cut(nd + 1, 0);
  cut[nd] = n_query;
  const int64_t total = match_ptr[n_query] - match_ptr[0];
  for (int k = 1; k < nd; ++k) {
    const int64_t want = match_ptr[0] + total * k / nd;
    int q = (int)(std::lower_bound(match_ptr, match_ptr + n_query + 1, want) - match_ptr);
    cut[k] = std::min(std::max(q, cut[k - 1] + 1), n_query - (nd - k));
  }
  std::vector<int> rcs(nd, PTZ_OK);
  auto run = [&](int k) {
    const int q0 = cut[k], nq = cut[k + 1] - cut[k];
    std::vector<int64_t> mp(nq + 1), pp;
    for (int q = 0; q <= nq; ++q) mp[q] = match_ptr[q0 + q] - match_ptr[q0];
    if (point_ptr) {
      pp.resize(nq + 1);
      for (int q = 0; q <= nq; ++q) pp[q] = point_ptr[q0 + q] - point_ptr[q0];
    }
    ptz_lm_options o = base;
    o.device_id = device_ids[k];
    rcs[k] = ptz_krt_solve_batch_2d3d(nq, mp.data(), uv_ref + 2 * match_ptr[q0], uv_cur + 2 * match_ptr[q0],
                                      point_ptr ? pp.data() : nullptr, point_ptr ? pts2d + 2 * point_ptr[q0] : nullptr,
                                      point_ptr ? pts3d + 3 * point_ptr[q0] : nullptr, cam_ref + 15 * (size_t)q0, cam_cur + 15 * (size_t)q0,
                                      factor_type, max_reproj_error, &o, summaries + q0, accepted + q0, nullptr);
  };
  std::vector<std::thread> th;
  for (int k = 1; k < nd; ++k) th.emplace_back(run, k);
  run(0);
  for (auto& x : th) x.join();
  for (int rc : rcs) if (rc) return rc;
  return PTZ_OK;
}

