/*
 * ptz_oracle.h -- CPU restatement of the PTZ-Calib hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (ptz-calib_amd/csrc, HIP) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned** for the floating-point path.  The reference
 * (gjgjh/PTZ-Calib) has no tests, golden vectors or data in-tree, and its arithmetic lives
 * in Ceres 1.14.0 + OpenCV 4.5.3 (install_deps.sh:44-128), neither of which exists in the
 * build image, so the reference binary cannot be run.  What IS pinned:
 *   - union-find root ids (track ids) against the reference's own union_find.h /
 *     flat_pair_map.h compiled from /root/reference (oracle/_ref, see oracle/Makefile);
 *   - evaluation metrics against scripts/eval_synthetic.py imported in the build
 *     container (tests/golden/eval_synthetic_vectors.json).
 * Everything tagged [Ceres-1.14] / [OpenCV-4.5.3] below restates the published algorithm of
 * that release from its documentation/source as remembered; it is anchored on the
 * reference's call sites, cited file:line (relative to /root/reference).
 */
#ifndef PTZ_ORACLE_H
#define PTZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Camera 15-vector layout, types.cc:32-73:
 * [fx, fy, cx, cy, r1, r2, r3, t1, t2, t3, k1, k2, k3, p1, p2] */
#define ORC_CAM_DIM 15

/* FACTOR_TYPE of PTZRayOptimizer, ptzray_optimizer.h:110 */
enum { ORC_PTZRay = 0, ORC_PTZRayDist = 1, ORC_PTZRayFxfyDist = 2, ORC_PTZRayDistDisp = 3 };
/* KRTOptimizer::FACTOR_TYPE, krt_optimizer.h:110 */
enum { ORC_KRT_F = 0, ORC_KRT_FDist = 1, ORC_KRT_Fxfy = 2, ORC_KRT_FxfyDist = 3 };
/* ceres::TerminationType [Ceres-1.14 types.h] */
enum { ORC_CONVERGENCE = 0, ORC_NO_CONVERGENCE = 1, ORC_FAILURE = 2 };
/* Jacobian mode: 0 = central differences exactly as ceres::NumericDiffCostFunction<...,CENTRAL,...>
 * (what the reference runs), 1 = closed-form derivatives (what the HIP kernels use). */
enum { ORC_JAC_NUMERIC = 0, ORC_JAC_ANALYTIC = 1 };

typedef struct {
  int32_t max_num_iterations;       /* ptzray_optimizer.cc:470, krt_optimizer.cc:388 */
  int32_t jacobian_mode;            /* ORC_JAC_* */
  int32_t num_threads;              /* OpenMP threads for residual/Jacobian evaluation */
  int32_t reserved;
  /* [Ceres-1.14] Solver::Options defaults */
  double initial_trust_region_radius; /* 1e4  */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;             /* 1 */
} orc_lm_options;

void orc_lm_options_default(orc_lm_options* o);

typedef struct {
  int32_t termination_type;      /* ORC_CONVERGENCE / NO_CONVERGENCE / FAILURE */
  int32_t num_iterations;        /* [Ceres] summary.iterations.size() - 1 */
  int32_t num_lm_steps;          /* trust-region loop passes executed (incl. the terminating one) */
  int32_t num_successful_steps;  /* [Ceres] iteration 0 counted as successful (2.x convention; 1.14 unverified -- the value is never read by the reference) */
  int32_t num_unsuccessful_steps;
  int32_t num_residuals;         /* scalar residuals */
  int32_t num_linear_solves;
  int32_t num_jacobian_evals;
  double initial_cost;
  double final_cost;
  double final_radius;
  double final_gradient_max_norm;
} orc_lm_summary;

/* Per-iteration trace (optional; pass NULL).  Arrays sized max_num_iterations + 2. */
typedef struct {
  int32_t capacity;
  int32_t count;
  double* cost;          /* cost recorded in the iteration summary */
  double* cost_change;
  double* radius;        /* trust-region radius after the iteration */
  double* rho;           /* relative_decrease */
  int32_t* accepted;     /* 1 = successful step, 0 = rejected, -1 = invalid */
} orc_lm_trace;

/* ---------------- F6: Rodrigues (types.cc:41,68; OpenCV-4.5.3 cv::Rodrigues) ---------------- */
void orc_rodrigues(const double rvec[3], double R[9]);               /* vector -> row-major matrix */
void orc_rodrigues_jac(const double rvec[3], double R[9], double dR[27]); /* + dR/dr_k, k-major */
void orc_rodrigues_inv(const double R[9], double rvec[3]);           /* matrix -> vector */

/* ---------------- F1..F5 residual functors ---------------- */
/* intr[9] = {fx,fy,cx,cy,k1,k2,k3,p1,p2}, extr[6] = {rvec,t}, ray[3]; uv float32 pair. */
void orc_res_ptzray(const double* intr, const double* extr, const double* ray, const float* uv, double* res);          /* ptzray_optimizer.cc:20-56  */
void orc_res_ptzray_dist(const double* intr, const double* extr, const double* ray, const float* uv, double* res);     /* :65-129 */
void orc_res_ptzray_fxfy_dist(const double* intr, const double* extr, const double* ray, const float* uv, double* res);/* :138-193 */
void orc_res_reproj2d3d(const double* intr, const double* extr, const double* tlw, const float* uv, const double* xyz, double* res); /* :268-326 */
/* displacement variants: disp[3] = (d0, d1, d2), camera-frame z += d0 + d1 fx + d2 fx^2 */
void orc_res_ptzray_dist_disp(const double* intr, const double* disp, const double* extr, const double* ray, const float* uv, double* res); /* :195-259 */
void orc_res_reproj2d3d_disp(const double* intr, const double* disp, const double* extr, const double* tlw, const float* uv, const double* xyz, double* res); /* :334-396 */
/* cam1 = reference camera in its own local frame (R = I, t = 0): k1[4] = {fx,fy,cx,cy}, dist1[5] in the
 * reference's Camera layout (k1,k2,k3,p1,p2).  cam[15] = current camera vector. */
void orc_res_2d2d(const double* cam, const double* k1, const float* uv1, const float* uv2, double* res);               /* krt_optimizer.cc:22-43 */
void orc_res_2d2d_dist(const double* cam, const double* k1, const double* dist1, const float* uv1, const float* uv2, double* res); /* :80-132 */
/* cv::undistortPoints(src, dst, K, dist, noArray(), K) for one point; result rounded to float32 as
 * cv::Point2f does (krt_optimizer.cc:89-92).  dist is passed exactly as the reference passes it. */
void orc_undistort_point(const double* k1, const double* dist1, const float* uv, float* out);

/* ---------------- P3: feature-track builder (tracks.cc:19-113, union_find.h, flat_pair_map.h) ---------------- */
/* Input: n_pairs match lists; pair p joins (src[p], query[k]) with (dst[p], train[k]) for k in
 * [match_ptr[p], match_ptr[p+1]).  Output (caller frees with orc_free): tracks sorted by track id,
 * entries sorted by image id.  Returns number of tracks, or -1 on error. */
int32_t orc_tracks_build(int32_t n_pairs, const int64_t* src, const int64_t* dst, const int64_t* match_ptr,
                         const int32_t* query_idx, const int32_t* train_idx, int32_t min_track_length,
                         int32_t** track_id, int64_t** track_ptr, int32_t** entry_image, int32_t** entry_feature);
void orc_free(void* p);

/* ---------------- PTZ-IBA global bundle adjustment (ptzray_optimizer.cc:454-489) ---------------- */
typedef struct {
  int32_t n_cam;              /* candidate cameras, compact ids */
  int32_t n_ray;              /* tracks with >= 1 candidate observation */
  int64_t n_obs;              /* 2D-2D observations */
  const float* obs_uv;        /* [2*n_obs] float32 pixels (data_io.cc:40) */
  const int32_t* obs_cam;     /* [n_obs] */
  const int32_t* obs_ray;     /* [n_obs] non-decreasing (track asc, image asc: ptzray_optimizer.cc:801-850) */
  const double* ray_weight;   /* [n_ray] ScaledLoss weight = full track length (:805-806) */
  int32_t n_obs3d;            /* 2D-3D annotation observations (:887-923) */
  const float* obs3d_uv;      /* [2*n_obs3d] */
  const double* obs3d_xyz;    /* [3*n_obs3d] world points */
  const int32_t* obs3d_cam;   /* [n_obs3d] */
  int32_t factor_type;        /* ORC_PTZRay... */
  /* shared_ic_ids_ (ptzray_optimizer.cc:427-428, 497-505): cameras with equal ids share ONE intrinsics parameter block
   * (intrinsics_param_ is keyed by the id, :645-650).  NULL = every camera its own block (the reference's default). */
  const int32_t* ic_of_cam;   /* [n_cam] or NULL */
} orc_ba_problem;

/* cam[15*n_cam], ray[3*n_ray], tlw[6] are updated in place with the best point found
 * ([Ceres] parameters <- x at minimum cost), regardless of termination type. */
int32_t orc_ba_solve(const orc_ba_problem* p, double* cam, double* ray, double* tlw, const orc_lm_options* o,
                     orc_lm_summary* s, orc_lm_trace* trace);

/* The same with the displacement block of PTZRayDistDisp (disp_param_, ptzray_optimizer.cc:655: starts at zero when
 * disp = NULL; in/out otherwise).  For the other factor types disp is carried along untouched. */
int32_t orc_ba_solve_disp(const orc_ba_problem* p, double* cam, double* ray, double* tlw, double* disp, const orc_lm_options* o,
                          orc_lm_summary* s, orc_lm_trace* trace);
int32_t orc_ba_residuals_disp(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                              const double* disp, double* res);

/* One linearisation at the given point, for kernel parity tests.  Outputs (any may be NULL):
 *  cost; per-camera free-parameter gradient g_c[ncf*n_cam] and diagonal blocks U[ncf*ncf*n_cam];
 *  per-ray g_r[3*n_ray], V[9*n_ray]; per-observation W[ncf*3*n_obs] (camera x ray coupling).
 *  Weighted (x sqrt(w)) but NOT Jacobi-scaled.  ncf = orc_ba_cam_free_dim(factor_type). */
int32_t orc_ba_cam_free_dim(int32_t factor_type);
int32_t orc_ba_linearize(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                         int32_t jacobian_mode, double* cost, double* g_c, double* U, double* g_r, double* V, double* W);
/* the same at a given displacement block (PTZRayDistDisp: ncf = 9, the last three camera columns are the block's columns
 * d res / d (d0, d1, d2), identical parameter for every camera) */
int32_t orc_ba_linearize_disp(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw,
                              const double* disp, int32_t jacobian_mode, double* cost, double* g_c, double* U, double* g_r,
                              double* V, double* W);
/* residuals only: res[2*n_obs + 2*n_obs3d], unweighted */
int32_t orc_ba_residuals(const orc_ba_problem* p, const double* cam, const double* ray, const double* tlw, double* res);

/* ---------------- KRT single-view LM (krt_optimizer.cc:265-404) ---------------- */
typedef struct {
  int32_t n_match;
  const float* uv_ref;   /* [2*n_match] keypoints of the reference view */
  const float* uv_cur;   /* [2*n_match] keypoints of the current view */
  const double* cam_ref; /* [15] reference camera in world frame (only K, dist are used by the functors;
                            R, t define the local frame, krt_optimizer.cc:269-282) */
  int32_t factor_type;   /* ORC_KRT_* */
  /* optional 2D-3D constraints, KRTOptimizer::Add2d3dConstraints (krt_optimizer.cc:350-383) */
  int32_t n_pt;
  const float* pts2d;        /* [2*n_pt] pixels */
  const double* pts3d_local; /* [3*n_pt] points already in the local frame (orc_krt_point_to_local) */
} orc_krt_problem;

/* cam_cur[15]: in = initial camera in *local* frame vector form (krt_optimizer.cc:284), out = refined. */
int32_t orc_krt_solve(const orc_krt_problem* p, double* cam_cur_local, const orc_lm_options* o, orc_lm_summary* s,
                      orc_lm_trace* trace);
/* Factor2d3dDist / Factor2d3dFxfyDist functor (krt_optimizer.cc:200-249): cv::projectPoints of one local-frame point with
 * the camera 15-vector (F / FDist: fy := fx).  OpenCV 4.5.3 cvProjectPoints2Internal arithmetic, 5 coefficients read as
 * (k1,k2,p1,p2,k3). */
void orc_res_2d3d_krt(const double* cam, int32_t fxfy, const float* pt2d, const double* pt3d_local, double* res);
/* R_local_world X_w + t_local_world with the reference camera's pose (krt_optimizer.cc:357-362) */
void orc_krt_point_to_local(const double* cam_ref_world, const double* pt3d_world, double* pt3d_local);
/* world <-> local frame helpers (krt_optimizer.cc:269-284, 535-567); cams are 15-vectors */
void orc_krt_world_to_local(const double* cam_ref_world, const double* cam_cur_world, double* cam_cur_local);
void orc_krt_local_to_world(const double* cam_ref_world, const double* cam_cur_local, int32_t factor_type,
                            double* cam_cur_world);
/* KRTOptimizer::CheckResults gates (krt_optimizer.cc:504-533): returns 1 if accepted */
int32_t orc_krt_check(const orc_lm_summary* s, const double* cam_cur_local, double max_reproj_error);

/* The relocalization loop of run_ptz_reloc.cc:68-118 over many queries, each one KRTOptimizer as that loop uses it: initial
 * camera into the reference view's local frame (krt_optimizer.cc:269-284), solve, CheckResults with max_reproj_error (:504-533),
 * back to the world frame if accepted (:535-567; the camera is left untouched otherwise, :396-403).  The reference runs the
 * queries one after the other (with Ceres' own threads inside each tiny solve); num_threads > 1 deals whole queries to that
 * many threads instead -- the CPU baseline of the benchmark.  cam_cur_world: [15 * n_query] in = initial, out = refined. */
int32_t orc_krt_solve_batch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                            const double* cam_ref_world, double* cam_cur_world, int32_t factor_type, double max_reproj_error,
                            const orc_lm_options* o, orc_lm_summary* summaries, int32_t* accepted, int32_t num_threads);

/* Pix2Ray initialisation (ptzray_optimizer.cc:768-797) for packed observations */
void orc_pix2ray(const orc_ba_problem* p, const double* cam, double* ray);

#ifdef __cplusplus
}
#endif
#endif
