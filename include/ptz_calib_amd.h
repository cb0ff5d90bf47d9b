/*
 * ptz_calib_amd.h -- C-ABI of libptzcalib_hip.so: the MI355X (gfx950) implementation of the PTZ-Calib
 * optimisation hot path (per-observation PTZ reprojection residual/Jacobian evaluation, Levenberg-
 * Marquardt normal-equation assembly, Schur elimination, dense reduced-camera solve, LM control).
 *
 * The reference (gjgjh/PTZ-Calib) has no FFI layer; its seam is the pair of C++ classes that each own
 * a ceres::Problem.  Every entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  Plain pointers and sizes only; no C++ or torch types.
 * All functions return 0 on success or a negative PTZ_E* code; no exceptions cross the boundary and
 * there is no CPU fallback: without a HIP device every compute entry point returns PTZ_ENODEVICE.
 *
 * Conventions
 *  - Camera = 15 doubles, layout of Camera::ToVector (src/core/types.cc:32-57):
 *      [fx, fy, cx, cy, r1, r2, r3, t1, t2, t3, k1, k2, k3, p1, p2]
 *  - pixels are float32 pairs (cv::Point2f; src/core/data_io.cc:40), everything else float64
 *  - observations are sorted (track id ascending, image id ascending), the order in which
 *    PTZRayOptimizer::AddConstraints2d2d adds residual blocks (src/core/ptzray_optimizer.cc:801-850)
 *  - "host" pointers are ordinary process memory; "dev" pointers are HIP device memory
 */
#ifndef PTZ_CALIB_AMD_H
#define PTZ_CALIB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTZ_CAM_DIM 15

/* error codes */
#define PTZ_OK 0
#define PTZ_EINVAL (-1)     /* malformed problem (CheckValid, ptzray_optimizer.cc:515-535) */
#define PTZ_ENODEVICE (-2)  /* no usable HIP device / HIP runtime error */
#define PTZ_ENOMEM (-3)
#define PTZ_EUNSUPPORTED (-4) /* factor type not implemented on the device path */
#define PTZ_ELIMIT (-5)     /* a problem dimension beyond what the device path handles: more than 65535 observations in ONE view
                               (16-bit positions in the camera-pair records), more than 2^31 - 1 observations or pair entries in
                               one batch.  The number of views and tracks is bounded by device memory only (the reduced camera
                               system is stored as dense tiles: 8 (NC n_views + 64)^2 bytes per scene).  Never a silent failure:
                               the C++ classes report it on stderr and return false. */
#define PTZ_ENOOBS (-6)     /* ptz_ba_batch_create_views: a view none of whose tracks has a candidate observation -- no residual block, not a problem
                               (the reference's Solve returns false there, ptzray_optimizer.cc:517); every other malformed view is PTZ_EINVAL */

/* enum FACTOR_TYPE { PTZRay, PTZRayDist, PTZRayFxfyDist, PTZRayDistDisp }  (ptzray_optimizer.h:110) */
#define PTZ_BA_PTZRay 0
#define PTZ_BA_PTZRayDist 1
#define PTZ_BA_PTZRayFxfyDist 2 /* fx, fy, k1 free (ptzray_optimizer.cc:136-191); camera block of 6 columns */
#define PTZ_BA_PTZRayDistDisp 3 /* fx, k1 free + ONE 3-parameter displacement block shared by every residual of the problem
                                 * (ptzray_optimizer.cc:195-265, disp_param_ :655; with annotations Reproj2d3dDispFactor :334-396);
                                 * CLOSED-FORM JACOBIANS ONLY, a documented deviation (type 3 is dead from the reference's tools): the
                                 * reference differentiates the d2 column numerically with delta = 1.49e-8 against f^2 ~ 6e6, so ITS
                                 * column is a per cent off and its trajectory follows that through the flat valley of (f, k1, disp);
                                 * this library converges to the closed-form oracle's point (1e-6) and may stop up to 8 % in f from the
                                 * numeric-differentiation reference at equal cost (tests/test_gpu_disp.py holds the number) */
/* KRTOptimizer::FACTOR_TYPE { F, FDist, Fxfy, FxfyDist }  (krt_optimizer.h:110) */
#define PTZ_KRT_F 0
#define PTZ_KRT_FDist 1
#define PTZ_KRT_Fxfy 2      /* fy free as well: Factor2d2dFxfy (krt_optimizer.cc:52-71), dead from the reference's tools */
#define PTZ_KRT_FxfyDist 3  /* Factor2d2dFxfyDist (krt_optimizer.cc:141-192) */
/* ceres::TerminationType as read by the reference (ptzray_optimizer.cc:482, krt_optimizer.cc:513) */
#define PTZ_CONVERGENCE 0
#define PTZ_NO_CONVERGENCE 1
#define PTZ_FAILURE 2

/* Solver options.  The reference sets max_num_iterations, linear_solver_type, num_threads only
 * (ptzray_optimizer.cc:469-473, krt_optimizer.cc:387-391); every other field is the Ceres 1.14
 * default and is exposed so that the defaults are data, not code. */
typedef struct ptz_lm_options {
  int32_t max_num_iterations;                /* 200 (run_ptz_ba.cc:52) / 100 (ptz_incremental_optimizer.cc:396) */
  int32_t device_id;                         /* HIP device ordinal */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;                    /* 1 */
  double initial_trust_region_radius;        /* 1e4 */
  double max_trust_region_radius;            /* 1e16 */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3 */
  double min_lm_diagonal;                    /* 1e-6 */
  double max_lm_diagonal;                    /* 1e32 */
  double function_tolerance;                 /* 1e-6 */
  double gradient_tolerance;                 /* 1e-10 */
  double parameter_tolerance;                /* 1e-8 */
  /* ptz_krt_solve_batch*: lanes of a wavefront that work on one query.  64 = a wave per query (lowest latency: registration
   * attempts, a handful of queries); 16 = four queries per wave (highest throughput once the launch fills the GPU); 0 = the
   * library picks from the launch size (16 from 16384 queries on).  The two forms add the matches up in different orders: a
   * query's bits depend on the form, never on its neighbours in the launch -- callers that compare runs bit by bit pin it. */
  int32_t krt_lanes_per_query;               /* 0 */
  int32_t reserved_;                         /* 0 */
} ptz_lm_options;

/* The ceres::Solver::Summary fields the reference reads (ptzray_optimizer.cc:962-963,482;
 * krt_optimizer.cc:396,506-513) plus iteration bookkeeping. */
typedef struct ptz_lm_summary {
  int32_t termination_type;     /* PTZ_CONVERGENCE / PTZ_NO_CONVERGENCE / PTZ_FAILURE */
  int32_t num_iterations;       /* summary.iterations.size() - 1 */
  int32_t num_lm_steps;         /* trust-region loop passes executed (includes the terminating pass) */
  int32_t num_successful_steps; /* iteration 0 counts as successful (Ceres 2.x convention; whether 1.14 did is unverified here --
                                   the reference only stores the value in KRTOptimizer::num_iter_, krt_optimizer.cc:396, and never reads it) */
  int32_t num_unsuccessful_steps;
  int32_t num_residuals;        /* scalar residuals */
  int32_t num_linear_solves;
  int32_t num_jacobian_evals;
  double initial_cost;
  double final_cost;
  double final_radius;
  double final_gradient_max_norm;
} ptz_lm_summary;

void ptz_lm_options_default(ptz_lm_options* o);

/* Library / device probe.  ptz_device_count() never initialises a GPU context beyond counting. */
const char* ptz_version(void);
int32_t ptz_device_count(void);
/* The reference builds a new optimizer object per solve; so do callers of this library.  Device blocks, pinned blocks,
 * streams and events released by finished solves are parked in a process-wide cache (budget PTZ_CACHE_MAX_MB of device
 * memory per GPU, default 32768) and reused by later ones.  ptz_trim_cache() hands everything parked back to the driver. */
void ptz_trim_cache(void);

/* ------------------------------------------------------------------------------------------------
 * PTZ-IBA global bundle adjustment
 *   replaces PTZRayOptimizer::Solve (ptzray_optimizer.cc:454-489) from AddConstraints2d2d onwards:
 *   residual blocks, SubsetParameterization masks, ScaledLoss weights, ceres::Solve(SPARSE_SCHUR).
 * One problem = one scene's candidate cameras + tracks, in packed form.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ptz_ba_problem {
  int32_t n_cam;             /* candidate cameras (compact ids 0..n_cam-1) */
  int32_t n_ray;             /* tracks with >= 1 candidate observation */
  int64_t n_obs;             /* 2D-2D observations */
  const float* obs_uv;       /* host [2*n_obs] */
  const int32_t* obs_cam;    /* host [n_obs] */
  const int32_t* obs_ray;    /* host [n_obs] non-decreasing */
  const double* ray_weight;  /* host [n_ray]: ScaledLoss weight = full track length (ptzray_optimizer.cc:805) */
  int32_t n_obs3d;           /* 2D-3D annotation observations (ptzray_optimizer.cc:887-923); 0 = none */
  const float* obs3d_uv;     /* host [2*n_obs3d] */
  const double* obs3d_xyz;   /* host [3*n_obs3d] */
  const int32_t* obs3d_cam;  /* host [n_obs3d] */
  int32_t factor_type;       /* PTZ_BA_* */
  /* PTZRayOptimizer::SetSharedIntrinsics (ptzray_optimizer.cc:497-505): cameras with equal ids share ONE intrinsics block
   * (fx, fy, cx, cy, k1..p2); its initial value is the first such camera's (:645-650).  NULL = every camera its own block,
   * the reference's default (:427-428). */
  const int32_t* ic_of_cam;  /* host [n_cam] or NULL */
} ptz_ba_problem;

typedef struct ptz_ba_batch ptz_ba_batch; /* opaque: device-resident problems + workspaces */

/* Build a device-resident batch of n independent problems (all with the same factor_type).  Uploads
 * the observation records, builds the structural indices (per-camera observation lists, camera-pair
 * lists for the Schur complement) and allocates every workspace, so that ptz_ba_batch_solve performs
 * no allocation.  The problem arrays may be released after this call returns. */
int32_t ptz_ba_batch_create(int32_t n, const ptz_ba_problem* problems, const ptz_lm_options* opt, ptz_ba_batch** out);
void ptz_ba_batch_destroy(ptz_ba_batch* b);

/* Initial point.  cam: host [15 * sum(n_cam)], ray: host [3 * sum(n_ray)], tlw: host [6 * n] (may be
 * NULL = zeros), each concatenated over the problems in order
 * (SetUpInitialCameraParams, ptzray_optimizer.cc:635-670). */
int32_t ptz_ba_batch_set_state(ptz_ba_batch* b, const double* cam, const double* ray, const double* tlw);
/* Run LM on every problem of the batch from the state last set (the state set by
 * ptz_ba_batch_set_state is kept, so repeated calls re-solve from the same initial point).
 * Device work is enqueued on the batch's stream; the call returns after the stream has drained.
 * summaries: host [n]. */
int32_t ptz_ba_batch_solve(ptz_ba_batch* b, ptz_lm_summary* summaries);
/* Solution = parameters at the minimum-cost point, as Ceres writes back (any termination type). */
int32_t ptz_ba_batch_get_state(ptz_ba_batch* b, double* cam, double* ray, double* tlw);
/* PTZRayDistDisp only (else PTZ_EUNSUPPORTED): the displacement block of every problem, disp [3 * n]: (d0, d1, d2) of
 * delta_z = d0 + d1 fx + d2 fx^2.  The initial value is zero (disp_param_ = {0, 0, 0}, ptzray_optimizer.cc:655) unless set;
 * get returns the block at the minimum-cost point like ptz_ba_batch_get_state (the reference adds it to t_z, :693, :714). */
int32_t ptz_ba_batch_set_disp(ptz_ba_batch* b, const double* disp);
int32_t ptz_ba_batch_get_disp(ptz_ba_batch* b, double* disp);
/* Device time of the last ptz_ba_batch_solve in milliseconds (HIP events on the batch's stream),
 * and, per kernel family, the accumulated device time and launch count when profiling was enabled
 * with ptz_ba_batch_set_profiling(b, 1) (serialises the stream between kernels and runs the batch as ONE scene group,
 * so that no other group's kernels share the device while a family is being timed). */
int32_t ptz_ba_batch_last_solve_ms(const ptz_ba_batch* b, double* ms);
int32_t ptz_ba_batch_set_profiling(ptz_ba_batch* b, int32_t enable);
#define PTZ_PROF_SLOTS 16
int32_t ptz_ba_batch_get_profile(const ptz_ba_batch* b, double* ms_per_slot, int64_t* launches_per_slot,
                                 const char** slot_names);

/* One-shot convenience: create + set_state + solve + get_state + destroy for a single problem.
 * cam/ray/tlw are updated in place. */
int32_t ptz_ba_solve(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, const ptz_lm_options* opt,
                     ptz_lm_summary* summary);
/* The same for PTZRayDistDisp: disp [3] initial value in (NULL: zeros), refined value out.  cam[9] (t_z) is returned as the
 * parameter block holds it; the reference's read-back also adds d0 + d1 fx + d2 fx^2 to it (ptzray_optimizer.cc:693, 714):
 * that last step is the caller's (the C++ class PTZRayOptimizer and api.fold_displacement do it). */
int32_t ptz_ba_solve_disp(const ptz_ba_problem* p, double* cam, double* ray, double* tlw, double* disp,
                          const ptz_lm_options* opt, ptz_lm_summary* summary);

/* Kernel-level entry points used by the parity tests (one linearisation at the current state of
 * problem `index`; weighted by sqrt(track length), not Jacobi-scaled).  Host outputs, any may be NULL:
 *   cost; g_c [nc*n_cam], U [nc*nc*n_cam]; g_r [3*n_ray], V [9*n_ray]; W [nw*3*n_obs]
 * where nw = ptz_ba_cam_block_dim(factor_type), nc = ptz_ba_batch_cam_block_dim(batch): the free camera parameters carried on the device
 * ([fx, r1, r2, r3] for PTZRay, [fx, k1, r1, r2, r3] for PTZRayDist, [fx, fy, k1, r1, r2, r3] for PTZRayFxfyDist,
 * [fx, k1, r1, r2, r3, d0, d1, d2] for PTZRayDistDisp -- every camera carries its copy of the displacement block, the
 * copies are one parameter; the reference's always-zero fy column (ptzray_optimizer.cc:24-25) is not materialised). */
int32_t ptz_ba_cam_block_dim(int32_t factor_type);
/* ... and of a batch: one more (fy, live through Reproj2d3dFactor, ptzray_optimizer.cc:273) when any problem of the
 * batch carries 2D-3D annotation observations: [fx, fy, (k1), r1, r2, r3]; the reduced system then also holds the
 * 6-dof T_l_w block (ptzray_optimizer.cc:909-910). */
int32_t ptz_ba_batch_cam_block_dim(const ptz_ba_batch* b);
int32_t ptz_ba_batch_linearize(ptz_ba_batch* b, int32_t index, double* cost, double* g_c, double* U, double* g_r,
                               double* V, double* W);
/* Pix2Ray ray initialisation (ptzray_optimizer.cc:768-797) on the device for the whole batch:
 * overwrites the ray part of the stored initial state from the stored cameras. */
int32_t ptz_ba_batch_pix2ray(ptz_ba_batch* b);

/* n independent problems over the devices device_ids[0 .. n_devices) of one node, from one process: scenes are dealt
 * longest-first to the least loaded device, one host thread per device creates / solves / reads back its own batch
 * (replaces the sequential scene loops of the reference's scripts, run_ptzba_synthetic.sh:4-13, run_ptzba_worldcup14.sh).
 * cam [15 * sum n_cam], ray [3 * sum n_ray], tlw [6 * n] or NULL: in = initial values, out = solutions, all in problem
 * order; summaries [n] or NULL.  opt->device_id is ignored.  No collective is involved: the scenes never interact. */
int32_t ptz_ba_solve_sharded(int32_t n, const ptz_ba_problem* problems, double* cam, double* ray, double* tlw,
                             const int32_t* device_ids, int32_t n_devices, const ptz_lm_options* opt, ptz_lm_summary* summaries);

/* ------------------------------------------------------------------------------------------------
 * Rig-resident tracks: bundle adjustments over candidate SUBSETS of one rig without rebuilding the problem on the host
 *   PTZ-IBA adjusts a growing subset of one rig's images ~2N times (AdjustGlobalBundle after every ~10 % of new registrations,
 *   src/core/ptz_incremental_optimizer.cc:91, 420-440), and the reference re-walks the match table into a fresh
 *   PTZRayOptimizer each time (ptzray_optimizer.cc:537-552, 799-850).  Here the rig's tracks (TracksBuilder output after
 *   Filter(4), tracks.cc:19-113) are uploaded ONCE; a bundle adjustment is then a VIEW of them -- the ascending list of candidate
 *   images -- and the packed problem AddConstraints2d2d would produce for that candidate set (observations in (track, image)
 *   order, full-track weights :805, the library's internal ray order, camera-major lists, camera pairs) is built ON THE DEVICE,
 *   word for word the arrays ptz_ba_batch_create builds from the host-packed problem: same batch, same bits.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ptz_rig ptz_rig; /* opaque: one rig's tracks, resident in HBM */
/* trk_ptr [n_track + 1] offsets into the view arrays; per view of a track (image ids ascending inside a track, the order of
 * the reference's map of maps): trk_img [n_view] image id in [0, n_img), trk_uv [2 * n_view] the keypoint's pixel (cv::Point2f). */
int32_t ptz_rig_create(int32_t n_img, int32_t n_track, const int64_t* trk_ptr, const int32_t* trk_img, const float* trk_uv,
                       int32_t device_id, ptz_rig** out);
void ptz_rig_destroy(ptz_rig* rig);
typedef struct ptz_rig_view {
  const ptz_rig* rig;
  int32_t n_cam;             /* candidate images */
  const int32_t* cam_image;  /* host [n_cam] image ids, ascending: compact camera c is image cam_image[c] (SetUpInitialCameraParams' order) */
} ptz_rig_view;
/* The batch ptz_ba_batch_create would build from the n packed problems of these views (2D-2D residuals only, no shared
 * intrinsics): tracks without a candidate view are not rays of the problem; a view without any candidate observation is
 * PTZ_EINVAL.  Ray i of problem k, for set/get_state, is the i-th track of the rig (ascending) that has a candidate view. */
int32_t ptz_ba_batch_create_views(int32_t n, const ptz_rig_view* views, int32_t factor_type, const ptz_lm_options* opt,
                                  ptz_ba_batch** out);
/* Initial state of a batch: cameras cam [15 * sum n_cam] and, computed ON THE DEVICE, the rays by Pix2Ray
 * (ptzray_optimizer.cc:768-797): ray = normalise(mean over the candidate views of normalise(R^-1 K^-1 [u, v, 1])), with the
 * per-camera matrices rkinv [9 * sum n_cam] = R^-1 K^-1 as the caller's (reference's) matrix code evaluates them -- the same
 * operations in the same order as a host loop over the track's views, without contraction: the same bits.  T_l_w = 0. */
int32_t ptz_ba_batch_set_state_pix2ray(ptz_ba_batch* b, const double* cam, const double* rkinv);
/* Diagnostic for the tests: FNV-1a hash over the batch's structure arrays (observations in the library's order, ray and
 * camera lists, camera pairs, entries, runs, weights, ray order), real extents only -- two batches that hash alike solve alike. */
int32_t ptz_debug_batch_structure_hash(ptz_ba_batch* b, uint64_t* hash);
/* ... and the stored initial rays of the batch in the caller's ray numbering, ray [3 * sum n_ray] (at the extents the batch was
 * created with), e.g. what ptz_ba_batch_set_state_pix2ray computed. */
int32_t ptz_debug_batch_initial_rays(ptz_ba_batch* b, double* ray);

/* Dense SPD solve used for the reduced camera system, exposed for parity tests and micro-benchmarks:
 * solves A x = rhs for `count` independent n x n systems (host, row-major, lower triangle read).
 * Returns per-system status in fail[count] (1 = not positive definite). */
int32_t ptz_chol_solve_batch(int32_t count, int32_t n, const double* A, const double* rhs, double* x, int32_t* fail,
                             int32_t device_id, double* device_ms);

/* Measured FP64 matrix-core rate of the device (v_mfma_f64_16x16x4_f64 from registers, every wave slot busy): the
 * number the reduced-camera solve is priced against in the roofline reports. */
int32_t ptz_mfma_f64_peak(int32_t device_id, double* tflops);
/* Host logic only (no device needed; used by the CPU tests): the elimination order ptz_ba_batch_create would give a reduced
 * camera system of nt 64-column tiles whose lower-triangular tile adjacency is mask[nt*nt] (row-major, non-zero = coupled),
 * tiles >= first_dense kept last.  perm[t] = position of tile t; lanes[0], lanes[1] = tiles of the two lanes that are
 * factored side by side (positions [0, lanes[0]) and [lanes[0], lanes[0] + lanes[1])).  sched (may be NULL): [nt * 4] the
 * block columns (positions, -1 = none) each step of the factorisation handles, n_steps[0] their number -- read off the filled
 * structure for a dissected order, one column per step otherwise.  Returns 1 if a dissection was chosen, 0 for the natural
 * order (perm = identity). */
int32_t ptz_ba_plan_tile_order(int32_t nt, int32_t first_dense, const uint8_t* mask, int32_t* perm, int32_t* lanes,
                               int32_t* sched, int32_t* n_steps);

/* Measured HBM rates of the device in GB/s: streaming read of 4 GB, and copy of 4 GB counted as read + write. */
int32_t ptz_hbm_bandwidth(int32_t device_id, double* read_gbps, double* copy_gbps);

/* ------------------------------------------------------------------------------------------------
 * Single-view LM, batched over queries
 *   replaces KRTOptimizer::Add2d2dConstraints + Solve (krt_optimizer.cc:265-348, 385-404), the loop
 *   body of run_ptz_reloc.cc:68-118 and RegisterNextImage (ptz_incremental_optimizer.cc:377-418).
 * Query q owns matches [match_ptr[q], match_ptr[q+1]).  cam_ref: reference camera (world frame),
 * cam_cur: in = initial current camera in world frame (SetInitParams), out = refined camera in world
 * frame (ObtainRefinedCameraParams, krt_optimizer.cc:535-567), written only when accepted[q] = 1
 * (CheckResults, krt_optimizer.cc:504-533, with max_reproj_error).
 * ------------------------------------------------------------------------------------------------ */
int32_t ptz_krt_solve_batch(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                            const double* cam_ref, double* cam_cur, int32_t factor_type, double max_reproj_error,
                            const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted,
                            double* device_ms);

/* The same solve with 2D-3D constraints added to every query:
 *   replaces KRTOptimizer::Add2d3dConstraints (krt_optimizer.cc:350-383; Factor2d3dDist / Factor2d3dFxfyDist,
 *   :200-249) on top of Add2d2dConstraints -- never called by the reference's tools.
 * Query q owns points [point_ptr[q], point_ptr[q+1]): pts2d = pixels (2 x f32), pts3d = WORLD points (3 x f64); the
 * library moves them into the local frame of the query's reference camera as the reference does (:357-362) and
 * projects them as cv::projectPoints does, including its (k1,k2,p1,p2,k3) reading of the stored distortion.
 * num_residuals counts both kinds.  point_ptr = NULL is ptz_krt_solve_batch. */
int32_t ptz_krt_solve_batch_2d3d(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                                 const int64_t* point_ptr, const float* pts2d, const double* pts3d, const double* cam_ref,
                                 double* cam_cur, int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt,
                                 ptz_lm_summary* summaries, int32_t* accepted, double* device_ms);

/* The same launch on queries that are already resident in HBM (matches produced on the device, a serving loop that keeps its
 * buffers): every d_* argument is a DEVICE pointer with the layout of its host counterpart above, d_point_ptr = NULL for no
 * 2D-3D constraints.  The kernel is enqueued on `hip_stream` (a hipStream_t; NULL = the default stream) of the device that
 * OWNS d_cam_cur (hipPointerGetAttributes) and the call returns without synchronising: results (d_cam_cur, d_summaries,
 * d_accepted) are valid once the stream has passed this point.  PTZ_EINVAL if the stream belongs to another device than the
 * buffers, or if opt->device_id names a different device explicitly (a non-zero value; 0 is the untouched default and is
 * not taken as a request for device 0).  The CSR offsets are not validated (they live on the device).
 * Like every entry point of this library the call leaves the calling thread's current HIP device as it found it. */
int32_t ptz_krt_solve_batch_device(int32_t n_query, const int64_t* d_match_ptr, const float* d_uv_ref, const float* d_uv_cur,
                                   const int64_t* d_point_ptr, const float* d_pts2d, const double* d_pts3d,
                                   const double* d_cam_ref, double* d_cam_cur, int32_t factor_type, double max_reproj_error,
                                   const ptz_lm_options* opt, ptz_lm_summary* d_summaries, int32_t* d_accepted, void* hip_stream);

/* Registration attempts over match tables that stay on the device:
 *   replaces the loop body of PtzIncrementalOptimizer::RegisterNextImage (ptz_incremental_optimizer.cc:377-418) -- one
 *   KRTOptimizer per table entry (registered image i -> image j) -- for callers that try many entries of the SAME tables again
 *   and again while only the cameras change (the incremental pipeline: ~5 000 attempts per 200-view rig over ~8 500 entries).
 * ptz_krt_table_create copies a table's matches to device `device_id` once: entry e owns matches [match_ptr[e], match_ptr[e+1])
 * (match_ptr[0] = 0), uv_ref / uv_cur as in ptz_krt_solve_batch.  ptz_krt_solve_attempts then solves query q = entry
 * attempts[q].entry of attempts[q].table with the cameras cam_ref / cam_cur [15 n_query] (in / out as in ptz_krt_solve_batch); the
 * tables of one call may differ (several rigs in one launch) but live on one device, where the launch runs (opt->device_id, if
 * non-zero, must name it).  Results are bit-identical to ptz_krt_solve_batch on the same matches. */
typedef struct ptz_krt_table ptz_krt_table;
typedef struct ptz_krt_attempt { const ptz_krt_table* table; int32_t entry; } ptz_krt_attempt;
int32_t ptz_krt_table_create(int32_t n_entry, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur, int32_t device_id,
                             ptz_krt_table** out);
void ptz_krt_table_destroy(ptz_krt_table* table);
int32_t ptz_krt_solve_attempts(int32_t n_query, const ptz_krt_attempt* attempts, const double* cam_ref, double* cam_cur,
                               int32_t factor_type, double max_reproj_error, const ptz_lm_options* opt, ptz_lm_summary* summaries,
                               int32_t* accepted, double* device_ms);

/* The queries over the devices device_ids[0 .. n_devices) of one node, from one process: contiguous chunks of about equal
 * total match count, one host thread per device, results written into the caller's arrays in query order.  Host pointers
 * as in ptz_krt_solve_batch_2d3d (point_ptr = NULL: no 2D-3D constraints); opt->device_id is ignored. */
int32_t ptz_krt_solve_batch_sharded(int32_t n_query, const int64_t* match_ptr, const float* uv_ref, const float* uv_cur,
                                    const int64_t* point_ptr, const float* pts2d, const double* pts3d, const double* cam_ref,
                                    double* cam_cur, int32_t factor_type, double max_reproj_error, const int32_t* device_ids,
                                    int32_t n_devices, const ptz_lm_options* opt, ptz_lm_summary* summaries, int32_t* accepted);

#ifdef __cplusplus
}
#endif
#endif /* PTZ_CALIB_AMD_H */
