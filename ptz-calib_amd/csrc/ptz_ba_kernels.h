// ptz_ba_kernels.h -- device side of the PTZ-IBA bundle adjustment (gfx950): the structures the kernels read by value and
// every kernel of one Levenberg-Marquardt pass (see the pass outline at the top of ptz_ba.hip, which is the only file that
// includes this one; the host side -- batch creation, the pass loop, the C-ABI -- lives there).
#ifndef PTZ_BA_KERNELS_H
#define PTZ_BA_KERNELS_H
#include "ptz_common.h"
#include "ptz_factor.h"

namespace ptz {

namespace {

constexpr int RAY_BLOCK = 1024;  // most rays per workgroup in the ray-centric kernels (Dev::ray_block is the batch's actual value:
                                 // a few scenes use small workgroups so that one rig's 13 k rays spread over a hundred compute units)
constexpr int SCHUR_F_ROW = 9;     // doubles per observation in k_schur_f's LDS table: 8, at an odd pitch
constexpr int EZS = 18;           // doubles per ray in the record k_schur gathers once per observation: E (6), z = E g_r (3), the
                                  // functor's point Xn (3), a = sqrt(w) |X|^-1 s_r (3), sqrt(w) -- eight 16-byte pieces (e_piece) written by k_ray_prep every
                                  // pass; batches of k_schur_f (Dev::e_fold) carry {E, z'', Xn, E''} in nine pieces instead
// LDS / global stride of a camera block and of a candidate block: Dims<TYPE>::CBS (35 doubles: odd -> no same-field bank
// conflicts; 41 with the displacement block) and Dims<TYPE>::CDS (19; 41)
// W row stride: see Dims<TYPE>::WS

struct SceneDev {
  int n_cam, n_ray, n_obs, n_pair;
  int cam_off, ray_off, obs_off, pair_off;
  int ent_off;   // first camera-pair entry
  int run_off;   // first run of k_schur (global index)
  int part_off;  // first partial-sum slot (one per WAVE of 64 rays, plus one for the 2D-3D terms: the reduction tree of the
                 // per-ray sums does not depend on the workgroup size, so a scene's bits do not depend on the batch it is in)
  int n_chunk;   // ceil(n_ray / Dev::ray_block)
  int n_wave;    // ceil(n_ray / 64)
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
  int ray_lin_ready;  // the ray blocks of the current point were left by k_eval's second pass (no k_lin_ray needed for this linearisation)
  double cand_norm2;  // |x_c|^2 of the last evaluated candidate (k_lm_post): |x|^2 of the point once the step is accepted
  int chain_timeouts; // linear solves in which a hand-over of the one-launch factorisation did not arrive within its bounded wait
                      // (CholBatch::fail bit 1): the step was handled as an invalid one, and the host reports the solve as PTZ_ENODEVICE
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
  const float2* cam_uv; // [total_obs] the pixel, camera-major: the camera pass streams (pixel, ray id) instead of gathering them
  const int* pair_ci;   // scene-local camera ids, ci >= cj
  const int* pair_cj;
  const int* pair_ptr;  // [total_pair + n_scene] per scene n_pair + 1 entries, global entry index
  const int* cam_pair;  // [total_cam + n_scene] per scene n_cam + 1 entries: scene-local pair range of each camera ci
  const unsigned* ent;  // low 16 bits: position of obs a in ci's observation list; high 16: position of obs b in cj's; ci > cj only
  const int* pair_brow; // first W row of camera cj of every pair (k_schur_w only)
  // runs of k_schur: a camera's entries cut into at most schur_threads<TYPE>() pieces of equal length that never straddle two pairs
  const int* cam_run;   // [total_cam + n_scene] per scene n_cam + 1 entries: scene-local run range of each camera ci
  const uint2* run_rec; // [total_run] {first entry (global index), pair (index among the camera's pairs, 16 bits) | entries << 16}
  const int* pair_run;  // [total_pair + n_scene] per scene n_pair + 1 entries: scene-local first run of every pair
  int schur_ent_cap;    // entries of the batch's largest camera (k_schur stages a camera's entry list in LDS, 2 bytes per entry)
  const double* ray_w;
  // state: two buffers, LmState.cur selects the current one
  double* cam_x;  // [2][total_cam][15]
  double* ray_x;  // [2][total_ray][3]
  size_t cam_stride, ray_stride;
  const double* cam_x0;
  const double* ray_x0;
  double* dsp_x;  // [2][total_cam][3] PTZRayDistDisp: every camera's copy of its scene's displacement block (else nullptr)
  size_t dsp_stride;
  // per-camera blocks
  double* camblk;    // [total_cam][CBS]   at x          (rows already carry the odd LDS pitch: a workgroup stages its scene's
  double* candblk;   // [total_cam][CDS]   at the candidate   table with a flat, 16-byte-per-lane copy)
  size_t camblk_stride;  // camblk is double-buffered like the state: [2][camblk_stride]; k_cam_update fills the candidate's half,
                         // so an accepted step needs no camera pass before it is re-linearised
  double* scale_c;   // [total_cam][NC]
  double* scale_r;   // [total_ray][3]
  // The camera side of the linearisation is DOUBLE-BUFFERED like the state (two halves of lin_cams cameras each, LmState::cur
  // selects the current point's): k_lin_cam evaluates the CANDIDATE's blocks into the other half before the step is judged
  // (Dev::spec_lin), so that accepting a step is a flip of `cur` and one control point per pass closes the step and opens the next
  double* U;         // [2][total_cam][NC*NC]
  double* gc;        // [2][total_cam][NC]
  double* costc;     // [2][total_cam]
  double* diag_c;    // [2][total_cam][NC]
  int lin_cams;      // cameras per half (the batch's total, not rebased per scene group)
  int spec_lin;      // 1: k_lin_cam linearises the candidate (half cur ^ 1) of every active scene; 0: the current point where LmState asks for it
  double* dc;        // [total_cam][NC] scaled-space camera step
  double* dct;       // [total_cam][NC | 1] the same step as k_eval applies it per observation: [intrinsic components | Jl v_rot]
  double* V;         // [total_ray][6]
  double* gr;        // [total_ray][3]
  double* diag_r;    // [total_ray][3]
  int e_fold;        // 1 (PTZRay batches of k_schur_f): pieces 3..8 hold the PRE-SCALED record {z'' = c z, Xn, E'' = diag(c) E diag(c)}, c = sqrt(w) a: six
                     // pieces per observation to gather instead of eight (k_ray_prep); pieces 0..2 stay the plain E of k_eval
  size_t e_stride;   // rays of the batch: E is stored as EZS / 2 PLANES of 16-byte pieces, piece p of ray gj at ((double2*)E)[p * e_stride + gj]
  double* E;         // [total_ray][EZS] per ray: E = (V + D^2)^-1 (6 unique entries), z = E g_r (3), then what k_schur needs to
                     // rebuild an observation's Jacobians (Xn, the ray-side factors, sqrt(w)) -- ONE record, one gather per observation
  double* W;         // [total_obs][Dims::WS] rows W_a = Jc^T Jr (NW x 3), camera-major
  double* Tbuf;      // [total_obs][NW * 3] or nullptr: T_a = W_a E rows of k_schur when a camera's do not fit in LDS
  double* rayrec;    // [total_ray][8] {X[3], Jacobi scale[3], weight, 0}: what the camera pass needs of a ray, one 64-byte sector
  double* camstep;   // [total_cam][2] {|x_i - x_c,i|^2, |x_c,i|^2} over the camera's parameter blocks that are in the problem: written with
                     // the candidate (k_cam_update, or k_eval's prologue when the camera update is folded into it), summed by k_lm_post
  double* cam_gmax;  // [2][total_cam] max_k |g_i[k] / s_i[k]| of the camera's gradient block (k_lin_cam), for k_lm_pre
  // LM control without launches of its own (launch shapes of a few scenes, Dev::fuse_ctl): the LAST workgroup of a scene to
  // finish k_eval closes the step (lm_post_wave), the last to finish k_lin_cam opens the next iteration (lm_pre_wave).
  int* tail_cnt;     // [n_scene][2] workgroups of the scene that have finished k_eval / k_lin_cam in this pass (zero between passes)
  int fuse_ctl;      // 1: this launch shape has no k_lm_pre / k_lm_post / k_cam_update launches
  double* partial;   // [total_wave + n_scene][4] per wave of rays {model cost change, candidate cost, |x - x_c|^2, |x_c|^2} of k_eval
                     // (one extra slot per scene for the 2D-3D terms)
  double* partial_lin;  // [total_wave][2] per wave of rays {max |g_r / scale|, |x|^2} of k_lin_ray
  // V, gr, rayrec and partial_lin are DOUBLE-BUFFERED like the state (LmState.cur selects the half of the current point): k_eval's second
  // pass leaves the ray-side linearisation of the candidate in the other half, which an accepted step makes the current one
  size_t V_stride, gr_stride, rayrec_stride, plin_stride;  // doubles between the halves
  int ray_block;     // rays per workgroup of k_lin_ray / k_ray_prep / k_eval in this batch (multiple of 64, <= RAY_BLOCK)
  // 2D-3D annotation residuals (georeferencing); per-scene arrays below are indexed by the GLOBAL scene index
  const float2* o3_uv;  // [total_o3]
  const double* o3_xyz; // [total_o3][3] world points
  const int* o3_cam;    // [total_o3] scene-local camera id
  double* tlw_x;     // [2][n_scene_total][6]
  size_t tlw_stride;
  double* tlwblk;    // [2][n_scene_total][TLWBLK] at x / at the candidate, selected by LmState.cur like camblk
  size_t tlwblk_stride;
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
  const unsigned char* cam_flag;  // [total_cam] bit 0: this camera's intrinsics block is counted in |x| (one per group);
                                  // bit 1: so is its copy of the displacement block (one camera per scene)
  const unsigned char* grp_cls;   // [total groups] at scene.grp_off + g: which slots the group shares -- 0 intrinsics, 1 displacement
  double* gfold;         // [total_cam][NC] gradient with the shared slots folded onto the representative
  // LM
  LmState* lm;
  int* active;       // [n_scene]
  int* ray_fail;     // [n_scene]
  // Pass pipeline control of this scene group.  The host enqueues LM passes ahead of the device and never waits for one:
  //   grp_ctl[0]  scenes of the group still active (device memory; atomically decremented by the block that retires a scene)
  //   grp_ctl[1]  LM passes whose step-evaluation kernel has started (device memory)
  //   host_ctl[0] copy of grp_ctl[1] in pinned host memory (the host throttles its run-ahead on it)
  //   host_ctl[1] set to 1 by whoever retires the group's last scene (the host stops enqueuing passes when it sees it)
  //   grp_ctl[2]  entries of the compacted scene list `act` (rebuilt by k_compact after every k_lm_pre); host_ctl[2] mirrors it
  int* grp_ctl;
  int* host_ctl;
  int debug_stall;   // PTZ_BA_DEBUG_STALL (tests of the host's watchdog): k_lm_post stops posting progress after this many passes (0: never)
  // Compacted launches: once few scenes of a large batch are still active, the host enqueues passes whose grids cover only
  // `slots` scenes; blockIdx.y (or .x for the one-workgroup-per-scene kernels) is then a slot and act[slot] the scene.  The
  // host sizes those grids from a stale -- hence larger or equal -- count.  use_act = 0: slot == scene (full-size launches).
  int* act;
  int use_act;
  Opt opt;
  // reduced camera system
  CholBatch chol;
  double* yc;        // [n_scene][np]
  // Elimination order of the reduced system: a permutation of its 64-column tiles, [n_scene][np / 64] (nullptr: natural
  // order).  Row / column c of the system "as the cameras number it" (camera i: columns i NC ...) lives at
  // tperm[c / 64] * 64 + c % 64; every kernel that writes the system or reads its solution goes through sys_col().  The host
  // picks the order from the tile graph (ptz_ba.hip plan_dissection): arcs of the ring first, separators last.
  const int* tperm;
};

__device__ __forceinline__ int sys_col(const Dev& d, int sc, int c)
{
  return d.tperm ? d.tperm[(size_t)sc * (d.chol.np / CHOL_NB) + c / CHOL_NB] * CHOL_NB + c % CHOL_NB : c;
}
// element (r, c) of the symmetric system, r and c already mapped, in its lower-triangular storage
__device__ __forceinline__ double& sys_at(double* A, int np, int r, int c) { return r >= c ? A[(size_t)r * np + c] : A[(size_t)c * np + r]; }


// The rays' records {E (6), z (3), Xn (3), a (3), sqrt(w)} as eight planes of 16-byte pieces (round 5; one 128-byte record per
// ray before): the ray-centric kernels (k_ray_prep's stores, k_eval's reads of E) become unit-stride -- ray preparation 20.7 ->
// 17.2 ms, evaluation 41 -> 38.7 ms per C4 solve -- and a camera's rays are dense clusters of the internal ray order (by track
// length, then first camera), so the 64 lanes of k_schur's gather touch ~21 lines per plane.  (k_schur itself does not gain: its
// 24 record loads per lane take 5-9 us to ISSUE on a loaded chip with either layout -- the gather runs at what a compute unit
// can keep in flight, NOTES_r05.)
__device__ __forceinline__ double2* e_piece(const Dev& d, int p, size_t gj) { return reinterpret_cast<double2*>(d.E) + (size_t)p * d.e_stride + gj; }

__device__ __forceinline__ const double* cur_cam(const Dev& d, const SceneDev& s, const LmState& st)
{
  return d.cam_x + (size_t)st.cur * d.cam_stride + (size_t)s.cam_off * 15;
}
__device__ __forceinline__ const double* cur_ray(const Dev& d, const SceneDev& s, const LmState& st)
{
  return d.ray_x + (size_t)st.cur * d.ray_stride + (size_t)s.ray_off * 3;
}

__device__ __forceinline__ double* cur_camblk(const Dev& d, const LmState& st) { return d.camblk + (size_t)st.cur * d.camblk_stride; }
// the ray-side linearisation (V, g_r, the rays' records for the camera pass, the per-wave gradient / |x| partials): half `h`
__device__ __forceinline__ double* lin_V(const Dev& d, int h) { return d.V + (size_t)h * d.V_stride; }
__device__ __forceinline__ double* lin_gr(const Dev& d, int h) { return d.gr + (size_t)h * d.gr_stride; }
__device__ __forceinline__ double* lin_rayrec(const Dev& d, int h) { return d.rayrec + (size_t)h * d.rayrec_stride; }
__device__ __forceinline__ double* lin_partial(const Dev& d, int h) { return d.partial_lin + (size_t)h * d.plin_stride; }
__device__ __forceinline__ double* cur_tlwblk(const Dev& d, const LmState& st) { return d.tlwblk + (size_t)st.cur * d.tlwblk_stride; }

__device__ __forceinline__ int scene_of_slot(const Dev& d, int slot)
{
  if (!d.use_act) return slot;
  return slot < d.grp_ctl[2] ? d.act[slot] : -1;
}

// A scene leaves the pass pipeline (one thread of the scene's LM block calls this, once per scene and solve).
__device__ __forceinline__ void retire_scene(const Dev& d, int sc)
{
  d.active[sc] = 0;
  if (atomicSub(&d.grp_ctl[0], 1) == 1)
    __hip_atomic_store(&d.host_ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// TYPE = factor (0 PTZRay, 1 PTZRayDist, 2 PTZRayFxfyDist, 3 PTZRayDistDisp) + 4 * has3d.
//   NW columns of a camera carry a non-zero 2D-2D Jacobian: [f, (k1), r1, r2, r3]; PTZRayFxfyDist [fx, fy, k1, r1, r2, r3];
//   PTZRayDistDisp [f, k1, r1, r2, r3, d0, d1, d2] -- the last three are the camera's copy of the ONE displacement block of
//   the problem (disp_param_, ptzray_optimizer.cc:655), made one parameter by a group of all cameras (see k_fold_system).
//   NC free camera parameters: without annotations the same set (the reference's always-zero fy column of PTZRay /
//   PTZRayDist is not materialised); with 2D-3D annotation residuals fy becomes live (Reproj2d3dFactor reads it,
//   ptzray_optimizer.cc:273): [f, fy, (k1), r1, r2, r3, (d0, d1, d2)], and the 6-dof T_l_w block joins the reduced system.
template <int TYPE> struct Dims {
  static constexpr int FACTOR = TYPE % 4, HAS3D = TYPE / 4;
  static constexpr int FXFY = FACTOR == 2;  // fy is a 2D-2D column
  static constexpr int DISP = FACTOR == 3;  // displacement block
  static constexpr int F3 = FACTOR ? 1 : 0;  // Reproj2d3dFactor variant: k1 free or not (same functor for Dist / FxfyDist)
  static constexpr int NW = 4 + (FACTOR >= 1) + FXFY + 3 * DISP;
  static constexpr int NC = NW + (HAS3D && !FXFY);
  static constexpr int NG = 6 * HAS3D;  // size of the global (tlw) block
  static constexpr int RW = NW - 3 - 3 * DISP;  // first rotation column among the 2D-2D columns
  static constexpr int RC = NC - 3 - 3 * DISP;  // ... among the NC free parameters
  // doubles per observation row of W = Jc^T Jr (NW x 3), rounded up to a 16-byte multiple: 12 (96 B) / 16 (128 B).
  // Measured on MI355X: unpadded 96-B rows beat 128-B-aligned rows (less write/stream traffic outweighs line straddling).
  static constexpr int WS = (NW * 3 + 1) & ~1;
  static constexpr int CAMBLK = DISP ? ptz::CAMBLK_DISP : ptz::CAMBLK, CBS = CAMBLK + 1;
  static constexpr int CANDBLK = DISP ? ptz::CAMBLK_DISP : ptz::CANDBLK, CDS = CANDBLK + 1;  // the displacement sits behind the scales
  // position of 2D-2D column k inside the NC block
  static __host__ __device__ constexpr int pos(int k) { return NC != NW ? (k == 0 ? 0 : k + 1) : k; }
  // index of free parameter k of the NC block in the Camera 15-vector; 15, 16, 17 = the displacement block (Dev::dsp_x)
  static __host__ __device__ constexpr int at(int k)
  {
    // PTZRay: f r1 r2 r3 | PTZRayDist: f k1 r.. | +3D: f fy r.. | f fy k1 r.. | PTZRayFxfyDist (with or without 3D): fx fy k1 r..
    // PTZRayDistDisp: f k1 r.. d.. | +3D: f fy k1 r.. d..
    constexpr int FYL = (NC != NW) || FXFY;  // fy occupies slot 1
    return k == 0 ? 0 : (FYL && k == 1) ? 1 : (FACTOR && k == 1 + FYL) ? 10 : k < RC + 3 ? 4 + (k - RC) : 15 + (k - RC - 3);
  }
};

// LM control as one wave per scene (defined with k_lm_pre / k_lm_post below; also run from the tails of k_lin_cam / k_eval)
template <int TYPE> __device__ __forceinline__ void lm_pre_wave(const Dev& d, int sc);
template <int TYPE> __device__ __forceinline__ void lm_post_wave(const Dev& d, int sc);
template <int TYPE> __device__ __forceinline__ void lm_step_wave(const Dev& d, int sc);
__device__ __forceinline__ bool tail_last_workgroup(int* cnt, int expected, int* lds_flag);
__device__ __forceinline__ void tail_store(double* p, double v);
__device__ __forceinline__ double tail_load(const double* p);
__device__ __forceinline__ void post_progress(const Dev& d);

__device__ __forceinline__ void fill_camblk(const double* c15, double* cb, bool with_jl, const double* dsp = nullptr)
{
  if (dsp) { cb[CB_D] = dsp[0]; cb[CB_D + 1] = dsp[1]; cb[CB_D + 2] = dsp[2]; }
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
  double* cb = cur_camblk(d, st) + (size_t)(s.cam_off + i) * Dims<TYPE>::CBS;
  fill_camblk(cur_cam(d, s, st) + (size_t)i * 15, cb, true,
              Dims<TYPE>::DISP ? d.dsp_x + (size_t)st.cur * d.dsp_stride + (size_t)(s.cam_off + i) * 3 : nullptr);
#pragma unroll
  for (int k = 0; k < NC; ++k) cb[CB_S + k] = d.scale_c[(size_t)(s.cam_off + i) * NC + k];
  if (Dims<TYPE>::HAS3D && i == 0) {
    const double* t = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    double* tb = cur_tlwblk(d, st) + (size_t)s.idx * TLWBLK;
    double R[9], Jl[9];
    const double rv[3] = {t[0], t[1], t[2]};
    rodrigues(rv, R);
    so3_left_jacobian(rv, Jl);
    for (int k = 0; k < 9; ++k) { tb[k] = R[k]; tb[9 + k] = Jl[k]; }
    tb[18] = t[3]; tb[19] = t[4]; tb[20] = t[5];
  }
}

// Stage `count` doubles of a scene's camera table (global rows already at the LDS pitch) into LDS with 16-byte loads, eight
// in flight per thread.  The source starts at an 8-byte boundary, so the copy starts at the 16-byte boundary at or below it:
// the table then sits `return value` (0 or 1) doubles into `dst`, which needs room for count + 4 doubles.
template <int U>
__device__ __forceinline__ int stage_flat(const double* __restrict__ src, double* dst, int count)
{
  const int shift = (int)((reinterpret_cast<uintptr_t>(src) >> 3) & 1);
  const double2* s2 = reinterpret_cast<const double2*>(src - shift);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const int n2 = (count + shift + 1) >> 1;
  const int bd = blockDim.x;
  for (int base = 0; base < n2; base += bd * U) {
    // Unconditional loads, all in flight together (hipcc sinks a conditional load next to its store and parks the buffer in
    // scratch: one exposed round trip per 16 bytes).  Lanes past the end of the table re-read its tail instead, each its own
    // element, so that no single line is hammered.
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * bd + (int)threadIdx.x;
      v[u] = s2[idx < n2 ? idx : max(n2 - 1 - (int)threadIdx.x, 0)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {  // unconditional as well (a conditional store sends v[] through scratch memory): lanes past the
      const int idx = base + u * bd + (int)threadIdx.x;  // end write the spare 16 bytes behind the table
      d2[min(idx, n2)] = v[u];
    }
  }
  return shift;
}

// Walk the observations [a0, a1) of one ray.  Large workgroups (batches: 16 waves per compute unit hide the latency) read
// the 16-byte records where they are needed.  SMALL workgroups (a few scenes, two to four waves per compute unit) would pay
// one exposed memory round trip per observation, so they fetch eight records at a time -- all loads in flight together --
// park them in a thread-private LDS slot and run the same body over them: same arithmetic, same order, same bits.
template <bool SMALL, typename Body>
__device__ __forceinline__ void for_each_obs(const Dev& d, int a0, int a1, float4* obsbuf, Body&& body)
{
  if (!SMALL) {
    // the next record is asked for before the current one is worked on (unconditional: past the end it re-reads the last)
    float2 uvn = d.obs_uv[a0];
    int cn = d.obs_cam[a0];
    for (int a = a0; a < a1; ++a) {
      const float2 uv = uvn;
      const int c = cn;
      const int an = min(a + 1, a1 - 1);
      uvn = d.obs_uv[an]; cn = d.obs_cam[an];
      body(uv, c);
    }
    return;
  }
  constexpr int P = 8;
  for (int ab = a0; ab < a1; ab += P) {
    float2 uvr[P];
    int cr[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {  // unconditional (clamped) so that the loads are issued back to back
      const int aa = min(ab + u, a1 - 1);
      uvr[u] = d.obs_uv[aa]; cr[u] = d.obs_cam[aa];
    }
#pragma unroll
    for (int u = 0; u < P; ++u)  // (slots past the end of the track hold a copy of its last record and are not read)
      obsbuf[u * blockDim.x + threadIdx.x] = make_float4(uvr[u].x, uvr[u].y, __int_as_float(cr[u]), 0.f);
    const int ne = min(P, a1 - ab);
    for (int u = 0; u < ne; ++u) {
      const float4 r = obsbuf[u * blockDim.x + threadIdx.x];
      body(make_float2(r.x, r.y), __float_as_int(r.z));
    }
  }
}
constexpr int OBS_PREFETCH_BYTES = 8 * 16;  // per thread of a SMALL workgroup

// ---- lin_ray: per-ray linearisation ---------------------------------------------------------------------
// thread = ray: for every observation of the ray evaluate residual + Jacobians, apply sqrt(w) and the
// Jacobi scales, accumulate V = sum Jr^T Jr and g_r = sum Jr^T r.  (The W_a = Jc^T Jr rows are written by k_lin_cam, whose
// lanes walk a camera's observations in the order of its W rows: one sequential stream instead of a 96-byte scatter.)
// GTAB: the scene's camera table does not fit in LDS (more than ~340 cameras): the blocks are read where they lie, through the
// caches (the reference has no cap on the number of cameras, ptzray_optimizer.cc:799-885).
template <int TYPE, bool SMALL, bool GTAB>
__global__ __launch_bounds__(SMALL ? 256 : RAY_BLOCK) void k_lin_ray(Dev d)
{
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize || st.ray_lin_ready || (int)(blockIdx.x * blockDim.x) >= s.n_ray) return;  // (ray_lin_ready: k_eval left this linearisation)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const double* tab;
  float4* obsbuf;  // SMALL only
  if constexpr (GTAB) {
    tab = cur_camblk(d, st) + (size_t)s.cam_off * CBS;
    obsbuf = reinterpret_cast<float4*>(lds);
  }
  else {
    tab = lds + stage_flat<SMALL ? 16 : 8>(cur_camblk(d, st) + (size_t)s.cam_off * CBS, lds, s.n_cam * CBS);
    obsbuf = reinterpret_cast<float4*>(lds + ((s.n_cam * CBS + 5) & ~1));
    __syncthreads();
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double gm = 0, xn = 0;  // this ray's share of the gradient max-norm and of |x|^2 (k_lm_pre)
  if (j < s.n_ray) {
  const int gj = s.ray_off + j;
  const double* X = cur_ray(d, s, st) + (size_t)j * 3;
  const double Xr[3] = {X[0], X[1], X[2]};
  const double sr[3] = {d.scale_r[(size_t)gj * 3], d.scale_r[(size_t)gj * 3 + 1], d.scale_r[(size_t)gj * 3 + 2]};
  const double sw = sqrt(d.ray_w[gj]);
  const int* rp = d.ray_ptr + s.ray_off + s.idx;
  double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for_each_obs<SMALL>(d, rp[j], rp[j + 1], obsbuf, [&](float2 uv, int ci) {
    const double* cb = tab + ci * CBS;
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
  });
#pragma unroll
  for (int k = 0; k < 6; ++k) lin_V(d, st.cur)[(size_t)gj * 6 + k] = V[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) lin_gr(d, st.cur)[(size_t)gj * 3 + k] = g[k];
  {  // the camera pass (next launch) gathers this ray once per observation: one aligned 64-byte record instead of three arrays
    double* rr = lin_rayrec(d, st.cur) + (size_t)gj * 8;
    rr[0] = Xr[0]; rr[1] = Xr[1]; rr[2] = Xr[2]; rr[3] = sr[0]; rr[4] = sr[1]; rr[5] = sr[2]; rr[6] = d.ray_w[gj]; rr[7] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { gm = fmax(gm, fabs(g[k] / sr[k])); xn += Xr[k] * Xr[k]; }
  }
  gm = wave_max(gm);
  xn = wave_sum(xn);
  if ((threadIdx.x & 63) == 0 && j < s.n_ray) {
    double* pl = lin_partial(d, st.cur) + (size_t)(s.part_off - s.idx + (j >> 6)) * 2;  // (part_off counts one extra slot per preceding scene)
    pl[0] = gm; pl[1] = xn;
  }
}

// ---- lin_cam: per-camera blocks -------------------------------------------------------------------------
// wave = camera: lanes stride over the camera's observation list, U = sum Jc^T Jc, g_c = sum Jc^T r,
// cost = 1/2 sum w |r|^2, reduced with a fixed butterfly.
// WRITE_W (ptz_ba_batch_linearize only, for the parity tests): every lane also stores the row W_a = Jc^T Jr of its observation
// (row index = position in the camera-major list, so a wave writes one contiguous stretch of W).  The solver itself never
// materialises W: k_schur rebuilds the rows from the rays (round 2 wrote 96 B per observation here and gathered them back
// 4.7 times each in k_schur).
template <int TYPE, bool WRITE_W>
__global__ __launch_bounds__(256) void k_lin_cam(Dev d)
{
#ifdef PTZ_LINCAM_STAMPS  // probe builds only (block 0, thread 0; 100 MHz wall clock)
  long long lc_t[6] = {0, 0, 0, 0, 0, 0};
#define LC_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) lc_t[i] = wall_clock64(); } while (0)
#else
#define LC_STAMP(i) do { } while (0)
#endif
  LC_STAMP(0);
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || (int)blockIdx.x * 4 >= s.n_cam) return;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int gi = s.cam_off + i;
  // Which point: the candidate's (Dev::spec_lin: every active scene, before its step is judged -- camera blocks, ray records and
  // the results all in the half that LmState::cur does NOT select) or the current one (iteration zero; scenes with annotation
  // residuals or shared blocks, whose later kernels add to these blocks: after the step was accepted)
  const int hh = d.spec_lin ? (st.cur ^ 1) : st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  const double* camblk_h = d.camblk + (size_t)hh * d.camblk_stride;
  // (no wave leaves before the end: with Dev::fuse_ctl the workgroup meets again at the tail below)
  if ((d.spec_lin || st.need_linearize) && i < s.n_cam) {
  double cb[CAMBLK];
#pragma unroll
  for (int k = 0; k < CAMBLK; ++k) cb[k] = camblk_h[(size_t)gi * CBS + k];
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  LC_STAMP(1);
  double U[NW * (NW + 1) / 2], g[NW], cost = 0;
#pragma unroll
  for (int k = 0; k < NW * (NW + 1) / 2; ++k) U[k] = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) g[k] = 0;
  // The W rows of 64 consecutive observations are one contiguous stretch of 64 * WS doubles: every lane parks its row in a
  // wave-private LDS strip, then the wave stores the stretch with unit-stride lanes -- whole 128-byte lines per store
  // instruction.  (Lane-private 96-byte row stores fill the lines piecemeal and make the L2 fetch them first.)
  constexpr int WS = Dims<TYPE>::WS, NT = NW * 3, WP = WS + 1;  // odd LDS pitch
  __shared__ double wstrip[WRITE_W ? 4 : 1][WRITE_W ? 64 * WP : 1];
  double* ws = wstrip[WRITE_W ? threadIdx.x >> 6 : 0];
  const int q_end = cp[i + 1];
  // Two memory round trips stand in front of an observation's arithmetic: its (pixel, ray id) record, then the ray's record
  // that the id points to.  Both are fetched ahead -- the ids two trips of 64 observations ahead, the ray records one -- with
  // unconditional loads (indices past the camera's end are clamped and their data unused), so a trip waits for neither.
  typedef double d8 __attribute__((ext_vector_type(8)));
  auto qclamp = [&](int q) { return max(min(q, q_end - 1), 0); };
  auto load_rr = [&](int rid) {
    const double2* r2 = reinterpret_cast<const double2*>(lin_rayrec(d, hh) + (size_t)rid * 8);
    const double2 a = r2[0], b = r2[1], c = r2[2], e = r2[3];
    d8 v; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = e.x; v[7] = e.y;
    return v;
  };
  float2 uv_c = d.cam_uv[qclamp(cp[i] + lane)];
  d8 rr_c = load_rr(d.cam_ray[qclamp(cp[i] + lane)]);  // written by k_lin_ray of the same linearisation
  float2 uv_n = d.cam_uv[qclamp(cp[i] + 64 + lane)];
  int rid_n = d.cam_ray[qclamp(cp[i] + 64 + lane)];
  for (int q0 = cp[i]; q0 < q_end; q0 += 64) {
    const int q = q0 + lane;
    const float2 uv = uv_c;
    const d8 rr = rr_c;
    rr_c = load_rr(rid_n);
    uv_c = uv_n;
    uv_n = d.cam_uv[qclamp(q + 128)];
    rid_n = d.cam_ray[qclamp(q + 128)];
    if (q < q_end) {
      const double Xr[3] = {rr[0], rr[1], rr[2]};
      double res[2], Jc[2][NW], Jr[2][3];
      ba_linearize<F>(cb, Xr, uv.x, uv.y, res, Jc, Jr);
      const double w = rr[6];
      const double sw = sqrt(w);
      cost += 0.5 * (w * (res[0] * res[0] + res[1] * res[1]));
      res[0] *= sw; res[1] *= sw;
#pragma unroll
      for (int k = 0; k < NW; ++k) { const double m = sw * cb[CB_S + Dims<TYPE>::pos(k)]; Jc[0][k] *= m; Jc[1][k] *= m; }
      if constexpr (WRITE_W) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double m = sw * rr[3 + k]; Jr[0][k] *= m; Jr[1][k] *= m; }
        double* Wl = ws + lane * WP;
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
          for (int l = 0; l < 3; ++l) Wl[3 * k + l] = Jc[0][k] * Jr[0][l] + Jc[1][k] * Jr[1][l];
#pragma unroll
        for (int k = NT; k < WS; ++k) Wl[k] = 0.0;  // row padding
      }
      int e = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        g[k] += Jc[0][k] * res[0] + Jc[1][k] * res[1];
#pragma unroll
        for (int l = 0; l <= k; ++l) U[e++] += Jc[0][k] * Jc[0][l] + Jc[1][k] * Jc[1][l];
      }
    }
    if constexpr (WRITE_W) {
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the strip is private to this wave, its lanes run in lock step
      const int n_el = min(64, q_end - q0) * WS;
      double* Wg = d.W + (size_t)q0 * WS;
#pragma unroll
      for (int t = 0; t < WS; ++t) {
        const int idx = t * 64 + lane;
        if (idx < n_el) Wg[idx] = ws[(idx / WS) * WP + (idx % WS)];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // the strip is rewritten in the next trip
    }
  }
  LC_STAMP(2);
  cost = wave_sum(cost);
#pragma unroll
  for (int k = 0; k < NW; ++k) g[k] = wave_sum(g[k]);
#pragma unroll
  for (int k = 0; k < NW * (NW + 1) / 2; ++k) U[k] = wave_sum(U[k]);
  if (lane == 0) {
    tail_store(&costc_[gi], cost);  // (read by the scene's closing workgroup in this launch when the control is fused: tail_last_workgroup)
    if (NC == NW) {
      // what LM control needs of this block, left with it: the gradient's share of the max-norm (k_lm_pre) and the LM diagonal
      // clamp(diag(J^T J)) (LevenbergMarquardtStrategy; refreshed exactly when the blocks are).  With annotation residuals later
      // kernels add to U and g: k_lm_pre then reads the blocks themselves.
      double gmx = 0;
      int e2 = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        gmx = fmax(gmx, fabs(g[k] / cb[CB_S + k]));
        e2 += k;  // index of U[k][k] in the packed lower triangle: k (k + 1) / 2 + k
        diagc_[(size_t)gi * NC + k] = fmin(fmax(U[e2 + k], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
      }
      tail_store(&gmax_[gi], gmx);
    }
    if (NC != NW) {  // the fy row/column has no 2D-2D contribution; k_lin_3d adds the annotation terms
#pragma unroll
      for (int k = 0; k < NC * NC; ++k) U_[(size_t)gi * NC * NC + k] = 0;
#pragma unroll
      for (int k = 0; k < NC; ++k) gc_[(size_t)gi * NC + k] = 0;
    }
    int e = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int pk = Dims<TYPE>::pos(k);
      gc_[(size_t)gi * NC + pk] = g[k];
#pragma unroll
      for (int l = 0; l <= k; ++l) {
        const int pl = Dims<TYPE>::pos(l);
        U_[(size_t)gi * NC * NC + pk * NC + pl] = U[e];
        U_[(size_t)gi * NC * NC + pl * NC + pk] = U[e];
        ++e;
      }
    }
  }
  }
  LC_STAMP(3);
  if constexpr (!Dims<TYPE>::HAS3D) {
    if (!d.fuse_ctl) return;
    // the last workgroup of the scene judges the step and opens the next iteration (what a k_lm_step launch would do)
    __shared__ int tail_flag;
    const bool last_wg = tail_last_workgroup(d.tail_cnt + 2 * sc + 1, (s.n_cam + 3) / 4, &tail_flag);
    LC_STAMP(4);
#ifdef PTZ_LINCAM_STAMPS
    const long long lc_s0 = wall_clock64();
#endif
    if (last_wg) lm_step_wave<TYPE>(d, sc);
#ifdef PTZ_LINCAM_STAMPS
    if (last_wg && threadIdx.x == 0) printf("k_lin_cam: lm_step_wave in the last workgroup (%d) %lld x10 ns\n", (int)blockIdx.x, wall_clock64() - lc_s0);
    LC_STAMP(5);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) printf("k_lin_cam stamps (x10 ns): camera block %lld, trips %lld, sums + stores %lld, tail counter %lld\n", lc_t[1] - lc_t[0], lc_t[2] - lc_t[1], lc_t[3] - lc_t[2], lc_t[4] - lc_t[3]);
#endif
  }
}

// ---- lin_3d: 2D-3D annotation residuals (AddConstraints2d3d, ptzray_optimizer.cc:887-923; weight 1) ----------
// One workgroup per scene: thread = annotation point (closed-form Jacobians of Reproj2d3dFactor w.r.t. the camera
// block [fx, fy, (k1), rvec] and the T_l_w block), then thread 0 adds the few blocks to U_i, g_i, cost_i and builds the
// T_l_w diagonal block / gradient in observation order (fixed order, no atomics).
template <int TYPE>
__global__ __launch_bounds__(256) void k_lin_3d(Dev d)
{
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  const int hh = st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  const double* tb = cur_tlwblk(d, st) + (size_t)s.idx * TLWBLK;
  const double* stl = d.scale_t + (size_t)s.idx * 6;
  for (int o = threadIdx.x; o < s.n_o3; o += 256) {
    const int go = s.o3_off + o;
    const int ci = d.o3_cam[go];
    const double* cb = cur_camblk(d, st) + (size_t)(s.cam_off + ci) * CBS;
    const float2 uv = d.o3_uv[go];
    const double xyz[3] = {d.o3_xyz[(size_t)go * 3], d.o3_xyz[(size_t)go * 3 + 1], d.o3_xyz[(size_t)go * 3 + 2]};
    double res[2], Jc[2][5 + Dims<TYPE>::F3 + 3 * Dims<TYPE>::DISP], Jt[2][6];
    reproj2d3d_eval<Dims<TYPE>::F3, true, Dims<TYPE>::DISP != 0>(cb, tb, xyz, uv.x, uv.y, res, Jc, Jt);
    for (int k = 0; k < NC; ++k) { d.Jc3[(size_t)go * 2 * NC + k] = Jc[0][k] * cb[CB_S + k]; d.Jc3[(size_t)go * 2 * NC + NC + k] = Jc[1][k] * cb[CB_S + k]; }
    for (int k = 0; k < 6; ++k) { d.Jt3[(size_t)go * 12 + k] = Jt[0][k] * stl[k]; d.Jt3[(size_t)go * 12 + 6 + k] = Jt[1][k] * stl[k]; }
    d.r3[(size_t)go * 2] = res[0]; d.r3[(size_t)go * 2 + 1] = res[1];
  }
  __syncthreads();
  // Accumulation, one ELEMENT per thread, every element summed over the annotations in their order (the order the single
  // accumulating thread of the first version used: same bits).  Threads 0..41: the T_l_w block and gradient; then one thread
  // per (camera, entry of its U block / gradient / cost): a camera's annotations are the ones that carry its id.
  const int t = threadIdx.x;
  if (t < 42) {
    const int k = t < 36 ? t / 6 : t - 36, l = t % 6;
    double acc = 0;
    for (int o = 0; o < s.n_o3; ++o) {
      const int go = s.o3_off + o;
      const double* q0 = d.Jt3 + (size_t)go * 12;
      const double* q1 = q0 + 6;
      if (t < 36) acc += q0[k] * q0[l] + q1[k] * q1[l];
      else acc += q0[k] * d.r3[(size_t)go * 2] + q1[k] * d.r3[(size_t)go * 2 + 1];
    }
    if (t < 36) d.Ut[(size_t)s.idx * 36 + t] = acc; else d.gt[(size_t)s.idx * 6 + k] = acc;
  }
  constexpr int PER = NC * NC + NC + 1;  // elements per camera: U, g, cost
  for (int w = t; w < s.n_cam * PER; w += 256) {
    const int ci = w / PER, el = w % PER;
    const int gi = s.cam_off + ci;
    double* dst = el < NC * NC ? U_ + (size_t)gi * NC * NC + el : (el < NC * NC + NC ? gc_ + (size_t)gi * NC + (el - NC * NC) : costc_ + gi);
    double acc = *dst;
    bool any = false;
    for (int o = 0; o < s.n_o3; ++o) {
      const int go = s.o3_off + o;
      if (d.o3_cam[go] != ci) continue;
      any = true;
      const double* j0 = d.Jc3 + (size_t)go * 2 * NC;
      const double* j1 = j0 + NC;
      const double r0 = d.r3[(size_t)go * 2], r1 = d.r3[(size_t)go * 2 + 1];
      if (el < NC * NC) acc += j0[el / NC] * j0[el % NC] + j1[el / NC] * j1[el % NC];
      else if (el < NC * NC + NC) acc += j0[el - NC * NC] * r0 + j1[el - NC * NC] * r1;
      else acc += 0.5 * (r0 * r0 + r1 * r1);
    }
    if (any) *dst = acc;
  }
}

// ---- Jacobi scaling (Ceres: s_j = 1 / (1 + |J_:j|), computed once at iteration 0) -----------------------
template <int TYPE>
__global__ void k_jacobi_scale(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int hh = d.lm[sc].cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  if (t < s.n_cam) {
    const int gi = s.cam_off + t;
#pragma unroll
    for (int k = 0; k < NC; ++k) d.scale_c[(size_t)gi * NC + k] = 1.0 / (1.0 + sqrt(U_[(size_t)gi * NC * NC + k * NC + k]));
  }
  if (t < s.n_ray) {
    const int gj = s.ray_off + t;
    const double* Vc = lin_V(d, d.lm[sc].cur);
    d.scale_r[(size_t)gj * 3 + 0] = 1.0 / (1.0 + sqrt(Vc[(size_t)gj * 6 + 0]));
    d.scale_r[(size_t)gj * 3 + 1] = 1.0 / (1.0 + sqrt(Vc[(size_t)gj * 6 + 2]));
    d.scale_r[(size_t)gj * 3 + 2] = 1.0 / (1.0 + sqrt(Vc[(size_t)gj * 6 + 5]));
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
// cls 0: the intrinsics block of the group's cameras; cls 1: the displacement block (PTZRayDistDisp: one group of all cameras)
template <int TYPE> __device__ __forceinline__ bool is_shared_slot(int k, int cls)
{
  const int a = Dims<TYPE>::at(k);
  return cls ? a >= 15 : (a < 4 || (a >= 10 && a < 15));
}
template <int TYPE> __device__ __forceinline__ int group_class(const Dev& d, const SceneDev& s, int g)
{
  return Dims<TYPE>::DISP ? (int)d.grp_cls[s.grp_off + g] : 0;
}

// thread = (group, slot): group-wide Jacobi scale from the summed squared column norms (members ascending)
template <int TYPE>
__global__ void k_group_scale(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  // one wave per (group, slot): the lanes stride over the members (a group can be all cameras of the scene), fixed butterfly
  const int t = blockIdx.x, lane = threadIdx.x;
  const int g = t / NC, k = t % NC;
  if (g >= s.n_grp || !is_shared_slot<TYPE>(k, group_class<TYPE>(d, s, g))) return;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  const int hh = d.lm[sc].cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  double sum = 0;
  for (int e = gp[g] + lane; e < gp[g + 1]; e += 64) sum += U_[(size_t)(s.cam_off + d.grp_mem[e]) * NC * NC + k * NC + k];
  sum = wave_sum(sum);
  const double sc_g = 1.0 / (1.0 + sqrt(sum));
  for (int e = gp[g] + lane; e < gp[g + 1]; e += 64) d.scale_c[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k] = sc_g;
}

// LM diagonal of a shared slot: clamp(sum of the members' diagonal entries); each member carries an equal share so that
// the fold of S adds them back up
template <int TYPE>
__global__ void k_group_diag(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (d.lm[sc].reuse_diagonal) return;
  const int t = blockIdx.x, lane = threadIdx.x;  // one wave per (group, slot), as k_group_scale
  const int g = t / NC, k = t % NC;
  if (g >= s.n_grp || !is_shared_slot<TYPE>(k, group_class<TYPE>(d, s, g))) return;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  const int hh = d.lm[sc].cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  double sum = 0;
  for (int e = gp[g] + lane; e < gp[g + 1]; e += 64) sum += U_[(size_t)(s.cam_off + d.grp_mem[e]) * NC * NC + k * NC + k];
  sum = wave_sum(sum);
  const double share = fmin(fmax(sum, d.opt.min_lm_diagonal), d.opt.max_lm_diagonal) / (double)(gp[g + 1] - gp[g]);
  for (int e = gp[g] + lane; e < gp[g + 1]; e += 64) diagc_[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k] = share;
}

// gradient with the shared slots folded onto the representative (others 0); every other slot copied.
// One workgroup per scene: copy, barrier, fold.
template <int TYPE>
__global__ void k_group_grad(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if (!d.active[sc] || !st.need_linearize) return;
  const int hh = st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  for (int t = threadIdx.x; t < s.n_cam * NC; t += blockDim.x) d.gfold[(size_t)s.cam_off * NC + t] = gc_[(size_t)s.cam_off * NC + t];
  __syncthreads();
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  for (int t = wv; t < s.n_grp * NC; t += nwv) {  // a wave per (group, slot), lanes over the members
    const int g = t / NC, k = t % NC;
    if (!is_shared_slot<TYPE>(k, group_class<TYPE>(d, s, g))) continue;
    double sum = 0;
    for (int e = gp[g] + lane; e < gp[g + 1]; e += 64) sum += gc_[(size_t)(s.cam_off + d.grp_mem[e]) * NC + k];
    sum = wave_sum(sum);
    for (int e = gp[g] + lane; e < gp[g + 1]; e += 64)
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
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (s.n_grp == 0) return;
  const int np = d.chol.np, n = s.n;
  double* A = d.chol.A + (size_t)sc * np * np;
  extern __shared__ double v[];  // [n + 1], then the scene's tile mask (one byte per tile pair)
  __shared__ double dsum;
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  auto at = [&](int i, int c) -> double& { return i >= c ? A[(size_t)i * np + c] : A[(size_t)c * np + i]; };
  // Entries outside the tile structure are exactly zero (those tiles are never written), so neither summing nor clearing
  // them does anything: a member row is visited only where its tile row meets the column's tile in the structure.  With 200
  // members (the displacement group) that is a fifth of the matrix.
  const int nt = np / CHOL_NB;
  unsigned char* tml = reinterpret_cast<unsigned char*>(v + n + 2);
  for (int t = threadIdx.x; t < nt * nt; t += blockDim.x) tml[t] = d.chol.tmask ? d.chol.tmask[(size_t)sc * nt * nt + t] : 1;
  __syncthreads();
  auto live = [&](int i, int c) { const int ti = i / CHOL_NB, tc = c / CHOL_NB; return tml[max(ti, tc) * nt + min(ti, tc)] != 0; };
  for (int g = 0; g < s.n_grp; ++g) {
    const int e0 = gp[g], e1 = gp[g + 1];
    const int rep = d.grp_mem[e1 - 1];
    for (int k = 0; k < NC; ++k) {
      if (!is_shared_slot<TYPE>(k, group_class<TYPE>(d, s, g))) continue;
      const bool all_cams = e1 - e0 == s.n_cam;  // (members are ascending camera ids: the group is every camera of the scene)
      // rows idx(m) = m NC + k of the members that lie in tile ti: cameras [lo, hi]
      auto cams_of_tile = [&](int ti, int& lo, int& hi) {
        lo = max((ti * CHOL_NB - k + NC - 1) / NC, 0);
        hi = min((ti * CHOL_NB + CHOL_NB - 1 - k) / NC, s.n_cam - 1);
      };
      const int nts = (n + CHOL_NB) / CHOL_NB;  // tiles of this scene's system (rows 0 .. n)
      for (int c = threadIdx.x; c <= n; c += blockDim.x) {
        double sum = 0;
        if (all_cams) {  // walk the tiles that meet column c's tile in the structure, then the few member rows inside each
          const int tc = c / CHOL_NB;
          for (int ti = 0; ti < nts; ++ti) {
            if (!tml[max(ti, tc) * nt + min(ti, tc)]) continue;
            int lo, hi;
            cams_of_tile(ti, lo, hi);
            for (int m = lo; m <= hi; ++m) sum += at(m * NC + k, c);
          }
        }
        else {
          for (int e = e0; e < e1; ++e) {
            const int im = d.grp_mem[e] * NC + k;
            if (live(im, c)) sum += at(im, c);
          }
        }
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
          if (all_cams) {
            const int tc = c / CHOL_NB;
            for (int ti = 0; ti < nts; ++ti) {
              if (!tml[max(ti, tc) * nt + min(ti, tc)]) continue;
              int lo, hi;
              cams_of_tile(ti, lo, hi);
              for (int m = lo; m <= hi; ++m) if (m != rep) at(m * NC + k, c) = 0.0;
            }
          }
          else {
            for (int e = e0; e < e1 - 1; ++e) {
              const int im = d.grp_mem[e] * NC + k;
              if (live(im, c)) at(im, c) = 0.0;
            }
          }
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
        if (a == bq || live(ia, ib)) at(ia, ib) = val;  // (entries outside the structure are zero already)
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
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const int* gp = d.grp_ptr + s.grp_off + s.idx;
  double* y = d.yc + (size_t)sc * d.chol.np;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  for (int t = wv; t < s.n_grp * NC; t += nwv) {  // a wave per (group, slot), lanes over the members
    const int g = t / NC, k = t % NC;
    if (!is_shared_slot<TYPE>(k, group_class<TYPE>(d, s, g))) continue;
    const double yr = y[d.grp_mem[gp[g + 1] - 1] * NC + k];
    for (int e = gp[g] + lane; e < gp[g + 1] - 1; e += 64) y[d.grp_mem[e] * NC + k] = yr;
  }
}

// ---- lm_pre: TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue ---------------------
// LM control is ONE WAVE per scene, whoever runs it: the kernels k_lm_pre / k_lm_post (64 threads per scene), or -- in launch
// shapes of a few scenes (Dev::fuse_ctl) -- the last workgroup of the scene to finish k_lin_cam / k_eval.  Lanes stride over
// the scene's cameras and per-wave partials, one butterfly per sum: the order of every sum is fixed by the scene alone, so a
// scene's bits depend neither on the launch shape nor on who runs its control.
constexpr int LM_THREADS = 64;
template <int TYPE>
__device__ __forceinline__ void lm_pre_wave(const Dev& d, int sc)
{
  constexpr int NC = Dims<TYPE>::NC;
  const SceneDev s = d.scene[sc];
  LmState& st = d.lm[sc];
  const int lane = threadIdx.x & 63;
  const int hh = st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  const bool slow = Dims<TYPE>::HAS3D || d.shared;  // gradient blocks that later kernels touch (annotation terms, group folds): read where they lie
  if (st.step_is_successful) {
    // a fresh linearisation exists: cost, gradient max-norm (unscaled gradient), |x|
    double c = 0, gm = 0, xn = 0;
    // |x|: at iteration zero from the parameters themselves; later it is |x_c| of the step k_lm_post accepted (LmState::cand_norm2:
    // Ceres' x_norm_ after HandleSuccessfulStep is the norm of that same vector)
    const bool first = st.n_summaries == 0;
    const double* cam = cur_cam(d, s, st);
    const int* cp = d.cam_ptr + s.cam_off + s.idx;
    for (int i = lane; i < s.n_cam; i += 64) {
      const int gi = s.cam_off + i;
      c += costc_[gi];
      if (slow) {
        const double* gsrc = d.shared ? d.gfold : d.gc;  // shared intrinsics: the group's gradient sits at its representative
        for (int k = 0; k < NC; ++k) gm = fmax(gm, fabs(gsrc[(size_t)gi * NC + k] / d.scale_c[(size_t)gi * NC + k]));
      }
      else gm = fmax(gm, gmax_[gi]);
      if (first && cp[i + 1] > cp[i]) {  // parameter blocks of cameras without residuals are not in the problem
        const bool intr = !d.shared || (d.cam_flag[gi] & 1);  // a shared intrinsics block is ONE block: counted once
        for (int k = 0; k < 15; ++k)
          if (intr || (k >= 4 && k < 10)) xn += cam[(size_t)i * 15 + k] * cam[(size_t)i * 15 + k];
        if (Dims<TYPE>::DISP && (d.cam_flag[gi] & 2)) {  // the displacement block, once per scene
          const double* dx = d.dsp_x + (size_t)st.cur * d.dsp_stride + (size_t)gi * 3;
          xn += dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        }
      }
    }
    for (int wv = lane; wv < s.n_wave; wv += 64) {  // the rays' share, one entry per wave of k_lin_ray / k_eval
      const double* pl = lin_partial(d, st.cur) + (size_t)(s.part_off - s.idx + wv) * 2;
      gm = fmax(gm, pl[0]);
      if (first) xn += pl[1];
    }
    if (Dims<TYPE>::HAS3D && lane == 0) {
      const double* tl = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
      for (int k = 0; k < 6; ++k) {
        if (first && s.n_o3 > 0) xn += tl[k] * tl[k];  // the T_l_w block is in the problem only when annotation residuals exist
        gm = fmax(gm, fabs(d.gt[(size_t)s.idx * 6 + k] / d.scale_t[(size_t)s.idx * 6 + k]));
      }
    }
    c = wave_sum(c);
    gm = wave_max(gm);
    xn = wave_sum(xn);
    if (lane == 0) {
      st.x_cost = c;
      st.it_cost = c;
      st.grad_max = gm;
      st.x_norm = sqrt(first ? xn : st.cand_norm2);
      st.need_linearize = 0;
      st.ray_lin_ready = 0;
      ++st.num_jac_evals;
      if (st.n_summaries == 0) { st.initial_cost = c; st.final_cost = c; }
    }
  }
  int go_on = 0;  // the scene takes another step and its LM diagonal is to be refreshed
  if (lane == 0) {
    if (st.step_is_successful) ++st.num_successful; else ++st.num_unsuccessful;
    if (st.it_cost < st.final_cost) st.final_cost = st.it_cost;
    ++st.n_summaries;
    if (st.iteration >= d.opt.max_num_iterations) { st.termination = PTZ_NO_CONVERGENCE; retire_scene(d, sc); }
    else if (st.step_is_successful && st.grad_max <= d.opt.gradient_tolerance) { st.termination = PTZ_CONVERGENCE; retire_scene(d, sc); }
    else if (st.radius <= d.opt.min_radius) { st.termination = PTZ_CONVERGENCE; retire_scene(d, sc); }
    else {
      ++st.iteration;
      ++st.num_lm_steps;
      st.step_is_successful = 0;
      d.ray_fail[sc] = 0;
      go_on = st.reuse_diagonal ? 0 : 1;
    }
  }
  // LevenbergMarquardtStrategy: the camera blocks' LM diagonal clamp(diag(J^T J)) is written by k_lin_cam with the blocks themselves
  // (a rejected or invalid step leaves U, hence the diagonal, as it was); only where later kernels add to U -- annotation
  // residuals (k_lin_3d) -- it is taken here, with the T_l_w block's
  if (Dims<TYPE>::HAS3D) {
    go_on = __shfl(go_on, 0, WAVE);
    if (!go_on) return;
    for (int i = lane; i < s.n_cam; i += 64) {
      const int gi = s.cam_off + i;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        diagc_[(size_t)gi * NC + k] = fmin(fmax(U_[(size_t)gi * NC * NC + k * NC + k], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
    }
    if (lane < 6)
      d.diag_t[(size_t)s.idx * 6 + lane] = fmin(fmax(d.Ut[(size_t)s.idx * 36 + lane * 7], d.opt.min_lm_diagonal), d.opt.max_lm_diagonal);
  }
}

template <int TYPE>
__global__ __launch_bounds__(LM_THREADS) void k_lm_pre(Dev d)
{
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0 || !d.active[sc]) return;
  lm_pre_wave<TYPE>(d, sc);
}

// ---- ray_prep: LevenbergMarquardtStrategy diagonal + SchurEliminator e-block inverse --------------------
// The launch also readies the reduced camera system for k_schur (blocks past the ray chunks: one per tile of the lower
// triangle): structural tiles zeroed, identity on the padding rows, CHOL_BIG under the right-hand-side row.
template <int TYPE>
__global__ __launch_bounds__(RAY_BLOCK) void k_ray_prep(Dev d, int n_ray_blocks)
{
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if ((int)blockIdx.x >= n_ray_blocks) {
    // 256 threads per tile of the lower triangle (a 1024-thread workgroup takes four tiles, a 128-thread one half the rate)
    const int np = d.chol.np, nt = np / CHOL_NB;
    const int per = max(1, (int)blockDim.x / 256), tpt = blockDim.x / per;
    const int t = ((int)blockIdx.x - n_ray_blocks) * per + (int)threadIdx.x / tpt;
    if (t >= nt * (nt + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    if (!d.chol.tmask[((size_t)sc * nt + ti) * nt + tj]) return;
    double* T = d.chol.A + (size_t)sc * np * np + (size_t)(ti * CHOL_NB) * np + tj * CHOL_NB;
    const int lt = threadIdx.x % tpt;
    for (int idx = lt; idx < CHOL_NB * CHOL_NB / 2; idx += tpt) {
      const int row = idx >> 5, c2 = (idx & 31) * 2;
      double2 v = make_double2(0.0, 0.0);
      if (ti == tj) {  // padding rows of the diagonal tile: identity, CHOL_BIG at (n, n)
        const int gr = ti * CHOL_NB + row;
        if (gr >= s.n) {
          if (c2 == row) v.x = (gr == s.n) ? CHOL_BIG : 1.0;
          if (c2 + 1 == row) v.y = (gr == s.n) ? CHOL_BIG : 1.0;
        }
      }
      *reinterpret_cast<double2*>(T + (size_t)row * np + c2) = v;
    }
    if (ti == tj && lt == 0 && s.n >= ti * CHOL_NB && s.n < (ti + 1) * CHOL_NB) d.chol.fail[sc] = 0;
    return;
  }
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s.n_ray) return;
  const int gj = s.ray_off + j;
  double V[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) V[k] = lin_V(d, st.cur)[(size_t)gj * 6 + k];
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
  const double* grc = lin_gr(d, st.cur);
  const double g0 = grc[(size_t)gj * 3], g1 = grc[(size_t)gj * 3 + 1], g2 = grc[(size_t)gj * 3 + 2];
  // The record goes out whole in eight 16-byte stores: E, z = E g_r, and -- from what k_lin_ray left in rayrec {X, Jacobi
  // scales, weight} -- the ray side of an observation's Jacobians as k_schur rebuilds them (ba_pair_side): the functor's point
  // Xn, a_l = sqrt(w) s_l / |X| (d res / d X_l = -MR[:, l] a_l after weighting and scaling) and sqrt(w).
  const double2* rr2 = reinterpret_cast<const double2*>(lin_rayrec(d, st.cur) + (size_t)gj * 8);
  const double2 ra = rr2[0], rb = rr2[1], rc = rr2[2], rd = rr2[3];
  const double Xray[3] = {ra.x, ra.y, rb.x};
  double Xn[3], inv_n;
  ba_ray_point<Dims<TYPE>::FACTOR>(Xray, Xn, inv_n);
  const double sw = sqrt(rd.x), swn = sw * inv_n;
  const double z[3] = {E[0] * g0 + E[1] * g1 + E[3] * g2, E[1] * g0 + E[2] * g1 + E[4] * g2, E[3] * g0 + E[4] * g1 + E[5] * g2};
  const double al[3] = {swn * rb.y, swn * rc.x, swn * rc.y};
  if (Dims<TYPE>::FACTOR == 0 && d.e_fold) {
    // k_schur_f's record: every use of (E, z) there carries the factors c_k = sqrt(w) a_k of the ray -- Jr = Jr0 diag(a), Jc = sqrt(w) Jc0
    // -- so they are folded in here, once per ray instead of once per observation, and a and sqrt(w) need not travel
    const double c0 = sw * al[0], c1 = sw * al[1], c2 = sw * al[2];
    const double rec[18] = {E[0], E[1], E[2], E[3], E[4], E[5],
                            c0 * z[0], c1 * z[1], c2 * z[2], Xn[0], Xn[1], Xn[2],
                            c0 * E[0] * c0, c0 * E[1] * c1, c1 * E[2] * c1, c0 * E[3] * c2, c1 * E[4] * c2, c2 * E[5] * c2};
#pragma unroll
    for (int k = 0; k < 9; ++k) *e_piece(d, k, gj) = make_double2(rec[2 * k], rec[2 * k + 1]);
  }
  else {
    const double rec[16] = {E[0], E[1], E[2], E[3], E[4], E[5], z[0], z[1], z[2], Xn[0], Xn[1], Xn[2], al[0], al[1], al[2], sw};
#pragma unroll
    for (int k = 0; k < 8; ++k) *e_piece(d, k, gj) = make_double2(rec[2 * k], rec[2 * k + 1]);
  }
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
// TG: a camera has more observations than the LDS table of T_a rows holds (~1700): the table lives in global memory instead
// (d.Tbuf, camera-major like W; written and re-read by the same workgroup, so it stays in that compute unit's caches).
template <int TYPE, bool TG>
__global__ __launch_bounds__(SCHUR_THREADS, Dims<TYPE>::DISP ? 2 : PTZ_SCHUR_WAVES) void k_schur_w(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW;
  constexpr int NU = NW * (NW + 1) / 2;
  constexpr int NT = NW * 3;
  constexpr int TS = NT;  // row stride of the T table in LDS (an odd stride was measured: 25 % slower, it breaks the 16-byte reads of phase 2)
  int ci, slot;
  xcd_remap(ci, slot);
  const int sc = scene_of_slot(d, slot);
  if (sc < 0 || !d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (ci >= s.n_cam) return;
  const LmState& st = d.lm[sc];
  const int hh = st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  const int o0 = cp[ci], no = cp[ci + 1] - o0;
  const int* cpair = d.cam_pair + s.cam_off + s.idx;
  const int pr0 = cpair[ci], npr = cpair[ci + 1] - pr0;   // this camera's pairs
  const int eb = 0;                                        // entries are addressed by global index
  double* T = TG ? d.Tbuf + (size_t)o0 * TS : lds;             // [no][TS]
  double* strip = TG ? lds : lds + (size_t)no * TS;            // [waves][NC + NU] reduction strip
  // the scene's tile order behind it (identity without one): positions in the reduced system are looked up in LDS
  int* tord = reinterpret_cast<int*>(strip + (SCHUR_THREADS / 64) * (NW + NU));
  {
    const int ntl = d.chol.np / CHOL_NB;
    for (int t = threadIdx.x; t < ntl; t += SCHUR_THREADS) tord[t] = d.tperm ? d.tperm[(size_t)sc * ntl + t] : t;  // (visible after the barrier below)
  }
  auto scol = [&](int c) { return tord[c / CHOL_NB] * CHOL_NB + c % CHOL_NB; };
  const unsigned* ents = d.ent + eb;                // (a slot | b slot << 16), this camera's contiguous range
  const int* pps = d.pair_ptr + s.pair_off + s.idx + pr0;  // entry offsets of this camera's pairs (global entry index)
  double bsum[NW], D[NU];
#pragma unroll
  for (int k = 0; k < NW; ++k) bsum[k] = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) D[k] = 0;
  for (int q = threadIdx.x; q < no; q += SCHUR_THREADS) {
    const int gj = d.cam_ray[o0 + q];  // global ray id of the q-th observation of this camera
    const double2 q0 = *e_piece(d, 0, gj), q1 = *e_piece(d, 1, gj), q2 = *e_piece(d, 2, gj), q3 = *e_piece(d, 3, gj), q4 = *e_piece(d, 4, gj);  // E (6), z (3)
    const double z0 = q3.x, z1 = q3.y, z2 = q4.x;
    const double e0 = q0.x, e1 = q0.y, e2 = q1.x, e3 = q1.y, e4 = q2.x, e5 = q2.y;
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
    __syncthreads();  // also orders the T stores before phase 2
  }
  const int np = d.chol.np;
  double* A = d.chol.A + (size_t)sc * np * np;
  {
    // Diagonal block and right-hand side, one element per thread (the last wave's, so that the first waves start on the camera
    // pairs at once): S_ii = U_i (2D-2D + annotation terms) + D_i^2 - sum T W^T (the latter only on the NW x NW 2D-2D
    // columns), b_i = g_i - sum W z.  Every thread adds up its own element's wave partials, in wave order.
    constexpr int NV = NW + NU, NE = NC * (NC + 1) / 2;
    const int t = (int)threadIdx.x - (SCHUR_THREADS - 64);
    const int gi = s.cam_off + ci;
    auto ipos = [](int c) { return Dims<TYPE>::NC != Dims<TYPE>::NW ? (c == 0 ? 0 : (c == 1 ? -1 : c - 1)) : c; };  // NC slot -> 2D-2D column
    auto strip_sum = [&](int k) { double r = 0; for (int i = 0; i < SCHUR_THREADS / 64; ++i) r += strip[i * NV + k]; return r; };
    if (t >= 0 && t < NE) {
      int p = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
      while ((p + 1) * (p + 2) / 2 <= t) ++p;
      while (p * (p + 1) / 2 > t) --p;
      const int qq = t - p * (p + 1) / 2;
      double v = U_[(size_t)gi * NC * NC + p * NC + qq];
      if (p == qq) {
        const double Dd = sqrt(diagc_[(size_t)gi * NC + p] / st.radius);
        v += Dd * Dd;
      }
      const int ip = ipos(p), iq = ipos(qq);
      if (ip >= 0 && iq >= 0) v -= strip_sum(NW + ip * (ip + 1) / 2 + iq);
      const int rp = scol(ci * NC + p), rq = scol(ci * NC + qq);
      A[(size_t)rp * np + rq] = v;
      A[(size_t)rq * np + rp] = v;
    }
    else if (t >= NE && t < NE + NC) {
      const int p = t - NE, ip = ipos(p);
      double v = gc_[(size_t)gi * NC + p];
      if (ip >= 0) v -= strip_sum(ip);
      A[(size_t)s.n * np + scol(ci * NC + p)] = v;
    }
  }
  // ---- phase 2: off-diagonal blocks of row-block ci (index data and T from LDS, W_b lines from L2/HBM)
  const int l = threadIdx.x & 15;
  // The chain of a pair is header (entry range, first W row of cj, cj) -> entry records -> W_b rows: three dependent round
  // trips before the first multiply, for about three trips of work.  So the NEXT pair's header is fetched while this pair
  // is worked on, its first records as soon as this pair's last rows have been asked for, and inside a pair the records of
  // the next trip ride with the rows of the current one: one exposed round trip per trip, none per pair.
  const int* pbrow = d.pair_brow + s.pair_off + pr0;
  const int* pcj = d.pair_cj + s.pair_off + pr0;
  int pl = threadIdx.x >> 4;
  int h_e0 = 0, h_e1 = 1, h_brow = 0, h_cj = 0;
  unsigned ab0 = 0, ab1 = 0;
  if (pl < npr) {
    h_e0 = pps[pl]; h_e1 = pps[pl + 1]; h_brow = pbrow[pl]; h_cj = pcj[pl];
    ab0 = ents[min(h_e0 + l, h_e1 - 1)]; ab1 = ents[min(h_e0 + l + 16, h_e1 - 1)];
  }
  for (; pl < npr; pl += SCHUR_THREADS / 16) {
    double acc[NW * NW];
#pragma unroll
    for (int k = 0; k < NW * NW; ++k) acc[k] = 0;
    const int e1 = h_e1, brow = h_brow, cj = h_cj;
    int e = h_e0 + l;
    // header of the group's next pair (clamped: the last pair re-reads its own)
    const int pn = min(pl + SCHUR_THREADS / 16, npr - 1);
    const int n_e0 = pps[pn], n_e1 = pps[pn + 1], n_brow = pbrow[pn], n_cj = pcj[pn];
    // two entries per trip while both exist (both W_b rows in flight together), then at most one single entry
    for (; e + 16 < e1; e += 32) {
      const unsigned nx0 = ents[min(e + 32, e1 - 1)], nx1 = ents[min(e + 48, e1 - 1)];
      const double* Wb0 = d.W + (size_t)(brow + (int)(ab0 >> 16)) * Dims<TYPE>::WS;
      const double* Wb1 = d.W + (size_t)(brow + (int)(ab1 >> 16)) * Dims<TYPE>::WS;
      double wb0[NT], wb1[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) { wb0[k] = Wb0[k]; wb1[k] = Wb1[k]; }
      const double* Ta0 = T + (ab0 & 0xffffu) * TS;
      const double* Ta1 = T + (ab1 & 0xffffu) * TS;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        const double t0 = Ta0[3 * p], t1 = Ta0[3 * p + 1], t2 = Ta0[3 * p + 2];
        const double u0 = Ta1[3 * p], u1 = Ta1[3 * p + 1], u2 = Ta1[3 * p + 2];
#pragma unroll
        for (int q = 0; q < NW; ++q)
          acc[p * NW + q] += (t0 * wb0[3 * q] + t1 * wb0[3 * q + 1] + t2 * wb0[3 * q + 2]) +
                             (u0 * wb1[3 * q] + u1 * wb1[3 * q + 1] + u2 * wb1[3 * q + 2]);
      }
      ab0 = nx0; ab1 = nx1;
    }
    // first records of the next pair: on their way during the tail entry and the reduction below
    const unsigned nab0 = ents[min(n_e0 + l, n_e1 - 1)], nab1 = ents[min(n_e0 + l + 16, n_e1 - 1)];
    if (e < e1) {
      const double* Wb0 = d.W + (size_t)(brow + (int)(ab0 >> 16)) * Dims<TYPE>::WS;
      double wb0[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) wb0[k] = Wb0[k];
      const double* Ta0 = T + (ab0 & 0xffffu) * TS;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        const double t0 = Ta0[3 * p], t1 = Ta0[3 * p + 1], t2 = Ta0[3 * p + 2];
#pragma unroll
        for (int q = 0; q < NW; ++q) acc[p * NW + q] += t0 * wb0[3 * q] + t1 * wb0[3 * q + 1] + t2 * wb0[3 * q + 2];
      }
    }
    h_e0 = n_e0; h_e1 = n_e1; h_brow = n_brow; h_cj = n_cj;
    ab0 = nab0; ab1 = nab1;
    // reduce-scatter over the 16 lanes of the group: after the steps with masks 8, 4, 2, 1 lane l holds the complete
    // sum of block element l (fixed order); every lane then stores its own element.  Elements >= NW*NW (NW = 5 keeps
    // 25 values) take a second pass with the lanes that are left.

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
      if (el < NW * NW)
        sys_at(A, np, scol(ci * NC + Dims<TYPE>::pos(el / NW)), scol(cj * NC + Dims<TYPE>::pos(el % NW))) = -v[0];
    }
  }
}

// ---- schur (round 3): the same row-block of the reduced system WITHOUT a materialised W ---------------------------------------
// k_schur_w above gathers one stored 96-byte product row W_b per camera-pair entry -- 470 k entries = 45 MB per scene and pass
// from a 9.6 MB array, at the fabric's row-gather rate, 7 x the kernel's algorithmic bytes -- and k_lin_cam writes those rows
// (1 GB per launch of a 1000-scene batch).  But the two observations (a, b) of an entry look at ONE ray, and the Jacobians
// of b depend on that ray and on camera cj only (not on the pixel): so nothing is stored per observation at all.
//   Phase 1 (thread = observation a of camera ci): the ray's 128-byte record {E, z, Xn, ray-side factors, sqrt(w)} (k_ray_prep)
//     is the only gather; Jc_a, Jr_a come from ba_pair_side's factored form with camera ci's block in scalar registers,
//     W_a = Jc_a^T Jr_a and T_a = W_a E stay in registers, S_ii and b_i as before.  LDS keeps, per observation,
//     T'_a = w |X|^-1 T_a diag(s_r)  and the functor's point Xn.
//   Phase 2 (thread = RUN of entries): with ba_pair_side's factors of observation b -- MR = M R_j and G = [intrinsic columns |
//     M x P], from Xn and camera cj's rotation and focal length --
//        T_a W_b^T = -[(T'_a MR^T) G] blockdiag(I, Jl_j) diag(s_j)
//     so an entry costs 3 + NW x 3 LDS reads and ~100 FP64 operations, no global load but its 4-byte record.  The camera's
//     entry list (all its pairs, one after the other) is cut by the host into at most SCHUR_THREADS runs of equal length that
//     never straddle two pairs (build_pairs): a thread sums ONE run in registers, so every lane of the workgroup has the same
//     amount of work whatever the lengths of the pairs (they differ by a factor of 30 on a C2 rig, and a camera has only ~23 of
//     them: 16-lane groups dealt pair by pair kept 40 % of the lanes busy).
//   Phase 3: the run sums meet in LDS -- in the space of the T table, which is dead by then -- and are added up per pair in run
//     order (fixed order: a scene's bits do not depend on the batch it is in); the product with Jl_j and the Jacobi scales is
//     applied once per pair, on the sums.
// FP64 VALU is the bound now; results differ from k_schur_w's in the last bits only (x = Px / Pz there,
// Px * (1 / Pz) with a Newton reciprocal here; the sum over a pair's entries is grouped differently).
#ifdef PTZ_SCHUR_STAMPS  // probe builds only: where a k_schur workgroup's time goes (one camera of slot 0, thread 0; 100 MHz wall clock)
#define SC_STAMP(i) do { if (stamp_on) sc_t[i] = wall_clock64(); } while (0)
#else
#define SC_STAMP(i) do { } while (0)
#endif
// Threads per workgroup (= the most runs a camera's entries are cut into).  512-thread workgroups at 128 VGPRs were measured
// for PTZRay (4 waves per SIMD): 42.5 ms against 41.9 ms per 256-scene solve -- the registers that the prefetches below need
// are worth more than the occupancy.
template <int TYPE> constexpr int schur_threads() { return 256; }
// (the wrong-result occupancy probe of round 3 -- a table of n wrapped rows, three workgroups per compute unit -- is kept as a patch:
//  tools/probes/hip/schur_fake_rows_r3.patch)
template <int TYPE, bool TG>
__global__ __launch_bounds__(schur_threads<TYPE>(), schur_threads<TYPE>() == 512 ? 4 : 2) void k_schur(Dev d)  // (threads, waves per SIMD)
{
  constexpr int THREADS = schur_threads<TYPE>();
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR, CBS = Dims<TYPE>::CBS;
  constexpr int NU = NW * (NW + 1) / 2;
  constexpr int NT = NW * 3;
  // row of the LDS table: T'_a (NW x 3), Xn (3), padded to an ODD number of doubles -- 15 for PTZRay -- so that rows whose
  // numbers differ modulo 16 start on different bank pairs (the host orders a pair's entries accordingly, build_pairs)
  constexpr int TS = (NT + 3) | 1;
  constexpr int ROT0 = BaDims<F>::ROT0;
  int ci, slot;
  xcd_remap(ci, slot);
  const int sc = scene_of_slot(d, slot);
  if (sc < 0 || !d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (ci >= s.n_cam) return;
  const LmState& st = d.lm[sc];
  const int hh = st.cur;
  double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;   // the camera side of the linearisation, half hh (LmState::cur selects the current point's)
  double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  double* costc_ = d.costc + (size_t)hh * d.lin_cams;
  double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  double* gmax_ = d.cam_gmax + (size_t)hh * d.lin_cams;
  (void)U_; (void)gc_; (void)costc_; (void)diagc_; (void)gmax_;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  const int o0 = cp[ci], no = cp[ci + 1] - o0;
#ifdef PTZ_SCHUR_STAMPS
  const bool stamp_on = slot == 0 && ci == (s.n_cam * 3) / 4 && threadIdx.x == 0;
  long long sc_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SC_STAMP(0);
  // ---- phase 1, first thing: the gathers.  Two dependent loads stand before an observation's arithmetic -- its ray id, then
  // the ray's 128-byte record, 2-5 us away on a loaded chip, more than a trip's arithmetic -- so the records of the first PF
  // trips (768 observations: every view of a C2 rig) are ALL asked for here, before anything else the kernel needs (its
  // structure, the camera block, the entry list: measured 5.5 us of dependent scalar and vector loads in front of the gathers
  // when they were issued where phase 1 begins).  Unconditional loads: past the camera's end the index is clamped and the
  // data unused.
  typedef double d16 __attribute__((ext_vector_type(16)));
  auto load_rec = [&](int gj) {
    d16 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double2 t = *e_piece(d, k, gj); v[2 * k] = t.x; v[2 * k + 1] = t.y; }
    return v;
  };
  // (a camera WITHOUT observations is legal -- a candidate image none of whose tracks survived: then o0 may equal the batch's
  //  observation count, so the clamp goes one below it; any valid observation's ray will do, its record is not used)
  auto oclamp = [&](int q) { return max(o0 + min(q, no - 1), 0); };
  constexpr int PF = 3;
  int gid[PF];
  d16 rcs[PF];
#pragma unroll
  for (int t = 0; t < PF; ++t) gid[t] = d.cam_ray[oclamp((int)threadIdx.x + t * schur_threads<TYPE>())];
#pragma unroll
  for (int t = 0; t < PF; ++t) rcs[t] = load_rec(gid[t]);
  // (the loads of the diagonal block's inputs, below, are issued HERE, behind the gathers: issued in front of phase 2 they made the
  //  last wave enter it one memory round trip after the others -- per-wave time stamps)
  constexpr int DIAG_NE = NC * (NC + 1) / 2;
  const int dt = (int)threadIdx.x - (THREADS - 64);
  int dp = 0, dq = 0;
  double dv = 0, dDc = 0;
  if (dt >= 0 && dt < DIAG_NE) {
    dp = (int)((sqrtf(8.0f * dt + 1.0f) - 1.0f) * 0.5f);
    while ((dp + 1) * (dp + 2) / 2 <= dt) ++dp;
    while (dp * (dp + 1) / 2 > dt) --dp;
    dq = dt - dp * (dp + 1) / 2;
    dv = U_[(size_t)(s.cam_off + ci) * NC * NC + dp * NC + dq];
    if (dp == dq) dDc = diagc_[(size_t)(s.cam_off + ci) * NC + dp];
  }
  else if (dt >= DIAG_NE && dt < DIAG_NE + NC) dv = gc_[(size_t)(s.cam_off + ci) * NC + (dt - DIAG_NE)];
  const int* cpair = d.cam_pair + s.cam_off + s.idx;
  const int pr0 = cpair[ci], npr = cpair[ci + 1] - pr0;   // this camera's pairs
  constexpr int PS = (NW * NW) | 1;  // doubles per run sum in LDS (odd pitch)
  const int ntl = d.chol.np / CHOL_NB;
  double* strip = lds;                                         // [waves][NW + NU] reduction strip of phase 1
  int* tord = reinterpret_cast<int*>(strip + (THREADS / 64) * (NW + NU));  // the scene's tile order (identity without one)
  // the camera's entry list as 16-bit a slots: staged here by all threads with coalesced loads, so that phase 2 -- one run per
  // thread, every thread at another place of the list -- reads it at LDS latency instead of chasing 4-byte global loads
  unsigned short* eslot = reinterpret_cast<unsigned short*>(strip + (THREADS / 64) * (NW + NU) + (ntl + 2) / 2);
  const int* crun = d.cam_run + s.cam_off + s.idx;
  const int run0 = crun[ci], nrun = crun[ci + 1] - run0;      // this camera's runs (scene-local numbers)
  const uint2* runs = d.run_rec + s.run_off + run0;
  const int* ppt = d.pair_ptr + s.pair_off + s.idx + pr0;      // entry offsets of this camera's pairs (global entry index)
  const int ent0 = npr > 0 ? ppt[0] : 0;                       // first entry of the camera
  const int nent = npr > 0 ? ppt[npr] - ent0 : 0;
  double* tab = reinterpret_cast<double*>(eslot + ((d.schur_ent_cap + 3) & ~3));  // T table, later the run sums
  double* T = TG ? d.Tbuf + (size_t)o0 * TS : tab;             // [no][TS]
  double* part = tab;                                          // [THREADS][PS]
  // (tile order and entry list are staged behind the trips of phase 1: a load's LDS store waits for every load issued before
  //  it, the gathers included, so nothing that is stored to LDS may be loaded between the gathers and the trips)
  auto scol = [&](int c) { return tord[c / CHOL_NB] * CHOL_NB + c % CHOL_NB; };
  const double* camtab = cur_camblk(d, st) + (size_t)s.cam_off * CBS;
  // this thread's run of phase 2 and the other camera's block are fetched during phase 1 (three dependent global loads would
  // otherwise stand between the barrier and the first entry): the run record behind the gathers, the rest behind the trips
  const int* pcj = d.pair_cj + s.pair_off + pr0;
  uint2 rr = make_uint2(0u, 0u);
  const double* cbj = camtab;
  double Rj[9], fj = 0, fyj = 0, kj[F ? 5 : 1], dj[F == 3 ? 3 : 1];
  auto load_cj = [&]() {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rj[k] = cbj[CB_R + k];
    fj = cbj[CB_F];
    fyj = F == 2 ? cbj[CB_FY] : fj;
#pragma unroll
    for (int k = 0; k < (F ? 5 : 1); ++k) kj[k] = F ? cbj[CB_K + k] : 0.0;
#pragma unroll
    for (int k = 0; k < (F == 3 ? 3 : 1); ++k) dj[k] = F == 3 ? cbj[CB_D + k] : 0.0;
  };
  // ---- phase 1
  {
    const double* cbi = camtab + (size_t)ci * CBS;  // uniform: scalar loads, hoisted out of the loop
    double Ri[9], Jli[9], kdi[F ? 5 : 1], dsi[F == 3 ? 3 : 1], sci[NW];
#pragma unroll
    for (int k = 0; k < 9; ++k) { Ri[k] = cbi[CB_R + k]; Jli[k] = cbi[CB_JL + k]; }
#pragma unroll
    for (int k = 0; k < (F ? 5 : 1); ++k) kdi[k] = F ? cbi[CB_K + k] : 0.0;
#pragma unroll
    for (int k = 0; k < (F == 3 ? 3 : 1); ++k) dsi[k] = F == 3 ? cbi[CB_D + k] : 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) sci[k] = cbi[CB_S + Dims<TYPE>::pos(k)];
    const double fi = cbi[CB_F], fyi = F == 2 ? cbi[CB_FY] : fi;
    double bsum[NW], D[NU];
#pragma unroll
    for (int k = 0; k < NW; ++k) bsum[k] = 0;
#pragma unroll
    for (int k = 0; k < NU; ++k) D[k] = 0;
    auto process = [&](const d16& rc, int q) {
      const double2 r0 = make_double2(rc[0], rc[1]), r1 = make_double2(rc[2], rc[3]), r2 = make_double2(rc[4], rc[5]), r3 = make_double2(rc[6], rc[7]),
                    r4 = make_double2(rc[8], rc[9]), r5 = make_double2(rc[10], rc[11]), r6 = make_double2(rc[12], rc[13]), r7 = make_double2(rc[14], rc[15]);
      const double e0 = r0.x, e1 = r0.y, e2 = r1.x, e3 = r1.y, e4 = r2.x, e5 = r2.y;
      const double z0 = r3.x, z1 = r3.y, z2 = r4.x;
      const double Xn[3] = {r4.y, r5.x, r5.y};
      const double al[3] = {r6.x, r6.y, r7.x};
      const double sw = r7.y;
      // this observation's weighted, scaled Jacobians from the factored form: Jr = -MR diag(a), Jc = sqrt(w) [G with its
      // rotation columns through Jl_i] diag(s_i); behind the camera (PTZRayDist) both are zero
      double MR[2][3], G[2][NW], Jc[2][NW], Jr[2][3];
      const bool front = ba_pair_side<F>(Ri, fi, fyi, kdi, dsi, Xn, MR, G);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int k = 0; k < NW; ++k) Jc[r][k] = G[r][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Jc[r][ROT0 + k] = G[r][ROT0] * Jli[k] + G[r][ROT0 + 1] * Jli[3 + k] + G[r][ROT0 + 2] * Jli[6 + k];
#pragma unroll
        for (int k = 0; k < NW; ++k) Jc[r][k] = front ? Jc[r][k] * (sw * sci[k]) : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) Jr[r][k] = front ? -(MR[r][k] * al[k]) : 0.0;
      }
      double w[NT];
#pragma unroll
      for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) w[3 * k + l] = Jc[0][k] * Jr[0][l] + Jc[1][k] * Jr[1][l];
      // T'_a = T_a diag(sqrt(w) a): the second observation's ray-side factors, folded into the row once
      const double f0 = sw * al[0], f1 = sw * al[1], f2 = sw * al[2];
      double* Tq = T + (size_t)q * TS;
      int e = 0;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        const double w0 = w[3 * p], w1 = w[3 * p + 1], w2 = w[3 * p + 2];
        bsum[p] += w0 * z0 + w1 * z1 + w2 * z2;
        const double t0 = w0 * e0 + w1 * e1 + w2 * e3, t1 = w0 * e1 + w1 * e2 + w2 * e4, t2 = w0 * e3 + w1 * e4 + w2 * e5;
        Tq[3 * p] = t0 * f0; Tq[3 * p + 1] = t1 * f1; Tq[3 * p + 2] = t2 * f2;
#pragma unroll
        for (int qq = 0; qq <= p; ++qq) D[e++] += t0 * w[3 * qq] + t1 * w[3 * qq + 1] + t2 * w[3 * qq + 2];
      }
      Tq[NT] = Xn[0]; Tq[NT + 1] = Xn[1]; Tq[NT + 2] = Xn[2];
    };
    if ((int)threadIdx.x < nrun) rr = runs[threadIdx.x];
    const int cjx = npr > 0 ? pcj[rr.y & 0xffffu] : 0;  // (arrives during the trips below)
    SC_STAMP(8);
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      const int q = (int)threadIdx.x + t * THREADS;
      if (q < no) process(rcs[t], q);
      SC_STAMP(9 + t);
    }
    for (int q = (int)threadIdx.x + PF * THREADS; q < no; q += THREADS) process(load_rec(d.cam_ray[o0 + q]), q);  // (very large views)
    cbj = camtab + (size_t)cjx * CBS;
    load_cj();  // on its way during the reductions and the barrier below
    // tile order and the first 8 x THREADS entries of the camera's list: loads here, LDS stores behind the reductions
    int tord_v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int t = (int)threadIdx.x + u * THREADS; tord_v[u] = d.tperm ? d.tperm[(size_t)sc * ntl + min(t, ntl - 1)] : t; }
    unsigned ent_v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ent_v[u] = d.ent[ent0 + max(min(u * THREADS + (int)threadIdx.x, nent - 1), 0)];
    // one pass of the block tree for all NW + NU sums (fixed order: lanes by butterfly, waves in wave order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NV = NW + NU;
    // Reduce-scatter instead of NV full butterflies: in every halving step a lane keeps one half of the values it still holds
    // (which half: one bit of its lane number), sends the other half to its partner and adds what the partner sends -- NV/2 +
    // NV/4 + .. shuffles instead of 6 NV (14 sums: 17 against 84); the value that is left is finished with plain butterfly
    // steps over the remaining lane bits.  A fixed tree per value, as before.
    constexpr int PAD = NV <= 16 ? 16 : (NV <= 32 ? 32 : 64);
    double v[PAD];
#pragma unroll
    for (int k = 0; k < NW; ++k) v[k] = bsum[k];
#pragma unroll
    for (int k = 0; k < NU; ++k) v[NW + k] = D[k];
#pragma unroll
    for (int k = NV; k < PAD; ++k) v[k] = 0.0;
    int vidx = 0;  // index of the value this lane ends up with
    {
      int off = 32;
#pragma unroll
      for (int h = PAD / 2; h >= 1; h >>= 1, off >>= 1) {
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < h; ++i) {
          const double keep = up ? v[i + h] : v[i];
          const double send = up ? v[i] : v[i + h];
          v[i] = keep + __shfl_xor(send, off, WAVE);
        }
        if (up) vidx += h;
      }
#pragma unroll
      for (; off >= 1; off >>= 1) v[0] += __shfl_xor(v[0], off, WAVE);
    }
    constexpr int LOWMASK = 64 / PAD - 1;  // lanes that differ only in these bits hold the same value
    if ((lane & LOWMASK) == 0 && vidx < NV) strip[wv * NV + vidx] = v[0];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int t = (int)threadIdx.x + u * THREADS; if (t < ntl) tord[t] = tord_v[u]; }
    for (int t = (int)threadIdx.x + 2 * THREADS; t < ntl; t += THREADS) tord[t] = d.tperm ? d.tperm[(size_t)sc * ntl + t] : t;  // (more than 512 tiles)
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = u * THREADS + (int)threadIdx.x; if (e < nent) eslot[e] = (unsigned short)(ent_v[u] & 0xffffu); }
    for (int e0 = 8 * THREADS; e0 < nent; e0 += 8 * THREADS) {  // (a view with more than 2048 entries: eight loads in flight again)
      unsigned v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = d.ent[ent0 + min(e0 + u * THREADS + (int)threadIdx.x, nent - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * THREADS + (int)threadIdx.x; if (e < nent) eslot[e] = (unsigned short)(v8[u] & 0xffffu); }
    }
    SC_STAMP(12);
    __syncthreads();  // also orders the T, tile-order and entry stores before phase 2
  }
  SC_STAMP(1);
  const int np = d.chol.np;
  double* A = d.chol.A + (size_t)sc * np * np;
  // Diagonal block and right-hand side, one element per thread of the LAST wave: S_ii = U_i (2D-2D + annotation terms) + D_i^2
  // - sum T W^T (the latter only on the NW x NW 2D-2D columns), b_i = g_i - sum W z.  Its inputs were asked for at the top of the
  // kernel; the element is finished behind phase 2 (schur_diag_finish), while the other waves reduce their run sums.  (Assembled
  // in front of phase 2 it made the last wave start its runs ~2 us late, and every other wave then waited that long at the
  // barrier behind the runs; with only the loads ISSUED there the last wave was still 1.1-2.2 us late.)
  auto schur_diag_finish = [&]() {  // every thread adds up its own element's wave partials, in wave order
    constexpr int NV = NW + NU;
    auto ipos = [](int c) { return Dims<TYPE>::NC != Dims<TYPE>::NW ? (c == 0 ? 0 : (c == 1 ? -1 : c - 1)) : c; };  // NC slot -> 2D-2D column
    auto strip_sum = [&](int k) { double r = 0; for (int i = 0; i < THREADS / 64; ++i) r += strip[i * NV + k]; return r; };
    if (dt >= 0 && dt < DIAG_NE) {
      double v = dv;
      if (dp == dq) {
        const double Dd = sqrt(dDc / st.radius);
        v += Dd * Dd;
      }
      const int ip = ipos(dp), iq = ipos(dq);
      if (ip >= 0 && iq >= 0) v -= strip_sum(NW + ip * (ip + 1) / 2 + iq);
      const int rp = scol(ci * NC + dp), rq = scol(ci * NC + dq);
      A[(size_t)rp * np + rq] = v;
      A[(size_t)rq * np + rp] = v;
    }
    else if (dt >= DIAG_NE && dt < DIAG_NE + NC) {
      const int p = dt - DIAG_NE, ip = ipos(p);
      double v = dv;
      if (ip >= 0) v -= strip_sum(ip);
      A[(size_t)s.n * np + scol(ci * NC + p)] = v;
    }
  };
  SC_STAMP(2);
  // ---- phase 2: off-diagonal blocks of row-block ci, one run of entries per thread
  const int* prun = d.pair_run + s.pair_off + s.idx + pr0;    // first run of each of this camera's pairs; prun[npr] = end of the last
  // Without TG the host has cut the entries into at most THREADS runs: ONE round, after which the T table is dead and its
  // LDS space takes the run sums.  With the T table in global memory (very many observations or pairs in one view) the runs
  // may need several rounds; a pair that continues from the previous round adds to what that round stored.
  for (int base = 0; base < nrun; base += THREADS) {
    const int r = base + (int)threadIdx.x;
    double acc[NW * NW];
#pragma unroll
    for (int k = 0; k < NW * NW; ++k) acc[k] = 0;
    if (r < nrun) {
      if (base > 0) {  // (a later round of a view with more runs than threads: the first round's were fetched ahead)
        rr = runs[r];
        cbj = camtab + (size_t)pcj[rr.y & 0xffffu] * CBS;
        load_cj();
      }
      const int cnt = (int)(rr.y >> 16);
      const unsigned short* es = eslot + ((int)rr.x - ent0);  // this run's a slots
      // the row of the NEXT entry is read from LDS while the current one is worked on (past the end the last row again)
      double Tn[NT + 3];
      {
        const double* Ta = T + (size_t)es[0] * TS;
#pragma unroll
        for (int k = 0; k < NT + 3; ++k) Tn[k] = Ta[k];
      }
      for (int k = 0; k < cnt; ++k) {
        double Tc[NT + 3];
#pragma unroll
        for (int i = 0; i < NT + 3; ++i) Tc[i] = Tn[i];
        {
          const double* Ta = T + (size_t)es[min(k + 1, cnt - 1)] * TS;
#pragma unroll
          for (int i = 0; i < NT + 3; ++i) Tn[i] = Ta[i];
        }
        const double Xn[3] = {Tc[NT], Tc[NT + 1], Tc[NT + 2]};
        double MR[2][3], G[2][NW];
        if (ba_pair_side<F>(Rj, fj, fyj, kj, dj, Xn, MR, G)) {
#pragma unroll
          for (int p = 0; p < NW; ++p) {
            const double t0 = Tc[3 * p], t1 = Tc[3 * p + 1], t2 = Tc[3 * p + 2];
            const double u0 = t0 * MR[0][0] + t1 * MR[0][1] + t2 * MR[0][2];
            const double u1 = t0 * MR[1][0] + t1 * MR[1][1] + t2 * MR[1][2];
#pragma unroll
            for (int q = 0; q < NW; ++q) acc[p * NW + q] = fma(u1, G[1][q], fma(u0, G[0][q], acc[p * NW + q]));
          }
        }
      }
    }
    SC_STAMP(3);
    // ---- phase 3: run sums -> pair sums -> parameters of camera cj -> the reduced system
    if (!TG) __syncthreads();  // every thread is done with the T table
    SC_STAMP(4);
    if (r < nrun) {
#pragma unroll
      for (int k = 0; k < NW * NW; ++k) part[threadIdx.x * PS + k] = acc[k];
    }
    __syncthreads();
    // what step (b) needs of camera cj -- Jl and the Jacobi scales, two cache lines away in global memory -- is asked for now,
    // so that it arrives while step (a) runs
    double pJl[9], pS[NW];
    const bool pre = (int)threadIdx.x < npr * NW;
    {
      const double* cbp = camtab + (size_t)(pre ? pcj[(int)threadIdx.x / NW] : 0) * CBS;  // (a camera without pairs must not read pcj[0]: it may lie behind the array)
#pragma unroll
      for (int k = 0; k < 9; ++k) pJl[k] = cbp[CB_JL + k];
#pragma unroll
      for (int k = 0; k < NW; ++k) pS[k] = cbp[CB_S + Dims<TYPE>::pos(k)];
    }
    // (a) thread = (pair, element): the pair's runs of this round in run order, into the row of its first run
    for (int it = threadIdx.x; it < npr * NW * NW; it += THREADS) {
      const int pl = it / (NW * NW), el = it % (NW * NW);
      const int ra = max(prun[pl] - run0 - base, 0), rb = min(prun[pl + 1] - run0 - base, THREADS);
      if (ra >= rb) continue;  // (the pair has no run in this round)
      // four independent partial sums (the LDS reads of a serial sum would each wait for the one before): runs ra, ra+4, .. /
      // ra+1, .. / ra+2, .. / ra+3, ..; a fixed order as well
      double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      int q = ra;
      for (; q + 3 < rb; q += 4) {
        v0 += part[q * PS + el]; v1 += part[(q + 1) * PS + el]; v2 += part[(q + 2) * PS + el]; v3 += part[(q + 3) * PS + el];
      }
      for (; q < rb; ++q) v0 += part[q * PS + el];
      part[ra * PS + el] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    SC_STAMP(5);
    // (b) thread = (pair, row of the block): rotation columns from the camera frame to the parameters (x Jl_j), every column by
    //     its Jacobi scale, into S_ij
    for (int it = threadIdx.x; it < npr * NW; it += THREADS) {
      const int pl = it / NW, p = it % NW;
      const int ra = max(prun[pl] - run0 - base, 0), rb = min(prun[pl + 1] - run0 - base, THREADS);
      if (ra >= rb) continue;
      const bool first = prun[pl] - run0 >= base;  // else the pair continues from the previous round
      const int cj = pcj[pl];
      if (it != (int)threadIdx.x) {  // (a camera with more than THREADS / NW pairs: later items fetch their own)
        const double* cbj = camtab + (size_t)cj * CBS;
#pragma unroll
        for (int k = 0; k < 9; ++k) pJl[k] = cbj[CB_JL + k];
#pragma unroll
        for (int k = 0; k < NW; ++k) pS[k] = cbj[CB_S + Dims<TYPE>::pos(k)];
      }
      double row[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) row[q] = part[ra * PS + p * NW + q];
      const double a0 = row[ROT0], a1 = row[ROT0 + 1], a2 = row[ROT0 + 2];
#pragma unroll
      for (int k = 0; k < 3; ++k) row[ROT0 + k] = a0 * pJl[k] + a1 * pJl[3 + k] + a2 * pJl[6 + k];
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const double v = row[q] * pS[q];
        double& dst = sys_at(A, np, scol(ci * NC + Dims<TYPE>::pos(p)), scol(cj * NC + Dims<TYPE>::pos(q)));
        dst = first ? v : dst + v;
      }
    }
    if (base + THREADS < nrun) __syncthreads();  // the next round's sums go to the same rows
  }
  schur_diag_finish();
#ifdef PTZ_SCHUR_STAMPS
  SC_STAMP(6);
  if (stamp_on)
    printf("k_schur cam %d: obs %d pairs %d runs %d | x10 ns: phase1 %lld (issue+stage %lld, trips %lld %lld %lld, reduce %lld, barrier %lld), diag %lld, runs(thread 0) %lld, wait %lld, sums %lld, store %lld\n", ci, no, npr, nrun,
           sc_t[1] - sc_t[0], sc_t[8] - sc_t[0], sc_t[9] - sc_t[8], sc_t[10] - sc_t[9], sc_t[11] - sc_t[10], sc_t[12] - sc_t[11], sc_t[1] - sc_t[12],
           sc_t[2] - sc_t[1], sc_t[3] - sc_t[2], sc_t[4] - sc_t[3], sc_t[5] - sc_t[4], sc_t[6] - sc_t[5]);
#endif
}

// ---- schur (round 5): the table row in FACTORED form -- 8 doubles per observation instead of 15 -----------------------------
// k_schur above keeps T'_a (NW x 3) and the functor's point Xn per observation: 15 doubles, 62 KB for a 514-observation view,
// which is what holds it at two workgroups per compute unit.  But T_a = W_a E = Jc_a^T (Jr_a E) has rank two, and for PTZRay
// (FACTOR 0: pinhole, no distortion) everything of observation a that an entry needs follows from its normalised image point
// (x_a, y_a) in camera i:
//   * G_a = Graw(x_a, y_a) diag(1, f_i, f_i, f_i),  Graw(x, y) = [[-x, xy, -(1 + x^2), y], [-y, 1 + y^2, -xy, -x]]  (ba_pair_side
//     with its zeros and its common factors written out: M x P = f Pz^-1 ... Pz cancels);
//   * the ray's direction is R_i^T (x_a, y_a, 1) up to the factor Pz_i, so camera j sees it at P' = R_j R_i^T (x_a, y_a, 1) --
//     (x_b, y_b) = (P'x, P'y) / P'z exactly as from Xn, and 1 / Pz_j = (1 / Pz_i)(1 / P'z);
//   * MR_b = f_j Pz_j^-1 [R_j(0) - x_b R_j(2); R_j(1) - y_b R_j(2)] = f_j Pz_j^-1 [Rji(0) - x_b Rji(2); Rji(1) - y_b Rji(2)] R_i.
// With Q_a = sqrt(w) Pz_i^-1 [Jr_a E diag(sqrt(w) a)] R_i^T = Pz_i^-1 [Jr0_a E''] R_i^T (2 x 3, phase 1; E'' = diag(c) E diag(c), c = sqrt(w) a,
// comes pre-scaled from k_ray_prep, Dev::e_fold: 96 bytes per observation to gather instead of 128) an entry's contribution is
//     T'_a MR_b^T G_b = (F_i B_i)^T { Graw_a^T [P'z^-1 Q_a (Rji(0:1) - (x_b, y_b) Rji(2))^T] Graw_b } f_j F_j
// (F = diag(1, f, f, f), B = blockdiag(1, Jl) diag(s)): the braces are what a run sums -- 8 doubles of LDS per entry (64 bytes
// instead of 120) and about the same ~90 FP64 operations -- and everything outside them is constant over a camera pair and is
// applied to the pair's sum in phase 3.  Table 514 x 9 doubles = 37 KB: three workgroups per compute unit.
// Same sums in a fixed order (bits independent of batch, launch shape, scene group: the tests that hold k_schur to that hold this
// kernel); its blocks differ from k_schur's in the last bits (regrouped products).  PTZRay only (with or without annotations);
// the other factor types and views beyond the LDS table keep k_schur.
// TG: the table in global memory (d.Tbuf) for views beyond the LDS table, and several rounds of runs for views with more runs than
// threads -- the same arithmetic in the same order (a scene keeps its bits in a batch that needs them).
template <int TYPE, bool TG>
__global__ __launch_bounds__(256, 3) void k_schur_f(Dev d)
{
  constexpr int THREADS = 256;
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, CBS = Dims<TYPE>::CBS;
  static_assert(Dims<TYPE>::FACTOR == 0 && NW == 4, "k_schur_f: PTZRay only");
  constexpr int NU = NW * (NW + 1) / 2;
  constexpr int TS = SCHUR_F_ROW;   // 8 doubles at an odd pitch
  int ci, slot;
  xcd_remap(ci, slot);
  const int sc = scene_of_slot(d, slot);
  if (sc < 0 || !d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  if (ci >= s.n_cam) return;
  const LmState& st = d.lm[sc];
  const int hh = st.cur;
  const double* U_ = d.U + (size_t)hh * d.lin_cams * NC * NC;
  const double* gc_ = d.gc + (size_t)hh * d.lin_cams * NC;
  const double* diagc_ = d.diag_c + (size_t)hh * d.lin_cams * NC;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int* cp = d.cam_ptr + s.cam_off + s.idx;
  const int o0 = cp[ci], no = cp[ci + 1] - o0;
#ifdef PTZ_SCHUR_STAMPS
  const bool stamp_on = slot == 0 && ci == (s.n_cam * 3) / 4 && threadIdx.x == 0;
  long long sc_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SC_STAMP(0);
  // ---- the gathers first (see k_schur): ray ids, then the rays' 128-byte records of the first PF trips
  struct Rec { double v[12]; };  // {z'' (3), Xn (3), E'' (6)}: pieces 3..8 of the ray (Dev::e_fold)
  auto load_rec = [&](int gj) {
    Rec r;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double2 t = *e_piece(d, 3 + k, gj); r.v[2 * k] = t.x; r.v[2 * k + 1] = t.y; }
    return r;
  };
  auto oclamp = [&](int q) { return max(o0 + min(q, no - 1), 0); };
  constexpr int PF = 3;
  int gid[PF];
  Rec rcs[PF];
#ifdef PTZ_SCHUR_STAMPS  // (probe builds: the waits make the stamps mean "arrived"; they cost nothing the loads do not wait for anyway)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  SC_STAMP(13);
#endif
#pragma unroll
  for (int t = 0; t < PF; ++t) gid[t] = d.cam_ray[oclamp((int)threadIdx.x + t * THREADS)];
#ifdef PTZ_SCHUR_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SC_STAMP(14);
#endif
#pragma unroll
  for (int t = 0; t < PF; ++t) rcs[t] = load_rec(gid[t]);
#ifdef PTZ_SCHUR_STAMPS
  SC_STAMP(15);
#endif
  constexpr int DIAG_NE = NC * (NC + 1) / 2;
  const int dt = (int)threadIdx.x - (THREADS - 64);
  int dp = 0, dq = 0;
  double dv = 0, dDc = 0;
  if (dt >= 0 && dt < DIAG_NE) {
    dp = (int)((sqrtf(8.0f * dt + 1.0f) - 1.0f) * 0.5f);
    while ((dp + 1) * (dp + 2) / 2 <= dt) ++dp;
    while (dp * (dp + 1) / 2 > dt) --dp;
    dq = dt - dp * (dp + 1) / 2;
    dv = U_[(size_t)(s.cam_off + ci) * NC * NC + dp * NC + dq];
    if (dp == dq) dDc = diagc_[(size_t)(s.cam_off + ci) * NC + dp];
  }
  else if (dt >= DIAG_NE && dt < DIAG_NE + NC) dv = gc_[(size_t)(s.cam_off + ci) * NC + (dt - DIAG_NE)];
  const int* cpair = d.cam_pair + s.cam_off + s.idx;
  const int pr0 = cpair[ci], npr = cpair[ci + 1] - pr0;
  constexpr int PS = (NW * NW) | 1;
  const int ntl = d.chol.np / CHOL_NB;
  double* strip = lds;                                         // [waves][NW + NU]
  int* tord = reinterpret_cast<int*>(strip + (THREADS / 64) * (NW + NU));
  unsigned short* eslot = reinterpret_cast<unsigned short*>(strip + (THREADS / 64) * (NW + NU) + (ntl + 2) / 2);
  const int* crun = d.cam_run + s.cam_off + s.idx;
  const int run0 = crun[ci], nrun = crun[ci + 1] - run0;
  const uint2* runs = d.run_rec + s.run_off + run0;
  const int* ppt = d.pair_ptr + s.pair_off + s.idx + pr0;
  const int ent0 = npr > 0 ? ppt[0] : 0;
  const int nent = npr > 0 ? ppt[npr] - ent0 : 0;
  double* tab = reinterpret_cast<double*>(eslot + ((d.schur_ent_cap + 3) & ~3));  // the table [no][TS], later the run sums [THREADS][PS]
  double* T = TG ? d.Tbuf + (size_t)o0 * TS : tab;
  double* part = tab;
  auto scol = [&](int c) { return tord[c / CHOL_NB] * CHOL_NB + c % CHOL_NB; };
  const double* camtab = cur_camblk(d, st) + (size_t)s.cam_off * CBS;
  const int* pcj = d.pair_cj + s.pair_off + pr0;
  uint2 rr = make_uint2(0u, 0u);
  const double* cbi = camtab + (size_t)ci * CBS;  // uniform: scalar loads
  double Ri[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Ri[k] = cbi[CB_R + k];
  const double fi = cbi[CB_F];
  double Rji[9];
  // ---- phase 1
  {
    double Jli[9], sci[NW];
#pragma unroll
    for (int k = 0; k < 9; ++k) Jli[k] = cbi[CB_JL + k];
#pragma unroll
    for (int k = 0; k < NW; ++k) sci[k] = cbi[CB_S + Dims<TYPE>::pos(k)];
    double bsum[NW], D[NU];
#pragma unroll
    for (int k = 0; k < NW; ++k) bsum[k] = 0;
#pragma unroll
    for (int k = 0; k < NU; ++k) D[k] = 0;
    auto process = [&](const Rec& rc, int q) {
      const double z0 = rc.v[0], z1 = rc.v[1], z2 = rc.v[2];
      const double Xn[3] = {rc.v[3], rc.v[4], rc.v[5]};
      const double e0 = rc.v[6], e1 = rc.v[7], e2 = rc.v[8], e3 = rc.v[9], e4 = rc.v[10], e5 = rc.v[11];
      const double Px = Ri[0] * Xn[0] + Ri[1] * Xn[1] + Ri[2] * Xn[2];
      const double Py = Ri[3] * Xn[0] + Ri[4] * Xn[1] + Ri[5] * Xn[2];
      const double Pz = Ri[6] * Xn[0] + Ri[7] * Xn[1] + Ri[8] * Xn[2];
      const double iz = rcp_nr(Pz);
      const double x = Px * iz, y = Py * iz, fiz = fi * iz;
      // Jr0 = -MR, MR = fiz [R(0) - x R(2); R(1) - y R(2)]   (the ray-side factors a live in E'', z'')
      double Jr[2][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Jr[0][k] = -(fiz * (Ri[k] - x * Ri[6 + k]));
        Jr[1][k] = -(fiz * (Ri[3 + k] - y * Ri[6 + k]));
      }
      // camera columns Jc0 = Graw F_i blockdiag(1, Jl_i) diag(s_i)   (sqrt(w) lives in E'', z'')
      const double xy = x * y, ox = fma(x, x, 1.0), oy = fma(y, y, 1.0);
      const double g0[3] = {fi * xy, -(fi * ox), fi * y}, g1[3] = {fi * oy, -(fi * xy), -(fi * x)};
      double Jc[2][NW];
      Jc[0][0] = -x * sci[0];
      Jc[1][0] = -y * sci[0];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Jc[0][1 + k] = (g0[0] * Jli[k] + g0[1] * Jli[3 + k] + g0[2] * Jli[6 + k]) * sci[1 + k];
        Jc[1][1 + k] = (g1[0] * Jli[k] + g1[1] * Jli[3 + k] + g1[2] * Jli[6 + k]) * sci[1 + k];
      }
      // Q = Jr E (2 x 3);  N = Q Jr^T;  S_ii -= Jc^T N Jc;  b_i -= Jc^T (Jr z)
      double Q[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        Q[r][0] = Jr[r][0] * e0 + Jr[r][1] * e1 + Jr[r][2] * e3;
        Q[r][1] = Jr[r][0] * e1 + Jr[r][1] * e2 + Jr[r][2] * e4;
        Q[r][2] = Jr[r][0] * e3 + Jr[r][1] * e4 + Jr[r][2] * e5;
      }
      const double n00 = Q[0][0] * Jr[0][0] + Q[0][1] * Jr[0][1] + Q[0][2] * Jr[0][2];
      const double n01 = Q[0][0] * Jr[1][0] + Q[0][1] * Jr[1][1] + Q[0][2] * Jr[1][2];
      const double n11 = Q[1][0] * Jr[1][0] + Q[1][1] * Jr[1][1] + Q[1][2] * Jr[1][2];
      const double jz0 = Jr[0][0] * z0 + Jr[0][1] * z1 + Jr[0][2] * z2;
      const double jz1 = Jr[1][0] * z0 + Jr[1][1] * z1 + Jr[1][2] * z2;
      int e = 0;
#pragma unroll
      for (int p = 0; p < NW; ++p) {
        bsum[p] += Jc[0][p] * jz0 + Jc[1][p] * jz1;
        const double a0 = Jc[0][p] * n00 + Jc[1][p] * n01, a1 = Jc[0][p] * n01 + Jc[1][p] * n11;
#pragma unroll
        for (int qq = 0; qq <= p; ++qq) D[e++] += a0 * Jc[0][qq] + a1 * Jc[1][qq];
      }
      // the row: Q_a = Pz^-1 [Jr0 E''] R_i^T, and (x, y)
      double* Tq = T + (size_t)q * TS;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double q0 = Q[r][0] * iz, q1 = Q[r][1] * iz, q2 = Q[r][2] * iz;
#pragma unroll
        for (int m = 0; m < 3; ++m) Tq[3 * r + m] = q0 * Ri[3 * m] + q1 * Ri[3 * m + 1] + q2 * Ri[3 * m + 2];
      }
      Tq[6] = x; Tq[7] = y;
    };
    if ((int)threadIdx.x < nrun) rr = runs[threadIdx.x];
    const int cjx = npr > 0 ? pcj[rr.y & 0xffffu] : 0;
    SC_STAMP(8);
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      const int q = (int)threadIdx.x + t * THREADS;
      if (q < no) process(rcs[t], q);
      SC_STAMP(9 + t);
    }
    for (int q = (int)threadIdx.x + PF * THREADS; q < no; q += THREADS) process(load_rec(d.cam_ray[o0 + q]), q);
    {  // Rji = R_j R_i^T of this thread's run (on its way during the reductions and the barrier below)
      const double* cbj = camtab + (size_t)cjx * CBS;
      double Rj[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rj[k] = cbj[CB_R + k];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rji[3 * r + c] = Rj[3 * r] * Ri[3 * c] + Rj[3 * r + 1] * Ri[3 * c + 1] + Rj[3 * r + 2] * Ri[3 * c + 2];
    }
    int tord_v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int t = (int)threadIdx.x + u * THREADS; tord_v[u] = d.tperm ? d.tperm[(size_t)sc * ntl + min(t, ntl - 1)] : t; }
    unsigned ent_v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ent_v[u] = d.ent[ent0 + max(min(u * THREADS + (int)threadIdx.x, nent - 1), 0)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NV = NW + NU;
    constexpr int PAD = 16;
    static_assert(NV <= PAD, "reduce-scatter width");
    double v[PAD];
#pragma unroll
    for (int k = 0; k < NW; ++k) v[k] = bsum[k];
#pragma unroll
    for (int k = 0; k < NU; ++k) v[NW + k] = D[k];
#pragma unroll
    for (int k = NV; k < PAD; ++k) v[k] = 0.0;
    // reduce-scatter over the wave (see k_schur) with register swaps and DPP moves instead of 34 trips through the LDS crossbar,
    // which phase 2 of the neighbouring workgroups keeps busy: the same sums of the same operands
    int vidx = 0;
    {
#pragma unroll
      for (int i = 0; i < 8; ++i) { swap_halves32(v[i], v[i + 8]); v[i] += v[i + 8]; }   // distance 32: lanes >= 32 now hold values 8..15
      if (lane & 32) vidx += 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) { swap_rows16(v[i], v[i + 4]); v[i] += v[i + 4]; }     // distance 16
      if (lane & 16) vidx += 4;
      {  // distance 8
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
          v[i] = keep + lane_xor_dpp<8>(send);
        }
        if (up) vidx += 2;
      }
      {  // distance 4
        const bool up = (lane & 4) != 0;
        const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + lane_xor_dpp<4>(send);
        if (up) vidx += 1;
      }
      v[0] += lane_xor_dpp<2>(v[0]);
      v[0] += lane_xor_dpp<1>(v[0]);
    }
    constexpr int LOWMASK = 64 / PAD - 1;
    if ((lane & LOWMASK) == 0 && vidx < NV) strip[wv * NV + vidx] = v[0];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int t = (int)threadIdx.x + u * THREADS; if (t < ntl) tord[t] = tord_v[u]; }
    for (int t = (int)threadIdx.x + 2 * THREADS; t < ntl; t += THREADS) tord[t] = d.tperm ? d.tperm[(size_t)sc * ntl + t] : t;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int e = u * THREADS + (int)threadIdx.x; if (e < nent) eslot[e] = (unsigned short)(ent_v[u] & 0xffffu); }
    for (int e0 = 8 * THREADS; e0 < nent; e0 += 8 * THREADS) {
      unsigned v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = d.ent[ent0 + min(e0 + u * THREADS + (int)threadIdx.x, nent - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + u * THREADS + (int)threadIdx.x; if (e < nent) eslot[e] = (unsigned short)(v8[u] & 0xffffu); }
    }
    SC_STAMP(12);
    __syncthreads();
  }
  SC_STAMP(1);
  const int np = d.chol.np;
  double* A = d.chol.A + (size_t)sc * np * np;
  auto schur_diag_finish = [&]() {
    constexpr int NV = NW + NU;
    auto ipos = [](int c) { return Dims<TYPE>::NC != Dims<TYPE>::NW ? (c == 0 ? 0 : (c == 1 ? -1 : c - 1)) : c; };
    auto strip_sum = [&](int k) { double r = 0; for (int i = 0; i < THREADS / 64; ++i) r += strip[i * NV + k]; return r; };
    if (dt >= 0 && dt < DIAG_NE) {
      double v = dv;
      if (dp == dq) {
        const double Dd = sqrt(dDc / st.radius);
        v += Dd * Dd;
      }
      const int ip = ipos(dp), iq = ipos(dq);
      if (ip >= 0 && iq >= 0) v -= strip_sum(NW + ip * (ip + 1) / 2 + iq);
      const int rp = scol(ci * NC + dp), rq = scol(ci * NC + dq);
      A[(size_t)rp * np + rq] = v;
      A[(size_t)rq * np + rp] = v;
    }
    else if (dt >= DIAG_NE && dt < DIAG_NE + NC) {
      const int p = dt - DIAG_NE, ip = ipos(p);
      double v = dv;
      if (ip >= 0) v -= strip_sum(ip);
      A[(size_t)s.n * np + scol(ci * NC + p)] = v;
    }
  };
  // ---- phase 2: one run of entries per thread.  Views whose table is in LDS have at most THREADS runs: ONE round, after which
  // the table is dead and its space takes the run sums; with the table in global memory the runs may need several rounds, and a
  // pair that continues from the previous round adds to what that round stored.
  const int* prun = d.pair_run + s.pair_off + s.idx + pr0;
  SC_STAMP(2);
  for (int base = 0; base < (TG ? nrun : min(nrun, 1)); base += THREADS) {
    const int r = base + (int)threadIdx.x;
    double acc[NW * NW];
#pragma unroll
    for (int k = 0; k < NW * NW; ++k) acc[k] = 0;
    if (r < nrun) {
      if (base > 0) {  // (a later round: the first round's run and camera were fetched ahead)
        rr = runs[r];
        const double* cbj = camtab + (size_t)pcj[rr.y & 0xffffu] * CBS;
        double Rj[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rj[k] = cbj[CB_R + k];
#pragma unroll
        for (int rw = 0; rw < 3; ++rw)
#pragma unroll
          for (int c = 0; c < 3; ++c) Rji[3 * rw + c] = Rj[3 * rw] * Ri[3 * c] + Rj[3 * rw + 1] * Ri[3 * c + 1] + Rj[3 * rw + 2] * Ri[3 * c + 2];
      }
      const int cnt = (int)(rr.y >> 16);
      const unsigned short* es = eslot + ((int)rr.x - ent0);
      double Tn[8];
      {
        const double* Ta = T + (size_t)es[0] * TS;
#pragma unroll
        for (int k = 0; k < 8; ++k) Tn[k] = Ta[k];
      }
      for (int k = 0; k < cnt; ++k) {
        double Tc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) Tc[i] = Tn[i];
        {
          const double* Ta = T + (size_t)es[min(k + 1, cnt - 1)] * TS;
#pragma unroll
          for (int i = 0; i < 8; ++i) Tn[i] = Ta[i];
        }
        const double xa = Tc[6], ya = Tc[7];
        const double px = fma(Rji[0], xa, fma(Rji[1], ya, Rji[2]));
        const double py = fma(Rji[3], xa, fma(Rji[4], ya, Rji[5]));
        const double pz = fma(Rji[6], xa, fma(Rji[7], ya, Rji[8]));
        const double iz = rcp_nr(pz);
        const double xb = px * iz, yb = py * iz;
        double K[2][2];
        {
          const double m00 = fma(-xb, Rji[6], Rji[0]), m01 = fma(-xb, Rji[7], Rji[1]), m02 = fma(-xb, Rji[8], Rji[2]);
          const double m10 = fma(-yb, Rji[6], Rji[3]), m11 = fma(-yb, Rji[7], Rji[4]), m12 = fma(-yb, Rji[8], Rji[5]);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            K[q][0] = (Tc[3 * q] * m00 + Tc[3 * q + 1] * m01 + Tc[3 * q + 2] * m02) * iz;
            K[q][1] = (Tc[3 * q] * m10 + Tc[3 * q + 1] * m11 + Tc[3 * q + 2] * m12) * iz;
          }
        }
        const double xyb = xb * yb, oxb = fma(xb, xb, 1.0), oyb = fma(yb, yb, 1.0);
        const double Gb0[4] = {-xb, xyb, -oxb, yb}, Gb1[4] = {-yb, oyb, -xyb, -xb};
        double KG[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          KG[0][q] = K[0][0] * Gb0[q] + K[0][1] * Gb1[q];
          KG[1][q] = K[1][0] * Gb0[q] + K[1][1] * Gb1[q];
        }
        const double xya = xa * ya, oxa = fma(xa, xa, 1.0), oya = fma(ya, ya, 1.0);
        const double Ga0[4] = {-xa, xya, -oxa, ya}, Ga1[4] = {-ya, oya, -xya, -xa};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[p * 4 + q] = fma(Ga1[p], KG[1][q], fma(Ga0[p], KG[0][q], acc[p * 4 + q]));
      }
    }
    // ---- phase 3: run sums -> pair sums -> (F_i B_i)^T . f_j F_j B_j -> the reduced system
    SC_STAMP(3);
    if (!TG) __syncthreads();  // every thread is done with the table
    SC_STAMP(4);
    if (r < nrun) {
#pragma unroll
      for (int k = 0; k < NW * NW; ++k) part[threadIdx.x * PS + k] = acc[k];
    }
    __syncthreads();
    double pJl[9], pS[NW], pf = 0;
    const bool pre = (int)threadIdx.x < npr * NW;
    {
      const double* cbp = camtab + (size_t)(pre ? pcj[(int)threadIdx.x / NW] : 0) * CBS;  // (a camera without pairs must not read pcj[0]: it may lie behind the array)
#pragma unroll
      for (int k = 0; k < 9; ++k) pJl[k] = cbp[CB_JL + k];
#pragma unroll
      for (int k = 0; k < NW; ++k) pS[k] = cbp[CB_S + Dims<TYPE>::pos(k)];
      pf = cbp[CB_F];
    }
    for (int it = threadIdx.x; it < npr * NW * NW; it += THREADS) {
      const int pl = it / (NW * NW), el = it % (NW * NW);
      const int ra = max(prun[pl] - run0 - base, 0), rb = min(prun[pl + 1] - run0 - base, THREADS);
      if (ra >= rb) continue;  // (the pair has no run in this round)
      double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      int q = ra;
      for (; q + 3 < rb; q += 4) {
        v0 += part[q * PS + el]; v1 += part[(q + 1) * PS + el]; v2 += part[(q + 2) * PS + el]; v3 += part[(q + 3) * PS + el];
      }
      for (; q < rb; ++q) v0 += part[q * PS + el];
      part[ra * PS + el] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    SC_STAMP(5);
    // thread = (pair, row p of the block): rows through (F_i B_i)^T, columns through f_j F_j B_j, every element stored once
    for (int it = threadIdx.x; it < npr * NW; it += THREADS) {
      const int pl = it / NW, p = it % NW;
      const int ra = max(prun[pl] - run0 - base, 0), rb = min(prun[pl + 1] - run0 - base, THREADS);
      if (ra >= rb) continue;
      const bool first = prun[pl] - run0 >= base;  // else the pair continues from the previous round
      const int cj = pcj[pl];
      if (it != (int)threadIdx.x) {
        const double* cbj = camtab + (size_t)cj * CBS;
#pragma unroll
        for (int k = 0; k < 9; ++k) pJl[k] = cbj[CB_JL + k];
#pragma unroll
        for (int k = 0; k < NW; ++k) pS[k] = cbj[CB_S + Dims<TYPE>::pos(k)];
        pf = cbj[CB_F];
      }
      const double* blk = part + ra * PS;
      double row[NW];
      if (p == 0) {
        const double sl = cbi[CB_S + Dims<TYPE>::pos(0)];
#pragma unroll
        for (int q = 0; q < NW; ++q) row[q] = blk[q] * sl;
      }
      else {
        const int k = p - 1;
        const double sl = fi * cbi[CB_S + Dims<TYPE>::pos(p)];
        const double l0 = cbi[CB_JL + k], l1 = cbi[CB_JL + 3 + k], l2 = cbi[CB_JL + 6 + k];
#pragma unroll
        for (int q = 0; q < NW; ++q) row[q] = (l0 * blk[NW + q] + l1 * blk[2 * NW + q] + l2 * blk[3 * NW + q]) * sl;
      }
      const double a0 = row[1], a1 = row[2], a2 = row[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) row[1 + k] = (a0 * pJl[k] + a1 * pJl[3 + k] + a2 * pJl[6 + k]) * pf;
      if constexpr (NC == NW) {
        // (round 6) the block row lies in ONE row of the stored triangle when camera i's tile comes behind camera j's: its four
        // values are 32 contiguous, 32-byte aligned bytes -- two 16-byte stores instead of four 8-byte ones (the same values)
        const int rr_ = scol(ci * NC + p), c0 = scol(cj * NC);
        if (rr_ > c0) {
          double2* d2 = reinterpret_cast<double2*>(A + (size_t)rr_ * np + c0);
          const double v0 = row[0] * (pf * pS[0]), v1 = row[1] * (pf * pS[1]), v2 = row[2] * (pf * pS[2]), v3 = row[3] * (pf * pS[3]);
          if (first) { d2[0] = make_double2(v0, v1); d2[1] = make_double2(v2, v3); }
          else { const double2 o0 = d2[0], o1 = d2[1]; d2[0] = make_double2(o0.x + v0, o0.y + v1); d2[1] = make_double2(o1.x + v2, o1.y + v3); }
          continue;
        }
      }
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const double v = row[q] * (pf * pS[q]);
        double& dst = sys_at(A, np, scol(ci * NC + Dims<TYPE>::pos(p)), scol(cj * NC + Dims<TYPE>::pos(q)));
        dst = first ? v : dst + v;
      }
    }
    if (base + THREADS < nrun) __syncthreads();  // the next round's sums go to the same rows
  }
  schur_diag_finish();
#ifdef PTZ_SCHUR_STAMPS
  SC_STAMP(6);
  if (stamp_on)
    printf("k_schur_f cam %d: obs %d pairs %d runs %d | x10 ns: scalars %lld ids %lld issue %lld | phase1 %lld (issue+stage %lld, trips %lld %lld %lld, reduce %lld, barrier %lld), diag %lld, runs(thread 0) %lld, wait %lld, sums %lld, store %lld\n", ci, no, npr, nrun,
           sc_t[13] - sc_t[0], sc_t[14] - sc_t[13], sc_t[15] - sc_t[14], sc_t[1] - sc_t[0], sc_t[8] - sc_t[0], sc_t[9] - sc_t[8], sc_t[10] - sc_t[9], sc_t[11] - sc_t[10], sc_t[12] - sc_t[11], sc_t[1] - sc_t[12],
           sc_t[2] - sc_t[1], sc_t[3] - sc_t[2], sc_t[4] - sc_t[3], sc_t[5] - sc_t[4], sc_t[6] - sc_t[5]);
#endif
}

// ---- schur_3d: rows of the T_l_w block in the reduced system (it is not coupled to the rays) ---------------------
//   S_tt = U_t + D_t^2,  S_t,cam(i) = sum_{annotations of camera i} Jt^T Jc,  b_t = g_t
template <int TYPE>
__global__ __launch_bounds__(64) void k_schur_3d(Dev d)
{
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
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
    sys_at(A, np, sys_col(d, sc, t0 + k), sys_col(d, sc, t0 + l)) = v;
  }
  A[(size_t)s.n * np + sys_col(d, sc, t0 + k)] = d.gt[(size_t)s.idx * 6 + k];
  for (int o = 0; o < s.n_o3; ++o) {  // observation order: deterministic accumulation into the (zeroed) coupling row
    const int go = s.o3_off + o;
    const int ci = d.o3_cam[go];
    const double q0 = d.Jt3[(size_t)go * 12 + k], q1 = d.Jt3[(size_t)go * 12 + 6 + k];
    const double* j0 = d.Jc3 + (size_t)go * 2 * NC;
    for (int l = 0; l < NC; ++l) sys_at(A, np, sys_col(d, sc, t0 + k), sys_col(d, sc, ci * NC + l)) += q0 * j0[l] + q1 * j0[NC + l];
  }
}

// ---- cam_update: candidate cameras and their residual-side blocks ----------------------------------------
// One camera's candidate from the solution of the reduced system, and everything later kernels need of it.  Called by k_cam_update
// (thread = camera, everything to global memory) and by k_eval's prologue when the camera update is folded into it (launch shapes
// of a few scenes: EVERY workgroup computes all candidates into its LDS tables -- lds_cand: the CANDBLK-entry prefix of the
// camera block, lds_dct: the step as k_eval applies it -- and the scene's first workgroup also stores them, to_global).  cbc: the
// camera's block at x (global or LDS).  One piece of code, so that a scene's bits do not depend on which of the two runs it.
template <int TYPE> struct CamIn {  // what a camera's candidate is computed from (all loads of cam_update_load, in flight together)
  double x15[15], y[Dims<TYPE>::NC], sc[Dims<TYPE>::NC], jl[9], dsp0[3];
};
template <int TYPE>
__device__ __forceinline__ void cam_update_load(const Dev& d, const SceneDev& s, const LmState& st, int sc, int i, CamIn<TYPE>& in)
{
  constexpr int NC = Dims<TYPE>::NC, CBS = Dims<TYPE>::CBS;
  const int gi = s.cam_off + i;
  const double* x = cur_cam(d, s, st) + (size_t)i * 15;
  const double* ysc = d.yc + (size_t)sc * d.chol.np + (size_t)i * NC;  // (the back-substitution leaves the solution in camera order: CholBatch::xperm)
  const double* cbc = cur_camblk(d, st) + (size_t)gi * CBS;  // Jacobi scales and SO(3) Jacobian of the camera at x
#pragma unroll
  for (int k = 0; k < 15; ++k) in.x15[k] = x[k];
#pragma unroll
  for (int k = 0; k < NC; ++k) { in.y[k] = ysc[k]; in.sc[k] = cbc[CB_S + k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) in.jl[k] = cbc[CB_JL + k];
  if (Dims<TYPE>::DISP) {
    const double* dx = d.dsp_x + (size_t)st.cur * d.dsp_stride + (size_t)gi * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) in.dsp0[k] = dx[k];
  }
}
template <int TYPE>
__device__ __forceinline__ void cam_update_apply(const Dev& d, const SceneDev& s, const LmState& st, int sc, int i, const CamIn<TYPE>& in,
                                                 bool to_global, double* lds_cand, double* lds_dct)
{
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, DCS = NC | 1;
  const int gi = s.cam_off + i;
  const double* x15 = in.x15;
  const double* y = in.y;
  double c15[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) c15[k] = x15[k];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double step = -y[k];
    if (to_global) d.dc[(size_t)gi * NC + k] = step;
    if (Dims<TYPE>::at(k) < 15) c15[Dims<TYPE>::at(k) < 15 ? Dims<TYPE>::at(k) : 0] += step * in.sc[k];
  }
  double dsp0[3] = {0, 0, 0}, dsp[3] = {0, 0, 0};
  if (Dims<TYPE>::DISP) {  // the camera's copy of the displacement block (every copy takes the same step: k_group_expand)
    double* dxc = d.dsp_x + (size_t)(st.cur ^ 1) * d.dsp_stride + (size_t)gi * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dsp0[k] = in.dsp0[k];
      dsp[k] = dsp0[k] + (-y[NC - 3 + k]) * in.sc[NC - 3 + k];
      if (to_global) dxc[k] = dsp[k];
    }
  }
  {  // the scaled step of the camera's 2D-2D columns as [intrinsic components | om = Jl v_rot] (ba_step_dir), for k_eval
    double sv[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) sv[k] = in.sc[Dims<TYPE>::pos(k)] * (-y[Dims<TYPE>::pos(k)]);
    double dr[NW];
    constexpr int RW = Dims<TYPE>::RW;  // [columns before the rotation, columns behind it | Jl v_rot]
#pragma unroll
    for (int k = 0; k < NW - 3; ++k) dr[k] = sv[k < RW ? k : k + 3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      dr[NW - 3 + r] = in.jl[3 * r] * sv[RW] + in.jl[3 * r + 1] * sv[RW + 1] + in.jl[3 * r + 2] * sv[RW + 2];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      if (to_global) d.dct[(size_t)gi * DCS + k] = dr[k];
      if (lds_dct) lds_dct[k] = dr[k];
    }
  }
  // the candidate's camera block: its first CANDBLK entries (rotation, intrinsics) are what k_eval needs of it; the full block
  // (SO(3) Jacobian, scales) goes to the other half of camblk -- if the step is accepted the linearisation kernels find it there
  double cb[CAMBLK];
#pragma unroll
  for (int k = CB_S; k < CAMBLK; ++k) cb[k] = 0.0;
  fill_camblk(c15, cb, to_global, Dims<TYPE>::DISP ? dsp : nullptr);
  if (lds_cand) {
#pragma unroll
    for (int k = 0; k < CANDBLK; ++k) lds_cand[k] = cb[k];
  }
  if (!to_global) return;
  double* xc = d.cam_x + (size_t)(st.cur ^ 1) * d.cam_stride + (size_t)gi * 15;
#pragma unroll
  for (int k = 0; k < 15; ++k) xc[k] = c15[k];
#pragma unroll
  for (int k = 0; k < NC; ++k) cb[CB_S + k] = in.sc[k];
  double* cfull = d.camblk + (size_t)(st.cur ^ 1) * d.camblk_stride + (size_t)gi * CBS;
#pragma unroll
  for (int k = 0; k < CAMBLK; ++k) cfull[k] = cb[k];
#pragma unroll
  for (int k = 0; k < CANDBLK; ++k) d.candblk[(size_t)gi * CDS + k] = cb[k];
  {  // |x - x_c|^2 and |x_c|^2 over this camera's parameter blocks that are in the problem (k_lm_post)
    const int* cp = d.cam_ptr + s.cam_off + s.idx;
    double dn = 0, cn = 0;
    if (cp[i + 1] > cp[i]) {  // (blocks of cameras without residuals are not in the problem)
      const bool intr = !d.shared || (d.cam_flag[gi] & 1);  // a shared intrinsics block is ONE block: counted once
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        if (!intr && (k < 4 || k >= 10)) continue;
        dn += (x15[k] - c15[k]) * (x15[k] - c15[k]);
        cn += c15[k] * c15[k];
      }
      if (Dims<TYPE>::DISP && (d.cam_flag[gi] & 2)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { dn += (dsp0[k] - dsp[k]) * (dsp0[k] - dsp[k]); cn += dsp[k] * dsp[k]; }
      }
    }
    *reinterpret_cast<double2*>(d.camstep + (size_t)gi * 2) = make_double2(dn, cn);
  }
  if (Dims<TYPE>::HAS3D && i == 0) {
    const double* t = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    double* tc = d.tlw_x + (size_t)(st.cur ^ 1) * d.tlw_stride + (size_t)s.idx * 6;
    double tn[6];
    for (int k = 0; k < 6; ++k) {
      const double step = -d.yc[(size_t)sc * d.chol.np + NC * s.n_cam + k];
      d.dt[(size_t)s.idx * 6 + k] = step;
      tn[k] = t[k] + step * d.scale_t[(size_t)s.idx * 6 + k];
      tc[k] = tn[k];
    }
    double* tb = d.tlwcand + (size_t)s.idx * TLWBLK;
    double* tf = d.tlwblk + (size_t)(st.cur ^ 1) * d.tlwblk_stride + (size_t)s.idx * TLWBLK;  // full block, used if the step is accepted
    double R[9], Jl[9];
    const double rv[3] = {tn[0], tn[1], tn[2]};
    rodrigues(rv, R);
    so3_left_jacobian(rv, Jl);
    for (int k = 0; k < 9; ++k) { tb[k] = R[k]; tf[k] = R[k]; tf[9 + k] = Jl[k]; }
    tb[18] = tn[3]; tb[19] = tn[4]; tb[20] = tn[5];
    tf[18] = tn[3]; tf[19] = tn[4]; tf[20] = tn[5];
  }
}

template <int TYPE>
__global__ void k_cam_update(Dev d)
{
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n_cam) return;
  CamIn<TYPE> in;
  cam_update_load<TYPE>(d, s, st, sc, i, in);
  cam_update_apply<TYPE>(d, s, st, sc, i, in, true, nullptr, nullptr);
}

// ---- eval: ray back-substitution, model cost change and candidate cost in one ray-centric pass -----------------
//   y_r = E (g_r - sum_a Jr_a^T (Jc_a y_c))            (SchurEliminator::BackSubstitute; W_a = Jc_a^T Jr_a is not
//                                                        re-read: the Jacobian blocks are recomputed, flops are free)
//   candidate ray = x + scale * (-y_r)
//   model_cost_change = -(J d)^T (r + J d / 2)          (TrustRegionMinimizer::ComputeTrustRegionStep)
//   candidate_cost    = 1/2 sum w |r(x + delta)|^2
// The scaled camera step d_c = -y_c is staged in LDS next to the camera tables of x and of the candidate.
#ifdef PTZ_EVAL_STAMPS  // probe builds only: where k_eval's time goes (block 0, thread 0; 100 MHz wall clock)
#define EV_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ev_t[i] = wall_clock64(); } while (0)
#define EV_STAMP_DECL long long ev_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define EV_STAMP_PRINT do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) printf("k_eval stamps (x10 ns): stage (FUSE: inputs asked for + table staged) %lld, step table (FUSE: candidates computed) %lld, ray prologue (FUSE: + barrier) %lld, pass1 %lld, mid %lld, pass2 %lld, reduce %lld\n", ev_t[1] - ev_t[0], ev_t[2] - ev_t[1], ev_t[3] - ev_t[2], ev_t[4] - ev_t[3], ev_t[5] - ev_t[4], ev_t[6] - ev_t[5], ev_t[7] - ev_t[6]); } while (0)
#else
#define EV_STAMP(i) do { } while (0)
#define EV_STAMP_DECL do { } while (0)
#define EV_STAMP_PRINT do { } while (0)
#endif
// FUSE (launch shapes of a few scenes, Dev::fuse_ctl): the launch also does what k_cam_update does before it -- every workgroup
// computes the scene's candidate cameras into its LDS tables, the first one stores them.  (The step is judged behind the
// speculative camera-side linearisation of the candidate, in k_lin_cam's tail: lm_step_wave.)
// LANES = 4 (a few scenes, FUSE): FOUR lanes per ray.  One rig alone keeps one wave per SIMD busy and a kernel lasts as long as its
// longest ray's observations one after the other (19 of them, twice); here a quad of lanes takes four observations at a time -- the
// functor, which is nearly all of the work, in parallel -- and then every lane of the quad adds the four observations' terms to its
// own copy of the ray's sums IN THE ORDER OF THE OBSERVATIONS (quad broadcasts): the same terms, added in the same order by the same
// expressions as the one-lane form, so a scene keeps its bits whatever form its batch runs in.
template <int Q> __device__ __forceinline__ double quad_bc(double v) { return dpp_mov<Q * 0x55>(v); }  // lane Q of the quad, in all four
template <int TYPE, bool SMALL, bool GTAB, bool FUSE = false, int LANES = 1>
__global__ __launch_bounds__(LANES > 1 ? 128 * LANES : (SMALL ? 256 : RAY_BLOCK)) void k_eval(Dev d)
{
  static_assert(LANES == 1 || (LANES == 4 && FUSE), "four lanes per ray: the form of a few scenes");
  const int RB = blockDim.x / LANES;         // rays per workgroup
  const int lp = threadIdx.x & (LANES - 1);  // this lane's place in its ray's group of lanes
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NC = Dims<TYPE>::NC, NW = Dims<TYPE>::NW, F = Dims<TYPE>::FACTOR;
  static_assert(!FUSE || (SMALL && !GTAB), "the folded camera update fills the LDS tables");
  if (FUSE && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) post_progress(d);  // (no control launch in this shape: see k_lm_post)
  const int sc = scene_of_slot(d, blockIdx.y);
  if (sc < 0) return;
  if (!d.active[sc]) return;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  if ((int)(blockIdx.x * RB) >= s.n_ray) return;
  EV_STAMP_DECL;
  EV_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int DCS = NC | 1;                // odd stride for the step table as well
  const double *tab, *ctab, *dct;
  double* scratch;
  float4* obsbuf;  // SMALL only: [8][blockDim.x]
  if constexpr (GTAB) {  // tables read where they lie (see k_lin_ray)
    tab = cur_camblk(d, st) + (size_t)s.cam_off * CBS;
    ctab = d.candblk + (size_t)s.cam_off * CDS;
    dct = d.dct + (size_t)s.cam_off * DCS;
    scratch = lds;
    obsbuf = reinterpret_cast<float4*>(scratch + 16);
  }
  else {
    double* tab0 = lds;                                    // [n_cam][CBS] (+ 4: alignment slack and spare slot of the flat copy)
    double* ctab0 = tab0 + ((s.n_cam * CBS + 5) & ~1);     // [n_cam][CDS] (+ 4)
    double* dct0 = ctab0 + ((s.n_cam * CDS + 5) & ~1);     // [n_cam][DCS] (+ 4) scaled camera step (k_cam_update)
    scratch = dct0 + ((s.n_cam * DCS + 5) & ~1);           // [16]
    obsbuf = reinterpret_cast<float4*>(scratch + 16);
    if constexpr (FUSE) {
      // The candidates of ALL the scene's cameras, by every workgroup: the inputs of a thread's first camera are asked for in
      // front of the table's staging loads (one memory round trip for both), the arithmetic runs behind them
      // (the ONE workgroup that also stores them -- with the SO(3) Jacobians, the full blocks, the step norms: three times the work
      //  of the others -- is the scene's LAST: rays are ordered by falling track length, so it is the one with the least to do after)
      const bool store_cand = (int)blockIdx.x == (s.n_ray - 1) / RB;
      CamIn<TYPE> in;
      cam_update_load<TYPE>(d, s, st, sc, min((int)threadIdx.x, s.n_cam - 1), in);
      tab = tab0 + stage_flat<16>(cur_camblk(d, st) + (size_t)s.cam_off * CBS, tab0, s.n_cam * CBS);
      EV_STAMP(1);
      if ((int)threadIdx.x < s.n_cam)
        cam_update_apply<TYPE>(d, s, st, sc, threadIdx.x, in, store_cand, ctab0 + threadIdx.x * CDS, dct0 + threadIdx.x * DCS);
      for (int i = threadIdx.x + blockDim.x; i < s.n_cam; i += blockDim.x) {
        cam_update_load<TYPE>(d, s, st, sc, i, in);
        cam_update_apply<TYPE>(d, s, st, sc, i, in, store_cand, ctab0 + i * CDS, dct0 + i * DCS);
      }
      ctab = ctab0;
      dct = dct0;
      EV_STAMP(2);
    }
    else {
      tab = tab0 + stage_flat<SMALL ? 16 : 8>(cur_camblk(d, st) + (size_t)s.cam_off * CBS, tab0, s.n_cam * CBS);
      ctab = ctab0 + stage_flat<SMALL ? 16 : 8>(d.candblk + (size_t)s.cam_off * CDS, ctab0, s.n_cam * CDS);
      dct = dct0 + stage_flat<SMALL ? 16 : 8>(d.dct + (size_t)s.cam_off * DCS, dct0, s.n_cam * DCS);
    }
    __syncthreads();
  }
  if constexpr (!FUSE) { EV_STAMP(1); EV_STAMP(2); }
  const int j = blockIdx.x * RB + (int)threadIdx.x / LANES;
  double mcc = 0, cost = 0, dn = 0, cn = 0;
  double lgm = 0;  // the candidate's share of the gradient max-norm (k_lm_pre, if the step is accepted; its |x|^2 share is cn)
  const int gj = s.ray_off + j;
  double sr[3] = {1, 1, 1}, w = 0, sw = 0, Xn[3] = {0, 0, 0};
  int a0 = 0, a1 = 0;
  if (j < s.n_ray) {
    const double* X = cur_ray(d, s, st) + (size_t)j * 3;
    const double Xr[3] = {X[0], X[1], X[2]};
    sr[0] = d.scale_r[(size_t)gj * 3]; sr[1] = d.scale_r[(size_t)gj * 3 + 1]; sr[2] = d.scale_r[(size_t)gj * 3 + 2];
    w = d.ray_w[gj];
    sw = sqrt(w);
    const int* rp = d.ray_ptr + s.ray_off + s.idx;
    a0 = rp[j]; a1 = rp[j + 1];
    // pass 1 (one linearisation per observation): with p_a = Jc_a d_c (camera part of J d),
    //   t  = g_r + sum_a Jr_a^T p_a                      -> y_r = E t, ray step d_r = -y_r
    //   s1 = sum_a p_a . (r_a + p_a / 2)
    // and, since J d = p_a + Jr_a d_r per observation, the ray's share of (J d)^T (r + J d / 2) is
    //   s1 + d_r . t + 1/2 d_r^T V d_r      (V = sum_a Jr_a^T Jr_a is the stored, undamped ray block)
    const double* grc = lin_gr(d, st.cur);
    double t0 = grc[(size_t)gj * 3], t1 = grc[(size_t)gj * 3 + 1], t2 = grc[(size_t)gj * 3 + 2];
    double s1 = 0;
    double Xu[3], inv_n;  // the functor's point for this ray, once for all of its observations
    ba_ray_point<F>(Xr, Xu, inv_n);
    EV_STAMP(3);
    // one observation's terms: what it adds to s1 and (before the ray's scales) to t
    auto terms1 = [&](float2 uv, int ci, double& X, double (&Y)[3]) {
      const double* cb = tab + ci * CBS;
      double res[2], pd[2], Jr[2][3];
      ba_step_dir_unit<F>(cb, Xu, inv_n, uv.x, uv.y, dct + ci * DCS, dct + ci * DCS + (NW - 3), res, pd, Jr);
      const double m0 = sw * pd[0], m1 = sw * pd[1];
      X = m0 * (res[0] * sw + m0 / 2.0) + m1 * (res[1] * sw + m1 / 2.0);
      Y[0] = Jr[0][0] * m0 + Jr[1][0] * m1;
      Y[1] = Jr[0][1] * m0 + Jr[1][1] * m1;
      Y[2] = Jr[0][2] * m0 + Jr[1][2] * m1;
    };
    auto add1 = [&](double X, double Y0, double Y1, double Y2) {
      s1 += X;
      t0 += sw * sr[0] * Y0;
      t1 += sw * sr[1] * Y1;
      t2 += sw * sr[2] * Y2;
    };
    if constexpr (LANES == 1) {
      for_each_obs<SMALL>(d, a0, a1, obsbuf, [&](float2 uv, int ci) {
        double X, Y[3];
        terms1(uv, ci, X, Y);
        add1(X, Y[0], Y[1], Y[2]);
      });
    }
    else {
      const int len = a1 - a0;
      int an = min(a0 + lp, a1 - 1);
      float2 uvn = d.obs_uv[an];
      int cin = d.obs_cam[an];
      for (int base = 0; base < len; base += LANES) {
        const float2 uv = uvn;
        const int ci = cin;
        an = min(a0 + base + LANES + lp, a1 - 1);   // the next round's record is on its way (past the end: the last one again, unused)
        uvn = d.obs_uv[an]; cin = d.obs_cam[an];
        double X, Y[3];
        terms1(uv, ci, X, Y);
        add1(quad_bc<0>(X), quad_bc<0>(Y[0]), quad_bc<0>(Y[1]), quad_bc<0>(Y[2]));
        if (base + 1 < len) add1(quad_bc<1>(X), quad_bc<1>(Y[0]), quad_bc<1>(Y[1]), quad_bc<1>(Y[2]));
        if (base + 2 < len) add1(quad_bc<2>(X), quad_bc<2>(Y[0]), quad_bc<2>(Y[1]), quad_bc<2>(Y[2]));
        if (base + 3 < len) add1(quad_bc<3>(X), quad_bc<3>(Y[0]), quad_bc<3>(Y[1]), quad_bc<3>(Y[2]));
      }
    }
    EV_STAMP(4);
    const double2 E01 = *e_piece(d, 0, gj), E23 = *e_piece(d, 1, gj), E45 = *e_piece(d, 2, gj);
    const double E[6] = {E01.x, E01.y, E23.x, E23.y, E45.x, E45.y};
    // step = -y_r (Ceres solves J y = r and negates)
    const double ds[3] = {-(E[0] * t0 + E[1] * t1 + E[3] * t2), -(E[1] * t0 + E[2] * t1 + E[4] * t2), -(E[3] * t0 + E[4] * t1 + E[5] * t2)};
    Xn[0] = Xr[0] + ds[0] * sr[0]; Xn[1] = Xr[1] + ds[1] * sr[1]; Xn[2] = Xr[2] + ds[2] * sr[2];
    double* xc = d.ray_x + (size_t)(st.cur ^ 1) * d.ray_stride + (size_t)gj * 3;
    if (lp == 0) { xc[0] = Xn[0]; xc[1] = Xn[1]; xc[2] = Xn[2]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { dn += (Xr[k] - Xn[k]) * (Xr[k] - Xn[k]); cn += Xn[k] * Xn[k]; }  // |x - x_c|^2, |x_c|^2 (k_lm_post)
    {
      const double* V = lin_V(d, st.cur) + (size_t)gj * 6;  // [v00 v10 v11 v20 v21 v22]
      const double q0 = V[0] * ds[0] + V[1] * ds[1] + V[3] * ds[2];
      const double q1 = V[1] * ds[0] + V[2] * ds[1] + V[4] * ds[2];
      const double q2 = V[3] * ds[0] + V[4] * ds[1] + V[5] * ds[2];
      mcc = s1 + (ds[0] * t0 + ds[1] * t1 + ds[2] * t2) + 0.5 * (ds[0] * q0 + ds[1] * q1 + ds[2] * q2);
    }
  }
  // the first pass's sums leave the registers now: one partial per wave of 64 rays (fixed butterfly), whatever the workgroup size
  if constexpr (LANES == 1) {
    mcc = wave_sum(mcc);
    dn = wave_sum(dn);
    cn = wave_sum(cn);
    if ((threadIdx.x & 63) == 0 && j < s.n_ray) {
      double* pp = d.partial + (size_t)(s.part_off + (j >> 6)) * 4;
      pp[0] = mcc; pp[2] = dn; pp[3] = cn;
      lin_partial(d, st.cur ^ 1)[(size_t)(s.part_off - s.idx + (j >> 6)) * 2 + 1] = cn;  // |x_c|^2 is the accepted point's |x|^2 (k_lm_pre)
    }
  }
  if (j < s.n_ray) {
    // pass 2: candidate cost, and -- for the price of the ray Jacobian on top of the residual -- the ray side of the candidate's
    // LINEARISATION (V, g_r, the ray's record for the camera pass, its share of the gradient norm and of |x|), into the other half
    // of the double buffers: if k_lm_post accepts the step that half becomes the current one and no k_lin_ray runs for it (it
    // would re-read these observations and camera blocks to compute exactly this); a rejected step leaves the current half alone.
    double Xcu[3], inv_nc;
    ba_ray_point<F>(Xn, Xcu, inv_nc);
    EV_STAMP(5);
    double Vc[6] = {0, 0, 0, 0, 0, 0}, gc3[3] = {0, 0, 0};
    // T[0]: the squared residual; T[1..6]: the observation's share of V; T[7..9]: of g_r
    auto terms2 = [&](float2 uv, int ci, double (&T)[10]) {
      double rc[2], Jr[2][3];
      ba_res_jr_unit<F>(ctab + ci * CDS, Xcu, inv_nc, uv.x, uv.y, rc, Jr);
      T[0] = rc[0] * rc[0] + rc[1] * rc[1];
      rc[0] *= sw; rc[1] *= sw;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double m = sw * sr[k]; Jr[0][k] *= m; Jr[1][k] *= m; }
      T[1] = Jr[0][0] * Jr[0][0] + Jr[1][0] * Jr[1][0];
      T[2] = Jr[0][1] * Jr[0][0] + Jr[1][1] * Jr[1][0];
      T[3] = Jr[0][1] * Jr[0][1] + Jr[1][1] * Jr[1][1];
      T[4] = Jr[0][2] * Jr[0][0] + Jr[1][2] * Jr[1][0];
      T[5] = Jr[0][2] * Jr[0][1] + Jr[1][2] * Jr[1][1];
      T[6] = Jr[0][2] * Jr[0][2] + Jr[1][2] * Jr[1][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) T[7 + k] = Jr[0][k] * rc[0] + Jr[1][k] * rc[1];
    };
    auto add2 = [&](const double (&T)[10]) {
      cost += 0.5 * (w * T[0]);
#pragma unroll
      for (int k = 0; k < 6; ++k) Vc[k] += T[1 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) gc3[k] += T[7 + k];
    };
    if constexpr (LANES == 1) {
      for_each_obs<SMALL>(d, a0, a1, obsbuf, [&](float2 uv, int ci) {
        double T[10];
        terms2(uv, ci, T);
        add2(T);
      });
    }
    else {
      const int len = a1 - a0;
      int an = min(a0 + lp, a1 - 1);
      float2 uvn = d.obs_uv[an];
      int cin = d.obs_cam[an];
      for (int base = 0; base < len; base += LANES) {
        const float2 uv = uvn;
        const int ci = cin;
        an = min(a0 + base + LANES + lp, a1 - 1);
        uvn = d.obs_uv[an]; cin = d.obs_cam[an];
        double T[10], B[10];
        terms2(uv, ci, T);
#pragma unroll
        for (int k = 0; k < 10; ++k) B[k] = quad_bc<0>(T[k]);
        add2(B);
        if (base + 1 < len) {
#pragma unroll
          for (int k = 0; k < 10; ++k) B[k] = quad_bc<1>(T[k]);
          add2(B);
        }
        if (base + 2 < len) {
#pragma unroll
          for (int k = 0; k < 10; ++k) B[k] = quad_bc<2>(T[k]);
          add2(B);
        }
        if (base + 3 < len) {
#pragma unroll
          for (int k = 0; k < 10; ++k) B[k] = quad_bc<3>(T[k]);
          add2(B);
        }
      }
    }
    if (lp == 0) {
      const int hc = st.cur ^ 1;
      double* Vo = lin_V(d, hc) + (size_t)gj * 6;
      double* go = lin_gr(d, hc) + (size_t)gj * 3;
#pragma unroll
      for (int k = 0; k < 6; ++k) Vo[k] = Vc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) go[k] = gc3[k];
      double* rr = lin_rayrec(d, hc) + (size_t)gj * 8;
      rr[0] = Xn[0]; rr[1] = Xn[1]; rr[2] = Xn[2]; rr[3] = sr[0]; rr[4] = sr[1]; rr[5] = sr[2]; rr[6] = w; rr[7] = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) lgm = fmax(lgm, fabs(gc3[k] / sr[k]));
    }
  }
  EV_STAMP(6);
  if constexpr (LANES == 1) {
    lgm = wave_max(lgm);
    cost = wave_sum(cost);
    if ((threadIdx.x & 63) == 0 && j < s.n_ray) {
      lin_partial(d, st.cur ^ 1)[(size_t)(s.part_off - s.idx + (j >> 6)) * 2] = lgm;
      d.partial[(size_t)(s.part_off + (j >> 6)) * 4 + 1] = cost;
    }
  }
  else {
    // the partials of a wave of 64 RAYS, as the one-lane form sums them (the same butterfly over the same 64 values): the rays' values
    // meet in LDS, ray r of the workgroup in lane r of the first waves
    double* red = reinterpret_cast<double*>(obsbuf);  // [5][RB] (the observation slots of the one-lane form, unused here: 128 bytes per ray)
    if (lp == 0) {
      const int r = threadIdx.x / LANES;
      red[r] = mcc; red[RB + r] = dn; red[2 * RB + r] = cn; red[3 * RB + r] = cost; red[4 * RB + r] = lgm;
    }
    __syncthreads();
    if ((int)threadIdx.x < RB) {
      const int r = threadIdx.x, jr = blockIdx.x * RB + r;
      double v0 = wave_sum(red[r]), v2 = wave_sum(red[RB + r]), v3 = wave_sum(red[2 * RB + r]), v1 = wave_sum(red[3 * RB + r]);
      const double vm = wave_max(red[4 * RB + r]);
      if ((r & 63) == 0 && jr < s.n_ray) {
        double* pp = d.partial + (size_t)(s.part_off + (jr >> 6)) * 4;
        pp[0] = v0; pp[1] = v1; pp[2] = v2; pp[3] = v3;
        double* pl = lin_partial(d, st.cur ^ 1) + (size_t)(s.part_off - s.idx + (jr >> 6)) * 2;
        pl[0] = vm; pl[1] = v3;
      }
    }
  }
  (void)scratch;
  EV_STAMP(7);
  EV_STAMP_PRINT;
}

// ---- eval_3d: annotation residuals' share of the model cost change and of the candidate cost -------------------
template <int TYPE>
__global__ __launch_bounds__(256) void k_eval_3d(Dev d)
{
  constexpr int CBS = Dims<TYPE>::CBS, CDS = Dims<TYPE>::CDS, CAMBLK = Dims<TYPE>::CAMBLK, CANDBLK = Dims<TYPE>::CANDBLK;
  (void)CBS; (void)CDS; (void)CAMBLK; (void)CANDBLK;
  constexpr int NC = Dims<TYPE>::NC;
  if (!Dims<TYPE>::HAS3D) return;
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0) return;
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
    for (int k = 0; k < CANDBLK; ++k) cb[k] = d.candblk[(size_t)gi * CDS + k];  // (with the displacement block: the whole block)
    const float2 uv = d.o3_uv[go];
    const double xyz[3] = {d.o3_xyz[(size_t)go * 3], d.o3_xyz[(size_t)go * 3 + 1], d.o3_xyz[(size_t)go * 3 + 2]};
    double rc[2], Jc[2][5 + Dims<TYPE>::F3 + 3 * Dims<TYPE>::DISP], Jt[2][6];
    reproj2d3d_eval<Dims<TYPE>::F3, false, Dims<TYPE>::DISP != 0>(cb, d.tlwcand + (size_t)s.idx * TLWBLK, xyz, uv.x, uv.y, rc, Jc, Jt);
    cost += 0.5 * (rc[0] * rc[0] + rc[1] * rc[1]);
  }
  mcc = block_sum(mcc, scratch);
  cost = block_sum(cost, scratch);
  if (threadIdx.x == 0) {
    double* pp = d.partial + (size_t)(s.part_off + s.n_wave) * 4;
    pp[0] = mcc; pp[1] = cost; pp[2] = 0.0; pp[3] = 0.0;
  }
}

// ---- lm_post: the body of TrustRegionMinimizer::Minimize after the step has been computed --------------
// One wave per scene (see lm_pre_wave).
template <int TYPE>
__device__ __forceinline__ void lm_post_wave(const Dev& d, int sc)
{
  const SceneDev s = d.scene[sc];
  LmState& st = d.lm[sc];
  const int lane = threadIdx.x & 63;
  // per-wave partials of k_eval in wave order, the cameras' {|x - x_c|^2, |x_c|^2} in camera order (lane-strided, then the butterfly)
  double mcc = 0, cost = 0, dn = 0, cn = 0;
  for (int c = lane; c < s.n_wave + Dims<TYPE>::HAS3D; c += 64) {
    const double2* pp = reinterpret_cast<const double2*>(d.partial + (size_t)(s.part_off + c) * 4);
    const double2 a = pp[0], b = pp[1];
    mcc += a.x; cost += a.y; dn += b.x; cn += b.y;  // the rays' |x - x_c|^2 and |x_c|^2 come from k_eval as well
  }
  for (int i = lane; i < s.n_cam; i += 64) {
    const double2 v = *reinterpret_cast<const double2*>(d.camstep + (size_t)(s.cam_off + i) * 2);
    dn += v.x; cn += v.y;
  }
  if (Dims<TYPE>::HAS3D && lane == 0 && s.n_o3 > 0) {
    const double* ta = d.tlw_x + (size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6;
    const double* tb_ = d.tlw_x + (size_t)(st.cur ^ 1) * d.tlw_stride + (size_t)s.idx * 6;
    for (int k = 0; k < 6; ++k) { dn += (ta[k] - tb_[k]) * (ta[k] - tb_[k]); cn += tb_[k] * tb_[k]; }
  }
  mcc = -wave_sum(mcc); cost = wave_sum(cost); dn = wave_sum(dn); cn = wave_sum(cn);
  if (lane != 0) return;
  const Opt& o = d.opt;
  ++st.num_linear_solves;
  st.reuse_diagonal = 1;  // LevenbergMarquardtStrategy::ComputeStep
  const bool solve_fail = d.ray_fail[sc] || d.chol.fail[sc];
  if (d.chol.fail[sc] & 2) ++st.chain_timeouts;
  const bool valid = !solve_fail && isfinite(mcc) && isfinite(dn) && mcc > 0.0;
  st.model_cost_change = mcc;
  st.it_cost = st.x_cost;
  if (!valid) {  // HandleInvalidStep
    ++st.num_consecutive_invalid;
    if (st.num_consecutive_invalid >= o.max_consecutive_invalid) { st.termination = PTZ_FAILURE; retire_scene(d, sc); return; }
    st.radius *= 0.5;  // StepIsInvalid
    st.reuse_diagonal = 0;
    return;
  }
  st.num_consecutive_invalid = 0;
  if (!isfinite(cost)) cost = 1.7976931348623157e308;
  st.candidate_cost = cost;
  st.cand_norm2 = cn;
  // ParameterToleranceReached
  if (sqrt(dn) <= o.parameter_tolerance * (st.x_norm + o.parameter_tolerance)) { st.termination = PTZ_CONVERGENCE; retire_scene(d, sc); return; }
  // FunctionToleranceReached
  const double cost_change = st.x_cost - cost;
  if (fabs(cost_change) <= o.function_tolerance * st.x_cost) { st.termination = PTZ_CONVERGENCE; retire_scene(d, sc); return; }
  const double rho = cost_change / mcc;  // TrustRegionStepEvaluator::StepQuality, monotonic steps
  if (rho > o.min_relative_decrease) {
    // HandleSuccessfulStep: x <- candidate; the Jacobian is re-evaluated by the kernels that follow
    st.cur ^= 1;
    st.need_linearize = 1;
    st.ray_lin_ready = 1;  // k_eval's second pass has left V, g_r, the ray records and the partials of this point in its half
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

// progress mark for the host's run-ahead throttle: one thread of the launch, also from passes that have nothing left to do
__device__ __forceinline__ void post_progress(const Dev& d)
{
  const int reached = ++d.grp_ctl[1];
  if (!d.debug_stall || reached <= d.debug_stall) __hip_atomic_store(&d.host_ctl[0], reached, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int TYPE>
__global__ __launch_bounds__(LM_THREADS) void k_lm_post(Dev d)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) post_progress(d);
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0 || !d.active[sc]) return;
  lm_post_wave<TYPE>(d, sc);
}

// ---- lm_step: one trust-region step judged AND the next iteration opened, at one point of the pass ------------------------------
// For scenes whose camera-side linearisation is speculative (Dev::spec_lin: no annotation residuals, no shared blocks): k_eval has
// left the candidate's cost and the ray side of its linearisation, k_lin_cam its camera side, all in the half that LmState::cur does
// not select -- so lm_post_wave's judgement and lm_pre_wave's bookkeeping for the next iteration need nothing in between, and an
// accepted step is a flip of `cur`.  Every input of both is asked for at once (one memory round trip), the state lives in
// registers in between and is stored once.  Same decisions in the same order as k_lm_post followed by k_lm_pre.
template <int TYPE>
__device__ __forceinline__ void lm_step_wave(const Dev& d, int sc)
{
  static_assert(!Dims<TYPE>::HAS3D, "annotation residuals keep k_lm_post / k_lm_pre");
  const SceneDev s = d.scene[sc];
  const LmState L0 = d.lm[sc];  // (uniform address: scalar loads, in flight with everything below)
  const int ray_fail = d.ray_fail[sc], chol_fail = d.chol.fail[sc];
  const int lane = threadIdx.x & 63;
  const int hc = L0.cur ^ 1;  // the candidate's half
  const double* costc_ = d.costc + (size_t)hc * d.lin_cams;
  const double* gmax_ = d.cam_gmax + (size_t)hc * d.lin_cams;
  double mcc = 0, cost = 0, dn = 0, cn = 0;   // the step: model cost change, candidate cost, |x - x_c|^2, |x_c|^2
  double c2 = 0, gm2 = 0;                     // the candidate's linearisation: cost, gradient max-norm
  for (int c = lane; c < s.n_wave; c += 64) {
    const double2* pp = reinterpret_cast<const double2*>(d.partial + (size_t)(s.part_off + c) * 4);
    const double2 a = pp[0], b = pp[1];
    mcc += a.x; cost += a.y; dn += b.x; cn += b.y;
    gm2 = fmax(gm2, lin_partial(d, hc)[(size_t)(s.part_off - s.idx + c) * 2]);
  }
  for (int i = lane; i < s.n_cam; i += 64) {
    const int gi = s.cam_off + i;
    const double2 v = *reinterpret_cast<const double2*>(d.camstep + (size_t)gi * 2);
    dn += v.x; cn += v.y;
    c2 += tail_load(&costc_[gi]);   // (k_lin_cam of this very launch may have written these two: sc1 loads, see tail_last_workgroup)
    gm2 = fmax(gm2, tail_load(&gmax_[gi]));
  }
  mcc = -wave_sum(mcc); cost = wave_sum(cost); dn = wave_sum(dn); cn = wave_sum(cn);
  c2 = wave_sum(c2); gm2 = wave_max(gm2);
  if (lane != 0) return;
  LmState L = L0;
  const Opt& o = d.opt;
  bool retire = false;
  // ---- the step (k_lm_post)
  ++L.num_linear_solves;
  L.reuse_diagonal = 1;  // LevenbergMarquardtStrategy::ComputeStep
  const bool solve_fail = ray_fail || chol_fail;
  if (chol_fail & 2) ++L.chain_timeouts;
  const bool valid = !solve_fail && isfinite(mcc) && isfinite(dn) && mcc > 0.0;
  L.model_cost_change = mcc;
  L.it_cost = L.x_cost;
  if (!valid) {  // HandleInvalidStep
    ++L.num_consecutive_invalid;
    if (L.num_consecutive_invalid >= o.max_consecutive_invalid) { L.termination = PTZ_FAILURE; retire = true; }
    else { L.radius *= 0.5; L.reuse_diagonal = 0; }  // StepIsInvalid
  }
  else {
    L.num_consecutive_invalid = 0;
    if (!isfinite(cost)) cost = 1.7976931348623157e308;
    L.candidate_cost = cost;
    L.cand_norm2 = cn;
    const double cost_change = L.x_cost - cost;
    if (sqrt(dn) <= o.parameter_tolerance * (L.x_norm + o.parameter_tolerance)) { L.termination = PTZ_CONVERGENCE; retire = true; }  // ParameterToleranceReached
    else if (fabs(cost_change) <= o.function_tolerance * L.x_cost) { L.termination = PTZ_CONVERGENCE; retire = true; }              // FunctionToleranceReached
    else {
      const double rho = cost_change / mcc;  // TrustRegionStepEvaluator::StepQuality, monotonic steps
      if (rho > o.min_relative_decrease) {
        // HandleSuccessfulStep: x <- candidate, whose linearisation is the other half's
        L.cur ^= 1;
        L.step_is_successful = 1;
        const double t = 2.0 * rho - 1.0;
        L.radius = L.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);  // StepAccepted
        L.radius = fmin(o.max_radius, L.radius);
        L.decrease_factor = 2.0;
        L.reuse_diagonal = 0;
      }
      else {
        // HandleUnsuccessfulStep / StepRejected
        L.it_cost = cost;
        L.radius = L.radius / L.decrease_factor;
        L.decrease_factor *= 2.0;
        L.reuse_diagonal = 1;
      }
    }
  }
  // ---- the next iteration (k_lm_pre)
  if (!retire) {
    if (L.step_is_successful) {  // the accepted point's cost, gradient max-norm, |x|
      L.x_cost = c2;
      L.it_cost = c2;
      L.grad_max = gm2;
      L.x_norm = sqrt(L.cand_norm2);
      ++L.num_jac_evals;
      ++L.num_successful;
    }
    else ++L.num_unsuccessful;
    L.need_linearize = 0;
    L.ray_lin_ready = 0;
    if (L.it_cost < L.final_cost) L.final_cost = L.it_cost;
    ++L.n_summaries;
    if (L.iteration >= o.max_num_iterations) { L.termination = PTZ_NO_CONVERGENCE; retire = true; }
    else if (L.step_is_successful && L.grad_max <= o.gradient_tolerance) { L.termination = PTZ_CONVERGENCE; retire = true; }
    else if (L.radius <= o.min_radius) { L.termination = PTZ_CONVERGENCE; retire = true; }
    else {
      ++L.iteration;
      ++L.num_lm_steps;
      L.step_is_successful = 0;
      d.ray_fail[sc] = 0;
    }
  }
  d.lm[sc] = L;
  if (retire) retire_scene(d, sc);
}

template <int TYPE>
__global__ __launch_bounds__(LM_THREADS) void k_lm_step(Dev d)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) post_progress(d);
  const int sc = scene_of_slot(d, blockIdx.x);
  if (sc < 0 || !d.active[sc]) return;
  if constexpr (!Dims<TYPE>::HAS3D) lm_step_wave<TYPE>(d, sc);
}

// The last workgroup of a scene to get here runs the scene's LM control in its first wave (Dev::fuse_ctl): every thread makes its
// stores visible device-wide, one thread counts the workgroup in; `expected` workgroups of the scene pass here per launch.
// Returns true in the first wave of the workgroup that closes the count (which has reset it for the next pass).
// Round 6: what the closing workgroup reads of the OTHER workgroups of this launch (k_lin_cam: the candidate's per-camera cost and
// gradient max-norm share) is stored write-through (tail_store: sc1) and read with L1-bypassing sc1 loads (tail_load), so the count
// needs neither an L2 write-back in front of it (__threadfence by 256 threads: ~3.5 us) nor an invalidate behind it (~1.7 us):
// every storing wave waits for its stores' acknowledgement, the barrier collects the waves, one lane counts the workgroup in.
// Everything else the launch stores is read by later launches only.
__device__ __forceinline__ void tail_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool tail_last_workgroup(int* cnt, int expected, int* lds_flag)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atomicAdd(cnt, 1);
    *lds_flag = prev == expected - 1;
    if (prev == expected - 1) *cnt = 0;  // (nobody else touches it before the next launch)
  }
  __syncthreads();
  if (!*lds_flag || threadIdx.x >= 64) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: the loads below stay behind the count)
  return true;
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
  d.tail_cnt[2 * sc] = 0; d.tail_cnt[2 * sc + 1] = 0;  // (a solve the watchdog gave up on may have left them mid-count)
}
// x <- initial state, Jacobi scales <- 1, LM state reset: everything a solve starts from, in one launch (was: five device-to-device
// copies, three fills and k_reset -- nine launches in front of every bundle adjustment of the incremental pipeline)
__global__ __launch_bounds__(256) void k_solve_init(Dev d, const double* __restrict__ cam0, const double* __restrict__ ray0, const double* __restrict__ dsp0,
                                                    const double* __restrict__ tlw0, size_t n_cam15, size_t n_ray3, size_t n_camnc)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_cam15) d.cam_x[i] = cam0[i];
  if (i < n_ray3) { d.ray_x[i] = ray0[i]; d.scale_r[i] = 1.0; }
  if (i < n_camnc) d.scale_c[i] = 1.0;
  if (d.dsp_x && i < d.dsp_stride) d.dsp_x[i] = dsp0[i];
  if (i < (size_t)6 * d.n_scene) {
    // both halves of the double buffer: without annotation residuals no kernel ever writes the candidate half, and the
    // accepted-step parity decides which half is read back
    const double t = tlw0[i];
    d.tlw_x[i] = t; d.tlw_x[d.tlw_stride + i] = t;
    d.scale_t[i] = 1.0;
  }
  if (i < (size_t)d.n_scene) {
    const int sc = (int)i;
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
    d.tail_cnt[2 * sc] = 0; d.tail_cnt[2 * sc + 1] = 0;  // (a solve the watchdog gave up on may have left them mid-count)
  }
}
// control words of one scene group (see Dev::grp_ctl); the host zeroes its pinned mirror itself before it enqueues anything
__global__ void k_ctl_reset(Dev d)
{
  if (threadIdx.x == 0) {
    d.grp_ctl[0] = d.n_scene; d.grp_ctl[1] = 0; d.grp_ctl[2] = d.n_scene;
    // tickets / done count of the one-launch factorisation: zero after every complete launch.  Left dirty by a launch that did
    // not finish (a fault, a solve the watchdog gave up on), they would send the next launch's workgroups off the triangle: start
    // clean, and move the generation on so that no flag of the unfinished launch can pass for one of the next
    int* ctl = d.chol.chain_ctl;
    if (ctl && (ctl[0] | ctl[1])) { ctl[0] = 0; ctl[1] = 0; ctl[2] += 2; }
  }
  for (int i = threadIdx.x; i < d.n_scene; i += blockDim.x) d.act[i] = i;
}

// The list of scenes still active, in scene order (one workgroup per scene group, after every k_lm_pre): what compacted
// launches index.  The count goes to the host as well, which sizes the grids of later passes from it.
__global__ __launch_bounds__(1024) void k_compact(Dev d)
{
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < d.n_scene; i0 += 1024) {
    const int i = i0 + tid;
    const bool on = i < d.n_scene && d.active[i] != 0;
    const unsigned long long m = __ballot(on);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < w; ++k) off += wsum[k];
    if (on) d.act[off + before] = i;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; base_s += t; }
    __syncthreads();
  }
  if (tid == 0) {
    d.grp_ctl[2] = base_s;
    __hip_atomic_store(&d.host_ctl[2], base_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// ---- gather_state: every scene's CURRENT state (LmState.cur selects the half) into contiguous arrays in the caller's order ----
__global__ __launch_bounds__(256) void k_gather_state(Dev d, const int* __restrict__ ray_perm, double* __restrict__ cam_out,
                                                       double* __restrict__ ray_out, double* __restrict__ tlw_out)
{
  const int sc = blockIdx.y;
  const SceneDev s = d.scene[sc];
  const LmState& st = d.lm[sc];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < s.n_cam * 15) cam_out[(size_t)s.cam_off * 15 + i] = cur_cam(d, s, st)[i];
  if (i < s.n_ray) {  // internal ray i is the caller's ray ray_perm[i]
    const double* x = cur_ray(d, s, st) + (size_t)i * 3;
    double* o = ray_out + ((size_t)s.ray_off + ray_perm[s.ray_off + i]) * 3;
    o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
  }
  if (i < 6) tlw_out[(size_t)s.idx * 6 + i] = d.tlw_x[(size_t)st.cur * d.tlw_stride + (size_t)s.idx * 6 + i];
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
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
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

// ---- structure of the Schur complement on the device (ptz_ba_batch_create) ------------------------------------------------
// What build_pairs does on the host after the observation arrays exist -- camera pairs, their entry lists in ray order, k_schur's
// runs -- rebuilt per CAMERA without any sort: the workgroup of camera ci marks, for every lower camera cj, the positions
// (in ci's observation list) of the observations whose track cj also sees, as a BITMAP in LDS (atomicOr: order-free).  A pair
// exists where a bitmap is non-empty, its entries are the set bits in ascending order -- which IS ray order, the camera-major
// list being in ray order -- an entry's place in its pair is a prefix popcount.  The arrays come out equal, word for word, to the
// host builder's (tests).  Two launches of the same kernel: COUNT leaves every camera's pair / entry / run counts, k_pair_scan
// turns them into offsets (one small read-back: the host sizes the arrays), WRITE fills them.
struct PairsDev {
  const SceneDev* scene;
  const int *obs_cam, *obs_ray, *ray_ptr, *cam_ptr, *cam_obs, *wpos;
  int max_runs;               // threads of a k_schur workgroup
  int* cam_cnt;               // [total_cam][3] pairs, entries, runs of every camera (COUNT)
  int* err;                   // [1] an image twice in one track
  const int* cam_off3;        // [total_cam][3] scene-local exclusive prefixes of cam_cnt (WRITE)
  const int* scene_tot;       // [n_scene][6] pairs, entries, runs, max pairs / entries / runs of one camera
  int *pci, *pcj, *pbrow, *pptr, *campair, *camrun, *prun;
  uint2* runs;
  unsigned* ent;
};

template <bool WRITE>
__global__ __launch_bounds__(256) void k_pairs(PairsDev a)
{
  const SceneDev s = a.scene[blockIdx.y];
  const int ci = blockIdx.x;
  if (ci >= s.n_cam) return;
  const int* cp = a.cam_ptr + s.cam_off + s.idx;
  const int* rp = a.ray_ptr + s.ray_off + s.idx;
  const int o0 = cp[ci], no = cp[ci + 1] - o0;
  const int W = (no + 31) / 32;
  extern __shared__ unsigned pl[];
  unsigned* bitmap = pl;                                   // [ci][W]
  int* cnt = reinterpret_cast<int*>(pl + (size_t)ci * W);  // [ci] entries of (ci, cj)
  int* pidx = cnt + ci;                                    // [ci] pair number of cj among ci's pairs, or -1
  int* pcjl = pidx + ci;                                   // [<= ci] cj of pair k
  int* poff = pcjl + ci;                                   // [<= ci] first entry of pair k, relative to the camera's first
  int* prl = poff + ci;                                    // [<= ci] first run of pair k, relative to the camera's first
  __shared__ int sh_npair, sh_nent, sh_L, sh_nrun, sh_pieces, sh_maxlen;
  for (int i = threadIdx.x; i < ci * W; i += 256) bitmap[i] = 0u;
  __syncthreads();
  for (int slot = threadIdx.x; slot < no; slot += 256) {
    const int g = a.cam_obs[o0 + slot];
    const int j = a.obs_ray[g];
    const int r0 = rp[j], r1 = rp[j + 1];
    for (int bb = r0; bb < r1; ++bb) {
      const int cj = a.obs_cam[bb];
      if (cj == ci && bb != g) *a.err = 1;  // an image appears once per track (tracks.cc:77)
      if (cj < ci) atomicOr(&bitmap[(size_t)cj * W + (slot >> 5)], 1u << (slot & 31));
    }
  }
  __syncthreads();
  for (int cj = threadIdx.x; cj < ci; cj += 256) {
    int c = 0;
    for (int w = 0; w < W; ++w) c += __popc(bitmap[(size_t)cj * W + w]);
    cnt[cj] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // pairs in ascending cj (a few hundred cameras at most: a serial sweep)
    int k = 0, e = 0, ml = 0;
    for (int cj = 0; cj < ci; ++cj) {
      const int c = cnt[cj];
      pidx[cj] = c > 0 ? k : -1;
      if (c > 0) { pcjl[k] = cj; poff[k] = e; e += c; ml = max(ml, c); ++k; }
    }
    sh_npair = k; sh_nent = e; sh_maxlen = ml;
    sh_L = max(1, min(ml, (e + a.max_runs - 1) / a.max_runs));  // lower bound of the run length (build_pairs)
  }
  __syncthreads();
  const int npair = sh_npair;
  // smallest run length L for which the camera's pairs fall into at most max_runs runs
  for (;;) {
    if (threadIdx.x == 0) sh_pieces = 0;
    __syncthreads();
    const int L = sh_L;
    int mine = 0;
    for (int k = threadIdx.x; k < npair; k += 256) mine += (cnt[pcjl[k]] + L - 1) / L;
    if (mine) atomicAdd(&sh_pieces, mine);
    __syncthreads();
    const bool ok = npair == 0 || sh_pieces <= a.max_runs || L >= sh_maxlen;
    __syncthreads();
    if (ok) break;
    if (threadIdx.x == 0) sh_L = L + 1;
    __syncthreads();
  }
  const int L = min(sh_L, 65535);
  if (threadIdx.x == 0) {
    int r = 0;
    for (int k = 0; k < npair; ++k) { prl[k] = r; r += (cnt[pcjl[k]] + L - 1) / L; }
    sh_nrun = r;
  }
  __syncthreads();
  const int gc = s.cam_off + ci;
  if (!WRITE) {
    if (threadIdx.x == 0) { a.cam_cnt[3 * gc] = npair; a.cam_cnt[3 * gc + 1] = sh_nent; a.cam_cnt[3 * gc + 2] = sh_nrun; }
    return;
  }
  const int p0 = a.cam_off3[3 * gc], e0 = a.cam_off3[3 * gc + 1], q0 = a.cam_off3[3 * gc + 2];  // scene-local firsts of this camera
  const int* tot = a.scene_tot + 6 * blockIdx.y;
  for (int k = threadIdx.x; k < npair; k += 256) {
    const int cj = pcjl[k], n = cnt[cj];
    a.pci[s.pair_off + p0 + k] = ci;
    a.pcj[s.pair_off + p0 + k] = cj;
    a.pbrow[s.pair_off + p0 + k] = cp[cj];
    a.pptr[s.pair_off + s.idx + p0 + k] = s.ent_off + e0 + poff[k];
    a.prun[s.pair_off + s.idx + p0 + k] = q0 + prl[k];
    int r = s.run_off + q0 + prl[k];
    for (int e = 0; e < n; e += L, ++r) a.runs[r] = make_uint2((unsigned)(s.ent_off + e0 + poff[k] + e), (unsigned)k | ((unsigned)min(L, n - e) << 16));
  }
  if (threadIdx.x == 0) {
    a.campair[s.cam_off + s.idx + ci] = p0;
    a.camrun[s.cam_off + s.idx + ci] = q0;
    if (ci == s.n_cam - 1) {
      a.campair[s.cam_off + s.idx + s.n_cam] = tot[0];
      a.camrun[s.cam_off + s.idx + s.n_cam] = tot[2];
      a.pptr[s.pair_off + s.idx + tot[0]] = s.ent_off + tot[1];
      a.prun[s.pair_off + s.idx + tot[0]] = tot[2];
    }
  }
  // entries: observation a of ci with every lower camera of its track; place in the pair = set bits below its own
  for (int slot = threadIdx.x; slot < no; slot += 256) {
    const int g = a.cam_obs[o0 + slot];
    const int j = a.obs_ray[g];
    const int r0 = rp[j], r1 = rp[j + 1];
    for (int bb = r0; bb < r1; ++bb) {
      const int cj = a.obs_cam[bb];
      if (cj >= ci) continue;
      const unsigned* bm = bitmap + (size_t)cj * W;
      int rank = __popc(bm[slot >> 5] & ((1u << (slot & 31)) - 1u));
      for (int w = 0; w < (slot >> 5); ++w) rank += __popc(bm[w]);
      a.ent[s.ent_off + e0 + poff[pidx[cj]] + rank] = (unsigned)slot | ((unsigned)(a.wpos[bb] - cp[cj]) << 16);
    }
  }
}

// per scene: exclusive prefixes of the cameras' counts and the scene's totals (one thread: a few hundred cameras)
__global__ __launch_bounds__(64) void k_pair_scan(const SceneDev* __restrict__ scene, int n_scene, const int* __restrict__ cam_cnt,
                                                  int* __restrict__ cam_off3, int* __restrict__ scene_tot)
{
  const int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc >= n_scene) return;
  const SceneDev s = scene[sc];
  int p = 0, e = 0, r = 0, mp = 0, me = 0, mr = 0;
  for (int c = 0; c < s.n_cam; ++c) {
    const int* cc = cam_cnt + 3 * (size_t)(s.cam_off + c);
    int* o = cam_off3 + 3 * (size_t)(s.cam_off + c);
    o[0] = p; o[1] = e; o[2] = r;
    p += cc[0]; e += cc[1]; r += cc[2];
    mp = max(mp, cc[0]); me = max(me, cc[1]); mr = max(mr, cc[2]);
  }
  int* t = scene_tot + 6 * sc;
  t[0] = p; t[1] = e; t[2] = r; t[3] = mp; t[4] = me; t[5] = mr;
}

}  // namespace

}  // namespace ptz
#endif  // PTZ_BA_KERNELS_H
